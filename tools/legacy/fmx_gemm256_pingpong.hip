// 256x256-tile implicit-GEMM convolution / linear for gfx950: the large-problem path of fmx_gemm_conv_f16.
//
// One workgroup = 8 waves = one 256(M: pixels) x 256(N: output channels) tile, K in steps of 64, the whole 128 KiB
// double buffer in LDS (1 workgroup per CU, 2 waves per SIMD).  The two waves of a SIMD belong to different GROUPS
// (g = wave / 4) that run the same instruction stream shifted by one barrier interval ("slot"): while one group
// issues its ds_reads (LOAD slot), the other group's 8 MFMAs (v_mfma_f32_32x32x16_f16, 256 cycles) own the matrix pipe.
//
//   wave (g, wc), wc = wave % 4, owns the 4 output quadrants (qi, qj) in {0,1}^2 of 64 x 32:
//       rows  m0 + qi*128 + g*64 + [0,64)         cols  n0 + qj*128 + wc*32 + [0,32)
//   so quadrant (qi, qj) needs only half-tile A_qi (128 activation rows) and half-tile B_qj (128 weight rows).
//   A K-tile is 4 PHASES, one quadrant each, in the order (0,0) (0,1) (1,1) (1,0): between consecutive phases only one
//   operand changes, so the LDS reads per K-tile are 8+4 | 4 | 8 | 0 ds_read_b128 (B_0 stays in registers).
//
// Measured on MI355X (tools/stamp_gemm.py, s_memtime per wave): the per-CU LDS-DMA path accepts one 1-KiB
// global_load_lds_dwordx4 every ~36 cycles at best (28 B/clk/CU), i.e. the 64 DMA instructions of a K-tile need ~2300
// cycles against 2048 cycles of MFMA -- the DMA issue stream is as critical as the matrix pipe;
// the stamps also show that a wave's DMA instruction does not issue while the OTHER wave of its SIMD streams MFMAs (it
// goes out right after that burst, ~60 cycles per piece), wherever it is placed in the LOAD slot; four placements were
// A/B-tested on the GPU (template parameter SCHED, tools/sched_gemm.py) and the simplest won by 3-8 %:
//
//   phase j:   LOAD slot:  ds_read what this quadrant needs | DMA j.0, j.1 | s_waitcnt vmcnt(6) | s_barrier
//              MFMA slot:  8 MFMA | source address of DMA (j+1).0 (32-bit offsets from a uniform base) | s_barrier
//   staged half-tile by phase:  P1: B_0(t+1)   P2: B_1(t+1)   P3: A_1(t+1)   P4: A_0(t+2)
//
// Why this is race-free (slots are numbered globally; group 0 loads in slot 2j and computes in slot 2j+1 for
// phase j, group 1 one slot later; every slot boundary is a workgroup barrier):
//   RAW  a half-tile is first read >= 4 phases after it was staged.  At the wait of phase j a wave has issued, newest
//        first, j.1, j.0, (j-1).1, (j-1).0, (j-2).1, (j-2).0, (j-3).1, ... so vmcnt(6) means "everything I staged in
//        phase <= j-3 has landed"; it is executed BEFORE the first barrier of the phase that precedes the read, so the
//        pieces of all 8 waves are in LDS and published by a barrier before anyone reads them.
//   WAR  a half-tile region is restaged >= 3 phases after the phase of its last ds_read (A_0: read P1(t), restaged
//        P4(t); all others 4 phases); the reads of phase j are complete in every wave before slot 2j+3 starts
//        (s_waitcnt lgkmcnt(0) follows the barrier that ends the load slot), and the restage is issued in slot >= 2j+6.
// The LDS image, swizzle and im2col-by-source-address gather are the same as in fmx_gemm.hip.
#include <stdlib.h>

#include "fmx_gemm_common.hpp"

namespace {

constexpr int BM = 256, BN = 256, BK = FMX_BK;
constexpr int HALF_BYTES = 128 * 128;        // one half-tile: 128 rows x 128 B
constexpr int STAGE_BYTES = 4 * HALF_BYTES;  // A_0 A_1 B_0 B_1
constexpr int LDS_BYTES = 2 * STAGE_BYTES;   // 128 KiB

template <int V>
struct IC { static constexpr int value = V; };

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct Cursor {  // K-tile being staged
  int t, ky, kx, cc;
};

struct Piece {  // one DMA instruction: per-lane byte offset, uniform byte offset, which descriptor (0 = a0, 1 = a1, 2 = weights)
  unsigned voff, soff;
  int which;
};

// ABL: timing-ablation bits for tools/ablate_gemm.py / stamp_gemm.py (results are garbage when != 0; only reachable when
// built with -DFMX_ABLATE):  1 = no vmcnt wait, 2 = no LDS-DMA, 4 = no ds_read, 8 = no MFMA, 128 = s_memtime stamps
// SCHED (A/B of the DMA issue points): 0 = piece 0 at the start of the LOAD slot + piece 1 inside the MFMA burst,
// 1 = both pieces at the end of the LOAD slot (after the ds_reads), 2 = same with s_setprio 3 around them,
// 3 = both pieces right after the MFMA burst
template <bool CONV, int ABL = 0, int SCHED = 1>
__global__ __launch_bounds__(512) void gemm256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, wc = wave & 3;
  const int hi = lane >> 5, li = lane & 31;

  // ---- tile id: XCD remap, then 8-row groups of tiles so that an XCD's 32 concurrent tiles form an 8 x 4 patch ----
  const int nwg = p.tiles_m * p.tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  int tm, tn;
  {
    constexpr int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int grp = wg / per_group;
    const int first_m = grp * GM;
    const int gsz = min(GM, p.tiles_m - first_m);
    const int in_g = wg - grp * per_group;
    tn = in_g / gsz;
    tm = first_m + (in_g - tn * gsz);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int Ctot = p.c0 + p.c1;

  // ---- staging geometry: per half-tile a wave issues 2 LDS-DMA instructions (e = 0,1), each 8 rows x 128 B;
  //      lane -> row (e*8 + wave)*8 + lane/8 of the half-tile, physical chunk lane&7 -------------------------------------
  const int r8 = lane >> 3;
  const int kc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);  // logical chunk (source side of the swizzle)
  const unsigned kcb = (unsigned)kc * 16u;                               // its byte offset inside the 128-B K row
  // Every DMA is a buffer_load_dwordx4 ... lds: wave-uniform descriptor + 32-bit per-lane byte offset (+ uniform soffset).
  // Measured (tools/ubench/dma_rate.hip): beside MFMA streams on the sibling waves a 64-bit-vaddr global_load_lds costs
  // 35 cycles per 1-KiB piece per CU, the 32-bit-offset forms 20 (= their rate with no MFMA at all).  Lanes with nothing
  // to load (conv zero padding, rows >= M / >= nout, the pipeline tail) use an offset beyond num_records: the hardware
  // bounds check writes zeros into LDS (tools/ubench/oob_probe.hip) -- no zero page, no 64-bit select.
  constexpr unsigned OOB = 0xC0000000u;  // the dispatcher guarantees every operand spans < OOB bytes
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a0), 0, p.a0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a1 ? p.a1 : p.a0), 0, p.a1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.wgt), 0, p.w_bytes, 0x00020000);
  int a_pix[4], a_yx[4];  // [q*2+e]: first pixel of the image (or pixel index for plain GEMM; -1 = none), packed (iy0, ix0)
  unsigned b_off[4];      // byte offset of (weight row, chunk), or OOB
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int row = (s >> 1) * 128 + ((s & 1) * 8 + wave) * 8 + r8;
    const int m = m0 + row;
    if (CONV) {
      const int per = p.oh * p.ow;
      const int mm = min(m, p.M - 1);
      const int img = mm / per;
      const int rem = mm - img * per;
      const int oy = rem / p.ow;
      const int ox = rem - oy * p.ow;
      a_pix[s] = img * p.h * p.w;
      const int iy0 = (m < p.M) ? oy * p.stride - p.pad : -20000;  // out-of-range rows fail every bounds check
      const int ix0 = ox * p.stride - p.pad;
      a_yx[s] = (iy0 << 16) | (ix0 & 0xffff);
    } else {
      a_pix[s] = (m < p.M) ? m : -1;
      a_yx[s] = 0;
    }
    const int nn = n0 + row;
    b_off[s] = (nn < p.nout) ? (unsigned)nn * (unsigned)p.ldw * 2u + kcb : OOB;
  }

  // A-operand piece (half q, e) for K-tile `c`
  auto a_piece = [&](int s, const Cursor& c) -> Piece {
    const bool second = c.cc >= p.c0;  // uniform
    const unsigned sstride = second ? (unsigned)p.s1 : (unsigned)p.s0;
    const unsigned coff = second ? (unsigned)(c.cc - p.c0) : (unsigned)c.cc;
    bool ok;
    unsigned pix;
    if (CONV) {
      int iy = (a_yx[s] >> 16) + c.ky;
      int ix = (int)(short)(a_yx[s] & 0xffff) + c.kx;
      if (p.up_h > 0) {
        ok = iy >= 0 && iy < p.up_h && ix >= 0 && ix < p.up_w;
        if (p.up_h == 2 * p.h && p.up_w == 2 * p.w) { iy >>= 1; ix >>= 1; }
        else { iy = ok ? (iy * p.h) / p.up_h : 0; ix = ok ? (ix * p.w) / p.up_w : 0; }
      } else {
        ok = iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
      }
      pix = (unsigned)(a_pix[s] + iy * p.w + ix);
    } else {
      ok = a_pix[s] >= 0;
      pix = (unsigned)a_pix[s];
    }
    ok = ok && c.t < p.kt;
    return Piece{ok ? pix * sstride * 2u + kcb : OOB, coff * 2u, second ? 1 : 0};
  };
  auto b_piece = [&](int s, const Cursor& c) -> Piece {
    return Piece{c.t < p.kt ? b_off[s] : OOB, (unsigned)c.t * (BK * 2u), 2};
  };
  auto advance = [&](Cursor& c) {
    c.t++;
    c.cc += BK;
    if (CONV && c.cc == Ctot) {
      c.cc = 0;
      if (++c.kx == p.kh) { c.kx = 0; ++c.ky; }
    }
  };
  // half-tile ids inside a stage: 0 = A_0, 1 = A_1, 2 = B_0, 3 = B_1;  HT = half-tile id, E = which of the wave's 2 pieces
  auto src_of = [&](auto HT, auto E, const Cursor& c) -> Piece {
    constexpr int ht = decltype(HT)::value, e = decltype(E)::value;
    if constexpr (ht < 2) return a_piece(ht * 2 + e, c);
    else return b_piece((ht - 2) * 2 + e, c);
  };
  auto dma = [&](const Piece& pc, int buf, auto HT, auto E) {
    constexpr int ht = decltype(HT)::value, e = decltype(E)::value;
    if (ABL & 2) return;
    auto* dst = (__attribute__((address_space(3))) void*)(smem + buf * STAGE_BYTES + ht * HALF_BYTES + (e * 8 + wave) * 1024);
    if constexpr (ht >= 2) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst, 16, pc.voff, pc.soff, 0, 0);
    } else {
      if (pc.which) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, dst, 16, pc.voff, pc.soff, 0, 0);  // uniform branch
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a0, dst, 16, pc.voff, pc.soff, 0, 0);
    }
  };

  f32x16 acc[2][2][2];  // [qi][qj][f]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][f][r] = 0.f;

  f16x8 af[2][4];     // activation fragments of the current qi: [f][kstep]   (MFMA "B" operand)
  f16x8 wf[2][4];     // weight fragments: [qj][kstep]                        (MFMA "A" operand)

  if (ABL & 4) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        af[f][ks] = *reinterpret_cast<const f16x8*>(smem + lane * 16 + f * 1024 + ks * 2048);
        wf[f][ks] = *reinterpret_cast<const f16x8*>(smem + lane * 16 + f * 1024 + ks * 2048 + 8192);
      }
  }
  auto read_a = [&](int buf, int qi) {
    if (ABL & 4) return;
    const char* base = smem + buf * STAGE_BYTES + qi * HALF_BYTES;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        af[f][ks] = *reinterpret_cast<const f16x8*>(base + lds_off(g * 64 + f * 32 + li, ks * 2 + hi));
  };
  auto read_b = [&](int buf, auto QJ) {
    constexpr int qj = decltype(QJ)::value;
    if (ABL & 4) return;
    const char* base = smem + buf * STAGE_BYTES + (2 + qj) * HALF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wf[qj][ks] = *reinterpret_cast<const f16x8*>(base + lds_off(wc * 32 + li, ks * 2 + hi));
  };
  // 8 MFMAs of quadrant (qi, qj); this wave's second DMA piece of the phase goes out after MFMA 2*wc
  auto mma = [&](auto QI, auto QJ, const Piece& src1, int buf, auto HT) {
    constexpr int qi = decltype(QI)::value, qj = decltype(QJ)::value;
    if (ABL & 8) {  // keep the fragments alive so the ds_reads are not dead code
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        asm volatile("" ::"v"(wf[qj][ks]));
#pragma unroll
        for (int f = 0; f < 2; ++f) asm volatile("" ::"v"(af[f][ks]));
      }
      dma(src1, buf, HT, IC<1>{});
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
        acc[qi][qj][f] = FMX_MFMA_32x32x16(wf[qj][ks], af[f][ks], acc[qi][qj][f]);
      if (SCHED == 0 && wc == ks) dma(src1, buf, HT, IC<1>{});
    }
    __builtin_amdgcn_s_setprio(0);
  };
  unsigned stamps[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) stamps[i] = 0;
  bool stamp_on = false;
  auto stamp = [&](auto IDX) {
    constexpr int idx = decltype(IDX)::value;
    if ((ABL & 128) && stamp_on) stamps[idx] = (unsigned)__builtin_amdgcn_s_memtime();
  };

  // ---- prologue: A_0(0) B_0(0) B_1(0) A_1(0) A_0(1) in flight; wait for the first two ------------------------------
  Cursor c1{0, 0, 0, 0};  // K-tile t+1 relative to the loop variable (starts as tile 0 for the prologue)
  dma(src_of(IC<0>{}, IC<0>{}, c1), 0, IC<0>{}, IC<0>{});
  dma(src_of(IC<0>{}, IC<1>{}, c1), 0, IC<0>{}, IC<1>{});
  dma(src_of(IC<2>{}, IC<0>{}, c1), 0, IC<2>{}, IC<0>{});
  dma(src_of(IC<2>{}, IC<1>{}, c1), 0, IC<2>{}, IC<1>{});
  dma(src_of(IC<3>{}, IC<0>{}, c1), 0, IC<3>{}, IC<0>{});
  dma(src_of(IC<3>{}, IC<1>{}, c1), 0, IC<3>{}, IC<1>{});
  dma(src_of(IC<1>{}, IC<0>{}, c1), 0, IC<1>{}, IC<0>{});
  dma(src_of(IC<1>{}, IC<1>{}, c1), 0, IC<1>{}, IC<1>{});
  advance(c1);            // c1 = tile 1
  dma(src_of(IC<0>{}, IC<0>{}, c1), 1, IC<0>{}, IC<0>{});
  dma(src_of(IC<0>{}, IC<1>{}, c1), 1, IC<0>{}, IC<1>{});
  Piece src0 = src_of(IC<2>{}, IC<0>{}, c1);  // first piece of P1's half-tile B_0(1)
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one slot behind group 0

  unsigned long long clk0 = 0, rt0 = 0;
  if (ABL & 128) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  for (int t = 0; t < p.kt; ++t) {
    const int buf = t & 1;
    Cursor c2 = c1;
    advance(c2);  // tile t+2
    if (ABL & 128) stamp_on = (blockIdx.x == 8 && t == 8);
    // PH: phase index; READS: this quadrant's ds_reads; (QI,QJ): quadrant; (HT, SBUF, CUR): half-tile staged in this phase;
    // (NHT, NCUR): half-tile staged in the NEXT phase (its first source address is computed in this phase's MFMA shadow)
#define FMX_PHASE(PH, READS, QI, QJ, HT, SBUF, CUR, NHT, NCUR)                 \
    stamp(IC<PH * 6 + 0>{});                                                   \
    if (SCHED == 0) dma(src0, SBUF, IC<HT>{}, IC<0>{});                        \
    stamp(IC<PH * 6 + 1>{});                                                   \
    READS;                                                                     \
    if (SCHED == 1 || SCHED == 2) {                                            \
      if (SCHED == 2) __builtin_amdgcn_s_setprio(3);                           \
      dma(src0, SBUF, IC<HT>{}, IC<0>{});                                      \
      dma(src_of(IC<HT>{}, IC<1>{}, CUR), SBUF, IC<HT>{}, IC<1>{});            \
      if (SCHED == 2) __builtin_amdgcn_s_setprio(0);                           \
    }                                                                          \
    stamp(IC<PH * 6 + 2>{});                                                   \
    if (!(ABL & 1)) {                                                          \
      if (SCHED == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");        \
      else if (SCHED == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   \
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                   \
    }                                                                          \
    __builtin_amdgcn_s_barrier();                                              \
    __builtin_amdgcn_sched_barrier(0);                                         \
    stamp(IC<PH * 6 + 3>{});                                                   \
    {                                                                          \
      const Piece src1 = src_of(IC<HT>{}, IC<1>{}, CUR);                       \
      mma(IC<QI>{}, IC<QJ>{}, src1, SBUF, IC<HT>{});                           \
      if (SCHED == 3) {                                                        \
        dma(src0, SBUF, IC<HT>{}, IC<0>{});                                    \
        dma(src1, SBUF, IC<HT>{}, IC<1>{});                                    \
      }                                                                        \
      src0 = src_of(IC<NHT>{}, IC<0>{}, NCUR);                                 \
    }                                                                          \
    stamp(IC<PH * 6 + 4>{});                                                   \
    __builtin_amdgcn_sched_barrier(0);                                         \
    __builtin_amdgcn_s_barrier();                                              \
    __builtin_amdgcn_sched_barrier(0);                                         \
    stamp(IC<PH * 6 + 5>{});
    // P1 (0,0): stage B_0(t+1);  P2 (0,1): B_1(t+1);  P3 (1,1): A_1(t+1);  P4 (1,0): A_0(t+2), B_0 still in registers
    FMX_PHASE(0, read_a(buf, 0); read_b(buf, IC<0>{}), 0, 0, 2, buf ^ 1, c1, 3, c1)
    FMX_PHASE(1, read_b(buf, IC<1>{}), 0, 1, 3, buf ^ 1, c1, 1, c1)
    FMX_PHASE(2, read_a(buf, 1), 1, 1, 1, buf ^ 1, c1, 0, c2)
    FMX_PHASE(3, (void)0, 1, 0, 0, buf, c2, 2, c2)
#undef FMX_PHASE
    c1 = c2;
  }
  if (g == 0) __builtin_amdgcn_s_barrier();  // balance group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's dummy DMA must land before the LDS is released

  if (ABL & 128) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int f = 0; f < 2; ++f) asm volatile("" ::"v"(acc[a][b][f]));
    if (blockIdx.x == 8 && lane == 0) {
      unsigned* dbg = reinterpret_cast<unsigned*>(p.out);
#pragma unroll
      for (int i = 0; i < 24; ++i) dbg[wave * 24 + i] = stamps[i];
      if (wave == 0) {  // whole K loop: shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) -> effective clock
        dbg[192] = (unsigned)(__builtin_amdgcn_s_memtime() - clk0);
        dbg[193] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
        dbg[194] = (unsigned)p.kt;
      }
    }
    return;
  }
  // ---- epilogue.  The MFMA leaves lane (li, hi) with 4 runs (q4) of 4 consecutive channels  q4*8 + hi*4 + [0,4)  of
  //      pixel li.  One v_permlane32_swap per accumulator pair (q4 even <-> q4 odd) turns that into 2 runs of 8
  //      consecutive channels  (2*pp + hi)*8 + [0,8)  per fragment, so every epilogue load / store is 16 bytes per lane:
  //      half the memory instructions of the 8-byte form (the tile's store tail is issue-bound: 12 us per round of
  //      256 tiles at 32 dwordx2 stores per lane, measured with the mainloop ablated).  Only the branch-free
  //      FastEpilogue form is implemented here (the dispatcher checks FastEpilogue::eligible8). ---------------------------
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[a][b][f][pp * 8 + r], y = acc[a][b][f][pp * 8 + 4 + r];
            swap_halfwaves(x, y);
            acc[a][b][f][pp * 8 + r] = x;
            acc[a][b][f][pp * 8 + 4 + r] = y;
          }
  const FastEpilogue ep(p);
  const bool geglu = p.act == FMX_ACT_GEGLU;
  if (!geglu) {
    // bias for this lane's 4 column groups, loaded once
    int nbs[2][2];
    bool nok[2][2];
    f16x8 bb[2][2];
#pragma unroll
    for (int qj = 0; qj < 2; ++qj)
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int nb = n0 + qj * 128 + wc * 32 + (2 * pp + hi) * 8;
        nok[qj][pp] = nb < ep.nout;
        nbs[qj][pp] = nok[qj][pp] ? nb : 0;
        bb[qj][pp] = ep.bias8(nbs[qj][pp]);
      }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int m = m0 + qi * 128 + g * 64 + f * 32 + li;
        const bool mok = m < p.M;
        const int mc = mok ? m : p.M - 1;
        const int img = mc / ep.per_img;
        f16x8 rv[2][2], rs[2][2], gt[2][2];
#pragma unroll
        for (int qj = 0; qj < 2; ++qj)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            rv[qj][pp] = ep.rv8(img, nbs[qj][pp]);
            rs[qj][pp] = ep.res8(mc, nbs[qj][pp]);
            gt[qj][pp] = ep.gate8(img, nbs[qj][pp]);
          }
#pragma unroll
        for (int qj = 0; qj < 2; ++qj)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
              v[r] = ep.act_gate(acc[qi][qj][f][pp * 8 + r] * ep.alpha + (float)bb[qj][pp][r] + (float)rv[qj][pp][r], (float)gt[qj][pp][r]) +
                     (float)rs[qj][pp][r];
            if (mok && nok[qj][pp]) ep.store8(m, nbs[qj][pp], v);
          }
      }
  } else {
    // weight rows are interleaved [16 value | 16 gate]: after the swap, registers 0..7 of a fragment are the values of
    // columns hi*8 + [0,8) and registers 8..15 their gates
    int nbs[2];
    bool nok[2];
    f16x8 bv[2], bg[2];
#pragma unroll
    for (int qj = 0; qj < 2; ++qj) {
      const int nb = n0 + qj * 128 + wc * 32 + hi * 8;
      nok[qj] = nb < ep.nout;
      nbs[qj] = nok[qj] ? nb : 0;
      bv[qj] = ep.bias8(nbs[qj]);
      bg[qj] = ep.bias8(nbs[qj] + 16);
    }
#pragma unroll
    for (int qi = 0; qi < 2; ++qi)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int m = m0 + qi * 128 + g * 64 + f * 32 + li;
        const bool mok = m < p.M;
        const int mc = mok ? m : p.M - 1;
        const int img = mc / ep.per_img;
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
          const int nb = nbs[qj];
          const int col = ((n0 + qj * 128 + wc * 32) >> 1) + hi * 8;
          const int colc = nok[qj] ? col : 0;
          const f16x8 rvv = ep.rv8(img, nb), rvg = ep.rv8(img, nb + 16), rs = ep.res8(mc, colc);
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const float val = acc[qi][qj][f][r] * ep.alpha + (float)bv[qj][r] + (float)rvv[r];
            const float gate = acc[qi][qj][f][8 + r] * ep.alpha + (float)bg[qj][r] + (float)rvg[r];
            v[r] = val * gelu_erf_f(gate) + (float)rs[r];
          }
          if (mok && nok[qj]) ep.store8(m, col, v);
        }
      }
  }
}

}  // namespace

int fmx_launch_gemm256(const GemmParams& p, bool conv, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  GemmParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.nout + BN - 1) / BN;
  const int grid = q.tiles_m * q.tiles_n;
#ifdef FMX_ABLATE
  {
    static int abl = -1, sched = -1;
    if (abl < 0) { const char* e = getenv("FMX_ABL"); abl = e ? atoi(e) : 0; const char* f = getenv("FMX_SCHED"); sched = f ? atoi(f) : -1; }
    if (!abl && sched >= 0 && !conv) {
#define FMX_SCHED_CASE(S)                                                                                                   \
  case S: {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false, 0, S>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL((gemm256_kernel<false, 0, S>), dim3(grid), dim3(512), LDS_BYTES, st, q);                              \
    return FMX_OK;                                                                                                         \
  }
      switch (sched) { FMX_SCHED_CASE(0) FMX_SCHED_CASE(2) FMX_SCHED_CASE(3) default: break; }
    }
    if (abl && !conv) {
#define FMX_ABL_CASE(A)                                                                                                     \
  case A: {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<false, A>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
    hipLaunchKernelGGL((gemm256_kernel<false, A>), dim3(grid), dim3(512), LDS_BYTES, st, q);                                 \
    return FMX_OK;                                                                                                         \
  }
      switch (abl) {
        FMX_ABL_CASE(1) FMX_ABL_CASE(2) FMX_ABL_CASE(4) FMX_ABL_CASE(6) FMX_ABL_CASE(8) FMX_ABL_CASE(10) FMX_ABL_CASE(12) FMX_ABL_CASE(14)
        FMX_ABL_CASE(128) default: break;
      }
    }
  }
#endif
  if (conv) hipLaunchKernelGGL(gemm256_kernel<true>, dim3(grid), dim3(512), LDS_BYTES, st, q);
  else hipLaunchKernelGGL(gemm256_kernel<false>, dim3(grid), dim3(512), LDS_BYTES, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (256x256)");
  return FMX_OK;
}
