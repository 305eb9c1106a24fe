// 256 x BN-tile (BN = 256 or 320), 320 x 256 and 512 x 128 implicit-GEMM convolution / linear for gfx950, software-pipelined ("gemm256p").
//
// The first-generation 8-wave kernel (in the history up to round 3; no longer part of the library) alternated two wave
// groups through 8 barrier-separated slots per K-tile (one group loads while the other owns the matrix pipe); its s_memtime
// stamps (profiles/r02d_*) show every slot costing max(load, 8 MFMA) + barrier skew, 3000+ cycles per K-tile against 2048 cycles
// of MFMA.  Here every wave runs ONE in-order stream with ONE barrier per K-tile:
//
//   BN = 256: waves 2 (M) x 4 (N), wave tile 128 x 64  = 4 x 2 accumulator blocks of 32 x 32,  8 MFMA / 6 ds_read per k-step
//   BN = 320: waves 4 (M) x 2 (N), wave tile  64 x 160 = 2 x 5 blocks,                        10 MFMA / 7 ds_read per k-step
//   BM = 320 (x 256): waves 2 x 4, wave tile 160 x 64 = 5 x 2 blocks -- the same thing for the operand-swapped V^T GEMM
//   512 x 128: waves 4 (M) x 2 (N), wave tile 128 x 64 = 4 x 2 blocks (the 256 x 256 kernel's inner loop) -- for layers with 128 output
//   channels (the VAE decoder's 1024^2 level), where a 256-wide tile would compute 50 % padding; 8 + 2 LDS-DMA pieces per wave and K-tile,
//   so the implicit-GEMM address arithmetic is the per-tap bit mask form (see FASTADDR below)
//   (v_mfma_f32_32x32x16_f16, weights as MFMA-A, activations as MFMA-B).  320 is the native width of the SD family: every
//   channel count is 320 k, so N = 320 / 640 / 1280 tile with no padding columns and (M, N) = (16384, 1280) is exactly
//   256 tiles = one round of the 256 CUs (256-wide tiles: 320 tiles = 1.25 rounds), with 10 % fewer LDS-DMA bytes per FLOP.
//
//   A K-tile (BK = 64) is 4 k-steps; the fragments of k-step s+1 are read from LDS into the other half of a register
//   double buffer while the MFMAs of k-step s issue, interleaved one load per MFMA (sched_group_barrier): a lone load
//   issues in the 32-cycle shadow of the running MFMA, the same loads issued back to back starve the matrix pipe
//   (tools/ubench/lds_mix.hip).  The second wave of the SIMD fills whatever gaps remain.
//
//   k-step 3 of K-tile t:   s_waitcnt lgkmcnt(0) vmcnt(0) | s_barrier | ds_read fragments (t+1, 0) from stage (t+1)&1
//                           | MFMAs (t, 3) | first 3 pieces of tile t+2 -> stage t&1
//   Why one barrier is enough: at that point every wave has finished its LAST reads of stage t&1 (the fragments of k-step 3
//   were read during k-step 2 and lgkmcnt(0) has retired them), so the stage may be overwritten (WAR); and every wave has
//   waited for its own pieces of tile t+1 (issued during k-steps (t-1,3), (t,0), (t,1)), so after the barrier all of tile
//   t+1 is visible in LDS (RAW).
#include <stdlib.h>

#include <utility>

#include "fmx_gemm_epi.hpp"

namespace {

constexpr int BK = FMX_BK;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct Cursor {  // K-tile being staged
  int t, ky, kx, cc;
};

struct Piece {  // one DMA instruction: per-lane byte offset, uniform byte offset, second source (a1) or not
  unsigned voff, soff;
  bool second;
};

template <int BM, int BN>
struct Geo {
  static constexpr int WN = (BN == 320 || BN == 128) ? 2 : 4;   // waves along N
  static constexpr int WM = 8 / WN;                     // waves along M
  static constexpr int MI = BM / (WM * 32);             // 32-row blocks per wave along M
  static constexpr int NJ = BN / (WN * 32);             // 32-col blocks per wave along N
  static constexpr int NPA = BM / 64, NPB = BN / 64;    // LDS-DMA pieces per wave per K-tile (8 rows x 128 B each)
  static constexpr int NP = NPA + NPB;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  // per-wave transpose buffer of the epilogue: one 32-row block row of NJ*32 fp32, or (GEGLU) the wave's whole MI*32 x NJ*16 sub-tile
  static constexpr int WAVE_EPI_BYTES = 32 * NJ * 128 > MI * 32 * NJ * 64 ? 32 * NJ * 128 : MI * 32 * NJ * 64;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES > 8 * WAVE_EPI_BYTES ? 2 * STAGE_BYTES : 8 * WAVE_EPI_BYTES;  // 128 / 160 KiB
};

// SCHED: where the NP LDS-DMA pieces of a K-tile are issued (A/B knob FMX_GEMM_SCHED, tools/bench_kernels.py gemmsched):
//   0: 3 behind the barrier (k-step 3 of the tile before), 3 in k-step 0, the rest in k-step 1 -- spread thin beside the MFMAs
//   1: 5 / 4 / 0 -- everything a k-step earlier, 2 k-steps (~1.5 k cycles) between the last issue and the wait
//   2: all NP behind the barrier
// MF: the MFMA shape of the K loop.  32 = v_mfma_f32_32x32x16 (rounds 1-2).  16 = v_mfma_f32_16x16x32 (round 3): the same FLOPs per k from four
//   times as many accumulator blocks a quarter the size.  Why: the K loop is power-bound, and on this part a stream of 16x16x32
//   MFMAs on random operands sustains 11 % more FLOP/s at the power limit than the same tile as 32x32x16 (tools/ubench/mfma_power.hip,
//   profiles/r08q: the GEMM's own 64 x 160 wave-tile sequence from registers, 1 913 vs 1 714 TFLOP/s).  A k-step is then 32 of K (two per
//   K-tile) and 40 MFMAs per wave; the weight fragments are SINGLE-buffered -- fragment j is re-read for the next k-step right behind its four
//   MFMAs, 36 MFMAs before its next use -- and only the four activation fragments are double-buffered, so the register count stays where it was.
// FA: the per-tap bit-mask form of the implicit-GEMM addresses (FASTADDR below) on tiles other than 512 x 128 -- single-source convolutions without
//   upsample-on-load (every 3x3 convolution of the UNet except the decoder's upsamplers): the general form's ~15 vector instructions per
//   activation piece (two sources, nearest-resize, bounds) become 3, in a K loop that otherwise issues ~10 instructions per MFMA (round 3).
//   FA = 3 (round 6): FA = 1 with scattered output rows (GemmParams::scat_ow), the four phase convolutions of fmx_conv3x3_up2x.
//   FA = 2 (round 4): the same for the x2 NEAREST UPSAMPLE ON LOAD of the UNet's / VAE decoder's Upsample convolutions (3x3, stride 1, pad 1 on the
//   upsampled grid): source row of tap ky is  (iy0 >> 1) + {0, iy0 & 1, 1}[ky]  (iy0 = oy - 1 on the upsampled grid), likewise for columns, so the two
//   parity bits ride in the mask word (bits 9, 10) and a piece costs 7 vector instructions instead of ~15.
//   (The same form for the TWO-SOURCE convolutions of the decoder -- pixel index x the K-tile's source stride -- was measured in round 4 and changes nothing:
//   102.82 vs 102.83 ms per step, every two-source shape within 1 %: those launches have K >= 5760 and hide the address arithmetic.  Not kept.)
// XA (round 5): cross-attention against the text context as the EPILOGUE of the query projection (fmx.h xa_*), see the block behind the K loop.
template <bool CONV, int BM, int BN, bool STATS = false, int SCHED = 0, int LN = 0, int MF = 32, int FA = 0, bool XA = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(const GemmParams p) {
  using G = Geo<BM, BN>;
  constexpr int MI = G::MI, NJ = G::NJ, NPA = G::NPA, NPB = G::NPB, NP = G::NP;
  constexpr int BR = MF;                                   // rows / columns of one accumulator block
  constexpr int MIB = MI * 32 / BR, NJB = NJ * 32 / BR;    // accumulator blocks per wave along M / N
  constexpr int WROWS = MI * 32;                           // output rows of a wave
  constexpr int STAGE_BYTES = G::STAGE_BYTES, A_BYTES = G::A_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;

  // ---- tile id: XCD remap, then 8-row groups of tiles so that an XCD's 32 concurrent tiles form an 8 x 4 patch ----
  // PERSISTENT over the launch's tiles (round 3): workgroup b owns tiles b, b + grid, b + 2 grid, ... (grid = one workgroup per CU, a
  // multiple of 8, so a workgroup's tiles stay on its XCD).  A workgroup dispatched per tile cannot start before its predecessor on the CU has
  // drained its output stores and released the LDS; here the stores of tile t are still in flight while the LDS-DMA prologue of tile t + 1
  // is issued (the first K-tile's vmcnt(0) then waits for both), and the dispatch / descriptor set-up of later rounds is gone.
  // XT (round 3, linear GEMMs on the 256 x 320 tile with the 16x16x32 loop): the K loop's LDS-DMA pipeline runs ACROSS output tiles.  The pieces a
  // K loop issues for "K-tile kt" and beyond were zero fills; with a next tile in this workgroup's walk, K-tile kt IS that tile's K-tile 0 (its
  // addresses swapped in two K-tiles before the end), so it lands under the last K-tile's MFMAs and is complete at that K-tile's barrier: the next
  // tile's prologue (issue 14 pieces, wait for the first 9: ~3 us with the matrix pipes idle) shrinks to issuing the first pieces
  // of its K-tile 1.  The stage that holds the prefetched K-tile is whichever is next in the alternation ((kt + s0) & 1, so a tile starts on
  // stage s0 = 0 or 1), and the epilogue's transpose slices (16 rows x NJ*32 fp32 = XSL bytes per wave with this loop; GEGLU in two halves of 32
  // rows) are laid around it: above stage 0, or below stage 1 with the eighth slice above it.
  constexpr bool XT = !XA && !CONV && !STATS && MF == 16 && BM == 256 && BN == 320;   // (XA: the epilogue needs the whole LDS)
  constexpr int XSL = NJ * 2048;
  static_assert(!XT || (G::LDS_BYTES - 8 * XSL >= STAGE_BYTES && 7 * XSL <= STAGE_BYTES && 2 * STAGE_BYTES + XSL <= G::LDS_BYTES), "slices fit around one stage");
  int xs0 = 0;          // stage of this tile's K-tile 0
  bool xpre = false;    // ... which is already there (fetched under the previous tile's K loop)
  const int nwg = p.tiles_m * p.tiles_n;
  auto tile_origin = [&](int l, int& tm_, int& tn_) {
    const int wg_ = xcd_remap(l, nwg);
    constexpr int GM = 8;
    const int per_group = GM * p.tiles_n;
    const int grp = wg_ / per_group;
    const int first_m = grp * GM;
    const int gsz = min(GM, p.tiles_m - first_m);
    const int in_g = wg_ - grp * per_group;
    tn_ = in_g / gsz;
    tm_ = first_m + (in_g - tn_ * gsz);
  };
  for (int lid = blockIdx.x; lid < nwg; lid += gridDim.x) {
  const bool has_next = XT && p.xtile && p.kt >= 2 && lid + (int)gridDim.x < nwg;   // uniform
  const int s0 = XT ? xs0 : 0;
  const bool pre = XT && xpre;
  const int kt_live = p.kt + (has_next ? 1 : 0);
  // the lane id passes through an opaque copy once per tile: every per-lane address below (LDS-DMA offsets, fragment and epilogue LDS offsets)
  // is the same for all tiles, and hoisted out of this loop they would ride through the K loop in ~45 registers the 160-accumulator tile
  // does not have (spilled to scratch when first tried)
  int lane_it = tid & 63;
  asm volatile("" : "+v"(lane_it));
  const int lane = lane_it;
  const int hi = lane >> 5, li = lane & 31;
  const int wg = xcd_remap(lid, nwg);
  int tm, tn;
  tile_origin(lid, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int Ctot = p.c0 + p.c1;

  // ---- staging geometry: piece s of a wave covers the 8 rows  s*64 + wave*8 + [0,8)  of its operand (A: s < NPA,
  //      B: s < NPB), lane -> row lane/8, physical 16-byte chunk lane&7; LDS destination (s*8 + wave) * 1 KiB ---------------
  const int r8 = lane >> 3;
  const int kc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);  // logical chunk (source side of the swizzle)
  const unsigned kcb = (unsigned)kc * 16u;
  // buffer_load_dwordx4 ... lds: uniform descriptor + 32-bit byte offset; offsets >= OOB read as zeros (tools/ubench/oob_probe.hip)
  constexpr unsigned OOB = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a0), 0, p.a0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a1 ? p.a1 : p.a0), 0, p.a1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.wgt), 0, p.w_bytes, 0x00020000);
  // FASTADDR (the 512-row tile: 8 activation pieces per wave and K-tile): the host sends only single-source convolutions without
  // upsample-on-load here, for which a piece's pixel offset is  base(lane) + (ky * w + kx) * stride  with a UNIFORM second term, and its
  // validity one bit of a per-lane 9-bit mask computed once: 3 VALU instructions per piece instead of ~15.
  constexpr bool FASTADDR = CONV && (BM == 512 || FA != 0);
  constexpr bool UP2 = CONV && FA == 2;
  int a_pix[NPA], a_yx[NPA];
  unsigned b_off[NPB];
#pragma unroll
  for (int s = 0; s < NPA; ++s) {
    const int m = m0 + s * 64 + wave * 8 + r8;
    if (FASTADDR) {
      const int per = p.oh * p.ow;
      const int mm = min(m, p.M - 1);
      const int img = mm / per;
      const int rem = mm - img * per;
      const int oy = rem / p.ow;
      const int ox = rem - oy * p.ow;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad_x;
      const int lim_h = UP2 ? p.up_h : p.h, lim_w = UP2 ? p.up_w : p.w;   // (UP2: output pixel and taps live on the upsampled grid)
      unsigned mask = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const bool ok = m < p.M && ky < p.kh && kx < p.kh && iy0 + ky >= 0 && iy0 + ky < lim_h && ix0 + kx >= 0 && ix0 + kx < lim_w;
          mask |= (ok ? 1u : 0u) << (ky * 3 + kx);
        }
      if (UP2) {
        mask |= ((unsigned)iy0 & 1u) << 9 | ((unsigned)ix0 & 1u) << 10;   // parities of the first tap's row / column
        a_pix[s] = (int)((unsigned)(img * p.h * p.w + (iy0 >> 1) * p.w + (ix0 >> 1)) * (unsigned)p.s0 * 2u + kcb);   // (arithmetic shifts: -1 >> 1 = -1)
      } else {
        a_pix[s] = (int)((unsigned)(img * p.h * p.w + iy0 * p.w + ix0) * (unsigned)p.s0 * 2u + kcb);   // modular: exact whenever the tap is valid
      }
      a_yx[s] = (int)mask;
    } else if (CONV) {
      const int per = p.oh * p.ow;
      const int mm = min(m, p.M - 1);
      const int img = mm / per;
      const int rem = mm - img * per;
      const int oy = rem / p.ow;
      const int ox = rem - oy * p.ow;
      a_pix[s] = img * p.h * p.w;
      const int iy0 = (m < p.M) ? oy * p.stride - p.pad : -20000;  // out-of-range rows fail every bounds check
      const int ix0 = ox * p.stride - p.pad_x;
      a_yx[s] = (iy0 << 16) | (ix0 & 0xffff);
    } else {
      a_pix[s] = (m < p.M) ? m : -1;
      a_yx[s] = 0;
    }
  }
#pragma unroll
  for (int s = 0; s < NPB; ++s) {
    const int nn = n0 + s * 64 + wave * 8 + r8;
    b_off[s] = (nn < p.nout) ? (unsigned)nn * (unsigned)p.ldw * 2u + kcb : OOB;
  }

  auto a_piece = [&](int s, const Cursor& c) -> Piece {
    if (UP2) {
      // source row / column of tap k: base + {0, parity, 1}[k]: the uniform part (k == 2) and the per-lane part (k == 1: the parity bit) -- no branch
      const unsigned ps2 = (unsigned)p.s0 * 2u;
      const unsigned uni = ((c.ky == 2 ? (unsigned)p.w : 0u) + (c.kx == 2 ? 1u : 0u)) * ps2;          // scalar ALU
      const unsigned my = c.ky == 1 ? (unsigned)p.w * ps2 : 0u, mx = c.kx == 1 ? ps2 : 0u;            // scalar ALU
      const unsigned py = ((unsigned)a_yx[s] >> 9) & 1u, px = ((unsigned)a_yx[s] >> 10) & 1u;
      const bool ok = ((unsigned)a_yx[s] >> (c.ky * 3 + c.kx)) & 1u;
      return Piece{ok ? (unsigned)a_pix[s] + uni + py * my + px * mx : OOB, (unsigned)c.cc * 2u, false};
    }
    if (FASTADDR) {
      const unsigned delta = (unsigned)(c.ky * p.w + c.kx) * (unsigned)p.s0 * 2u;   // uniform (scalar ALU)
      const bool ok = ((unsigned)a_yx[s] >> (c.ky * 3 + c.kx)) & 1u;
      return Piece{ok ? (unsigned)a_pix[s] + delta : OOB, (unsigned)c.cc * 2u, false};
    }
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
    return Piece{ok ? pix * sstride * 2u + kcb : OOB, coff * 2u, second};
  };
  auto advance = [&](Cursor& c) {
    c.t++;
    c.cc += BK;
    if (XT && c.t == p.kt) c.cc = 0;   // (XT) "K-tile kt" is the next output tile's K-tile 0
    if (CONV && c.cc == Ctot) {
      c.cc = 0;
      if (++c.kx == p.kh) { c.kx = 0; ++c.ky; }
    }
  };
  // piece IDX (0 .. NPA-1: A pieces, NPA .. NP-1: B pieces) of K-tile `c` into stage `buf`; tiles past the end are zero
  // fills (no traffic), IDX >= NP is a no-op
  auto issue_piece = [&](auto IDX, const Cursor& c, int buf) {
    constexpr int idx = decltype(IDX)::value;
    if constexpr (idx < NP) {
      char* sbase = smem + buf * STAGE_BYTES + wave * 1024;
      const bool live = c.t < kt_live;  // uniform
      if constexpr (idx < NPA) {
        constexpr int s = idx;
        const Piece pc = a_piece(s, c);
        auto* dst = (__attribute__((address_space(3))) void*)(sbase + s * 8192);
        const __amdgpu_buffer_rsrc_t rs = pc.second ? rs_a1 : rs_a0;  // uniform select (s_cselect), no branch
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, live ? pc.voff : OOB, pc.soff, 0, 0);
      } else {
        constexpr int s = idx - NPA;
        auto* dst = (__attribute__((address_space(3))) void*)(sbase + A_BYTES + s * 8192);
        const unsigned bo = b_off[s];  // (local copy: hipcc's host pass silently drops the kernel stub when a captured array
        const unsigned voff = live ? bo : OOB;  //  element is handed straight to the buffer builtin)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst, 16, voff, CONV ? (unsigned)c.t * (BK * 2u) : (unsigned)c.cc * 2u, 0, 0);
      }
    }
  };

  // LN consumer: the producer's per-row partial sums of row `lane` of this wave's MI*32 rows are needed in the epilogue; a round trip to
  // another XCD's output costs ~2 us there (15 us per ff1 launch).  One word of the row's 64-byte record is loaded here, ahead of the K loop:
  // it pulls the line into this CU's cache, and ONE register rides through the loop (the 160-accumulator tile has no room for mean and rstd).
  float ln_touch = 0.f;
  const float* ln_row = nullptr;
  if (LN == 2) {
    const int mrow = min(m0 + wm * WROWS + lane, p.M - 1);
    ln_row = p.ln_partial + (long)mrow * (p.ln_parts * 2);
    ln_touch = ln_row[0];
  }
  typedef float accv __attribute__((ext_vector_type(MF == 32 ? 16 : 4)));
  accv acc[MIB][NJB];
#pragma unroll
  for (int i = 0; i < MIB; ++i)
#pragma unroll
    for (int j = 0; j < NJB; ++j)
#pragma unroll
      for (int r = 0; r < (MF == 32 ? 16 : 4); ++r) acc[i][j][r] = 0.f;
  const int l16 = lane & 15, kg = lane >> 4;   // MF 16: block row / column of this lane, its 8-wide k group

  f16x8 af[2][MI];  // [buffer][mi]  activation fragments (MFMA "B" operand)
  f16x8 wf[2][NJ];  // [buffer][nj]  weight fragments     (MFMA "A" operand)

  auto read_frags = [&](int buf, int ks, int fb) {
    const char* sa = smem + buf * STAGE_BYTES;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < MI; ++i) af[fb][i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * (MI * 32) + i * 32 + li, ks * 2 + hi));
#pragma unroll
    for (int j = 0; j < NJ; ++j) wf[fb][j] = *reinterpret_cast<const f16x8*>(sb + lds_off(wn * (NJ * 32) + j * 32 + li, ks * 2 + hi));
  };
  auto mma = [&](int fb) {
    if constexpr (MF == 32) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc[i][j] = FMX_MFMA_32x32x16(wf[fb][j], af[fb][i], acc[i][j]);
    }
  };

  // MFMA / DS-read / VMEM interleave of one k-step: (MFMA, ds_read) x (MI + NJ), then the remaining MFMAs each preceded by
  // one LDS-DMA piece, any left-over pieces last
#define FMX_INTERLEAVE(NV)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int q = 0; q < MI + NJ; ++q) {                                         \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                          \
    }                                                                                             \
    _Pragma("unroll") for (int q = 0; q < MI * NJ - MI - NJ; ++q) {                               \
      if (q < (NV)) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
    }                                                                                             \
    _Pragma("unroll") for (int q = MI * NJ - MI - NJ; q < (NV); ++q) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); \
  }

  constexpr int P0 = MF == 16 ? (NP > 5 ? 5 : NP) : SCHED == 0 ? 3 : SCHED == 1 ? 5 : NP;   // pieces issued behind the barrier (k-step 3 of the tile before)
  constexpr int P1 = SCHED == 0 ? 3 : SCHED == 1 ? NP - 5 : 0;      // in k-step 0; the rest in k-step 1
  auto issue_range = [&](auto LO, auto HI, const Cursor& c, int buf) {
    constexpr int lo = decltype(LO)::value, hi = decltype(HI)::value;
    if constexpr (lo + 0 < hi) issue_piece(IC<lo + 0>{}, c, buf);
    if constexpr (lo + 1 < hi) issue_piece(IC<lo + 1>{}, c, buf);
    if constexpr (lo + 2 < hi) issue_piece(IC<lo + 2>{}, c, buf);
    if constexpr (lo + 3 < hi) issue_piece(IC<lo + 3>{}, c, buf);
    if constexpr (lo + 4 < hi) issue_piece(IC<lo + 4>{}, c, buf);
    if constexpr (lo + 5 < hi) issue_piece(IC<lo + 5>{}, c, buf);
    if constexpr (lo + 6 < hi) issue_piece(IC<lo + 6>{}, c, buf);
    if constexpr (lo + 7 < hi) issue_piece(IC<lo + 7>{}, c, buf);
    if constexpr (lo + 8 < hi) issue_piece(IC<lo + 8>{}, c, buf);
    if constexpr (lo + 9 < hi) issue_piece(IC<lo + 9>{}, c, buf);
  };
  // ---- prologue: tile 0 complete, the first P0 pieces of tile 1 in flight ------------------------------------------------------
  Cursor c1{0, 0, 0, 0};
  if (!pre) issue_range(IC<0>{}, IC<NP>{}, c1, s0);
  advance(c1);  // c1 = tile 1
  issue_range(IC<0>{}, IC<P0>{}, c1, s0 ^ 1);
  if (!pre) {   // (XT, pre: K-tile 0 was complete at the previous tile's last K-tile barrier, and that tile ended on a barrier)
    if constexpr (P0 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (P0 == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (P0 == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (MF == 32) read_frags(0, 0, 0);
  // iteration t:  k-step 0: + pieces 3-5 of tile t+1   k-step 1: + pieces 6.. of tile t+1   k-step 2: nothing
  //               wait + barrier                          k-step 3: + pieces 0-2 of tile t+2 (into the stage just released)
  if constexpr (MF == 16) {
    // ---- 16x16x32 K loop: two k-steps of 32 per K-tile, MIB x NJB = 40 MFMAs each, every fragment SINGLE-buffered (216 registers with the
    //      accumulators, as the 32x32x16 loop).  A k-step runs activation-major in two halves of the weight fragments:
    //        half A:  for i: for j <  NJB/2: acc[i][j] += w[j] a[i]      half B:  for i: for j >= NJB/2: acc[i][j] += w[j] a[i]
    //      A fragment is re-read for the NEXT k-step right behind its last MFMA of this one: w[j < NJB/2] behind the last activation row of half A,
    //      a[i] behind its group of half B, w[j >= NJB/2] behind the last row of half B -- each 15 to 25 MFMAs (> 240 cycles) before its next use.
    //      The one barrier per K-tile sits INSIDE k-step 1, in front of its first re-read (MFMA 15 of 40): k-step 1's own fragments were read during
    //      k-step 0, whose last reads complete under the 15 MFMAs in front of the barrier (waiting for them at the k-step boundary cost 460 cycles per
    //      K-tile when first tried: 3 330 against 2 860 of the 32x32x16 loop); behind it k-step 1 re-reads from stage buf^1 and stages tile t+2 into buf.
    constexpr int HJ = NJB / 2, HM = MIB * HJ, NM = 2 * HM;   // weight fragments per half, MFMAs per half / per k-step
    static_assert(NJB % 2 == 0, "two halves of the weight fragments");
    f16x8 a4[MIB], w1[NJB];
    // byte offset of this lane's fragment 0 of either operand in a stage, k-step 0.  Fragment i is 16 rows further: + i * 2048 exactly (the XOR swizzle
    // keys on (row >> 1) & 7, which 16 rows do not change) -- an immediate of the ds_read; k-step 1 = ^ 64 (chunk bit 2 survives the swizzle).
    static_assert((WROWS % 16) == 0 && ((NJ * 32) % 16) == 0, "fragment rows in steps of 16");
    const unsigned abase = (unsigned)lds_off(wm * WROWS + l16, kg);
    const unsigned wbase = (unsigned)(A_BYTES + lds_off(wn * (NJ * 32) + l16, kg));
    auto rd = [&](unsigned addr, int frag) { return *reinterpret_cast<const f16x8*>(smem + addr + frag * 2048); };
    {
      const unsigned st0 = (unsigned)s0 * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < MIB; ++i) a4[i] = rd(abase + st0, i);
#pragma unroll
      for (int j = 0; j < NJB; ++j) w1[j] = rd(wbase + st0, j);
    }
#ifndef FMX_MF16_Q0
#define FMX_MF16_Q0 5
#endif
#ifndef FMX_MF16_GAP
#define FMX_MF16_GAP 4
#endif
    constexpr int Q0 = NP > FMX_MF16_Q0 ? FMX_MF16_Q0 : NP;   // pieces of tile t+2 issued in k-step 1 (behind the barrier); the rest in k-step 0 of the next iteration
    constexpr int GAP = FMX_MF16_GAP;                          // one LDS-DMA piece behind every GAP-th MFMA
    // MFMAs [LO, HI) of a k-step with what rides behind them: the fragment re-reads, and one LDS-DMA piece behind every second MFMA from PM0 on
    auto kpart = [&](auto LOC, auto HIC, unsigned kxor, unsigned stage_off, auto&& piece, auto NPIECES, auto PM0C) {
      constexpr int LO = decltype(LOC)::value, HI = decltype(HIC)::value, npieces = decltype(NPIECES)::value, PM0 = decltype(PM0C)::value;
      const unsigned aaddr = (abase ^ kxor) + stage_off, waddr = (wbase ^ kxor) + stage_off;   // the only per-k-step address arithmetic
      static_for<HI - LO>([&](auto MC) {
        constexpr int m = LO + decltype(MC)::value;
        constexpr bool hb = m >= HM;
        constexpr int mm = hb ? m - HM : m;
        constexpr int i = mm / HJ, j = (hb ? HJ : 0) + mm % HJ;
        acc[i][j] = FMX_MFMA_16x16x32(w1[j], a4[i], acc[i][j]);
        if constexpr (i == MIB - 1) w1[j] = rd(waddr, j);
        if constexpr (hb && mm % HJ == HJ - 1) a4[i] = rd(aaddr, i);
        if constexpr (m >= PM0 && ((m - PM0) % GAP) == 0 && (m - PM0) / GAP < npieces) piece(IC<(m - PM0) / GAP>{});
      });
      static_for<HI - LO>([&](auto MC) {
        constexpr int m = LO + decltype(MC)::value;
        constexpr bool hb = m >= HM;
        constexpr int mm = hb ? m - HM : m;
        constexpr int i = mm / HJ;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (i == MIB - 1 && hb && mm % HJ == HJ - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if constexpr (i == MIB - 1 || (hb && mm % HJ == HJ - 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (m >= PM0 && ((m - PM0) % GAP) == 0 && (m - PM0) / GAP < npieces) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      });
    };
    constexpr int MB = HM - HJ;   // the first MFMA of a k-step with a re-read behind it: the barrier of the K-tile sits in front of it in k-step 1
    static_assert(MB + 1 + GAP * (Q0 - 1) < NM && 1 + GAP * (NP - Q0 - 1) < NM, "the pieces fit their k-step");
    for (int t = 0; t < p.kt; ++t) {
      const int buf = (t + s0) & 1;
      Cursor c2 = c1;
      advance(c2);  // tile t+2
      // (opaque per iteration: loop-invariant otherwise, and the fragment addresses of both k-steps and stages would ride through the loop)
      unsigned k64 = 64u, k0 = 0u;
      asm volatile("" : "+s"(k64), "+s"(k0));
      const unsigned off_cur = (unsigned)buf * STAGE_BYTES, off_nxt = (unsigned)(buf ^ 1) * STAGE_BYTES;
      auto piece1 = [&](auto IDX) { issue_piece(IC<Q0 + decltype(IDX)::value>{}, c1, buf ^ 1); };
      auto piece2 = [&](auto IDX) { issue_piece(IC<decltype(IDX)::value>{}, c2, buf); };
      // k-step 0: re-reads k-step 1's fragments from this tile's stage; the rest of tile t+1's pieces
      __builtin_amdgcn_sched_barrier(0);
      kpart(IC<0>{}, IC<NM>{}, k64, off_cur, piece1, IC<NP - Q0>{}, IC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      // k-step 1, MFMAs in front of its first re-read: nothing rides behind them -- the reads k-step 0 issued last complete under them
      kpart(IC<0>{}, IC<MB>{}, k0, off_nxt, piece2, IC<0>{}, IC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      // every wave: its reads of stage `buf` are complete, its pieces of tile t+1 have landed -> the K-tile's one barrier; behind it stage buf^1
      // is readable and stage buf may be overwritten with tile t+2
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if constexpr (XT) {
        // two K-tiles before the end: every piece issued from here on belongs to the NEXT output tile's K-tile 0 (or is a no-op) -> its addresses
        if (has_next && t == p.kt - 2) {
          int tmn, tnn;
          tile_origin(lid + (int)gridDim.x, tmn, tnn);
#pragma unroll
          for (int s = 0; s < NPA; ++s) {
            const int m = tmn * BM + s * 64 + wave * 8 + r8;
            a_pix[s] = (m < p.M) ? m : -1;
          }
#pragma unroll
          for (int s = 0; s < NPB; ++s) {
            const int nn = tnn * BN + s * 64 + wave * 8 + r8;
            b_off[s] = (nn < p.nout) ? (unsigned)nn * (unsigned)p.ldw * 2u + kcb : OOB;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      kpart(IC<MB>{}, IC<NM>{}, k0, off_nxt, piece2, IC<Q0>{}, IC<MB + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      c1 = c2;
    }
  } else
  for (int t = 0; t < p.kt; ++t) {
    const int buf = t & 1;
    Cursor c2 = c1;
    advance(c2);  // tile t+2
    __builtin_amdgcn_sched_barrier(0);
    read_frags(buf, 1, 1);
    mma(0);
    issue_range(IC<P0>{}, IC<P0 + P1>{}, c1, buf ^ 1);
    FMX_INTERLEAVE(P1)
    __builtin_amdgcn_sched_barrier(0);
    read_frags(buf, 2, 0);
    mma(1);
    issue_range(IC<P0 + P1>{}, IC<NP>{}, c1, buf ^ 1);
    FMX_INTERLEAVE(NP - P0 - P1)
    __builtin_amdgcn_sched_barrier(0);
    read_frags(buf, 3, 1);
    mma(0);
    FMX_INTERLEAVE(0)
    __builtin_amdgcn_sched_barrier(0);
    // this wave's reads of stage `buf` are complete and its pieces of tile t+1 have landed -> one barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(buf ^ 1, 0, 0);
    mma(1);
    issue_range(IC<0>{}, IC<P0>{}, c2, buf);
    FMX_INTERLEAVE(P0)
    __builtin_amdgcn_sched_barrier(0);
    c1 = c2;
  }
#undef FMX_INTERLEAVE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's zero-fill pieces must land before the LDS is released

  // ---- epilogue: through LDS, so that every global access is row-contiguous.  The MFMA leaves a lane with 4-channel runs of
  //      ONE pixel; stored as they are (even widened to 16 B by a half-wave swap) every store
  //      instruction scatters 32-byte pieces over 32 cache lines (transaction-bound, not bandwidth-bound).  Here each wave
  //      transposes its sub-tile through a private slice of the (now idle) LDS in fp32, one 32-row block row at a time
  //      (GEGLU: the whole sub-tile at once, it is half as wide), and NJ*4 (GEGLU: NJ*2) lanes then own one output row:
  //      residual loads and stores are contiguous runs of NJ*64 (NJ*32) bytes.  16-byte chunks are XOR-swizzled by row on
  //      both sides; the arithmetic (fp32, one rounding) is unchanged.  The row pass itself lives in epi_rows() below.
  __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage: the LDS is free
  // XT: the slices stay clear of the stage that holds the next tile's K-tile 0
  const int sN = (p.kt + s0) & 1;
  char* my = !XT ? smem + wave * G::WAVE_EPI_BYTES
                 : smem + ((has_next && sN == 1) ? (wave < 7 ? wave * XSL : 2 * STAGE_BYTES) : G::LDS_BYTES - 8 * XSL + wave * XSL);
  const FastEpilogue ep(p);
  const bool geglu = !STATS && LN != 1 && LN != 3 && p.act == FMX_ACT_GEGLU;   // (no GEGLU code in the statistics-emitting / operand-swapped kernels: registers; the host refuses)
  // LN consumer: mean / rstd of row `lane` of this wave's rows (fixed summation order); lane l serves row l through __shfl below
  float lnm = 0.f, lnr = 1.f;
  if (LN == 2) {
    float s1 = ln_touch, s2 = 0.f;   // (entry 0's sum is the word loaded before the K loop)
    const float* q = ln_row;
    s2 += q[1];
    for (int k2 = 1; k2 < p.ln_parts; ++k2) {
      s1 += q[2 * k2];
      s2 += q[2 * k2 + 1];
    }
    lnm = s1 * p.ln_inv_c;
    lnr = rsqrtf(fmaxf(s2 * p.ln_inv_c - lnm * lnm, 0.f) + p.ln_eps);
    // the same pairs for the operand-swapped GEMM that follows on these rows (fmx.h ln_ab_out): one tile column writes them
    if (p.ln_ab_out && tn == 0 && wn == 0) {
      const int mrow = m0 + wm * WROWS + lane;
      if (mrow < p.M) *reinterpret_cast<f32x2*>(p.ln_ab_out + (long)mrow * 2) = f32x2{lnr, -lnm * lnr};
    }
  }
#ifndef FMX_ELEM_BF16
  if constexpr (XA) {
    // ---- cross-attention out of the accumulators (reference backend/nn/unet.py:145-155, 254-267) -----------------------------------
    // The tile is Q[256 queries of ONE image][320 columns = 5 heads of 64]; this wave holds 64 queries x 160 columns = 2.5 heads: lane (l16, kg) has, of
    // every 16 x 16 accumulator block (i, j), query i*16 + l16 and columns j*16 + kg*4 + [0, 4).  That IS the B-operand layout of the next MFMA up to a
    // permutation of its contraction index: two column blocks side by side give a lane 8 values at d = {4 kg + [0,4)} u {16 + 4 kg + [0,4)} of a 32-wide
    // d step, and a contraction does not care about the order of its index as long as the other operand uses the same one -- so the K fragments are read
    // from LDS in that order (two 8-byte reads instead of one 16-byte read) and Q never leaves the registers.  S^T = K Q^T comes out as lane = (query
    // l16, keys 4 kg + [0,4) of key block kb): the same trick gives P^T as the B operand of O^T = V^T P^T (key blocks paired for 16x16x32, the fifth
    // block through 16x16x16), and O^T lands with lane = (query l16, d = 4 kg + [0, 4) of d block db) -- the accumulator layout of the GEMM itself, so O
    // is written back into the Q columns' registers and leaves through the ordinary transposing epilogue.
    //   head 2 of the tile straddles the two wave columns (wn 0: d 0..31, wn 1: d 32..63): the partners exchange that Q half through LDS (4 KB per wave,
    //   under the barrier the staging needs anyway), both compute the head's full S (5 redundant MFMAs per query block) and each its own half of O.
    //   LDS: K image [d chunk c = 0..39][key 0..79] x 16 B (a wave's 16 keys of one chunk are 256 contiguous bytes: conflict-free 8-byte reads),
    //   V^T image [key chunk 0..9][d 0..319] x 16 B, 100 LDS-DMA pieces of 1 KiB; the exchange buffer behind them.  Numerics = fmx_attention_f16's
    //   short-context kernel: Q rounded to fp16, pre-scaled by scale log2(e) and rounded again, fp32 scores, exp2, row sum of the unrounded
    //   exponentials, P rounded to fp16, fp32 O scaled by 1 / l.
    // MEASURED (profiles/r32_cross_attention_in_to_q_epilogue_ab.jsonl, SDXL 1024^2 batch 8, same box, interleaved): correct (agrees with the two
    // launches it replaces to fp32 summation order) and SLOWER -- the fused launch 102.8 us against 56.7 (projection) + 28.3 (fmx_attention_f16) = 85.0 us,
    // step 102.4 -> 104.9 ms.  The 77-key attention is not a memory pass that a producer-side fusion deletes: per wave it is ~4 000 vector instructions
    // (mask / max / exp2 / sum / two conversions per score on 16 x 16 blocks, the LayerNorm-consumer arithmetic and two roundings per Q element) around 260
    // MFMAs, i.e. bound by vector issue; as its own launch that work runs four workgroups deep per CU under its memory traffic, as an epilogue it runs
    // alone on a CU whose matrix pipes wait (one query block at a time; two at a time spill 143 registers next to the 160 accumulators: 149 us).  The
    // executor therefore keeps the two launches (backend/nn/unet.py, FMX_XATTN_FUSE=1 selects this path for A/B); the kernel stays for the record and
    // its test.
    static_assert(!XA || (MF == 16 && BM == 256 && BN == 320 && LN == 2 && !STATS && !CONV), "cross-attention epilogue: the 256 x 320 LayerNorm-consumer linear kernel");
    constexpr int XKEYS = 80, NKB = 5;
    constexpr int K_BYTES = 40 * XKEYS * 16, V_BYTES = 10 * 320 * 16;
    static_assert(K_BYTES + V_BYTES + 8 * 4096 <= G::LDS_BYTES, "K / V^T images and the exchange buffer fit the LDS");
    char* const xk = smem;
    char* const xv = smem + K_BYTES;
    char* const xq = smem + K_BYTES + V_BYTES;
    {
      const int img = m0 / p.xa_rows;
      const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.xa_k), 0, p.xa_k_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.xa_vt), 0, p.xa_vt_bytes, 0x00020000);
      for (int pc = wave; pc < 100; pc += 8) {   // (uniform: `wave` is a scalar)
        const bool isk = pc < 50;
        const int pq = isk ? pc : pc - 50;
        const int sl = pq * 64 + lane;
        unsigned off;
        if (isk) {
          const int c = sl / XKEYS, key = sl - c * XKEYS;
          off = (unsigned)((img * p.xa_k_bs + key) * p.xa_k_rs + n0 + c * 8) * 2u;
        } else {
          const int kc = sl / 320, d = sl - kc * 320;
          off = (unsigned)((n0 + d) * p.xa_vt_ds + img * p.xa_vt_bs + kc * 8) * 2u;
        }
        auto* dst = (__attribute__((address_space(3))) void*)((isk ? xk : xv) + pq * 1024);
        const __amdgpu_buffer_rsrc_t rs = isk ? rs_k : rs_v;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 0);
      }
    }
    const float c2 = p.xa_c2;
    float mr[MIB], rr[MIB];
#pragma unroll
    for (int i = 0; i < MIB; ++i) {
      mr[i] = __shfl(lnm, i * 16 + l16);
      rr[i] = __shfl(lnr, i * 16 + l16);
    }
    const int colw = n0 + wn * (NJ * 32);
    // column blocks JB, JB + 1 of this wave -> one pre-scaled fp16 B-operand fragment per query block I0, I0 + 1 (the LayerNorm-consumer arithmetic of
    // epi_rows).  Two query blocks at a time: four (80 score registers per head) spill next to the 160 accumulators.
#ifndef FMX_XA_QH
#define FMX_XA_QH 1
#endif
    constexpr int QH = FMX_XA_QH;
    auto qfrag_pair = [&](auto JB, auto I0, f16x8(&out)[QH]) {
      constexpr int jb = decltype(JB)::value, i0 = decltype(I0)::value;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int col = colw + (jb + half) * 16 + kg * 4;
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.ln_colsum + col);
        const f16x4 bb = *reinterpret_cast<const f16x4*>(ep.bias + (long)col * ep.mb);
#pragma unroll
        for (int ii = 0; ii < QH; ++ii)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i0 + ii][jb + half][r];   // (alpha = 1: the host refuses anything else here)
            v = (v - mr[i0 + ii] * cs[r]) * rr[i0 + ii];
            v += (float)bb[r];
            const f16 qh = (f16)v;                           // Q as fmx_attention_f16 would have read it from memory
            out[ii][half * 4 + r] = (f16)((float)qh * c2);   // ... and pre-scaled as that kernel does
          }
      }
    };
    auto pack8 = [](f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; };
    // MODE 0: a whole head in column blocks J0 .. J0 + 3;  1: head 2's d 0..31 in J0, J0 + 1 (wn 0);  2: head 2's d 32..63 in J0, J0 + 1 (wn 1)
    auto head = [&](auto HC, auto J0, auto MODE, auto I0) {
      constexpr int hc = decltype(HC)::value, j0 = decltype(J0)::value, mode = decltype(MODE)::value, i0 = decltype(I0)::value;
      f16x8 qf[QH][2];
      if constexpr (mode == 0) {
        f16x8 t0[QH], t1[QH];
        qfrag_pair(IC<j0>{}, IC<i0>{}, t0);
        qfrag_pair(IC<j0 + 2>{}, IC<i0>{}, t1);
#pragma unroll
        for (int ii = 0; ii < QH; ++ii) { qf[ii][0] = t0[ii]; qf[ii][1] = t1[ii]; }
      } else {
        f16x8 own[QH];
        qfrag_pair(IC<j0>{}, IC<i0>{}, own);   // (recomputed: the copy that went to the partner is not kept)
#pragma unroll
        for (int ii = 0; ii < QH; ++ii) {
          const f16x8 pq = *reinterpret_cast<const f16x8*>(xq + (wave ^ 1) * 4096 + (i0 + ii) * 1024 + lane * 16);
          qf[ii][mode == 1 ? 0 : 1] = own[ii];
          qf[ii][mode == 1 ? 1 : 0] = pq;
        }
      }
      f32x4 S[QH][NKB];
#pragma unroll
      for (int ii = 0; ii < QH; ++ii)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) S[ii][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int c = hc * 8 + ks * 4 + (kg >> 1);
          const char* a = xk + (c * XKEYS + kb * 16 + l16) * 16 + (kg & 1) * 8;
          const f16x8 kf = pack8(*reinterpret_cast<const f16x4*>(a), *reinterpret_cast<const f16x4*>(a + 2 * XKEYS * 16));
#pragma unroll
          for (int ii = 0; ii < QH; ++ii) S[ii][kb] = FMX_MFMA_16x16x32(kf, qf[ii][ks], S[ii][kb]);
        }
      f16x4 P[QH][NKB];
      float inv[QH];
#pragma unroll
      for (int ii = 0; ii < QH; ++ii) {
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kb * 16 + 16 > p.xa_nk) {   // (uniform: only the key block that holds the end of the context pays for the mask)
              if (kb * 16 + kg * 4 + r >= p.xa_nk) S[ii][kb][r] = -INFINITY;
            }
            mx = fmaxf(mx, S[ii][kb][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float l = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(S[ii][kb][r] - mx);
            l += e;
            P[ii][kb][r] = (f16)e;
          }
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        inv[ii] = 1.0f / l;
      }
      constexpr int db0 = mode == 2 ? 2 : 0, ndb = mode == 0 ? 4 : 2;
      static_for<ndb>([&](auto DBI) {
        constexpr int dbi = decltype(DBI)::value, db = db0 + dbi;
        f32x4 O[QH];
#pragma unroll
        for (int ii = 0; ii < QH; ++ii) O[ii] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int d = hc * 64 + db * 16 + l16;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          const int kc = kp * 4 + (kg >> 1);
          const char* a = xv + (kc * 320 + d) * 16 + (kg & 1) * 8;
          const f16x8 vf = pack8(*reinterpret_cast<const f16x4*>(a), *reinterpret_cast<const f16x4*>(a + 2 * 320 * 16));
#pragma unroll
          for (int ii = 0; ii < QH; ++ii) O[ii] = FMX_MFMA_16x16x32(vf, pack8(P[ii][2 * kp], P[ii][2 * kp + 1]), O[ii]);
        }
        {
          const int kc = 8 + (kg >> 1);
          const f16x4 vf4 = *reinterpret_cast<const f16x4*>(xv + (kc * 320 + d) * 16 + (kg & 1) * 8);
#pragma unroll
          for (int ii = 0; ii < QH; ++ii) {
#ifdef FMX_XA_TAIL16
            O[ii] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf4, P[ii][4], O[ii], 0, 0, 0);
#else
            // (the fifth key block as a 16x16x32 step with a zero second half.  With a v_mfma_f32_16x16x16_f16 taking over the accumulator of the
            //  v_mfma_f32_16x16x32_f16 straight in front of it, registers 0 and 1 of the result came back wrong in a fixed subset of the blocks -- every
            //  launch, no spills involved, gone with this form; FMX_XA_TAIL16 rebuilds the failing sequence)
            const f16x4 z4 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            O[ii] = FMX_MFMA_16x16x32(pack8(vf4, vf4), pack8(P[ii][4], z4), O[ii]);
#endif
          }
        }
#pragma unroll
        for (int ii = 0; ii < QH; ++ii)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i0 + ii][j0 + dbi][r] = O[ii][r] * inv[ii];
      });
    };
    // the straddling head's own half first: its fragments go to the partner through LDS, under the barrier that also publishes the staged K / V^T
    static_for<MIB / QH>([&](auto IH) {
      constexpr int i0 = decltype(IH)::value * QH;
      f16x8 o0[QH];
      if (wn == 0) qfrag_pair(IC<8>{}, IC<i0>{}, o0);
      else qfrag_pair(IC<0>{}, IC<i0>{}, o0);
#pragma unroll
      for (int ii = 0; ii < QH; ++ii) *reinterpret_cast<f16x8*>(xq + wave * 4096 + (i0 + ii) * 1024 + lane * 16) = o0[ii];
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");   // (the raw barrier does not keep the compiler from hoisting the LDS reads below above it)
    __builtin_amdgcn_sched_barrier(0);
    static_for<MIB / QH>([&](auto IH) {
      constexpr int i0 = decltype(IH)::value * QH;
      if (wn == 0) {
        head(IC<2>{}, IC<8>{}, IC<1>{}, IC<i0>{});
        head(IC<0>{}, IC<0>{}, IC<0>{}, IC<i0>{});
        head(IC<1>{}, IC<4>{}, IC<0>{}, IC<i0>{});
      } else {
        head(IC<2>{}, IC<0>{}, IC<2>{}, IC<i0>{});
        head(IC<3>{}, IC<2>{}, IC<0>{}, IC<i0>{});
        head(IC<4>{}, IC<6>{}, IC<0>{}, IC<i0>{});
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done reading K / V^T / the exchange buffer: the transposing epilogue may take the LDS
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
  if (!geglu) {
    constexpr int RB = NJ * 128;       // staged row: NJ*32 fp32
    constexpr int LPR = NJ * 4;        // lanes per output row (8 columns each)
    const int cg = lane % LPR;
    const int nb = n0 + wn * (NJ * 32) + cg * 8;
    const bool nok = nb < ep.nout;
    const int nbc = nok ? nb : 0;
    float st[STATS ? 16 : 1];
#pragma unroll
    for (int r = 0; r < (STATS ? 16 : 1); ++r) st[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MIB; ++i) {
      if constexpr (MF == 32) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int chunk = j * 8 + q4 * 2 + hi;
            const accv& a = acc[i][j];
            *reinterpret_cast<f32x4*>(my + li * RB + ((chunk ^ (li & 7)) << 4)) = f32x4{a[q4 * 4], a[q4 * 4 + 1], a[q4 * 4 + 2], a[q4 * 4 + 3]};
          }
      } else {   // 16 x 16 blocks: lane = pixel l16, its 4 registers = channels j*16 + kg*4 + [0, 4) -> one 16-byte chunk per block
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
          const int chunk = j * 4 + kg;
          const accv& a = acc[i][j];
          *reinterpret_cast<f32x4*>(my + l16 * RB + ((chunk ^ (l16 & 7)) << 4)) = f32x4{a[0], a[1], a[2], a[3]};
        }
      }
      // same wave wrote and reads: LDS operations of one wave execute in order, no barrier needed
      const int mbase = m0 + wm * WROWS + i * BR;
      // FA = 3 (round 6, fmx_conv3x3_up2x): the BR rows of this block are neighbours in ONE row of the phase's pixel grid (scat_ow % 32 == 0), whose
      // pixels sit every second pixel of every second row of the output: the row pass keeps `out + m * ld_out`, the block's base moves
      long scat = 0;
      if constexpr (FA == 3) scat = (long)(mbase / p.scat_ow) * p.scat_extra;
#define FMX_EPI_ARGS my, lane, mbase, p.M, nbc, nok, ep.per_img, ep.alpha, (float)ep.mgt, ep.bias + nbc * ep.mb, ep.rowvec + nbc * ep.mrv, ep.ld_rv, \
                     ep.gate + nbc * ep.mgt, ep.ld_gt, ep.res + nbc * ep.mres, ep.ld_res, ep.out + nbc + scat, ep.ld_out, st
      if constexpr (XA) {   // O is final: plain fp16 store (no bias, no LayerNorm arithmetic -- that went into Q)
        epi_rows<RB, LPR, BR, 7, false, 2>(my, lane, mbase, p.M, nbc, nok, ep.per_img, 1.0f, 0.0f, p.zp, p.zp, 0L, p.zp, 0L, p.zp, 0L, ep.out + nbc, ep.ld_out, st);
      } else
      if (STATS) {  // (the host refuses GELU-tanh / gate together with statistics)
        if (ep.mrv) epi_rows<RB, LPR, BR, 7, false, 3, true>(FMX_EPI_ARGS);
        else if (ep.mres) epi_rows<RB, LPR, BR, 7, false, 1, true>(FMX_EPI_ARGS);
        else epi_rows<RB, LPR, BR, 7, false, 2, true>(FMX_EPI_ARGS);
      } else if (LN == 1) {   // producer: bias + residual (the host sends nothing else here)
        epi_rows<RB, LPR, BR, 7, false, 1, false, 1>(FMX_EPI_ARGS, p.row_stats + (long)(tn * G::WN + wn) * 2, G::WN * p.tiles_n * 2);
      } else if (LN == 2) {   // consumer: bias only
        epi_rows<RB, LPR, BR, 7, false, 2, false, 2>(FMX_EPI_ARGS, nullptr, 0, lnm, lnr, i * BR, p.ln_colsum + nbc);
      } else if (LN == 3) {   // consumer, operand-swapped: the host sends no bias / residual here
        epi_rows<RB, LPR, BR, 7, false, 2, false, 3>(FMX_EPI_ARGS, nullptr, 0, 0.f, 0.f, 0, p.ln_col_ab + (long)nbc * 2, p.ln_row_cb);
      } else if (ep.gelu_tanh) epi_rows<RB, LPR, BR, 7, true, 0>(FMX_EPI_ARGS);       // uniform branches
      else if (ep.mrv | ep.mgt) epi_rows<RB, LPR, BR, 7, false, 0>(FMX_EPI_ARGS);
      else if (ep.mres) epi_rows<RB, LPR, BR, 7, false, 1>(FMX_EPI_ARGS);
      else epi_rows<RB, LPR, BR, 7, false, 2>(FMX_EPI_ARGS);
#undef FMX_EPI_ARGS
    }
    if (STATS) {
      // GroupNorm statistics of this BM-row tile (= chunk `m0 % per_img / BM` of image `m0 / per_img`; the host guarantees
      // per_img % BM == 0, hence M % BM == 0).  A lane holds the sums of its 8 columns over the rows it stored.  Fold
      //   (1) the RPI row-lanes of each column group inside the wave: through the wave's own LDS slice (idle now; LDS operations of
      //       one wave execute in order, so no barrier is needed to reuse it),
      //   (2) the WM waves that share the tile's columns: every wave leaves its sums at a fixed place of its slice, one barrier,
      //       the wm == 0 wave adds them in wave order,
      // and write {sum, sum of squares} per column: partial[(img * nch + chunk) * nout + col][2].  Fixed order -> deterministic.
      constexpr int RPI = 64 / LPR;
      float* sl = reinterpret_cast<float*>(my);
#pragma unroll
      for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(sl + lane * 16 + v * 4) = f32x4{st[v * 4], st[v * 4 + 1], st[v * 4 + 2], st[v * 4 + 3]};
      if (lane < LPR) {
        f32x4 a4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) a4[v] = *reinterpret_cast<const f32x4*>(sl + lane * 16 + v * 4);
#pragma unroll
        for (int rs = 1; rs < RPI; ++rs)
#pragma unroll
          for (int v = 0; v < 4; ++v) a4[v] += *reinterpret_cast<const f32x4*>(sl + (rs * LPR + lane) * 16 + v * 4);
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(sl + 1024 + lane * 16 + v * 4) = a4[v];
      }
      __syncthreads();
      if (wm == 0 && lane < LPR && nok) {
        f32x4 a4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) a4[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w2 = 0; w2 < G::WM; ++w2) {
          const float* o = reinterpret_cast<const float*>(smem + (w2 * G::WN + wn) * G::WAVE_EPI_BYTES) + 1024 + lane * 16;
#pragma unroll
          for (int v = 0; v < 4; ++v) a4[v] += *reinterpret_cast<const f32x4*>(o + v * 4);
        }
        const int img = m0 / ep.per_img;
        const int chunk = (m0 - img * ep.per_img) / BM;
        float* dst = p.stats + ((long)(img * p.stats_nch + chunk) * ep.nout + nb) * 2;   // a4[0..1] = sums, a4[2..3] = sums of squares
        *reinterpret_cast<f32x4*>(dst) = f32x4{a4[0][0], a4[2][0], a4[0][1], a4[2][1]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{a4[0][2], a4[2][2], a4[0][3], a4[2][3]};
        *reinterpret_cast<f32x4*>(dst + 8) = f32x4{a4[1][0], a4[3][0], a4[1][1], a4[3][1]};
        *reinterpret_cast<f32x4*>(dst + 12) = f32x4{a4[1][2], a4[3][2], a4[1][3], a4[3][3]};
      }
    }
  } else {
    // weight rows are interleaved [16 value | 16 gate] per 32-row block: registers q4 = 0,1 of a block are the values of
    // output columns q4*8 + hi*4 + [0,4), registers q4 = 2,3 their gates (same lane).  Staged row = NJ*16 outputs.
    constexpr int RB = NJ * 64;
    constexpr int LPR = NJ * 2;
    // GH passes over the wave's rows: one (the whole MI*32 x NJ*16 sub-tile staged at once), or (XT: XSL-byte slices) two halves of MIB/2 block rows
    constexpr int GH = XT ? 2 : 1;
    constexpr int ROWS = MI * 32 / GH;
    constexpr int IH = MIB / GH;
    static_assert(!XT || ROWS * RB <= XSL, "a GEGLU half fits the slice");
    const int lrow = MF == 32 ? li : l16;   // this lane's pixel row inside a block row
    int mcs[MIB], imgs[MIB];
#pragma unroll
    for (int i = 0; i < MIB; ++i) {
      const int mrow = m0 + wm * WROWS + i * BR + lrow;
      mcs[i] = mrow < p.M ? mrow : p.M - 1;
      imgs[i] = mcs[i] / ep.per_img;
    }
    // the per-image row vector (a ResBlock's embedding add) is absent from every GEGLU call of the UNet / Flux executors: its loads
    // (two 8-byte loads and their wait per 4 outputs, from the zero page when absent) are compiled out of the common instantiation
    float gm[MIB], gr[MIB];   // LN consumer: mean / rstd of this lane's pixel row in each block row
#pragma unroll
    for (int i = 0; i < MIB; ++i) {
      gm[i] = LN == 2 ? __shfl(lnm, i * BR + lrow) : 0.f;
      gr[i] = LN == 2 ? __shfl(lnr, i * BR + lrow) : 1.f;
    }
    // MF 32: registers q4 = 0, 1 of block j are values of columns q4*8 + hi*4 + [0, 4), registers q4 = 2, 3 their gates.
    // MF 16: block 2 jj holds the 16 values of weight-row group jj, block 2 jj + 1 their gates (same lane: columns kg*4 + [0, 4)); the loop
    //        below runs q4 over ONE value run per (jj, lane): q4 = kg.
    auto stage_geglu = [&](auto HAS_RV, auto HALF) {
      constexpr bool RV = decltype(HAS_RV)::value != 0;
      constexpr int I0 = decltype(HALF)::value * IH;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int q4 = 0; q4 < (MF == 32 ? 2 : 1); ++q4) {
          const int nb = n0 + wn * (NJ * 32) + j * 32 + (MF == 32 ? q4 * 8 + hi * 4 : kg * 4);
          const int nbc = nb < ep.nout ? nb : 0;
          const f16x4 bv = ep.bias4(nbc), bg = ep.bias4(nbc + 16);  // few live registers: the 160-accumulator tile has none to spare
          f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, sg = sv;
          if (LN == 2) {
            sv = *reinterpret_cast<const f32x4*>(p.ln_colsum + nbc);
            sg = *reinterpret_cast<const f32x4*>(p.ln_colsum + nbc + 16);
          }
          f32x4 bvf, bgf;
#pragma unroll
          for (int r = 0; r < 4; ++r) { bvf[r] = (float)bv[r]; bgf[r] = (float)bg[r]; }
#pragma unroll
          for (int i = I0; i < I0 + IH; ++i) {
            f16x4 rvv, rvg;
            if (RV) { rvv = ep.rv4(imgs[i], nbc); rvg = ep.rv4(imgs[i], nbc + 16); }
            // (acc alpha - mean colsum) rstd + bias  =  acc (alpha rstd) + (bias - mean rstd colsum): one FMA per element and one per column
            const float sc = LN == 2 ? ep.alpha * gr[i] : ep.alpha;
            const float mr = gm[i] * gr[i];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float cv = bvf[r], cg = bgf[r];
              if (LN == 2) { cv = fmaf(-mr, sv[r], cv); cg = fmaf(-mr, sg[r], cg); }
              if (RV) { cv += (float)rvv[r]; cg += (float)rvg[r]; }
              const float val = fmaf(MF == 32 ? acc[i][j][(q4 * 4 + r) % (MF == 32 ? 16 : 4)] : acc[i][(2 * j) % NJB][r], sc, cv);
              const float gate = fmaf(MF == 32 ? acc[i][j][(8 + q4 * 4 + r) % (MF == 32 ? 16 : 4)] : acc[i][(2 * j + 1) % NJB][r], sc, cg);
              // (a transcendental-free erf -- odd polynomial of degree 19, 12 packable operations -- was measured in round 3: 117.1 vs 117.1 ms
              //  per step, profiles/r08j; the epilogue is not bound by v_rcp / v_exp issue, and the exact form is 30x more accurate)
              o[r] = val * gelu_erf_f(gate);
            }
            const int row = (i - I0) * BR + lrow, chunk = j * 4 + (MF == 32 ? q4 * 2 + hi : kg);
            *reinterpret_cast<f32x4*>(my + row * RB + ((chunk ^ (row & 3)) << 4)) = o;  // NJ*4 chunks per row: XOR of the low 2 bits stays inside
          }
        }
    };
    static_for<GH>([&](auto HALF) {
      constexpr int h = decltype(HALF)::value;
      if (ep.mrv) stage_geglu(IC<1>{}, HALF); else stage_geglu(IC<0>{}, HALF);
      const int cg = lane % LPR;
      const int col = ((n0 + wn * (NJ * 32)) >> 1) + cg * 8;
      const bool nok = col < ep.ncols;
      const int colc = nok ? col : 0;
      // bias / rowvec / act were applied above: the row pass only adds the residual (alpha = 1, every other operand -> zero page)
      const int mb = m0 + wm * WROWS + h * ROWS;
      if (ep.mres) epi_rows<RB, LPR, ROWS, 3, false, 1>(my, lane, mb, p.M, colc, nok, ep.per_img, 1.0f, 0.0f, p.zp, p.zp, 0L, p.zp, 0L,
                                                         ep.res + colc, ep.ld_res, ep.out + colc, ep.ld_out);
      else epi_rows<RB, LPR, ROWS, 3, false, 2>(my, lane, mb, p.M, colc, nok, ep.per_img, 1.0f, 0.0f, p.zp, p.zp, 0L, p.zp, 0L, p.zp, 0L,
                                                 ep.out + colc, ep.ld_out);
    });
  }
  // every wave is done with its epilogue slice of the LDS before the next tile's LDS-DMA pieces (any wave's) land in it
  if (lid + (int)gridDim.x < nwg) __syncthreads();
  if constexpr (XT) {
    xpre = has_next;
    xs0 = has_next ? sN : 0;
  }
  }  // tile loop
}

// SC / SL: DMA issue schedule of the convolution / linear instantiation.  Measured (profiles/r04j_gemm_dma_schedule_ab.jsonl, 256x320 tile, two
// runs each): the linear GEMMs of the SDXL forward gain 2-3 % from schedule 1 (993 -> 1014, 847 -> 870 GEGLU, 1186 -> 1225 at K = 5120, 1042 ->
// 1065 TFLOP/s), the implicit-GEMM convolutions (longer address arithmetic per piece) are level or 1 % slower -> linear 1, convolution 0.
// the two LayerNorm-folding instantiations: linear GEMMs on the 256 x 320 tile only
// grid of a launch: one persistent workgroup per CU (FMX_GEMM_PERSIST=0: one workgroup per tile, the round-2 launch shape, for A/B)
static int persistent_grid(int tiles) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 256;
    if (!(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)) n = 256;
    const char* e = fmx_knob("FMX_GEMM_PERSIST");
    cus = (e && atoi(e) == 0) ? (1 << 30) : (n & ~7);
  }
  return tiles < cus ? tiles : cus;
}

// FMX_GEMM_XTILE=0: the persistent linear kernels fetch every tile's first K-tile in its own prologue again (A/B of the cross-tile prefetch)
static int xtile_on() {
  static int v = -1;
  if (v < 0) {
    const char* e = fmx_knob("FMX_GEMM_XTILE");
    v = (e && atoi(e) == 0) ? 0 : 1;
  }
  return v;
}

template <int LN, int MF = 32>
int launch_ln(const GemmParams& p, hipStream_t st) {
  using G = Geo<256, 320>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<false, 256, 320, false, 1, LN, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  GemmParams q = p;
  q.xtile = xtile_on();
  q.tiles_m = (p.M + 255) / 256;
  q.tiles_n = (p.nout + 319) / 320;
  hipLaunchKernelGGL((gemm256p_kernel<false, 256, 320, false, 1, LN, MF>), dim3(persistent_grid(q.tiles_m * q.tiles_n)), dim3(512), G::LDS_BYTES, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (256x320, LayerNorm folded)");
  return FMX_OK;
}

// the LayerNorm-consumer query projection with the cross-attention epilogue (fp16 build only)
int launch_ln_xa(const GemmParams& p, hipStream_t st) {
#ifndef FMX_ELEM_BF16
  using G = Geo<256, 320>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<false, 256, 320, false, 1, 2, 16, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  GemmParams q = p;
  q.xtile = 0;
  q.tiles_m = (p.M + 255) / 256;
  q.tiles_n = (p.nout + 319) / 320;
  hipLaunchKernelGGL((gemm256p_kernel<false, 256, 320, false, 1, 2, 16, 0, true>), dim3(persistent_grid(q.tiles_m * q.tiles_n)), dim3(512), G::LDS_BYTES, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (256x320, LayerNorm folded, cross-attention epilogue)");
  return FMX_OK;
#else
  (void)p; (void)st;
  return -1;
#endif
}

// LayerNorm folded into the operand-swapped V^T GEMM: 320 x 256 tile
template <int MF>
int launch_ln_swapped(const GemmParams& p, hipStream_t st) {
  using G = Geo<320, 256>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<false, 320, 256, false, 1, 3, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  GemmParams q = p;
  q.xtile = xtile_on();
  q.tiles_m = (p.M + 319) / 320;
  q.tiles_n = (p.nout + 255) / 256;
  hipLaunchKernelGGL((gemm256p_kernel<false, 320, 256, false, 1, 3, MF>), dim3(persistent_grid(q.tiles_m * q.tiles_n)), dim3(512), G::LDS_BYTES, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (320x256, LayerNorm folded, operand-swapped)");
  return FMX_OK;
}

template <int BM, int BN, bool STATS, int SC = 0, int SL = 1, int MF = 32, int FA = 0>
int launch_bn(const GemmParams& p, bool conv, hipStream_t st) {
  using G = Geo<BM, BN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<true, BM, BN, STATS, SC, 0, MF, FA>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<false, BM, BN, STATS, SL, 0, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    attr_set = true;
  }
  GemmParams q = p;
  q.xtile = xtile_on();
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.nout + BN - 1) / BN;
  const int grid = persistent_grid(q.tiles_m * q.tiles_n);
  if (conv) hipLaunchKernelGGL((gemm256p_kernel<true, BM, BN, STATS, SC, 0, MF, FA>), dim3(grid), dim3(512), G::LDS_BYTES, st, q);
  else hipLaunchKernelGGL((gemm256p_kernel<false, BM, BN, STATS, SL, 0, MF>), dim3(grid), dim3(512), G::LDS_BYTES, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (256-row pipelined)");
  return FMX_OK;
}

}  // namespace

int fmx_launch_gemm256p(const GemmParams& p, bool conv, int bm, int bn, hipStream_t st) {
  static int mf = 0;
  if (!mf) {
    // A/B knob: 32 = v_mfma_f32_32x32x16 K loops everywhere (rounds 1-2); 16 (default) = v_mfma_f32_16x16x32 for the LINEAR GEMMs of the 256 x 320 tile.
    // Same box, SDXL 1024^2 batch 8 (profiles/r08r): 110.76 -> 107.27 ms per step, chip clock of the timed steps 1.91 -> 1.99 GHz.
    const char* e = fmx_knob("FMX_GEMM_MFMA");
    mf = (e && atoi(e) == 32) ? 32 : 16;
  }
  if (p.xa_k) return launch_ln_xa(p, st);
  if (p.scat_ow && (mf != 16 || !conv)) return fmx_set_error(FMX_E_BADARG, "gemm: scattered output rows exist in the 16x16x32 convolution kernels only");
  if (mf == 16) {
    if (p.row_stats) return launch_ln<1, 16>(p, st);
    if (p.ln_partial) return launch_ln<2, 16>(p, st);
    if (p.ln_col_ab) return launch_ln_swapped<16>(p, st);
    static int fa = -1;
    if (fa < 0) {
      const char* e2 = fmx_knob("FMX_CONV_FASTADDR");   // A/B knob: 0 = the general address form for every convolution of the 256-row tiles (round 2); 3 = not for the x2-upsample convolutions (round 4 A/B)
      fa = e2 ? atoi(e2) : 1;
    }
    if (p.scat_ow) {   // a phase convolution of fmx_conv3x3_up2x: the bit-mask address form with scattered output rows (the host sends only eligible launches)
      if (bm == 256 && bn == 320) return p.stats ? launch_bn<256, 320, true, 0, 1, 16, 3>(p, conv, st) : launch_bn<256, 320, false, 0, 1, 16, 3>(p, conv, st);
      if (bm == 256 && bn == 256) return p.stats ? launch_bn<256, 256, true, 0, 1, 16, 3>(p, conv, st) : launch_bn<256, 256, false, 0, 1, 16, 3>(p, conv, st);
      return -1;
    }
    if (fa && conv && p.c1 == 0 && p.up_h == 0 && p.kh <= 3) {   // single source, no resize-on-load: the bit-mask address form
      if (bm == 256 && bn == 320) return p.stats ? launch_bn<256, 320, true, 0, 1, 16, 1>(p, conv, st) : launch_bn<256, 320, false, 0, 1, 16, 1>(p, conv, st);
      if (bm == 256 && bn == 256) return p.stats ? launch_bn<256, 256, true, 0, 1, 16, 1>(p, conv, st) : launch_bn<256, 256, false, 0, 1, 16, 1>(p, conv, st);
    }
    // single source, x2 nearest upsample on load, 3x3 / stride 1 / pad 1 (the Upsample convolutions of the UNet decoder and the VAE decoder): FA = 2
    if (fa && fa != 3 && conv && p.c1 == 0 && p.up_h == 2 * p.h && p.up_w == 2 * p.w && p.kh == 3 && p.stride == 1 && p.pad == 1 && p.oh == p.up_h && p.ow == p.up_w) {
      if (bm == 256 && bn == 320) return p.stats ? launch_bn<256, 320, true, 0, 1, 16, 2>(p, conv, st) : launch_bn<256, 320, false, 0, 1, 16, 2>(p, conv, st);
      if (bm == 256 && bn == 256) return p.stats ? launch_bn<256, 256, true, 0, 1, 16, 2>(p, conv, st) : launch_bn<256, 256, false, 0, 1, 16, 2>(p, conv, st);
    }
    if (bm == 256 && bn == 320) return p.stats ? launch_bn<256, 320, true, 0, 1, 16>(p, conv, st) : launch_bn<256, 320, false, 0, 1, 16>(p, conv, st);
    if (bm == 320 && !p.stats) return launch_bn<320, 256, false, 0, 1, 16>(p, conv, st);
    if (bm == 512) return p.stats ? launch_bn<512, 128, true, 0, 1, 16>(p, conv, st) : launch_bn<512, 128, false, 0, 1, 16>(p, conv, st);
    if (bm == 256 && bn == 256) return p.stats ? launch_bn<256, 256, true, 0, 1, 16>(p, conv, st) : launch_bn<256, 256, false, 0, 1, 16>(p, conv, st);
  }
  if (p.row_stats) return launch_ln<1>(p, st);
  if (p.ln_partial) return launch_ln<2>(p, st);
  if (p.ln_col_ab) return launch_ln_swapped<32>(p, st);
  static int sched = -1;
  if (sched < 0) {
    const char* e = fmx_knob("FMX_GEMM_SCHED");   // A/B knob (tools/bench_kernels.py gemmsched): force one DMA issue schedule on the 256x320 tile
    sched = e ? atoi(e) + 1 : 0;
  }
  if (sched && bm == 256 && bn == 320 && !p.stats)
    return sched == 1 ? launch_bn<256, 320, false, 0, 0>(p, conv, st) : sched == 2 ? launch_bn<256, 320, false, 1, 1>(p, conv, st)
                                                                                    : launch_bn<256, 320, false, 2, 2>(p, conv, st);
  if (bm == 320) return launch_bn<320, 256, false>(p, conv, st);
  if (bm == 512) return p.stats ? launch_bn<512, 128, true>(p, conv, st) : launch_bn<512, 128, false>(p, conv, st);
  if (p.stats) return bn == 320 ? launch_bn<256, 320, true>(p, conv, st) : launch_bn<256, 256, true>(p, conv, st);   // output statistics: 256-row tiles only
  return bn == 320 ? launch_bn<256, 320, false>(p, conv, st) : launch_bn<256, 256, false>(p, conv, st);
}
