// 256 x 160 linear GEMM tile for gfx950 with TWO CO-RESIDENT 4-WAVE WORKGROUPS PER CU ("gemm4w", round 4).
//
// Why it exists: in the 8-wave kernels of fmx_gemm256p.hip one workgroup owns the CU -- 160 accumulators per
// lane, all 160 KB of LDS -- so while it runs its epilogue (7-11 us of a 57 us K = 1280 tile: LDS transposes, residual loads, the 42 MB output
// burst) the matrix pipes of that CU are idle, and every wave of the CU waits at the same barriers.  Here a CU holds two INDEPENDENT workgroups:
// each has its own barrier, its own LDS stages and its own tile, each puts ONE wave on every SIMD (the second wave of the SIMD belongs to the
// other workgroup), so one workgroup's prologue, barrier skew, epilogue and store burst run beside the other's K loop.
//
//   tile 256 x 160, waves 4 (M) x 1 (N): a wave owns 64 rows x 160 columns = the SAME wave tile as the 256 x 320 kernel (4 x 10 accumulator blocks
//   of 16 x 16, v_mfma_f32_16x16x32, weights as MFMA-A, activations as MFMA-B), so the epilogue row pass (fmx_gemm_epi.hpp), the GEGLU pairing and the
//   LayerNorm folds (producer LN = 1, consumer LN = 2) are the ones of that kernel and the two kernels interoperate on the same `row_stats` arrays.
//   <= 256 registers (launch bound 2 waves per SIMD).
//
//   LDS: a workgroup may use 80 KB.  Two stages of a 64-deep K-tile are (256 + 160) x 128 B x 2 = 104 KB -- too much; so the K-tile is 32 deep (one
//   k-step of the 16x16x32 loop, 64-byte LDS rows) and the ring has THREE stages of 26 KB = 78 KB: the LDS-DMA pieces of K-tile t + 3 are issued
//   behind the barrier of K-tile t into the stage that barrier released and have two K-tiles (~1.3 us with both workgroups on the pipes) to land.
//   64-byte rows: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 3) (source side of the DMA and ds_read side; see lds_off64).  Fragment i of an operand is 1 KiB behind fragment 0 (16 rows do not change the key).
//
//   K-tile t:   MFMAs 0..14 (nothing rides behind them: the fragment re-reads issued last complete under them)
//               s_waitcnt vmcnt(pieces of one K-tile) lgkmcnt(0) | s_barrier      -> K-tile t + 1 has landed for every wave, stage t % 3 is free
//               MFMAs 15..39, each fragment re-read for K-tile t + 1 right behind its last use, one LDS-DMA piece of K-tile t + 3 behind every third
//
// What it costs: (256 + 160) / (256 x 160) = 1.44 x the L2 -> LDS bytes per FLOP of the 256 x 320 tile, and twice the barriers per K (each over 4
// waves instead of 8).  Whether the overlap pays for that is a measurement: tools/bench_kernels.py gemm4w, profiles/r10_*.
#include <stdlib.h>

#include "fmx_gemm_epi.hpp"

namespace {

constexpr int BM = 256, BN = 160, BKT = 32;
constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;   // 16 KB + 10 KB
constexpr int NSTAGE = 3;
constexpr int SCRATCH = NSTAGE * STAGE_BYTES;                                           // 1 KiB nobody reads (see b_last)
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES + 1024;                                  // 80 896 <= 81 920: two workgroups per CU
constexpr int MIB = 4, NJB = 10;        // 16 x 16 accumulator blocks of a wave: 64 rows x 160 columns
constexpr int NJ = 5;                   // the same in 32-column units (the epilogue's constants)
constexpr int WROWS = 64;
constexpr int NPA = 4, NPB = 3;         // LDS-DMA pieces (16 rows x 64 B) per wave and K-tile: A 16 / 4 waves; B 10: piece s * 4 + wave < 10
constexpr int XSL = NJ * 2048;          // a wave's transpose slice of the epilogue: 16 rows x 160 fp32
static_assert(4 * XSL <= LDS_BYTES, "epilogue slices fit the stages");

// key (row >> 1) & 3: the one of the candidate keys that tools/ubench/lds_b128_patterns.hip times as conflict-free for a 16-row fragment read; the first
// version of this file keyed on (row >> 2) & 3 and ran with SQ_LDS_BANK_CONFLICT = 0.46 of SQ_LDS_IDX_ACTIVE (profiles/r10b_pmc_gemm4w_vs_256x320.json)
__device__ __forceinline__ int lds_off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// LN: 0 plain, 1 LayerNorm producer (per-row statistics of the stored output), 2 LayerNorm consumer (incl. GEGLU) -- fmx_gemm_epi.hpp
template <int LN>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = wave row wm; one wave column
  const int nwg = p.tiles_m * p.tiles_n;
  // tile order: XCD remap, then groups of 8 tile rows (an XCD's 64 concurrent tiles = 8 row blocks x up to 8 column blocks share its L2)
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
  for (int lid = blockIdx.x; lid < nwg; lid += gridDim.x) {   // persistent: a workgroup walks the launch's tiles
    // (the lane id passes through an opaque copy once per tile: nothing per-lane is hoisted across the K loop, see fmx_gemm256p.hip)
    int lane_it = tid & 63;
    asm volatile("" : "+v"(lane_it));
    const int lane = lane_it;
    const int l16 = lane & 15, kg = lane >> 4;
    int tm, tn;
    tile_origin(lid, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging: a piece = 16 rows x 64 B; lane -> row lane / 4, physical chunk lane & 3, logical chunk = physical ^ ((row >> 1) & 3) -----------
    const int r16 = lane >> 2;
    const unsigned kcb = (unsigned)((lane & 3) ^ ((lane >> 3) & 3)) * 16u;   // (row >> 1) & 3 with row = 16 k + lane / 4
    constexpr unsigned OOB = 0xC0000000u;   // beyond num_records: the hardware writes zeros into LDS (tools/ubench/oob_probe.hip)
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a0), 0, p.a0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.wgt), 0, p.w_bytes, 0x00020000);
    unsigned a_off[NPA], b_off[NPB];
#pragma unroll
    for (int s = 0; s < NPA; ++s) {
      const int m = m0 + (s * 4 + wave) * 16 + r16;
      a_off[s] = (m < p.M) ? (unsigned)m * (unsigned)p.s0 * 2u + kcb : OOB;
    }
#pragma unroll
    for (int s = 0; s < NPB; ++s) {
      const int nn = n0 + (s * 4 + wave) * 16 + r16;
      b_off[s] = (nn < p.nout && (s * 4 + wave) < BN / 16) ? (unsigned)nn * (unsigned)p.ldw * 2u + kcb : OOB;
    }
    // B is 10 pieces for 4 waves: in the third round only waves 0 and 1 have one.  Waves 2 and 3 issue theirs all the same -- offset out of range (no
    // memory traffic, zeros) into a spare KiB of LDS -- so that every wave has NPA + NPB pieces per K-tile: one vmcnt bookkeeping, and no branch inside
    // the scheduled MFMA / LDS / DMA interleave of the K loop
    const bool b_last = wave < (BN / 16 - 8);   // uniform
    const int kt = p.kt * (FMX_BK / BKT);        // K-tiles of 32
    // piece IDX of K-tile t into stage byte offset `sto`; K-tiles past the end are zero fills (no traffic) as well
    auto issue_piece = [&](auto IDX, int t, unsigned sto) {
      constexpr int idx = decltype(IDX)::value;
      const bool live = t < kt;
      const unsigned soff = (unsigned)t * (BKT * 2u);
      if constexpr (idx < NPA) {
        auto* dst = (__attribute__((address_space(3))) void*)(smem + sto + (idx * 4 + wave) * 1024);
        const unsigned ao = a_off[idx];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, dst, 16, live ? ao : OOB, soff, 0, 0);
      } else if constexpr (idx < NPA + NPB) {
        constexpr int s = idx - NPA;
        const unsigned at = (s < NPB - 1 || b_last) ? sto + A_BYTES + (s * 4 + wave) * 1024 : (unsigned)SCRATCH;   // uniform select
        auto* dst = (__attribute__((address_space(3))) void*)(smem + at);
        const unsigned bo = b_off[s];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, dst, 16, live ? bo : OOB, soff, 0, 0);
      }
    };
    auto issue_tile = [&](int t, unsigned sto) { static_for<NPA + NPB>([&](auto I) { issue_piece(I, t, sto); }); };

    // LN consumer: one word of this wave's row record ahead of the K loop (pulls the line into this CU's cache; see fmx_gemm256p.hip)
    float ln_touch = 0.f;
    const float* ln_row = nullptr;
    if (LN == 2) {
      const int mrow = min(m0 + wave * WROWS + lane, p.M - 1);
      ln_row = p.ln_partial + (long)mrow * (p.ln_parts * 2);
      ln_touch = ln_row[0];
    }

    f32x4 acc[MIB][NJB];
#pragma unroll
    for (int i = 0; i < MIB; ++i)
#pragma unroll
      for (int j = 0; j < NJB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-tiles 0, 1, 2 in flight; wait for K-tile 0 ------------------------------------------------------------------------------------
    issue_tile(0, 0u);
    issue_tile(1, (unsigned)STAGE_BYTES);
    issue_tile(2, 2u * STAGE_BYTES);
    asm volatile("s_waitcnt vmcnt(14)" ::: "memory");   // 2 x (NPA + NPB) may still fly
    __builtin_amdgcn_s_barrier();

    f16x8 a4[MIB], w1[NJB];
    const unsigned abase = (unsigned)lds_off64(wave * WROWS + l16, kg);
    const unsigned wbase = (unsigned)(A_BYTES + lds_off64(l16, kg));
    auto rd = [&](unsigned addr, int frag) { return *reinterpret_cast<const f16x8*>(smem + addr + frag * 1024); };
#pragma unroll
    for (int i = 0; i < MIB; ++i) a4[i] = rd(abase, i);
#pragma unroll
    for (int j = 0; j < NJB; ++j) w1[j] = rd(wbase, j);

    // MFMAs [LO, HI) of the K-tile.  Order (as the 16x16x32 loop of fmx_gemm256p.hip): two halves of the weight fragments, activation-major inside a half;
    // a fragment is re-read for the NEXT K-tile right behind its last MFMA of this one: w[j < 5] behind the last activation row of half A (MFMAs 15-19),
    // a[i] behind its group of half B, w[j >= 5] behind the last row of half B.  One LDS-DMA piece behind every GAP-th MFMA from PM0 on.
    constexpr int HJ = NJB / 2, HM = MIB * HJ, NM = 2 * HM, MB = HM - HJ, GAP = 3, NPW = NPA + NPB;
    static_assert(MB + 1 + GAP * (NPW - 1) < NM, "the pieces fit behind the barrier");
    auto kpart = [&](auto LOC, auto HIC, unsigned aaddr, unsigned waddr, auto&& piece, auto NPIECES, auto PM0C) {
      constexpr int LO = decltype(LOC)::value, HI = decltype(HIC)::value, npieces = decltype(NPIECES)::value, PM0 = decltype(PM0C)::value;
      static_for<HI - LO>([&](auto MC) {
        constexpr int m = LO + decltype(MC)::value;
        constexpr bool hb = m >= HM;
        constexpr int mm = hb ? m - HM : m;
        constexpr int i = mm / HJ, j = (hb ? HJ : 0) + mm % HJ;
        acc[i][j] = FMX_MFMA_16x16x32(w1[j], a4[i], acc[i][j]);
        if constexpr (LO >= MB) {
          if constexpr (i == MIB - 1) w1[j] = rd(waddr, j);
          if constexpr (hb && mm % HJ == HJ - 1) a4[i] = rd(aaddr, i);
        }
        if constexpr (npieces > 0 && m >= PM0 && ((m - PM0) % GAP) == 0 && (m - PM0) / GAP < npieces) piece(IC<(m - PM0) / GAP>{});
      });
      static_for<HI - LO>([&](auto MC) {
        constexpr int m = LO + decltype(MC)::value;
        constexpr bool hb = m >= HM;
        constexpr int mm = hb ? m - HM : m;
        constexpr int i = mm / HJ;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (LO >= MB) {
          if constexpr (i == MIB - 1 && hb && mm % HJ == HJ - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if constexpr (i == MIB - 1 || (hb && mm % HJ == HJ - 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (npieces > 0 && m >= PM0 && ((m - PM0) % GAP) == 0 && (m - PM0) / GAP < npieces) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      });
    };
    unsigned st_cur = 0u, st_nxt = (unsigned)STAGE_BYTES;   // byte offsets of stage t % 3 and (t + 1) % 3
    static_assert(NPW == 7, "the vmcnt immediates above and below are NPW and 2 NPW");
    for (int t = 0; t < kt; ++t) {
      auto nopiece = [&](auto) {};
      auto piece3 = [&](auto IDX) { issue_piece(IDX, t + 3, st_cur); };
      __builtin_amdgcn_sched_barrier(0);
      kpart(IC<0>{}, IC<MB>{}, 0u, 0u, nopiece, IC<0>{}, IC<0>{});
      __builtin_amdgcn_sched_barrier(0);
      // this wave: its reads of stage t % 3 are complete, its pieces of K-tile t + 1 have landed (those of t + 2 may still fly) -> the K-tile's barrier
      asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      kpart(IC<MB>{}, IC<NM>{}, abase + st_nxt, wbase + st_nxt, piece3, IC<NPW>{}, IC<MB + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      st_cur = st_nxt;
      st_nxt = st_nxt == 2u * STAGE_BYTES ? 0u : st_nxt + STAGE_BYTES;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the tail's zero fills and the last (unused) re-reads: the LDS is about to change hands
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: the row pass of the 256 x 320 kernel on this wave's 64 x 160 sub-tile, 16 rows at a time through its LDS slice --------------------------
    char* my = smem + wave * XSL;
    const FastEpilogue ep(p);
    const bool geglu = LN != 1 && p.act == FMX_ACT_GEGLU;
    float lnm = 0.f, lnr = 1.f;
    if (LN == 2) {   // mean / rstd of row `lane` of this wave's rows from the producer's partial sums (fixed order)
      float s1 = ln_touch, s2 = 0.f;
      const float* q = ln_row;
      s2 += q[1];
      for (int k2 = 1; k2 < p.ln_parts; ++k2) {
        s1 += q[2 * k2];
        s2 += q[2 * k2 + 1];
      }
      lnm = s1 * p.ln_inv_c;
      lnr = rsqrtf(fmaxf(s2 * p.ln_inv_c - lnm * lnm, 0.f) + p.ln_eps);
      if (p.ln_ab_out && tn == 0) {
        const int mrow = m0 + wave * WROWS + lane;
        if (mrow < p.M) *reinterpret_cast<f32x2*>(p.ln_ab_out + (long)mrow * 2) = f32x2{lnr, -lnm * lnr};
      }
    }
    if (!geglu) {
      constexpr int RB = NJ * 128, LPR = NJ * 4;
      const int cg = lane % LPR;
      const int nb = n0 + cg * 8;
      const bool nok = nb < ep.nout;
      const int nbc = nok ? nb : 0;
      float st[1] = {0.f};
#pragma unroll
      for (int i = 0; i < MIB; ++i) {
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
          const int chunk = j * 4 + kg;
          *reinterpret_cast<f32x4*>(my + l16 * RB + ((chunk ^ (l16 & 7)) << 4)) = acc[i][j];
        }
        const int mbase = m0 + wave * WROWS + i * 16;
#define FMX_EPI_ARGS my, lane, mbase, p.M, nbc, nok, ep.per_img, ep.alpha, (float)ep.mgt, ep.bias + nbc * ep.mb, ep.rowvec + nbc * ep.mrv, ep.ld_rv, \
                     ep.gate + nbc * ep.mgt, ep.ld_gt, ep.res + nbc * ep.mres, ep.ld_res, ep.out + nbc, ep.ld_out, st
        if (LN == 1) epi_rows<RB, LPR, 16, 7, false, 1, false, 1>(FMX_EPI_ARGS, p.row_stats + (long)tn * 2, p.tiles_n * 2);
        else if (LN == 2) epi_rows<RB, LPR, 16, 7, false, 2, false, 2>(FMX_EPI_ARGS, nullptr, 0, lnm, lnr, i * 16, p.ln_colsum + nbc);
        else if (ep.gelu_tanh) epi_rows<RB, LPR, 16, 7, true, 0>(FMX_EPI_ARGS);
        else if (ep.mrv | ep.mgt) epi_rows<RB, LPR, 16, 7, false, 0>(FMX_EPI_ARGS);
        else if (ep.mres) epi_rows<RB, LPR, 16, 7, false, 1>(FMX_EPI_ARGS);
        else epi_rows<RB, LPR, 16, 7, false, 2>(FMX_EPI_ARGS);
#undef FMX_EPI_ARGS
      }
    } else {
      // GEGLU: block 2 jj holds the 16 values of weight-row group jj, block 2 jj + 1 their gates (same lane: columns kg * 4 + [0, 4)); staged row = 80 outputs,
      // two halves of 32 rows through the 10 KB slice
      constexpr int RB = NJ * 64, LPR = NJ * 2, ROWS = 32, IH = MIB / 2;
      static_assert(ROWS * RB <= XSL, "a GEGLU half fits the slice");
      float gm[MIB], gr[MIB];
#pragma unroll
      for (int i = 0; i < MIB; ++i) {
        gm[i] = LN == 2 ? __shfl(lnm, i * 16 + l16) : 0.f;
        gr[i] = LN == 2 ? __shfl(lnr, i * 16 + l16) : 1.f;
      }
      static_for<2>([&](auto HALF) {
        constexpr int h = decltype(HALF)::value, I0 = h * IH;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int nb = n0 + j * 32 + kg * 4;
          const int nbc = nb < ep.nout ? nb : 0;
          const f16x4 bv = ep.bias4(nbc), bg = ep.bias4(nbc + 16);
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
            const float sc = LN == 2 ? ep.alpha * gr[i] : ep.alpha;
            const float mr = gm[i] * gr[i];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float cv = bvf[r], cgt = bgf[r];
              if (LN == 2) { cv = fmaf(-mr, sv[r], cv); cgt = fmaf(-mr, sg[r], cgt); }
              const float val = fmaf(acc[i][2 * j][r], sc, cv);
              const float gate = fmaf(acc[i][2 * j + 1][r], sc, cgt);
              o[r] = val * gelu_erf_f(gate);
            }
            const int row = (i - I0) * 16 + l16, chunk = j * 4 + kg;
            *reinterpret_cast<f32x4*>(my + row * RB + ((chunk ^ (row & 3)) << 4)) = o;
          }
        }
        const int cg = lane % LPR;
        const int col = (n0 >> 1) + cg * 8;
        const bool nok = col < ep.ncols;
        const int colc = nok ? col : 0;
        const int mb = m0 + wave * WROWS + h * ROWS;
        if (ep.mres) epi_rows<RB, LPR, ROWS, 3, false, 1>(my, lane, mb, p.M, colc, nok, ep.per_img, 1.0f, 0.0f, p.zp, p.zp, 0L, p.zp, 0L,
                                                           ep.res + colc, ep.ld_res, ep.out + colc, ep.ld_out);
        else epi_rows<RB, LPR, ROWS, 3, false, 2>(my, lane, mb, p.M, colc, nok, ep.per_img, 1.0f, 0.0f, p.zp, p.zp, 0L, p.zp, 0L, p.zp, 0L,
                                                   ep.out + colc, ep.ld_out);
      });
    }
    // every wave is done with its slice before the next tile's LDS-DMA pieces (any wave's) land in it
    if (lid + (int)gridDim.x < nwg) __syncthreads();
  }
}

template <int LN>
int launch4w(const GemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<LN>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  static int slots = 0;
  if (!slots) {
    int dev = 0, n = 256;
    if (!(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8)) n = 256;
    slots = 2 * (n & ~7);   // two workgroups per CU
  }
  GemmParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.nout + BN - 1) / BN;
  const int tiles = q.tiles_m * q.tiles_n;
  // development knob FMX_GEMM_4W_SOLO=1: ask for more LDS than half a CU's, so that ONE workgroup -- one wave per SIMD -- is resident per CU: how busy can a
  // single 4-wave group keep the matrix pipes (the number any scheme with one MFMA wave per SIMD depends on)
  static int solo = -1;
  if (solo < 0) {
    const char* e = fmx_knob("FMX_GEMM_4W_SOLO");
    solo = e ? atoi(e) : 0;
    if (solo) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<LN>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  }
  const int lds = solo ? 100 * 1024 : LDS_BYTES;
  const int cap = solo ? slots / 2 : slots;
  hipLaunchKernelGGL((gemm4w_kernel<LN>), dim3(tiles < cap ? tiles : cap), dim3(256), lds, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16 (256x160, two workgroups per CU)");
  return FMX_OK;
}

}  // namespace

// linear GEMMs only (single source, no statistics); p.row_stats -> LayerNorm producer, p.ln_partial -> LayerNorm consumer
int fmx_launch_gemm4w(const GemmParams& p, hipStream_t st) {
  if (p.row_stats) return launch4w<1>(p, st);
  if (p.ln_partial) return launch4w<2>(p, st);
  return launch4w<0>(p, st);
}
