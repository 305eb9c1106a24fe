// Fused multi-head attention (flash-style, online softmax) for gfx950 -- replaces `attention_function`
// (backend/attention.py:324-339 SDPA / :37-93 explicit form): O = softmax(Q K^T / sqrt(d)) V per (batch, head).
//
// Layout trick that removes every cross-lane shuffle / LDS round trip from the softmax->PV hand-off:
//   * scores are computed TRANSPOSED, S^T = K Q^T, with v_mfma_f32_32x32x16_f16 (A = K rows, B = Q rows):
//     a lane then owns ONE query (column lane&31) and 16 of the tile's 32 keys in its 16 accumulators.
//   * the order in which keys are fed to MFMA rows is free, so K rows are read through the permutation
//     pi(i) = (i&3) | ((i>>3)&3)<<2 | ((i>>2)&1)<<4; with it, accumulator r of lane-half `hi` is key 16*hi + r,
//     i.e. each lane holds 16 CONSECUTIVE keys -- exactly the k-slot order the next MFMA wants.
//   * the output is accumulated TRANSPOSED too, O^T = V^T P^T (A = V^T rows = head-dim, B = P^T): the P
//     registers are used as the B operand as they are (packed to fp16), and the O^T accumulator again has
//     "lane = query", so the online-softmax rescale and the final 1/l are lane-local multiplies.
//   * V is consumed as V^T [d][key]; the projection GEMM produces it directly by swapping its operands
//     (V^T = Wv X^T is the same NT GEMM), so no transpose kernel and no ds_read_tr is needed.
// K and V^T tiles (64 keys) are shared by the 4 waves of a workgroup (4 x 32 queries), staged by LDS-DMA,
// double buffered, swizzled for conflict-free ds_read_b128.  Row sums stay per-lane partials (the
// cross-half add happens once at the end); the accumulator rescale is skipped wave-uniformly when no
// running max moved.
#include <stdlib.h>

#include "fmx_common.hpp"

namespace {

struct AttnParams {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* o;
  long q_bs, q_rs, k_bs, k_rs, vt_bs, vt_hs, vt_ds, o_bs, o_rs;
  int batch, heads, nq, nk, nk_pad, qtiles;
  float scale_log2e;
  int causal;
  const f16* zp;
  unsigned q_span;            // attn_short2_kernel: bytes addressable from a (batch, head)'s Q base
  unsigned k_span, vt_span;   // bytes addressable from a (batch, head)'s K / V^T base (q64v2: buffer descriptors), 0 = do not use
  const f16* mask;            // optional additive mask, element (b, h, i, j) at mask[b*mask_bs + h*mask_hs + i*mask_qs + j] (strides may be 0)
  long mask_bs, mask_hs, mask_qs;
  float inv_scale;            // 1 / scale: the mask is added to the UNSCALED score
  int nfull, nsplit;          // q64v2: workgroups [0, nfull) own 256 queries x all keys; [nfull, nfull + nsplit) are key-split (see the kernel)
  int bid0;            // first workgroup index of this launch (0 except for a key-split tail launched on its own)
  int sk_tpw, sk_chunks;      // attn_short_kernel: 128-query tiles per persistent workgroup, workgroups per (batch, head)
};

constexpr int KVB = 64;  // keys per tile
#ifndef FMX_ATTN_WS_ROT_DEFAULT
#define FMX_ATTN_WS_ROT_DEFAULT 1   // the wave-specialised kernel's rotated sub-tile pipeline.  Measured (profiles/r44_*, r45_*): Flux shape 568 -> 562 us eager,
                                    // PMC 811 k -> 784 k cycles (-3.4 %) at 1.87 -> 1.83 GHz: matrix pipes 54.7 -> 56.9 % busy, busy x GHz 1.02 -> 1.04 -- the kernel sits
                                    // on the chip's power line, a denser schedule is answered with a lower clock; +1-3 % on every d = 128 shape, so it stays on
                                    // (FMX_ATTN_WS_ROT=0 restores round 2's order)
#endif

template <int V>
struct IC { static constexpr int value = V; };

template <int CPR>
__device__ __forceinline__ int k_phys_chunk(int row, int c) {
  if (CPR == 8) return c ^ ((row >> 1) & 7);
  if (CPR == 16) return c ^ (row & 15);
  if (CPR == 20) { int x = c + (row >> 2); return x >= 20 ? x - 20 : (x); }
  return c;
}
template <int CPR>
__device__ __forceinline__ int k_logical_chunk(int row, int pc) {
  if (CPR == 8) return pc ^ ((row >> 1) & 7);
  if (CPR == 16) return pc ^ (row & 15);
  if (CPR == 20) { int x = pc - ((row >> 2) % 20); return x < 0 ? x + 20 : x; }
  return pc;
}

__device__ __forceinline__ int key_perm(int i) { return (i & 3) | (((i >> 3) & 3) << 2) | (((i >> 2) & 1) << 4); }

template <int DP>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DSTEPS = DP / 16;        // QK^T MFMAs per 32-key sub-tile
  constexpr int DVT = (DP + 31) / 32;    // 32-row tiles of O^T
  constexpr int CPR = DP / 8;            // 16-byte chunks per K row
  constexpr int KBYTES = KVB * DP * 2;
  constexpr int VBYTES = DVT * 32 * 128;
  constexpr int STAGE = KBYTES + VBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, li = lane & 31;

  const int nwg = p.qtiles * p.heads * p.batch;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int qt = wg % p.qtiles;
  const int bh = wg / p.qtiles;
  const int h = bh % p.heads, b = bh / p.heads;

  const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;

  // ---- Q fragments (B operand): lane = query li, k-slots hi*8.. of each 16-wide d step -----------------------
  const int q0 = qt * 128 + wave * 32;
  const int qrow = min(q0 + li, p.nq - 1);
  const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + (long)h * DP + hi * 8;
  f16x8 qf[DSTEPS];
#pragma unroll
  for (int ds = 0; ds < DSTEPS; ++ds) qf[ds] = *reinterpret_cast<const f16x8*>(qp + ds * 16);

  const f16* zp = p.zp;
  auto stage = [&](int s, int kt) {
    char* sk = smem + s * STAGE;
    char* sv = sk + KBYTES;
    const int key0 = kt * KVB;
    // K tile: 64 rows x CPR chunks, linear chunk id q = ii*64 + lane
#pragma unroll
    for (int ii = wave; ii < CPR; ii += 4) {
      const int qq = ii * 64 + lane;
      const int row = qq / CPR, pc = qq - row * CPR;
      const int c = k_logical_chunk<CPR>(row, pc);
      glds16(kbase + (long)(key0 + row) * p.k_rs + c * 8, sk + ii * 1024);
    }
    // V^T tile: DVT*32 rows (head dim) x 8 chunks (64 keys)
#pragma unroll
    for (int ii = wave; ii < DVT * 4; ii += 4) {
      const int qq = ii * 64 + lane;
      const int row = qq >> 3, pc = qq & 7;
      const int c = pc ^ ((row >> 1) & 7);
      const f16* g = (row < DP) ? vbase + (long)row * p.vt_ds + key0 + c * 8 : zp + c * 8;
      glds16(g, sv + ii * 1024);
    }
  };

  f32x16 oacc[DVT];
#pragma unroll
  for (int i = 0; i < DVT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (p.nk + KVB - 1) / KVB;
  stage(0, 0);
  wait_vmcnt0();
  __syncthreads();

  const int krow = key_perm(li);
  const float c2 = p.scale_log2e;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
    const char* sk = smem + cur * STAGE;
    const char* sv = sk + KBYTES;

    // ---- S^T = K Q^T for the two 32-key sub-tiles ---------------------------------------------------------
    f32x16 sacc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[s][r] = 0.f;
      const int row = s * 32 + krow;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
        sacc[s] = FMX_MFMA_32x32x16(kf, qf[ds], sacc[s]);
      }
    }
    // lane now holds scores of query li against keys kt*64 + s*32 + hi*16 + r
    if ((kt + 1) * KVB > p.nk) {  // ragged tail: mask keys >= nk (wave-uniform branch)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * KVB + s * 32 + hi * 16 + r >= p.nk) sacc[s][r] = -INFINITY;
    }
    if (p.mask) {  // additive mask (attention_function's `mask`: 0 / -inf from a bool mask, or arbitrary biases), 16 consecutive keys per lane
      const f16* mp = p.mask + (long)b * p.mask_bs + (long)h * p.mask_hs + (long)min(q0 + li, p.nq - 1) * p.mask_qs + kt * KVB + hi * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f16x8 m0 = *reinterpret_cast<const f16x8*>(mp + s * 32), m1 = *reinterpret_cast<const f16x8*>(mp + s * 32 + 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[s][r] += (float)(r < 8 ? m0[r & 7] : m1[r & 7]) * p.inv_scale;
      }
    }
    if (p.causal && (kt + 1) * KVB - 1 > q0) {  // causal mask (CLIP text encoder): key j > query i never attends; the first key
      const int qi = q0 + li;                   // tile always holds key 0 <= i, so the running max is finite from tile 0 on
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * KVB + s * 32 + hi * 16 + r > qi) sacc[s][r] = -INFINITY;
    }
    float mx = sacc[0][0];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[s][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
    const float mc = m_new * c2;
    float psum = 0.f;
    f16x8 pf[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[s][r] * c2 - mc);
        psum += e;
        pf[s][r >> 3][r & 7] = (f16)e;
      }
    l_run = l_run * alpha + psum;
    if (__any(m_new > m_run)) {
#pragma unroll
      for (int i = 0; i < DVT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    m_run = m_new;

    // ---- O^T += V^T P^T : A = V^T rows (head dim), B = P^T (k-slot hi*8+e <-> key s*32 + hi*16 + j*8 + e) ----
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      const int row = dt * 32 + li;
      const char* rp = sv + row * 128;
      const int sw = (row >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(rp + (((s * 4 + hi * 2 + j) ^ sw) << 4));
          oacc[dt] = FMX_MFMA_32x32x16(vf, pf[s][j], oacc[dt]);
        }
    }
    wait_vmcnt0();
    __syncthreads();
  }

  // ---- finish: 1/l, store O[b][q][h*DP + d]; lane = query, registers = 4-wide runs of d --------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qg = q0 + li;
  if (qg < p.nq) {
    f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + g * 8 + hi * 4;
        if (d < DP) {
          f16x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)(oacc[dt][g * 4 + e] * inv);
          *reinterpret_cast<f16x4*>(op + d) = hv;
        }
      }
  }
}

// ---- d_head 64, 64 queries per wave ------------------------------------------------------------------------------
// Same algorithm and LDS image as attn_kernel<64>, but a wave owns TWO 32-query fragments: every K / V^T fragment read
// from LDS feeds 2 MFMAs and a workgroup (4 waves) covers 256 queries per K/V tile, which halves the LDS reads and the
// LDS-DMA instructions per MFMA -- on MI355X both cost matrix-pipe time (a DMA instruction does not issue while the
// SIMD's other wave streams MFMAs, tools/ubench/dma_rate.hip).  ~215 VGPRs -> 2 waves per SIMD, softmax VALU of one wave
// under the MFMAs of the other.
template <int PRIO>
__global__ __launch_bounds__(256, 2) void attn_q64_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = 64, DSTEPS = 4, DVT = 2, CPR = 8;
  constexpr int KBYTES = KVB * DP * 2;
  constexpr int VBYTES = DVT * 32 * 128;
  constexpr int STAGE = KBYTES + VBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;

  const int nwg = p.qtiles * p.heads * p.batch;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int qt = wg % p.qtiles;
  const int bh = wg / p.qtiles;
  const int h = bh % p.heads, b = bh / p.heads;

  const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;

  const int q0 = qt * 256 + wave * 64;
  f16x8 qf[2][DSTEPS];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int qrow = min(q0 + a * 32 + li, p.nq - 1);
    const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + (long)h * DP + hi * 8;
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) qf[a][ds] = *reinterpret_cast<const f16x8*>(qp + ds * 16);
  }

  // staging: K tile 64 rows x 8 chunks and V^T tile 64 rows x 8 chunks = 2 x 8 KiB, 2 + 2 DMA instructions per wave
  const int srow = lane >> 3, spc = lane & 7;
  auto stage = [&](int s, int kt) {
    char* sk = smem + s * STAGE;
    char* sv = sk + KBYTES;
    const int key0 = kt * KVB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int row = (e * 4 + wave) * 8 + srow;
      const int c = spc ^ ((row >> 1) & 7);
      glds16(kbase + (long)(key0 + row) * p.k_rs + c * 8, sk + (e * 4 + wave) * 1024);
      glds16(vbase + (long)row * p.vt_ds + key0 + c * 8, sv + (e * 4 + wave) * 1024);
    }
  };

  f32x16 oacc[DVT][2];
#pragma unroll
  for (int i = 0; i < DVT; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][a][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  const int ntiles = (p.nk + KVB - 1) / KVB;
  stage(0, 0);
  wait_vmcnt0();
  __syncthreads();

  const int krow = key_perm(li);
  const float c2 = p.scale_log2e;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < ntiles) stage(cur ^ 1, kt + 1);
    const char* sk = smem + cur * STAGE;
    const char* sv = sk + KBYTES;

    f32x16 sacc[2][2];  // [32-key sub-tile][query fragment]
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[s][a][r] = 0.f;
      const int row = s * 32 + krow;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + row * 128 + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
#pragma unroll
        for (int a = 0; a < 2; ++a) sacc[s][a] = FMX_MFMA_32x32x16(kf, qf[a][ds], sacc[s][a]);
      }
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if ((kt + 1) * KVB > p.nk) {  // ragged tail: mask keys >= nk (wave-uniform branch)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * KVB + s * 32 + hi * 16 + r >= p.nk) sacc[s][a][r] = -INFINITY;
    }
    f16x8 pf[2][2][2];  // [query fragment][sub-tile][8-key half]
    bool grew = false;
    float alpha[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float mx = sacc[0][a][0];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[s][a][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run[a], mx);
      alpha[a] = __builtin_amdgcn_exp2f((m_run[a] - m_new) * c2);
      const float mc = m_new * c2;
      float psum = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(sacc[s][a][r] * c2 - mc);
          psum += e;
          pf[a][s][r >> 3][r & 7] = (f16)e;
        }
      l_run[a] = l_run[a] * alpha[a] + psum;
      grew = grew || (m_new > m_run[a]);
      m_run[a] = m_new;
    }
    if (__any(grew)) {
#pragma unroll
      for (int i = 0; i < DVT; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[i][a][r] *= alpha[a];
    }

    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      const int row = dt * 32 + li;
      const char* rp = sv + row * 128;
      const int sw = (row >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(rp + (((s * 4 + hi * 2 + j) ^ sw) << 4));
#pragma unroll
          for (int a = 0; a < 2; ++a) oacc[dt][a] = FMX_MFMA_32x32x16(vf, pf[a][s][j], oacc[dt][a]);
        }
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    wait_vmcnt0();
    __syncthreads();
  }

#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float l_tot = l_run[a] + __shfl_xor(l_run[a], 32);
    const float inv = 1.0f / l_tot;
    const int qg = q0 + a * 32 + li;
    if (qg < p.nq) {
      f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int d = dt * 32 + g4 * 8 + hi * 4;
          f16x4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (f16)(oacc[dt][a][g4 * 4 + e] * inv);
          *reinterpret_cast<f16x4*>(op + d) = hv;
        }
    }
  }
}

// ---- d_head 64, 64 queries per wave, second generation ("q64v2") ------------------------------------------------------
// Same tiling and LDS image as attn_q64_kernel.  What changed is the VALU work per score, because the counters say the loop is
// ISSUE-bound, not matrix-pipe- or exp-bound (profiles/r04a_pmc_*: 9.7 VALU instructions per MFMA -- a 32-cycle MFMA leaves ~7 issue
// slots -- matrix pipe 39 % busy at N = 4096):
//   * Q is pre-multiplied by scale * log2(e) once per workgroup, and each score chain gets one extra rank-2 MFMA k-step that adds
//     minus the running maximum of its query (see `mfrag` below): the matrix pipe hands back  s*c - m  directly -- no
//     multiply-subtract per score (64 v_fma per tile gone, for 4 more MFMAs on a pipe that is 39 % busy).
//   * the running maximum moves only when a tile's scores exceed it by more than THR (log2 units; P stays below 2^THR, fp16 / bf16
//     have the exponent range and the same relative precision there; sums and O are fp32): the O / l rescale, the refresh of the -m operand and
//     the 64 subtractions that re-base an already computed tile run in that rare branch only.  The first tile always takes it
//     (exact maximum, which may be negative: a row whose scores are all far below zero must not underflow).
//   * the two LDS stages are unrolled (compile-time stage -> immediate LDS offsets, no per-tile address arithmetic), K / V^T tiles
//     arrive by buffer_load ... lds with a uniform scalar offset per tile (no per-lane 64-bit addresses), the cross-half maximum
//     exchange is one v_permlane32_swap instead of an LDS bpermute, and the epilogue stores 16 bytes per lane (half-wave swap).
//   * KEY-SPLIT workgroups for the last, partial round of a launch.  A workgroup holds its CU slot for the whole key loop; when the tiles
//     beyond the last full round of S slots (2 per CU) are fewer than the CUs, the launcher turns them into twice as many workgroups of 128
//     queries whose wave PAIRS each walk one half of the keys (own LDS stages per pair) and merge (m, l, O) through LDS at the end: half the
//     duration each, on CUs that would otherwise idle.  Small launches (batch 1) are all key-split.
//   * (round 4) a PERSISTENT form -- workgroups walking the (batch, head, query tile) list, the last key tile of an entry staging the first K / V^T tile of the
//     next one, the next Q rows requested ahead of the O stores -- was built and measured (commit "attn_q64v2: persistent walk ...", profiles/
//     r12a_attention_persistent_walk_negative_result.jsonl): 1024 keys 98.4 -> 99.4 us, 4096 keys 667 -> 723 us.  The walk's loop-carried state took the kernel
//     from 127 + 32 to 249 registers, i.e. from THREE resident workgroups per CU to two, and a static entry assignment replaces the dispatcher's dynamic one;
//     what it saves (one K / V^T + Q round trip per entry, ~2 us of 38) is less than what those cost.  Not kept.
//   * DP = 128 (Flux: 24 heads of 128): the same kernel with 8 k-steps per score chain and four 32-channel output blocks.  64 queries per
//     wave then need ~350 registers (128 accumulator + 64 Q-fragment + 64 score ...), so it runs ONE wave per SIMD on the unified 512-entry
//     file (__launch_bounds__(256, 1)): nothing overlaps across waves, but a K / V^T fragment read still feeds two MFMAs, where the
//     32-query generic kernel it replaces is LDS-read-bound (1 read per MFMA; 531 TFLOP/s in the Flux forward).
// ---- round 5 experiment: a fraction of a tile's exponentials on the PACKED-fp16 vector path instead of the quarter-rate
//      transcendental unit.  exp2 of two scores at once: x (<= THR, clamped at -15) -> t = x + 1536 (fp16 ulp 1 there: t - 1536 = round(x) = n, and the low
//      mantissa bits of t ARE 512 + n) -> f = x - n in [-0.5, 0.5] -> degree-3 minimax 2^f (7.5e-5) in three v_pk_fma_f16 -> the exponent add as
//      bits(poly) + (bits(t) << 10) (512 << 10 wraps to 0 in 16 bits, n << 10 is what is left) -> x a [0, 1] factor clamp(x + 15) that flushes what
//      would have left fp16's normal range.  ~12 full-rate packed instructions per PAIR of scores against two v_exp_f32 (16.5 issue cycles each).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
#ifndef FMX_ELEM_BF16
__device__ __forceinline__ h16x2 pk_exp2(float x0, float x1) {
  h16x2 x = {(_Float16)x0, (_Float16)x1};
  const h16x2 lo = {(_Float16)-15.0f, (_Float16)-15.0f}, magic = {(_Float16)1536.0f, (_Float16)1536.0f};
  const h16x2 zero = {(_Float16)0.0f, (_Float16)0.0f}, one = {(_Float16)1.0f, (_Float16)1.0f}, fifteen = {(_Float16)15.0f, (_Float16)15.0f};
  const h16x2 xc = __builtin_elementwise_max(x, lo);
  const h16x2 t = xc + magic;
  const h16x2 f = xc - (t - magic);
  const h16x2 c0 = {(_Float16)0.99992807f, (_Float16)0.99992807f}, c1 = {(_Float16)0.69326099f, (_Float16)0.69326099f},
              c2 = {(_Float16)0.24261112f, (_Float16)0.24261112f}, c3 = {(_Float16)0.05517162f, (_Float16)0.05517162f};
  h16x2 pl = c3 * f + c2;
  pl = pl * f + c1;
  pl = pl * f + c0;
  const u16x2_t bits = __builtin_bit_cast(u16x2_t, pl) + (__builtin_bit_cast(u16x2_t, t) << (unsigned short)10);
  const h16x2 keep = __builtin_elementwise_min(__builtin_elementwise_max(x + fifteen, zero), one);
  return __builtin_bit_cast(h16x2, bits) * keep;
}
#endif

template <int THR, int DP, int NPOLY = 0>
__global__ __launch_bounds__(256, DP == 64 ? 2 : 1) void attn_q64v2_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DSTEPS = DP / 16, DVT = DP / 32;
  constexpr int CPR = DP / 8;                // 16-byte chunks per K row
  constexpr int KBYTES = KVB * DP * 2;
  constexpr int STAGE = KBYTES + DVT * 32 * 128;
  constexpr int KROWS = 1024 / (DP * 2);     // K rows per 1-KiB LDS-DMA piece: 8 (DP 64) / 4 (DP 128)
  constexpr int NPK = KVB / KROWS / 4;       // K pieces per wave and tile when four waves share it: 2 / 4
  constexpr int NPV = DP / 8 / 4;            // V^T pieces (8 rows of 128 B) per wave: 2 / 4

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;

  const int bid = blockIdx.x + p.bid0;   // bid0 > 0: only the key-split tail of a launch whose full rounds another kernel ran
  const bool split = bid >= p.nfull;   // workgroup-uniform
  int wg, q_in_tile;
  if (!split) {
    wg = xcd_remap(bid, p.nfull);
    q_in_tile = wave * 64;
  } else {
    const int sidx = xcd_remap(bid - p.nfull, p.nsplit);
    wg = p.nfull + (sidx >> 1);
    q_in_tile = (sidx & 1) * 128 + (wave & 1) * 64;
  }
  const int pair = split ? (wave >> 1) : 0;        // which half of the keys (key-split workgroups)
  const int swave = split ? (wave & 1) : wave;     // position among the waves that share a K / V^T tile
  const int sways = split ? 2 : 4;                 // how many waves share one
  const int qt = wg % p.qtiles;
  const int bh = wg / p.qtiles;
  const int h = bh % p.heads, b = bh / p.heads;

  const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  // uniform descriptors; lanes past the extents read zeros (launch_attn_q64 guarantees the spans fit 32 bits)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);

  const int q0 = qt * 256 + q_in_tile;
  // staging: K tile 64 rows x 8 chunks and V^T tile 64 rows x 8 chunks = 2 x 8 KiB, 2 + 2 DMA instructions per wave and tile (4 + 4 in a
  // key-split workgroup, where two waves share a tile); per-lane byte offsets are tile-invariant, the tile's key offset rides in the
  // scalar offset operand
  char* const sbase = smem + pair * (2 * STAGE);   // a wave pair of a key-split workgroup has its own two stages
  // piece e of this wave: K rows (e * sways + swave) * KROWS + lane / CPR, chunk lane % CPR; V^T rows (e * sways + swave) * 8 + lane / 8
  constexpr int MAXP = 2 * (NPK > NPV ? NPK : NPV);   // (twice as many in a key-split workgroup, where two waves share a tile)
  unsigned k_voff[2 * NPK], v_voff[2 * NPV];
#pragma unroll
  for (int e = 0; e < 2 * NPK; ++e) {
    const int row = ((e * sways + swave) * KROWS + lane / CPR) & (KVB - 1);
    k_voff[e] = (unsigned)row * (unsigned)p.k_rs * 2u + (unsigned)k_logical_chunk<CPR>(row, lane % CPR) * 16u;
  }
#pragma unroll
  for (int e = 0; e < 2 * NPV; ++e) {
    const int row = ((e * sways + swave) * 8 + (lane >> 3)) & (DP - 1);
    v_voff[e] = (unsigned)row * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  (void)MAXP;
  const unsigned k_tile = (unsigned)KVB * (unsigned)p.k_rs * 2u;
  auto stage = [&](auto SI, int kt) {
    constexpr int S = decltype(SI)::value;
#pragma unroll
    for (int e = 0; e < 2 * NPK; ++e)
      if (e < NPK || split) {
        auto* dk = (__attribute__((address_space(3))) void*)(sbase + S * STAGE + (e * sways + swave) * 1024);
        const unsigned kv = k_voff[e];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * k_tile, 0, 0);
      }
#pragma unroll
    for (int e = 0; e < 2 * NPV; ++e)
      if (e < NPV || split) {
        auto* dv = (__attribute__((address_space(3))) void*)(sbase + S * STAGE + KBYTES + (e * sways + swave) * 1024);
        const unsigned vv = v_voff[e];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KVB * 2u), 0, 0);
      }
  };

  f32x16 oacc[DVT][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < DVT; ++i) oacc[i][a][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};   // m_run in the log2 domain (scores arrive scaled)
  // "- m" through the matrix pipe: one extra k-step per score chain whose A operand is 1 in k-slots 0 and 1 (every key row alike) and
  // whose B operand carries -m split into an fp16 high and low part in those two slots (residual 2^-22 |m|; products with 1 are exact,
  // the accumulation is fp32).  12 registers instead of a 32-register accumulator-shaped copy of the maxima.
  f16x8 ones, mfrag[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ones[e] = (f16)((hi == 0 && e < 2) ? 1.0f : 0.0f);
    mfrag[0][e] = mfrag[1][e] = (f16)0.0f;
  }

  const int ntiles_all = (p.nk + KVB - 1) / KVB;
  const int ntiles = split ? ntiles_all >> 1 : ntiles_all;   // the launcher splits only an even number of key tiles
  const int kt0 = pair * ntiles;
  stage(IC<0>{}, kt0);   // first: its round trip to HBM overlaps the Q loads below

  const float c2 = p.scale_log2e;
  f16x8 qf[2][DSTEPS];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int qrow = min(q0 + a * 32 + li, p.nq - 1);
    const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + (long)h * DP + hi * 8;
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) {
      const f16x8 raw = *reinterpret_cast<const f16x8*>(qp + ds * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[a][ds][e] = (f16)((float)raw[e] * c2);
    }
  }
  wait_vmcnt0();
  __syncthreads();

  const int krow = key_perm(li);
  auto tile = [&](auto SI, int j) {   // j: position in this wave's key range
    constexpr int S = decltype(SI)::value;
    const int kt = kt0 + j;
    if (j + 1 < ntiles) stage(IC<S ^ 1>{}, kt + 1);
    const char* sk = sbase + S * STAGE;
    const char* sv = sk + KBYTES;

    f32x16 sacc[2][2];  // [32-key sub-tile][query fragment] = score * c - running max
    // Issue order of the score phase, pinned: the four K fragments of sub-tile 0, the two "- maximum" MFMAs, then two MFMAs per fragment
    // with one fragment read of sub-tile 1 behind each pair (its register quad was just released), then sub-tile 1's eight MFMAs.  Left
    // alone the compiler recycles ONE register quad and waits for every read in turn (16 exposed LDS round trips per tile).
    __builtin_amdgcn_sched_barrier(0);
    f16x8 kf[2][DSTEPS];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const int row = s * 32 + krow;
        kf[s][ds] = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
      }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      sacc[0][a] = FMX_MFMA_32x32x16(ones, mfrag[a], z);
      sacc[1][a] = sacc[0][a];
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
        for (int a = 0; a < 2; ++a) sacc[s][a] = FMX_MFMA_32x32x16(kf[s][ds], qf[a][ds], sacc[s][a]);
    __builtin_amdgcn_sched_group_barrier(0x100, DSTEPS, 0);   // DS read x DSTEPS
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);        // MFMA x2 (- maximum)
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * DSTEPS, 0);
    __builtin_amdgcn_sched_barrier(0);
    if ((kt + 1) * KVB > p.nk) {  // ragged tail: mask keys >= nk (wave-uniform branch)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt * KVB + s * 32 + hi * 16 + r >= p.nk) sacc[s][a][r] = -INFINITY;
    }
    // how far does this tile stick out above the running maxima?
    float mx[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float m0 = fmaxf(sacc[0][a][0], sacc[1][a][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) m0 = fmaxf(fmaxf(m0, sacc[0][a][r]), sacc[1][a][r]);
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
      mx[a] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));   // {own, other half-wave's} in some order
    }
    const bool first = j == 0;
    if (first || __any(fmaxf(mx[0], mx[1]) > (float)THR)) {   // wave-uniform; rare after the first few tiles
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        // the first tile sets the maximum exactly (it may be negative); later ones only raise it
        const float delta = first ? mx[a] : fmaxf(mx[a], 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        m_run[a] += delta;
        l_run[a] *= alpha;
        const f16 mh = (f16)(-m_run[a]);
        const f16 ml = (f16)(-m_run[a] - (float)mh);
        mfrag[a][0] = hi == 0 ? mh : (f16)0.0f;
        mfrag[a][1] = hi == 0 ? ml : (f16)0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
          for (int i = 0; i < DVT; ++i) oacc[i][a][r] *= alpha;
#pragma unroll
          for (int s = 0; s < 2; ++s) sacc[s][a][r] -= delta;
        }
      }
    }
    f16x8 pf[2][2][2];  // [query fragment][sub-tile][8-key half]
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifndef FMX_ELEM_BF16
          if (NPOLY > 0 && (r >> 1) < NPOLY) {   // (compile time) the first NPOLY pairs of every 16-key group: packed-fp16 polynomial
            if ((r & 1) == 0) {
              const h16x2 e2 = pk_exp2(sacc[s][a][r], sacc[s][a][r + 1]);
              pf[a][s][r >> 3][r & 7] = (f16)e2[0];
              pf[a][s][r >> 3][(r & 7) + 1] = (f16)e2[1];
              const float es = (float)e2[0] + (float)e2[1];
              if (s == 0) ps0 += es; else ps1 += es;
            }
            continue;
          }
#endif
          const float e = __builtin_amdgcn_exp2f(sacc[s][a][r]);
          // (row sums from the PACKED P through v_dot2c_f32_f16 -- 32 instead of 64 vector instructions per tile, and a denominator made of exactly the
          //  fp16 values the P.V MFMA multiplies -- were measured in round 4 and LOSE: 4096 keys 665 -> 692 us, 1024 keys level, batch 2 114 -> 128 us
          //  (profiles/r10c_attention_dot2_row_sums_negative_result.jsonl); fewer instructions, a longer dependent chain behind the packing.  Not kept.)
          if (s == 0) ps0 += e; else ps1 += e;
          pf[a][s][r >> 3][r & 7] = (f16)e;
        }
      l_run[a] += ps0 + ps1;
    }

    // ---- O^T += V^T P^T, same discipline: four V^T fragments up front, the other four behind the first eight MFMAs ----
    __builtin_amdgcn_sched_barrier(0);
    f16x8 vf[DVT][2][2];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      const int row = dt * 32 + li;
      const char* rp = sv + row * 128;
      const int sw = (row >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) vf[dt][s][j] = *reinterpret_cast<const f16x8*>(rp + (((s * 4 + hi * 2 + j) ^ sw) << 4));
    }
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int a = 0; a < 2; ++a) oacc[dt][a] = FMX_MFMA_32x32x16(vf[dt][s][j], pf[a][s][j], oacc[dt][a]);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int q4 = 0; q4 < DVT * 4 - 4; ++q4) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt0();
    __syncthreads();
  };
  for (int j = 0; j < ntiles; j += 2) {
    tile(IC<0>{}, j);
    if (j + 1 < ntiles) tile(IC<1>{}, j + 1);
  }

  // ---- key-split workgroups: the pair that walked the upper half of the keys hands (O, m, l) to its partner through LDS (the stages are
  //      idle: the last tile ended with a barrier); same code, so lane i holds the same (query, channel) elements in both waves ----------
  if (split) {
    constexpr int NEX = DVT * 32 + 4;   // floats per lane: O, m, l
    float* ex = reinterpret_cast<float*>(smem) + (wave & 1) * (NEX * 64) + lane;
    if (pair == 1) {
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) ex[((dt * 2 + a) * 16 + r) * 64] = oacc[dt][a][r];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        ex[(DVT * 32 + a) * 64] = m_run[a];
        ex[(DVT * 32 + 2 + a) * 64] = l_run[a];
      }
    }
    __syncthreads();
    if (pair == 1) return;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float m1 = ex[(DVT * 32 + a) * 64], l1 = ex[(DVT * 32 + 2 + a) * 64];
      const float m = fmaxf(m_run[a], m1);
      const float f0 = __builtin_amdgcn_exp2f(m_run[a] - m), f1 = __builtin_amdgcn_exp2f(m1 - m);
      l_run[a] = l_run[a] * f0 + l1 * f1;
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][a][r] = oacc[dt][a][r] * f0 + ex[((dt * 2 + a) * 16 + r) * 64] * f1;
    }
  }

  // ---- finish: 1/l, O[b][q][h*64 + d]; lane = query.  A lane holds 4-wide runs of d (registers g*4.., d = dt*32 + g*8 + hi*4); one
  //      half-wave swap per register pair turns two 8-byte pieces into one 16-byte store per lane ------------------------------------
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[a]), __float_as_uint(l_run[a]), false, false);
    const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
    const int qg = q0 + a * 32 + li;
    f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        union { f16x4 h4; unsigned u[2]; } lo, up;   // groups g and g + 1 of this lane
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo.h4[e] = (f16)(oacc[dt][a][g * 4 + e] * inv);
          up.h4[e] = (f16)(oacc[dt][a][(g + 1) * 4 + e] * inv);
        }
        // lanes 0-31 end up with [own group g | upper half's group g] = columns g*8 .. g*8+7; lanes 32-63 with
        // [lower half's group g+1 | own group g+1] = columns (g+1)*8 .. (g+1)*8+7
        const u32x2 x = __builtin_amdgcn_permlane32_swap(lo.u[0], up.u[0], false, false);
        const u32x2 y = __builtin_amdgcn_permlane32_swap(lo.u[1], up.u[1], false, false);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {x[0], y[0], x[1], y[1]};
        if (qg < p.nq) *reinterpret_cast<u32x4*>(op + dt * 32 + (g + hi) * 8) = v;
      }
  }
}

// ---- third generation, d_head 64: the vector work of one 32-key sub-tile issued UNDER the MFMAs of the next one, inside each wave --------
// tools/microbench_pipes.hip (profiles/r07f): one wave issues a 32x32x16 MFMA at most every ~47 cycles (the pipe takes one per 32 when two
// waves feed it), a v_exp_f32 costs a wave ~16 cycles, and both run beside each other -- from two waves and from ONE wave (16 x (MFMA + 2
// v_exp) take what 16 MFMAs take).  In attn_q64v2_kernel a wave's tile is a serial chain  scores (18 MFMA) -> maximum -> 64 exp2 -> P V (16
// MFMA): ~3300 cycles of which the two halves overlap only across the SIMD's two waves.  Here the chain is re-cut at sub-tile granularity
// (sub-tile = 32 keys; scores of 0 / 1 live in separate accumulators anyway, so no extra registers):
//     phase A(t):  scores of sub-tile 0 of tile t          |  exponentials of sub-tile 1 of tile t-1
//     phase B(t):  P V of sub-tile 1 of tile t-1           |  maximum check of sub-tile 0 of tile t
//     phase C(t):  scores of sub-tile 1 of tile t          |  exponentials of sub-tile 0 of tile t
//     phase D(t):  P V of sub-tile 0 of tile t             |  maximum check of sub-tile 1 of tile t
// with the MFMA column and the VALU column of a phase interleaved instruction by instruction (sched_group_barrier).  The running maximum is
// checked per sub-tile (same lazy rule and THR as above).  V^T of tile t-1 is read one tile late, so the LDS ring has three K / V^T stages
// (48 KB; two workgroups per CU) and one barrier per tile as before.
template <int THR>
__global__ __launch_bounds__(256, 2) void attn_q64v3_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = 64, DSTEPS = 4, DVT = 2, CPR = 8;
  constexpr int KBYTES = KVB * DP * 2, STAGE = 2 * KBYTES, NSLOT = 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int wg = xcd_remap(blockIdx.x, p.nfull);
  const int qt = wg % p.qtiles, bh = wg / p.qtiles;
  const int h = bh % p.heads, b = bh / p.heads;
  const int q0 = qt * 256 + wave * 64;
  const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);
  unsigned k_voff[2], v_voff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = (e * 4 + wave) * 8 + lane / CPR;
    k_voff[e] = (unsigned)row * (unsigned)p.k_rs * 2u + (unsigned)k_logical_chunk<CPR>(row, lane % CPR) * 16u;
    const int vrow = (e * 4 + wave) * 8 + (lane >> 3);
    v_voff[e] = (unsigned)vrow * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 7) ^ ((vrow >> 1) & 7)) * 16u;
  }
  const unsigned k_tile = (unsigned)KVB * (unsigned)p.k_rs * 2u;
  auto stage = [&](int kt) __attribute__((always_inline)) {
    char* const sb = smem + (kt % NSLOT) * STAGE;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      auto* dk = (__attribute__((address_space(3))) void*)(sb + (e * 4 + wave) * 1024);
      const unsigned kv = k_voff[e];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * k_tile, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      auto* dv = (__attribute__((address_space(3))) void*)(sb + KBYTES + (e * 4 + wave) * 1024);
      const unsigned vv = v_voff[e];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KVB * 2u), 0, 0);
    }
  };
  const int ntiles = (p.nk + KVB - 1) / KVB;
  stage(0);
  if (ntiles > 1) stage(1);

  const float c2 = p.scale_log2e;
  f16x8 qf[2][DSTEPS];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int qrow = min(q0 + a * 32 + li, p.nq - 1);
    const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + (long)h * DP + hi * 8;
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) {
      const f16x8 raw = *reinterpret_cast<const f16x8*>(qp + ds * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[a][ds][e] = (f16)((float)raw[e] * c2);
    }
  }
  f32x16 oacc[DVT][2], sacc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < DVT; ++i) oacc[i][a][r] = 0.f;
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  f16x8 ones, mfrag[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ones[e] = (f16)((hi == 0 && e < 2) ? 1.0f : 0.0f);
    mfrag[0][e] = mfrag[1][e] = (f16)0.0f;
  }
  // per-lane LDS offsets inside a stage: K fragment (sub-tile, k-step), V^T fragment (channel block, sub-tile, 8-key half)
  const int krow = key_perm(li);
  int kofs[2][DSTEPS], vofs[DVT][2][2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) {
      const int row = s2 * 32 + krow;
      kofs[s2][ds] = row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4);
    }
#pragma unroll
  for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = dt * 32 + li;
        vofs[dt][s2][j] = KBYTES + row * 128 + (((s2 * 4 + hi * 2 + j) ^ ((row >> 1) & 7)) << 4);
      }
  f16x8 pf[2][2][2];   // [sub-tile][query fragment][8-key half]
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

  // ---- the four columns ----
  auto scores = [&](int s2, const char* sb) __attribute__((always_inline)) {   // 4 reads, 2 + 8 MFMAs
    f16x8 kf[DSTEPS];
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) kf[ds] = *reinterpret_cast<const f16x8*>(sb + kofs[s2][ds]);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      sacc[s2][a] = FMX_MFMA_32x32x16(ones, mfrag[a], z);
    }
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
      for (int a = 0; a < 2; ++a) sacc[s2][a] = FMX_MFMA_32x32x16(kf[ds], qf[a][ds], sacc[s2][a]);
  };
  auto exps = [&](int s2) __attribute__((always_inline)) {                     // 32 exp2, 32 adds, 16 packs
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[s2][a][r]);
        if (r & 1) ps1 += e; else ps0 += e;
        pf[s2][a][r >> 3][r & 7] = (f16)e;
      }
      l_run[a] += ps0 + ps1;
    }
  };
  auto pv = [&](int s2, const char* sb) __attribute__((always_inline)) {       // 4 reads, 8 MFMAs
    f16x8 vf[DVT][2];
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j) vf[dt][j] = *reinterpret_cast<const f16x8*>(sb + vofs[dt][s2][j]);
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) oacc[dt][a] = FMX_MFMA_32x32x16(vf[dt][j], pf[s2][a][j], oacc[dt][a]);
  };
  float mx[2];
  auto maxima = [&](int s2) __attribute__((always_inline)) {                   // ~34 VALU
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float m0 = fmaxf(sacc[s2][a][0], sacc[s2][a][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) m0 = fmaxf(fmaxf(m0, sacc[s2][a][r]), sacc[s2][a][r + 1]);
      const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
      mx[a] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
  };
  auto moved = [&](int s2, bool first) __attribute__((always_inline)) {        // rare after the first few tiles (wave-uniform branch)
    if (first || __any(fmaxf(mx[0], mx[1]) > (float)THR)) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float delta = first ? mx[a] : fmaxf(mx[a], 0.f);
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        m_run[a] += delta;
        l_run[a] *= alpha;
        const f16 mh = (f16)(-m_run[a]);
        const f16 ml = (f16)(-m_run[a] - (float)mh);
        mfrag[a][0] = hi == 0 ? mh : (f16)0.0f;
        mfrag[a][1] = hi == 0 ? ml : (f16)0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
          for (int i = 0; i < DVT; ++i) oacc[i][a][r] *= alpha;
          sacc[s2][a][r] -= delta;
        }
      }
    }
  };
  auto mask_tail = [&](int s2, int kt) __attribute__((always_inline)) {
    if ((kt + 1) * KVB > p.nk) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * KVB + s2 * 32 + hi * 16 + r >= p.nk) sacc[s2][a][r] = -INFINITY;
    }
  };
  // phase shapes: `scores` beside `exps` (10 MFMAs, ~80 VALU) and `pv` beside `maxima` (8 MFMAs, ~34 VALU)
#define FMX_V3_PHASE_SE()                                      \
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);           \
  __builtin_amdgcn_sched_group_barrier(0x402, 10, 0);          \
  _Pragma("unroll") for (int i_ = 0; i_ < 10; ++i_) {          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x402, 7, 0);         \
  }
#define FMX_V3_PHASE_PM()                                      \
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);           \
  __builtin_amdgcn_sched_group_barrier(0x402, 10, 0);          \
  _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
    __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);         \
  }

  wait_vmcnt0();
  __syncthreads();

  // ---- tile 0: nothing of a previous tile beside phases A and B ----
  {
    const char* sb = smem;
    __builtin_amdgcn_sched_barrier(0);
    scores(0, sb);
    __builtin_amdgcn_sched_barrier(0);
    mask_tail(0, 0);
    maxima(0);
    moved(0, true);
    __builtin_amdgcn_sched_barrier(0);
    scores(1, sb);
    exps(0);
    FMX_V3_PHASE_SE()
    __builtin_amdgcn_sched_barrier(0);
    mask_tail(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    pv(0, sb);
    maxima(1);
    FMX_V3_PHASE_PM()
    __builtin_amdgcn_sched_barrier(0);
    moved(1, false);
    wait_vmcnt0();
    __syncthreads();
    if (2 < ntiles) stage(2);
  }
  // (the ragged-tail mask is a branch: LLVM sinks the exponentials of a phase below it, out of the MFMAs' block -- so only the last tile's
  //  copy of the body has it)
  auto body = [&](int t, auto MASK) __attribute__((always_inline)) {
    const char* sb = smem + (t % NSLOT) * STAGE;
    const char* sp = smem + ((t - 1) % NSLOT) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    scores(0, sb);                 // A
    exps(1);
    FMX_V3_PHASE_SE()
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(MASK)::value) mask_tail(0, t);
    __builtin_amdgcn_sched_barrier(0);
    pv(1, sp);                     // B
    maxima(0);
    FMX_V3_PHASE_PM()
    __builtin_amdgcn_sched_barrier(0);
    moved(0, false);
    __builtin_amdgcn_sched_barrier(0);
    scores(1, sb);                 // C
    exps(0);
    FMX_V3_PHASE_SE()
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(MASK)::value) mask_tail(1, t);
    __builtin_amdgcn_sched_barrier(0);
    pv(0, sb);                     // D
    maxima(1);
    FMX_V3_PHASE_PM()
    __builtin_amdgcn_sched_barrier(0);
    moved(1, false);
    wait_vmcnt0();
    __syncthreads();
    if (t + 2 < ntiles) stage(t + 2);
  };
  for (int t = 1; t + 1 < ntiles; ++t) body(t, IC<0>{});
  if (ntiles > 1) body(ntiles - 1, IC<1>{});
  {  // the last sub-tile
    const char* sp = smem + ((ntiles - 1) % NSLOT) * STAGE;
    exps(1);
    pv(1, sp);
  }
#undef FMX_V3_PHASE_SE
#undef FMX_V3_PHASE_PM

#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run[a]), __float_as_uint(l_run[a]), false, false);
    const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
    const int qg = q0 + a * 32 + li;
    f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        union { f16x4 h4; unsigned u[2]; } lo, up;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo.h4[e] = (f16)(oacc[dt][a][g * 4 + e] * inv);
          up.h4[e] = (f16)(oacc[dt][a][(g + 1) * 4 + e] * inv);
        }
        const u32x2 x = __builtin_amdgcn_permlane32_swap(lo.u[0], up.u[0], false, false);
        const u32x2 y = __builtin_amdgcn_permlane32_swap(lo.u[1], up.u[1], false, false);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {x[0], y[0], x[1], y[1]};
        if (qg < p.nq) *reinterpret_cast<u32x4*>(op + dt * 32 + (g + hi) * 8) = v;
      }
  }
}

// ---- wave-specialised form ("ws"): score / softmax waves and P V waves -------------------------------------------------------------------
// A workgroup of 8 waves covers 256 queries; waves 0-3 ("S", 64 queries each) compute scores and the online softmax of key tile t and leave
// P (fp16, as the MFMA operand fragments they already are: [fragment][lane], 8 KB) plus the rescale factor of a moved maximum in LDS; waves
// 4-7 ("O", the same 64 queries) pick tile t up one interval later and accumulate O^T += V^T P^T.  A SIMD hosts one S and one O wave, and
// neither carries the other's registers (S: Q fragments + scores; O: the output accumulators) -- which is what lets d_head 128 run two waves
// per SIMD at 64 queries per wave at all (the symmetric kernel above needs ~350 registers per wave there: one wave per SIMD, nothing overlaps).
// One barrier per key tile.  Intervals I_t = (B_{t-1}, B_t):
//     I_t:  S waves: request K_{t+1}; scores + softmax of tile t -> P slot t & 1          O waves: request V_{t+1}; P V of tile t-1
// K tiles live in a 2-slot ring, V^T tiles (read one interval later) in a 3-slot ring, P in 2 slots per query group.
// Measured (Flux shape, 2 x 24 heads x 4352 tokens; profiles/r07_attention_ws_trace.md): 585 -> 497 us (795 -> 936 TFLOP/s).  The S wave is
// the critical path: per tile ~1150 cycles for sub-tile 0's 18 MFMAs (two accumulator chains: one MFMA per 64 cycles, the O wave's MFMAs
// take the slots between), ~1350 for sub-tile 1's MFMAs under sub-tile 0's exponentials, ~730 for sub-tile 1's exponentials (v_exp_f32 is
// quarter rate: 16 cycles per wave instruction) and ~500 around the barrier; the O wave needs ~2300 of the ~3730.  Wave priorities, a third
// accumulator chain in the first phase and delaying the O wave were measured and change nothing / lose.
//
// ROT = 1 (round 6): the S wave's sub-tile pipeline ROTATED ACROSS the tile boundary.  In the form above sub-tile 0's score MFMAs have nothing to
// run under (their softmax cannot start before they finish: ~1150 cycles per tile of matrix-only issue) and sub-tile 1's exponentials nothing to
// cover them (~730 cycles of vector-only issue).  Rotated, every phase is "maximum check of sub-tile n -> issue the scores of sub-tile n + 1 ->
// exponentials of sub-tile n under them": the scores of (t + 1, 0) run under the exponentials of (t, 1).  That needs K_{t+1} visible during
// interval t, i.e. requested in interval t - 1: K moves to a THREE-slot ring, and V^T, which was requested two intervals before its use, to a
// two-slot ring requested one interval before (same 80 KB).  Past the last tile the "next" scores read a stale slot and are never used.
template <int THR, int DP, int ROT = 0>
__global__ __launch_bounds__(512, 2) void attn_ws_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DSTEPS = DP / 16, DVT = DP / 32, CPR = DP / 8;
  constexpr int KBYTES = KVB * DP * 2, VBYTES = DVT * 32 * 128;
  constexpr int KROWS = 1024 / (DP * 2);            // K rows per 1-KiB piece
  constexpr int NPK = KVB / KROWS / 4, NPV = DP / 8 / 4;   // pieces per staging wave and tile
  constexpr int PSLOT = 8 * 1024 + 1280;            // 8 operand fragments x 64 lanes x 16 B, then alpha[2][2][64] floats, flag[2], (pad)
  constexpr int KSLOTS = ROT ? 3 : 2, VSLOTS = ROT ? 2 : 3;
  constexpr int OFF_V = KSLOTS * KBYTES, OFF_P = OFF_V + VSLOTS * VBYTES, OFF_L = OFF_P + 2 * 4 * PSLOT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_s = wave < 4;
  const int grp = wave & 3;
  const int hi = lane >> 5, li = lane & 31;
  const int wg = xcd_remap(blockIdx.x, p.nfull);
  const int qt = wg % p.qtiles, bh = wg / p.qtiles;
  const int h = bh % p.heads, b = bh / p.heads;
  const int q0 = qt * 256 + grp * 64;
  const int ntiles = (p.nk + KVB - 1) / KVB;
  char* const pbase = smem + OFF_P + grp * PSLOT;    // slot s of this group at pbase + s * 4 * PSLOT
  float* const lbase = reinterpret_cast<float*>(smem + OFF_L) + grp * 128;

  if (is_s) {
    // =================================================== S waves ===================================================
    const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
    unsigned k_voff[NPK];
#pragma unroll
    for (int e = 0; e < NPK; ++e) {
      const int row = (e * 4 + grp) * KROWS + lane / CPR;
      k_voff[e] = (unsigned)row * (unsigned)p.k_rs * 2u + (unsigned)k_logical_chunk<CPR>(row, lane % CPR) * 16u;
    }
    const unsigned k_tile = (unsigned)KVB * (unsigned)p.k_rs * 2u;
    auto stage_k = [&](int slot, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int e = 0; e < NPK; ++e) {
        auto* dk = (__attribute__((address_space(3))) void*)(smem + slot * KBYTES + (e * 4 + grp) * 1024);
        const unsigned kv = k_voff[e];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * k_tile, 0, 0);
      }
    };
    stage_k(0, 0);
    if (ROT && ntiles > 1) stage_k(1, 1);
    const float c2 = p.scale_log2e;
    f16x8 qf[2][DSTEPS];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int qrow = min(q0 + a * 32 + li, p.nq - 1);
      const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + (long)h * DP + hi * 8;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 raw = *reinterpret_cast<const f16x8*>(qp + ds * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[a][ds][e] = (f16)((float)raw[e] * c2);
      }
    }
    float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
    f16x8 ones, mfrag[2];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ones[e] = (f16)((hi == 0 && e < 2) ? 1.0f : 0.0f);
      mfrag[0][e] = mfrag[1][e] = (f16)0.0f;
    }
    const int krow = key_perm(li);
    wait_vmcnt0();
    __builtin_amdgcn_s_barrier();                                   // B_{-1}: K_0 (and V_0) visible
    // One tile = two 32-key sub-tiles, each with its own maximum check and P / rescale record: the score MFMAs of sub-tile 1 are independent
    // of sub-tile 0's softmax, so they are issued interleaved with its exp2 / sum / pack instructions (one MFMA, a handful of VALU, ...):
    // this wave's matrix work runs under its own vector work, and what stays exposed (sub-tile 1's softmax) is covered by the O wave.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (ROT != 0) {
      // ---- rotated pipeline: phase (t, s2) = maximum check of sub-tile (t, s2) | scores of the NEXT sub-tile issued | exponentials of (t, s2) under them ----
      f32x16 sacc[2][2];
      f16x8 kf[2][DSTEPS];
      auto read_k = [&](int which, const char* sk, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int ds = 0; ds < DSTEPS; ++ds) {
          const int row = half * 32 + krow;
          kf[which][ds] = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
        }
      };
      auto scores = [&](int which) __attribute__((always_inline)) {     // sacc[which] = K (sub-tile in kf[which]) Q^T - running maximum
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sacc[which][a] = FMX_MFMA_32x32x16(ones, mfrag[a], z);
        }
#pragma unroll
        for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
          for (int a = 0; a < 2; ++a) sacc[which][a] = FMX_MFMA_32x32x16(kf[which][ds], qf[a][ds], sacc[which][a]);
      };
      // prologue: both halves of K_0, the scores of (0, 0) with nothing to run under
      __builtin_amdgcn_sched_barrier(0);
      read_k(0, smem, 0);
      read_k(1, smem, 1);
      scores(0);
      __builtin_amdgcn_sched_barrier(0);
      int kslot_next = ntiles > 1 ? 1 : 0;                            // slot of K_{t+1} while in interval t
      for (int t = 0; t <= ntiles; ++t) {
        if (t < ntiles) {
          if (t + 2 < ntiles) stage_k(kslot_next == 2 ? 0 : kslot_next + 1, t + 2);
          const char* skn = smem + kslot_next * KBYTES;
          char* const ps = pbase + (t & 1) * (4 * PSLOT);
          float* const ctrl = reinterpret_cast<float*>(ps + 8 * 1024);
          const bool ragged = (t + 1) * KVB > p.nk;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            __builtin_amdgcn_sched_barrier(0);
            if (ragged) {
#pragma unroll
              for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                  if (t * KVB + s2 * 32 + hi * 16 + r >= p.nk) sacc[s2][a][r] = -INFINITY;
            }
            float mx[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              float m0 = fmaxf(sacc[s2][a][0], sacc[s2][a][1]);
#pragma unroll
              for (int r = 2; r < 16; r += 2) m0 = fmaxf(fmaxf(m0, sacc[s2][a][r]), sacc[s2][a][r + 1]);
              const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
              mx[a] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const bool first = t == 0 && s2 == 0;
            const bool moved = first || __any(fmaxf(mx[0], mx[1]) > (float)THR);
            if (moved) {
#pragma unroll
              for (int a = 0; a < 2; ++a) {
                const float delta = first ? mx[a] : fmaxf(mx[a], 0.f);
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
                m_run[a] += delta;
                l_run[a] *= alpha;
                const f16 mh = (f16)(-m_run[a]);
                const f16 ml = (f16)(-m_run[a] - (float)mh);
                mfrag[a][0] = hi == 0 ? mh : (f16)0.0f;
                mfrag[a][1] = hi == 0 ? ml : (f16)0.0f;
                ctrl[(s2 * 2 + a) * 64 + lane] = alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[s2][a][r] -= delta;
              }
            }
            if (lane == 0) reinterpret_cast<int*>(ctrl)[256 + s2] = (moved && !first) ? 1 : 0;
            __builtin_amdgcn_sched_barrier(0);
            // the next sub-tile's scores against the maximum as it stands now: (t, 1) from kf[1] behind (t, 0); (t + 1, 0) from kf[0] behind (t, 1);
            // and the fragments of the sub-tile after that into the operand registers these scores' predecessor has left
            if (s2 == 0) {
              scores(1);
              read_k(0, skn, 0);
            } else {
              scores(0);
              read_k(1, skn, 1);
            }
            // ... under the exponentials of this one
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              float ps0 = 0.f, ps1 = 0.f;
              f16x8 pf[2];
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(sacc[s2][a][r]);
                if (r & 1) ps1 += e; else ps0 += e;
                pf[r >> 3][r & 7] = (f16)e;
              }
#pragma unroll
              for (int j = 0; j < 2; ++j) *reinterpret_cast<f16x8*>(ps + ((a * 2 + s2) * 2 + j) * 1024 + lane * 16) = pf[j];
              l_run[a] += ps0 + ps1;
            }
#pragma unroll
            for (int i = 0; i < 2 * DSTEPS + 2; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (i < DSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x402, (80 + DSTEPS) / (2 * DSTEPS + 2), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t + 1 == ntiles) {
            lbase[lane] = l_run[0];
            lbase[64 + lane] = l_run[1];
          }
          kslot_next = kslot_next == 2 ? 0 : kslot_next + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // B_t
      }
      return;
    }
    for (int t = 0; t <= ntiles; ++t) {
      if (t < ntiles) {
        if (t + 1 < ntiles) stage_k((t + 1) & 1, t + 1);
        const char* sk = smem + (t & 1) * KBYTES;
        char* const ps = pbase + (t & 1) * (4 * PSLOT);
        float* const ctrl = reinterpret_cast<float*>(ps + 8 * 1024);   // [sub-tile][query fragment][lane] alpha, then [sub-tile] flag
        const bool ragged = (t + 1) * KVB > p.nk;
        f32x16 sacc[2][2];
        // ---- scores of sub-tile 0; sub-tile 1's K fragments arrive behind its MFMAs ----
        __builtin_amdgcn_sched_barrier(0);
        f16x8 kf[2][DSTEPS];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int ds = 0; ds < DSTEPS; ++ds) {
            const int row = s2 * 32 + krow;
            kf[s2][ds] = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
          }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          sacc[0][a] = FMX_MFMA_32x32x16(ones, mfrag[a], z);
        }
#pragma unroll
        for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
          for (int a = 0; a < 2; ++a) sacc[0][a] = FMX_MFMA_32x32x16(kf[0][ds], qf[a][ds], sacc[0][a]);
        __builtin_amdgcn_sched_group_barrier(0x100, DSTEPS, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int ds = 0; ds < DSTEPS; ++ds) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (ragged) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                if (t * KVB + s2 * 32 + hi * 16 + r >= p.nk) sacc[s2][a][r] = -INFINITY;
          }
          float mx[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            float m0 = fmaxf(sacc[s2][a][0], sacc[s2][a][1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) m0 = fmaxf(fmaxf(m0, sacc[s2][a][r]), sacc[s2][a][r + 1]);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
            mx[a] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
          }
          const bool first = t == 0 && s2 == 0;
          const bool moved = first || __any(fmaxf(mx[0], mx[1]) > (float)THR);
          if (moved) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              // a sub-tile that is masked out entirely (ragged tail) must not move the maximum to -inf
              const float delta = first ? mx[a] : fmaxf(mx[a], 0.f);
              const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
              m_run[a] += delta;
              l_run[a] *= alpha;
              const f16 mh = (f16)(-m_run[a]);
              const f16 ml = (f16)(-m_run[a] - (float)mh);
              mfrag[a][0] = hi == 0 ? mh : (f16)0.0f;
              mfrag[a][1] = hi == 0 ? ml : (f16)0.0f;
              ctrl[(s2 * 2 + a) * 64 + lane] = alpha;      // the O wave rescales its accumulators before it adds this sub-tile
#pragma unroll
              for (int r = 0; r < 16; ++r) sacc[s2][a][r] -= delta;
            }
          }
          if (lane == 0) reinterpret_cast<int*>(ctrl)[256 + s2] = (moved && !first) ? 1 : 0;
          __builtin_amdgcn_sched_barrier(0);
          if (s2 == 0) {
            // scores of sub-tile 1 (against the maximum as it stands now) ...
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.f;
              sacc[1][a] = FMX_MFMA_32x32x16(ones, mfrag[a], z);
            }
#pragma unroll
            for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
              for (int a = 0; a < 2; ++a) sacc[1][a] = FMX_MFMA_32x32x16(kf[1][ds], qf[a][ds], sacc[1][a]);
          }
          // ... under the exponentials of sub-tile 0
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            float ps0 = 0.f, ps1 = 0.f;
            f16x8 pf[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float e = __builtin_amdgcn_exp2f(sacc[s2][a][r]);
              if (r & 1) ps1 += e; else ps0 += e;
              pf[r >> 3][r & 7] = (f16)e;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<f16x8*>(ps + ((a * 2 + s2) * 2 + j) * 1024 + lane * 16) = pf[j];
            l_run[a] += ps0 + ps1;
          }
          if (s2 == 0) {
#pragma unroll
            for (int i = 0; i < 2 * DSTEPS + 2; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x402, (80 + DSTEPS) / (2 * DSTEPS + 2), 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 == ntiles) {
          lbase[lane] = l_run[0];
          lbase[64 + lane] = l_run[1];
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                 // B_t
    }
    return;
  }

  // ===================================================== O waves =====================================================
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);
  unsigned v_voff[NPV];
#pragma unroll
  for (int e = 0; e < NPV; ++e) {
    const int row = (e * 4 + grp) * 8 + (lane >> 3);
    v_voff[e] = (unsigned)row * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  auto stage_v = [&](int slot, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < NPV; ++e) {
      auto* dv = (__attribute__((address_space(3))) void*)(smem + OFF_V + slot * VBYTES + (e * 4 + grp) * 1024);
      const unsigned vv = v_voff[e];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KVB * 2u), 0, 0);
    }
  };
  if (!ROT) stage_v(0, 0);
  f32x16 oacc[DVT][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < DVT; ++i) oacc[i][a][r] = 0.f;
  wait_vmcnt0();
  __builtin_amdgcn_s_barrier();                                     // B_{-1}
  int vslot = 0;                                                    // slot of V_{t-1} while in interval t
  for (int t = 0; t <= ntiles; ++t) {
    // V_{t+1} goes to slot (t + 1) % 3: V_{t-2}, its previous tenant, was read in interval t - 1
    // (ROT: V_t itself, one interval before its use, into slot t & 1 -- V_{t-1}, read in this interval, sits in the other one)
    const int vnext = t + 1 < ntiles ? (t + 1) % 3 : 0;
    if (ROT) {
      if (t < ntiles) stage_v(t & 1, t);
    } else if (t + 1 < ntiles) {
      stage_v(vnext, t + 1);
    }
    if (t >= 1) {
      const int u = t - 1;                                          // the tile whose P the S wave left before B_{t-1}
      const char* ps = pbase + (u & 1) * (4 * PSLOT);
      const float* ctrl = reinterpret_cast<const float*>(ps + 8 * 1024);
      const char* sv = smem + OFF_V + vslot * VBYTES;
      // One LDS round trip per tile is exposed (the flags and the first operands, right behind the barrier); everything else is requested
      // under MFMAs: sub-tile 0's V^T fragments 4 .. 7 and P / V^T 0 .. 3 of sub-tile 1 behind sub-tile 0's MFMAs, the rest behind sub-tile 1's.
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      const i32x2 flags = *reinterpret_cast<const i32x2*>(ctrl + 256);
      f16x8 pf[2][2][2], vf[2][DVT][2];                              // [sub-tile][query fragment | channel block][8-key half]
      auto read_p = [&](int s2) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < 2; ++j) pf[s2][a][j] = *reinterpret_cast<const f16x8*>(ps + ((a * 2 + s2) * 2 + j) * 1024 + lane * 16);
      };
      auto read_v = [&](int s2, int dt) __attribute__((always_inline)) {
        const int row = dt * 32 + li;
        const char* rp = sv + row * 128;
        const int sw = (row >> 1) & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) vf[s2][dt][j] = *reinterpret_cast<const f16x8*>(rp + (((s2 * 4 + hi * 2 + j) ^ sw) << 4));
      };
      auto rescale = [&](int s2) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const float alpha = ctrl[(s2 * 2 + a) * 64 + lane];
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < DVT; ++i) oacc[i][a][r] *= alpha;
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      read_p(0);
#pragma unroll
      for (int dt = 0; dt < DVT / 2; ++dt) read_v(0, dt);
      __builtin_amdgcn_sched_barrier(0);
      if (flags[0]) rescale(0);                                     // wave-uniform: the maximum moved at this sub-tile
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = DVT / 2; dt < DVT; ++dt) read_v(0, dt);
      read_p(1);
#pragma unroll
      for (int dt = 0; dt < DVT / 2; ++dt) read_v(1, dt);
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int a = 0; a < 2; ++a) oacc[dt][a] = FMX_MFMA_32x32x16(vf[0][dt][j], pf[0][a][j], oacc[dt][a]);
      // DVT + 4 + DVT reads behind 4 * DVT MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
#pragma unroll
      for (int i = 4; i < 2 * DVT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (flags[1]) rescale(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = DVT / 2; dt < DVT; ++dt) read_v(1, dt);
#pragma unroll
      for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int a = 0; a < 2; ++a) oacc[dt][a] = FMX_MFMA_32x32x16(vf[1][dt][j], pf[1][a][j], oacc[dt][a]);
#pragma unroll
      for (int i = 0; i < DVT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    vslot = ROT ? (t & 1) : t % 3;                                  // in interval t + 1 the tile to consume is V_t
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // B_t
  }
  // ---- finish: 1 / l from the S wave (written before the last barrier), O[b][q][h*DP + d] -------------------------------------------------------
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const float lr = lbase[a * 64 + lane];
    const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lr), __float_as_uint(lr), false, false);
    const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
    const int qg = q0 + a * 32 + li;
    f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        union { f16x4 h4; unsigned u[2]; } lo, up;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo.h4[e] = (f16)(oacc[dt][a][g * 4 + e] * inv);
          up.h4[e] = (f16)(oacc[dt][a][(g + 1) * 4 + e] * inv);
        }
        const u32x2 x = __builtin_amdgcn_permlane32_swap(lo.u[0], up.u[0], false, false);
        const u32x2 y = __builtin_amdgcn_permlane32_swap(lo.u[1], up.u[1], false, false);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {x[0], y[0], x[1], y[1]};
        if (qg < p.nq) *reinterpret_cast<u32x4*>(op + dt * 32 + (g + hi) * 8) = v;
      }
  }
}


// ---- d_head 64, contexts of at most 128 keys (the 77-token text context of every cross-attention, round 3) -------------------------------
// The launch is a streaming kernel -- one read of Q, one write of O, ~90 FLOP per byte -- that the 64-query kernels above ran at a third of
// the HBM rate (profiles/r07m: 2.6 ms per SDXL forward at 2.6 TB/s): a workgroup there lives for ONE query tile, and its life is a chain of
// latencies (Q from HBM -> scores -> softmax -> P V -> store drain) with two workgroups per CU to overlap it.  Here
//   * a workgroup is PERSISTENT over `sk_tpw` consecutive 128-query tiles of one (batch, head): the two K / V^T tiles (32 KB) are staged in
//     LDS once, and the Q rows of tile t + 1 are requested before tile t is computed, so a load is in flight behind every tile's arithmetic;
//   * 32 queries per wave (one query fragment), one key block's fragments live at a time: 114 registers -> four workgroups per CU instead of
//     two, 32 KB of LDS each;
//   * the softmax is ONE pass over all NB 32-key blocks (exact maximum, no running state, no rescale), and only the blocks that hold keys are
//     computed (77 keys: 3 of the 4 blocks of the two 64-key tiles -- the looped kernels did 128 keys' worth of exponentials and MFMAs).
// Same operand layouts as attn_q64v2_kernel (S^T = K Q^T with permuted key rows, O^T = V^T P^T, lane = query).
template <int NB>
__global__ __launch_bounds__(256, 4) void attn_short_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = 64, DSTEPS = 4, CPR = 8;
  constexpr int KBYTES = KVB * DP * 2;          // one 64-key K tile: 64 rows x 128 B
  constexpr int STAGE = KBYTES + 2 * 32 * 128;   // + its V^T tile: 64 channel rows x 128 B (64 keys)
  constexpr int NT = (NB + 1) / 2;               // 64-key tiles that hold keys

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = wg % p.sk_chunks;
  const int bh = wg / p.sk_chunks;
  const int h = bh % p.heads, b = bh / p.heads;
  const int t0 = chunk * p.sk_tpw, t1 = min(t0 + p.sk_tpw, p.qtiles);   // 128-query tiles [t0, t1) of this (batch, head)

  // K / V^T of this (batch, head): tiles 0 .. NT-1 by LDS-DMA, two + two 1-KiB pieces per wave and tile (layout and swizzles of q64v2)
  const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
  const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int krow = (e * 4 + wave) * 8 + lane / CPR;
      const unsigned kv = (unsigned)krow * (unsigned)p.k_rs * 2u + (unsigned)k_logical_chunk<CPR>(krow, lane % CPR) * 16u;
      auto* dk = (__attribute__((address_space(3))) void*)(smem + kt * STAGE + (e * 4 + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * ((unsigned)KVB * (unsigned)p.k_rs * 2u), 0, 0);
      const int vrow = (e * 4 + wave) * 8 + (lane >> 3);
      const unsigned vv = (unsigned)vrow * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 7) ^ ((vrow >> 1) & 7)) * 16u;
      auto* dv = (__attribute__((address_space(3))) void*)(smem + kt * STAGE + KBYTES + (e * 4 + wave) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KVB * 2u), 0, 0);
    }

  const float c2 = p.scale_log2e;
  const f16* qhead = p.q + (long)b * p.q_bs + (long)h * DP + hi * 8;
  f16x8 raw[DSTEPS];   // Q rows of the NEXT tile, as loaded
  auto load_q = [&](int t) {
    const int qrow = min(t * 128 + wave * 32 + li, p.nq - 1);
    const f16* qp = qhead + (long)qrow * p.q_rs;
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds) raw[ds] = *reinterpret_cast<const f16x8*>(qp + ds * 16);
  };
  if (t0 < t1) load_q(t0);
  wait_vmcnt0();
  __syncthreads();   // K / V^T visible to every wave; nothing below writes LDS, so this is the only barrier

  const int krow = key_perm(li);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  for (int t = t0; t < t1; ++t) {
    f16x8 qf[DSTEPS];
#pragma unroll
    for (int ds = 0; ds < DSTEPS; ++ds)
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[ds][e] = (f16)((float)raw[ds][e] * c2);
    if (t + 1 < t1) load_q(t + 1);   // in flight behind this tile's arithmetic (its registers were just consumed)

    // ---- scores of all NB blocks: S^T = K Q^T in exp2 units ----
    f32x16 sacc[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[g][r] = 0.f;
      const char* sk = smem + (g >> 1) * STAGE;
      const int row = (g & 1) * 32 + krow;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
        sacc[g] = FMX_MFMA_32x32x16(kf, qf[ds], sacc[g]);
      }
      __builtin_amdgcn_sched_barrier(0);   // one block's fragments at a time: left alone the scheduler hoists all 4 NB fragment reads (48 registers)
    }
    // keys >= nk: out of the softmax.  Per key block and uniform: only the block(s) that reach past nk pay for the compare + select pairs (77 keys: one
    // block of three -- the kernel is vector-issue-bound, profiles/r32: 16 instead of 48 pairs per lane)
#pragma unroll
    for (int g = 0; g < NB; ++g)
      if ((g + 1) * 32 > p.nk) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (g * 32 + hi * 16 + r >= p.nk) sacc[g][r] = -INFINITY;
      }
    // ---- one-pass softmax: exact maximum of the query (both half-waves hold 16 keys of each block), P = 2^(s - m) ----
    float m0 = sacc[0][0];
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 = fmaxf(m0, sacc[g][r]);
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
    const float mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    float l = 0.f;
    f16x8 pf[NB][2];
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[g][r] - mx);
        l += e;
        pf[g][r >> 3][r & 7] = (f16)e;
      }
    // ---- O^T = V^T P^T over the NB blocks ----
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
      const int row = dt * 32 + li;
      const int swz = (row >> 1) & 7;
#pragma unroll
      for (int g = 0; g < NB; ++g) {
        const char* rp = smem + (g >> 1) * STAGE + KBYTES + row * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(rp + ((((g & 1) * 4 + hi * 2 + j) ^ swz) << 4));
          oacc[dt] = FMX_MFMA_32x32x16(vf, pf[g][j], oacc[dt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 1 / l and the store: 16 bytes per lane after a half-wave swap (as q64v2) ----
    const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
    const int qg = t * 128 + wave * 32 + li;
    f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + (long)h * DP;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        union { f16x4 h4; unsigned u[2]; } lo, up;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo.h4[e] = (f16)(oacc[dt][g * 4 + e] * inv);
          up.h4[e] = (f16)(oacc[dt][(g + 1) * 4 + e] * inv);
        }
        const u32x2 x = __builtin_amdgcn_permlane32_swap(lo.u[0], up.u[0], false, false);
        const u32x2 y = __builtin_amdgcn_permlane32_swap(lo.u[1], up.u[1], false, false);
        const u32x4 v = {x[0], y[0], x[1], y[1]};
        if (qg < p.nq) *reinterpret_cast<u32x4*>(op + dt * 32 + (g + hi) * 8) = v;
      }
  }
}

template <int NB>
int launch_attn_short_nb(AttnParams& p, int grid, hipStream_t st) {
  constexpr int SMEM = ((NB + 1) / 2) * (KVB * 64 * 2 + 2 * 32 * 128);
  hipLaunchKernelGGL(attn_short_kernel<NB>, dim3(grid), dim3(256), SMEM, st, p);
  FMX_LAUNCH_CHECK("fmx_attention_f16 (short context)");
  return FMX_OK;
}

// nk <= 128, d_head 64, no mask: tiles of 128 queries, `sk_tpw` of them per persistent workgroup so that the launch is at most one round of
// the 4-per-CU workgroup slots (every workgroup resident from the start: no tail round)
int launch_attn_short(AttnParams p, hipStream_t st) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256;
    if (!(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)) cus = 256;
    const char* e = fmx_knob("FMX_ATTN_SHORT_WGS");   // A/B knob: workgroups per CU the launch is sized for (the kernel allows 4)
    const int per_cu = e ? atoi(e) : 4;
    slots = (per_cu >= 1 && per_cu <= 4 ? per_cu : 4) * cus;
  }
  p.qtiles = (p.nq + 127) / 128;
  const long bh = (long)p.batch * p.heads;
  int chunks = (int)((slots + bh - 1) / bh);           // workgroups per (batch, head) the slots allow ...
  if (chunks * bh > slots && chunks > 1) --chunks;     // ... rounded down to stay within one round
  if (chunks > p.qtiles) chunks = p.qtiles;
  if (chunks < 1) chunks = 1;
  p.sk_tpw = (p.qtiles + chunks - 1) / chunks;
  p.sk_chunks = (p.qtiles + p.sk_tpw - 1) / p.sk_tpw;
  const int grid = (int)(bh * p.sk_chunks);
  const int nb = (p.nk + 31) / 32;
  switch (nb) {
    case 1: return launch_attn_short_nb<1>(p, grid, st);
    case 2: return launch_attn_short_nb<2>(p, grid, st);
    case 3: return launch_attn_short_nb<3>(p, grid, st);
    default: return launch_attn_short_nb<4>(p, grid, st);
  }
}


// ---- the same, with Q and O moved in FULL CACHE LINES through LDS (round 3, second form) -------------------------------------------------
// In attn_short_kernel a lane owns a query, so every 16-byte Q load / O store instruction of a wave touches 32 rows x 32 bytes: each 128-byte
// line of a (query, head) is requested in four pieces by four instructions.  Here a wave's 32 x 64 Q tile arrives by LDS-DMA (4 pieces of 8
// rows x 128 B: whole lines, bounds-checked by the descriptor, one tile AHEAD of the arithmetic) into one of two 4-KiB slots, the fragments
// are read from there (chunk-swizzled like the K tile), and O goes back through the slot the tile's Q has just left: lanes write their
// 16-byte pieces, then 8 lanes per row store whole lines.  A workgroup walks a contiguous range of the launch's (batch, head, tile) list and
// restages K / V^T when the (batch, head) changes, so any grid size balances: 2 workgroups per CU (64 KB of LDS each), all resident.
// (Round 4: Q requested TWO tiles ahead -- three slots per wave, one descriptor over the whole Q tensor so that the prefetch also crosses (batch, head)
// changes -- was built and measured: 27.6 / 46.4-47.6 us against 27.5 / 46.0 us per launch, profiles/r10e_short_context_attention_two_tiles_ahead.jsonl.
// The lead time of the Q request is not what holds the launch at 0.4 of the HBM rate; the one-tile form stays.)
template <int NB>
__global__ __launch_bounds__(256, 2) void attn_short2_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int DP = 64, DSTEPS = 4, CPR = 8;
  constexpr int KBYTES = KVB * DP * 2;
  constexpr int STAGE = KBYTES + 2 * 32 * 128;
  constexpr int NT = (NB + 1) / 2;
  constexpr int KV_BYTES = 2 * STAGE;            // (two tiles reserved whatever NB is: the slots' base does not depend on it)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  char* const slots = smem + KV_BYTES + wave * 8192;   // this wave's two Q / O slots
  // tiles [g0, g1) of the flattened (batch * heads, 128-query tile) list
  const int total = p.batch * p.heads * p.qtiles;
  const int per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int g0 = wg * per, g1 = min(total, g0 + per);
  if (g0 >= g1) return;

  const float c2 = p.scale_log2e;
  const int krow = key_perm(li);
  const int r8 = lane >> 3, pc = lane & 7;       // DMA / line pass: row within an 8-row piece, physical 16-byte chunk
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

  int bh_cur = -1;
  __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.q), 0, 0, 0x00020000);
  auto q_desc = [&](int bh) {
    const int h = bh % p.heads, b = bh / p.heads;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.q + (long)b * p.q_bs + (long)h * DP), 0, p.q_span, 0x00020000);
  };
  // Q tile `tile` of the current (batch, head) -> slot: 4 pieces, rows >= nq read as zeros (descriptor bounds)
  auto dma_q = [&](const __amdgpu_buffer_rsrc_t& rs, int tile, int slot) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = e * 8 + r8;
      const unsigned off = (unsigned)(tile * 128 + wave * 32 + row) * (unsigned)p.q_rs * 2u + (unsigned)k_logical_chunk<CPR>(row, pc) * 16u;
      auto* dst = (__attribute__((address_space(3))) void*)(slots + slot * 4096 + e * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 0);
    }
  };

  int slot = 0;
  for (int g = g0; g < g1; ++g) {
    const int bh = g / p.qtiles, tile = g - bh * p.qtiles;
    const int h = bh % p.heads, b = bh / p.heads;
    if (bh != bh_cur) {
      // new (batch, head): every wave is done with the old K / V^T; stage the new ones and this tile's Q (nothing was prefetched across
      // the change: the descriptor differs)
      if (bh_cur >= 0) {
        wait_vmcnt0();
        __syncthreads();
      }
      bh_cur = bh;
      rs_q = q_desc(bh);
      const f16* kbase = p.k + (long)b * p.k_bs + (long)h * DP;
      const f16* vbase = p.vt + (long)b * p.vt_bs + (long)h * p.vt_hs;
      const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int kr = (e * 4 + wave) * 8 + lane / CPR;
          const unsigned kv = (unsigned)kr * (unsigned)p.k_rs * 2u + (unsigned)k_logical_chunk<CPR>(kr, lane % CPR) * 16u;
          auto* dk = (__attribute__((address_space(3))) void*)(smem + kt * STAGE + (e * 4 + wave) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * ((unsigned)KVB * (unsigned)p.k_rs * 2u), 0, 0);
          const int vr = (e * 4 + wave) * 8 + (lane >> 3);
          const unsigned vv = (unsigned)vr * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 7) ^ ((vr >> 1) & 7)) * 16u;
          auto* dv = (__attribute__((address_space(3))) void*)(smem + kt * STAGE + KBYTES + (e * 4 + wave) * 1024);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KVB * 2u), 0, 0);
        }
      dma_q(rs_q, tile, slot);
      wait_vmcnt0();
      __syncthreads();
    } else {
      // this tile's Q was requested a tile ago; behind it in the (in-order) counter are only the 4 line stores of the previous tile's O
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    // ---- Q fragments out of the slot (pre-scaled), then the NEXT tile's Q into the other slot ----
    f16x8 qf[DSTEPS];
    {
      const char* qs = slots + slot * 4096 + li * 128;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 raw = *reinterpret_cast<const f16x8*>(qs + (k_phys_chunk<CPR>(li, ds * 2 + hi) << 4));
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[ds][e] = (f16)((float)raw[e] * c2);
      }
    }
    const bool more = g + 1 < g1 && (g + 1) / p.qtiles == bh;   // uniform
    // pinned in program order: the vmcnt(4) above counts on exactly this tile's four O stores being the only memory operations behind it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma_q(rs_q, tile + 1, slot ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");

    f32x16 sacc[NB];
#pragma unroll
    for (int gk = 0; gk < NB; ++gk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[gk][r] = 0.f;
      const char* sk = smem + (gk >> 1) * STAGE;
      const int row = (gk & 1) * 32 + krow;
#pragma unroll
      for (int ds = 0; ds < DSTEPS; ++ds) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + row * (DP * 2) + (k_phys_chunk<CPR>(row, ds * 2 + hi) << 4));
        sacc[gk] = FMX_MFMA_32x32x16(kf, qf[ds], sacc[gk]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int gk = 0; gk < NB; ++gk)
      if ((gk + 1) * 32 > p.nk) {   // (uniform; see attn_short_kernel)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (gk * 32 + hi * 16 + r >= p.nk) sacc[gk][r] = -INFINITY;
      }
    float m0 = sacc[0][0];
#pragma unroll
    for (int gk = 0; gk < NB; ++gk)
#pragma unroll
      for (int r = 0; r < 16; ++r) m0 = fmaxf(m0, sacc[gk][r]);
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m0), __float_as_uint(m0), false, false);
    const float mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    float l = 0.f;
    f16x8 pf[NB][2];
#pragma unroll
    for (int gk = 0; gk < NB; ++gk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(sacc[gk][r] - mx);
        l += e;
        pf[gk][r >> 3][r & 7] = (f16)e;
      }
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
      const int row = dt * 32 + li;
      const int swz = (row >> 1) & 7;
#pragma unroll
      for (int gk = 0; gk < NB; ++gk) {
        const char* rp = smem + (gk >> 1) * STAGE + KBYTES + row * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f16x8 vf = *reinterpret_cast<const f16x8*>(rp + ((((gk & 1) * 4 + hi * 2 + j) ^ swz) << 4));
          oacc[dt] = FMX_MFMA_32x32x16(vf, pf[gk][j], oacc[dt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- O: 16-byte pieces into the slot this tile's Q has left (row = query, chunk-swizzled), then whole lines out ----
    const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
    char* os = slots + slot * 4096;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gq = 0; gq < 4; gq += 2) {
        union { f16x4 h4; unsigned u[2]; } lo, up;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo.h4[e] = (f16)(oacc[dt][gq * 4 + e] * inv);
          up.h4[e] = (f16)(oacc[dt][(gq + 1) * 4 + e] * inv);
        }
        const u32x2 x = __builtin_amdgcn_permlane32_swap(lo.u[0], up.u[0], false, false);
        const u32x2 y = __builtin_amdgcn_permlane32_swap(lo.u[1], up.u[1], false, false);
        const u32x4 v = {x[0], y[0], x[1], y[1]};
        const int chunk = dt * 4 + gq + hi;   // columns chunk * 8 .. + 7 of query li
        *reinterpret_cast<u32x4*>(os + li * 128 + (k_phys_chunk<CPR>(li, chunk) << 4)) = v;
      }
    // (LDS operations of one wave execute in order: the reads below see the writes above)
    f16* obase = p.o + (long)b * p.o_bs + (long)h * DP;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = e * 8 + r8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(os + row * 128 + (pc << 4));
      const int qg = tile * 128 + wave * 32 + row;
      if (qg < p.nq) *reinterpret_cast<u32x4*>(obase + (long)qg * p.o_rs + k_logical_chunk<CPR>(row, pc) * 8) = v;
    }
    slot ^= 1;
  }
}

template <int NB>
int launch_attn_short2_nb(AttnParams& p, int grid, hipStream_t st) {
  constexpr int SMEM = 2 * (KVB * 64 * 2 + 2 * 32 * 128) + 4 * 8192;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_short2_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  hipLaunchKernelGGL(attn_short2_kernel<NB>, dim3(grid), dim3(256), SMEM, st, p);
  FMX_LAUNCH_CHECK("fmx_attention_f16 (short context, line-staged)");
  return FMX_OK;
}

int launch_attn_short2(AttnParams p, hipStream_t st) {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256;
    if (!(hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)) cus = 256;
    slots = 2 * cus;
  }
  p.qtiles = (p.nq + 127) / 128;
  const long total = (long)p.batch * p.heads * p.qtiles;
  const double q_span = ((double)(p.nq - 1) * p.q_rs + 64) * 2.0;
  if (q_span >= 2.0e9 || total >= (1L << 30)) return -1;
  p.q_span = (unsigned)q_span;
  const int grid = (int)(total < slots ? total : slots);
  const int nb = (p.nk + 31) / 32;
  switch (nb) {
    case 1: return launch_attn_short2_nb<1>(p, grid, st);
    case 2: return launch_attn_short2_nb<2>(p, grid, st);
    case 3: return launch_attn_short2_nb<3>(p, grid, st);
    default: return launch_attn_short2_nb<4>(p, grid, st);
  }
}

template <int DP>
int launch_attn_v2(AttnParams p, hipStream_t st) {
  constexpr int DVT = DP / 32;
  const int smem = 2 * (KVB * DP * 2 + DVT * 32 * 128);
  static int variant = -1, slots = 512, allow_split = 1;
  if (variant < 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      slots = (DP == 64 ? 2 : 1) * cus;   // 4-wave workgroups per CU at this kernel's register count
    const char* sp = fmx_knob("FMX_ATTN_SPLIT");   // A/B knob: 0 disables the key-split workgroups
    allow_split = sp ? atoi(sp) : 1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64v2_kernel<6, DP>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem);
    // A/B knob (tools/bench_kernels.py attn): FMX_ATTN_VARIANT=1 selects the first-generation kernels (d_head 64: round 1's 823 / 729 TFLOP/s at
    // N = 4096 / 1024; d_head 128: the generic 32-query kernel), 0 (default) the second-generation ones, 2 the wave-specialised kernel for
    // d_head 64 as well (d_head 128 uses it by default; at 64 it is slower than the symmetric kernel: 832 vs 1025 TFLOP/s at N = 4096), 3 the
    // symmetric 64-query kernel for d_head 128 too.
    const char* e = fmx_knob("FMX_ATTN_VARIANT");
    variant = e ? atoi(e) : 0;
    if (DP == 64) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  }
  p.qtiles = (p.nq + 255) / 256;
  const int grid = p.qtiles * p.heads * p.batch;
  // 32-bit byte offsets from the (batch, head) bases: K rows 0 .. nk_pad-1, V^T rows 0 .. DP-1 with nk_pad keys each
  const double k_span = ((double)(p.nk_pad - 1) * p.k_rs + DP) * 2.0, v_span = ((DP - 1.0) * p.vt_ds + p.nk_pad) * 2.0;
  if (variant != 1 && k_span < 2.0e9 && v_span < 2.0e9 && p.k_rs > 0 && p.vt_ds > 0) {
    p.k_span = (unsigned)k_span;
    p.vt_span = (unsigned)v_span;
    // tiles beyond the last full round of `slots` workgroups: as key-split workgroups (2 per tile, half as long each) when they would leave
    // CUs idle (fewer tiles than 3/4 of the CUs).  Measured (tools/bench_kernels.py attn, FMX_ATTN_SPLIT A/B): 640
    // tiles (SDXL 1024-token level at batch 8) 58.7 -> 52.5 us; with one tile per CU left (1280 tiles = 2.5 rounds) splitting gains nothing --
    // a workgroup alone on its CU already runs ~1.6x faster than one of two.  Needs an even number of key tiles to halve.
    const int ntiles = (p.nk + KVB - 1) / KVB;
    const int rem = grid % slots;
    const bool do_split = allow_split && rem > 0 && 8 * rem <= 3 * slots && ntiles >= 4 && (ntiles & 1) == 0;
    // d_head 128: the wave-specialised kernel (two waves per SIMD) unless this launch is one the key-split rule shortens (at most one round)
    if (variant == 2 || (variant == 0 && DP == 128 && (!do_split || grid > slots))) {
      constexpr int WS_SMEM = 2 * KVB * DP * 2 + 3 * DVT * 32 * 128 + 2 * 4 * (8 * 1024 + 1280) + 2048;
      static bool ws_attr = false;
      static int ws_rot = 0;
      if (!ws_attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_ws_kernel<6, DP>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_ws_kernel<6, DP, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_SMEM);
        // A/B knob (round 6): 1 = the S wave's sub-tile pipeline rotated across the tile boundary (K in a three-slot ring, see the kernel)
        const char* er = fmx_knob("FMX_ATTN_WS_ROT");
        ws_rot = er ? atoi(er) : FMX_ATTN_WS_ROT_DEFAULT;
        ws_attr = true;
      }
      // more than one round with a short tail: the full rounds here, the tail as key-split workgroups of the symmetric kernel behind it
      const bool tail = variant == 0 && do_split && grid > slots;
      p.nfull = tail ? grid - rem : grid;
      p.nsplit = 0;
      if (ws_rot) hipLaunchKernelGGL((attn_ws_kernel<6, DP, 1>), dim3(p.nfull), dim3(512), WS_SMEM, st, p);
      else hipLaunchKernelGGL((attn_ws_kernel<6, DP>), dim3(p.nfull), dim3(512), WS_SMEM, st, p);
      FMX_LAUNCH_CHECK("fmx_attention_f16 (wave-specialised)");
      if (tail) {
        p.bid0 = p.nfull;
        p.nsplit = 2 * rem;
        hipLaunchKernelGGL((attn_q64v2_kernel<6, DP>), dim3(p.nsplit), dim3(256), 2 * smem, st, p);
        FMX_LAUNCH_CHECK("fmx_attention_f16 (key-split tail)");
      }
      return FMX_OK;
    }
    if (DP == 64 && p.nk <= 128) {
      static int sk = -1;
      if (sk < 0) {
        // A/B knob: 0 = the looped 64-query kernels for short contexts as well (round 2), 1 = attn_short_kernel (lane-owned Q loads / O stores),
        // 2 (default) = attn_short2_kernel (Q and O in whole lines through LDS).  In a graph, every launch on its own Q / O tensors
        // (profiles/r08e): batch 16 x 20 heads x 1024 queries 32.4 / 31.3 / 27.5 us, 16 x 10 x 4096 queries 53.7 / 60.1 / 46.0 us.
        const char* e7 = fmx_knob("FMX_ATTN_SHORT");
        sk = e7 ? atoi(e7) : 2;
      }
      if (sk == 2) {
        const int rc = launch_attn_short2(p, st);
        if (rc >= 0) return rc;
      }
      if (sk) return launch_attn_short(p, st);
    }
    if (DP == 64 && !do_split) {
      static int v3 = -1;
      if (v3 < 0) {
        // A/B knob: 0 never, 1 (default) for short contexts, 2 for every launch without key-split workgroups.  Measured (tools/bench_kernels.py
        // attn, batch 16): 77 keys 31.1 -> 26.7 us (1024 queries, 20 heads) and 51.6 -> 43.4 us (4096 queries, 10 heads); 1024 keys 101.4 ->
        // 99.0 us; 4096 keys 692.6 -> 717.0 us -- with many key tiles the SIMD's two waves already overlap each other and the vector pipe is
        // the bound either way, with two tiles the per-wave chain is what counts.
        const char* e6 = fmx_knob("FMX_ATTN_V3");
        v3 = e6 ? atoi(e6) : 1;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64v3_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * smem / 2);
      }
      if (v3 == 2 || (v3 == 1 && ntiles <= 4)) {
        p.nfull = grid;
        p.nsplit = 0;
        hipLaunchKernelGGL(attn_q64v3_kernel<6>, dim3(grid), dim3(256), 3 * smem / 2, st, p);
        FMX_LAUNCH_CHECK("fmx_attention_f16 (sub-tile pipelined)");
        return FMX_OK;
      }
    }
    p.nfull = do_split ? grid - rem : grid;
    p.nsplit = do_split ? 2 * rem : 0;
#ifndef FMX_ELEM_BF16
    if (DP == 64) {
      static int npoly = -1;
      if (npoly < 0) {
        // A/B knob (development only): pairs per 16-key group whose exponentials run as the packed-fp16 polynomial (0 = none, the default; 2 = a
        // quarter, 3, 4 = half).  Measured and NOT adopted: profiles/r29_attention_packed_polynomial_exp2.jsonl -- every share is SLOWER
        // (4096 keys, hot: 671 us with v_exp_f32 only, 750 / 797 / 826 us with 2 / 3 / 4 pairs on the polynomial): the packed sequence costs
        // 8 VALU issues per pair (range shift, 3 pk_fma, pk_add magic, pk_lshl, pk_mul, clamp) against 2 quarter-rate transcendentals that
        // already overlap with the MFMA pipe, so the kernel's limiter is VALU issue slots, not the transcendental unit.
        const char* e8 = fmx_knob("FMX_ATTN_POLY");
        npoly = e8 ? atoi(e8) : 0;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64v2_kernel<6, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64v2_kernel<6, 64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_q64v2_kernel<6, 64, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * smem);
      }
      if (npoly >= 2 && npoly <= 4) {
        const dim3 g(p.nfull + p.nsplit);
        const size_t sm = do_split ? 2 * smem : smem;
        if (npoly == 2) hipLaunchKernelGGL((attn_q64v2_kernel<6, 64, 2>), g, dim3(256), sm, st, p);
        else if (npoly == 3) hipLaunchKernelGGL((attn_q64v2_kernel<6, 64, 3>), g, dim3(256), sm, st, p);
        else hipLaunchKernelGGL((attn_q64v2_kernel<6, 64, 4>), g, dim3(256), sm, st, p);
        FMX_LAUNCH_CHECK("fmx_attention_f16 (packed-polynomial exp2)");
        return FMX_OK;
      }
    }
#endif
    hipLaunchKernelGGL((attn_q64v2_kernel<6, DP>), dim3(p.nfull + p.nsplit), dim3(256), do_split ? 2 * smem : smem, st, p);
  } else if (DP == 64) {
    hipLaunchKernelGGL(attn_q64_kernel<0>, dim3(grid), dim3(256), smem, st, p);
  } else {
    return -1;   // caller falls back to the generic kernel
  }
  FMX_LAUNCH_CHECK("fmx_attention_f16 (64 queries per wave)");
  return FMX_OK;
}

template <int DP>
int launch_attn(const AttnParams& p, hipStream_t st) {
  constexpr int DVT = (DP + 31) / 32;
  const int smem = 2 * (KVB * DP * 2 + DVT * 32 * 128);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<DP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  const int grid = p.qtiles * p.heads * p.batch;
  hipLaunchKernelGGL((attn_kernel<DP>), dim3(grid), dim3(256), smem, st, p);
  FMX_LAUNCH_CHECK("fmx_attention_f16");
  return FMX_OK;
}

__global__ void softmax_rows_kernel(f16* __restrict__ x, int ncols, long ld) {
  __shared__ float red[8];
  f16* row = x + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int j = tid; j < ncols; j += blockDim.x) mx = fmaxf(mx, (float)row[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < ncols; j += blockDim.x) sum += __expf((float)row[j] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) sum += red[i];
  const float inv = 1.0f / sum;
  for (int j = tid; j < ncols; j += blockDim.x) row[j] = (f16)(__expf((float)row[j] - mx) * inv);
}

}  // namespace

extern "C" int fmx_attention_f16(const fmx_attn_args* a, void* stream) {
  FMX_REQUIRE(a && a->q && a->k && a->vt && a->o && a->zero_page, "attention: null pointer");
  FMX_REQUIRE(a->batch > 0 && a->heads > 0 && a->nq > 0 && a->nk > 0, "attention: bad dims");
  FMX_REQUIRE(a->nk_pad >= a->nk && (a->nk_pad % 64) == 0, "attention: nk_pad (%d) must be a multiple of 64 >= nk (%d)", a->nk_pad, a->nk);
  FMX_REQUIRE(fmx_aligned16(a->q) && fmx_aligned16(a->k) && fmx_aligned16(a->vt) && fmx_aligned16(a->o), "attention: 16-byte alignment");
  FMX_REQUIRE((a->q_rs % 8) == 0 && (a->k_rs % 8) == 0 && (a->vt_ds % 8) == 0 && (a->o_rs % 4) == 0 && (a->q_bs % 8) == 0 &&
                  (a->k_bs % 8) == 0 && (a->vt_bs % 8) == 0 && (a->vt_hs % 8) == 0, "attention: strides must be multiples of 8 elements");
  AttnParams p;
  p.q = (const f16*)a->q; p.k = (const f16*)a->k; p.vt = (const f16*)a->vt; p.o = (f16*)a->o;
  p.q_bs = a->q_bs; p.q_rs = a->q_rs; p.k_bs = a->k_bs; p.k_rs = a->k_rs;
  p.vt_bs = a->vt_bs; p.vt_hs = a->vt_hs; p.vt_ds = a->vt_ds; p.o_bs = a->o_bs; p.o_rs = a->o_rs;
  p.batch = a->batch; p.heads = a->heads; p.nq = a->nq; p.nk = a->nk; p.nk_pad = a->nk_pad;
  p.qtiles = (a->nq + 127) / 128;
  const bool force32 = a->scale < 0.f;
  p.scale_log2e = fabsf(a->scale) * 1.44269504088896340736f;
  p.zp = (const f16*)a->zero_page;
  p.causal = a->causal ? 1 : 0;
  p.k_span = p.vt_span = p.q_span = 0;
  p.nfull = p.nsplit = p.bid0 = 0;
  p.sk_tpw = p.sk_chunks = 0;
  p.mask = (const f16*)a->mask; p.mask_bs = a->mask_bs; p.mask_hs = a->mask_hs; p.mask_qs = a->mask_qs;
  p.inv_scale = 1.0f / fabsf(a->scale);
  FMX_REQUIRE(!p.mask || (fmx_aligned16(p.mask) && (p.mask_bs % 8) == 0 && (p.mask_hs % 8) == 0 && (p.mask_qs % 8) == 0),
              "attention: mask must be 16-byte aligned with strides that are multiples of 8 elements (rows padded to nk_pad keys)");
  FMX_REQUIRE(!p.causal || a->nq == a->nk, "attention: the causal mask is defined for self-attention (nq == nk)");
  hipStream_t st = (hipStream_t)stream;
  switch (a->dpad) {
    case 48: return launch_attn<48>(p, st);
    case 64:
      // 64-query-per-wave variant when there are enough queries to fill 256-query workgroups (test hook: scale < 0 forces
      // the 32-query kernel)
      if (a->nq >= 256 && !force32 && !p.causal && !p.mask) return launch_attn_v2<64>(p, st);
      return launch_attn<64>(p, st);
    case 80: return launch_attn<80>(p, st);
    case 128:
      if (a->nq >= 256 && !force32 && !p.causal && !p.mask) {
        const int rc = launch_attn_v2<128>(p, st);
        if (rc >= 0) return rc;
      }
      return launch_attn<128>(p, st);
    case 160: return launch_attn<160>(p, st);
    default: return fmx_set_error(FMX_E_UNSUPPORTED, "attention: dpad %d not in {48,64,80,128,160}", a->dpad);
  }
}

extern "C" int fmx_softmax_rows_f16(void* x, int64_t nrows, int32_t ncols, int64_t ld, void* stream) {
  FMX_REQUIRE(x && nrows > 0 && ncols > 0 && ld >= ncols, "softmax_rows: bad args");
  FMX_REQUIRE(nrows < (1LL << 31), "softmax_rows: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)nrows), dim3(256), 0, (hipStream_t)stream, (f16*)x, ncols, (long)ld);
  FMX_LAUNCH_CHECK("fmx_softmax_rows_f16");
  return FMX_OK;
}
