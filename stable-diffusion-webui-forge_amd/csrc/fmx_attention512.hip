// Fused single-head attention of width 512 for gfx950: the VAE mid-block attention (backend/nn/vae.py:118-137 -> attention.py:412-422,
// one head, C = 512, N = H*W up to 16 384 tokens at 1024^2) without the N x N score matrix ever leaving the chip.
//
// Round 1 ran it as S = QK^T (GEMM) -> row softmax -> PV (GEMM) per image with a 512 MB fp16 score buffer.  A flash-style kernel for a
// 512-wide head does not fit the multi-head kernel's shape: 32 queries x 512 channels of fp32 output are 256 accumulator registers per
// wave.  So the head dimension is SPLIT ACROSS WAVES:
//   * workgroup = 8 waves = 2 query fragments (32 queries each) x 4 channel slices (128 channels each); 64 queries per workgroup;
//   * per 32-key step every wave computes the PARTIAL scores of its query fragment over its channel slice (8 MFMAs 32x32x16, K slice
//     from LDS, pre-scaled Q slice in registers), the four partials of a query fragment meet in LDS (one 4 KB record per wave, one
//     barrier), every wave of the fragment sums them in a fixed order -- all four hold the same scores, bit for bit -- and runs the same
//     online softmax (lane = query: maxima and sums are lane-local, as in fmx_attention.hip);
//   * then O^T[slice] += V^T[slice] P^T: 8 MFMAs per wave on its 128 x 32 output slice (64 accumulator registers).
// K / V^T tiles (32 keys: 32 KB + 32 KB) arrive by LDS-DMA, double buffered, 16-byte chunks XOR-swizzled on the source side and on the
// ds_read_b128 side; LDS: 2 x 64 KB stages + 32 KB exchange = 160 KB, one workgroup per CU.
// Rates: 16 MFMAs per wave against 16 fragment reads and a 20 KB partial-score exchange per step -> LDS-bound (PMC: matrix pipes 16-22 % busy,
// 30 % of wave time waiting on LDS), not matrix-pipe-bound; 538 TFLOP/s after the bank-conflict fix below (407 before), against 397 for the
// materialised path it replaces -- and no 512 MB score buffer or per-image host loop, see DESIGN.md section 4.2.
#include <stdlib.h>

#include "fmx_common.hpp"

namespace {

struct Attn512Params {
  const f16* q;
  const f16* k;
  const f16* vt;
  f16* o;
  long q_bs, q_rs, k_bs, k_rs, vt_bs, vt_ds, o_bs, o_rs;   // element strides: batch, token row (q / k / o); batch, channel row (V^T)
  int batch, nq, nk, qtiles;
  float scale_log2e;
  unsigned k_span, vt_span;   // bytes addressable from an image's K / V^T base (buffer descriptors: lanes beyond read zeros)
};

constexpr int C = 512, KV = 32, QT = 64;
constexpr int K_BYTES = KV * C * 2;        // 32 keys x 1024 B
constexpr int V_BYTES = C * KV * 2;        // 512 channel rows x 64 B
constexpr int STAGE = K_BYTES + V_BYTES;   // 64 KB
constexpr int XCH = 2 * STAGE;             // exchange area: 8 waves x 4 KB

template <int V>
struct IC { static constexpr int value = V; };

__device__ __forceinline__ int key_perm(int i) { return (i & 3) | (((i >> 3) & 3) << 2) | (((i >> 2) & 1) << 4); }
// LDS swizzles.  A ds_read_b128 is served 8 lanes per cycle (8 x 16 B = the 32 banks); rows of the K tile are 1024 B apart and rows of the V^T
// tile 64 B, so without a swizzle 8 consecutive lanes (= 8 rows) land on 1 or 2 sixteen-byte bank groups.  The XOR keys below give 8
// consecutive lanes 8 distinct groups: for K the row a lane reads is key_perm(lane), whose bits 0, 1 and 4 are the lane's low three bits; for
// V^T the rows are consecutive, two per 128 B, so bits 1-2 of the row spread the four rows that share a half.  (The first version keyed on
// other bits: SQ_LDS_BANK_CONFLICT was 64 % of SQ_LDS_IDX_ACTIVE, profiles/r05z_pmc_attention512.json.)
// (A variant that also gives the 8 even / odd lanes of a 16-lane pass distinct groups measured the same: 8.16 vs 8.18 ms, the same residual
// SQ_LDS_BANK_CONFLICT count.)
__device__ __forceinline__ int k_swz(int row) { return (row & 3) | (((row >> 4) & 1) << 2); }   // row = key_perm(lane)
__device__ __forceinline__ int v_swz(int row) { return (row >> 1) & 3; }

// NS = channel slices per query fragment: 4 (128 channels per wave, 8 waves = two per SIMD: the round-2 form) or 2 (256 channels per wave, 4 waves = ONE
// per SIMD on the whole 512-entry register file; round 3, slower: see the launcher).  With 4 slices a 32-key step costs a CU 256 KB of fragment reads + the exchange of 8 x 4 KB of
// fp32 partial scores, each partial read by 4 waves (LDS ~1 500 cycles against 1 024 of MFMA: LDS-bound, profiles/r05z); with 2 slices every fragment read
// feeds a wave that owns twice the channels: the same MFMAs, 160 KB of reads, half the exchange, two partials per sum.
template <int NS>
__global__ __launch_bounds__(NS * 128, 1) void attn512_kernel(const Attn512Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 2 * NS;              // waves
  constexpr int SC = C / NS;              // channels per slice
  constexpr int QS = SC / 16;             // 16-channel MFMA k-steps per slice
  constexpr int OD = SC / 32;             // 32-channel output blocks per slice
  constexpr int PW = 32 / NW;             // 1-KiB staging pieces per wave and operand (a stage is 32 K pieces + 32 V^T pieces)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qf = wave / NS, ds = wave % NS;   // query fragment, channel slice
  const int hi = lane >> 5, li = lane & 31;

  const int nwg = p.qtiles * p.batch;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int qt = wg % p.qtiles, b = wg / p.qtiles;
  const f16* kbase = p.k + (long)b * p.k_bs;
  const f16* vbase = p.vt + (long)b * p.vt_bs;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(kbase), 0, p.k_span, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(vbase), 0, p.vt_span, 0x00020000);

  // ---- Q slice of this wave, pre-scaled by scale * log2(e): B operand, lane = query li, k-slots hi*8.. of each 16-channel step ----
  const int q0 = qt * QT + qf * 32;
  const int qrow = min(q0 + li, p.nq - 1);
  const f16* qp = p.q + (long)b * p.q_bs + (long)qrow * p.q_rs + ds * SC + hi * 8;
  f16x8 qfr[QS];
#pragma unroll
  for (int s = 0; s < QS; ++s) {
    const f16x8 raw = *reinterpret_cast<const f16x8*>(qp + s * 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) qfr[s][e] = (f16)((float)raw[e] * p.scale_log2e);
  }

  // ---- staging: 64 one-KiB pieces per stage, 8 per wave.  K piece r = key row r (64 chunks of 16 B);
  //      V^T piece r = channel rows 16 r .. 16 r + 15 (4 chunks of 16 B each); chunk positions XOR-ed with k_swz / v_swz of the row ---------
  unsigned k_voff[PW], v_voff[PW];
#pragma unroll
  for (int e = 0; e < PW; ++e) {
    const int row = e * NW + wave;                                  // key row of this wave's e-th K piece
    k_voff[e] = (unsigned)row * (unsigned)p.k_rs * 2u + (unsigned)(lane ^ k_swz(row)) * 16u;
    const int vrow = (e * NW + wave) * 16 + (lane >> 2);            // channel row of this lane in the wave's e-th V^T piece
    v_voff[e] = (unsigned)vrow * (unsigned)p.vt_ds * 2u + (unsigned)((lane & 3) ^ v_swz(vrow)) * 16u;
  }
  const unsigned k_step = (unsigned)KV * (unsigned)p.k_rs * 2u;
  auto stage = [&](auto SI, int kt) {
    constexpr int S = decltype(SI)::value;
#pragma unroll
    for (int e = 0; e < PW; ++e) {
      auto* dk = (__attribute__((address_space(3))) void*)(smem + S * STAGE + (e * NW + wave) * 1024);
      auto* dv = (__attribute__((address_space(3))) void*)(smem + S * STAGE + K_BYTES + (e * NW + wave) * 1024);
      const unsigned kv = k_voff[e], vv = v_voff[e];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, dk, 16, kv, (unsigned)kt * k_step, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, dv, 16, vv, (unsigned)kt * (KV * 2u), 0, 0);
    }
  };

  f32x16 oacc[OD];
#pragma unroll
  for (int i = 0; i < OD; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // log2 domain

  const int nsteps = (p.nk + KV - 1) / KV;
  stage(IC<0>{}, 0);
  wait_vmcnt0();
  __syncthreads();

  const int krow = key_perm(li);
  // exchange records: 4 KB per wave, laid out [quarter v][lane] (16 B each): consecutive lanes write / read consecutive 16-byte groups
  float* xw = reinterpret_cast<float*>(smem + XCH) + wave * 1024 + lane * 4;
  const float* xr = reinterpret_cast<const float*>(smem + XCH) + (qf * NS) * 1024 + lane * 4;
  auto step = [&](auto SI, int kt) {
    constexpr int S = decltype(SI)::value;
    if (kt + 1 < nsteps) stage(IC<S ^ 1>{}, kt + 1);
    const char* sk = smem + S * STAGE;
    const char* sv = sk + K_BYTES;
    // partial scores over this wave's SC channels: S^T[key][query]
    f32x16 part;
#pragma unroll
    for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
    for (int s = 0; s < QS; ++s) {
      const int chunk = ds * (2 * QS) + s * 2 + hi;
      const f16x8 kf = *reinterpret_cast<const f16x8*>(sk + krow * 1024 + ((chunk ^ k_swz(krow)) << 4));
      part = FMX_MFMA_32x32x16(kf, qfr[s], part);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(xw + v * 256) = f32x4{part[v * 4], part[v * 4 + 1], part[v * 4 + 2], part[v * 4 + 3]};
    // raw barrier: the next stage's LDS-DMA stays in flight across it (a __syncthreads() would drain it: vmcnt(0)); only the exchange
    // records have to be visible
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    f32x16 sc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f32x4 a = *reinterpret_cast<const f32x4*>(xr + v * 256);
#pragma unroll
      for (int w2 = 1; w2 < NS; ++w2) a += *reinterpret_cast<const f32x4*>(xr + w2 * 1024 + v * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[v * 4 + e] = a[e];
    }
    // lane holds query li against keys kt*32 + hi*16 + r
    if ((kt + 1) * KV > p.nk) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kt * KV + hi * 16 + r >= p.nk) sc[r] = -INFINITY;
    }
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first step: exp2(-inf) = 0 on O = l = 0
    float psum = 0.f;
    f16x8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(sc[r] - m_new);
      psum += e;
      pf[r >> 3][r & 7] = (f16)e;
    }
    l_run = l_run * alpha + psum;
    if (__any(m_new > m_run)) {
#pragma unroll
      for (int i = 0; i < OD; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    m_run = m_new;
    // O^T[slice] += V^T[slice rows][32 keys] P^T : A = V^T rows (channel), B = P^T (k-slot hi*8+e <-> key hi*16 + j*8 + e)
#pragma unroll
    for (int dt = 0; dt < OD; ++dt) {
      const int row = ds * SC + dt * 32 + li;
      const char* rp = sv + row * 64;
      const int swz = v_swz(row);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f16x8 vf = *reinterpret_cast<const f16x8*>(rp + (((hi * 2 + j) ^ swz) << 4));
        oacc[dt] = FMX_MFMA_32x32x16(vf, pf[j], oacc[dt]);
      }
    }
    wait_vmcnt0();
    __syncthreads();
  };
  for (int kt = 0; kt < nsteps; kt += 2) {
    step(IC<0>{}, kt);
    if (kt + 1 < nsteps) step(IC<1>{}, kt + 1);
  }

  // ---- finish: O[b][q][slice channels] = O^T / l ; 16-byte stores after a half-wave swap (see fmx_attention.hip) -------------------------
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x2 lw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
  const float inv = 1.0f / (__uint_as_float(lw[0]) + __uint_as_float(lw[1]));
  const int qg = q0 + li;
  f16* op = p.o + (long)b * p.o_bs + (long)qg * p.o_rs + ds * SC;
#pragma unroll
  for (int dt = 0; dt < OD; ++dt)
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

}  // namespace

extern "C" int fmx_attention_single_head512_f16(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* vt,
                                                int64_t vt_bs, int64_t vt_ds, void* o, int64_t o_bs, int64_t o_rs, int32_t batch, int32_t nq,
                                                int32_t nk, int32_t nk_pad, float scale, void* stream) {
  FMX_REQUIRE(q && k && vt && o, "attention512: null pointer");
  FMX_REQUIRE(batch > 0 && nq > 0 && nk > 0 && nk_pad >= nk && (nk_pad % 32) == 0, "attention512: bad dims (nk_pad = keys present per V^T row, multiple of 32)");
  FMX_REQUIRE(fmx_aligned16(q) && fmx_aligned16(k) && fmx_aligned16(vt) && fmx_aligned16(o), "attention512: 16-byte alignment");
  FMX_REQUIRE((q_rs % 8) == 0 && (k_rs % 8) == 0 && (vt_ds % 8) == 0 && (o_rs % 8) == 0 && (q_bs % 8) == 0 && (k_bs % 8) == 0 && (vt_bs % 8) == 0 && (o_bs % 8) == 0,
              "attention512: strides must be multiples of 8 elements");
  const double k_span = ((double)(nk - 1) * k_rs + C) * 2.0 /* rows >= nk read as zeros */, v_span = ((double)(C - 1) * vt_ds + nk_pad) * 2.0;
  FMX_REQUIRE(k_span < 2.0e9 && v_span < 2.0e9 && k_rs >= C && vt_ds >= nk_pad, "attention512: an image's K / V^T must span less than 2 GB");
  Attn512Params p;
  p.q = (const f16*)q; p.k = (const f16*)k; p.vt = (const f16*)vt; p.o = (f16*)o;
  p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.vt_bs = vt_bs; p.vt_ds = vt_ds; p.o_bs = o_bs; p.o_rs = o_rs;
  p.batch = batch; p.nq = nq; p.nk = nk; p.qtiles = (nq + QT - 1) / QT;
  p.scale_log2e = scale * 1.44269504088896340736f;
  p.k_span = (unsigned)k_span; p.vt_span = (unsigned)v_span;
  const int smem = 2 * STAGE + 8 * 4096;
  static int slices = 0;
  if (!slices) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn512_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn512_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    // A/B knob: 4 (default) = eight waves of 128 channels, 2 = four waves of 256 channels.  Measured (8 x 16 384 tokens, tools/bench_kernels.py attn512,
    // profiles/r08p): 8.12 ms vs 9.67 ms -- with one wave per SIMD nothing covers the fragment-read and exchange latencies around the two barriers of a
    // step; the LDS traffic saved (-40 %) does not pay for it.
    const char* e = fmx_knob("FMX_ATTN512_SLICES");
    slices = (e && atoi(e) == 2) ? 2 : 4;
  }
  if (slices == 4) hipLaunchKernelGGL(attn512_kernel<4>, dim3(p.qtiles * batch), dim3(512), smem, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn512_kernel<2>, dim3(p.qtiles * batch), dim3(256), smem, (hipStream_t)stream, p);
  FMX_LAUNCH_CHECK("fmx_attention_single_head512_f16");
  return FMX_OK;
}

