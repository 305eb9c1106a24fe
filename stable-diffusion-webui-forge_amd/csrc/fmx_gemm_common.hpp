// Shared pieces of the MFMA implicit-GEMM kernels (fmx_gemm.hip: 128x128 / 128x160 / 128x64 / 64x64 tiles, 4 waves;
// fmx_gemm256p.hip: 256x256 / 256x320 / 320x256 tiles, 8 waves, software-pipelined): launch parameters and the fused epilogue.
#pragma once
#include "fmx_common.hpp"

struct GemmParams {
  const f16* a0;
  const f16* a1;
  int c0, c1, s0, s1;  // channels and pixel strides (elements) of the two sources
  int n, h, w, oh, ow;
  int kh, stride, pad;
  int pad_x;        // horizontal padding (= pad except for the phase convolutions of fmx_conv3x3_up2x, whose 2 x 2 windows start at -1 or 0 per axis)
  int up_h, up_w;
  const f16* wgt;
  int ldw;
  int nout;
  const f16* bias;
  const f16* rowvec;
  int ld_rowvec;
  const f16* residual;
  int ld_res;
  float alpha;
  int act;
  void* out;
  int ld_out;
  int out_f32;
  const f16* zp;
  const f16* gate;
  int ld_gate;
  int M;            // n*oh*ow
  int kt;           // number of 64-wide K tiles = kh*kh*(c0+c1)/64
  int cpt;          // K tiles per tap = (c0+c1)/64
  int tiles_m, tiles_n;
  unsigned a0_bytes, a1_bytes, w_bytes;  // operand extents for the buffer descriptors of the 256x256 kernel
  float* stats;     // GroupNorm statistics of the output, partial[image][chunk][nout][2] (256-row tiles only; null = none)
  int stats_nch;    // chunks per image = oh*ow / 256
  // split-K (4-wave kernels, small problems): `splits` workgroups share an output tile, each over a contiguous range of K-tiles
  int splits;       // 1 = off
  float* ws;        // partial accumulators, [tile][split][BM*BN] fp32 in the kernel's register layout
  int* tickets;     // one arrival counter per tile, zero between launches
  // LayerNorm folded into the GEMMs around it (256x320 tile, see fmx_gemm256p.hip "LN"):
  float* row_stats;          // producer: per-row {sum, sum of squares} of the stored fp16 output, [M][2 * tiles_n][2]
  const float* ln_partial;   // consumer: the producer's array for this GEMM's INPUT rows, ln_parts entries per row
  int ln_parts;
  const float* ln_colsum;    // consumer: fp32 column sums of the (gamma-scaled) weight, [nout]
  float ln_eps, ln_inv_c;    // consumer: LayerNorm epsilon, 1 / normalised width
  // the operand-swapped consumer (320 x 256 tile, fmx.h ln_col_ab / ln_row_cb): LayerNorm rows = output columns
  const float* ln_col_ab;    // [nout][2] {rstd, -mean rstd}
  const float* ln_row_cb;    // [M][2] {colsum, folded bias}
  float* ln_ab_out;          // LN consumer: also write {rstd, -mean rstd} of its input rows (tiles of the first tile column), or null
  // cross-attention epilogue of the query projection (fmx.h xa_*): K / V^T of the text context, element strides, scale * log2(e)
  const f16* xa_k;
  const f16* xa_vt;
  int xa_k_rs, xa_k_bs, xa_vt_ds, xa_vt_bs, xa_nk, xa_rows;
  unsigned xa_k_bytes, xa_vt_bytes;
  float xa_c2;
  // scattered output rows (fmx_conv3x3_up2x: a phase's pixels are every second pixel of every second row of the output): row m is stored at
  // out + m * ld_out + (m / scat_ow) * scat_extra; 0 = off.  8-wave kernels with FA = 3 only, scat_ow a multiple of 32.
  int scat_ow;
  long scat_extra;
  int xtile;                 // persistent 256 x 320 linear kernels: fetch the next output tile's first K-tile under the last K-tile of this one (set by the launcher)
};

// K-tile depth of every GEMM kernel: 64 halfs = one 128-byte LDS row.
constexpr int FMX_BK = 64;

// Epilogue for one lane-owned run of 4 CONSECUTIVE output columns of output row m (the MFMA operands are issued
// swapped so that this is what a lane holds):  acc*alpha (+bias) (+rowvec[image]) -> (GEGLU) -> (+residual) -> store.
//   v   : the 4 accumulators (value half when geglu), g : the 4 gate accumulators (geglu only)
//   nb  : first weight row (GEMM column before GEGLU halving) of v; nbg : same for g
//   col : first output column
struct GemmEpilogue {
  const GemmParams& p;
  int per_img, ncols;
  bool geglu, vec_ok;
  __device__ __forceinline__ explicit GemmEpilogue(const GemmParams& q) : p(q) {
    per_img = p.oh * p.ow;
    geglu = p.act == FMX_ACT_GEGLU;
    ncols = geglu ? (p.nout >> 1) : p.nout;
    vec_ok = ((p.ld_out & 3) == 0) && ((p.ld_res & 3) == 0) && ((p.ld_rowvec & 3) == 0);
  }
  __device__ __forceinline__ const f16* rowvec_of(int m) const {
    return p.rowvec ? p.rowvec + (long)(m / per_img) * p.ld_rowvec : nullptr;
  }
  __device__ __forceinline__ void biased(const float (&a)[4], int nb, const f16* rv, float (&v)[4]) const {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a[r] * p.alpha;
    if (nb + 3 < p.nout && vec_ok) {
      if (p.bias) {
        const f16x4 b = *reinterpret_cast<const f16x4*>(p.bias + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
      }
      if (rv) {
        const f16x4 b = *reinterpret_cast<const f16x4*>(rv + nb);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)b[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nb + r < p.nout) {
          if (p.bias) v[r] += (float)p.bias[nb + r];
          if (rv) v[r] += (float)rv[nb + r];
        }
      }
    }
  }
  // v already biased (and GEGLU-combined); adds the residual and stores 4 columns starting at `col` of row m
  __device__ __forceinline__ void store(int m, int col, float (&v)[4]) const {
    if (col >= ncols) return;
    const long o = (long)m * p.ld_out + col;
    const bool full = (col + 3 < ncols) && vec_ok;
    if (p.residual) {
      const f16* rp = p.residual + (long)m * p.ld_res + col;
      if (full) {
        const f16x4 rr = *reinterpret_cast<const f16x4*>(rp);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < ncols) v[r] += (float)rp[r];
      }
    }
    if (p.out_f32) {
      float* op = reinterpret_cast<float*>(p.out) + o;
      if (full) {
        *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < ncols) op[r] = v[r];
      }
    } else {
      f16* op = reinterpret_cast<f16*>(p.out) + o;
      if (full) {
        f16x4 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (f16)v[r];
        *reinterpret_cast<f16x4*>(op) = hv;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < ncols) op[r] = (f16)v[r];
      }
    }
  }
};

// Branch-free epilogue for the common case (fp16 output, every leading dimension and nout a multiple of 4): absent
// operands (bias / rowvec / residual) are redirected to the zero page with index multiplier 0, so the code below is
// straight-line -- the compiler can put all of a row's loads in flight before the first use instead of one
// s_waitcnt vmcnt(0) per 4 columns.
struct FastEpilogue {
  const f16* bias;
  const f16* rowvec;
  const f16* res;
  f16* out;
  long ld_rv, ld_res, ld_out;
  const f16* gate;
  long ld_gt;
  int mb, mrv, mres, mgt;  // 1 when the operand exists, else 0
  bool gelu_tanh;
  int per_img, nout, ncols;
  float alpha;
  __device__ __forceinline__ explicit FastEpilogue(const GemmParams& p) {
    mb = p.bias ? 1 : 0;
    mrv = p.rowvec ? 1 : 0;
    mres = p.residual ? 1 : 0;
    mgt = p.gate ? 1 : 0;
    gate = p.gate ? p.gate : p.zp;
    ld_gt = (long)p.ld_gate * mgt;
    gelu_tanh = p.act == FMX_ACT_GELU_TANH;
    bias = p.bias ? p.bias : p.zp;
    rowvec = p.rowvec ? p.rowvec : p.zp;
    res = p.residual ? p.residual : p.zp;
    out = reinterpret_cast<f16*>(p.out);
    ld_rv = (long)p.ld_rowvec * mrv;
    ld_res = (long)p.ld_res * mres;
    ld_out = p.ld_out;
    per_img = p.oh * p.ow;
    nout = p.nout;
    ncols = p.act == FMX_ACT_GEGLU ? (p.nout >> 1) : p.nout;
    alpha = p.alpha;
  }
  static __host__ __device__ __forceinline__ bool eligible(const GemmParams& p) {
    return !p.out_f32 && (p.ld_out & 3) == 0 && (p.ld_res & 3) == 0 && (p.ld_rowvec & 3) == 0 && (p.ld_gate & 3) == 0 && (p.nout & 7) == 0;
  }
  // nb / col are multiples of 4; callers clamp them into range for the loads and predicate the store
  __device__ __forceinline__ f16x4 bias4(int nb) const { return *reinterpret_cast<const f16x4*>(bias + nb * mb); }
  __device__ __forceinline__ f16x4 rv4(int img, int nb) const { return *reinterpret_cast<const f16x4*>(rowvec + img * ld_rv + nb * mrv); }
  __device__ __forceinline__ f16x4 gate4(int img, int nb) const { return *reinterpret_cast<const f16x4*>(gate + img * ld_gt + nb * mgt); }
  // act (GELU-tanh when selected) then the optional gate: absent gate reads zeros -> factor 1
  __device__ __forceinline__ float act_gate(float v, float gt) const {
    if (gelu_tanh) v = gelu_tanh_f(v);
    return v * fmaf((float)mgt, gt - 1.0f, 1.0f);
  }
  __device__ __forceinline__ f16x4 res4(int m, int col) const { return *reinterpret_cast<const f16x4*>(res + m * ld_res + col * mres); }
  __device__ __forceinline__ void store4(int m, int col, const float (&v)[4]) const {
    f16x4 hv;
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = (f16)v[r];
    *reinterpret_cast<f16x4*>(out + m * ld_out + col) = hv;
  }
  // 8-wide (16-byte) forms for kernels whose lanes own 8 consecutive columns (after a v_permlane32_swap of the MFMA
  // accumulators): half the store / load instructions of the 4-wide forms -- the epilogue of a 256x256 tile is
  // store-ISSUE-bound, not bandwidth-bound.  Need eligible8(): every leading dimension a multiple of 8, 16-byte bases.
  static inline bool eligible8(const GemmParams& p) {
    return eligible(p) && (p.ld_out & 7) == 0 && (p.ld_res & 7) == 0 && (p.ld_rowvec & 7) == 0 && (p.ld_gate & 7) == 0 &&
           fmx_aligned16(p.out) && fmx_aligned16(p.bias) && fmx_aligned16(p.rowvec) && fmx_aligned16(p.residual) && fmx_aligned16(p.gate);
  }
  __device__ __forceinline__ f16x8 bias8(int nb) const { return *reinterpret_cast<const f16x8*>(bias + nb * mb); }
  __device__ __forceinline__ f16x8 rv8(int img, int nb) const { return *reinterpret_cast<const f16x8*>(rowvec + img * ld_rv + nb * mrv); }
  __device__ __forceinline__ f16x8 gate8(int img, int nb) const { return *reinterpret_cast<const f16x8*>(gate + img * ld_gt + nb * mgt); }
  __device__ __forceinline__ f16x8 res8(int m, int col) const { return *reinterpret_cast<const f16x8*>(res + m * ld_res + col * mres); }
  __device__ __forceinline__ void store8(int m, int col, const float (&v)[8]) const {
    f16x8 hv;
#pragma unroll
    for (int r = 0; r < 8; ++r) hv[r] = (f16)v[r];
    *reinterpret_cast<f16x8*>(out + m * ld_out + col) = hv;
  }
};

// v_permlane32_swap on a pair of fp32 registers: afterwards lanes 0-31 hold {a.lo, a.hi} and lanes 32-63 {b.lo, b.hi}
// in (a, b), i.e. the upper half-wave of `a` and the lower half-wave of `b` trade places.
__device__ __forceinline__ void swap_halfwaves(float& a, float& b) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// same tile, software-pipelined single-barrier schedule (fmx_gemm256p.hip)
int fmx_launch_gemm256p(const GemmParams& p, bool conv, int bm, int bn, hipStream_t st);  // (bm, bn) = (256,256) (256,320) (320,256)
// 256 x 160 tile, two 4-wave workgroups per CU (fmx_gemm4w.hip): linear GEMMs, optional LayerNorm producer / consumer / GEGLU
int fmx_launch_gemm4w(const GemmParams& p, hipStream_t st);
