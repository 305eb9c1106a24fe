// GroupNorm(+SiLU) over NHWC fp16 and LayerNorm over rows, gfx950.  HBM-bound kernels: 16-byte vector
// loads/stores, fp32 statistics, no atomics (deterministic two-level reduction).
// Replaces F.group_norm (backend/operations.py:308; eps 1e-5 ResBlock/out, 1e-6 SpatialTransformer/VAE) fused
// with the SiLU that follows it in ResBlock.in_layers/out_layers (backend/nn/unet.py:394-398,417-421) and the
// torch.cat of the skip tensor in front of it (:741); and F.layer_norm (backend/operations.py:327).
#include <stdlib.h>

#include "fmx_common.hpp"

namespace {

// ---- statistics: per-(image, chunk, channel) partial sum / sum of squares ------------------------------------------------
// partial[img][chunk][c] = {sum x, sum x^2} over the chunk's pixels (fp32).  This kernel is the FALLBACK producer: the 256-row GEMM
// tiles emit the same records from their epilogue (fmx_gemm256p.hip, one chunk = one 256-row tile), so that a GroupNorm whose input
// was just written by a convolution never re-reads it for statistics.
// grid (nchunks, n); block 256 threads; thread t owns channel octet (t % oct) and walks pixels t / oct + k*lanes, four independent
// 16-byte loads in flight per thread (the kernel is latency-, not bandwidth-bound at 8 waves per CU otherwise).
__global__ __launch_bounds__(256) void gn_stats_kernel(const f16* __restrict__ x, int c, long ld, int hw, float* __restrict__ partial,
                                                        int nchunks) {
  extern __shared__ float sred[];  // [256][16] only when several pixel-lanes share an octet
  const int oct = c >> 3;
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int per = (hw + nchunks - 1) / nchunks;
  const int p_begin = chunk * per;
  const int p_end = min(hw, p_begin + per);
  const int tid = threadIdx.x;
  float* out = partial + ((long)(img * nchunks + chunk) * c) * 2;

  const int lanes = max(1, 256 / oct);          // pixel lanes per octet (when oct <= 256)
  for (int ob = 0; ob < oct; ob += 256) {       // octet blocks (oct > 256 only for C > 2048)
    const int my_oct = ob + (oct >= 256 ? tid : tid % oct);
    const int my_lane = oct >= 256 ? 0 : tid / oct;
    const int nl = oct >= 256 ? 1 : lanes;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const bool active = my_oct < oct && my_lane < nl;
    if (active) {
      const f16* src = x + (long)img * hw * ld + my_oct * 8;
      int px = p_begin + my_lane;
      for (; px + 3 * nl < p_end; px += 4 * nl) {
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f16x8*>(src + (long)(px + u * nl) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[u][e];
            s[e] += f;
            q[e] += f * f;
          }
      }
      for (; px < p_end; px += nl) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(src + (long)px * ld);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v[e];
          s[e] += f;
          q[e] += f * f;
        }
      }
    }
    if (nl == 1) {
      if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          out[(my_oct * 8 + e) * 2 + 0] = s[e];
          out[(my_oct * 8 + e) * 2 + 1] = q[e];
        }
      }
    } else {
      // combine the pixel lanes of each octet through LDS (fixed order -> deterministic)
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sred[tid * 16 + e] = s[e];
        sred[tid * 16 + 8 + e] = q[e];
      }
      __syncthreads();
      if (tid < oct) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float ss = 0.f, qq = 0.f;
          for (int l = 0; l < nl; ++l) {
            const int src_t = l * oct + tid;
            if (src_t < 256) {
              ss += sred[src_t * 16 + e];
              qq += sred[src_t * 16 + 8 + e];
            }
          }
          out[(tid * 8 + e) * 2 + 0] = ss;
          out[(tid * 8 + e) * 2 + 1] = qq;
        }
      }
    }
  }
}

// ---- finalize: fold the chunk partials of one (image, group) into per-channel (scale, shift) ---------------------------------------
// grid (groups, n), one wave per block.  The channels of the (virtually concatenated) input come from up to two partial buffers with
// their own chunk counts (the two producers may have tiled differently).  Item i of a group and source = (channel i / nch, chunk i % nch);
// lane l takes items l, l + 64, ...; the wave sum is a fixed butterfly -> deterministic.  ss[img][c] = {rstd * gamma, beta - mean * rstd * gamma}.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ p0, int nch0, int c0, const float* __restrict__ p1, int nch1, int c1,
                                                          int groups, int hw, float eps, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                          float* __restrict__ ss) {
  const int C = c0 + c1;
  const int cpg = C / groups;
  const int g = blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
  float s = 0.f, q = 0.f;
  // the group's (channel, chunk) records of each source, flattened over the lanes: all loads of a lane are independent (a loop over
  // channels with a few chunks each leaves most lanes idle behind one L2 round trip per channel)
  auto accumulate = [&](const float* base, int nch, int cs, int lo, int hi) {   // source-local channels [lo, hi)
    const int items = (hi - lo) * nch;
    const float* b0 = base + ((long)img * nch * cs + lo) * 2;
    for (int i = lane; i < items; i += 64) {
      const int ci = i / nch, ch = i - ci * nch;
      const f32x2 v = *reinterpret_cast<const f32x2*>(b0 + ((long)ch * cs + ci) * 2);
      s += v[0];
      q += v[1];
    }
  };
  const int g_lo = g * cpg, g_hi = g_lo + cpg;
  if (g_lo < c0) accumulate(p0, nch0, c0, g_lo, min(g_hi, c0));
  if (g_hi > c0) accumulate(p1, nch1, c1, max(g_lo, c0) - c0, g_hi - c0);
  s = wave_sum(s);
  q = wave_sum(q);
  const float cnt = (float)cpg * (float)hw;
  const float mean = s / cnt;
  const float rstd = rsqrtf(fmaxf(q / cnt - mean * mean, 0.f) + eps);
  for (int ci = lane; ci < cpg; ci += 64) {
    const int c = g * cpg + ci;
    const float sc = rstd * (float)gamma[c];
    *reinterpret_cast<f32x2*>(ss + ((long)img * C + c) * 2) = f32x2{sc, (float)beta[c] - mean * sc};
  }
}

// ---- apply: normalise + affine (+ SiLU), writing the (concatenated) fp16 NHWC tensor: 1 read + 1 write ---------------------------
// grid (pixel tiles, n); the image's per-channel scale / shift are staged once per block in LDS; element i = tid + k*256 over the
// (pixel, octet) grid, four independent 16-byte loads in flight per thread.
// FUSE (round 5): no gn_finalize launch in front -- every block folds its image's chunk partials into the scale / shift table itself (32 groups,
// fixed summation order: every block of an image computes the same bits).  The partials of one image are (channels x chunks) 8-byte records out
// of L2; the launcher fuses up to 12 288 of them (96 KB per block), which covers the UNet's tensors up to 32^2 x 2560 / 64^2 x 640 (batch 1-8: the
// tensors whose apply launch is 15-20 us and whose finalize launch was another 7-12 us of pure latency, 46-61 launches per forward); larger
// tensors keep the separate finalize (8 us against a 45-90 us apply).
struct GnFuse {
  const float* p0; const float* p1;   // chunk partials of the two sources [n][nch][c][2]
  int nch0, nch1;
  const f16* gamma; const f16* beta;
  float eps;
};

template <bool SILU, int UNR = 4, bool NT = false, bool FUSE = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int c0, int c1, long ld0, long ld1,
                                                        int hw, const float* __restrict__ ss_g, f16* __restrict__ y, int pix_per_block, GnFuse fu) {
  extern __shared__ __attribute__((aligned(16))) float ss[];  // [C] scale, [C] shift (FUSE: + 64 floats of group sums)
  const int C = c0 + c1;
  const int img = blockIdx.y;
  float* scale = ss;
  float* shift = ss + C;
  const int tid = threadIdx.x;
  if (FUSE) {
    // 1. per channel: sum of its chunk records, in chunk order
    for (int c = tid; c < C; c += 256) {
      const bool second = c >= c0;
      const int cl = second ? c - c0 : c, cs = second ? c1 : c0, nch = second ? fu.nch1 : fu.nch0;
      const float* b = (second ? fu.p1 : fu.p0) + ((long)img * nch * cs + cl) * 2;
      float s = 0.f, q = 0.f;
      for (int ch = 0; ch < nch; ++ch) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(b + (long)ch * cs * 2);
        s += v[0];
        q += v[1];
      }
      scale[c] = s;
      shift[c] = q;
    }
    __syncthreads();
    // 2. per group (32 of them, 8 threads each): channels j, j + 8, ... of the group, then a fixed 8-lane butterfly
    const int cpg = C >> 5;
    {
      const int g = tid >> 3, j = tid & 7;
      float s = 0.f, q = 0.f;
      for (int ci = j; ci < cpg; ci += 8) {
        s += scale[g * cpg + ci];
        q += shift[g * cpg + ci];
      }
#pragma unroll
      for (int d = 1; d < 8; d <<= 1) {
        s += __shfl_xor(s, d);
        q += __shfl_xor(q, d);
      }
      __syncthreads();   // every read of the per-channel sums is done before step 3 overwrites them
      if (j == 0) {
        ss[2 * C + g] = s;
        ss[2 * C + 32 + g] = q;
      }
    }
    __syncthreads();
    // 3. per channel: scale = rstd gamma, shift = beta - mean rstd gamma
    const float cnt = (float)cpg * (float)hw;
    for (int c = tid; c < C; c += 256) {
      const int g = c / cpg;
      const float mean = ss[2 * C + g] / cnt;
      const float rstd = rsqrtf(fmaxf(ss[2 * C + 32 + g] / cnt - mean * mean, 0.f) + fu.eps);
      const float sc = rstd * (float)fu.gamma[c];
      scale[c] = sc;
      shift[c] = (float)fu.beta[c] - mean * sc;
    }
  } else {
    const float* simg = ss_g + (long)img * C * 2;
    for (int c = tid; c < C; c += 256) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(simg + (long)c * 2);
      scale[c] = v[0];
      shift[c] = v[1];
    }
  }
  __syncthreads();
  const int oct = C >> 3;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(hw, p0 + pix_per_block);
  // walk (px, o) incrementally instead of dividing
  int px = p0 + tid / oct;
  int o = tid - (tid / oct) * oct;
  const int dpx = 256 / oct, dov = 256 - dpx * oct;
  const long ibase = (long)img * hw;
  while (px < p1) {
    int pxs[UNR], chs[UNR];
    f16x8 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      pxs[u] = px;
      chs[u] = o * 8;
      if (px < p1) {
        const f16* src = (chs[u] < c0) ? x0 + (ibase + px) * ld0 + chs[u] : x1 + (ibase + px) * ld1 + (chs[u] - c0);
        if (NT) {
          typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
          union { u32x4_ u4; f16x8 h; } cv;
          cv.u4 = __builtin_nontemporal_load(reinterpret_cast<const u32x4_*>(src));
          v[u] = cv.h;
        } else {
          v[u] = *reinterpret_cast<const f16x8*>(src);
        }
      }
      px += dpx;
      o += dov;
      if (o >= oct) { o -= oct; ++px; }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (pxs[u] < p1) {
        const int ch = chs[u];
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(scale + ch), sc1 = *reinterpret_cast<const f32x4*>(scale + ch + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(shift + ch), sh1 = *reinterpret_cast<const f32x4*>(shift + ch + 4);
        f16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = fmaf((float)v[u][e], e < 4 ? sc0[e & 3] : sc1[e & 3], e < 4 ? sh0[e & 3] : sh1[e & 3]);
          if (SILU) f = silu_f(f);
          r[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(y + (ibase + pxs[u]) * C + ch) = r;
      }
    }
  }
}


// ---- LayerNorm: one wave per row, row kept in registers, exact two-pass mean/variance in fp32 -----------------
// MOD = false: y = LN(x) * gamma + beta, row r written to output row (r / rows_per_b) * ld_mod + r % rows_per_b (token rows re-spaced
// to a padded per-image stride for the attention key tiles; identity when rows_per_b == ld_mod).   MOD = true (Flux adaLN, backend/nn/flux.py:209-210,286,326): LN has no affine and
// y = (1 + scale[b]) * LN(x) + shift[b] with per-batch vectors gamma := scale + b*ld_mod, beta := shift + b*ld_mod, b = row / rows_per_b.
template <int MAXOCT, bool MOD>  // octets per lane
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma,
                                                  const f16* __restrict__ beta, f16* __restrict__ y, long rows, int c, float eps,
                                                  long rows_per_b, long ld_mod) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int oct = c >> 3;
  const f16* xr = x + row * c;
  f16x8 v[MAXOCT];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXOCT; ++j) {
    const int o = lane + j * 64;
    if (o < oct) {
      v[j] = *reinterpret_cast<const f16x8*>(xr + o * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += (float)v[j][e];
    }
  }
  const float mean = wave_sum(s) / (float)c;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXOCT; ++j) {
    const int o = lane + j * 64;
    if (o < oct) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[j][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
  f16* yr = y + (MOD ? row : (row / rows_per_b) * ld_mod + row % rows_per_b) * c;
  const long moff = MOD ? (row / rows_per_b) * ld_mod : 0;
#pragma unroll
  for (int j = 0; j < MAXOCT; ++j) {
    const int o = lane + j * 64;
    if (o < oct) {
      const f16x8 g = *reinterpret_cast<const f16x8*>(gamma + moff + o * 8);
      const f16x8 b = *reinterpret_cast<const f16x8*>(beta + moff + o * 8);
      f16x8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (f16)(((float)v[j][e] - mean) * rstd * ((MOD ? 1.0f : 0.0f) + (float)g[e]) + (float)b[e]);
      *reinterpret_cast<f16x8*>(yr + o * 8) = r;
    }
  }
}

// row statistics of a LayerNorm-folded GEMM's producer -> {rstd, -mean rstd} per row (the operand-swapped consumer reads them per output column)
__global__ __launch_bounds__(256) void ln_rowstats_finalize_kernel(const float* __restrict__ partial, int parts, long rows, float inv_c, float eps,
                                                                    float* __restrict__ ab) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  const float* q = partial + m * (parts * 2);
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < parts; ++k) {   // same order as the in-kernel form of the 256 x 320 consumer (fmx_gemm256p.hip, LN == 2)
    s1 += q[2 * k];
    s2 += q[2 * k + 1];
  }
  const float mean = s1 * inv_c;
  const float rstd = rsqrtf(fmaxf(s2 * inv_c - mean * mean, 0.f) + eps);
  *reinterpret_cast<f32x2*>(ab + m * 2) = f32x2{rstd, -mean * rstd};
}

}  // namespace

#ifndef FMX_ELEM_BF16   // fp32 in, fp32 out: one copy serves both element types
extern "C" int fmx_layernorm_rowstats_finalize(const float* row_partial, int32_t parts, int64_t rows, int32_t c, float eps, float* ab, void* stream) {
  FMX_REQUIRE(row_partial && ab && parts >= 1 && parts <= 64 && rows > 0 && c > 0, "layernorm_rowstats_finalize: bad args");
  hipLaunchKernelGGL(ln_rowstats_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row_partial, parts, (long)rows,
                     1.0f / (float)c, eps, ab);
  FMX_LAUNCH_CHECK("fmx_layernorm_rowstats_finalize");
  return FMX_OK;
}
#endif

int fmx_launch_gn_stats(const void* x, int32_t c, int64_t ld, int32_t n, int32_t hw, float* partial, int32_t nchunks, hipStream_t st) {
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, n), dim3(256), 256 * 16 * sizeof(float), st, (const f16*)x, c, (long)ld, hw, partial, nchunks);
  FMX_LAUNCH_CHECK("fmx_groupnorm_stats_f16");
  return FMX_OK;
}

// scale / shift table [n][c][2] of a one-source GroupNorm from its chunk partials -- for kernels that apply the norm themselves (fmx_conv_patch.hip)
int fmx_launch_gn_finalize(const float* partial, int32_t nchunks, int32_t c, int32_t n, int32_t groups, int32_t hw, float eps, const void* gamma,
                           const void* beta, float* scale_shift, hipStream_t st) {
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n), dim3(64), 0, st, partial, nchunks, c, (const float*)nullptr, 1, 0, groups, hw, eps,
                     (const f16*)gamma, (const f16*)beta, scale_shift);
  FMX_LAUNCH_CHECK("fmx_groupnorm_finalize");
  return FMX_OK;
}

extern "C" int fmx_groupnorm_stats_f16(const void* x, int32_t c, int64_t ld, int32_t n, int32_t hw, float* partial, int32_t nchunks, void* stream) {
  FMX_REQUIRE(x && partial && c > 0 && (c % 8) == 0 && ld >= c && (ld % 8) == 0 && n > 0 && hw > 0, "groupnorm_stats: bad args");
  FMX_REQUIRE(nchunks >= 1 && nchunks <= 1024, "groupnorm_stats: nchunks out of range");
  FMX_REQUIRE(fmx_aligned16(x), "groupnorm_stats: alignment");
  return fmx_launch_gn_stats(x, c, ld, n, hw, partial, nchunks, (hipStream_t)stream);
}

extern "C" int fmx_groupnorm_apply_f16(const void* x0, const void* x1, int32_t c0, int32_t c1, int64_t ld0, int64_t ld1, int32_t n, int32_t hw,
                                       const float* partial0, int32_t nchunks0, const float* partial1, int32_t nchunks1, int32_t groups, float eps,
                                       const void* gamma, const void* beta, int32_t silu, float* scale_shift, void* y, void* stream) {
  FMX_REQUIRE(x0 && partial0 && gamma && beta && y && scale_shift, "groupnorm_apply: null pointer");
  const int C = c0 + c1;
  FMX_REQUIRE(c0 > 0 && c1 >= 0 && (c0 % 8) == 0 && (c1 % 8) == 0 && groups > 0 && (C % groups) == 0, "groupnorm_apply: bad channels");
  FMX_REQUIRE(c1 == 0 || (x1 && partial1 && nchunks1 >= 1), "groupnorm_apply: second source incomplete");
  FMX_REQUIRE(nchunks0 >= 1 && n > 0 && hw > 0 && ld0 >= c0 && (ld0 % 8) == 0 && (c1 == 0 || (ld1 >= c1 && (ld1 % 8) == 0)), "groupnorm_apply: bad geometry");
  FMX_REQUIRE(fmx_aligned16(x0) && fmx_aligned16(y) && (!x1 || fmx_aligned16(x1)), "groupnorm_apply: alignment");
  FMX_REQUIRE((size_t)(2 * C) * sizeof(float) <= 160 * 1024, "groupnorm_apply: too many channels for the LDS scale/shift table");
  hipStream_t st = (hipStream_t)stream;
  static int ppb_bytes = 0;
  if (!ppb_bytes) {
    // A/B knob: activation bytes per block.  16 KB = one round of 4 x 16 bytes per thread: many short blocks beat few long ones on every
    // shape of the SDXL UNet and VAE (in a graph, each launch on its own tensors, profiles/r08l: 128^2 x 320 91 -> 76 us, 32^2 x 1280 34 -> 28 us,
    // 1024^2 x 128 818 -> 745 us against round 2's 64 KB; 8 / 12 KB and streaming loads are level or mixed)
    const char* e = fmx_knob("FMX_GN_BLOCK_KB");
    ppb_bytes = (e ? atoi(e) : 16) * 512;        // (elements: 2 bytes each)
  }
  auto ppb_for = [&](int c) { const int v = (ppb_bytes + c - 1) / c; return v < 1 ? 1 : v; };
  static int fuse_mode = -1;
  if (fuse_mode < 0) {
    const char* e = fmx_knob("FMX_GN_FUSE");   // A/B knob: 0 = always the separate finalize launch (rounds 2-4), 1 (default) = folded into the apply blocks where small
    fuse_mode = e ? atoi(e) : 1;
  }
  const long records = (long)c0 * nchunks0 + (long)c1 * (c1 ? nchunks1 : 0);
  // ... and only where the launch is a few blocks per CU: every block repeats the fold, so on the batch-8 tensors (2 560 blocks each) the fused form
  // cost 0.68 ms per forward more than the 46 finalize launches it removed (profiles/r26: 103.85 -> 104.33 ms per step); at batch 1-4 (320-640 blocks) it wins
  const long blocks_total = (long)n * ((hw + (ppb_for(C) - 1)) / ppb_for(C));
  const bool fuse = fuse_mode != 0 && groups == 32 && records <= 12288 && blocks_total <= 1024;
  if (!fuse)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n), dim3(64), 0, st, partial0, nchunks0, c0, partial1, c1 ? nchunks1 : 1, c1, groups, hw, eps,
                       (const f16*)gamma, (const f16*)beta, scale_shift);
  const GnFuse fu{partial0, partial1, nchunks0, c1 ? nchunks1 : 1, (const f16*)gamma, (const f16*)beta, eps};
  int ppb = (ppb_bytes + C - 1) / C;
  if (ppb < 1) ppb = 1;
  const int tiles = (hw + ppb - 1) / ppb;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<true, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<true, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<true, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<true, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<false, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<true, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_apply_kernel<false, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  static int variant = -1;
  if (variant < 0) {
    const char* e = fmx_knob("FMX_GN_VARIANT");   // A/B knob (tools/bench_kernels.py gnapply): 0 = 4 loads in flight per thread, 1 = 8, 2 = 4 streaming (nt) loads, 3 = 8 nt
    variant = e ? atoi(e) : 0;
  }
#define FMX_GN_LAUNCH(S, U, N)                                                                                                                   \
  hipLaunchKernelGGL((gn_apply_kernel<S, U, N>), dim3(tiles, n), dim3(256), (size_t)(2 * C) * sizeof(float), st, (const f16*)x0, (const f16*)x1, c0, c1, \
                     (long)ld0, (long)ld1, hw, scale_shift, (f16*)y, ppb, fu)
#define FMX_GN_LAUNCH_FUSED(S)                                                                                                                    \
  hipLaunchKernelGGL((gn_apply_kernel<S, 4, false, true>), dim3(tiles, n), dim3(256), (size_t)(2 * C + 64) * sizeof(float), st, (const f16*)x0,    \
                     (const f16*)x1, c0, c1, (long)ld0, (long)ld1, hw, scale_shift, (f16*)y, ppb, fu)
  if (fuse) {
    if (silu) FMX_GN_LAUNCH_FUSED(true); else FMX_GN_LAUNCH_FUSED(false);
  } else if (silu) {
    if (variant == 1) FMX_GN_LAUNCH(true, 8, false); else if (variant == 2) FMX_GN_LAUNCH(true, 4, true); else if (variant == 3) FMX_GN_LAUNCH(true, 8, true);
    else FMX_GN_LAUNCH(true, 4, false);
  } else {
    FMX_GN_LAUNCH(false, 4, false);
  }
#undef FMX_GN_LAUNCH
#undef FMX_GN_LAUNCH_FUSED
  FMX_LAUNCH_CHECK("fmx_groupnorm_apply_f16");
  return FMX_OK;
}


// ---- RMS LayerNorm (T5LayerNorm, backend/nn/t5.py:15-25): y = x * rsqrt(mean(x^2) + eps) * weight -- no mean subtraction, no bias.  One wave per
//      row, the row kept in registers, fp32 sum of squares (the reference takes it in x's own type: fp32 here is the closer-to-exact form).
template <int MAXOCT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const f16* __restrict__ x, const f16* __restrict__ weight, f16* __restrict__ y, long rows, int c, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int oct = c >> 3;
  const f16* xr = x + row * c;
  f16x8 v[MAXOCT];
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXOCT; ++j) {
    const int o = lane + j * 64;
    if (o < oct) {
      v[j] = *reinterpret_cast<const f16x8*>(xr + o * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) q = fmaf((float)v[j][e], (float)v[j][e], q);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
  f16* yr = y + row * c;
#pragma unroll
  for (int j = 0; j < MAXOCT; ++j) {
    const int o = lane + j * 64;
    if (o < oct) {
      const f16x8 g = *reinterpret_cast<const f16x8*>(weight + o * 8);
      f16x8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (f16)((float)v[j][e] * rstd * (float)g[e]);
      *reinterpret_cast<f16x8*>(yr + o * 8) = r;
    }
  }
}

extern "C" int fmx_rmsnorm_f16(const void* x, const void* weight, void* y, int64_t rows, int32_t c, float eps, void* stream) {
  FMX_REQUIRE(x && weight && y && rows > 0 && c > 0 && (c % 8) == 0 && c <= 4096, "rmsnorm: bad args");
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(y) && fmx_aligned16(weight), "rmsnorm: alignment");
  const long blocks = (rows + 3) / 4;
  if (blocks >= (1L << 31)) return fmx_set_error(FMX_E_BADARG, "rmsnorm: too many rows");
  hipStream_t st = (hipStream_t)stream;
  const int oct = c >> 3;
#define FMX_RMS(MO) hipLaunchKernelGGL((rmsnorm_kernel<MO>), dim3((unsigned)blocks), dim3(256), 0, st, (const f16*)x, (const f16*)weight, (f16*)y, (long)rows, c, eps)
  if (oct <= 64) FMX_RMS(1);
  else if (oct <= 128) FMX_RMS(2);
  else if (oct <= 256) FMX_RMS(4);
  else FMX_RMS(8);
#undef FMX_RMS
  FMX_LAUNCH_CHECK("fmx_rmsnorm_f16");
  return FMX_OK;
}

template <bool MOD>
static int launch_ln(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps, long rows_per_b,
                     long ld_mod, void* stream, const char* name) {
  const long blocks = (rows + 3) / 4;
  if (blocks >= (1L << 31)) return fmx_set_error(FMX_E_BADARG, "%s: too many rows", name);
  hipStream_t st = (hipStream_t)stream;
  const int oct = c >> 3;
#define FMX_LN(MO)                                                                                                              \
  hipLaunchKernelGGL((ln_kernel<MO, MOD>), dim3((unsigned)blocks), dim3(256), 0, st, (const f16*)x, (const f16*)gamma, (const f16*)beta, \
                     (f16*)y, (long)rows, c, eps, rows_per_b, ld_mod)
  if (oct <= 64) FMX_LN(1);
  else if (oct <= 128) FMX_LN(2);
  else if (oct <= 192) FMX_LN(3);
  else FMX_LN(8);
#undef FMX_LN
  FMX_LAUNCH_CHECK(name);
  return FMX_OK;
}

extern "C" int fmx_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c,
                                 float eps, void* stream) {
  FMX_REQUIRE(x && gamma && beta && y && rows > 0 && c > 0 && (c % 8) == 0 && c <= 4096, "layernorm: bad args");
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(y) && fmx_aligned16(gamma) && fmx_aligned16(beta), "layernorm: alignment");
  return launch_ln<false>(x, gamma, beta, y, rows, c, eps, rows, rows, stream, "fmx_layernorm_f16");
}

extern "C" int fmx_layernorm_padded_f16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps,
                                        int64_t rows_per_image, int64_t out_rows_per_image, void* stream) {
  FMX_REQUIRE(x && gamma && beta && y && rows > 0 && c > 0 && (c % 8) == 0 && c <= 4096, "layernorm_padded: bad args");
  FMX_REQUIRE(rows_per_image > 0 && out_rows_per_image >= rows_per_image && rows % rows_per_image == 0, "layernorm_padded: bad row geometry");
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(y) && fmx_aligned16(gamma) && fmx_aligned16(beta), "layernorm_padded: alignment");
  return launch_ln<false>(x, gamma, beta, y, rows, c, eps, rows_per_image, out_rows_per_image, stream, "fmx_layernorm_padded_f16");
}

extern "C" int fmx_layernorm_mod_f16(const void* x, const void* scale, const void* shift, int64_t ld_mod, int64_t rows_per_batch,
                                     void* y, int64_t rows, int32_t c, float eps, void* stream) {
  FMX_REQUIRE(x && scale && shift && y && rows > 0 && rows_per_batch > 0 && c > 0 && (c % 8) == 0 && c <= 4096, "layernorm_mod: bad args");
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(y) && fmx_aligned16(scale) && fmx_aligned16(shift) && (ld_mod % 8) == 0, "layernorm_mod: alignment");
  return launch_ln<true>(x, scale, shift, y, rows, c, eps, rows_per_batch, ld_mod, stream, "fmx_layernorm_mod_f16");
}
