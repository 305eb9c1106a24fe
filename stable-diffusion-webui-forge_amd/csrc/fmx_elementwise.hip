// Small HBM-bound kernels around the UNet / VAE executors and the sampler loop (gfx950).
#include "fmx_common.hpp"

namespace {

constexpr int TPB = 256;
inline unsigned grid_for(long n, int per_thread = 1) {
  long b = (n + (long)TPB * per_thread - 1) / ((long)TPB * per_thread);
  if (b > 65535L * 16) b = 65535L * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// timestep_embedding (backend/nn/unet.py:55-67): cat([cos, sin]) of t*exp(-ln(P)*k/half), fp32 math
__global__ void timestep_embedding_kernel(const float* __restrict__ t, f16* __restrict__ emb, int b, int dim, float log_period) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * half) return;
  const int bi = i / half, k = i - bi * half;
  const float freq = expf(-log_period * (float)k / (float)half);
  const float a = t[bi] * freq;
  emb[(long)bi * dim + k] = (f16)cosf(a);
  emb[(long)bi * dim + half + k] = (f16)sinf(a);
}

__global__ void silu_kernel(const f16* __restrict__ x, f16* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (f16)silu_f((float)x[i]);
}

// h NHWC fp16 += ctrl NCHW fp32.  One block handles a 64-pixel x 64-channel patch through LDS so both sides are coalesced.
__global__ void add_control_nchw_kernel(f16* __restrict__ h, const float* __restrict__ ctrl, int c, long npix) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows of 64
  for (int r = ty; r < 64; r += 4) {                        // r = channel within the patch, tx = pixel
    const int ch = c0 + r;
    const long p = p0 + tx;
    tile[r][tx] = (ch < c && p < npix) ? ctrl[((long)b * c + ch) * npix + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {                        // r = pixel within the patch, tx = channel
    const long p = p0 + r;
    const int ch = c0 + tx;
    if (p < npix && ch < c) {
      f16* q = h + ((long)b * npix + p) * c + ch;
      *q = (f16)((float)*q + tile[tx][r]);
    }
  }
}

__global__ void act_kernel(const f16* __restrict__ x, f16* __restrict__ y, long n, int kind) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = (float)x[i];
    y[i] = (f16)(kind == 0 ? v / (1.0f + __expf(-1.702f * v)) : kind == 1 ? gelu_erf_f(v) : fmaxf(v, 0.f));
  }
}

// 2x2 average pooling, stride 2, fp16 NHWC [n][h][w][c] -> [n][h/2][w/2][c] (h, w even): T2I-Adapter's conv-less Downsample (t2i_adapter.py:42-62)
__global__ void avgpool2x2_kernel(const f16* __restrict__ x, f16* __restrict__ y, int n, int h, int w, int c) {
  const int oh = h / 2, ow = w / 2;
  const long total = (long)n * oh * ow * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const int ox = (int)((i / c) % ow);
    const int oy = (int)((i / ((long)c * ow)) % oh);
    const long b = i / ((long)c * ow * oh);
    const f16* p = x + ((b * h + 2 * oy) * w + 2 * ox) * (long)c + ch;
    const float v = (float)p[0] + (float)p[c] + (float)p[(long)w * c] + (float)p[(long)w * c + c];
    y[i] = (f16)(0.25f * v);
  }
}

__global__ void embed_tokens_kernel(const int* __restrict__ ids, const f16* __restrict__ tok, const f16* __restrict__ pos, f16* __restrict__ out,
                                    int tokens, int c, int vocab) {
  const int row = blockIdx.x;                      // b * tokens + t
  const int t = row % tokens;
  const int id = min(max(ids[row], 0), vocab - 1);
  for (int j = threadIdx.x * 8; j < c; j += blockDim.x * 8) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(tok + (long)id * c + j);
    const f16x8 b = *reinterpret_cast<const f16x8*>(pos + (long)t * c + j);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)a[e] + (float)b[e]);
    *reinterpret_cast<f16x8*>(out + (long)row * c + j) = o;
  }
}

__global__ void cast_f32_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (f16)x[i];
}

// x fp32 NCHW [b][c][h][w] -> fp16 3x3-im2col rows [reps*b*h*w][64] of x / sqrt(sigma^2 + sd^2)
// one thread per (row, tap); each writes c halfs.  Columns >= 9*c are zeroed by tap-0 threads.
__global__ void unet_pack_input_kernel(const float* __restrict__ x, const float* __restrict__ sigma, float sd2, int b, int c,
                                       int h, int w, int reps, f16* __restrict__ out) {
  const long total = (long)reps * b * h * w * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long row = i / 9;
    const int px = (int)(row % w);
    const int py = (int)((row / w) % h);
    const int bi = (int)((row / ((long)w * h)) % b);
    const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
    const float s = sigma[bi];
    const float inv = rsqrtf(s * s + sd2);
    f16* o = out + row * 64 + tap * c;
    for (int ch = 0; ch < c; ++ch) o[ch] = ok ? (f16)(x[(((long)bi * c + ch) * h + iy) * w + ix] * inv) : (f16)0.f;
    if (tap == 0)
      for (int k = 9 * c; k < 64; ++k) out[row * 64 + k] = (f16)0.f;
  }
}

// fp16 NHWC [n][h][w][ldx] (first c channels) -> 3x3 im2col rows [n*h*w][64]
__global__ void im2col3x3_smallc_kernel(const f16* __restrict__ x, int ldx, int n, int c, int h, int w, f16* __restrict__ out) {
  const long total = (long)n * h * w * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const long row = i / 9;
    const int px = (int)(row % w);
    const int py = (int)((row / w) % h);
    const long bi = row / ((long)w * h);
    const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < h && ix >= 0 && ix < w;
    f16* o = out + row * 64 + tap * c;
    const f16* src = x + ((bi * h + iy) * w + ix) * ldx;
    for (int ch = 0; ch < c; ++ch) o[ch] = ok ? src[ch] : (f16)0.f;
    if (tap == 0)
      for (int k = 9 * c; k < 64; ++k) out[row * 64 + k] = (f16)0.f;
  }
}

// calculate_denoised per half (k_prediction.py:81-92: denoised = A(sigma) * x + B(sigma) * model_output with
//   epsilon / const: A = 1, B = -sigma;   v_prediction: A = sd^2 / (sigma^2 + sd^2), B = -sigma sd / sqrt(sigma^2 + sd^2);   edm: same A, +B)
// then the CFG combine (sampling_function.py:276-288,312)
__global__ void cfg_combine_kernel(const f16* __restrict__ eps, int ld, const float* __restrict__ x, const float* __restrict__ sigma,
                                   int b, int c, int h, int w, int reps, float cond_scale, float* __restrict__ den,
                                   float* __restrict__ cond_pred, float* __restrict__ uncond_pred, int pred_type, float sigma_data) {
  const long total = (long)b * c * h * w;
  const long hw = (long)h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i % hw;
    const int ch = (int)((i / hw) % c);
    const int bi = (int)(i / (hw * c));
    const float s = sigma[bi];
    float ca = 1.f, cb = -s;
    if (pred_type != 0) {
      const float v = s * s + sigma_data * sigma_data;
      ca = sigma_data * sigma_data / v;
      cb = (pred_type == 1 ? -1.f : 1.f) * s * sigma_data / sqrtf(v);
    }
    const float xv = x[i];
    float result;
    if (reps == 2) {
      const float eu = (float)eps[((long)bi * hw + pix) * ld + ch];
      const float ec = (float)eps[((long)(b + bi) * hw + pix) * ld + ch];
      // accumulators 0 + out*1 divided by counts 1e-37 + 1 (sampling_function.py:155-159,284-288): exact in fp32
      const float du = pred_type == 0 ? xv - eu * s : xv * ca + eu * cb;
      const float dc = pred_type == 0 ? xv - ec * s : xv * ca + ec * cb;
      result = du + (dc - du) * cond_scale;
      if (cond_pred) cond_pred[i] = dc;
      if (uncond_pred) uncond_pred[i] = du;
    } else {
      const float ec = (float)eps[((long)bi * hw + pix) * ld + ch];
      const float dc = pred_type == 0 ? xv - ec * s : xv * ca + ec * cb;
      result = 0.f + (dc - 0.f) * cond_scale;  // uncond half skipped when cond_scale == 1 (:295-298)
      if (cond_pred) cond_pred[i] = dc;
      if (uncond_pred) uncond_pred[i] = 0.f;
    }
    den[i] = result;
  }
}

__global__ void euler_step_kernel(const float* __restrict__ x, const float* __restrict__ den, float sigma, float sigma_next,
                                  const float* __restrict__ noise, float noise_scale, float* __restrict__ out, long n) {
  const float dt = sigma_next - sigma;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float xv = x[i];
    const float d = (xv - den[i]) / sigma;  // to_d (modules/sd_schedulers.py:10-12)
    float r = xv + d * dt;
    if (noise) r = r + noise[i] * noise_scale;
    out[i] = r;
  }
}

__global__ void lincomb3_kernel(const float* __restrict__ x, const float* __restrict__ d0, const float* __restrict__ d1, float a,
                                float bc, float cc, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float r;
    if (d1) {
      // sample_dpmpp_2m (k_diffusion/sampling.py:666-669): denoised_d first, then the update
      const float dd = bc * d0[i] + cc * d1[i];
      r = a * x[i] + dd;
    } else {
      r = a * x[i] + bc * d0[i];
    }
    out[i] = r;
  }
}

struct LincombArgs {
  const float* src[8];
  float coef[8];
};

// out = sum_k coef[k] * src[k], accumulated left to right (host-side scalar coefficients of the multistep / multi-stage samplers)
template <int N>
__global__ void lincomb_kernel(LincombArgs a, float* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= n) {
      float4 r = *reinterpret_cast<const float4*>(a.src[0] + i);
      r.x *= a.coef[0], r.y *= a.coef[0], r.z *= a.coef[0], r.w *= a.coef[0];
#pragma unroll
      for (int k = 1; k < N; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(a.src[k] + i);
        r.x += a.coef[k] * v.x, r.y += a.coef[k] * v.y, r.z += a.coef[k] * v.z, r.w += a.coef[k] * v.w;
      }
      *reinterpret_cast<float4*>(out + i) = r;
    } else {
      for (long j = i; j < n; ++j) {
        float r = a.coef[0] * a.src[0][j];
#pragma unroll
        for (int k = 1; k < N; ++k) r += a.coef[k] * a.src[k][j];
        out[j] = r;
      }
    }
  }
}

// DPM-Solver adaptive step-size control (k_diffusion/sampling.py:531-532): sum over the latent of ((x_low - x_high) / delta)^2 with
// delta = max(atol, rtol * max(|x_low|, |x_prev|)).  Stage 1: one fp32 partial per block (fixed grid, fixed summation order inside the
// block: strided per-thread sums, wave shuffle tree, LDS across the 4 waves); stage 2: a single block adds the partials in order.
// Same bits on every run for a given size.
constexpr int ERRNORM_BLOCKS = 256;

__global__ void error_norm_partial_kernel(const float* __restrict__ x_low, const float* __restrict__ x_high, const float* __restrict__ x_prev,
                                          float atol, float rtol, float* __restrict__ partial, long n) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float lo = x_low[i];
    const float delta = fmaxf(atol, rtol * fmaxf(fabsf(lo), fabsf(x_prev[i])));
    const float r = (lo - x_high[i]) / delta;
    acc += r * r;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float wave_sum[TPB / 64];
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < TPB / 64; ++w) t += wave_sum[w];
    partial[blockIdx.x] = t;
  }
}

__global__ void error_norm_final_kernel(const float* __restrict__ partial, int nblocks, float inv_numel, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < nblocks; ++b) t += partial[b];
    out[0] = sqrtf(t) * sqrtf(inv_numel);  // ||.||_2 / sqrt(numel)
  }
}

// out[b][c][oy][ox] = sum_ky sum_kx yw[oy][ky] * xw[ox][kx] * in[b][c][ys[oy] + ky][xs[ox] + kx]  (hires-fix latent resize: every
// interpolate mode of modules/shared.py:56-63 is separable with <= a handful of taps per axis when upscaling)
__global__ void resize_separable_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ ys,
                                        const float* __restrict__ yw, const int* __restrict__ xs, const float* __restrict__ xw, int planes,
                                        int h, int w, int oh, int ow, int ky, int kx) {
  const long total = (long)planes * oh * ow;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const long p = i / ((long)ow * oh);
    const float* src = in + p * (long)h * w + (long)ys[oy] * w + xs[ox];
    const float* wy = yw + (long)oy * ky;
    const float* wx = xw + (long)ox * kx;
    float acc = 0.f;
    for (int a = 0; a < ky; ++a) {
      float row = 0.f;
      for (int b = 0; b < kx; ++b) row += wx[b] * src[(long)a * w + b];
      acc += wy[a] * row;
    }
    out[i] = acc;
  }
}

// h += alpha * c, same (NHWC) layout: ControlNet residuals that already live channels-last (the native ControlNet's outputs)
template <typename T>
__global__ void add_scaled_kernel(f16* __restrict__ h, const T* __restrict__ c, float alpha, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    if (i + 8 <= n) {
      f16 hv[8];
      *reinterpret_cast<uint4*>(hv) = *reinterpret_cast<const uint4*>(h + i);
      T cv[8];
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(cv) = *reinterpret_cast<const uint4*>(c + i);
      } else {
        *reinterpret_cast<uint4*>(cv) = *reinterpret_cast<const uint4*>(c + i);
        *reinterpret_cast<uint4*>(cv + 4) = *reinterpret_cast<const uint4*>(c + i + 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) hv[k] = (f16)((float)hv[k] + alpha * (float)cv[k]);
      *reinterpret_cast<uint4*>(h + i) = *reinterpret_cast<const uint4*>(hv);
    } else {
      for (long j = i; j < n; ++j) h[j] = (f16)((float)h[j] + alpha * (float)c[j]);
    }
  }
}

__global__ void scale_kernel(const float* __restrict__ x, float s, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

// how many of n fp16 values are inf / NaN (exponent bits all ones): the VAE's overflow guard (backend/nn/vae.py auto_bf16_fallback)
__global__ void count_nonfinite_kernel(const uint16_t* __restrict__ x, long n, int* __restrict__ count) {
  int bad = 0;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    if (i + 8 <= n) {
      const uint4 v = *reinterpret_cast<const uint4*>(x + i);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) bad += ((w[k] & 0x7C00u) == 0x7C00u) + ((w[k] & 0x7C000000u) == 0x7C000000u);
    } else {
      for (long j = i; j < n; ++j) bad += (x[j] & 0x7C00u) == 0x7C00u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

// the three VAE boundary kernels exist for both 16-bit element types (T = _Float16, or __bf16 for the bfloat16 decoder / encoder)
template <typename T>
__global__ void vae_pack_latent_kernel(const float* __restrict__ z, float inv_scale, float shift, int b, int c, int h, int w,
                                       T* __restrict__ out, int ld) {
  const long total = (long)b * h * w * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % ld);
    const long pix = i / ld;
    const long hw = (long)h * w;
    const long bi = pix / hw, p = pix % hw;
    out[i] = ch < c ? (T)(z[(bi * c + ch) * hw + p] / inv_scale + shift) : (T)0.f;
  }
}

template <typename T>
__global__ void vae_unpack_image_kernel(const T* __restrict__ y, int ld, long npix, int c, float* __restrict__ out) {
  const long total = npix * c;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / c;
    const int ch = (int)(i - pix * c);
    const float v = ((float)y[pix * ld + ch] + 1.0f) / 2.0f;
    out[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

__global__ void blend_masked_kernel(const float* a, const float* am, const float* b, const float* bm, float* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] * am[i] + b[i] * bm[i];
}

template <typename T>
__global__ void vae_sample_posterior_kernel(const T* __restrict__ mo, int ld, const float* __restrict__ noise, int b, int lc, long npix, float scale,
                                            float shift, float* __restrict__ out) {
  const long total = (long)b * lc * npix;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i % npix;
    const long bc = i / npix;
    const int c = (int)(bc % lc);
    const long bi = bc / lc;
    const T* row = mo + (bi * npix + p) * ld;
    const float mean = (float)row[c];
    const float logvar = fminf(fmaxf((float)row[lc + c], -30.0f), 20.0f);
    out[i] = (mean + __expf(0.5f * logvar) * noise[i] - shift) * scale;
  }
}

// Philox4x32-10 (modules/rng_philox.py:32-64) + Box-Muller sine branch (:67-76)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)c[0] * 0xD2511F53ull;
  const uint64_t p1 = (uint64_t)c[2] * 0xCD9E8D57ull;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void philox_randn_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t offset, float* __restrict__ out,
                                    uint32_t* __restrict__ raw, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {offset, 0u, (uint32_t)i, 0u};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      if (r != 9) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    }
    if (raw) { raw[i * 4 + 0] = c[0]; raw[i * 4 + 1] = c[1]; raw[i * 4 + 2] = c[2]; raw[i * 4 + 3] = c[3]; }
    const float inv = 2.3283064e-10f;
    const float inv2pi = 2.3283064e-10f * 6.2831855f;
    const float u = (float)c[0] * inv + inv / 2.0f;
    const float v = (float)c[1] * inv2pi + inv2pi / 2.0f;
    out[i] = sqrtf(-2.0f * logf(u)) * sinf(v);
  }
}

// ---- strided 4-D copy with element conversion: the layout adapter of the attention- and op-level entry points --------------------------
// (NCHW <-> NHWC, [B, N, H*d] -> zero-padded heads, V -> V^T, bool mask -> additive mask).  One thread per element, innermost
// destination index fastest; not a hot-path kernel.
struct Copy4Params {
  const void* src;
  void* dst;
  long ss[4], ds[4];
  int d[4];
  int src_kind, dst_kind;
};
__device__ __forceinline__ float load_kind(const void* p, long i, int kind) {
  switch (kind) {
    case 0: return (float)reinterpret_cast<const _Float16*>(p)[i];
    case 1: return reinterpret_cast<const float*>(p)[i];
    case 2: return (float)reinterpret_cast<const __bf16*>(p)[i];
    default: return reinterpret_cast<const unsigned char*>(p)[i] ? 0.0f : -INFINITY;   // bool "attend" mask -> additive mask
  }
}
__global__ void strided_copy4_kernel(const Copy4Params p) {
  const long total = (long)p.d[0] * p.d[1] * p.d[2] * p.d[3];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int i3 = (int)(r % p.d[3]); r /= p.d[3];
    const int i2 = (int)(r % p.d[2]); r /= p.d[2];
    const int i1 = (int)(r % p.d[1]);
    const int i0 = (int)(r / p.d[1]);
    const float v = load_kind(p.src, i0 * p.ss[0] + i1 * p.ss[1] + i2 * p.ss[2] + i3 * p.ss[3], p.src_kind);
    const long o = i0 * p.ds[0] + i1 * p.ds[1] + i2 * p.ds[2] + i3 * p.ds[3];
    if (p.dst_kind == 0) reinterpret_cast<_Float16*>(p.dst)[o] = (_Float16)v;
    else if (p.dst_kind == 1) reinterpret_cast<float*>(p.dst)[o] = v;
    else reinterpret_cast<__bf16*>(p.dst)[o] = (__bf16)v;
  }
}

}  // namespace

extern "C" int fmx_timestep_embedding(const float* t, void* emb, int32_t b, int32_t dim, float max_period, void* stream) {
  FMX_REQUIRE(t && emb && b > 0 && dim > 0 && (dim % 2) == 0, "timestep_embedding: bad args");
  const int total = b * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, t, (f16*)emb, b, dim,
                     logf(max_period));
  FMX_LAUNCH_CHECK("fmx_timestep_embedding");
  return FMX_OK;
}

extern "C" int fmx_silu_f16(const void* x, void* y, int64_t n, void* stream) {
  FMX_REQUIRE(x && y && n > 0, "silu: bad args");
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, (long)n);
  FMX_LAUNCH_CHECK("fmx_silu_f16");
  return FMX_OK;
}

extern "C" int fmx_add_control_nchw(void* h, const float* ctrl, int32_t b, int32_t c, int64_t npix, void* stream) {
  FMX_REQUIRE(h && ctrl && b > 0 && c > 0 && npix > 0, "add_control: bad args");
  hipLaunchKernelGGL(add_control_nchw_kernel, dim3((unsigned)((npix + 63) / 64), (unsigned)((c + 63) / 64), (unsigned)b), dim3(256), 0,
                     (hipStream_t)stream, (f16*)h, ctrl, c, (long)npix);
  FMX_LAUNCH_CHECK("fmx_add_control_nchw");
  return FMX_OK;
}

extern "C" int fmx_act_f16(const void* x, void* y, int64_t n, int32_t kind, void* stream) {
  FMX_REQUIRE(x && y && n > 0 && kind >= 0 && kind <= 2, "act: bad args");
  hipLaunchKernelGGL(act_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, (long)n, kind);
  FMX_LAUNCH_CHECK("fmx_act_f16");
  return FMX_OK;
}

extern "C" int fmx_embed_tokens(const int32_t* ids, const void* tok_emb, const void* pos_emb, void* out, int32_t batch, int32_t tokens, int32_t c,
                                int32_t vocab, void* stream) {
  FMX_REQUIRE(ids && tok_emb && pos_emb && out && batch > 0 && tokens > 0 && c > 0 && (c % 8) == 0 && vocab > 0, "embed_tokens: bad args");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3(batch * tokens), dim3(128), 0, (hipStream_t)stream, ids, (const f16*)tok_emb, (const f16*)pos_emb,
                     (f16*)out, tokens, c, vocab);
  FMX_LAUNCH_CHECK("fmx_embed_tokens");
  return FMX_OK;
}

extern "C" int fmx_strided_copy4(const void* src, int32_t src_kind, const int64_t* src_strides, void* dst, int32_t dst_kind,
                                 const int64_t* dst_strides, const int32_t* dims, void* stream) {
  FMX_REQUIRE(src && dst && src_strides && dst_strides && dims, "strided_copy4: null pointer");
  FMX_REQUIRE(src_kind >= 0 && src_kind <= 3 && dst_kind >= 0 && dst_kind <= 2, "strided_copy4: element kinds are 0 f16, 1 f32, 2 bf16 (source also 3 = bool mask)");
  Copy4Params p;
  p.src = src; p.dst = dst; p.src_kind = src_kind; p.dst_kind = dst_kind;
  long total = 1;
  for (int i = 0; i < 4; ++i) {
    FMX_REQUIRE(dims[i] > 0, "strided_copy4: dims must be positive");
    p.d[i] = dims[i]; p.ss[i] = src_strides[i]; p.ds[i] = dst_strides[i];
    total *= dims[i];
  }
  hipLaunchKernelGGL(strided_copy4_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, p);
  FMX_LAUNCH_CHECK("fmx_strided_copy4");
  return FMX_OK;
}

extern "C" int fmx_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream) {
  FMX_REQUIRE(x && y && n > 0, "cast: bad args");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n, 4)), dim3(TPB), 0, (hipStream_t)stream, x, (f16*)y, (long)n);
  FMX_LAUNCH_CHECK("fmx_cast_f32_to_f16");
  return FMX_OK;
}

extern "C" int fmx_unet_pack_input(const float* x, const float* sigma, float sigma_data, int32_t b, int32_t c, int32_t h, int32_t w,
                                   int32_t reps, void* out, void* stream) {
  FMX_REQUIRE(x && sigma && out && b > 0 && c > 0 && c * 9 <= 64 && h > 0 && w > 0 && (reps == 1 || reps == 2), "unet_pack_input: bad args");
  const long total = (long)reps * b * h * w * 9;
  hipLaunchKernelGGL(unet_pack_input_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, x, sigma, sigma_data * sigma_data,
                     b, c, h, w, reps, (f16*)out);
  FMX_LAUNCH_CHECK("fmx_unet_pack_input");
  return FMX_OK;
}

extern "C" int fmx_im2col3x3_smallc(const void* x, int32_t ldx, int32_t n, int32_t c, int32_t h, int32_t w, void* out, void* stream) {
  FMX_REQUIRE(x && out && n > 0 && c > 0 && c * 9 <= 64 && ldx >= c && h > 0 && w > 0, "im2col3x3_smallc: bad args");
  const long total = (long)n * h * w * 9;
  hipLaunchKernelGGL(im2col3x3_smallc_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const f16*)x, ldx, n, c, h, w,
                     (f16*)out);
  FMX_LAUNCH_CHECK("fmx_im2col3x3_smallc");
  return FMX_OK;
}

extern "C" int fmx_cfg_combine(const void* eps, int32_t ld_eps, const float* x, const float* sigma, int32_t b, int32_t c, int32_t h,
                               int32_t w, int32_t reps, float cond_scale, float* denoised, float* cond_pred, float* uncond_pred,
                               int32_t prediction_type, float sigma_data, void* stream) {
  FMX_REQUIRE(eps && x && sigma && denoised && b > 0 && c > 0 && ld_eps >= c && (reps == 1 || reps == 2) && prediction_type >= 0 &&
                  prediction_type <= 2 && sigma_data > 0.f,
              "cfg_combine: bad args");
  const long total = (long)b * c * h * w;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const f16*)eps, ld_eps, x, sigma, b, c,
                     h, w, reps, cond_scale, denoised, cond_pred, uncond_pred, prediction_type, sigma_data);
  FMX_LAUNCH_CHECK("fmx_cfg_combine");
  return FMX_OK;
}

extern "C" int fmx_sampler_euler_step(const float* x, const float* denoised, float sigma, float sigma_next, const float* noise,
                                      float noise_scale, float* x_out, int64_t n, void* stream) {
  FMX_REQUIRE(x && denoised && x_out && n > 0 && sigma != 0.f, "euler_step: bad args");
  hipLaunchKernelGGL(euler_step_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, denoised, sigma, sigma_next, noise,
                     noise_scale, x_out, (long)n);
  FMX_LAUNCH_CHECK("fmx_sampler_euler_step");
  return FMX_OK;
}

extern "C" int fmx_sampler_error_norm(const float* x_low, const float* x_high, const float* x_prev, float atol, float rtol, float* workspace,
                                      float* out, int64_t n, void* stream) {
  FMX_REQUIRE(x_low && x_high && x_prev && workspace && out && n > 0 && atol > 0.f, "error_norm: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(error_norm_partial_kernel, dim3(ERRNORM_BLOCKS), dim3(TPB), 0, st, x_low, x_high, x_prev, atol, rtol, workspace, (long)n);
  FMX_LAUNCH_CHECK("fmx_sampler_error_norm (partial)");
  hipLaunchKernelGGL(error_norm_final_kernel, dim3(1), dim3(64), 0, st, workspace, ERRNORM_BLOCKS, 1.0f / (float)n, out);
  FMX_LAUNCH_CHECK("fmx_sampler_error_norm (final)");
  return FMX_OK;
}

extern "C" int fmx_resize_separable_f32(const float* in, float* out, const int32_t* ystart, const float* yweights, const int32_t* xstart,
                                        const float* xweights, int32_t planes, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t ky, int32_t kx,
                                        void* stream) {
  FMX_REQUIRE(in && out && ystart && yweights && xstart && xweights && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0 && ky > 0 && kx > 0 &&
                  ky <= h && kx <= w,
              "resize_separable: bad args");
  const long total = (long)planes * oh * ow;
  hipLaunchKernelGGL(resize_separable_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, in, out, ystart, yweights, xstart,
                     xweights, planes, h, w, oh, ow, ky, kx);
  FMX_LAUNCH_CHECK("fmx_resize_separable_f32");
  return FMX_OK;
}

extern "C" int fmx_avgpool2x2_nhwc_f16(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  FMX_REQUIRE(x && y && n > 0 && c > 0 && h > 0 && w > 0 && (h % 2) == 0 && (w % 2) == 0, "avgpool2x2: even H / W only");
  const long total = (long)n * (h / 2) * (w / 2) * c;
  hipLaunchKernelGGL(avgpool2x2_kernel, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, n, h, w, c);
  FMX_LAUNCH_CHECK("fmx_avgpool2x2_nhwc_f16");
  return FMX_OK;
}

extern "C" int fmx_add_scaled_f16(void* h, const void* c, int32_t c_is_f32, float alpha, int64_t n, void* stream) {
  FMX_REQUIRE(h && c && n > 0 && (reinterpret_cast<uintptr_t>(h) & 15) == 0 && (reinterpret_cast<uintptr_t>(c) & 15) == 0, "add_scaled: bad args");
  const long groups = (n + 7) / 8;
  if (c_is_f32)
    hipLaunchKernelGGL(add_scaled_kernel<float>, dim3(grid_for(groups)), dim3(TPB), 0, (hipStream_t)stream, (f16*)h, (const float*)c, alpha, (long)n);
  else
    hipLaunchKernelGGL(add_scaled_kernel<f16>, dim3(grid_for(groups)), dim3(TPB), 0, (hipStream_t)stream, (f16*)h, (const f16*)c, alpha, (long)n);
  FMX_LAUNCH_CHECK("fmx_add_scaled_f16");
  return FMX_OK;
}

template <int N>
static void launch_lincomb(const LincombArgs& a, float* out, long n, hipStream_t st) {
  hipLaunchKernelGGL(lincomb_kernel<N>, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, st, a, out, n);
}

extern "C" int fmx_sampler_lincomb(const float* const* srcs, const float* coefs, int32_t n_terms, float* x_out, int64_t n, void* stream) {
  FMX_REQUIRE(srcs && coefs && x_out && n > 0 && n_terms >= 1 && n_terms <= 8, "lincomb: 1..8 terms");
  LincombArgs a;
  for (int k = 0; k < 8; ++k) {
    a.src[k] = k < n_terms ? srcs[k] : nullptr;
    a.coef[k] = k < n_terms ? coefs[k] : 0.f;
    FMX_REQUIRE(k >= n_terms || (srcs[k] && (reinterpret_cast<uintptr_t>(srcs[k]) & 15) == 0), "lincomb: null / unaligned source");
  }
  FMX_REQUIRE((reinterpret_cast<uintptr_t>(x_out) & 15) == 0, "lincomb: unaligned output");
  hipStream_t st = (hipStream_t)stream;
  switch (n_terms) {
    case 1: launch_lincomb<1>(a, x_out, (long)n, st); break;
    case 2: launch_lincomb<2>(a, x_out, (long)n, st); break;
    case 3: launch_lincomb<3>(a, x_out, (long)n, st); break;
    case 4: launch_lincomb<4>(a, x_out, (long)n, st); break;
    case 5: launch_lincomb<5>(a, x_out, (long)n, st); break;
    case 6: launch_lincomb<6>(a, x_out, (long)n, st); break;
    case 7: launch_lincomb<7>(a, x_out, (long)n, st); break;
    default: launch_lincomb<8>(a, x_out, (long)n, st); break;
  }
  FMX_LAUNCH_CHECK("fmx_sampler_lincomb");
  return FMX_OK;
}

extern "C" int fmx_sampler_lincomb3(const float* x, const float* denoised, const float* old_denoised, float a, float bcoef, float ccoef,
                                    float* x_out, int64_t n, void* stream) {
  FMX_REQUIRE(x && denoised && x_out && n > 0, "lincomb3: bad args");
  hipLaunchKernelGGL(lincomb3_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, denoised, old_denoised, a, bcoef, ccoef,
                     x_out, (long)n);
  FMX_LAUNCH_CHECK("fmx_sampler_lincomb3");
  return FMX_OK;
}

extern "C" int fmx_scale_f32(const float* x, float s, float* y, int64_t n, void* stream) {
  FMX_REQUIRE(x && y && n > 0, "scale: bad args");
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, s, y, (long)n);
  FMX_LAUNCH_CHECK("fmx_scale_f32");
  return FMX_OK;
}

template <typename T>
static int vae_pack_latent_impl(const float* z, float scaling_factor, float shift, int32_t b, int32_t c, int32_t h, int32_t w, void* out, int32_t ld,
                                void* stream) {
  FMX_REQUIRE(z && out && b > 0 && c > 0 && ld >= c && scaling_factor != 0.f, "vae_pack_latent: bad args");
  const long total = (long)b * h * w * ld;
  hipLaunchKernelGGL(vae_pack_latent_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, (hipStream_t)stream, z, scaling_factor, shift, b, c, h, w,
                     (T*)out, ld);
  FMX_LAUNCH_CHECK("fmx_vae_pack_latent");
  return FMX_OK;
}
extern "C" int fmx_vae_pack_latent(const float* z, float scaling_factor, float shift, int32_t b, int32_t c, int32_t h, int32_t w, void* out,
                                   int32_t ld, void* stream) {
  return vae_pack_latent_impl<f16>(z, scaling_factor, shift, b, c, h, w, out, ld, stream);
}
extern "C" int fmx_vae_pack_latent_bf16(const float* z, float scaling_factor, float shift, int32_t b, int32_t c, int32_t h, int32_t w, void* out,
                                        int32_t ld, void* stream) {
  return vae_pack_latent_impl<__bf16>(z, scaling_factor, shift, b, c, h, w, out, ld, stream);
}

template <typename T>
static int vae_unpack_image_impl(const void* y, int32_t ld, int64_t npix, int32_t c, float* out, void* stream) {
  FMX_REQUIRE(y && out && npix > 0 && c > 0 && ld >= c, "vae_unpack_image: bad args");
  hipLaunchKernelGGL(vae_unpack_image_kernel<T>, dim3(grid_for(npix * c, 2)), dim3(TPB), 0, (hipStream_t)stream, (const T*)y, ld, (long)npix,
                     c, out);
  FMX_LAUNCH_CHECK("fmx_vae_unpack_image");
  return FMX_OK;
}
extern "C" int fmx_vae_unpack_image(const void* y, int32_t ld, int64_t npix, int32_t c, float* out, void* stream) {
  return vae_unpack_image_impl<f16>(y, ld, npix, c, out, stream);
}
extern "C" int fmx_vae_unpack_image_bf16(const void* y, int32_t ld, int64_t npix, int32_t c, float* out, void* stream) {
  return vae_unpack_image_impl<__bf16>(y, ld, npix, c, out, stream);
}

extern "C" int fmx_count_nonfinite_f16(const void* x, int64_t n, int32_t* count, void* stream) {
  FMX_REQUIRE(x && count && n > 0 && fmx_aligned16(x), "count_nonfinite: bad args (x must be 16-byte aligned)");
  hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return fmx_set_error((int)e, "fmx_count_nonfinite_f16: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(grid_for((n + 7) / 8)), dim3(TPB), 0, (hipStream_t)stream, (const uint16_t*)x, (long)n, count);
  FMX_LAUNCH_CHECK("fmx_count_nonfinite_f16");
  return FMX_OK;
}

extern "C" int fmx_blend_masked(const float* a, const float* a_mask, const float* b, const float* b_mask, float* out, int64_t n, void* stream) {
  FMX_REQUIRE(a && a_mask && b && b_mask && out && n > 0, "blend_masked: bad args");
  hipLaunchKernelGGL(blend_masked_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, a, a_mask, b, b_mask, out, (long)n);
  FMX_LAUNCH_CHECK("fmx_blend_masked");
  return FMX_OK;
}

template <typename T>
static int vae_sample_posterior_impl(const void* moments, int32_t ld, const float* noise, int32_t b, int32_t lc, int64_t npix, float scale, float shift,
                                     float* out, void* stream) {
  FMX_REQUIRE(moments && noise && out && b > 0 && lc > 0 && npix > 0 && ld >= 2 * lc, "vae_sample_posterior: bad args");
  hipLaunchKernelGGL(vae_sample_posterior_kernel<T>, dim3(grid_for((long)b * lc * npix)), dim3(TPB), 0, (hipStream_t)stream, (const T*)moments, ld,
                     noise, b, lc, (long)npix, scale, shift, out);
  FMX_LAUNCH_CHECK("fmx_vae_sample_posterior");
  return FMX_OK;
}
extern "C" int fmx_vae_sample_posterior(const void* moments, int32_t ld, const float* noise, int32_t b, int32_t lc, int64_t npix, float scale, float shift,
                                        float* out, void* stream) {
  return vae_sample_posterior_impl<f16>(moments, ld, noise, b, lc, npix, scale, shift, out, stream);
}
extern "C" int fmx_vae_sample_posterior_bf16(const void* moments, int32_t ld, const float* noise, int32_t b, int32_t lc, int64_t npix, float scale,
                                             float shift, float* out, void* stream) {
  return vae_sample_posterior_impl<__bf16>(moments, ld, noise, b, lc, npix, scale, shift, out, stream);
}

extern "C" int fmx_philox_randn(uint64_t seed, uint32_t offset, float* out, uint32_t* raw_u32, int64_t n, void* stream) {
  FMX_REQUIRE(out && n > 0 && n <= 0xFFFFFFFFLL, "philox_randn: bad args");
  hipLaunchKernelGGL(philox_randn_kernel, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, (uint32_t)(seed & 0xFFFFFFFFu),
                     (uint32_t)(seed >> 32), offset, out, raw_u32, (long)n);
  FMX_LAUNCH_CHECK("fmx_philox_randn");
  return FMX_OK;
}
