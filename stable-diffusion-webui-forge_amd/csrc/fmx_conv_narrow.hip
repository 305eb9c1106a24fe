// 3x3 convolution with AT MOST 4 OUTPUT CHANNELS for gfx950: the VAE decoder's `conv_out` (128 -> 3 channels at the full image size,
// /root/reference/backend/nn/vae.py:248-271 `self.conv_out`, 8 x 1024^2 pixels per SDXL batch).
//
// Through the implicit-GEMM kernels that layer is a GEMM with N = 3: the narrowest tile computes 97 % padding, and the im2col gather re-fetches every
// input pixel nine times through the L2 -> LDS path (2.2 ms per 8 x 1024^2 decode for 58 GFLOP: 0.03 of the tile's rate).  The layer is a STREAM: 2.1 GB of
// input, 67 MB of output.  So:
//   * a workgroup (4 waves) owns 4 rows x 32 pixels of the output; the input patch it needs -- 6 rows x 34 (staged: 36) pixels x C channels -- is staged ONCE in LDS by
//     LDS-DMA (buffer_load ... lds, 1 KiB = 4 pixels x 256 B per instruction for C = 128; pixels outside the image read as zeros through the
//     descriptor's bounds check), 63 KB at C = 128 (6 x 36 pixels x 256 B of patch + 4 x 9 x 128 fp16 weights), two workgroups per CU;
//   * every tap reads its shifted pixels from that patch: weights (padded to 16 output rows: 3 real + zeros) as the MFMA-A operand, 16 consecutive pixels
//     of a row as the MFMA-B operand of v_mfma_f32_16x16x32; a wave owns one row of 32 pixels = 2 accumulator blocks, 9 taps x C / 32 k-steps;
//   * 16-byte chunk c of patch pixel q lives at chunk c ^ (q & 15): 16 neighbouring pixels of one fragment read hit 16 different bank groups;
//   * the accumulator block hands lanes 0-15 the 4 output channels of one pixel each: bias, one 8-byte store per pixel, 128 contiguous bytes per block.
// Bound: HBM (the input read once, the 1.6x halo from L2).  Wider inputs (the UNet's own `out` convolution: 320 -> 4 channels, backend/nn/unet.py:760-764) are
// walked in channel chunks of C = 128 / 64 / 32 (the widest that divides the channel count): patch and weights of a chunk staged, 9 taps accumulated, next chunk.
#include "fmx_common.hpp"

namespace {

constexpr int TH = 4, TW = 32;            // output tile of a workgroup
constexpr int PH = TH + 2;
// staged pixels per patch row: TW + 2 halo, rounded up to whole DMA pieces (4 pixels of 128 channels, 8 of 64, 16 of 32)
constexpr int patch_width(int c) { return (TW + 2 + 1024 / (c * 2) - 1) / (1024 / (c * 2)) * (1024 / (c * 2)); }

// C: channels per chunk (staged at once); ctot: channels of the input (a multiple of C)
// GN (round 6): the input is the UN-normalised tensor and `ss` its GroupNorm's (image, channel) {scale, shift} table (gn_finalize): the patch is staged
// through registers -- silu(fma(x, scale, shift)) with gn_apply's own arithmetic and rounding, zeros outside the image -- instead of by LDS-DMA, and the
// norm_out -> swish -> conv_out tail of the VAE decoder (backend/nn/vae.py:266-271) loses its GroupNorm apply pass (one read + one write of the full-size tensor)
template <int C, bool GN = false>
__global__ __launch_bounds__(256, 2) void conv3x3_narrow_kernel(const f16* __restrict__ x, long x_bytes, const f16* __restrict__ wgt, const f16* __restrict__ bias,
                                                                f16* __restrict__ out, int n, int h, int w, int ctot, int nout, int ld_out, int tiles_x, int tiles_y,
                                                                const float* __restrict__ ss = nullptr) {
  constexpr int PW = patch_width(C);
  constexpr int PIXB = C * 2;                       // bytes per pixel OF A CHUNK (LDS); a pixel of the input is ctot * 2 bytes
  constexpr int PPP = 1024 / PIXB;                  // pixels per 1-KiB DMA piece: 4 (C = 128), 8 (64), 16 (32)
  constexpr int CHUNKS = PIXB / 16;                 // 16-byte chunks per pixel
  constexpr int PIECES_ROW = PW / PPP;              // pieces per patch row
  static_assert(PW % PPP == 0 && CHUNKS <= 16 && (C % 32) == 0, "piece geometry");
  constexpr int PATCH = PH * PW * PIXB;
  constexpr int KSTEPS = C / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  f16* const wl = reinterpret_cast<f16*>(smem + PATCH);   // [4][9 * C] weights, row 3 = zeros when nout < 4 ... rows >= nout are zero
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kg = lane >> 4;

  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int img = t / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;

  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(x), 0, (unsigned)x_bytes, 0x00020000);
  constexpr unsigned OOB = 0xC0000000u;
  const long img_base = (long)img * h * w;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (int c0 = 0; c0 < ctot; c0 += C) {
    if (c0) __syncthreads();   // every wave is done with the previous chunk's patch and weights
    // ---- weights of this chunk -> LDS (4 rows of 9 x C halfs; rows >= nout zero) ---------------------------------------------------------------------
    for (int i = tid; i < 4 * 9 * C / 8; i += 256) {
      const int row = i / (9 * C / 8), r8 = i - row * (9 * C / 8);
      const int tap = r8 / (C / 8), c8 = r8 - tap * (C / 8);
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (f16)0.0f;
      if (row < nout) v = *reinterpret_cast<const f16x8*>(wgt + ((long)row * 9 + tap) * ctot + c0 + c8 * 8);
      *reinterpret_cast<f16x8*>(wl + ((long)row * 9 + tap) * C + c8 * 8) = v;
    }
    if constexpr (GN) {
      // ---- input patch through registers: item i = (patch pixel i / CHUNKS, logical chunk i % CHUNKS); 256 % CHUNKS == 0, so a thread keeps ONE chunk ----
      constexpr int ITEMS = PH * PW * CHUNKS, ITERS = (ITEMS + 255) / 256;
      const int cc = tid % CHUNKS;
      const float* ssp = ss + ((long)img * ctot + c0 + cc * 8) * 2;
      f32x4 sv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const f32x4*>(ssp + i * 4);   // {scale, shift} of channels 2i, 2i + 1
      f16x8 v[ITERS];
      int lo[ITERS];                                         // -1: no item; else LDS byte offset, bit 30 = outside the image
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int pi = tid / CHUNKS + it * (256 / CHUNKS);
        const int pr = pi / PW, q = pi - pr * PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + q;
        const bool in_patch = pi < PH * PW;
        const bool ok = in_patch && gy >= 0 && gy < h && gx >= 0 && gx < w;
        lo[it] = in_patch ? ((pi * PIXB + ((cc ^ (q & (CHUNKS - 1))) << 4)) | (ok ? 0 : (1 << 30))) : -1;
        if (ok) v[it] = *reinterpret_cast<const f16x8*>(x + (img_base + (long)gy * w + gx) * ctot + c0 + cc * 8);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        if (lo[it] < 0) continue;
        f16x8 r;
        if (lo[it] & (1 << 30)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = (f16)0.0f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const f32x4 sq = sv[e >> 1];
            r[e] = (f16)silu_f(fmaf((float)v[it][e], sq[(e & 1) * 2], sq[(e & 1) * 2 + 1]));   // gn_apply_kernel's arithmetic, bit for bit
          }
        }
        *reinterpret_cast<f16x8*>(patch + (lo[it] & ((1 << 30) - 1))) = r;
      }
    } else
    // ---- input patch -> LDS: piece p = (patch row pr, pixels pc * PPP .. + PPP); lane -> pixel lane / CHUNKS, physical chunk lane % CHUNKS ---------
    for (int p = wave; p < PH * PIECES_ROW; p += 4) {       // uniform trip count per wave
      const int pr = p / PIECES_ROW, pc = p - pr * PIECES_ROW;
      const int q = pc * PPP + lane / CHUNKS;               // patch-local pixel index in its row
      const int gy = y0 - 1 + pr, gx = x0 - 1 + q;
      const int chunk = (lane % CHUNKS) ^ (q & (CHUNKS - 1));   // logical chunk this lane fetches (source side of the swizzle)
      const bool ok = gy >= 0 && gy < h && gx >= 0 && gx < w;
      const unsigned off = ok ? (unsigned)((img_base + (long)gy * w + gx) * ((long)ctot * 2) + c0 * 2 + chunk * 16) : OOB;
      auto* dst = (__attribute__((address_space(3))) void*)(patch + (pr * PW + pc * PPP) * PIXB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- wave `wave` = output row y0 + wave, two blocks of 16 pixels -------------------------------------------------------------------------------
    const f16* wrow = wl + (long)(l16 < 4 ? l16 : 3) * 9 * C;   // output rows >= 4 of the 16-row operand: masked below
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          f16x8 wf = *reinterpret_cast<const f16x8*>(wrow + (ky * 3 + kx) * C + ks * 32 + kg * 8);
          if (l16 >= 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[e] = (f16)0.0f;
          }
#pragma unroll
          for (int bl = 0; bl < 2; ++bl) {
            const int q = bl * 16 + l16 + kx;                  // patch pixel of this lane's output pixel under tap kx
            const int chunk = (ks * 4 + kg) ^ (q & (CHUNKS - 1));
            const f16x8 af = *reinterpret_cast<const f16x8*>(patch + ((wave + ky) * PW + q) * PIXB + chunk * 16);
            acc[bl] = FMX_MFMA_16x16x32(wf, af, acc[bl]);
          }
        }
  }
  // ---- lanes 0-15 (kg = 0) hold output channels 0-3 of pixel l16 of each block ----------------------------------------------------------------------
  const int oy = y0 + wave;
  if (kg == 0 && oy < h) {
    float b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = (bias && r < nout) ? (float)bias[r] : 0.f;
#pragma unroll
    for (int bl = 0; bl < 2; ++bl) {
      const int ox = x0 + bl * 16 + l16;
      if (ox < w) {
        f16* op = out + (img_base + (long)oy * w + ox) * ld_out;
        if (ld_out == 4) {
          f16x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (f16)(r < nout ? acc[bl][r] + b[r] : 0.f);
          *reinterpret_cast<f16x4*>(op) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nout) op[r] = (f16)(acc[bl][r] + b[r]);
        }
      }
    }
  }
}

template <int C, bool GN = false>
int launch_narrow(const void* x, long x_bytes, const void* wgt, const void* bias, void* out, int n, int h, int w, int ctot, int nout, int ld_out, hipStream_t st,
                  const float* ss = nullptr) {
  constexpr int SMEM = PH * patch_width(C) * C * 2 + 4 * 9 * C * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_narrow_kernel<C, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH;
  hipLaunchKernelGGL((conv3x3_narrow_kernel<C, GN>), dim3((unsigned)(n * tiles_x * tiles_y)), dim3(256), SMEM, st, (const f16*)x, x_bytes, (const f16*)wgt, (const f16*)bias,
                     (f16*)out, n, h, w, ctot, nout, ld_out, tiles_x, tiles_y, ss);
  FMX_LAUNCH_CHECK("fmx_conv3x3_narrow_f16");
  return FMX_OK;
}

}  // namespace

int fmx_launch_gn_finalize(const float* partial, int32_t nchunks, int32_t c, int32_t n, int32_t groups, int32_t hw, float eps, const void* gamma,
                           const void* beta, float* scale_shift, hipStream_t st);   // fmx_norm.hip

// silu(group_norm(x)) -> conv3x3 with at most 4 output channels, one launch (+ the statistics' finalize): see fmx.h
extern "C" int fmx_conv3x3_narrow_gn_silu_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const float* x_partial, int32_t x_nchunks, int32_t groups,
                                              float eps, const void* gamma, const void* beta, float* scale_shift, const void* wgt, const void* bias, int32_t nout,
                                              void* out, int32_t ld_out, void* stream) {
  FMX_REQUIRE(x && wgt && out && x_partial && gamma && beta && scale_shift && n > 0 && h > 0 && w > 0 && x_nchunks >= 1, "conv3x3_narrow_gn_silu: bad arguments");
  FMX_REQUIRE(nout >= 1 && nout <= 4 && ld_out >= nout, "conv3x3_narrow_gn_silu: 1..4 output channels, ld_out >= nout (got %d, %d)", nout, ld_out);
  FMX_REQUIRE(c >= 32 && (c % 32) == 0 && c <= 2048 && groups > 0 && (c % groups) == 0, "conv3x3_narrow_gn_silu: input channels must be a multiple of 32 and of the group count (got %d, %d groups)", c, groups);
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(wgt) && fmx_aligned16(scale_shift) && (ld_out != 4 || (reinterpret_cast<uintptr_t>(out) & 7u) == 0), "conv3x3_narrow_gn_silu: operands must be 16-byte aligned");
  FMX_REQUIRE((long)n * ((w + TW - 1) / TW) * ((h + TH - 1) / TH) < (1L << 31), "conv3x3_narrow_gn_silu: too many tiles for one launch (split the batch)");
  hipStream_t st = (hipStream_t)stream;
  const int rc = fmx_launch_gn_finalize(x_partial, x_nchunks, c, n, groups, h * w, eps, gamma, beta, scale_shift, st);
  if (rc != FMX_OK) return rc;
  const long bytes = (long)n * h * w * c * 2;
  if ((c % 128) == 0) return launch_narrow<128, true>(x, bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st, scale_shift);
  if ((c % 64) == 0) return launch_narrow<64, true>(x, bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st, scale_shift);
  return launch_narrow<32, true>(x, bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st, scale_shift);
}

extern "C" int fmx_conv3x3_narrow_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt, const void* bias, int32_t nout, void* out,
                                      int32_t ld_out, void* stream) {
  FMX_REQUIRE(x && wgt && out && n > 0 && h > 0 && w > 0, "conv3x3_narrow: bad arguments");
  FMX_REQUIRE(nout >= 1 && nout <= 4 && ld_out >= nout, "conv3x3_narrow: 1..4 output channels, ld_out >= nout (got %d, %d)", nout, ld_out);
  FMX_REQUIRE(c >= 32 && (c % 32) == 0 && c <= 2048, "conv3x3_narrow: input channels must be a multiple of 32, 32..2048 (got %d)", c);
  FMX_REQUIRE(fmx_aligned16(x) && fmx_aligned16(wgt) && (ld_out != 4 || (reinterpret_cast<uintptr_t>(out) & 7u) == 0), "conv3x3_narrow: operands must be 16-byte aligned");
  const double bytes = (double)n * h * w * c * 2.0;
  FMX_REQUIRE(bytes < 3.0e9 && (long)n * ((w + TW - 1) / TW) * ((h + TH - 1) / TH) < (1L << 31), "conv3x3_narrow: input beyond the 32-bit offset range of one launch (split the batch)");
  hipStream_t st = (hipStream_t)stream;
  if ((c % 128) == 0) return launch_narrow<128>(x, (long)bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st);
  if ((c % 64) == 0) return launch_narrow<64>(x, (long)bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st);
  return launch_narrow<32>(x, (long)bytes, wgt, bias, out, n, h, w, c, nout, ld_out, st);
}
