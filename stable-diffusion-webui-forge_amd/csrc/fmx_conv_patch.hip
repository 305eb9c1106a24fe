// GroupNorm + SiLU + 3x3 convolution in ONE kernel for the high-resolution levels of the VAE decoder (round 6): the `norm -> swish -> conv` halves of
// /root/reference/backend/nn/vae.py:98-114 (ResnetBlock.forward: norm1 / conv1 and norm2 / conv2, the second with `x + h`) at 128 output channels and
// 8 x 1024^2 pixels per SDXL batch.
//
// Until round 5 that half was two launches: gn_apply (one read + one write of the tensor, 2.1 GB each way at 128 channels: 0.8 ms) and the implicit-GEMM
// convolution, whose 512 x 128 tile fetches every input pixel NINE times through the L2 -> LDS DMA path -- (512 + 128) / 8 one-KiB pieces per 64-deep
// K-tile at ~13 ns per piece and CU against 0.86 us of MFMA work: ingest-bound, 890 TFLOP/s where the wider layers reach 1 130-1 250.  Here:
//   * a workgroup (4 waves, two workgroups per CU) owns 8 rows x 32 pixels of the output and all 128 output channels.  The input patch it needs
//     -- 10 x 34 pixels -- is staged ONCE per 64-channel chunk: 16-byte global loads into registers, y = silu(x * scale[c] + shift[c]) with the
//     (image, channel) scale / shift table gn_finalize made from the producer's statistics, rounded to the element type exactly as gn_apply rounds
//     it (the same fmaf + silu_f, so the staged values ARE the tensor the unfused path stores), written to LDS as 128-byte pixels with 16-byte
//     chunk c of patch column q at chunk c ^ ((q >> 1) & 7).  Pixels outside the image are staged as ZEROS: the convolution pads the NORMALISED tensor;
//   * the nine taps read their shifted pixels from that patch (MFMA-B operand of v_mfma_f32_16x16x32: 16 neighbouring pixels of one row); the
//     weights of one tap and chunk -- 128 x 64 = 16 KB, the MFMA-A operand -- stream through a two-slot ring by LDS-DMA, one barrier per tap;
//     a wave owns 2 rows = 64 pixels x 128 channels (4 x 8 accumulator blocks, 12 fragment reads per 32 MFMAs);
//   * so a tile moves 43.5 KB of activations once per chunk instead of 9 x 32 KB, and its DMA pieces are the 16 of the weights: a fifth of the
//     implicit-GEMM form's ingest, and the GroupNorm apply pass does not exist;
//   * epilogue in the accumulator layout (a lane holds 4 consecutive output channels of a pixel): + bias (+ residual), ONE rounding, 8-byte stores; the
//     GroupNorm statistics of the OUTPUT -- per channel sum and sum of squares of the values as stored -- are reduced over the tile (16-lane
//     butterflies, then the four waves through LDS in wave order: deterministic) and written as this tile's record of the [image][tile][channel][2]
//     array gn_finalize reads (fmx_norm.hip), like the GEMM epilogue's 256-row records.
#include "fmx_common.hpp"

int fmx_launch_gn_finalize(const float* partial, int32_t nchunks, int32_t c, int32_t n, int32_t groups, int32_t hw, float eps, const void* gamma,
                           const void* beta, float* scale_shift, hipStream_t st);   // fmx_norm.hip

namespace {

constexpr int TH = 8, TW = 32;                 // output tile of a workgroup
constexpr int PH = TH + 2, PW = TW + 2;        // staged patch (1-pixel halo)
constexpr int CC = 64;                         // channels per staged chunk: 128-byte pixels in LDS
constexpr int NOUT = 128;                      // output channels of a workgroup
constexpr int PATCH_BYTES = PH * PW * CC * 2;  // 43 520
constexpr int WBUF_BYTES = NOUT * CC * 2;      // 16 384: one tap of one chunk
constexpr int SMEM = PATCH_BYTES + 2 * WBUF_BYTES;
constexpr int ITEMS = PH * PW * (CC / 8);      // 16-byte items of a patch chunk
constexpr int ITERS = (ITEMS + 255) / 256;

struct PatchConvParams {
  const f16* x;          // [n][h][w][cin]
  const float* ss;       // [n][cin][2] = {scale, shift} of the GroupNorm in front (gn_finalize)
  const f16* wgt;        // [NOUT][9][cin]
  unsigned w_bytes;
  const f16* bias;       // [NOUT] or null
  const f16* res;        // [n*h*w][ld_res] or null
  long ld_res;
  f16* out;              // [n*h*w][ld_out]
  long ld_out;
  float* stats;          // [n][tiles][NOUT][2] or null
  int n, h, w, cin, tiles_x, tiles_y, nwg;
};

__global__ __launch_bounds__(256, 2) void conv3x3_gn_patch_kernel(const PatchConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch = smem;
  char* const wbuf = smem + PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kg = lane >> 4;
  int t = xcd_remap(blockIdx.x, p.nwg);
  const int tile_in_img = t % (p.tiles_x * p.tiles_y);
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int img = t / p.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int cin = p.cin, h = p.h, w = p.w;
  const int ntaps = (cin / CC) * 9;

  // ---- weights: tap `ti` (chunk ti / 9, tap ti % 9) -> ring slot, 16 one-KiB pieces of 8 rows x 128 bytes, four per wave -----------------------
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.wgt), 0, p.w_bytes, 0x00020000);
  unsigned w_voff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = (wave * 4 + e) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);            // the chunk this lane fetches (source side of the swizzle)
    w_voff[e] = ((unsigned)row * 9u * (unsigned)cin + (unsigned)logical * 8u) * 2u;
  }
  auto stage_w = [&](int slot, int ti) __attribute__((always_inline)) {
    const int chunk = ti / 9, tap = ti - chunk * 9;
    const unsigned soff = ((unsigned)tap * (unsigned)cin + (unsigned)chunk * CC) * 2u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      auto* dst = (__attribute__((address_space(3))) void*)(wbuf + slot * WBUF_BYTES + (wave * 4 + e) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, w_voff[e], soff, 0, 0);
    }
  };

  // ---- input patch of one chunk: global -> registers -> silu(x * scale + shift) -> LDS -------------------------------------------------------------
  const int cc = tid & 7;                                          // this thread's 16-byte channel octet of the chunk, in every item
  auto stage_patch = [&](int c0) __attribute__((always_inline)) {
    const float* ssp = p.ss + ((long)img * cin + c0 + cc * 8) * 2;
    f32x4 sv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sv[i] = *reinterpret_cast<const f32x4*>(ssp + i * 4);   // {scale, shift} pairs of channels 2i, 2i + 1
    f16x8 v[ITERS];
    int lds_off[ITERS];                                            // -1: no item; >= 0: LDS byte offset; bit 30: outside the image (zeros)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int pi = (tid >> 3) + it * 32;                         // patch pixel
      const int pr = pi / PW, q = pi - pr * PW;
      const int gy = y0 - 1 + pr, gx = x0 - 1 + q;
      const bool in_patch = pi < PH * PW;
      const bool ok = in_patch && gy >= 0 && gy < h && gx >= 0 && gx < w;
      lds_off[it] = in_patch ? ((pi * (CC * 2) + ((cc ^ ((q >> 1) & 7)) << 4)) | (ok ? 0 : (1 << 30))) : -1;
      if (ok) v[it] = *reinterpret_cast<const f16x8*>(p.x + (((long)img * h + gy) * w + gx) * cin + c0 + cc * 8);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      if (lds_off[it] < 0) continue;
      f16x8 r;
      if (lds_off[it] & (1 << 30)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (f16)0.0f;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f32x4 s = sv[e >> 1];
#ifdef FMX_CGN_ABLATE_SILU   // timing build: what the normalisation arithmetic costs (wrong results)
          r[e] = v[it][e];
#else
          r[e] = (f16)silu_f(fmaf((float)v[it][e], s[(e & 1) * 2], s[(e & 1) * 2 + 1]));   // gn_apply_kernel's arithmetic, bit for bit
#endif
        }
      }
      *reinterpret_cast<f16x8*>(patch + (lds_off[it] & ((1 << 30) - 1))) = r;
    }
  };

  f32x4 acc[4][8];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[pb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane parts of the fragment addresses.  Weights: row cb * 16 + l16 -> swizzle key (l16 >> 1) & 7 (cb * 16 does not change it)
  const int w_lane = l16 * 128 + ((kg ^ ((l16 >> 1) & 7)) << 4);

  stage_w(0, 0);
  stage_patch(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  int ky = 0, kx = 0;
  for (int ti = 0; ti < ntaps; ++ti) {
    if (ti + 1 < ntaps) stage_w((ti + 1) & 1, ti + 1);
    // byte offsets into smem; every term but the chunk bits (4-6) is a multiple of 128, so k-step 1 (logical chunks 4-7) is `^ 64`
    const int wo = PATCH_BYTES + (ti & 1) * WBUF_BYTES + w_lane;
    // activations: pixel column q = (pb & 1) * 16 + l16 + kx of patch row 2 * wave + (pb >> 1) + ky; the key of (16 + q') equals the key of q'
    const int q0 = l16 + kx;
    const int po = ((2 * wave + ky) * PW + q0) * (CC * 2) + ((kg ^ ((q0 >> 1) & 7)) << 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 wf[8], af[4];
      // (the block / pixel-block offsets are multiples of 128 and leave bit 6 alone: one base per k-step, every read an immediate offset)
      const char* wks = smem + (wo ^ (ks * 64));
      const char* pks = smem + (po ^ (ks * 64));
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) wf[cb] = *reinterpret_cast<const f16x8*>(wks + cb * 2048);
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) af[pb] = *reinterpret_cast<const f16x8*>(pks + ((pb >> 1) * PW + (pb & 1) * 16) * (CC * 2));
#pragma unroll
      for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) acc[pb][cb] = FMX_MFMA_16x16x32(wf[cb], af[pb], acc[pb][cb]);
    }
    if (++kx == 3) {
      kx = 0;
      if (++ky == 3) ky = 0;
    }
#ifdef FMX_CGN_ABLATE_RESTAGE   // timing build: the first chunk's patch serves every chunk (wrong results): what restaging costs
    if (false) {
#else
    if (ti + 1 < ntaps && ky == 0 && kx == 0) {       // next tap opens a new channel chunk: every wave is done with this patch, then restage
#endif
      __syncthreads();
      stage_patch(((ti + 1) / 9) * CC);
    }
#ifdef FMX_CGN_ABLATE_WAIT    // timing build: the barrier without waiting for the next tap's weights (wrong results): what the DMA round trip costs
    __builtin_amdgcn_s_barrier();
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#endif
  }
#ifdef FMX_CGN_ABLATE_WAIT
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#endif

  // ---- epilogue: lane (l16, kg) of block (pb, cb) holds output channels cb * 16 + kg * 4 + 0..3 of pixel (row 2 * wave + (pb >> 1), column (pb & 1) * 16 + l16) ----
  f32x4 bsum[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    bsum[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
      const f16x4 b4 = *reinterpret_cast<const f16x4*>(p.bias + cb * 16 + kg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) bsum[cb][r] = (float)b4[r];
    }
  }
  f32x4 ssum[8], sqs[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) ssum[cb] = sqs[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    const int oy = y0 + 2 * wave + (pb >> 1), ox = x0 + (pb & 1) * 16 + l16;
    const bool ok = oy < h && ox < w;
    const long m = ((long)img * h + oy) * w + ox;
    f16x4 rv[8];
    if (p.res && ok) {
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) rv[cb] = *reinterpret_cast<const f16x4*>(p.res + m * p.ld_res + cb * 16 + kg * 4);
    }
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      f16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float f = acc[pb][cb][r] + bsum[cb][r];
        if (p.res && ok) f += (float)rv[cb][r];
        o[r] = (f16)f;
        const float g = ok ? (float)o[r] : 0.f;      // statistics of the values AS STORED
        ssum[cb][r] += g;
        sqs[cb][r] += g * g;
      }
#ifdef FMX_CGN_ABLATE_STORE   // timing build: no output traffic (wrong results)
      if (ok && o[0] == (f16)12345.0f) *reinterpret_cast<f16x4*>(p.out + m * p.ld_out + cb * 16 + kg * 4) = o;
#else
      if (ok) *reinterpret_cast<f16x4*>(p.out + m * p.ld_out + cb * 16 + kg * 4) = o;
#endif
    }
  }
  if (p.stats) {
    // all waves are past the last barrier of the tap loop: the weight ring is free -> red[wave][NOUT][2]
    float* red = reinterpret_cast<float*>(wbuf);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = ssum[cb][r], q1 = sqs[cb][r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          s1 += __shfl_xor(s1, d);
          q1 += __shfl_xor(q1, d);
        }
        if (l16 == 0) {
          const int c = cb * 16 + kg * 4 + r;
          red[(wave * NOUT + c) * 2 + 0] = s1;
          red[(wave * NOUT + c) * 2 + 1] = q1;
        }
      }
    __syncthreads();
    if (tid < NOUT) {
      float s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s1 += red[(wv * NOUT + tid) * 2 + 0];
        q1 += red[(wv * NOUT + tid) * 2 + 1];
      }
      float* dst = p.stats + (((long)img * (p.tiles_x * p.tiles_y) + tile_in_img) * NOUT + tid) * 2;
      *reinterpret_cast<f32x2*>(dst) = f32x2{s1, q1};
    }
  }
}

}  // namespace

extern "C" int fmx_conv3x3_gn_silu_f16(const fmx_conv_gn_args* a, int32_t* stats_nchunks, void* stream) {
  FMX_REQUIRE(a && a->x && a->x_partial && a->gamma && a->beta && a->scale_shift && a->wgt && a->out, "conv3x3_gn_silu: null pointer");
  FMX_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->x_nchunks >= 1, "conv3x3_gn_silu: bad geometry");
  FMX_REQUIRE(a->cout == NOUT, "conv3x3_gn_silu: 128 output channels (got %d)", a->cout);
  FMX_REQUIRE(a->cin >= CC && (a->cin % CC) == 0 && a->cin <= 1024, "conv3x3_gn_silu: input channels must be a multiple of 64, 64..1024 (got %d)", a->cin);
  FMX_REQUIRE(a->groups > 0 && (a->cin % a->groups) == 0, "conv3x3_gn_silu: channels not divisible into %d groups", a->groups);
  FMX_REQUIRE(a->ld_out >= NOUT && (a->ld_out % 4) == 0 && (!a->residual || (a->ld_res >= NOUT && (a->ld_res % 4) == 0)), "conv3x3_gn_silu: bad leading dimensions");
  FMX_REQUIRE(fmx_aligned16(a->x) && fmx_aligned16(a->wgt) && fmx_aligned16(a->scale_shift) && (reinterpret_cast<uintptr_t>(a->out) & 7u) == 0 &&
                  (!a->residual || (reinterpret_cast<uintptr_t>(a->residual) & 7u) == 0) && (!a->bias || (reinterpret_cast<uintptr_t>(a->bias) & 7u) == 0),
              "conv3x3_gn_silu: operand alignment");
  const int tiles_x = (a->w + TW - 1) / TW, tiles_y = (a->h + TH - 1) / TH;
  const long nwg = (long)a->n * tiles_x * tiles_y;
  FMX_REQUIRE(nwg < (1L << 31), "conv3x3_gn_silu: too many tiles for one launch (split the batch)");
  FMX_REQUIRE(!a->stats || a->stats_cap >= tiles_x * tiles_y, "conv3x3_gn_silu: the statistics buffer holds %d records per image, %d tiles write one each",
              a->stats_cap, tiles_x * tiles_y);
  hipStream_t st = (hipStream_t)stream;
  const int rc = fmx_launch_gn_finalize(a->x_partial, a->x_nchunks, a->cin, a->n, a->groups, a->h * a->w, a->eps, a->gamma, a->beta, a->scale_shift, st);
  if (rc != FMX_OK) return rc;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_gn_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr = true;
  }
  PatchConvParams p;
  p.x = (const f16*)a->x;
  p.ss = a->scale_shift;
  p.wgt = (const f16*)a->wgt;
  p.w_bytes = (unsigned)((long)NOUT * 9 * a->cin * 2);
  p.bias = (const f16*)a->bias;
  p.res = (const f16*)a->residual;
  p.ld_res = a->ld_res;
  p.out = (f16*)a->out;
  p.ld_out = a->ld_out;
  p.stats = a->stats;
  p.n = a->n; p.h = a->h; p.w = a->w; p.cin = a->cin; p.tiles_x = tiles_x; p.tiles_y = tiles_y; p.nwg = (int)nwg;
  hipLaunchKernelGGL(conv3x3_gn_patch_kernel, dim3((unsigned)nwg), dim3(256), SMEM, st, p);
  FMX_LAUNCH_CHECK("fmx_conv3x3_gn_silu_f16");
  if (stats_nchunks) *stats_nchunks = a->stats ? tiles_x * tiles_y : 0;
  return FMX_OK;
}
