// Pieces shared by the software-pipelined GEMM kernels (fmx_gemm256p.hip: one 8-wave workgroup per CU; fmx_gemm4w.hip: two 4-wave workgroups per CU):
// compile-time loops and the row pass of the epilogue.  Included inside each file's anonymous namespace user; everything here is force-inlined.
#pragma once
#include <utility>

#include "fmx_gemm_common.hpp"

namespace {

template <int V>
struct IC { static constexpr int value = V; };

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Row pass of the epilogue: RB-byte fp32 rows staged in this wave's LDS slice -> 8 output columns per lane, LPR lanes per
// row.  v = act(a * alpha + bias + rowvec[image]) * gate + residual, fp16 store.  The pointers are __restrict__ PARAMETERS
// on purpose: the output may be the residual itself (in-place `x += f(x)`), and without the no-alias promise the compiler
// makes every later global load wait (s_waitcnt vmcnt(0)) for every earlier store to COMPLETE -- one full store round trip
// per iteration, 9-15 us of epilogue per tile (tools/clock_gemm.py).  The promise is safe here: an element is read and
// written by the same lane only, and its store depends on its load through registers.
// MODE 0: bias + rowvec + gate + residual;  1: bias + residual;  2: bias only (the absent operands cost a load each otherwise);
// 3: bias + rowvec + residual, no gate (the statistics-emitting kernels: 8 registers fewer than MODE 0)
// STATS: the lane also accumulates sum / sum of squares of the 8 columns it stores (of the ROUNDED fp16 values: what a statistics pass
// over the stored tensor would read) into st[0..7] / st[8..15] -- the GroupNorm statistics of the output, see the kernel's epilogue.
// LN (LayerNorm folded into the GEMMs on both sides of it, backend/nn/unet.py:262-279 `norm2` / `norm3` of a BasicTransformerBlock):
//   LN = 1, the PRODUCER of the tensor the LayerNorm reads (the residual-adding projection): also emits, per output row and per wave column
//        range, {sum, sum of squares} of the fp16 values it stores -> row_stats[m][part][2].  The LPR lanes of a row leave their pairs in the
//        LDS bytes of the staged row they have just consumed (dead from then on); 32 lanes add them up after the pass.
//   LN = 2, the CONSUMER (the projection that follows the LayerNorm), run on the UN-normalised tensor with weights pre-multiplied by gamma:
//        LN(x) W = rstd (x W' - mean colsum(W')) + (beta W + b)  -- `lnm` / `lnr` hold mean / rstd of row `lane` of this wave's rows
//        (computed once per tile from the producer's partials), fetched per row with a lane shuffle; `cs` = colsum of this lane's 8 columns.
//   LN = 3, the consumer in the OPERAND-SWAPPED GEMM (V^T = Wv x^T, 320 x 256 tile): the LayerNorm rows are this GEMM's output COLUMNS, so a lane's
//        8 columns carry 8 (rstd, -mean rstd) pairs for the whole tile (`cs` = ln_col_ab at its first column) and each output row m one
//        (colsum, folded bias) pair (`rowcb`):  v = acc rstd[n] + (-mean rstd)[n] colsum[m] + bias'[m].
template <int RB, int LPR, int ROWS, int SWZ, bool TANH, int MODE, bool STATS = false, int LN = 0>
__device__ __forceinline__ void epi_rows(const char* my, int lane, int mbase, int M, int /*col*/, bool nok, int per_img, float alpha, float has_gate,
                                         const f16* __restrict__ bias, const f16* __restrict__ rowvec, long ld_rv, const f16* __restrict__ gate, long ld_gt,
                                         const f16* __restrict__ res, long ld_res, f16* __restrict__ out, long ld_out, float* st = nullptr,
                                         float* __restrict__ rowst = nullptr, int rowst_ld = 0, float lnm = 0.f, float lnr = 0.f, int lnbase = 0,
                                         const float* __restrict__ cs = nullptr, const float* __restrict__ rowcb = nullptr) {
  constexpr int RPI = 64 / LPR;  // rows per wave instruction
  constexpr int ITERS = (ROWS + RPI - 1) / RPI;
  // opaque copy: keeps the compiler from hoisting the ITERS x 2 LDS offsets of EVERY call of this function above the whole
  // epilogue (they are loop-invariant across the block rows) -- 35 spilled registers in the 160-accumulator tile otherwise
  asm volatile("" : "+v"(lane));
  const int cg = lane % LPR, rsub = lane / LPR;
  const f16x8 bb = *reinterpret_cast<const f16x8*>(bias);
  f32x4 cs0 = f32x4{0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
  if (LN == 2) {
    cs0 = *reinterpret_cast<const f32x4*>(cs);
    cs1 = *reinterpret_cast<const f32x4*>(cs + 4);
  }
  f32x4 ca0 = cs0, ca1 = cs0, cb0 = cs0, cb1 = cs0;   // LN 3: rstd / -mean rstd of this lane's 8 columns
  if (LN == 3) {
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(cs), t1 = *reinterpret_cast<const f32x4*>(cs + 4);
    const f32x4 t2 = *reinterpret_cast<const f32x4*>(cs + 8), t3 = *reinterpret_cast<const f32x4*>(cs + 12);
    ca0 = f32x4{t0[0], t0[2], t1[0], t1[2]};
    cb0 = f32x4{t0[1], t0[3], t1[1], t1[3]};
    ca1 = f32x4{t2[0], t2[2], t3[0], t3[2]};
    cb1 = f32x4{t2[1], t2[3], t3[1], t3[3]};
  }
  // vmcnt retires in order, loads and stores alike: an iteration that loads its operands AFTER the previous iteration's
  // store waits for that store to complete.  So the operands of iteration it+1 are requested before iteration it stores.
  // One iteration of lead is ~200 cycles of work against 900 cycles of HBM latency (the residual was written a whole kernel ago):
  // the pass was a chain of ITERS memory latencies.  PF-1 iterations of lead (ring of PF register slots; the accumulators' fragment
  // double buffer is dead by now, so the registers exist) put PF-1 latencies in flight per wave.
#ifndef FMX_EPI_PREFETCH
#define FMX_EPI_PREFETCH 6
#endif
  constexpr int PF = MODE == 1 ? FMX_EPI_PREFETCH : (MODE == 3 ? (FMX_EPI_PREFETCH > 3 ? 3 : FMX_EPI_PREFETCH) : 2);
  f16x8 rv[PF], gt[PF], rs[PF];
  auto fetch = [&](int it, int slot) {
    const int row = min(it * RPI + rsub, ROWS - 1);
    const int m = mbase + row;
    const int mc = m < M ? m : M - 1;
    const int img = mc / per_img;
    if (MODE == 0 || MODE == 3) rv[slot] = *reinterpret_cast<const f16x8*>(rowvec + img * ld_rv);
    if (MODE == 0) gt[slot] = *reinterpret_cast<const f16x8*>(gate + img * ld_gt);
    if (MODE <= 1 || MODE == 3) rs[slot] = *reinterpret_cast<const f16x8*>(res + mc * ld_res);
  };
#pragma unroll
  for (int it = 0; it < PF - 1 && it < ITERS; ++it) fetch(it, it);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int cur = it % PF;
    if (it + PF - 1 < ITERS) fetch(it + PF - 1, (it + PF - 1) % PF);
    const int row = min(it * RPI + rsub, ROWS - 1);
    const f32x4 lo = *reinterpret_cast<const f32x4*>(my + row * RB + (((2 * cg) ^ (row & SWZ)) << 4));
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(my + row * RB + (((2 * cg + 1) ^ (row & SWZ)) << 4));
    const int m = mbase + row;
    const bool ok = nok && m < M && rsub < RPI && it * RPI + rsub < ROWS;
    float mean_r = 0.f, rstd_r = 1.f;
    if (LN == 2) {
      mean_r = __shfl(lnm, lnbase + row);
      rstd_r = __shfl(lnr, lnbase + row);
    }
    f32x2 rcb = f32x2{0.f, 0.f};
    if (LN == 3) rcb = *reinterpret_cast<const f32x2*>(rowcb + (long)(m < M ? m : M - 1) * 2);
    f16x8 hv;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float v = (r < 4 ? lo[r & 3] : hi4[r & 3]) * alpha;
      if (LN == 2) v = (v - mean_r * (r < 4 ? cs0[r & 3] : cs1[r & 3])) * rstd_r;
      if (LN == 3) v = fmaf(v, r < 4 ? ca0[r & 3] : ca1[r & 3], fmaf(r < 4 ? cb0[r & 3] : cb1[r & 3], rcb[0], rcb[1]));
      v += (float)bb[r];
      if (MODE == 0 || MODE == 3) v += (float)rv[cur][r];
      if (TANH) v = gelu_tanh_f(v);
      if (MODE == 0) v *= fmaf(has_gate, (float)gt[cur][r] - 1.0f, 1.0f);
      if (MODE <= 1 || MODE == 3) v += (float)rs[cur][r];
      hv[r] = (f16)v;
    }
#ifndef FMX_EPI_PLAIN_STORES
    // streaming (nt) stores: the tile is not read again by this kernel and the launch ends with ~42 MB of dirty output to write back
    // (tools/ubench/launch_floor.hip: 160 KB per CU from all 256 CUs, 7.99 us plain, 7.16 us nt; SDXL step 114.2 -> 113.8 ms, profiles/r08i)
    if (ok) {
      typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
      union { f16x8 h; u32x4_ u; } cv;
      cv.h = hv;
      __builtin_nontemporal_store(cv.u, reinterpret_cast<u32x4_*>(out + m * ld_out));
    }
#else   // A/B build (tools/build_variant.sh)
    if (ok) *reinterpret_cast<f16x8*>(out + m * ld_out) = hv;
#endif
    if (LN == 1) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float f = (float)hv[r];
        s1 += f;
        s2 = fmaf(f, f, s2);
      }
      if (!nok) s1 = s2 = 0.f;   // columns beyond nout
      // this row's staged accumulators were consumed by the two reads above (all of its lanes, same instructions): its bytes are free
      if (rsub < RPI && it * RPI + rsub < ROWS) *reinterpret_cast<float2*>(const_cast<char*>(my) + row * RB + cg * 8) = float2{s1, s2};
    }
    if (STATS) {
      // every row of the tile is a valid output row here (the host only asks for statistics when M % 256 == 0), lanes with an
      // out-of-range column or a row-lane >= RPI are dropped by the reduction that follows; only the last iteration can revisit a row
      const bool fresh = (it + 1) * RPI <= ROWS || it * RPI + rsub < ROWS;
      if (fresh) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float f = (float)hv[r];
          st[r] += f;
          st[8 + r] = fmaf(f, f, st[8 + r]);
        }
      }
    }
  }
  if (LN == 1) {
    if (lane < ROWS) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int c = 0; c < LPR; ++c) {
        const float2 v = *reinterpret_cast<const float2*>(my + lane * RB + c * 8);
        a += v.x;
        b += v.y;
      }
      const int m = mbase + lane;
      if (m < M) *reinterpret_cast<float2*>(rowst + (long)m * rowst_ld) = float2{a, b};
    }
  }
}

}  // namespace
