// Implicit-GEMM convolution / linear for gfx950 (MI355X):  OUT[M, N] = epilogue(A (*) W^T), fp16 in, fp32 acc.
//
// Replaces F.linear / F.conv2d call sites of the reference (backend/operations.py:153-176) together with
// the ops the reference runs around them as separate kernels: nearest Upsample (backend/nn/unet.py:340-355),
// torch.cat of the skip tensor (:741), bias, ResBlock emb broadcast add (:469-477), residual add (:478, :240),
// GEGLU (:104-111).
//
// Design (CDNA4-first, not a translation of anything):
//  * one workgroup = 4 waves (2x2) computes a BM x BN tile with v_mfma_f32_16x16x32_f16; operands are issued
//    swapped (W fragment as the MFMA "A", activation fragment as "B") so every lane ends up with 4
//    CONSECUTIVE output columns of one output row -> 8-byte epilogue loads/stores instead of 2-byte ones.
//  * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, and -- because the
//    per-lane SOURCE address is free -- the im2col gather (3x3 taps, stride 2, zero padding via a zero
//    page, nearest-upsample, channel-concat of two tensors) costs address arithmetic only.
//  * LDS rows are 64 halfs (128 B); the 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the source
//    side and on the ds_read_b128 side (same involution), which makes the fragment reads conflict-free.
//  * double-buffered K loop (BK = 64): issue tile t+1, compute tile t, one vmcnt(0)+barrier per tile.
//  * 1-D grid remapped so that consecutive tiles (same A rows, neighbouring W rows) share an XCD's L2.
//  * split-K for problems with fewer tiles than the chip has workgroup slots (UNet batch 2 = interactive batch 1: M = 2048 rows is 160 tiles
//    of 128 x 128 for 256 CUs): S workgroups share a tile, each over a contiguous range of K-tiles; each leaves its fp32 accumulators in
//    its own slot of a workspace and takes a ticket; the last to arrive adds the slots IN SLOT ORDER (its own from registers), so the sum
//    does not depend on who was last -- deterministic, no floating-point atomics -- and runs the ordinary fused epilogue.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fmx_gemm_common.hpp"

namespace {


constexpr int BK = FMX_BK;

// byte offset of (row, logical 16B chunk) inside a [rows][64] fp16 LDS tile
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// BUF: LDS-DMA through buffer descriptors (32-bit byte offsets, hardware zero fill for out-of-range lanes) -- needs every
// operand to span < 0xC0000000 bytes (dispatcher: fits32); BUF = false keeps 64-bit global addresses + the zero page.
// NST: stages of the LDS ring.  2 (rounds 1-4): issue K-tile t+1, compute K-tile t, vmcnt(0) + barrier -- ONE K-tile in flight, so with one or two
// workgroups on a CU a K-tile costs a full trip to L2 / HBM (~0.8 us measured, profiles/r05c: (2048,1280,5120) - (2048,1280,1280) = 47 us for 60
// K-tiles) against ~0.25 us of MFMA work: the small-M regime (UNet batch 2..8 at the inner levels, where a launch has fewer tiles than the chip has
// CUs and nothing else covers the latency) ran at 250-550 TFLOP/s.  NST >= 3 (round 5): a ring with NST - 1 K-tiles in flight, counted
// s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads() would drain the ring: an LDS-DMA is a pending LDS write on the VM counter), one
// barrier per K-tile:  wait for tile i's own pieces -> barrier (every wave's pieces of tile i are in LDS; every wave is done reading tile i-1) ->
// issue tile i+NST-1 into the stage tile i-1 has left -> compute tile i.
template <int BM, int BN, bool CONV, bool BUF, bool SPLITK = false, int NST = 2>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MI = BM / 32;  // 16-row fragments per wave along M
  constexpr int NI = BN / 32;
  constexpr int LA = BM / 32;  // LDS-DMA instructions per thread for the A tile
  constexpr int LB = BN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int nwg = p.tiles_m * p.tiles_n;
  int wg, ks = 0;
  if (SPLITK) {  // the splits of a tile are neighbours in the remapped order: same XCD, resident together
    const int lin = xcd_remap(blockIdx.x, nwg * p.splits);
    wg = lin / p.splits;
    ks = lin - wg * p.splits;
  } else {
    wg = xcd_remap(blockIdx.x, nwg);
  }
  const int tm = wg / p.tiles_n;
  const int tn = wg - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread load geometry: thread owns LDS chunk (row = j*32 + tid/8, phys chunk = tid&7) -------------
  const int prow = tid >> 3;
  const int kc = (tid & 7) ^ ((tid >> 4) & 7);  // logical chunk this thread must fetch (swizzle, see lds_off)
  const int Ctot = p.c0 + p.c1;

  // A rows
  long a_pix[LA];   // CONV: (n*H + 0)*W base handled below; plain: pixel index m
  int a_iy0[LA], a_ix0[LA];
  bool a_ok[LA];
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int m = m0 + j * 32 + prow;
    a_ok[j] = m < p.M;
    if (CONV) {
      const int per = p.oh * p.ow;
      const int img = m / per;
      const int rem = m - img * per;
      const int oy = rem / p.ow;
      const int ox = rem - oy * p.ow;
      a_pix[j] = (long)img * p.h * p.w;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad_x;
    } else {
      a_pix[j] = m;
      a_iy0[j] = a_ix0[j] = 0;
    }
  }
  // W rows
  const f16* b_ptr[LB];
  unsigned b_voff[LB];
  constexpr unsigned OOB = 0xC0000000u;
  const unsigned kcb = (unsigned)kc * 16u;
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int nn = n0 + j * 32 + prow;
    b_ptr[j] = (nn < p.nout) ? p.wgt + (long)nn * p.ldw + kc * 8 : nullptr;
    b_voff[j] = (nn < p.nout) ? (unsigned)nn * (unsigned)p.ldw * 2u + kcb : OOB;
  }
  const f16* zp = p.zp + kc * 8;
  const __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a0), 0, p.a0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.a1 ? p.a1 : p.a0), 0, p.a1_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.wgt), 0, p.w_bytes, 0x00020000);

  auto stage = [&](int s, int t) {
    char* sa = smem + s * STAGE_BYTES;
    char* sb = sa + BM * 128;
    // which tap / channel slice does K tile t cover?
    int tap = 0, cc = t * BK;
    if (CONV) {
      tap = t / p.cpt;
      cc = (t - tap * p.cpt) * BK;
    }
    const bool second = cc >= p.c0;  // uniform
    const f16* src = second ? p.a1 : p.a0;
    const int sstride = second ? p.s1 : p.s0;
    const int coff = second ? cc - p.c0 : cc;
    int ky = 0, kx = 0;
    if (CONV) { ky = tap / p.kh; kx = tap - ky * p.kh; }
    const __amdgpu_buffer_rsrc_t rs_a = second ? rs_a1 : rs_a0;  // uniform select
    const __amdgpu_buffer_rsrc_t rs_w = rs_b;  // local copy: passing the captured descriptor straight to the builtin makes the host pass drop the kernel stub
#pragma unroll
    for (int j = 0; j < LA; ++j) {
      bool ok = a_ok[j];
      long pix = a_pix[j];
      if (CONV) {
        int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        if (p.up_h > 0) {
          ok = ok && iy >= 0 && iy < p.up_h && ix >= 0 && ix < p.up_w;
          if (p.up_h == 2 * p.h && p.up_w == 2 * p.w) {  // the x2 case of every SD/SDXL/VAE Upsample
            iy >>= 1;
            ix >>= 1;
          } else {  // nearest to an arbitrary skip size (odd latent sizes): src = floor(dst*in/out)
            iy = ok ? (iy * p.h) / p.up_h : 0;
            ix = ok ? (ix * p.w) / p.up_w : 0;
          }
        } else {
          ok = ok && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        }
        pix += (long)iy * p.w + ix;
      }
      char* dst = sa + (j * 256 + wave * 64) * 16;
      if (BUF) {
        const unsigned voff = ok ? (unsigned)pix * (unsigned)sstride * 2u + kcb : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)dst, 16, voff, (unsigned)coff * 2u, 0, 0);
      } else {
        glds16(ok ? src + pix * sstride + coff + kc * 8 : zp, dst);
      }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      char* dst = sb + (j * 256 + wave * 64) * 16;
      if (BUF) {
        const unsigned bv = b_voff[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)dst, 16, bv, (unsigned)t * (BK * 2u), 0, 0);
      } else {
        glds16(b_ptr[j] ? b_ptr[j] + (long)t * BK : zp, dst);
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = SPLITK ? (int)((long)p.kt * ks / p.splits) : 0;
  const int t_end = SPLITK ? (int)((long)p.kt * (ks + 1) / p.splits) : p.kt;
  const int frow = lane & 15;
  const int fk = lane >> 4;
  auto compute = [&](int cur) {
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * (BM / 2) + i * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bf[j] = *reinterpret_cast<const f16x8*>(sb + lds_off(wn * (BN / 2) + j * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = FMX_MFMA_16x16x32(bf[j], af[i], acc[i][j]);
    }
  };
  if constexpr (NST > 2) {
    static_assert(BUF, "the ring runs on the buffer-descriptor path");
    constexpr int PPT = LA + LB;   // LDS-DMA instructions a wave issues per K-tile: what one tile in flight adds to its VM counter
    static_assert((NST - 2) * PPT <= 63 && NST <= 6, "vmcnt is a 6-bit counter; the wait ladder below covers four tiles in flight behind the current one");
    const int nt = t_end - t_begin;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nt) stage(s, t_begin + s);
    int cur = 0, nxt = NST - 1;    // stage of tile i; stage tile i + NST - 1 goes to (= the one tile i - 1 has left)
    for (int i = 0; i < nt; ++i) {
      const int ahead = min(NST - 2, nt - 1 - i);   // tiles issued behind tile i that may still be in flight (uniform)
      if (ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((4 * PPT) & 63) : "memory");
      else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((3 * PPT) & 63) : "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (i + NST - 1 < nt) stage(nxt, t_begin + i + NST - 1);
      compute(cur);
      cur = cur + 1 == NST ? 0 : cur + 1;
      nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    __builtin_amdgcn_s_barrier();   // (split-K: every wave is past its last LDS read before the first word carries the ticket)
  } else {
  stage(0, t_begin);
  wait_vmcnt0();
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    if (t + 1 < t_end) stage(cur ^ 1, t + 1);
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * (BM / 2) + i * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bf[j] = *reinterpret_cast<const f16x8*>(sb + lds_off(wn * (BN / 2) + j * 16 + frow, kk * 4 + fk));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = FMX_MFMA_16x16x32(bf[j], af[i], acc[i][j]);
    }
    wait_vmcnt0();
    __syncthreads();
  }
  }

  if (SPLITK) {
    // ---- split-K hand-over.  The workgroups of a tile may sit on different XCDs, whose L2s are not coherent with each other.  A
    //      device-scope FENCE writes back / invalidates the whole L2 (measured: 80 -> 330 us per launch).  Instead every access that
    //      carries the partials is itself device-scope (the `sc1` bit, what a relaxed agent-scope atomic load / store compiles to on
    //      gfx942 / gfx950): such stores are performed at the memory side before vmcnt retires them, such loads do not hit a stale line.
    //      Order: own stores, s_waitcnt vmcnt(0), barrier, ticket (agent-scope atomic); the last arrival loads the other slots. ------------
    float* const slots = p.ws + (long)wg * p.splits * (BM * BN);
    float* const mine = slots + (long)ks * (BM * BN);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        float* dst = mine + ((i * NI + j) * 256 + tid) * 4;
        // (s_nop: a VMEM store of more than 64 bits must not be followed at once by a VALU write of its data registers -- the compiler
        //  keeps that hazard distance for its own stores, but does not look inside inline asm and re-fills v[4:7] from the AGPRs right away)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[i][j]) : "memory");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (also: every wave is past its last LDS read, the first word can carry the ticket)
    int* const s_ticket = reinterpret_cast<int*>(smem);
    if (tid == 0) *s_ticket = __hip_atomic_fetch_add(p.tickets + wg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = *s_ticket;
    __syncthreads();   // every wave holds the ticket in a register before anything (the statistics epilogue's `red[0]`) may write that LDS word again
    if (ticket != p.splits - 1) return;
    if (tid == 0) __hip_atomic_store(p.tickets + wg, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
    // sum in SLOT order, own slot from registers: the result does not depend on which workgroup arrived last.  All loads of a slot are
    // in flight together (one wait per slot, not per load).
    f32x4 total[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) total[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};   // 0 + x is exact
    for (int s2 = 0; s2 < p.splits; ++s2) {
      if (s2 == ks) {   // workgroup-uniform
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) total[i][j] += acc[i][j];
      } else {
        const float* src = slots + (long)s2 * (BM * BN) + tid * 4;
        f32x4 part[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            // relaxed agent-scope atomic loads = `global_load_dwordx2 ... sc1`, tracked by the compiler's own waitcnt insertion (an
            // inline-asm load's result may be copied before a separate asm waitcnt: the first version of this code read garbage that way)
            const unsigned long long* q = reinterpret_cast<const unsigned long long*>(src + (i * NI + j) * 1024);
            const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi2 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            part[i][j] = f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi2),
                               __uint_as_float((unsigned)(hi2 >> 32))};
          }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) total[i][j] += part[i][j];
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = total[i][j];
  }

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4)*4 + r], r = 0..3 ---------------------
  if (FastEpilogue::eligible(p)) {  // uniform; straight-line loads (see FastEpilogue)
    const FastEpilogue fe(p);
    const bool gg = p.act == FMX_ACT_GEGLU;
    int nbs[NI];
    bool nok[NI];
    f16x4 bb[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nb = n0 + wn * (BN / 2) + j * 16 + fk * 4;
      nok[j] = nb < fe.nout;
      nbs[j] = nok[j] ? nb : 0;
      bb[j] = fe.bias4(nbs[j]);
    }
    // output statistics (below): every 128-row tile but the 2-stage 128 x 160 one, whose 236 registers are what lets two workgroups share a CU (with the
    // 32 sums it ran one per CU and 40 % slower: profiles/r21 -> r23 breakdowns); that tile keeps the gn_stats pass behind it
    constexpr bool STATS_OK = BM == 128 && !(BN == 160 && NST == 2);
    float cs[NI][4], cq[NI][4];   // column sums / sums of squares of this lane's rows
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[j][r] = cq[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * (BM / 2) + i * 16 + frow;
      const bool mok = m < p.M;
      const int mc = mok ? m : p.M - 1;
      const int img = mc / fe.per_img;
      if (!gg) {
        f16x4 rv[NI], rs[NI], gt[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          rv[j] = fe.rv4(img, nbs[j]);
          rs[j] = fe.res4(mc, nbs[j]);
          gt[j] = fe.gate4(img, nbs[j]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] = fe.act_gate(acc[i][j][r] * fe.alpha + (float)bb[j][r] + (float)rv[j][r], (float)gt[j][r]) + (float)rs[j][r];
          if (mok && nok[j]) fe.store4(m, nbs[j], v);
          if (STATS_OK && p.stats) {   // (uniform) GroupNorm statistics of the fp16 values just stored
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float h = (mok && nok[j]) ? (float)(f16)v[r] : 0.f;
              cs[j][r] += h;
              cq[j][r] = fmaf(h, h, cq[j][r]);
            }
          }
        }
      } else {
        // fragments come in [value | gate] pairs along j: odd j holds the gate of fragment j-1
#pragma unroll
        for (int j = 1; j < NI; j += 2) {
          const int col = ((n0 + wn * (BN / 2)) >> 1) + (j >> 1) * 16 + fk * 4;
          const int colc = nok[j] ? col : 0;
          const f16x4 rvv = fe.rv4(img, nbs[j - 1]), rvg = fe.rv4(img, nbs[j]), rs = fe.res4(mc, colc);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float val = acc[i][j - 1][r] * fe.alpha + (float)bb[j - 1][r] + (float)rvv[r];
            const float gate = acc[i][j][r] * fe.alpha + (float)bb[j][r] + (float)rvg[r];
            v[r] = val * gelu_erf_f(gate) + (float)rs[r];
          }
          if (mok && nok[j]) fe.store4(m, col, v);
        }
      }
    }
    if (STATS_OK && p.stats) {
      // ---- round 5: the output's GroupNorm statistics from this tile (128 rows of ONE image: the dispatcher sets p.stats only then), so that a
      //      small-batch ResBlock needs no gn_stats pass over what was just written.  A lane holds, per column block j, 4 columns of MI rows summed;
      //      the 16 lanes of a column quad add up by a fixed butterfly, the two row halves of the workgroup through LDS in a fixed order:
      //      partial[image][128-row chunk][column][{sum, sum of squares}], plain stores into this tile's own slots -- deterministic.
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = cs[j][r], q1 = cq[j][r];
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) {
            s1 += __shfl_xor(s1, d);
            q1 += __shfl_xor(q1, d);
          }
          cs[j][r] = s1;
          cq[j][r] = q1;
        }
      float* red = reinterpret_cast<float*>(smem);   // [2 (wm)][BN][2]; every wave is past the K loop's last barrier
      if (frow == 0) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = wn * (BN / 2) + j * 16 + fk * 4 + r;
            red[(wm * BN + col) * 2 + 0] = cs[j][r];
            red[(wm * BN + col) * 2 + 1] = cq[j][r];
          }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < p.nout) {
        const int per = p.oh * p.ow;
        const int img = m0 / per, chunk = (m0 - img * per) / BM;
        float* dst = p.stats + ((long)(img * p.stats_nch + chunk) * p.nout + n0 + tid) * 2;
        *reinterpret_cast<f32x2*>(dst) = f32x2{red[tid * 2] + red[(BN + tid) * 2], red[tid * 2 + 1] + red[(BN + tid) * 2 + 1]};
      }
    }
    return;
  }
  const GemmEpilogue ep(p);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * (BM / 2) + i * 16 + frow;
    if (m >= p.M) continue;
    const f16* rv = ep.rowvec_of(m);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int nb = n0 + wn * (BN / 2) + j * 16 + fk * 4;  // first of this lane's 4 weight rows
      float v[4];
      int col;
      if (ep.geglu) {
        // fragments come in [value | gate] pairs along j: odd j holds the gate of fragment j-1
        if ((j & 1) == 0) continue;
        float g[4];
        const float ag[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const float av[4] = {acc[i][j - 1][0], acc[i][j - 1][1], acc[i][j - 1][2], acc[i][j - 1][3]};
        ep.biased(ag, nb, rv, g);
        ep.biased(av, nb - 16, rv, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= gelu_erf_f(g[r]);
        col = ((n0 + wn * (BN / 2)) >> 1) + (j >> 1) * 16 + fk * 4;
      } else {
        const float av[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        ep.biased(av, nb, rv, v);
        col = nb;
      }
      ep.store(m, col, v);
    }
  }
}

template <int BM, int BN, bool CONV, bool BUF, bool SPLITK = false, int NST = 2>
int launch_impl(const GemmParams& p, hipStream_t st) {
  const int smem = NST * (BM + BN) * 128;
  static_assert(NST * (BM + BN) * 128 <= 160 * 1024, "LDS ring beyond 160 KB");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, CONV, BUF, SPLITK, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  GemmParams q = p;
  q.tiles_m = (p.M + BM - 1) / BM;
  q.tiles_n = (p.nout + BN - 1) / BN;
  const int grid = q.tiles_m * q.tiles_n * (SPLITK ? p.splits : 1);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, CONV, BUF, SPLITK, NST>), dim3(grid), dim3(256), smem, st, q);
  FMX_LAUNCH_CHECK("fmx_gemm_conv_f16");
  return FMX_OK;
}

// the LDS ring (round 5; buffer-descriptor path only): one workgroup per CU, four stages (96 / 128 / 144 KB).  As many stages as 160 KB hold (5 on
// the 128x128 tile, 6 on 128x64) were measured level or slower (profiles/r20_ring_vs_two_stage_microbench.jsonl): three tiles in flight already cover
// the memory latency, the K-tile is LDS-bound
template <int BM, int BN>
constexpr int ring_depth() { return 4; }
template <int BM, int BN>
int launch_ring(const GemmParams& p, bool conv, bool split, hipStream_t st) {
  constexpr int R = ring_depth<BM, BN>();
  if (split) return conv ? launch_impl<BM, BN, true, true, true, R>(p, st) : launch_impl<BM, BN, false, true, true, R>(p, st);
  return conv ? launch_impl<BM, BN, true, true, false, R>(p, st) : launch_impl<BM, BN, false, true, false, R>(p, st);
}

// split-K instantiations exist for the 128 x 128 and 128 x 64 tiles on the buffer-descriptor path
template <int BM, int BN>
int launch_split(const GemmParams& p, bool conv, hipStream_t st) {
  return conv ? launch_impl<BM, BN, true, true, true>(p, st) : launch_impl<BM, BN, false, true, true>(p, st);
}

template <int BM, int BN, bool CONV>
int launch(const GemmParams& p, hipStream_t st) {
  return p.a0_bytes ? launch_impl<BM, BN, CONV, true>(p, st) : launch_impl<BM, BN, CONV, false>(p, st);
}

// interleave GEGLU rows: out row 32*q + i (i<16) = in row 16*q + i ; out row 32*q+16+i = in row inner + 16*q + i
__global__ void geglu_interleave_kernel(const f16* __restrict__ w_in, const f16* __restrict__ b_in, f16* __restrict__ w_out,
                                        f16* __restrict__ b_out, int inner, int k) {
  const int orow = blockIdx.x;
  const int q = orow >> 5, i = orow & 31;
  const int irow = (i < 16) ? (16 * q + i) : (inner + 16 * q + (i - 16));
  for (int c = threadIdx.x; c < k; c += blockDim.x) w_out[(long)orow * k + c] = w_in[(long)irow * k + c];
  if (b_in && threadIdx.x == 0) b_out[orow] = b_in[irow];
}

}  // namespace

int fmx_launch_gn_stats(const void* x, int32_t c, int64_t ld, int32_t n, int32_t hw, float* partial, int32_t nchunks, hipStream_t st);  // fmx_norm.hip

// a phase convolution of fmx_conv3x3_up2x: horizontal padding of its own, scattered output rows, its slice of a statistics array shared by the four phases
struct ScatterSpec {
  int pad_x, ow;
  long extra;
  int stats_nch_total;
};

static int gemm_conv_one(const fmx_gemm_args* a, float* stats, int max_chunks, int fallback_chunks, int* chunks_out, void* stream,
                         float* row_stats = nullptr, int row_parts_cap = 0, int* row_parts_out = nullptr, const ScatterSpec* sc = nullptr);

// Where the 256x160 two-workgroups-per-CU kernel replaces the 256x320 one by default: set from the same-box A/B of profiles/r10_gemm4w_ab.jsonl.
static bool fmx_gemm4w_preferred(int M, int nout, int kt, bool geglu) {
  (void)M; (void)nout; (void)kt; (void)geglu;
  return false;
}

// stats != null: also leave the GroupNorm statistics of the output in stats[n][chunks][nout][2] (see fmx_gemm_conv_stats_f16 in fmx.h)
//
// The 8-wave kernels address an operand with 32-bit byte offsets from its base.  An activation tensor beyond that range (the VAE decoder's
// 8 x 1024 x 1024 x 256 input of its last level is 4.3 GB) is processed as equal groups of whole images -- images are independent rows of the
// implicit GEMM -- each a launch of its own with the bases moved; the groups are identical in shape, so they pick the same tile and the same
// statistics chunking.
static int gemm_conv_impl(const fmx_gemm_args* a, float* stats, int max_chunks, int fallback_chunks, int* chunks_out, void* stream) {
  FMX_REQUIRE(a && a->a0 && a->wgt && a->out && a->zero_page, "gemm: null pointer");
  if (a->n > 1 && a->h > 0 && a->w > 0 && a->c0 > 0) {
    const double s0 = a->a0_stride ? a->a0_stride : a->c0, s1 = a->a1_stride ? a->a1_stride : a->c1;
    const double per_img = (double)a->h * a->w;
    auto fits = [&](int imgs) { return imgs * per_img * s0 * 2.0 < 3.0e9 && (a->c1 == 0 || imgs * per_img * s1 * 2.0 < 3.0e9); };
    if (!fits(a->n)) {
      int groups = 0;
      for (int d = 2; d <= a->n; ++d)
        if (a->n % d == 0 && fits(a->n / d)) { groups = d; break; }
      if (groups) {
        const int gi = a->n / groups;
        const long in_pix = (long)gi * a->h * a->w, out_pix = (long)gi * a->oh * a->ow;
        int chunks = 0;
        for (int g = 0; g < groups; ++g) {
          fmx_gemm_args b = *a;
          b.n = gi;
          b.a0 = (const char*)a->a0 + g * in_pix * (long)s0 * 2;
          if (a->a1) b.a1 = (const char*)a->a1 + g * in_pix * (long)s1 * 2;
          b.out = (char*)a->out + g * out_pix * a->ld_out * (a->out_f32 > 0 ? 4 : 2);
          if (a->residual) b.residual = (const char*)a->residual + g * out_pix * a->ld_res * 2;
          if (a->rowvec) b.rowvec = (const char*)a->rowvec + (long)g * gi * a->ld_rowvec * 2;
          if (a->gate) b.gate = (const char*)a->gate + (long)g * gi * a->ld_gate * 2;
          int got = 0;
          float* st = stats ? stats + (long)g * gi * chunks * a->nout * 2 : nullptr;
          const int rc = gemm_conv_one(&b, st, max_chunks, fallback_chunks, stats ? &got : nullptr, stream);
          if (rc != FMX_OK) return rc;
          if (stats) {
            FMX_REQUIRE(g == 0 || got == chunks, "gemm: image groups disagree on the statistics chunking");
            chunks = got;
          }
        }
        if (stats) *chunks_out = chunks;
        return FMX_OK;
      }
    }
  }
  return gemm_conv_one(a, stats, max_chunks, fallback_chunks, chunks_out, stream);
}

static int gemm_conv_one(const fmx_gemm_args* a, float* stats, int max_chunks, int fallback_chunks, int* chunks_out, void* stream, float* row_stats,
                         int row_parts_cap, int* row_parts_out, const ScatterSpec* sc) {
  FMX_REQUIRE(a && a->a0 && a->wgt && a->out && a->zero_page, "gemm: null pointer");
  const int ctot = a->c0 + a->c1;
  FMX_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c0 % 64) == 0 && (ctot % 64) == 0, "gemm: channels (%d,%d) must be multiples of 64", a->c0, a->c1);
  FMX_REQUIRE(a->c1 == 0 || a->a1, "gemm: a1 missing");
  FMX_REQUIRE(a->kh == 1 || a->kh == 3 || (sc && a->kh == 2), "gemm: kh must be 1 or 3");
  FMX_REQUIRE(a->n > 0 && a->h > 0 && a->w > 0 && a->oh > 0 && a->ow > 0 && a->nout > 0, "gemm: bad dims");
  FMX_REQUIRE(a->act == FMX_ACT_NONE || a->act == FMX_ACT_GELU_TANH || (a->act == FMX_ACT_GEGLU && (a->nout % 32) == 0), "gemm: bad act/nout");
  GemmParams p;
  p.a0 = (const f16*)a->a0; p.a1 = (const f16*)a->a1;
  p.c0 = a->c0; p.c1 = a->c1;
  p.s0 = a->a0_stride ? a->a0_stride : a->c0;
  p.s1 = a->a1_stride ? a->a1_stride : a->c1;
  p.n = a->n; p.h = a->h; p.w = a->w; p.oh = a->oh; p.ow = a->ow;
  p.kh = a->kh; p.stride = a->stride > 0 ? a->stride : 1; p.pad = a->pad;
  p.pad_x = sc ? sc->pad_x : a->pad;
  p.scat_ow = sc ? sc->ow : 0;
  p.scat_extra = sc ? sc->extra : 0;
  p.up_h = a->up_h; p.up_w = a->up_w;
  p.wgt = (const f16*)a->wgt;
  p.ldw = a->ldw ? a->ldw : a->kh * a->kh * ctot;
  p.nout = a->nout;
  p.bias = (const f16*)a->bias; p.rowvec = (const f16*)a->rowvec; p.ld_rowvec = a->ld_rowvec;
  p.residual = (const f16*)a->residual; p.ld_res = a->ld_res;
  p.alpha = a->alpha; p.act = a->act;
  p.out = a->out; p.ld_out = a->ld_out; p.out_f32 = a->out_f32 > 0 ? a->out_f32 : 0;   // (negative: the tile-forcing test hook below, fp16 output)
  p.zp = (const f16*)a->zero_page;
  p.gate = (const f16*)a->gate; p.ld_gate = a->ld_gate;
  p.M = a->n * a->oh * a->ow;
  p.cpt = ctot / 64;
  p.kt = a->kh * a->kh * p.cpt;
  p.tiles_m = p.tiles_n = 0;
  p.stats = nullptr;
  p.stats_nch = 0;
  p.splits = 1;
  p.ws = nullptr;
  p.tickets = nullptr;
  p.row_stats = nullptr;
  p.ln_partial = nullptr;
  p.ln_parts = 0;
  p.ln_colsum = nullptr;
  p.ln_eps = p.ln_inv_c = 0.f;
  p.ln_col_ab = p.ln_row_cb = nullptr;
  p.ln_ab_out = nullptr;
  p.xa_k = p.xa_vt = nullptr;
  p.xa_k_rs = p.xa_k_bs = p.xa_vt_ds = p.xa_vt_bs = p.xa_nk = p.xa_rows = 0;
  p.xa_k_bytes = p.xa_vt_bytes = 0;
  p.xa_c2 = 0.f;
  FMX_REQUIRE(fmx_aligned16(p.a0) && fmx_aligned16(p.wgt) && fmx_aligned16(p.zp) && (!p.a1 || fmx_aligned16(p.a1)), "gemm: operands must be 16-byte aligned");
  FMX_REQUIRE((p.s0 % 8) == 0 && (p.s1 % 8) == 0 && (p.ldw % 8) == 0, "gemm: strides must be multiples of 8 elements");
  FMX_REQUIRE((long)p.M * 1 > 0 && (long)a->n * a->oh * a->ow < (1L << 31), "gemm: M overflow");
  FMX_REQUIRE((!a->gate && a->act != FMX_ACT_GELU_TANH) || FastEpilogue::eligible(p) || a->out_f32 < 0,
              "gemm: gate / GELU-tanh need fp16 output and leading dimensions / nout multiples of 4 / 8");
  const bool conv = !(a->kh == 1 && p.stride == 1 && a->pad == 0 && a->up_h == 0 && a->oh == a->h && a->ow == a->w);
  hipStream_t st = (hipStream_t)stream;
  // the 8-wave kernels (and the buffer-descriptor path of the 4-wave ones) address their operands with 32-bit byte offsets;
  // offsets >= 0xC0000000 are the "load zeros" marker, so every operand must span less than that
  const double npix = (double)a->n * a->h * a->w;
  const double a0_span = ((npix - 1) * p.s0 + p.c0) * 2.0, a1_span = p.c1 ? ((npix - 1) * p.s1 + p.c1) * 2.0 : 16.0;
  const double w_span = ((double)(p.nout - 1) * p.ldw + (double)a->kh * a->kh * ctot) * 2.0;
  const bool fits32 = a0_span < 3.0e9 && a1_span < 3.0e9 && w_span < 3.0e9;
  p.a0_bytes = fits32 ? (unsigned)a0_span : 0;
  p.a1_bytes = fits32 ? (unsigned)a1_span : 0;
  p.w_bytes = fits32 ? (unsigned)w_span : 0;
  // Tile choice by a cost model fitted to tools/bench_kernels.py sweeps (profiles/r02o_gemm_sweep.jsonl).  In units of
  // "one output element x one K-tile at the 256x320 kernel's steady-state rate":
  //   time = rounds x wpc x ( area x (kt + e) / eff + F ),    rounds = ceil(tiles / (256 CUs x wpc)),
  // wpc = resident workgroups per CU, eff = steady-state rate relative to the 256x320 kernel, e = epilogue cost per output
  // element in K-tiles (it scales with the tile area), F = per-tile fixed latency (prologue + first loads, ~1.5 K-tiles of a
  // 256x256 tile; half of it hides behind the CU's other workgroup for the 4-wave kernels).  Padding rows / columns are
  // counted through the tile area.
  // sel: 0 = 128x128, 1 = 128x64, 2 = 64x64, 4 = 128x160, 5 / 6 / 7 / 8 = 256x256 / 256x320 / 320x256 / 512x128 pipelined,
  //      9 = 256x160 with two 4-wave workgroups per CU (fmx_gemm4w.hip; linear GEMMs)
  auto tiles = [&](int bm, int bn) { return (double)((p.M + bm - 1) / bm) * (double)((p.nout + bn - 1) / bn); };
  const double kt = p.kt;
  const bool geglu = a->act == FMX_ACT_GEGLU;
  auto cost = [&](int bm, int bn, int wpc, double eff, double e, double fixed) {
    const double rounds = ceil(tiles(bm, bn) / (256.0 * wpc));
    return rounds * wpc * ((double)bm * bn * (kt + e) / eff + fixed);
  };
  const double e4v = geglu ? 5.0 : 2.0, F4v = 49152.0;
  // (round 5) the 2-stage 4-wave tiles in a launch of ONE round: the constants above were fitted to multi-round launches, where a CU's workgroups
  // start staggered and cover each other's prologue / epilogue; in one round they start and end together.  Measured (tools/bench_kernels.py
  // smallconv / ring, profiles/r23_small_batch_conv_tiles.jsonl): twice the fixed cost, the 128 x 160 tile at 0.62 (3x3 convolutions: 0.55) instead
  // of 0.80, convolutions on the other two at 0.83 of their linear rate.  The 8-wave tiles' prices were right as they are (73.8 us predicted 75).
  const char* er5 = fmx_knob("FMX_GEMM_R5COST");   // A/B knob: 0 = round 4's prices for these launches
  const bool r5cost = !(er5 && atoi(er5) == 0);
  auto cost4 = [&](int bm, int bn, int wpc, double eff) {
    if (!r5cost || tiles(bm, bn) > 256.0 * wpc) return cost(bm, bn, wpc, eff, e4v, F4v);
    const double e1 = bn == 160 ? (conv ? 0.55 : 0.62) : eff * (conv ? 0.83 : 1.0);
    return cost(bm, bn, wpc, e1, e4v, 100000.0);
  };
  const bool big_ok = FastEpilogue::eligible8(p) && fits32;
  const double e4 = geglu ? 5.0 : 2.0, F4 = 49152.0, e8 = 4.0, F8 = 98304.0;
  int sel = 2, best_s = 1;
  double best = cost(64, 64, 5, 0.42, e4, F4);
  auto consider = [&](int id, double c, int s = 1) { if (c < best) { best = c; sel = id; best_s = s; } };
  consider(1, cost4(128, 64, 3, 0.58));
  consider(0, cost4(128, 128, 2, 0.70));
  if (!geglu && (p.nout % 160) == 0) consider(4, cost4(128, 160, 2, 0.80));
  if (big_ok) consider(5, cost(256, 256, 1, 0.97, e8, F8));
  if (big_ok && (!geglu || (p.nout % 32) == 0)) consider(6, cost(256, 320, 1, 1.0, e8, F8));
  if (big_ok && !geglu) consider(7, cost(320, 256, 1, 0.97, e8, F8));  // only where its row quantisation wins (M = 320 k: the V^T GEMM)
  // 512x128: narrow outputs (the VAE decoder's 128-channel level); its address path takes single-source convolutions without upsample-on-load
  const bool narrow_ok = big_ok && !geglu && (!conv || (p.c1 == 0 && a->up_h == 0));
  if (narrow_ok) consider(8, cost(512, 128, 1, 0.86, e8, F8));
  // split-K over S workgroups per tile (128x128 / 128x64 tiles) when the problem leaves workgroup slots empty: K-loop / S, plus the
  // partial-accumulator exchange (one fp32 tile written per workgroup, S - 1 read by the last: ES K-tile equivalents, fitted to
  // tools/bench_kernels.py splitk sweeps).  Needs the caller's workspace (fmx_gemm_args.workspace) and the buffer-descriptor path.
  constexpr long TICKET_BYTES = 65536;
  const long ws_floats = (a->workspace && a->workspace_bytes > TICKET_BYTES) ? (a->workspace_bytes - TICKET_BYTES) / 4 : 0;
  const char* force_split_env = fmx_knob("FMX_GEMM_SPLITK");   // A/B / test knob, read per call: 0 = never split, S >= 2 = always S (where possible)
  const int force_split = force_split_env ? atoi(force_split_env) : -1;
  const bool split_ok = ws_floats > 0 && fits32 && a->out_f32 <= 0 && force_split != 0;
  auto split_fits = [&](int bm, int bn, int S) {
    const double t = tiles(bm, bn);
    return S >= 2 && p.kt / S >= 2 && t <= TICKET_BYTES / 4 && t * S * bm * bn <= (double)ws_floats;
  };
  if (split_ok && force_split < 0) {
    auto cost_split = [&](int bm, int bn, int wpc, double eff, int S) {
      const double rounds = ceil(tiles(bm, bn) * S / (256.0 * wpc));
      // measured (profiles/r05c_split_k_sweep.jsonl, in-graph, cold weights): the hand-over costs 12 us at S = 2 and 18 us at S = 3 for a
      // 128 x 128 tile (~0.93 us per K-tile there), half again as much per K-tile for 128 x 64 -- it pays from ~50 K-tiles per workgroup on
      const double es = (bn == 128 ? 6.5 : 9.0) * S;
      return rounds * wpc * ((double)bm * bn * (kt / S + e4 + es) / eff + F4);
    };
    for (int S = 2; S <= 8; ++S) {
      if (p.kt / S < 4) break;
      if (split_fits(128, 128, S)) consider(0, cost_split(128, 128, 2, 0.70, S), S);
      if (split_fits(128, 64, S)) consider(1, cost_split(128, 64, 3, 0.58, S), S);
    }
  }
  // ---- round 5: the 4-wave tiles on a 4-stage LDS ring (ids 10 / 11 / 12 / 13 = 128x128 / 128x160 / 128x64 / 64x160; force_tile = id + 1), ONE workgroup per CU with three K-tiles in
  //      flight.  What it is for: launches with fewer tiles than CUs (UNet batch 2..8 at the 8^2..32^2 levels), where the 2-stage kernels above pay a
  //      full memory round trip per K-tile (~0.8-0.9 us; the model's "wpc x area / eff" happens to price that correctly) and the ring pays
  //      ~0.4 us.  Same epilogue, same split-K hand-over (ids 10 and 12).  FMX_GEMM_RING: 0 = never, 1 (default) = by the cost model,
  //      2 = wherever eligible (A/B).
  const char* ering = fmx_knob("FMX_GEMM_RING");
  const int ring_mode = ering ? atoi(ering) : 1;
  const bool ring_ok = fits32 && ring_mode != 0;
  if (ring_ok) {
    // fitted to tools/bench_kernels.py ring (profiles/r20_ring_vs_two_stage_microbench.jsonl; in a graph, cold weights): a K-tile costs the ring
    // 0.60 / 0.72 / 0.43 us on the 128x128 / 128x160 / 128x64 tile (2-stage forms: 0.9-1.0 / - / 0.8) and a launch 12-14 K-tiles' worth of fixed time
    // (dispatch, first tile's round trip, epilogue); a 3x3 convolution's address arithmetic slows the K-tile by a quarter.  Neither more stages (5 / 6:
    // level or worse) nor hot weights (19.7 vs 21.0 us) move it.  What bounds the K-tile is the CU's LDS-DMA INGEST, not the LDS reads or the MFMA
    // pipe: a K-tile is (BM + BN) / 8 one-KiB pieces per CU (32 / 36 / 24 / 28 on 128x128 / 128x160 / 128x64 / 64x160), a CU takes a piece per ~13 ns
    // beside MFMA waves (tools/ubench/dma_rate.hip, profiles/r02b: 10.4 ns alone) = 0.42 / 0.48 / 0.32 / 0.37 us, and every tile sits at 1.35-1.4 x
    // that.  Hand-ordering the fragment reads (all of a k-step requested behind the barrier, the LDS-DMA issue under their latency, counted lgkmcnt
    // waits, k-step 1 under k-step 0's MFMAs -- the compiler's own order exposes ~600 cycles of LDS latency per K-tile) was built and measured:
    // linear launches 0-5 % SLOWER, convolutions -3 % .. +5 %, steps level (profiles/r30_ring_ordered_fragment_reads_negative_result.jsonl) -- the
    // latency it hides was never on the critical path.  Fewer bytes per FLOP needs a bigger tile, which a small-M launch cannot fill the chip with.
    if (ring_mode == 2) best = 1e300;
    auto cost_ring = [&](int bm, int bn, int S) {
      // (64 x 160, id 14: a K-tile 0.51 us, 13 K-tiles fixed -- N = 1280 at M = 2048 is 256 tiles of it, one per CU, where 128 x 128 leaves 96 CUs idle:
      //  16.8 against 20.3 us at K = 1280, 47.3 against 55.5 at K = 5120; convolutions 0.77 of that)
      const double effr = (bm == 64 || bm == 160) ? (conv ? 0.34 : 0.44) : (bn == 128 ? 0.606 : bn == 160 ? 0.63 : 0.416) * (conv ? 0.80 : 1.0);
      const double E = (bm == 64 || bm == 160) ? 13.0 : bn == 128 ? 14.3 : 12.5;
      const double rounds = ceil(tiles(bm, bn) * S / 256.0);
      const double es = S > 1 ? 2.7 * S + 1.1 : 0.0;
      return rounds * (double)bm * bn * (kt / S + E + es) / effr;
    };
    consider(10, cost_ring(128, 128, 1));
    consider(12, cost_ring(128, 64, 1));
    if (!geglu && (p.nout % 160) == 0) consider(11, cost_ring(128, 160, 1));
    if (!geglu && (p.nout % 160) == 0) consider(13, cost_ring(64, 160, 1));
    if (!geglu && (p.M % 160) == 0) consider(14, cost_ring(160, 64, 1));   // the same tile lying down: the operand-swapped V^T projection (M = 1280 weight rows)
    if (split_ok && force_split < 0)
      for (int S = 2; S <= 8; ++S) {
        if (p.kt / S < 6) break;
        if (split_fits(128, 128, S)) consider(10, cost_ring(128, 128, S), S);
        if (split_fits(128, 64, S)) consider(12, cost_ring(128, 64, S), S);
      }
  }
  // 256x160 x 2 workgroups per CU: a plain linear GEMM (single source, 1x1, no output statistics) with the 8-wave kernels' epilogue contract
  const bool w4_ok = big_ok && !conv && p.c1 == 0 && !stats && !a->ln_col_ab && (!geglu || (p.nout % 32) == 0);
  {
    // where the dispatcher takes it by itself: FMX_GEMM_4W = 0 never, 1 (default) by the rule below, 2 wherever it is eligible (A/B knob, development only)
    const char* e4 = fmx_knob("FMX_GEMM_4W");
    const int mode4 = e4 ? atoi(e4) : 1;
    if (w4_ok && sel == 6 && (mode4 == 2 || (mode4 == 1 && fmx_gemm4w_preferred(p.M, p.nout, p.kt, geglu)))) sel = 9;
  }
  if (a->out_f32 < 0) { sel = (-a->out_f32 - 1) % 16; best_s = 1; }  // test hook: force a tile shape (out_f32 = -1..-10 -> fp16 out)
  if (sc) {   // scattered output rows exist in the 256 x 256 / 256 x 320 kernels only
    FMX_REQUIRE(big_ok && !geglu && p.c1 == 0 && a->up_h == 0, "gemm: a phase convolution needs the 256-row kernels (fp16 output, aligned operands < 3 GB, one source)");
    sel = cost(256, 320, 1, 1.0, e8, F8) <= cost(256, 256, 1, 0.97, e8, F8) ? 6 : 5;
    best_s = 1;
  }
  if (split_ok && force_split >= 2 && (sel == 0 || sel == 1 || sel == 10 || sel == 12) && split_fits(128, (sel == 0 || sel == 10) ? 128 : 64, force_split))
    best_s = force_split;
  if (best_s > 1 && sel != 0 && sel != 1 && sel != 10 && sel != 12) best_s = 1;   // (ids 12 / 14, the 160-wide ring tiles, have no split-K form)
  if (fmx_knob("FMX_GEMM_DEBUG"))
    fprintf(stderr, "fmx_gemm: M=%d N=%d K=%d conv=%d -> tile id %d, split-K %d (workspace %ld floats, fits32 %d)\n", p.M, p.nout, p.kt * 64, (int)conv,
            sel + 1, best_s, ws_floats, (int)fits32);
  FMX_REQUIRE(sel <= 14, "gemm: unknown tile id");
  FMX_REQUIRE(sel < 10 || fits32, "gemm: the ring tiles (ids 11-13) address their operands through buffer descriptors (operands < 3 GB)");
  FMX_REQUIRE((sel != 11 && sel != 13 && sel != 14) || (a->act != FMX_ACT_GEGLU), "gemm: the 128x160 / 64x160 / 160x64 ring tiles do not support GEGLU");
  FMX_REQUIRE(sel != 9 || w4_ok, "gemm: the 256x160 two-workgroup tile takes plain linear GEMMs (one source, no output statistics, fp16 out, 16-byte aligned operands)");
  FMX_REQUIRE(sel != 8 || (!geglu && (!conv || (p.c1 == 0 && a->up_h == 0))), "gemm: the 512x128 tile takes no GEGLU, second source or upsample-on-load");
  FMX_REQUIRE(sel != 3, "gemm: tile id 4 (the first-generation ping-pong kernel) is no longer part of the library");
  FMX_REQUIRE(sel != 4 || a->act != FMX_ACT_GEGLU, "gemm: the 128x160 tile does not support GEGLU");
  if (a->out_f32 < 0) p.out_f32 = 0;
  // output statistics: from the epilogue of a 256-row tile when every image is a whole number of tiles, else a pass behind the GEMM
  const int per_img = a->oh * a->ow;
  bool stats_after = false;
  if (stats && sc) {   // the caller (fmx_conv3x3_up2x) checked that an image is a whole number of 256-row tiles and offset `stats` to this phase's records
    p.stats = stats;
    p.stats_nch = sc->stats_nch_total;
  } else if (stats) {
    FMX_REQUIRE(!geglu && !p.out_f32 && p.ld_out == p.nout && (p.nout % 8) == 0 && !p.gate && a->act == FMX_ACT_NONE,
                "gemm: output statistics need a dense fp16 [M][nout] output without activation / gate");
    FMX_REQUIRE(fallback_chunks >= 1 && fallback_chunks <= 1024 && max_chunks >= fallback_chunks, "gemm: bad statistics chunk counts");
    // (round 5) the 128-row 4-wave tiles emit them too -- the small-batch launches, whose tensors used to get a gn_stats pass of their own
    const bool rows128 = (sel == 0 || sel == 1 || sel == 10 || sel == 11 || sel == 12) && FastEpilogue::eligible(p);   // the 128-ROW tiles (not 64 x 160 / 160 x 64)
    const int rows = sel == 8 ? 512 : rows128 ? 128 : 256;
    if ((sel == 5 || sel == 6 || sel == 8 || rows128) && (per_img % rows) == 0 && per_img / rows <= max_chunks) {
      p.stats = stats;
      p.stats_nch = per_img / rows;
      *chunks_out = p.stats_nch;
    } else {
      stats_after = true;
      *chunks_out = fallback_chunks;
    }
  }
  // LayerNorm folded into the GEMMs around it (fmx.h: fmx_gemm_linear_rowstats_f16 / the ln_* fields): 256x320 tile, linear only
  const bool ln_shape_ok = !conv && big_ok && p.c1 == 0 && !p.gate && !p.rowvec;
  if (row_parts_out) {
    const int parts = sel == 9 ? (p.nout + 159) / 160 : 2 * ((p.nout + 319) / 320);   // one entry per 160 output columns either way
    if ((sel == 6 || sel == 9) && ln_shape_ok && a->act == FMX_ACT_NONE && parts <= row_parts_cap && !stats) {   // (residual optional: absent reads the zero page)
      p.row_stats = row_stats;
      *row_parts_out = parts;
    } else {
      *row_parts_out = 0;   // another tile shape suits this problem better: plain GEMM, the caller runs its LayerNorm kernel
    }
  }
  if (a->ln_partial) {
    FMX_REQUIRE(ln_shape_ok && !p.residual && !stats && (a->act == FMX_ACT_NONE || (geglu && (p.nout % 32) == 0)) && a->ln_colsum && a->ln_parts >= 1 &&
                    a->ln_parts <= 8 && fmx_aligned16(a->ln_partial) && fmx_aligned16(a->ln_colsum),
                "gemm: a LayerNorm-folded GEMM is a plain linear (bias, optional GEGLU) with fp16 output on the 256x320 tile");
    p.ln_partial = (const float*)a->ln_partial;
    p.ln_parts = a->ln_parts;
    p.ln_colsum = (const float*)a->ln_colsum;
    p.ln_eps = a->ln_eps;
    p.ln_inv_c = 1.0f / (float)p.c0;
    p.ln_ab_out = (float*)a->ln_ab_out;
    if (sel != 9) sel = 6;
    best_s = 1;
  }
  if (a->xa_k) {   // cross-attention as the epilogue of this (query) projection
#ifdef FMX_ELEM_BF16
    FMX_REQUIRE(false, "gemm: the cross-attention epilogue exists in the fp16 build only");
#endif
    FMX_REQUIRE(a->ln_partial && a->act == FMX_ACT_NONE && !row_parts_out && a->xa_vt && (p.nout % 320) == 0 && a->xa_rows > 0 && (a->xa_rows % 256) == 0 &&
                    (p.M % a->xa_rows) == 0 && a->xa_nk >= 1 && a->xa_nk <= 80 && a->xa_k_bs >= 80 && a->xa_vt_bs >= 80 && a->xa_k_rs >= p.nout && (a->xa_k_rs % 8) == 0 &&
                    (a->xa_vt_ds % 8) == 0 && (a->xa_vt_bs % 8) == 0 && fmx_aligned16(a->xa_k) && fmx_aligned16(a->xa_vt) && a->xa_k_bytes > 0 &&
                    a->xa_k_bytes < 0xC0000000ll && a->xa_vt_bytes > 0 && a->xa_vt_bytes < 0xC0000000ll && !a->ln_ab_out && a->alpha == 1.0f,
                "gemm: the cross-attention epilogue takes a LayerNorm-consumer query projection (256x320 tile): nout %% 320 == 0, queries per image %% 256 == 0, 1..80 keys, "
                "K rows / V^T columns padded to >= 80 keys per image, 16-byte aligned K / V^T with strides %% 8 == 0");
    p.xa_k = (const f16*)a->xa_k;
    p.xa_vt = (const f16*)a->xa_vt;
    p.xa_k_rs = a->xa_k_rs; p.xa_k_bs = a->xa_k_bs; p.xa_vt_ds = a->xa_vt_ds; p.xa_vt_bs = a->xa_vt_bs; p.xa_nk = a->xa_nk; p.xa_rows = a->xa_rows;
    p.xa_k_bytes = (unsigned)a->xa_k_bytes; p.xa_vt_bytes = (unsigned)a->xa_vt_bytes;
    p.xa_c2 = a->xa_scale * 1.4426950408889634f;
    sel = 6;
  }
  if (a->ln_col_ab) {
    FMX_REQUIRE(!a->ln_partial && !row_parts_out && ln_shape_ok && !p.residual && !p.bias && !stats && a->act == FMX_ACT_NONE && a->ln_row_cb &&
                    (p.M % 320) == 0 && fmx_aligned16(a->ln_col_ab) && fmx_aligned16(a->ln_row_cb),
                "gemm: the operand-swapped LayerNorm-folded GEMM is a plain linear (no bias / residual) with fp16 output, M a multiple of 320, on the 320x256 tile");
    p.ln_col_ab = (const float*)a->ln_col_ab;
    p.ln_row_cb = (const float*)a->ln_row_cb;
    sel = 7;
    best_s = 1;
  }
  int rc;
  if (sel == 9) {
    rc = fmx_launch_gemm4w(p, st);
  } else if (sel >= 10) {
    if (best_s > 1) {
      p.splits = best_s;
      p.tickets = reinterpret_cast<int*>(a->workspace);
      p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(a->workspace) + TICKET_BYTES);
    }
    rc = sel == 10 ? launch_ring<128, 128>(p, conv, best_s > 1, st) : sel == 12 ? launch_ring<128, 64>(p, conv, best_s > 1, st)
       : sel == 13 ? (conv ? launch_impl<64, 160, true, true, false, ring_depth<64, 160>()>(p, st) : launch_impl<64, 160, false, true, false, ring_depth<64, 160>()>(p, st))
       : sel == 14 ? (conv ? launch_impl<160, 64, true, true, false, ring_depth<160, 64>()>(p, st) : launch_impl<160, 64, false, true, false, ring_depth<160, 64>()>(p, st))
                   : (conv ? launch_impl<128, 160, true, true, false, ring_depth<128, 160>()>(p, st) : launch_impl<128, 160, false, true, false, ring_depth<128, 160>()>(p, st));
  } else if (sel >= 5) {
    FMX_REQUIRE(FastEpilogue::eligible8(p) && fits32, "gemm: the 256-row kernels need fp16 output, 16-byte aligned epilogue operands, leading dimensions / nout multiples of 8, operands < 2^32 elements");
    if (sel == 6) FMX_REQUIRE(a->act != FMX_ACT_GEGLU || (p.nout % 32) == 0, "gemm: GEGLU needs nout % 32 == 0");
    rc = fmx_launch_gemm256p(p, conv, sel == 8 ? 512 : sel == 7 ? 320 : 256, sel == 8 ? 128 : sel == 6 ? 320 : 256, st);
  } else if (best_s > 1) {
    p.splits = best_s;
    p.tickets = reinterpret_cast<int*>(a->workspace);
    p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(a->workspace) + TICKET_BYTES);
    rc = sel == 0 ? launch_split<128, 128>(p, conv, st) : launch_split<128, 64>(p, conv, st);
  } else if (sel == 4) {
    rc = conv ? launch<128, 160, true>(p, st) : launch<128, 160, false>(p, st);
  } else if (conv) {
    rc = sel == 0 ? launch<128, 128, true>(p, st) : sel == 1 ? launch<128, 64, true>(p, st) : launch<64, 64, true>(p, st);
  } else {
    rc = sel == 0 ? launch<128, 128, false>(p, st) : sel == 1 ? launch<128, 64, false>(p, st) : launch<64, 64, false>(p, st);
  }
  if (rc != FMX_OK || !stats_after) return rc;
  return fmx_launch_gn_stats(p.out, p.nout, p.ld_out, a->n, per_img, stats, fallback_chunks, st);
}

extern "C" int fmx_gemm_conv_f16(const fmx_gemm_args* a, void* stream) { return gemm_conv_impl(a, nullptr, 0, 0, nullptr, stream); }

// conv3x3(pad 1) of the x2 nearest-upsampled input = four 2 x 2 convolutions on the INPUT grid, one per output parity (see fmx.h)
extern "C" int fmx_conv3x3_up2x_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt4, const void* bias, int32_t nout, void* out,
                                    float* stats, int32_t stats_cap, int32_t* stats_nchunks, const void* zero_page, void* stream) {
  FMX_REQUIRE(x && wgt4 && out && zero_page && n > 0 && h > 0 && w > 0, "conv3x3_up2x: bad arguments");
  FMX_REQUIRE(c > 0 && (c % 64) == 0 && nout > 0 && (nout % 8) == 0, "conv3x3_up2x: input channels must be a multiple of 64, output channels of 8 (got %d, %d)", c, nout);
  FMX_REQUIRE((w % 32) == 0, "conv3x3_up2x: the input width must be a multiple of 32 (got %d)", w);
  const long per = (long)h * w;
  FMX_REQUIRE((double)n * per * c * 2.0 < 3.0e9, "conv3x3_up2x: input beyond the 32-bit offset range of one launch (split the batch)");
  int nch = 0;
  if (stats) {
    FMX_REQUIRE((per % 256) == 0, "conv3x3_up2x: output statistics need h * w to be a multiple of 256 (got %ld)", per);
    nch = (int)(4 * (per / 256));
    FMX_REQUIRE(stats_cap >= nch, "conv3x3_up2x: the statistics buffer holds %d records per image, the four phases write %d", stats_cap, nch);
  }
  for (int ph = 0; ph < 4; ++ph) {
    const int py = ph >> 1, px = ph & 1;
    fmx_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.a0 = x; a.c0 = c; a.a0_stride = c;
    a.n = n; a.h = h; a.w = w; a.oh = h; a.ow = w;
    a.kh = 2; a.stride = 1; a.pad = 1 - py;                       // rows {iy - 1, iy} for the even output rows, {iy, iy + 1} for the odd ones
    a.wgt = (const char*)wgt4 + (long)ph * nout * 4 * c * 2; a.ldw = 4 * c; a.nout = nout;
    a.bias = bias;
    a.alpha = 1.0f; a.act = FMX_ACT_NONE;
    a.out = (char*)out + ((long)py * 2 * w + px) * nout * 2;      // pixel (py, px) of the [n][2h][2w][nout] output
    a.ld_out = 2 * nout;                                          // the next pixel of this phase is two pixels on
    a.zero_page = zero_page;
    const ScatterSpec sc{1 - px, w, (long)2 * w * nout, nch};     // ... and its next row two rows down: 4 w nout - w * ld_out more per row
    int got = 0;
    const int rc = gemm_conv_one(&a, stats ? stats + (long)ph * (per / 256) * nout * 2 : nullptr, nch, 1, &got, stream, nullptr, 0, nullptr, &sc);
    if (rc != FMX_OK) return rc;
  }
  if (stats_nchunks) *stats_nchunks = nch;
  return FMX_OK;
}

#ifndef FMX_ELEM_BF16  // the LDM transformer blocks run in fp16
extern "C" int fmx_gemm_linear_rowstats_f16(const fmx_gemm_args* a, float* row_partial, int32_t parts_cap, int32_t* parts_out, void* stream) {
  FMX_REQUIRE(a && row_partial && parts_out && parts_cap >= 2, "gemm_linear_rowstats: bad arguments");
  FMX_REQUIRE(!a->ln_partial, "gemm_linear_rowstats: a GEMM is either the producer or the consumer of a folded LayerNorm");
  return gemm_conv_one(a, nullptr, 0, 0, nullptr, stream, row_partial, parts_cap, parts_out);
}
#endif

extern "C" int fmx_gemm_conv_stats_f16(const fmx_gemm_args* a, float* partial, int32_t max_chunks, int32_t fallback_chunks, int32_t* chunks_out,
                                       void* stream) {
  FMX_REQUIRE(partial && chunks_out, "gemm_conv_stats: null pointer");
  return gemm_conv_impl(a, partial, max_chunks, fallback_chunks, chunks_out, stream);
}

#ifndef FMX_ELEM_BF16  // a 16-bit row shuffle: one copy serves both element types
extern "C" int fmx_geglu_interleave_rows(const void* w_in, const void* b_in, void* w_out, void* b_out, int32_t inner,
                                         int32_t k, void* stream) {
  FMX_REQUIRE(w_in && w_out && inner > 0 && (inner % 16) == 0 && k > 0, "geglu_interleave: bad args");
  FMX_REQUIRE((b_in == nullptr) == (b_out == nullptr), "geglu_interleave: bias in/out mismatch");
  hipLaunchKernelGGL(geglu_interleave_kernel, dim3(2 * inner), dim3(256), 0, (hipStream_t)stream, (const f16*)w_in,
                     (const f16*)b_in, (f16*)w_out, (f16*)b_out, inner, k);
  FMX_LAUNCH_CHECK("fmx_geglu_interleave_rows");
  return FMX_OK;
}
#endif
