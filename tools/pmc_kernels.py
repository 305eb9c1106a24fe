"""A handful of launches of each hot kernel at the SDXL 1024^2 batch-8 shapes, for rocprofv3 --pmc passes (MFMA-busy / VALU / wave-cycle
counters per dispatch):  rocprofv3 --kernel-trace --pmc <counters> -- python tools/pmc_kernels.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import rnd

DEV = "cuda"
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def attn(b, h, nq, nk, d=64):
    nkp = -(-nk // 64) * 64
    q, k, vt = rnd(b, nq, h, d), rnd(b, nkp, h, d), rnd(h, d, b, nkp)
    out = torch.empty(b * nq, h * d, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        ops.attention(q, k, vt, batch=b, heads=h, nq=nq, nk=nk, nk_pad=nkp, dpad=d, scale=d ** -0.5, q_bs=nq * h * d, q_rs=h * d, k_bs=nkp * h * d,
                      k_rs=h * d, vt_bs=nkp, vt_hs=d * b * nkp, vt_ds=b * nkp, out=out)


def gemm(m, n, k):
    x, w, b = rnd(m, k), rnd(n, k, scale=k ** -0.5), rnd(n)
    out = torch.empty(m, n, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=n)


def conv(n, h, w, c, co):
    x, wk, b = rnd(n, h, w, c), rnd(co, 9 * c, scale=(9 * c) ** -0.5), rnd(co)
    out = torch.empty(n * h * w, co, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        ops.conv_gemm(x, wk, co, kh=3, pad=1, bias=b, out=out, ld_out=co)


def gn(n, h, w, c):
    x, g, bb = rnd(n, h, w, c), rnd(c), rnd(c)
    out = torch.empty_like(x)
    for _ in range(REPS):
        ops.groupnorm(x, g, bb, 1e-5, silu=True, out=out)


def geglu(m, inner, k):
    x, w, b = rnd(m, k), rnd(2 * inner, k, scale=k ** -0.5), rnd(2 * inner)
    wi, bi = ops.geglu_interleave(w, b)
    out = torch.empty(m, inner, dtype=torch.float16, device=DEV)
    for _ in range(REPS):
        ops.conv_gemm(x, wi, 2 * inner, bias=bi, act=ops.ACT_GEGLU, out=out, ld_out=inner)


def ln_fold(m, c, inner):
    """the LayerNorm-folding pair: producer (bias + in-place residual + row statistics), consumers (plain projection, GEGLU)"""
    from forge_amd.backend.nn.unet import _fold_layernorm
    o, w_out, b_out, h = rnd(m, c), rnd(c, c, scale=c ** -0.5), rnd(c), rnd(m, c)
    gamma, beta = 1 + 0.1 * rnd(c), 0.1 * rnd(c)
    wq = rnd(c, c, scale=c ** -0.5)
    wg, bg = ops.geglu_interleave(rnd(2 * inner, c, scale=c ** -0.5), rnd(2 * inner))
    fq, fg = _fold_layernorm(wq, None, gamma, beta), _fold_layernorm(wg, bg, gamma, beta)
    rs = ops.RowStats(m, c)
    for _ in range(REPS):
        ops.linear(o, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs)
        ops.conv_gemm(h, fq[0], c, bias=fq[2], ln=(rs, fq[1], 1e-5))
        ops.conv_gemm(h, fg[0], 2 * inner, bias=fg[2], act=ops.ACT_GEGLU, ln=(rs, fg[1], 1e-5))


attn(16, 10, 4096, 4096)
attn(16, 20, 1024, 1024)
attn(16, 20, 1024, 77)
gemm(16384, 1280, 1280)
gemm(65536, 640, 640)
gemm(16384, 1280, 5120)
conv(16, 32, 32, 1280, 1280)
conv(16, 64, 64, 640, 640)
gn(16, 128, 128, 320)
geglu(16384, 5120, 1280)
ln_fold(16384, 1280, 5120)
conv(4, 1024, 1024, 128, 128)     # the VAE decoder's last level: 512x128 tile
torch.cuda.synchronize()
print("pmc_kernels done")
