"""A few launches of the fused GroupNorm + SiLU + conv3x3 kernel (csrc/fmx_conv_patch.hip) at the VAE decoder's full-resolution shape (8 x 1024^2 x 128 -> 128)
for rocprofv3 --pmc passes: matrix-pipe busy, wave wait fractions, LDS bank conflicts."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops
from tools.bench_kernels import rnd

n, hh, ww, cin = 8, 1024, 1024, 128
x = rnd(n, hh, ww, cin) + 0.3
g, b = 1 + 0.1 * rnd(cin), 0.1 * rnd(cin)
wk = rnd(128, 9 * cin, scale=1 / math.sqrt(9 * cin))
bias = rnd(128)
st = ops.groupnorm_stats(x)
out = ops.empty((n * hh * ww, 128), torch.float16)
for _ in range(3):
    ops.conv3x3_gn_silu(x, g, b, 1e-6, wk, bias, out=out, stats=st)
torch.cuda.synchronize()
print("pmc_convgn done")
