"""Debug: the statistics-emitting GEMM vs the plain one on the two cases where one output element differed by one fp16 ulp."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops

def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator("cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).half().to("cuda")

for (n, hh, ww, cin, cout, kh, tile) in [(2, 16, 32, 128, 640, 1, 7), (3, 16, 16, 64, 256, 3, 6), (2, 32, 32, 64, 320, 3, 7)]:
    x = rnd(n, hh, ww, cin, seed=300)
    wt = rnd(cout, cin, kh, kh, scale=1 / math.sqrt(cin * kh * kh), seed=301)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    b = rnd(cout, seed=302)
    kw = dict(kh=kh, pad=kh // 2, force_tile=tile, bias=b)
    s1, _ = ops.conv_gemm(x, wk, cout, stats=True, **kw)
    s2, _ = ops.conv_gemm(x, wk, cout, stats=True, **kw)
    p1 = ops.conv_gemm(x, wk, cout, **kw)
    p2 = ops.conv_gemm(x, wk, cout, **kw)
    print(f"case {(n, hh, ww, cin, cout, kh, tile)}: stats==stats {torch.equal(s1, s2)}  plain==plain {torch.equal(p1, p2)}  stats==plain {torch.equal(s1, p1)}")
    # rounding residue: (acc + bias) - plain, tiny numbers whose fp16 image shows the low fp32 bits of acc + bias
    neg = (-p1).contiguous()
    es, _ = ops.conv_gemm(x, wk, cout, stats=True, residual=neg, **kw)
    ep = ops.conv_gemm(x, wk, cout, residual=neg, **kw)
    e1 = ops.conv_gemm(x, wk, cout, kh=kh, pad=kh // 2, force_tile=1, bias=b, residual=neg)    # 128x128 4-wave kernel, different MFMA shape
    d = (es.float() - ep.float()).abs()
    print(f"   residue tensors: stats vs plain differ in {int((d > 0).sum())} of {d.numel()} elements, max {float(d.max()):.3e}; "
          f"vs 16x16x32-MFMA kernel: {int(((ep.float() - e1.float()).abs() > 0).sum())} differ, max {float((ep.float() - e1.float()).abs().max()):.3e}; residue rms {float(ep.float().pow(2).mean().sqrt()):.3e}")
    dd = (s1.float() - p1.float()).abs()
    for r, c in (dd > 0).nonzero().tolist()[:4]:
        print(f"   [{r},{c}] stats {float(s1[r, c]):.6f} plain {float(p1[r, c]):.6f}  residue stats {float(es[r, c]):.4e} plain {float(ep[r, c]):.4e}  half-ulp {abs(float(s1[r,c]) - float(p1[r,c])) / 2:.4e}")
