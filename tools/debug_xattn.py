"""Where does the fused cross-attention epilogue differ from projection + fmx_attention_f16?  (development aid)"""
import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import torch.nn.functional as F
import conftest  # noqa
from forge_amd import hipops as ops
from forge_amd.backend.nn.unet import _fold_layernorm
from test_gpu_kernels import rnd, DEV

bu, n, c, nk = (int(v) for v in (sys.argv[1:5] or (4, 1024, 1280, 77)))
m, heads, tp = bu * n, c // 64, 128
o_in = rnd(m, c, seed=190)
w_out, b_out = rnd(c, c, scale=1 / math.sqrt(c), seed=191), rnd(c, seed=192)
h = (rnd(m, c, scale=2.0, seed=193) + 0.7).contiguous()
rs = ops.RowStats(m, c)
ops.linear(o_in, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs, force_tile=7)
gamma, beta = (1 + 0.2 * rnd(c, seed=194)), 0.1 * rnd(c, seed=195)
wq = rnd(c, c, scale=2.0 / math.sqrt(c), seed=196)
wf, cs, bf = _fold_layernorm(wq, None, gamma, beta)
kc = torch.zeros(bu * tp, c, dtype=torch.float16, device=DEV)
vt = torch.zeros(c, bu * tp, dtype=torch.float16, device=DEV)
kv, vv = rnd(bu, nk, c, seed=197), rnd(bu, nk, c, seed=198)
kc.view(bu, tp, c)[:, :nk] = kv
vt.view(c, bu, tp)[:, :, :nk] = vv.permute(2, 0, 1)
scale = 64 ** -0.5
q = ops.conv_gemm(h, wf, c, bias=bf, ln=(rs, cs, 1e-5))
two = ops.attention(q, kc, vt, batch=bu, heads=heads, nq=n, nk=nk, nk_pad=tp, dpad=64, scale=scale,
                    q_bs=n * c, q_rs=c, k_bs=tp * c, k_rs=c, vt_bs=tp, vt_hs=64 * bu * tp, vt_ds=bu * tp)
for rep in range(2):
    fused = ops.conv_gemm(h, wf, c, bias=bf, ln=(rs, cs, 1e-5), xattn=(kc, vt, nk, tp, n, scale))
    torch.cuda.synchronize()
    bad = (fused.float() - two.float()).abs() > 2e-3 + 2e-3 * two.float().abs()
    print("rep", rep, "bad", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(fused).sum()))
    idx = bad.nonzero()
    if len(idx):
        r, cc = idx[:, 0], idx[:, 1]
        print(" rows mod 256 // 16 histogram:", torch.bincount((r % 256) // 16, minlength=16).tolist())
        print(" row mod 16 histogram:", torch.bincount(r % 16, minlength=16).tolist())
        print(" col mod 320 // 16 histogram:", torch.bincount((cc % 320) // 16, minlength=20).tolist())
        print(" col mod 16 histogram:", torch.bincount(cc % 16, minlength=16).tolist())
        print(" tile row histogram (first 16):", torch.bincount(r // 256)[:16].tolist(), " tile col:", torch.bincount(cc // 320).tolist())
        print(" first bad:", idx[:5].tolist(), fused[r[0], cc[0]].item(), two[r[0], cc[0]].item())
