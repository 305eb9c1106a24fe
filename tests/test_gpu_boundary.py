"""GPU parity of the attention-level and op-level drop-in boundaries: `forge_amd.backend.attention.attention_function`
/ `attention_function_single_head_spatial` with the reference's argument shapes (backend/attention.py:324-339, :37-93, :412-422) and the
`ForgeOperations` modules (backend/operations.py:125-330) under `using_forge_operations`, against torch fp32 on the same inputs."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402,F401
from forge_amd.backend import attention as fattn  # noqa: E402
from forge_amd.backend import operations as fops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator("cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, want, rtol, atol, what):
    err = (got.float() - want.float()).abs()
    bad = err > atol + rtol * want.float().abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g}"


def _ref(q, k, v, heads, mask=None):
    b, nq, hd = q.shape
    d = hd // heads
    sp = lambda t: t.float().view(b, -1, heads, d).transpose(1, 2)
    m = mask
    if m is not None and m.dtype == torch.bool and m.dim() == 2 and m.shape == (b, k.shape[1]):
        m = m[:, None, None, :]
    if m is not None and m.dtype != torch.bool:
        m = m.float()
        if m.dim() == 3:
            m = m[:, None]
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=m)
    return o.transpose(1, 2).reshape(b, nq, hd)


@pytest.mark.parametrize("b,heads,nq,nk,d,dtype", [(2, 8, 1024, 1024, 40, torch.float16), (2, 10, 300, 77, 64, torch.float16), (1, 5, 256, 256, 128, torch.float32),
                                                   (2, 4, 130, 154, 160, torch.bfloat16), (1, 20, 1024, 1024, 64, torch.float16)])
def test_attention_function_reference_layout(b, heads, nq, nk, d, dtype):
    q, k, v = rnd(b, nq, heads * d, seed=1, dtype=dtype), rnd(b, nk, heads * d, seed=2, dtype=dtype), rnd(b, nk, heads * d, seed=3, dtype=dtype)
    out = fattn.attention_function(q, k, v, heads)
    assert out.shape == q.shape and out.dtype == q.dtype
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-3
    close(out, _ref(q, k, v, heads), tol, tol, "attention_function")
    # skip_reshape: [B, heads, N, d]
    q4, k4, v4 = (t.view(b, -1, heads, d).transpose(1, 2).contiguous() for t in (q, k, v))
    close(fattn.attention_function(q4, k4, v4, heads, skip_reshape=True), _ref(q, k, v, heads), tol, tol, "attention_function skip_reshape")


def test_attention_function_masks():
    b, heads, nq, nk, d = 2, 4, 200, 77, 64
    q, k, v = rnd(b, nq, heads * d, seed=4), rnd(b, nk, heads * d, seed=5), rnd(b, nk, heads * d, seed=6)
    g = torch.Generator("cpu").manual_seed(7)
    key_mask = (torch.rand(b, nk, generator=g) > 0.3).to(DEV)
    key_mask[:, 0] = True
    close(fattn.attention_function(q, k, v, heads, mask=key_mask), _ref(q, k, v, heads, key_mask), 2e-3, 2e-3, "bool key mask [B, Nk]")
    add = rnd(nq, nk, scale=2.0, seed=8, dtype=torch.float32)
    close(fattn.attention_function(q, k, v, heads, mask=add), _ref(q, k, v, heads, add), 2e-3, 2e-3, "additive mask [Nq, Nk]")
    add3 = rnd(b, nq, nk, scale=2.0, seed=9)
    close(fattn.attention_function(q, k, v, heads, mask=add3), _ref(q, k, v, heads, add3), 2e-3, 2e-3, "additive mask [B, Nq, Nk]")
    full = (torch.rand(b, heads, nq, nk, generator=g) > 0.2).to(DEV)
    full[..., 0] = True
    close(fattn.attention_function(q, k, v, heads, mask=full), _ref(q, k, v, heads, full), 2e-3, 2e-3, "bool mask [B, heads, Nq, Nk]")
    # 1024 queries with a mask: the masked form of the d_head 64 kernel, not the unmasked fast path
    q2 = rnd(b, 1024, heads * d, seed=10)
    km = torch.ones(b, nk, dtype=torch.bool, device=DEV)
    km[:, 40:] = False
    close(fattn.attention_function(q2, k, v, heads, mask=km), _ref(q2, k[:, :40], v[:, :40], heads), 2e-3, 2e-3, "key mask == truncated keys")


@pytest.mark.parametrize("c,hh,ww", [(512, 16, 24), (128, 16, 16), (512, 32, 32)])
def test_attention_single_head_spatial(c, hh, ww):
    b = 2
    q, k, v = rnd(b, c, hh, ww, seed=11), rnd(b, c, hh, ww, seed=12), rnd(b, c, hh, ww, seed=13)
    out = fattn.attention_function_single_head_spatial(q, k, v)
    tok = lambda t: t.float().view(b, 1, c, -1).transpose(2, 3)
    ref = F.scaled_dot_product_attention(tok(q), tok(k), tok(v)).transpose(2, 3).reshape(b, c, hh, ww)
    assert out.shape == q.shape and out.dtype == q.dtype
    close(out, ref, 3e-3, 3e-3, "single-head spatial attention")


def test_attention_function_rejects_host_tensors():
    with pytest.raises(TypeError):
        fattn.attention_function(torch.zeros(1, 8, 64), torch.zeros(1, 8, 64), torch.zeros(1, 8, 64), 1)


def test_forge_operations_modules_match_torch():
    torch.manual_seed(0)
    with fops.using_forge_operations(device=DEV, dtype=torch.float16):
        assert torch.nn.Linear is fops.ForgeOperations.Linear and torch.nn.Conv2d is fops.ForgeOperations.Conv2d
        lin = torch.nn.Linear(320, 1280)
        lin_odd = torch.nn.Linear(77, 40, bias=False)
        c3 = torch.nn.Conv2d(320, 640, 3, padding=1)
        c3s2 = torch.nn.Conv2d(64, 128, 3, stride=2, padding=1)
        c1 = torch.nn.Conv2d(4, 320, 1)
        gn = torch.nn.GroupNorm(32, 320, eps=1e-6)
        ln = torch.nn.LayerNorm(640)
        with pytest.raises(NotImplementedError):
            torch.nn.Conv2d(8, 8, 5)
    assert torch.nn.Linear is not fops.ForgeOperations.Linear, "torch.nn must be restored on exit"
    for m in (lin, lin_odd, c3, c3s2, c1, gn, ln):
        for p_ in m.parameters():
            p_.data.copy_(torch.randn(p_.shape, generator=torch.Generator().manual_seed(p_.numel())).mul_(0.05 if p_.dim() > 1 else 0.3).add_(1.0 if m in (gn, ln) and p_.dim() == 1 and p_ is m.weight else 0.0))
    x = rnd(3, 50, 320, seed=20)
    close(lin(x), F.linear(x.float(), lin.weight.float(), lin.bias.float()), 3e-3, 3e-3, "Linear")
    x32 = rnd(5, 77, seed=21, dtype=torch.float32)
    y = lin_odd(x32)
    assert y.dtype == torch.float32
    close(y, F.linear(x32, lin_odd.weight.float()), 3e-3, 3e-3, "Linear fp32 in/out, K and N not multiples of 8")
    xi = rnd(2, 320, 24, 16, seed=22)
    close(c3(xi), F.conv2d(xi.float(), c3.weight.float(), c3.bias.float(), padding=1), 4e-3, 4e-3, "Conv2d 3x3")
    xs = rnd(2, 64, 17, 20, seed=23)
    close(c3s2(xs), F.conv2d(xs.float(), c3s2.weight.float(), c3s2.bias.float(), stride=2, padding=1), 4e-3, 4e-3, "Conv2d 3x3 stride 2 (odd size)")
    x4 = rnd(2, 4, 16, 16, seed=24)
    close(c1(x4), F.conv2d(x4.float(), c1.weight.float(), c1.bias.float()), 3e-3, 3e-3, "Conv2d 1x1, 4 input channels")
    close(gn(xi), F.group_norm(xi.float(), 32, gn.weight.float(), gn.bias.float(), 1e-6), 3e-3, 3e-3, "GroupNorm")
    xl = rnd(4, 33, 640, scale=2.0, seed=25)
    close(ln(xl), F.layer_norm(xl.float(), (640,), ln.weight.float(), ln.bias.float(), 1e-5), 3e-3, 3e-3, "LayerNorm")
    # an in-place weight edit (LoRA merge) must be picked up
    with torch.no_grad():
        lin.weight.mul_(0.5)
    close(lin(x), F.linear(x.float(), lin.weight.float(), lin.bias.float()), 3e-3, 3e-3, "Linear after an in-place weight update")
    # checkpoints load under the usual names
    sd = c3.state_dict()
    assert set(sd) == {"weight", "bias"} and tuple(sd["weight"].shape) == (640, 320, 3, 3)
