"""GPU (MI355X) per-kernel parity: every C-ABI kernel against a plain PyTorch fp32 reference of the same op
on the same fp16-rounded inputs.  Tolerances are written per test; fp16 outputs carry 2^-11 relative
rounding, accumulation is fp32 in both."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import forge_amd  # noqa: E402
from forge_amd import hipops as ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator("cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).half().to(DEV)


def close(got, want, rtol, atol, what=""):
    got, want = got.float(), want.float()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (ref max {float(want.abs().max()):.4g})"


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (384, 320, 320), (77 * 2, 640, 768), (1000, 132, 64), (4096, 1280, 1280), (130, 4, 2880)])
@pytest.mark.parametrize("tile", [1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15])
def test_linear_plain(m, n, k, tile):
    if tile in (4, 6, 7, 8, 9, 10) and n % 8:
        pytest.skip("256x256 kernel needs nout % 8 == 0 (dispatcher never selects it otherwise)")
    x = rnd(m, k, seed=1)
    w = rnd(n, k, scale=1 / math.sqrt(k), seed=2)
    b = rnd(n, seed=3)
    out = ops.linear(x, w, b, force_tile=tile)
    ref = x.float() @ w.float().t() + b.float()
    close(out, ref, 2e-3, 2e-3, "linear")


def test_narrow_tile_refuses_what_its_address_path_cannot_do():
    x = rnd(2, 8, 8, 128, seed=8)
    wk = rnd(64, 128 * 9, scale=0.03, seed=9)
    with pytest.raises(Exception, match="512x128"):
        ops.conv_gemm(x, wk, 64, kh=3, pad=1, up=(16, 16), force_tile=9)


@pytest.mark.parametrize("n,hh,ww,c,co", [(2, 64, 64, 128, 128), (1, 96, 80, 256, 128), (2, 40, 24, 128, 384)])
def test_narrow_outputs_pick_a_correct_kernel(n, hh, ww, c, co):
    """Layers with 128 output channels at large pixel counts (the VAE decoder's last level): whatever tile the dispatcher picks (the 512x128
    one on a 256-CU part) must agree with conv2d, borders and ragged last tiles included."""
    x = rnd(n, hh, ww, c, seed=25)
    wt = rnd(co, c, 3, 3, scale=1 / math.sqrt(c * 9), seed=26)
    b = rnd(co, seed=27)
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    out = ops.conv_gemm(x, wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous(), co, kh=3, pad=1, bias=b)
    close(out.reshape(ref.shape), ref, 3e-3, 3e-3, "narrow conv")


def test_conv_input_beyond_32bit_offsets_runs_as_image_groups():
    """An activation tensor of more than 3e9 bytes (the VAE decoder's 8 x 1024 x 1024 x 256 level) cannot be addressed with the 8-wave kernels'
    32-bit offsets: fmx_gemm_conv(_stats)_f16 runs it as groups of whole images (csrc/fmx_gemm.hip).  4 x 1024 x 768 x 512 input (3.2 GB), 3x3
    to 64 channels with a per-image row vector and a residual, statistics requested; reference: conv2d per image on the same fp16 data."""
    n, hh, ww, c, co = 4, 1024, 768, 512, 64
    g = torch.Generator(DEV).manual_seed(77)
    x = torch.randn(n, hh, ww, c, generator=g, device=DEV, dtype=torch.float16)
    wt = (torch.randn(co, c, 3, 3, generator=g, device=DEV) / math.sqrt(c * 9)).half()
    b, emb = rnd(co, seed=78), rnd(n, co, seed=79)
    res = torch.randn(n * hh * ww, co, generator=g, device=DEV, dtype=torch.float16)
    out, st = ops.conv_gemm(x, wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous(), co, kh=3, pad=1, bias=b, rowvec=emb, residual=res, stats=True)
    out = out.view(n, hh, ww, co)
    for i in range(n):
        ref = F.conv2d(x[i:i + 1].permute(0, 3, 1, 2).float(), wt.float(), b.float(), padding=1) + emb[i].float()[None, :, None, None]
        ref = ref.permute(0, 2, 3, 1)[0] + res.view(n, hh, ww, co)[i].float()
        close(out[i], ref, 3e-3, 3e-3, f"image {i} of an input beyond 32-bit offsets")
        del ref
    o = out.reshape(n, hh * ww, co).double()
    want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
    torch.testing.assert_close(_partial_to_sums(st, n, co), want, rtol=2e-5, atol=2e-2)


@pytest.mark.parametrize("k", [64 * kt for kt in (1, 2, 3, 4, 5, 6, 7, 9, 20, 37)])
@pytest.mark.parametrize("tile", [11, 12, 13])
def test_ring_tiles_every_pipeline_depth(k, tile):
    """Round 5: the 4-wave tiles on a 4-stage LDS ring (force_tile 11 / 12 / 13 = 128x128 / 128x160 / 128x64; csrc/fmx_gemm.hip `NST`): three K-tiles
    in flight, counted vmcnt waits.  K-tile counts below, at and above the ring depth (the prologue issues min(3, kt) tiles, the tail waits with
    vmcnt(16) -> (8) -> (0)), ragged M / N, residual in place; against fp32 and BIT-IDENTICAL to the 2-stage kernel of the same tile shape (same
    MFMA sequence per output block, same epilogue; 128 x 160: at most one fp16 ulp apart in a handful of elements, see below)."""
    m, n = 300, 328
    x, w, b = rnd(m, k, seed=40), rnd(n, k, scale=1 / math.sqrt(k), seed=41), rnd(n, seed=42)
    res = rnd(m, n, seed=43)
    ref = x.float() @ w.float().t() + b.float() + res.float()
    got = res.clone()
    ops.linear(x, w, b, residual=got, out=got, ld_out=n, force_tile=tile)
    close(got, ref, 2e-3, 2e-3, f"ring tile {tile}, {k // 64} K-tiles")
    old = {11: 1, 12: 5, 13: 2}[tile]
    same = res.clone()
    ops.linear(x, w, b, residual=same, out=same, ld_out=n, force_tile=old)
    if tile == 12:
        # the 2-stage 128 x 160 kernel is compiled WITHOUT the output-statistics code (its registers: two workgroups per CU), the ring one with it: the
        # compiler contracts the epilogue's multiply-adds differently around it, so the two may land on neighbouring fp16 values in a few elements
        d = (got.float() - same.float()).abs()
        ulp = torch.maximum(same.float().abs(), torch.tensor(2.0 ** -14, device=DEV)).log2().floor().exp2() * 2.0 ** -10
        assert bool((d <= ulp).all()) and int((d > 0).sum()) <= max(2, d.numel() // 1000), f"{int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.4g}"
    else:
        assert torch.equal(got, same), f"ring tile {tile} differs from the 2-stage tile {old}"


@pytest.mark.parametrize("splits", [2, 3, 5])
@pytest.mark.parametrize("tile", [1, 2, 11, 13])
def test_split_k_matches_and_is_deterministic(splits, tile, monkeypatch):
    """Split-K of the 4-wave kernels (csrc/fmx_gemm.hip; used for small batches): S workgroups per output tile over contiguous K ranges, partial
    accumulators through the workspace, the last arrival sums them in slot order and runs the fused epilogue.  Forced through the A/B knob on
    both tile shapes: linear with bias + in-place residual and ragged M / N, GEGLU, a 3x3 convolution with a per-image row vector; equal to the
    fp32 reference within the usual tolerance, bit-identical run to run (no floating-point atomics), and the arrival counters are left zero
    (a second launch works)."""
    monkeypatch.setenv("FMX_ALLOW_KNOBS", "1")
    monkeypatch.setenv("FMX_GEMM_SPLITK", str(splits))
    m, n, k = 1000, 328, 1280
    x, w, b = rnd(m, k, seed=30), rnd(n, k, scale=1 / math.sqrt(k), seed=31), rnd(n, seed=32)
    res = rnd(m, n, seed=33)
    ref = x.float() @ w.float().t() + b.float() + res.float()
    outs = []
    for _ in range(3):
        r2 = res.clone()
        ops.linear(x, w, b, residual=r2, out=r2, ld_out=n, force_tile=tile)
        outs.append(r2)
    close(outs[0], ref, 2e-3, 2e-3, f"split-K x{splits} linear")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    inner = 320
    wi, bi = rnd(2 * inner, k, scale=1 / math.sqrt(k), seed=34), rnd(2 * inner, seed=35)
    wint, bint = ops.geglu_interleave(wi, bi)
    out = ops.conv_gemm(x, wint, 2 * inner, bias=bint, act=ops.ACT_GEGLU, force_tile=tile)
    hcat = x.float() @ wi.float().t() + bi.float()
    close(out, hcat[:, :inner] * F.gelu(hcat[:, inner:]), 3e-3, 3e-3, f"split-K x{splits} GEGLU")
    nb, hh, ww, c, co = 2, 12, 20, 128, 192
    xc = rnd(nb, hh, ww, c, seed=36)
    wt = rnd(co, c, 3, 3, scale=1 / math.sqrt(c * 9), seed=37)
    bc, emb = rnd(co, seed=38), rnd(nb, co, seed=39)
    refc = (F.conv2d(xc.permute(0, 3, 1, 2).float(), wt.float(), bc.float(), padding=1) + emb.float()[:, :, None, None]).permute(0, 2, 3, 1)
    oc = ops.conv_gemm(xc, wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous(), co, kh=3, pad=1, bias=bc, rowvec=emb, force_tile=tile)
    close(oc.reshape(refc.shape), refc, 3e-3, 3e-3, f"split-K x{splits} conv")
    monkeypatch.setenv("FMX_ALLOW_KNOBS", "1")
    monkeypatch.setenv("FMX_GEMM_SPLITK", "0")
    plain = res.clone()
    ops.linear(x, w, b, residual=plain, out=plain, ld_out=n, force_tile=tile)
    close(outs[0], plain.float(), 1e-3, 1e-3, "split-K vs one workgroup per tile")


def test_split_k_hand_over_under_load_is_stable(monkeypatch):
    """The hand-over reuses ONE workspace and its arrival counters launch after launch.  Forty launches back to back with changing split
    factors, tile shapes and problem sizes (so slot addresses are re-used by other tiles, other XCDs, other split counts while earlier kernels
    are still draining), every output word compared with the first time that configuration ran, and with the unsplit kernel at the end."""
    cases = []
    for i, (m, n, k) in enumerate(((1000, 328, 1280), (2048, 1280, 2560), (640, 640, 5120), (3000, 136, 1920))):
        x, w, b = rnd(m, k, seed=50 + i), rnd(n, k, scale=1 / math.sqrt(k), seed=60 + i), rnd(n, seed=70 + i)
        cases.append((x, w, b))
    first = {}
    for it in range(40):
        ci, splits, tile = it % 4, (2, 3, 4, 7)[(it // 4) % 4], 1 + (it // 2) % 2
        monkeypatch.setenv("FMX_ALLOW_KNOBS", "1")
        monkeypatch.setenv("FMX_GEMM_SPLITK", str(splits))
        x, w, b = cases[ci]
        out = ops.linear(x, w, b, force_tile=tile)
        key = (ci, splits, tile)
        if key in first:
            assert torch.equal(out, first[key]), f"case {key} changed on launch {it}"
        else:
            first[key] = out.clone()
    monkeypatch.setenv("FMX_ALLOW_KNOBS", "1")
    monkeypatch.setenv("FMX_GEMM_SPLITK", "0")
    for (ci, splits, tile), got in first.items():
        x, w, b = cases[ci]
        close(got, ops.linear(x, w, b, force_tile=tile).float(), 1e-3, 1e-3, f"split-K x{splits} tile {tile} case {ci} vs unsplit")


@pytest.mark.parametrize("m,c,nq", [(16384, 1280, 1280), (16384 - 100, 1280, 640), (65536, 640, 640)])
def test_layernorm_folded_into_the_gemms_around_it(m, c, nq):
    """norm2 / norm3 of a transformer block without a LayerNorm kernel (csrc/fmx_gemm256p.hip `LN`): the residual-adding projection that
    writes h also leaves per-row {sum, sum of squares} partials of the fp16 values it stored (fmx_gemm_linear_rowstats_f16); the projection
    behind the LayerNorm runs on h itself with gamma-scaled weights and applies mean / rstd in its epilogue (plain and GEGLU).  Against
    F.layer_norm + linear in fp32 on the same stored h; the partials against sums over the stored tensor; a ragged last row tile included."""
    from forge_amd.backend.nn.unet import _fold_layernorm
    k_in = c
    o_in = rnd(m, k_in, seed=90)
    w_out, b_out = rnd(c, k_in, scale=1 / math.sqrt(k_in), seed=91), rnd(c, seed=92)
    h = (rnd(m, c, scale=2.0, seed=93) + 0.7).contiguous()          # the residual stream, with a non-zero mean
    h0 = h.clone()
    rs = ops.RowStats(m, c)
    ops.linear(o_in, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs)
    close(h, o_in.float() @ w_out.float().t() + b_out.float() + h0.float(), 2e-3, 2e-3, "producer output")
    assert rs.parts == 2 * (c // 320), "expected the 256x320 tile (and its row statistics) for this shape"
    got = rs.partial.view(m, -1, 2)[:, :rs.parts].double().sum(1)
    hf = h.double()
    torch.testing.assert_close(got, torch.stack([hf.sum(1), (hf * hf).sum(1)], -1), rtol=1e-5, atol=1e-2)
    gamma, beta = (1 + 0.2 * rnd(c, seed=94)), 0.1 * rnd(c, seed=95)
    ln = F.layer_norm(h.float(), (c,), gamma.float(), beta.float(), 1e-5)
    wq = rnd(nq, c, scale=1 / math.sqrt(c), seed=96)
    wf, cs, bf = _fold_layernorm(wq, None, gamma, beta)
    q = ops.conv_gemm(h, wf, nq, bias=bf, ln=(rs, cs, 1e-5))
    close(q, ln @ wq.float().t(), 4e-3, 4e-3, "LayerNorm folded into a plain projection")
    inner = nq
    wg, bg = rnd(2 * inner, c, scale=1 / math.sqrt(c), seed=97), rnd(2 * inner, seed=98)
    wgi, bgi = ops.geglu_interleave(wg, bg)
    wf2, cs2, bf2 = _fold_layernorm(wgi, bgi, gamma, beta)
    g = ops.conv_gemm(h, wf2, 2 * inner, bias=bf2, act=ops.ACT_GEGLU, ln=(rs, cs2, 1e-5))
    hc = ln @ wg.float().t() + bg.float()
    close(g, hc[:, :inner] * F.gelu(hc[:, inner:]), 5e-3, 5e-3, "LayerNorm folded into the GEGLU projection")
    # the kernel it replaces, for scale: LayerNorm output rounded to fp16, then the same GEMM
    n_kernel = ops.layernorm(h, gamma, beta)
    q_old = ops.linear(n_kernel, wq)
    err_new = float((q.float() - ln @ wq.float().t()).abs().max()), float((q_old.float() - ln @ wq.float().t()).abs().max())
    assert err_new[0] <= 2.0 * err_new[1] + 1e-3, err_new


@pytest.mark.parametrize("bu,n,c,nk", [(4, 1024, 1280, 77), (2, 4096, 640, 77), (4, 256, 320, 80), (2, 512, 640, 33)])
def test_cross_attention_as_the_epilogue_of_the_query_projection(bu, n, c, nk):
    """attn2 of a BasicTransformerBlock without a query tensor (csrc/fmx_gemm256p.hip `XA`, fmx.h xa_*; reference backend/nn/unet.py:145-155, 254-267):
    the LayerNorm-consumer projection `to_q(norm2(h))` keeps its 256 x 320 tile of Q in the accumulators and runs softmax(q k^T / 8) v against the
    cached K / V^T of the text context in its own epilogue.  Against (a) the two launches it replaces -- the same projection storing Q, then
    fmx_attention_f16 -- whose rounding sites it shares (Q rounded to fp16, pre-scaled, P rounded to fp16): agreement at the level of fp32 summation
    order, and (b) LayerNorm + Linear + softmax attention in fp32 on the stored h."""
    from forge_amd.backend.nn.unet import _fold_layernorm
    m, heads, tp = bu * n, c // 64, 128
    o_in = rnd(m, c, seed=190)
    w_out, b_out = rnd(c, c, scale=1 / math.sqrt(c), seed=191), rnd(c, seed=192)
    h = (rnd(m, c, scale=2.0, seed=193) + 0.7).contiguous()
    rs = ops.RowStats(m, c)
    ops.linear(o_in, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs, force_tile=7)   # (the 256 x 320 tile whatever the dispatcher would pick at this M)
    assert rs.parts == 2 * (c // 320)
    gamma, beta = (1 + 0.2 * rnd(c, seed=194)), 0.1 * rnd(c, seed=195)
    wq = rnd(c, c, scale=2.0 / math.sqrt(c), seed=196)          # (scores with a spread of a few units: a peaked softmax)
    wf, cs, bf = _fold_layernorm(wq, None, gamma, beta)
    kc = torch.zeros(bu * tp, c, dtype=torch.float16, device=DEV)
    vt = torch.zeros(c, bu * tp, dtype=torch.float16, device=DEV)
    kv = rnd(bu, nk, c, seed=197)
    vv = rnd(bu, nk, c, seed=198)
    kc.view(bu, tp, c)[:, :nk] = kv
    vt.view(c, bu, tp)[:, :, :nk] = vv.permute(2, 0, 1)
    scale = 64 ** -0.5
    q = ops.conv_gemm(h, wf, c, bias=bf, ln=(rs, cs, 1e-5))
    two = ops.attention(q, kc, vt, batch=bu, heads=heads, nq=n, nk=nk, nk_pad=tp, dpad=64, scale=scale,
                        q_bs=n * c, q_rs=c, k_bs=tp * c, k_rs=c, vt_bs=tp, vt_hs=64 * bu * tp, vt_ds=bu * tp)
    fused = ops.conv_gemm(h, wf, c, bias=bf, ln=(rs, cs, 1e-5), xattn=(kc, vt, nk, tp, n, scale))
    assert fused.shape == two.shape == (m, c)
    ln = F.layer_norm(h.float(), (c,), gamma.float(), beta.float(), 1e-5)
    qr = (ln @ wq.float().t()).view(bu, n, heads, 64).permute(0, 2, 1, 3)
    kr = kv.float().view(bu, nk, heads, 64).permute(0, 2, 1, 3)
    vr = vv.float().view(bu, nk, heads, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(qr @ kr.transpose(-1, -2) * scale, -1) @ vr).permute(0, 2, 1, 3).reshape(m, c)
    e_two, e_fused = float((two.float() - ref).abs().max()), float((fused.float() - ref).abs().max())
    d = float((fused.float() - two.float()).abs().max())
    print(f"cross-attention epilogue bu={bu} n={n} c={c} nk={nk}: fused vs fp32 {e_fused:.2e}, two launches vs fp32 {e_two:.2e}, fused vs two launches {d:.2e} "
          f"(|ref| max {float(ref.abs().max()):.2f})")
    close(fused, ref, 4e-3, 4e-3, "fused cross-attention vs fp32")
    close(fused, two.float(), 2e-3, 2e-3, "fused cross-attention vs projection + fmx_attention_f16")
    assert e_fused <= 1.5 * e_two + 1e-3


@pytest.mark.parametrize("m,c,hd", [(16384, 1280, 1280), (65536, 640, 640), (16384 - 64, 1280, 1280)])
def test_layernorm_folded_into_the_operand_swapped_vt_projection(m, c, hd):
    """norm1 without a LayerNorm kernel (round 3): the projection that writes h WITHOUT a residual (proj_in) leaves the row statistics too; the
    V^T projection runs as W' h^T on the un-normalised h (320x256 tile, `LN = 3`): the LayerNorm rows are its output COLUMNS, so the epilogue
    takes {rstd, -mean rstd} per column (fmx_layernorm_rowstats_finalize) and {colsum, W beta} per row.  Against F.layer_norm + linear in fp32 on
    the stored h, transposed; the q|k projection of the same block through the ordinary fold from the same statistics."""
    from forge_amd.backend.nn.unet import _fold_layernorm
    x_in = rnd(m, c, seed=110)
    w_in, b_in = rnd(c, c, scale=1 / math.sqrt(c), seed=111), (rnd(c, seed=112) + 0.5).contiguous()
    rs = ops.RowStats(m, c)
    h = ops.linear(x_in, w_in, b_in, row_stats=rs)                      # producer without a residual
    close(h, x_in.float() @ w_in.float().t() + b_in.float(), 2e-3, 2e-3, "producer output (no residual)")
    assert rs.parts == 2 * (c // 320)
    hf = h.double()
    torch.testing.assert_close(rs.partial.view(m, -1, 2)[:, :rs.parts].double().sum(1), torch.stack([hf.sum(1), (hf * hf).sum(1)], -1), rtol=1e-5, atol=1e-2)
    gamma, beta = (1 + 0.2 * rnd(c, seed=113)), 0.1 * rnd(c, seed=114)
    ln = F.layer_norm(h.float(), (c,), gamma.float(), beta.float(), 1e-5)
    ab = ops.ln_rowstats_finalize(rs, c, 1e-5)
    mean, var = h.float().mean(1), h.float().var(1, unbiased=False)
    torch.testing.assert_close(ab[:, 0], torch.rsqrt(var + 1e-5), rtol=2e-4, atol=1e-6)
    torch.testing.assert_close(ab[:, 1], -mean * torch.rsqrt(var + 1e-5), rtol=2e-4, atol=2e-4)
    wv = rnd(hd, c, scale=1 / math.sqrt(c), seed=115)
    wvf = (wv.float() * gamma.float()[None, :]).half().contiguous()
    cb = torch.stack([wvf.float().sum(1), wv.float() @ beta.float()], 1).contiguous()
    vt = ops.conv_gemm(wvf, h, m, ln_swapped=(ab, cb))
    assert tuple(vt.shape) == (hd, m)
    close(vt, (ln @ wv.float().t()).t(), 4e-3, 4e-3, "LayerNorm folded into the operand-swapped V^T projection")
    # the kernels it replaces, for scale
    vt_old = ops.conv_gemm(wv, ops.layernorm(h, gamma, beta), m)
    ref = (ln @ wv.float().t()).t()
    e_new, e_old = float((vt.float() - ref).abs().max()), float((vt_old.float() - ref).abs().max())
    assert e_new <= 2.0 * e_old + 1e-3, (e_new, e_old)
    wqk = rnd(2 * hd, c, scale=1 / math.sqrt(c), seed=116)
    wf, cs, bf = _fold_layernorm(wqk, None, gamma, beta)
    ab2 = torch.zeros(m, 2, dtype=torch.float32, device=DEV)
    close(ops.conv_gemm(h, wf, 2 * hd, bias=bf, ln=(rs, cs, 1e-5), ln_ab_out=ab2), ln @ wqk.float().t(), 4e-3, 4e-3, "q|k projection from the same statistics")
    torch.testing.assert_close(ab2, ab, rtol=2e-6, atol=1e-6)   # the pairs the q|k GEMM leaves for the V^T GEMM = the finalize kernel's


def test_gemm_32x32x16_loop_behind_its_knob():
    """The K loops run on v_mfma_f32_16x16x32 since round 3; the 32x32x16 loop stays in the library behind FMX_GEMM_MFMA=32 (the A/B of
    profiles/r08r).  The knob is read once per process, so the old loop is exercised in a child process: plain, residual, GEGLU and a 3x3
    convolution with GroupNorm statistics at shapes that take the 256x320 tile, against torch fp32 -- and against this process's 16x16x32
    results (same fp32 accumulation in another order: equal within fp16 rounding of the output)."""
    import subprocess
    import sys
    code = r"""
import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
import forge_amd
from forge_amd import hipops as ops, _lib
g = torch.Generator('cuda').manual_seed(5)
r = lambda *s, scale=1.0: (torch.randn(*s, device='cuda', generator=g) * scale).half()
x, w, b, res = r(4096, 1280), r(1280, 1280, scale=1280 ** -0.5), r(1280), r(4096, 1280)
out = {}
out['plain'] = ops.linear(x, w, b).float().cpu()
out['res'] = ops.linear(x, w, b, residual=res).float().cpu()
wg, bg = r(2560, 1280, scale=1280 ** -0.5), r(2560)
wgi, bgi = ops.geglu_interleave(wg, bg)
out['geglu'] = ops.conv_gemm(x, wgi, 2560, bias=bgi, act=ops.ACT_GEGLU).float().cpu()
xc, wc = r(4, 32, 32, 640), r(640, 9 * 640, scale=(9 * 640) ** -0.5)
y, st = ops.conv_gemm(xc, wc, 640, kh=3, pad=1, bias=r(640), stats=True)
out['conv'] = y.float().cpu()
ref = {'plain': x.float() @ w.float().t() + b.float(), 'res': x.float() @ w.float().t() + b.float() + res.float()}
for k, v in ref.items():
    assert float((out[k] - v.cpu()).abs().max()) < 2e-2, k
out['knobs_active'], out['knobs_ignored'] = _lib.active_knobs(False), _lib.active_knobs(True)
torch.save(out, sys.argv[1])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    outs = {}
    # "stray": the knob set WITHOUT FMX_ALLOW_KNOBS=1 -- a production process must keep its default kernels
    for mf in ("32", "16", "stray"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            env = {k: v for k, v in os.environ.items() if not k.startswith("FMX_")}
            env.update({"FMX_GEMM_MFMA": "32"} if mf == "stray" else {"FMX_GEMM_MFMA": mf, "FMX_ALLOW_KNOBS": "1"})
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=env, timeout=300)
            outs[mf] = torch.load(f.name)
    kn = {mf: (outs[mf].pop("knobs_active"), outs[mf].pop("knobs_ignored")) for mf in outs}
    assert kn["32"][0].get("FMX_GEMM_MFMA") == "32" and not kn["32"][1]
    assert not kn["stray"][0] and kn["stray"][1].get("FMX_GEMM_MFMA") == "32", kn["stray"]
    for k in outs["32"]:
        close(outs["32"][k], outs["16"][k], 2e-3, 2e-3, f"32x32x16 loop vs 16x16x32 loop: {k}")
        assert torch.equal(outs["stray"][k], outs["16"][k]), f"a knob without FMX_ALLOW_KNOBS=1 changed the kernel: {k}"
    # (the 32x32x16 and 16x16x32 loops accumulate each output over K in the same order in fp32 and may well agree bit for bit: that the allowed knob
    #  took effect is what fmx_active_knobs() says above, not a difference in the results)


def test_cross_tile_prefetch_matches_the_per_tile_prologue_bit_for_bit():
    """Round 3: a persistent workgroup of the linear 256x320 kernels fetches its NEXT output tile's first K-tile under the last K-tile of the
    current one (csrc/fmx_gemm256p.hip `XT`); FMX_GEMM_XTILE=0 (read once per process) restores the per-tile prologue.  Same arithmetic in the
    same order, so every output word must be identical: launches with 2 - 4 tiles per workgroup, odd and even K-tile counts (the stage a tile
    starts on alternates when the count is odd), the two- and three-K-tile minimum, one K-tile (prefetch off by construction), ragged last row
    and column tiles; plain, in-place residual, GEGLU with and without residual, and the LayerNorm producer / consumer / consumer-GEGLU chain."""
    import subprocess
    import sys
    import tempfile
    code = r"""
import sys, math, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
import forge_amd
from forge_amd import hipops as ops
from forge_amd.backend.nn.unet import _fold_layernorm
g = torch.Generator('cuda').manual_seed(11)
r = lambda *s, scale=1.0: (torch.randn(*s, device='cuda', generator=g) * scale).half()
out = {}
def check(name, got, ref, tol):
    err = float((got.float() - ref).abs().max())
    assert err < tol, (name, err)
    out[name] = got.cpu()
for (m, n, k) in ((33000, 1920, 960), (20000, 2560, 128), (20000, 2560, 192), (20000, 2560, 64), (30000, 1288, 1280)):
    x, w, b = r(m, k), r(n, k, scale=k ** -0.5), r(n)
    ref = x.float() @ w.float().t() + b.float()
    check(f'plain {m}x{n}x{k}', ops.linear(x, w, b, force_tile=7), ref, 2e-2)
    h = r(m, n)
    h0 = h.clone()
    ops.linear(x, w, b, residual=h, out=h, ld_out=n, force_tile=7)
    check(f'residual in place {m}x{n}x{k}', h, ref + h0.float(), 2e-2)
    if n %% 64 == 0:
        wgi, bgi = ops.geglu_interleave(w, b)
        gref = ref[:, :n // 2] * F.gelu(ref[:, n // 2:])
        check(f'geglu {m}x{n}x{k}', ops.conv_gemm(x, wgi, n, bias=bgi, act=ops.ACT_GEGLU, force_tile=7), gref, 3e-2)
        res = r(m, n // 2)
        check(f'geglu + residual {m}x{n}x{k}', ops.conv_gemm(x, wgi, n, bias=bgi, act=ops.ACT_GEGLU, residual=res, force_tile=7), gref + res.float(), 3e-2)
for (m, c) in ((40000 - 60, 1280), (70000, 640)):
    o_in, w_out, b_out = r(m, c), r(c, c, scale=c ** -0.5), r(c)
    h = (r(m, c, scale=2.0) + 0.7).contiguous()
    h0 = h.clone()
    rs = ops.RowStats(m, c)
    ops.linear(o_in, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs, force_tile=7)
    assert rs.parts == 2 * (c // 320)
    check(f'LN producer {m}x{c}', h, o_in.float() @ w_out.float().t() + b_out.float() + h0.float(), 2e-2)
    out[f'LN producer partials {m}x{c}'] = rs.partial.view(m, -1, 2)[:, :rs.parts].clone().cpu()
    gamma, beta = (1 + 0.2 * r(c)), 0.1 * r(c)
    ln = F.layer_norm(h.float(), (c,), gamma.float(), beta.float(), 1e-5)
    wq = r(2 * c, c, scale=c ** -0.5)
    wf, cs, bf = _fold_layernorm(wq, None, gamma, beta)
    ab = torch.zeros(m, 2, dtype=torch.float32, device='cuda')
    check(f'LN consumer {m}x{c}', ops.conv_gemm(h, wf, 2 * c, bias=bf, ln=(rs, cs, 1e-5), ln_ab_out=ab), ln @ wq.float().t(), 3e-2)
    out[f'LN consumer pairs {m}x{c}'] = ab.cpu()
    wg, bg = r(4 * c, c, scale=c ** -0.5), r(4 * c)
    wgi, bgi = ops.geglu_interleave(wg, bg)
    wf2, cs2, bf2 = _fold_layernorm(wgi, bgi, gamma, beta)
    hc = ln @ wg.float().t() + bg.float()
    check(f'LN consumer GEGLU {m}x{c}', ops.conv_gemm(h, wf2, 4 * c, bias=bf2, act=ops.ACT_GEGLU, ln=(rs, cs2, 1e-5)), hc[:, :2 * c] * F.gelu(hc[:, 2 * c:]), 4e-2)
    del ln, hc
torch.save(out, sys.argv[1])
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for knob in ("0", "1"):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            subprocess.run([sys.executable, "-c", code, f.name], check=True, env=dict(os.environ, FMX_GEMM_XTILE=knob, FMX_ALLOW_KNOBS="1"), timeout=600)
            outs[knob] = torch.load(f.name)
    assert outs["0"].keys() == outs["1"].keys() and len(outs["0"]) == 28
    for k in outs["0"]:
        assert torch.equal(outs["0"][k], outs["1"][k]), f"cross-tile prefetch changed the result of: {k}"


def test_layernorm_fold_is_declined_for_small_problems():
    """Below the sizes at which the dispatcher uses the 256x320 tile the producer emits nothing (parts == 0) and the caller keeps its LayerNorm."""
    m, c = 512, 640
    x, w, b, h = rnd(m, c, seed=99), rnd(c, c, scale=0.04, seed=100), rnd(c, seed=101), rnd(m, c, seed=102)
    h0 = h.clone()
    rs = ops.RowStats(m, c)
    ops.linear(x, w, b, residual=h, out=h, ld_out=c, row_stats=rs)
    assert rs.parts == 0
    close(h, x.float() @ w.float().t() + b.float() + h0.float(), 2e-3, 2e-3, "producer output without statistics")


def test_small_batch_shapes_pick_a_correct_kernel():
    """The interactive-batch shapes of the SDXL forward (UNet batch 2: M = 2048 rows), through the dispatcher's own choice (split-K where its cost
    model says so) against the fp32 reference."""
    for m, n, k in ((2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (1280, 2048, 1280), (8192, 640, 640)):
        x, w, b = rnd(m, k, seed=40), rnd(n, k, scale=1 / math.sqrt(k), seed=41), rnd(n, seed=42)
        close(ops.linear(x, w, b), x.float() @ w.float().t() + b.float(), 2e-3, 2e-3, f"linear {m}x{n}x{k}")


def test_linear_asymmetric_identity():
    # A = I, asymmetric B: catches swapped row/col in the MFMA C layout
    m = n = k = 128
    x = torch.eye(m, dtype=torch.float16, device=DEV)
    w = (torch.arange(n * k, device=DEV).reshape(n, k) % 251).half() / 16
    out = ops.linear(x, w)
    torch.testing.assert_close(out.float(), w.float().t(), rtol=0, atol=0)


def test_linear_epilogues():
    m, n, k = 512, 640, 320
    x, w, b = rnd(m, k, seed=4), rnd(n, k, scale=0.05, seed=5), rnd(n, seed=6)
    res = rnd(m, n, seed=7)
    out = ops.linear(x, w, b, residual=res, alpha=0.5)
    ref = 0.5 * (x.float() @ w.float().t()) + b.float() + res.float()
    close(out, ref, 2e-3, 2e-3, "bias+residual+alpha")
    # in-place residual (out aliases residual), as the transformer blocks use it
    r2 = res.clone()
    ops.linear(x, w, b, residual=r2, out=r2, ld_out=n)
    close(r2, x.float() @ w.float().t() + b.float() + res.float(), 2e-3, 2e-3, "in-place residual")


def test_linear_geglu():
    m, k, inner = 384, 320, 1280
    x = rnd(m, k, seed=8)
    w = rnd(2 * inner, k, scale=1 / math.sqrt(k), seed=9)
    b = rnd(2 * inner, scale=0.1, seed=10)
    wi, bi = ops.geglu_interleave(w, b)
    out = ops.conv_gemm(x, wi, 2 * inner, bias=bi, act=ops.ACT_GEGLU)
    h = x.float() @ w.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    close(out, a * F.gelu(g), 3e-3, 3e-3, "geglu")
    assert out.shape == (m, inner)


def test_linear_two_source_and_vt():
    m, k0, k1, n = 256, 128, 192, 256
    x0, x1 = rnd(m, k0, seed=11), rnd(m, k1, seed=12)
    w = rnd(n, k0 + k1, scale=0.06, seed=13)
    out = ops.conv_gemm(x0, w, n, x1=x1)
    ref = torch.cat([x0, x1], 1).float() @ w.float().t()
    close(out, ref, 2e-3, 2e-3, "two-source linear")
    # V^T = Wv X^T via operand swap (how the attention V operand is produced)
    vt = ops.conv_gemm(w, torch.cat([x0, x1], 1).contiguous(), m)
    close(vt, ref.t(), 2e-3, 2e-3, "swapped operands")


@pytest.mark.parametrize("cfg", [
    dict(n=2, h=16, w=16, c=64, co=64, kh=3, stride=1, pad=1),
    dict(n=2, h=32, w=32, c=320, co=320, kh=3, stride=1, pad=1),
    dict(n=3, h=16, w=16, c=128, co=192, kh=3, stride=2, pad=1),
    dict(n=2, h=9, w=7, c=64, co=64, kh=3, stride=2, pad=1),
    dict(n=2, h=8, w=8, c=128, co=64, kh=3, stride=1, pad=1, up=(16, 16)),
    dict(n=1, h=5, w=6, c=64, co=64, kh=3, stride=1, pad=1, up=(9, 11)),
    dict(n=2, h=16, w=16, c=128, co=4, kh=3, stride=1, pad=1),
    dict(n=2, h=12, w=12, c=64, co=128, kh=1, stride=1, pad=0),
])
@pytest.mark.parametrize("tile", [0, 3, 5, 6, 7, 8, 9, 11, 12, 13, 14, 15])
def test_conv(cfg, tile):
    if tile in (4, 6, 7, 8, 9) and cfg["co"] % 8:
        pytest.skip("256x256 kernel needs nout % 8 == 0")
    if tile == 9 and cfg.get("up"):
        pytest.skip("the 512x128 tile's address path takes no upsample-on-load (the dispatcher never sends one there)")
    n, h, w, c, co, kh = cfg["n"], cfg["h"], cfg["w"], cfg["c"], cfg["co"], cfg["kh"]
    x = rnd(n, h, w, c, seed=20)
    wt = rnd(co, c, kh, kh, scale=1 / math.sqrt(c * kh * kh), seed=21)  # torch layout [Cout,Cin,kh,kw]
    b = rnd(co, seed=22)
    emb = rnd(n, co, seed=23)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    up = cfg.get("up")
    xin = x.permute(0, 3, 1, 2).float()
    if up:
        xin = F.interpolate(xin, size=list(up), mode="nearest")
    ref = F.conv2d(xin, wt.float(), b.float(), stride=cfg["stride"], padding=cfg["pad"]) + emb.float()[:, :, None, None]
    ref = ref.permute(0, 2, 3, 1)
    res = rnd(*ref.shape, seed=24)
    out = ops.conv_gemm(x, wk, co, kh=kh, stride=cfg["stride"], pad=cfg["pad"], up=up, bias=b, rowvec=emb,
                        residual=res.reshape(-1, co), force_tile=tile)
    close(out.reshape(ref.shape), ref + res.float(), 3e-3, 3e-3, f"conv {cfg}")


# ---- 256-row kernels (force_tile = 6 / 7 / 8: 256x256, 256x320, 320x256 tiles of the software-pipelined single-barrier kernel): epilogues, GEGLU, two sources, ragged M / N, long K, tail handling ----
@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (512, 512, 128), (300, 264, 192), (1000, 1280, 1280), (4096, 640, 2560), (2048, 320, 320)])
@pytest.mark.parametrize("t256", [6, 7, 8, 10])
def test_gemm256_epilogues(m, n, k, t256):
    x, w, b = rnd(m, k, seed=100), rnd(n, k, scale=1 / math.sqrt(k), seed=101), rnd(n, seed=102)
    res = rnd(m, n, seed=103)
    out = ops.linear(x, w, b, residual=res, alpha=0.5, force_tile=t256)
    close(out, 0.5 * (x.float() @ w.float().t()) + b.float() + res.float(), 2e-3, 2e-3, "gemm256 bias+residual+alpha")
    r2 = res.clone()
    ops.linear(x, w, None, residual=r2, out=r2, ld_out=n, force_tile=t256)
    close(r2, x.float() @ w.float().t() + res.float(), 2e-3, 2e-3, "gemm256 in-place residual")


@pytest.mark.parametrize("t256", [6, 7, 8, 10])
def test_gemm256_identity_asymmetric(t256):
    m = n = k = 512
    x = torch.eye(m, dtype=torch.float16, device=DEV)
    w = (torch.arange(n * k, device=DEV).reshape(n, k) % 251).half() / 16
    out = ops.linear(x, w, force_tile=t256)
    torch.testing.assert_close(out.float(), w.float().t(), rtol=0, atol=0)


@pytest.mark.parametrize("m,k,inner", [(512, 320, 1280), (300, 640, 2560 + 16)])
@pytest.mark.parametrize("t256", [6, 7, 8, 10])
def test_gemm256_geglu(m, k, inner, t256):
    x = rnd(m, k, seed=110)
    w = rnd(2 * inner, k, scale=1 / math.sqrt(k), seed=111)
    b = rnd(2 * inner, scale=0.1, seed=112)
    wi, bi = ops.geglu_interleave(w, b)
    out = ops.conv_gemm(x, wi, 2 * inner, bias=bi, act=ops.ACT_GEGLU, force_tile=t256)
    h = x.float() @ w.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    close(out, a * F.gelu(g), 3e-3, 3e-3, "gemm256 geglu")


def test_two_workgroup_tile_refuses_what_it_does_not_do():
    """force_tile 10 = the 256x160 tile with two 4-wave workgroups per CU (csrc/fmx_gemm4w.hip): plain linear GEMMs only."""
    x = rnd(2, 16, 16, 128, seed=8)
    wk = rnd(160, 128 * 9, scale=0.03, seed=9)
    with pytest.raises(Exception, match="256x160"):
        ops.conv_gemm(x, wk, 160, kh=3, pad=1, force_tile=10)


@pytest.mark.parametrize("m,c,nq", [(16384, 1280, 1280), (16384 - 100, 1280, 640), (8192, 640, 640), (4096, 320, 960)])
def test_two_workgroup_tile_layernorm_folds_and_interoperates_with_the_256x320_kernel(m, c, nq):
    """The 256x160 two-workgroups-per-CU kernel as LayerNorm PRODUCER (per-row sums of the stored output, one entry per 160 columns) and CONSUMER (plain
    and GEGLU), and mixed with the 256x320 kernel on the same statistics array -- either kernel may produce what the other consumes (the dispatcher
    picks per shape)."""
    from forge_amd.backend.nn.unet import _fold_layernorm
    o_in = rnd(m, c, seed=190)
    w_out, b_out = rnd(c, c, scale=1 / math.sqrt(c), seed=191), rnd(c, seed=192)
    h0 = (rnd(m, c, scale=2.0, seed=193) + 0.7).contiguous()
    gamma, beta = (1 + 0.2 * rnd(c, seed=194)), 0.1 * rnd(c, seed=195)
    wq = rnd(nq, c, scale=1 / math.sqrt(c), seed=196)
    wf, cs, bf = _fold_layernorm(wq, None, gamma, beta)
    wg, bg = rnd(2 * nq, c, scale=1 / math.sqrt(c), seed=197), rnd(2 * nq, seed=198)
    wgi, bgi = ops.geglu_interleave(wg, bg)
    wf2, cs2, bf2 = _fold_layernorm(wgi, bgi, gamma, beta)
    outs = {}
    for prod in (10, 7):
        h = h0.clone()
        rs = ops.RowStats(m, c)
        ops.linear(o_in, w_out, b_out, residual=h, out=h, ld_out=c, row_stats=rs, force_tile=prod)
        close(h, o_in.float() @ w_out.float().t() + b_out.float() + h0.float(), 2e-3, 2e-3, f"producer output (tile {prod})")
        assert rs.parts == c // 160, (prod, rs.parts)
        hf = h.double()
        torch.testing.assert_close(rs.partial.view(m, -1, 2)[:, :rs.parts].double().sum(1), torch.stack([hf.sum(1), (hf * hf).sum(1)], -1), rtol=1e-5, atol=1e-2)
        ln = F.layer_norm(h.float(), (c,), gamma.float(), beta.float(), 1e-5)
        hc = ln @ wg.float().t() + bg.float()
        for cons in (10, 7):
            q = ops.conv_gemm(h, wf, nq, bias=bf, ln=(rs, cs, 1e-5), force_tile=cons)
            close(q, ln @ wq.float().t(), 4e-3, 4e-3, f"LayerNorm consumer (producer tile {prod}, consumer tile {cons})")
            g = ops.conv_gemm(h, wf2, 2 * nq, bias=bf2, act=ops.ACT_GEGLU, ln=(rs, cs2, 1e-5), force_tile=cons)
            close(g, hc[:, :nq] * F.gelu(hc[:, nq:]), 5e-3, 5e-3, f"LayerNorm GEGLU consumer (producer tile {prod}, consumer tile {cons})")
            outs[(prod, cons)] = (q, g)
        outs[prod] = h
    assert torch.equal(outs[10], outs[7]), "the two kernels accumulate every output element in the same order (K ascending, fp32): bit-identical stores"


@pytest.mark.parametrize("m,n,k", [(16384, 1280, 1280), (65536, 640, 640), (16384, 2560, 1280), (16384, 1280, 5120), (1000, 328, 192)])
def test_two_workgroup_tile_is_bit_identical_to_the_256x320_kernel(m, n, k):
    """Both kernels run the same 16x16x32 MFMA sequence per output block over K ascending in fp32 and the same epilogue arithmetic, so the stored fp16
    values must agree bit for bit -- plain, residual, GEGLU; many tiles per workgroup (the persistent walk and the three-stage ring across tiles)."""
    x, w, b = rnd(m, k, seed=200), rnd(n, k, scale=1 / math.sqrt(k), seed=201), rnd(n, seed=202)
    res = rnd(m, n, seed=203)
    for kw in ({}, {"residual": res}):
        a10, a7 = ops.linear(x, w, b, force_tile=10, **kw), ops.linear(x, w, b, force_tile=7, **kw)
        assert torch.equal(a10, a7), f"{kw.keys()}: max diff {float((a10.float() - a7.float()).abs().max())}"
    close(a10, x.float() @ w.float().t() + b.float() + res.float(), 3e-3, 3e-3, "against fp32")
    if n % 32 == 0:
        wi, bi = ops.geglu_interleave(w, b)
        g10 = ops.conv_gemm(x, wi, n, bias=bi, act=ops.ACT_GEGLU, force_tile=10)
        g7 = ops.conv_gemm(x, wi, n, bias=bi, act=ops.ACT_GEGLU, force_tile=7)
        assert torch.equal(g10, g7)


@pytest.mark.parametrize("t256", [6, 7, 8])
def test_gemm256_two_source_conv_and_swapped(t256):
    n, h, w, c0, c1, co = 2, 24, 24, 128, 64, 256
    x0, x1 = rnd(n, h, w, c0, seed=120), rnd(n, h, w, c1, seed=121)
    wt = rnd(co, c0 + c1, 3, 3, scale=0.03, seed=122)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    out = ops.conv_gemm(x0, wk, co, x1=x1, kh=3, pad=1, force_tile=t256)
    ref = F.conv2d(torch.cat([x0, x1], -1).permute(0, 3, 1, 2).float(), wt.float(), padding=1).permute(0, 2, 3, 1)
    close(out.reshape(ref.shape), ref, 3e-3, 3e-3, "gemm256 conv concat")
    # V^T = Wv X^T (weights as the "activation" operand), M = 256 rows of W, N = tokens
    xa = rnd(1152, 192, seed=123)
    wv = rnd(256, 192, scale=0.07, seed=124)
    vt = ops.conv_gemm(wv, xa, 1152, force_tile=t256)
    close(vt, wv.float() @ xa.float().t(), 2e-3, 2e-3, "gemm256 swapped operands")


@pytest.mark.parametrize("t256", [6, 7, 8])
def test_gemm256_matches_small_tile_bitwise_class(t256):
    # same K order and fp32 accumulation in both kernels: results agree to fp16 rounding of the same fp32 sums
    m, n, k = 768, 512, 1280
    x, w = rnd(m, k, seed=130), rnd(n, k, scale=1 / math.sqrt(k), seed=131)
    a = ops.linear(x, w, force_tile=1)
    b = ops.linear(x, w, force_tile=t256)
    close(a, b.float(), 1e-3, 1e-3, "gemm256 vs 128x128")


def test_conv_two_source_concat():
    n, h, w, c0, c1, co = 2, 16, 16, 128, 64, 128
    x0, x1 = rnd(n, h, w, c0, seed=30), rnd(n, h, w, c1, seed=31)
    wt = rnd(co, c0 + c1, 3, 3, scale=0.03, seed=32)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    out = ops.conv_gemm(x0, wk, co, x1=x1, kh=3, pad=1)
    ref = F.conv2d(torch.cat([x0, x1], -1).permute(0, 3, 1, 2).float(), wt.float(), padding=1).permute(0, 2, 3, 1)
    close(out.reshape(ref.shape), ref, 3e-3, 3e-3, "conv concat")


# ----------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v.float())


@pytest.mark.parametrize("b,h,nq,nk,d,dpad", [
    (2, 2, 128, 128, 64, 64), (1, 3, 256, 192, 64, 64), (2, 2, 200, 77, 64, 64), (1, 10, 1024, 1024, 64, 64),
    (2, 3, 300, 200, 64, 64), (1, 2, 4096, 4096, 64, 64), (2, 2, 1000, 77, 64, 64),
    (2, 2, 64, 64, 40, 48), (1, 2, 160, 128, 80, 80), (1, 2, 96, 77, 160, 160), (1, 8, 4096, 4096, 40, 48),
])
@pytest.mark.parametrize("force32", [False, True])
def test_attention(b, h, nq, nk, d, dpad, force32):
    if force32 and not (dpad == 64 and nq >= 256):
        pytest.skip("only d_head 64 with >= 256 queries has two kernels")
    nk_pad = -(-nk // 64) * 64
    q = torch.zeros(b, nq, h, dpad, dtype=torch.float16, device=DEV)
    k = torch.zeros(b, nk_pad, h, dpad, dtype=torch.float16, device=DEV)
    v = torch.zeros(b, nk_pad, h, dpad, dtype=torch.float16, device=DEV)
    q[..., :d] = rnd(b, nq, h, d, seed=40)
    k[:, :nk, :, :d] = rnd(b, nk, h, d, seed=41)
    v[:, :nk, :, :d] = rnd(b, nk, h, d, seed=42)
    k[:, nk:] = 7.0  # garbage in padded keys must be masked out, not rely on zeros
    v[:, nk:, :, :d] = -3.0
    vt = v.permute(2, 3, 0, 1).contiguous()  # [h, dpad, b, nk_pad] == V^T[(h,d)][b*nk_pad + j]
    scale = d ** -0.5
    out = ops.attention(q, k, vt, batch=b, heads=h, nq=nq, nk=nk, nk_pad=nk_pad, dpad=dpad, scale=scale,
                        q_bs=nq * h * dpad, q_rs=h * dpad, k_bs=nk_pad * h * dpad, k_rs=h * dpad,
                        vt_bs=nk_pad, vt_hs=dpad * b * nk_pad, vt_ds=b * nk_pad, force32=force32)
    ref = _attn_ref(q.permute(0, 2, 1, 3)[..., :d], k.permute(0, 2, 1, 3)[:, :, :nk, :d], v.permute(0, 2, 1, 3)[:, :, :nk, :d], scale)
    got = out.reshape(b, nq, h, dpad).permute(0, 2, 1, 3)
    close(got[..., :d], ref, 2e-3, 2e-3, "attention")
    assert float(got[..., d:].abs().max()) == 0.0 if dpad > d else True


@pytest.mark.parametrize("force32", [False, True])
def test_attention_spiked_max(force32):
    # one key dominates late in the sequence: forces the online-softmax rescale branch
    b, h, n, d = 1, 1, 512, 64
    q, k, v = rnd(b, n, h, d, seed=50), rnd(b, n, h, d, seed=51), rnd(b, n, h, d, seed=52)
    k[0, 400, 0] = q[0, 17, 0] * 6
    vt = v.permute(2, 3, 0, 1).contiguous()
    out = ops.attention(q, k, vt, batch=b, heads=h, nq=n, nk=n, nk_pad=n, dpad=d, scale=d ** -0.5, q_bs=n * d, q_rs=d,
                        k_bs=n * d, k_rs=d, vt_bs=n, vt_hs=d * n, vt_ds=n, force32=force32)
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), d ** -0.5)
    close(out.reshape(b, n, h, d).permute(0, 2, 1, 3), ref, 2e-3, 2e-3, "attention spike")


def _attn_d64(q, k, v, nk=None, force32=False):
    b, n, h, d = q.shape
    nkp = k.shape[1]
    vt = v.permute(2, 3, 0, 1).contiguous()
    out = ops.attention(q, k, vt, batch=b, heads=h, nq=n, nk=nk or nkp, nk_pad=nkp, dpad=d, scale=d ** -0.5, q_bs=n * h * d, q_rs=h * d,
                        k_bs=nkp * h * d, k_rs=h * d, vt_bs=nkp, vt_hs=d * b * nkp, vt_ds=b * nkp, force32=force32)
    return out.reshape(b, n, h, d).permute(0, 2, 1, 3)


@pytest.mark.parametrize("case", ["negative_scores", "staircase", "late_spikes", "threshold_edge", "ragged_spike"])
def test_attention_running_maximum_paths(case):
    """The d_head-64 kernel keeps a running maximum that moves only when a tile exceeds it by 2^6 (scores in the log2 domain) and folds
    '- maximum' into the score MFMAs (csrc/fmx_attention.hip, attn_q64v2_kernel).  Inputs that force each branch: rows whose scores are all
    far below zero (the first tile must set a NEGATIVE maximum exactly), maxima that climb tile after tile by more / less than the threshold,
    several dominant keys late in the sequence, a ragged key count with the dominant key in the last partial tile.  fp64-free reference: torch
    fp32 softmax on the same fp16 inputs; checked on every element."""
    b, h, n, d = 2, 2, 512, 64
    q, k, v = rnd(b, n, h, d, seed=53), rnd(b, n, h, d, seed=54), rnd(b, n, h, d, seed=55)
    nk = None
    if case == "negative_scores":
        q, k = q.abs().contiguous(), (-(k.abs()) - 0.5).contiguous()   # every q.k is strongly negative (about -12 * scale per pair of rows)
    elif case == "staircase":
        for t in range(8):                                  # key t*64+5 beats everything before it for query block t..: maxima climb every tile
            k[:, t * 64 + 5] = q[:, 100] * (0.5 + 0.45 * t)
    elif case == "late_spikes":
        k[0, 400, 0] = q[0, 17, 0] * 6
        k[1, 449, 1] = q[1, 300, 1] * 8
        k[0, 510, 1] = q[0, 511, 1] * 5
    elif case == "threshold_edge":
        for t, f in enumerate((0.2, 0.9, 1.0, 1.1, 1.6, 1.7, 3.0, 3.05)):   # growth per tile straddles the 2^6 threshold (q.q*scale ~ 8 per unit factor)
            k[:, t * 64 + 1] = q[:, 7] * f
    elif case == "ragged_spike":
        nk = 455
        k[0, 450, 0] = q[0, 33, 0] * 7
        k[:, nk:] = 9.0
        v[:, nk:] = -5.0
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    assert bool(torch.isfinite(ref).all())
    for f32 in (False, True):
        close(_attn_d64(q, k, v, nk=nk, force32=f32), ref, 2e-3, 2e-3, f"attention {case} force32={f32}")


@pytest.mark.parametrize("nk,spikes", [(77, ((40, 3, 4.0), (70, 300, 5.0))), (64, ((33, 5, 5.0),)), (130, ((5, 1, 3.0), (100, 2, 4.5), (129, 3, 6.0))),
                                       (200, ()), (256, ((63, 7, 3.0), (64, 8, 4.5), (250, 9, 6.0)))])
def test_attention_d64_short_context_pipelined(nk, spikes):
    """Contexts of up to four key tiles (cross-attention: 77 tokens) run the sub-tile pipelined kernel (attn_q64v3_kernel: the exponentials of
    one 32-key sub-tile under the MFMAs of the next, the running maximum checked per sub-tile).  Dominant keys placed in the second sub-tile of
    the first tile, in the ragged last tile and right at a sub-tile edge force the rescale branch at every position of the pipeline; one case
    has all-negative scores (the first sub-tile must set a negative maximum exactly).  Against torch fp32 and the 32-query kernel."""
    b, h, n, d = 2, 3, 1000, 64
    nkp = -(-nk // 64) * 64
    q = rnd(b, n, h, d, seed=81)
    k, v = torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, scale=1.2, seed=82), rnd(b, nk, h, d, seed=83)
    if not spikes:
        q, k = q.abs().contiguous(), (-(k.abs()) - 0.5).contiguous()
    k[:, nk:], v[:, nk:] = 6.0, -4.0
    for key, qi, f in spikes:
        k[key % b, key, key % h] = q[key % b, qi, key % h] * f
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    assert bool(torch.isfinite(ref).all())
    got = _attn_d64(q, k, v, nk=nk)
    # (3e-3: with a handful of keys the output is not an average -- it carries the fp16 rounding of P and of the pre-scaled Q of a
    #  dominant key at full size; the 4096-key tests above hold 2e-3)
    close(got, ref, 3e-3, 3e-3, f"attention d64, {nk} keys")
    close(got, _attn_d64(q, k, v, nk=nk, force32=True).float(), 3e-3, 3e-3, "pipelined vs 32-query kernel")


@pytest.mark.parametrize("b,h,nq,nk", [(16, 20, 1024, 77), (16, 10, 4096, 77), (1, 1, 4096, 77), (2, 3, 1000, 96), (2, 2, 300, 97), (1, 2, 2000, 128),
                                       (3, 5, 640, 32), (2, 2, 257, 65), (1, 7, 128 * 9 + 5, 31)])
def test_attention_d64_short_context_persistent(b, h, nq, nk):
    """Contexts of at most 128 keys on d_head 64 (every cross-attention against the 77-token text context) run attn_short_kernel (round 3): a
    persistent workgroup walks several 128-query tiles of one (batch, head) with the K / V^T tiles staged once and the next tile's Q rows
    requested ahead, a one-pass softmax over the 1-4 key blocks that hold keys.  Cases: the two bench shapes (3 and 6 tiles per workgroup, a
    short last chunk), one (batch, head) split over the whole chip, key counts on and next to every 32-key block edge, ragged query counts
    (a partial last tile, a partial last wave), padded keys holding garbage, a dominant key in the last valid position.  Against torch fp32
    and the 32-query kernel."""
    d = 64
    nkp = -(-nk // 64) * 64
    q = rnd(b, nq, h, d, seed=71)
    k, v = torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, scale=1.2, seed=72), rnd(b, nk, h, d, seed=73)
    k[:, nk:], v[:, nk:] = 6.0, -4.0
    k[b - 1, nk - 1, h - 1] = q[b - 1, nq - 1, h - 1] * 5     # the last query's dominant key is the last valid one
    k[0, 0, 0] = q[0, 0, 0] * 4
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    got = _attn_d64(q, k, v, nk=nk)
    assert bool(torch.isfinite(got).all())
    close(got, ref, 3e-3, 3e-3, f"short-context attention b{b} h{h} nq{nq} nk{nk}")
    if b * h * nq <= 400000:
        close(got, _attn_d64(q, k, v, nk=nk, force32=True).float(), 3e-3, 3e-3, "short-context kernel vs 32-query kernel")


@pytest.mark.parametrize("b,n,nk", [(2, 1024, 1024), (1, 4096, 4096), (2, 1000, 1000), (3, 200, 77)])
def test_attention_single_head_512_wide(b, n, nk):
    """The VAE mid-block attention (one 512-wide head, backend/nn/vae.py:118-137) on its fused kernel (two query fragments x four channel
    slices whose partial scores meet in LDS; FMX_ATTN512_SLICES=2 selects the four-wave form), online softmax over 32-key steps.  Ragged query and key counts
    (rows beyond nk in the last step are masked, padded V^T columns hold garbage), a dominant late key.  Against torch fp32 on the same fp16 inputs."""
    c = 512
    nkp = -(-nk // 32) * 32
    q = rnd(b * n, c, seed=121)
    k = torch.full((b * nkp, c), 7.0, dtype=torch.float16, device=DEV)
    v = torch.full((b, nkp, c), -5.0, dtype=torch.float16, device=DEV)
    kv, vv = rnd(b, nk, c, seed=122), rnd(b, nk, c, seed=123)
    k.view(b, nkp, c)[:, :nk] = kv
    v[:, :nk] = vv
    k.view(b, nkp, c)[b - 1, nk - 1] = q.view(b, n, c)[b - 1, n - 1] * 0.5      # a dominant key in the last valid row, for the last query
    vt = v.permute(2, 0, 1).reshape(c, b * nkp).contiguous()                     # V^T [512][image * nk_pad + key]
    o = torch.empty(b * n, c, dtype=torch.float16, device=DEV)
    ops.attention_single_head512(q, k, vt, o, batch=b, nq=n, nk=nk, nk_pad=nkp, q_bs=n * c, q_rs=c, k_bs=nkp * c, k_rs=c, vt_bs=nkp, vt_ds=b * nkp,
                                 scale=c ** -0.5)
    kk = k.view(b, nkp, c)[:, :nk].float()
    ref = torch.softmax(q.view(b, n, c).float() @ kk.transpose(1, 2) * c ** -0.5, -1) @ v[:, :nk].float()
    assert bool(torch.isfinite(o).all())
    close(o.view(b, n, c), ref, 3e-3, 3e-3, f"512-wide single-head attention b{b} n{n} nk{nk}")


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("nk", [1, 20, 33, 63])
def test_attention_fewer_keys_than_one_tile(d, nk):
    """A context shorter than one 64-key tile on the 64-query kernels (d_head 64: the sub-tile pipelined kernel, whose second sub-tile is then
    partly or wholly masked; d_head 128: the wave-specialised one): masked scores are -inf, a sub-tile without a valid key must not move the
    running maximum, padded keys hold garbage.  One key: the output is that key's value row."""
    b, h, n = 2, 2, 512
    q = rnd(b, n, h, d, seed=91)
    k, v = torch.full((b, 64, h, d), 9.0, dtype=torch.float16, device=DEV), torch.full((b, 64, h, d), -7.0, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, seed=92), rnd(b, nk, h, d, seed=93)
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    got = _attn_d64(q, k, v, nk=nk)
    assert bool(torch.isfinite(got).all())
    close(got, ref, 3e-3, 3e-3, f"attention d{d}, {nk} keys")
    if nk == 1:
        close(got, v[:, 0].float()[:, :, None, :].expand(b, h, n, d), 1e-3, 1e-3, "one key: its value row")


def test_attention_d64_generations_agree(monkeypatch):
    """Second-generation d_head-64 kernel (default) against the first-generation one (the 32-query kernel reached through the test hook):
    same inputs, results within fp16 rounding of each other on a 4096-token problem (Q pre-scaling and the deferred maximum change the
    rounding points, not the mathematics)."""
    b, h, n, d = 1, 2, 4096, 64
    q, k, v = rnd(b, n, h, d, seed=57), rnd(b, n, h, d, scale=1.5, seed=58), rnd(b, n, h, d, seed=59)
    a, c = _attn_d64(q, k, v), _attn_d64(q, k, v, force32=True)
    close(a, c.float(), 2e-3, 1e-3, "q64v2 vs 32-query kernel")


@pytest.mark.parametrize("b,h,nq,nk", [(3, 25, 1024, 1024), (9, 16, 1000, 1024), (2, 3, 1000, 1000), (1, 2, 200, 512), (9, 16, 1024, 320)])
def test_attention_d64_partial_round_key_split(b, h, nq, nk):
    """The d_head-64 kernel turns the 256-query tiles beyond the last full round of workgroup slots (2 per CU) into key-split workgroups: two
    per tile, 128 queries each, wave pairs walking one half of the keys and merging (m, l, O) through LDS (csrc/fmx_attention.hip).  On a
    256-CU part: 300 tiles -> none split (the remainder would not fit one round); 576 tiles -> 512 whole + 128 key-split workgroups in ONE
    launch, with a ragged last query tile; small launches -> all key-split, incl. ragged queries and a ragged last key tile in the upper half;
    an odd number of key tiles (320 keys = 5) -> never split."""
    d = 64
    nkp = -(-nk // 64) * 64
    q, k, v = rnd(b, nq, h, d, seed=61), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, scale=1.3, seed=62), rnd(b, nk, h, d, seed=63)
    k[:, nk:], v[:, nk:] = 6.0, -4.0
    k[0, nk - 3, 0] = q[0, nq - 1, 0] * 5          # a dominant key at the very end of the upper half, for the last query
    k[b - 1, 2, h - 1] = q[b - 1, 0, h - 1] * 5    # and one at the start of the lower half, for the first
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    close(_attn_d64(q, k, v, nk=nk), ref, 2e-3, 2e-3, f"attention d64 b{b} h{h} nq{nq} nk{nk}")


@pytest.mark.parametrize("b,h,nq,nk", [(16, 20, 1024, 1024), (9, 20, 1000, 896), (9, 20, 1000, 1000), (9, 20, 1000, 960), (3, 10, 4096, 2048)])
def test_attention_d64_launches_of_many_entries(b, h, nq, nk):
    """Launches of more 256-query entries than the chip holds at once, without a key-split tail (the bench shapes: 1280 and 2560 entries): several rounds of
    workgroups, ragged query counts, a ragged last key tile, an odd tile count, dominant keys in the first and the last key tile.  (Written for the persistent
    form of attn_q64v2_kernel that round 4 measured and did not keep -- profiles/r12a_* -- and kept as coverage of the many-round launches.)  Against torch
    fp32, every element."""
    d = 64
    nkp = -(-nk // 64) * 64
    q, k, v = rnd(b, nq, h, d, seed=161), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, scale=1.3, seed=162), rnd(b, nk, h, d, seed=163)
    k[:, nk:], v[:, nk:] = 6.0, -4.0
    for bb in range(b):                                   # a dominant key in the FIRST tile of every (batch, head): the tile the previous entry prefetched
        k[bb, 3 + bb, bb % h] = q[bb, (17 * bb) % nq, bb % h] * 5
    k[b - 1, nk - 2, h - 1] = q[b - 1, nq - 1, h - 1] * 5  # and one in the last (partial) tile of the last entry
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    got = _attn_d64(q, k, v, nk=nk)
    assert bool(torch.isfinite(got).all())
    close(got, ref, 2e-3, 2e-3, f"persistent attention d64 b{b} h{h} nq{nq} nk{nk}")


@pytest.mark.parametrize("b,h,nq,nk", [(1, 3, 1024, 1024), (2, 2, 4352, 4352), (1, 2, 300, 1000), (2, 3, 512, 77), (1, 3, 1024, 950), (2, 24, 1408, 1000),
                                       (2, 24, 1500, 1024)])
def test_attention_d128_64_queries_per_wave(b, h, nq, nk):
    """d_head 128 (Flux) on the 64-query-per-wave kernels against the fp32 reference and against the generic 32-query kernel (test hook):
    ragged query / key counts, a short context, a dominant late key.  Launches the key-split rule shortens (the first three) run the symmetric
    kernel, the others the wave-specialised one: an odd number of key tiles, a short context, more than one round of workgroups (288; then
    a key-split tail of the symmetric kernel behind the full round)."""
    d = 128
    nkp = -(-nk // 64) * 64
    q, k, v = rnd(b, nq, h, d, seed=67), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV), torch.zeros(b, nkp, h, d, dtype=torch.float16, device=DEV)
    k[:, :nk], v[:, :nk] = rnd(b, nk, h, d, scale=1.2, seed=68), rnd(b, nk, h, d, seed=69)
    k[:, nk:], v[:, nk:] = 5.0, -3.0
    k[0, nk - 2, 0] = q[0, nq - 1, 0] * 3
    ref = _attn_ref(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3)[:, :, :nk], v.permute(0, 2, 1, 3)[:, :, :nk], d ** -0.5)
    got = _attn_d64(q, k, v, nk=nk)
    close(got, ref, 2e-3, 2e-3, f"attention d128 b{b} h{h} nq{nq} nk{nk}")
    close(got, _attn_d64(q, k, v, nk=nk, force32=True).float(), 2e-3, 1e-3, "64-query vs 32-query kernel, d128")


def test_softmax_rows():
    x = rnd(300, 1000, scale=3, seed=60)
    ref = x.float().softmax(-1)
    ops.softmax_rows_(x)
    close(x, ref, 2e-3, 1e-5, "softmax rows")


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,h,w,c0,c1,silu,eps", [(2, 16, 16, 64, 0, True, 1e-5), (2, 32, 32, 320, 0, True, 1e-5), (3, 8, 8, 1280, 640, True, 1e-5),
                                                   (2, 8, 8, 1280, 1280, False, 1e-6), (1, 64, 64, 128, 0, True, 1e-6), (2, 7, 5, 960, 0, False, 1e-6)])
def test_groupnorm(n, h, w, c0, c1, silu, eps):
    x0 = rnd(n, h, w, c0, scale=2, seed=70) + 0.5
    x1 = rnd(n, h, w, c1, seed=71) if c1 else None
    c = c0 + c1
    g, b = (1 + 0.1 * rnd(c, seed=72)), 0.1 * rnd(c, seed=73)
    out = ops.groupnorm(x0, g, b, eps, x1=x1, silu=silu)
    xin = (torch.cat([x0, x1], -1) if c1 else x0).permute(0, 3, 1, 2).float()
    ref = F.group_norm(xin, 32, g.float(), b.float(), eps)
    if silu:
        ref = F.silu(ref)
    close(out, ref.permute(0, 2, 3, 1), 2e-3, 2e-3, "groupnorm")


def _partial_to_sums(st, n, c):
    p = st.partial.reshape(-1)[:n * st.nchunks * c * 2].view(n, st.nchunks, c, 2).double()
    return p.sum(1)   # [n, c, 2]


@pytest.mark.parametrize("n,hh,ww,cin,cout,kh,tile", [(2, 32, 32, 64, 320, 3, 7), (2, 16, 32, 128, 640, 1, 7), (3, 16, 16, 64, 256, 3, 6), (2, 32, 32, 64, 320, 1, 0),
                                                        (1, 48, 16, 64, 1280, 3, 7), (2, 10, 10, 64, 320, 3, 7), (2, 8, 8, 128, 320, 3, 0),
                                                        (2, 32, 32, 64, 128, 3, 9), (1, 64, 64, 128, 136, 3, 9), (3, 32, 16, 64, 128, 1, 9), (2, 24, 16, 64, 128, 3, 9)])
def test_gemm_output_statistics(n, hh, ww, cin, cout, kh, tile):
    """fmx_gemm_conv_stats_f16: the per-(image, channel) sum / sum of squares of the fp16 OUTPUT, from the epilogue of the 256-row tiles when
    an image is a whole number of them (32x32, 16x32, 48x16, 16x16 pixels), by the pass behind the GEMM otherwise (10x10, 8x8, or a 4-wave
    tile choice) -- against sums over the tensor the kernel stored; bias + per-image row vector + residual epilogues included."""
    x = rnd(n, hh, ww, cin, seed=300)
    wt = rnd(cout, cin, kh, kh, scale=1 / math.sqrt(cin * kh * kh), seed=301)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    b, emb, res = rnd(cout, seed=302), rnd(n, cout, seed=303), rnd(n * hh * ww, cout, seed=304)
    for kw in (dict(bias=b), dict(bias=b, rowvec=emb), dict(bias=b, residual=res), dict(bias=b, rowvec=emb, residual=res)):
        out, st = ops.conv_gemm(x, wk, cout, kh=kh, pad=kh // 2, force_tile=tile, stats=True, **kw)
        plain = ops.conv_gemm(x, wk, cout, kh=kh, pad=kh // 2, force_tile=tile, **kw)
        # same fp32 accumulators and epilogue arithmetic in both instantiations (tools/debug_stats.py: the rounding residues agree in every
        # element); a value that sits on an fp16 rounding tie may still land on either neighbour (seen: 1 element of 655 360, exactly
        # half an ulp from both), so: at most one fp16 ulp apart, in at most a handful of elements
        if not torch.equal(out, plain):
            d = (out.float() - plain.float()).abs()
            ulp = torch.maximum(plain.float().abs(), torch.tensor(2.0 ** -14, device=DEV)).log2().floor().exp2() * 2.0 ** -10
            assert bool((d <= ulp).all()) and int((d > 0).sum()) <= max(2, d.numel() // 100000), \
                f"kw={sorted(kw)}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.4g}"
        if tile in (6, 7) and (hh * ww) % 256 == 0:
            assert st.nchunks == hh * ww // 256, "expected the fused (epilogue) statistics path"
        if tile == 9 and (hh * ww) % 512 == 0:
            assert st.nchunks == hh * ww // 512, "expected the fused (epilogue) statistics path of the 512-row tile"
        o = out.view(n, hh * ww, cout).double()
        want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
        got = _partial_to_sums(st, n, cout)
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-3)


def test_groupnorm_uses_producer_statistics_and_keeps_them():
    """ResBlock pattern: conv -> GroupNorm+SiLU with the conv's statistics; a tensor with two consumers (a skip connection) keeps them."""
    n, hh, ww, cin, c = 2, 32, 32, 64, 320
    x = rnd(n, hh, ww, cin, seed=310)
    wk = rnd(c, cin * 9, scale=0.05, seed=311)
    g, b = (1 + 0.1 * rnd(c, seed=312)), 0.1 * rnd(c, seed=313)
    h, st = ops.conv_gemm(x, wk, c, kh=3, pad=1, stats=True)
    h4 = ops.attach_stats(h.view(n, hh, ww, c), st)
    snap = st.partial.clone()
    y1 = ops.groupnorm(h4, g, b, 1e-5, silu=True)
    assert torch.equal(st.partial, snap), "apply must not consume the statistics"
    ref = F.silu(F.group_norm(h4.permute(0, 3, 1, 2).float(), 32, g.float(), b.float(), 1e-5)).permute(0, 2, 3, 1)
    close(y1, ref, 2e-3, 2e-3, "groupnorm on producer statistics")
    g2, b2 = torch.cat([g, g]), torch.cat([b, b])
    y2 = ops.groupnorm(h4, g2, b2, 1e-6, x1=h4, silu=False)
    ref2 = F.group_norm(torch.cat([h4, h4], -1).permute(0, 3, 1, 2).float(), 32, g2.float(), b2.float(), 1e-6).permute(0, 2, 3, 1)
    close(y2, ref2, 2e-3, 2e-3, "two-source groupnorm on producer statistics")
    ops.clear_stats(h4)
    close(ops.groupnorm(h4, g, b, 1e-5, silu=True), ref, 2e-3, 2e-3, "groupnorm with its own statistics pass")


def test_groupnorm_large_offset_values():
    """E[x^2] - E[x]^2 in fp32 with a mean 30x the standard deviation (worst case for the one-pass formula at fp16 input range)."""
    n, h, w, c = 2, 64, 64, 320
    x = (rnd(n, h, w, c, scale=1.0, seed=320) + 30.0)
    g, b = (1 + 0.1 * rnd(c, seed=321)), 0.1 * rnd(c, seed=322)
    ref = F.group_norm(x.permute(0, 3, 1, 2).float(), 32, g.float(), b.float(), 1e-5).permute(0, 2, 3, 1)
    close(ops.groupnorm(x, g, b, 1e-5), ref, 1e-2, 1e-2, "groupnorm, mean >> std")


@pytest.mark.parametrize("rows,c", [(100, 320), (257, 640), (64, 1280), (5, 64), (33, 2048)])
def test_layernorm(rows, c):
    x = rnd(rows, c, scale=3, seed=80) + 1
    g, b = (1 + 0.1 * rnd(c, seed=81)), 0.1 * rnd(c, seed=82)
    out = ops.layernorm(x, g, b, 1e-5)
    close(out, F.layer_norm(x.float(), (c,), g.float(), b.float(), 1e-5), 2e-3, 2e-3, "layernorm")


def test_timestep_embedding_and_silu():
    t = torch.tensor([999.0, 500.0, 3.0, 0.0], device=DEV)
    emb = ops.timestep_embedding(t, 320)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=DEV) / half)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    close(emb, ref, 1e-3, 1.5e-3, "timestep embedding")  # fp16 output of values in [-1, 1]; device sin/cos of args up to 999
    x = rnd(1000, scale=3, seed=90)
    close(ops.silu(x), F.silu(x.float()), 2e-3, 1e-3, "silu")


def test_pack_cfg_sampler_kernels():
    b, c, h, w = 2, 4, 16, 12
    g = torch.Generator("cpu").manual_seed(5)
    x = (torch.randn(b, c, h, w, generator=g) * 10).to(DEV)
    sigma = torch.tensor([14.6, 3.2], device=DEV)
    packed = ops.unet_pack_input(x, sigma, reps=2)
    xc = x / (sigma.view(-1, 1, 1, 1) ** 2 + 1.0) ** 0.5
    cols = F.unfold(torch.cat([xc, xc]), 3, padding=1)  # [2b, c*9, hw] ordered (c, ky, kx)
    cols = cols.reshape(2 * b, c, 9, h * w).permute(0, 3, 2, 1).reshape(2 * b * h * w, 9 * c)  # -> (tap, c)
    close(packed[:, :36], cols, 1e-3, 1e-3, "pack_input")
    assert float(packed[:, 36:].abs().max()) == 0.0
    eps = rnd(2 * b, h, w, 4, seed=91)
    den = ops.cfg_combine(eps, 4, x, sigma, 2, 7.0)
    e = eps.float().permute(0, 3, 1, 2)
    du = x - e[:b] * sigma.view(-1, 1, 1, 1)
    dc = x - e[b:] * sigma.view(-1, 1, 1, 1)
    torch.testing.assert_close(den, du + (dc - du) * 7.0, rtol=1e-5, atol=1e-4)  # fma contraction order differs
    den1 = ops.cfg_combine(eps, 4, x, sigma, 1, 1.0)
    torch.testing.assert_close(den1, x - e[:b] * sigma.view(-1, 1, 1, 1), rtol=1e-6, atol=1e-5)
    nz = torch.randn(b, c, h, w, generator=g).to(DEV)
    out = ops.euler_step(x, den, 14.6, 9.7, noise=nz, noise_scale=0.3)
    torch.testing.assert_close(out, x + (x - den) / 14.6 * (9.7 - 14.6) + nz * 0.3, rtol=1e-6, atol=1e-5)
    out = ops.lincomb3(x, den, den1, 0.7, 0.2, -0.1)
    torch.testing.assert_close(out, 0.7 * x + (0.2 * den - 0.1 * den1), rtol=1e-6, atol=1e-5)
    # generic 1..8-term linear combination (multi-stage / multistep samplers); sizes with and without a 4-element tail, in place
    for shape in ((2, 4, 8, 8), (1, 4, 5, 7), (3,)):
        srcs = [torch.randn(*shape, generator=g).to(DEV) for _ in range(8)]
        coefs = [0.7, -1.3, 0.25, 2.0, -0.5, 0.125, 1.5, -0.75]
        for n in range(1, 9):
            want = sum(c * t.double() for c, t in zip(coefs[:n], srcs[:n])).float()
            torch.testing.assert_close(ops.lincomb(srcs[:n], coefs[:n]), want, rtol=1e-6, atol=1e-5)
        if len(shape) == 4:
            many = [torch.randn(*shape, generator=g).to(DEV) for _ in range(19)]
            cf = [((-1) ** k) * (0.1 + 0.07 * k) for k in range(19)]
            torch.testing.assert_close(ops.lincomb(many, cf), sum(c * t.double() for c, t in zip(cf, many)).float(), rtol=1e-5, atol=1e-5)
            lo, hi, pv = srcs[0], srcs[0] + 0.01 * srcs[1], srcs[2]
            delta = torch.maximum(torch.tensor(0.0078, device=DEV), 0.05 * torch.maximum(lo.abs(), pv.abs()))
            want = float(torch.linalg.norm(((lo - hi) / delta).double()) / lo.numel() ** 0.5)
            got = ops.error_norm(lo, hi, pv, 0.0078, 0.05)
            assert abs(got - want) < 1e-5 * want and got == ops.error_norm(lo, hi, pv, 0.0078, 0.05)  # deterministic reduction
        acc = srcs[0].clone()
        ops.lincomb([acc, srcs[1]], [1.0, 2.0], out=acc)
        torch.testing.assert_close(acc, srcs[0] + 2.0 * srcs[1], rtol=1e-6, atol=1e-5)


def test_vae_pack_unpack_and_im2col():
    b, c, h, w = 2, 4, 8, 8
    z = torch.randn(b, c, h, w, device=DEV)
    p = ops.vae_pack_latent(z, 0.18215, 0.0, ld=8)
    close(p[..., :4], (z / 0.18215).permute(0, 2, 3, 1), 1e-3, 1e-3, "vae pack")
    assert float(p[..., 4:].abs().max()) == 0.0
    col = ops.im2col3x3_smallc(p, 4)
    ref = F.unfold(p[..., :4].permute(0, 3, 1, 2).float(), 3, padding=1).reshape(b, c, 9, h * w).permute(0, 3, 2, 1).reshape(-1, 36)
    close(col[:, :36], ref, 0, 0, "im2col")
    y = rnd(b * 64 * 64, 4, seed=95)
    out = torch.empty(b, 64, 64, 3, device=DEV)
    ops.vae_unpack_image(y, 4, b * 64 * 64, 3, out)
    torch.testing.assert_close(out.reshape(-1, 3), torch.clamp((y[:, :3].float() + 1) / 2, 0, 1))


def test_image_rng_variations_on_device():
    """modules/rng.py ImageRNG with variation seeds / seed resize on the device against the reference's values (made on the CPU): the CPU source's
    draws are bit-identical but the slerp (norm, acos, sin, a division by sin(omega)) then runs in the device's libm; the Philox source adds the
    2e-6 of the device Box-Muller per draw.  Tolerances 5e-5 / 2e-4 on N(0, 1) values."""
    from forge_amd.modules import rng as prod_rng, shared
    from oracle.make_golden import RNG_VARIATION_CASES
    from conftest import load_golden
    g = load_golden("rng_variations.pt")
    saved = shared.opts.randn_source, shared.opts.eta_noise_seed_delta
    try:
        shared.opts.eta_noise_seed_delta = g["eta_noise_seed_delta"]
        for source, tol in (("CPU", 5e-5), ("NV", 2e-4)):
            shared.opts.randn_source = source
            for cname, kw in RNG_VARIATION_CASES.items():
                r = prod_rng.ImageRNG(g["shape"], g["seeds"], device=DEV, **kw)
                for want in g[(source, cname)]:
                    got = r.next()
                    assert got.is_cuda and float((got.cpu() - want).abs().max()) <= tol, (source, cname)
    finally:
        shared.opts.randn_source, shared.opts.eta_noise_seed_delta = saved


def test_philox_bit_exact():
    from oracle.rng import philox4x32_10, philox_randn
    n = 4 * 64 * 64
    out, raw = ops.philox_randn(12345, 3, n, DEV, want_raw=True)
    idx = np.arange(n, dtype=np.uint32)
    z = np.zeros(n, dtype=np.uint32)
    want = np.stack(philox4x32_10(np.full(n, 3, np.uint32), z, idx, z, np.full(n, 12345, np.uint32), z), 1)
    np.testing.assert_array_equal(raw.cpu().numpy().view(np.uint32), want)  # integer path: bit exact
    np.testing.assert_allclose(out.cpu().numpy(), philox_randn(12345, 3, n), rtol=0, atol=2e-6)  # Box-Muller: device libm


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,hh,ww,c,co", [(2, 64, 64, 128, 3), (1, 37, 50, 128, 3), (3, 5, 7, 64, 4), (1, 130, 33, 32, 1), (1, 256, 256, 128, 3),
                                          (2, 64, 64, 320, 4), (1, 19, 45, 96, 4), (1, 32, 32, 640, 2)])
def test_narrow_output_conv3x3_direct_kernel(n, hh, ww, c, co, dtype):
    """fmx_conv3x3_narrow (csrc/fmx_conv_narrow.hip, round 4): the VAE decoder's conv_out (128 -> 3 channels, /root/reference/backend/nn/vae.py:248-271) as a
    direct kernel -- the input patch of a 4 x 32 pixel tile staged once, zero padding through the descriptor's bounds check, the pad column of the
    [npix, 4] output written as zeros.  Against F.conv2d in fp32 and against the implicit-GEMM path; image sizes that are no multiple of the tile, images
    narrower than a tile, every supported channel count, both element types."""
    x = rnd(n, hh, ww, c, seed=300).to(dtype)
    wt = rnd(co, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=301).to(dtype)
    b = rnd(co, seed=302).to(dtype)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    out = torch.full((n * hh * ww, 4), float("nan"), dtype=dtype, device=DEV)      # a recycled buffer: every column must be written
    got = ops.conv3x3_narrow(x, wk, b, co, out=out)
    ref = F.conv2d(x.permute(0, 3, 1, 2).float(), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, co)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    close(got[:, :co], ref, tol, tol, f"direct 3x3 conv {c} -> {co}")
    assert bool((got[:, co:] == 0).all()), "pad columns must be zero (the VAE's overflow guard scans the whole buffer)"
    if dtype == torch.float16 and c % 64 == 0:     # (the implicit GEMM takes channel counts in multiples of 64)
        y = torch.zeros(n * hh * ww, 4, dtype=dtype, device=DEV)
        ops.conv_gemm(x, wk, co, kh=3, pad=1, bias=b, out=y, ld_out=4)
        close(got[:, :co], y[:, :co].float(), 1e-3, 1e-3, "direct kernel vs the implicit-GEMM path")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,hh,ww,cin,with_res", [(2, 24, 40, 128, True), (1, 16, 64, 256, False), (1, 9, 33, 64, True), (3, 8, 32, 128, False), (1, 64, 96, 128, True),
                                                  (1, 40, 31, 192, True)])
def test_groupnorm_silu_conv3x3_in_one_kernel(n, hh, ww, cin, with_res, dtype):
    """fmx_conv3x3_gn_silu (csrc/fmx_conv_patch.hip, round 6): norm -> swish -> conv of the VAE's ResnetBlock (/root/reference/backend/nn/vae.py:98-114) as ONE
    launch -- the 8 x 32-pixel tile's input patch normalised while it is staged in LDS.  The staged values are the tensor groupnorm() stores (same
    arithmetic), so against groupnorm() + conv_gemm() the result differs by fp32 summation order only; against F.conv2d of torch's own
    silu(group_norm(x)) at the element type's tolerance; the output statistics against sums over the stored tensor.  Image sizes that are no multiple of
    the tile, one to four channel chunks, with and without the residual, both element types."""
    cout = 128
    x = (rnd(n, hh, ww, cin, scale=1.5, seed=330) + 0.3).to(dtype)
    g, b = (1 + 0.1 * rnd(cin, seed=331)).to(dtype), (0.1 * rnd(cin, seed=332)).to(dtype)
    wt = rnd(cout, cin, 3, 3, scale=1 / math.sqrt(9 * cin), seed=333).to(dtype)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    bias = rnd(cout, seed=334).to(dtype)
    res = rnd(n * hh * ww, cout, seed=335).to(dtype) if with_res else None
    st_x = ops.groupnorm_stats(x)
    tiles = -(-hh // 8) * -(-ww // 32)
    part = torch.full((n, tiles, cout, 2), float("nan"), dtype=torch.float32, device=DEV)
    out = torch.full((n * hh * ww, cout), float("nan"), dtype=dtype, device=DEV)
    got, st = ops.conv3x3_gn_silu(x, g, b, 1e-6, wk, bias, residual=res, out=out, stats=st_x, stats_partial=part)
    assert st.nchunks == tiles and bool(torch.isfinite(got.float()).all())
    # the unfused pair on the same statistics
    gn = ops.groupnorm(x, g, b, 1e-6, silu=True, stats=st_x)
    two = ops.conv_gemm(gn, wk, cout, kh=3, pad=1, bias=bias, residual=res)
    # two kernels: another fp32 summation order over the K = 9 * cin products, and -- at these small sizes -- another summation order of the
    # statistics (groupnorm() folds its finalize into the apply blocks below 1 024 blocks, fmx_norm.hip GnFuse; the fused convolution runs the
    # stand-alone finalize): scale / shift differ in their last bit, so one normalised activation in ~10^4 lands on the other neighbour, and an
    # output moves by |w| x ulp(activation) <= ~5e-4 of the output's scale in fp16 (8 x that in bf16); at the VAE's sizes both paths use the same
    # finalize and the staged patch IS the stored tensor (tests/test_gpu_vae_sharp_parity.py holds that level layer by layer at 2e-4 rms)
    d = (got.float() - two.float()).abs()
    e = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    ulp = torch.maximum(two.float().abs(), torch.tensor(2.0 ** -14, device=DEV)).log2().floor().exp2() * e
    scale = float(two.float().pow(2).mean().sqrt())
    assert bool((d <= ulp + 0.5 * e * scale).all()), f"fused vs groupnorm + conv_gemm: max {float(((d - 0.5 * e * scale) / ulp).max()):.2f} ulp"
    assert float((d > ulp).float().mean()) < 0.02, "more than a rounding apart in a few elements only"
    ref = F.conv2d(F.silu(F.group_norm(x.permute(0, 3, 1, 2).float(), 32, g.float(), b.float(), 1e-6)), wt.float(), bias.float(), padding=1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + (res.float() if with_res else 0)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    close(got, ref, tol, tol, "fused groupnorm + silu + conv3x3 vs torch fp32")
    o = got.view(n, hh * ww, cout).double()
    want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
    torch.testing.assert_close(_partial_to_sums(st, n, cout), want, rtol=2e-5, atol=2e-3)
    # and the statistics feed the next fused launch (norm2 -> conv2 of the same block)
    g2, b2 = (1 + 0.1 * rnd(cout, seed=337)).to(dtype), (0.1 * rnd(cout, seed=338)).to(dtype)
    w2 = rnd(cout, 9 * cout, scale=0.03, seed=336).to(dtype)
    nxt, _ = ops.conv3x3_gn_silu(got.view(n, hh, ww, cout), g2, b2, 1e-6, w2, bias, stats=st, want_stats=False)
    nxt2 = ops.conv_gemm(ops.groupnorm(got.view(n, hh, ww, cout), g2, b2, 1e-6, silu=True, stats=st), w2, cout, kh=3, pad=1, bias=bias)
    close(nxt, nxt2.float(), 4 * e, 4 * e, "second fused launch on the first one's statistics")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,hh,ww,c,nout", [(2, 32, 32, 64, 320), (1, 16, 64, 128, 256), (2, 8, 32, 64, 128), (1, 32, 96, 64, 640), (3, 16, 32, 192, 64), (1, 64, 64, 256, 256)])
def test_upsample_conv_as_four_phase_convolutions(n, hh, ww, c, nout, dtype):
    """fmx_conv3x3_up2x (round 6): conv3x3(pad 1) of the x2 nearest-upsampled input (/root/reference/backend/nn/unet.py:340-355, nn/vae.py:42-57) as four
    2 x 2 convolutions on the input grid with tap-summed weights, each writing every second pixel of every second output row.  Against F.conv2d of
    F.interpolate in fp32; against the same four convolutions evaluated in fp32 on the SAME rounded tap sums (what is left is accumulation order and
    the output rounding); against the nine-tap implicit-GEMM path it replaces; the GroupNorm statistics of the scattered output against sums over it."""
    x = rnd(n, hh, ww, c, seed=340).to(dtype)
    wt = rnd(nout, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=341).to(dtype)
    wk = wt.permute(0, 2, 3, 1).reshape(nout, -1).contiguous()
    bias = rnd(nout, seed=342).to(dtype)
    w4 = ops.fold_up2x_weights(wk, c)
    assert ops.conv3x3_up2x_supported(x, nout) and not ops.conv3x3_up2x_eligible(x, nout)     # (too few tiles to be USED by the executors: see hipops)
    out, st = ops.conv3x3_up2x(x, w4, bias, nout)
    e = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    xu = F.interpolate(x.permute(0, 3, 1, 2).float(), scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xu, wt.float(), bias.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, nout)
    close(out, ref, 3 * e, 3 * e, "four phase convolutions vs conv3x3 of the upsampled input (fp32)")
    # the arithmetic it implements, in fp32 on the rounded tap sums
    same = torch.empty(n, nout, 2 * hh, 2 * ww, device=DEV)
    xl = x.permute(0, 3, 1, 2).float()
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        wp = w4[ph].float().view(nout, 2, 2, c).permute(0, 3, 1, 2)
        same[:, :, py::2, px::2] = F.conv2d(F.pad(xl, (1 - px, px, 1 - py, py)), wp, bias.float())
    same = same.permute(0, 2, 3, 1).reshape(-1, nout)
    d = (out.float() - same).abs()
    ulp = torch.maximum(same.abs(), torch.tensor(2.0 ** -14, device=DEV)).log2().floor().exp2() * e
    assert bool((d <= ulp + 1e-5 * float(same.pow(2).mean().sqrt())).all()), f"vs the same sums in fp32: max {float((d / ulp).max()):.2f} ulp"
    # the path it replaces: nine taps on the upsample-on-load address form (another rounding of the weights: the sums)
    old = ops.conv_gemm(x, wk, nout, kh=3, pad=1, up=(2 * hh, 2 * ww), bias=bias)
    close(out, old.float(), 3 * e, 3 * e, "four phase convolutions vs the nine-tap upsample-on-load path")
    assert st.nchunks == 4 * hh * ww // 256
    o = out.view(n, 4 * hh * ww, nout).double()
    want = torch.stack([o.sum(1), (o * o).sum(1)], -1)
    torch.testing.assert_close(_partial_to_sums(st, n, nout), want, rtol=2e-5, atol=2e-3)
    assert not ops.conv3x3_up2x_supported(x[:, :, :24].contiguous(), nout) and not ops.conv3x3_up2x_supported(x, nout, (2 * hh + 1, 2 * ww))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,hh,ww,c,co", [(2, 64, 64, 128, 3), (1, 37, 50, 128, 3), (1, 19, 45, 96, 4), (2, 64, 64, 320, 4), (1, 40, 33, 64, 2)])
def test_narrow_output_conv3x3_with_groupnorm_silu_in_its_staging(n, hh, ww, c, co, dtype):
    """fmx_conv3x3_narrow_gn_silu (round 6): the norm_out -> swish -> conv_out tail of the VAE decoder (/root/reference/backend/nn/vae.py:266-271) in one launch --
    against groupnorm() + conv3x3_narrow() on the same statistics (the staged patch holds the tensor groupnorm() stores; at these sizes the two paths finalise
    their statistics in different orders, see test_groupnorm_silu_conv3x3_in_one_kernel) and against torch fp32."""
    x = (rnd(n, hh, ww, c, scale=1.5, seed=350) + 0.3).to(dtype)
    g, b = (1 + 0.1 * rnd(c, seed=351)).to(dtype), (0.1 * rnd(c, seed=352)).to(dtype)
    wt = rnd(co, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=353).to(dtype)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    bias = rnd(co, seed=354).to(dtype)
    st = ops.groupnorm_stats(x)
    out = torch.full((n * hh * ww, 4), float("nan"), dtype=dtype, device=DEV)
    got = ops.conv3x3_narrow_gn_silu(x, g, b, 1e-6, wk, bias, co, out=out, stats=st)
    two = ops.conv3x3_narrow(ops.groupnorm(x, g, b, 1e-6, silu=True, stats=st), wk, bias, co)
    e = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    close(got[:, :co], two[:, :co].float(), 2 * e, 2 * e, "fused vs groupnorm + conv3x3_narrow")
    assert bool((got[:, co:] == 0).all()), "pad columns must be zero"
    ref = F.conv2d(F.silu(F.group_norm(x.permute(0, 3, 1, 2).float(), 32, g.float(), b.float(), 1e-6)), wt.float(), bias.float(), padding=1)
    close(got[:, :co], ref.permute(0, 2, 3, 1).reshape(-1, co), 3 * e, 3 * e, "fused groupnorm + silu + narrow conv3x3 vs torch fp32")


def test_narrow_output_conv3x3_contract():
    x = rnd(1, 8, 8, 80, seed=1)
    with pytest.raises(Exception, match="multiple of 32"):
        ops.conv3x3_narrow(x, rnd(3, 9 * 80, seed=2), None, 3)
    with pytest.raises(Exception, match="1..4 output"):
        ops.conv3x3_narrow(rnd(1, 8, 8, 64, seed=1), rnd(5, 9 * 64, seed=2), None, 5, out=torch.zeros(64, 8, dtype=torch.float16, device=DEV), ld_out=8)


@pytest.mark.parametrize("tile", [0, 6, 7])
@pytest.mark.parametrize("n,hh,ww,c,co,stats", [(2, 16, 16, 128, 320, True), (1, 13, 9, 64, 64, False), (2, 32, 32, 256, 256, True), (1, 7, 40, 128, 640, False)])
def test_conv_x2_nearest_upsample_on_load_fast_address_form(n, hh, ww, c, co, stats, tile):
    """The Upsample convolutions (backend/nn/unet.py:340-355 / backend/nn/vae.py:35-57: F.interpolate(scale 2, nearest) then a 3x3 convolution) on the per-tap
    bit-mask address form (csrc/fmx_gemm256p.hip FA = 2, round 4): source row of tap ky = (iy0 >> 1) + {0, iy0 & 1, 1}[ky].  Odd source sizes, sources narrower
    than a tile row, borders on all four sides, with and without the GroupNorm statistics of the output; against conv2d of the upsampled input in fp32 and
    against the small-tile kernel (general address form)."""
    x = rnd(n, hh, ww, c, seed=310)
    wt = rnd(co, c, 3, 3, scale=1 / math.sqrt(9 * c), seed=311)
    b = rnd(co, seed=312)
    wk = wt.permute(0, 2, 3, 1).reshape(co, -1).contiguous()
    ref = F.conv2d(F.interpolate(x.permute(0, 3, 1, 2).float(), scale_factor=2.0, mode="nearest"), wt.float(), b.float(), padding=1).permute(0, 2, 3, 1)
    if stats and (4 * hh * ww) % 256 == 0 and tile in (0, 6, 7):
        out, st = ops.conv_gemm(x, wk, co, kh=3, pad=1, up=(2 * hh, 2 * ww), bias=b, stats=True, force_tile=tile)
        o = out.view(n, -1, co).double()
        got = st.partial.reshape(-1)[:n * st.nchunks * co * 2].view(n, st.nchunks, co, 2).double().sum(1)   # dense [n][nchunks][co][2] at the buffer's head
        torch.testing.assert_close(got[..., 0], o.sum(1), rtol=1e-4, atol=5e-2)
        torch.testing.assert_close(got[..., 1], (o * o).sum(1), rtol=1e-4, atol=5e-2)
    else:
        out = ops.conv_gemm(x, wk, co, kh=3, pad=1, up=(2 * hh, 2 * ww), bias=b, force_tile=tile)
    close(out.reshape(ref.shape), ref, 3e-3, 3e-3, f"x2 upsample-on-load conv, tile {tile}")
    small = ops.conv_gemm(x, wk, co, kh=3, pad=1, up=(2 * hh, 2 * ww), bias=b, force_tile=1)
    close(out.reshape(ref.shape), small.reshape(ref.shape).float(), 1e-3, 1e-3, "fast address form vs the general one (128x128 tile)")
