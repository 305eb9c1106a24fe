"""CPU: the host side of regional / time-ranged conditioning (backend/sampling/sampling_function.py get_area_and_mult,
_regional_cond_uncond_batch) against the reference's calc_cond_uncond_batch on a closed-form model (tests/golden/regional_conds.pt, made by
oracle/make_golden.py gen_regional).  The two latent-sized kernels on that path are replaced HERE, for this test only, by their torch
one-liners (they are checked on the GPU in tests/test_gpu_kernels.py; the product itself has no such route)."""
import pytest
import torch

from forge_amd import hipops
from forge_amd.backend.sampling import sampling_function as sf
from forge_amd.backend.sampling.condition import ConditionCrossAttn
from oracle.make_golden import RegionalToyModel, regional_case

from conftest import load_golden


@pytest.fixture(autouse=True)
def torch_kernels(monkeypatch):
    def lincomb(srcs, coefs, out=None):
        r = sum(float(c) * s for c, s in zip(coefs, srcs))
        return r if out is None else out.copy_(r)

    def blend_masked(a, a_mask, b, b_mask, out=None):
        r = a * a_mask + b * b_mask
        return r if out is None else out.copy_(r)
    monkeypatch.setattr(hipops, "lincomb", lincomb)
    monkeypatch.setattr(hipops, "blend_masked", blend_masked)


def _build(entries, ctx):
    out = []
    for e in entries:
        d = {k: v for k, v in e.items() if k != "ctx"}
        d["model_conds"] = {"c_crossattn": ConditionCrossAttn(ctx[e["ctx"]])}
        out.append(d)
    return out


def test_regional_and_time_ranged_entries_match_the_reference():
    g = load_golden("regional_conds.pt")
    x, ctx, cond, uncond, sigmas = regional_case()
    model = RegionalToyModel()
    for s in sigmas:
        t = torch.full((x.shape[0],), s)
        fused, c_out, u_out = sf.calc_cond_uncond_batch(model, _build(cond, ctx), _build(uncond, ctx), x, t, {}, cond_scale=7.0)
        assert fused is None
        torch.testing.assert_close(c_out, g[s][0], rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(u_out, g[s][1], rtol=2e-6, atol=2e-6)


def test_area_weights_feather_only_interior_sides():
    x = torch.zeros(1, 1, 20, 24)
    area, mult = sf.get_area_and_mult({"area": (16, 12, 0, 0), "strength": 2.0}, x, 1.0)     # top-left corner: bottom and right sides feathered
    assert area == (16, 12, 0, 0) and float(mult[0, 0, 0, 0]) == 2.0
    assert float(mult[0, 0, 15, 0]) == pytest.approx(2.0 / 8) and float(mult[0, 0, 0, 11]) == pytest.approx(2.0 / 8)
    assert float(mult[0, 0, 15, 11]) == pytest.approx(2.0 / 64)
    assert sf.get_area_and_mult({"timestep_start": 3.0}, x, 5.0) is None and sf.get_area_and_mult({"timestep_end": 3.0}, x, 1.0) is None
    assert sf.get_area_and_mult({"timestep_start": 3.0, "timestep_end": 1.0}, x, 2.0) is not None
    with pytest.raises(ValueError):
        sf.get_area_and_mult({"mask": torch.ones(1, 10, 24)}, x, 1.0)


def test_entries_outside_their_window_leave_a_zero_prediction():
    x, ctx, _, _, _ = regional_case()
    t = torch.full((x.shape[0],), 9.0)
    cond = _build([dict(ctx=0, timestep_start=5.0)], ctx)     # inactive at sigma 9
    _, c_out, u_out = sf.calc_cond_uncond_batch(RegionalToyModel(), cond, _build([dict(ctx=1)], ctx), x, t, {}, cond_scale=7.0)
    assert float(c_out.abs().max()) == 0.0 and float(u_out.abs().max()) > 0.0     # 0 / 1e-37 in the reference (:155-159, 284-288)
