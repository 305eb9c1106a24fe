"""Host-side contracts that need no GPU: a clone of the UNet patcher never leaks hooks into its parent (the reference
deep-copies model_options, backend/patcher/base.py:83), and the per-iteration conditioning slice tolerates the absent unconditional
batch of a cfg_scale 1 job (setup_conds leaves uc = None, as the reference does); which ControlNet chains may run inside the captured graph
(round 2: a host decision per link, patcher/controlnet.py graph_entry)."""
import pytest
import torch

import forge_amd  # noqa: F401
from forge_amd.backend.patcher.unet import UnetPatcher
from forge_amd.modules import processing
from forge_amd.modules.prompt_parser import DictWithShape


class _Model:
    device = torch.device("cpu")

    def memory_required(self, shape):
        return 0


def test_clone_does_not_share_hook_containers_with_its_parent():
    parent = UnetPatcher(_Model())
    f0, f1, f2 = (lambda *a: a[0]), (lambda *a: a[0]), (lambda *a: a[0])
    parent.set_model_patch(f0, "attn1_patch")
    parent.append_transformer_option("block_modifiers", f0)
    parent.model_options["transformer_options"]["patches_replace"] = {"attn1": {("input", 1): f0}}
    child = parent.clone()
    child.set_model_patch(f1, "attn1_patch")
    child.append_transformer_option("block_modifiers", f1)
    child.model_options["transformer_options"]["patches_replace"]["attn1"][("middle", 0)] = f2
    child.model_options["transformer_options"]["patches"]["attn1_patch"].append(f2)
    pt, ct = parent.model_options["transformer_options"], child.model_options["transformer_options"]
    assert pt["patches"]["attn1_patch"] == [f0] and pt["block_modifiers"] == [f0] and list(pt["patches_replace"]["attn1"]) == [("input", 1)]
    assert ct["patches"]["attn1_patch"] == [f0, f1, f2] and ct["block_modifiers"] == [f0, f1]
    assert ct["patches"]["attn1_patch"][0] is f0, "hook callables themselves stay shared objects"
    grandchild = child.clone()
    grandchild.append_transformer_option("block_modifiers", f2)
    assert ct["block_modifiers"] == [f0, f1]


def test_slice_cond_handles_every_conditioning_shape_and_none():
    assert processing._slice_cond(None, 0, 2) is None
    t = torch.arange(12.0).reshape(4, 3)
    assert torch.equal(processing._slice_cond(t, 1, 3), t[1:3])
    d = DictWithShape({"crossattn": torch.zeros(4, 77, 8), "vector": torch.ones(4, 6)})
    s = processing._slice_cond(d, 2, 4)
    assert isinstance(s, DictWithShape) and s["crossattn"].shape[0] == 2 and s["vector"].shape[0] == 2


# ---- which ControlNet chains KModel may capture in its graph (patcher/controlnet.py graph_entry; host decision, no kernels) ------------------------
class _FakeCtx:
    key, serial = ("k",), 1


class _FakeControlExecutor:
    """Stands in for nn/cnets/cldm.ControlNet: the three things graph_entry asks of it."""
    arena_epoch = 0

    def __init__(self):
        self.hints = 0

    def forward_static(self, *a):
        raise AssertionError("graph_entry must not run the trunk")

    def prepare_context(self, context, y):
        return _FakeCtx()

    def hint_for_batch(self, hint, bu):
        self.hints += 1
        return torch.zeros(bu, hint.shape[2] // 8, hint.shape[3] // 8, 4)


def _plain_link(strength=0.8, rng=(0.0, 1.0), pooling=False):
    from forge_amd.backend.patcher import controlnet as pc

    class P:   # predictor stand-in: percent -> sigma, linear from 10 down to 0
        sigma_data = 1.0

    cn = pc.ControlNet(_FakeControlExecutor(), global_average_pooling=pooling)
    cn.set_cond_hint(torch.rand(1, 3, 64, 64), strength, rng)
    cn.pre_run(type("M", (), {"predictor": P()})(), lambda pct: 10.0 * (1.0 - pct))
    return cn


def test_controlnet_graph_entry_accepts_plain_links_and_reports_activity():
    ctx = torch.zeros(4, 77, 8)
    cn = _plain_link(rng=(0.2, 0.7))                                   # active for sigma in [3, 8]
    e = cn.graph_entry(5.0, 4, 8, 8, ctx, None, 2)
    assert e is not None and e["active"] and e["cn"] is cn and e["gh"].shape[0] == 4
    v = e["valid"]()
    assert v[0] == ("k",) and v[1] == 1 and v[3] == 0.8
    for sigma in (9.0, 2.0):                                           # before the start / after the end of the range: in the chain, inactive
        e = cn.graph_entry(sigma, 4, 8, 8, ctx, None, 2)
        assert e is not None and not e["active"] and e["gh"] is None
    # the prepared hint survives cleanup() (per source image, on the executor): the next job finds the same tensor, hence the same guided hint
    h1 = cn._prepare_hint(4, 8, 8, "cpu", 2)
    cn.cleanup()
    assert cn.cond_hint is None
    cn.pre_run(type("M", (), {"predictor": type("P", (), {"sigma_data": 1.0})()})(), lambda pct: 10.0 * (1.0 - pct))
    assert cn._prepare_hint(4, 8, 8, "cpu", 2) is h1
    cn.cond_hint = None
    cn.cond_hint_original = cn.cond_hint_original.clone()              # another image (even with equal content): prepared again
    assert cn._prepare_hint(4, 8, 8, "cpu", 2) is not h1


@pytest.mark.parametrize("why", ["modifier", "wrapper", "pooling", "weighting", "foreign_model", "subclass_get_control"])
def test_controlnet_graph_entry_declines_links_that_need_python_per_step(why):
    from forge_amd.backend.patcher import controlnet as pc
    ctx = torch.zeros(4, 77, 8)
    cn = _plain_link(pooling=(why == "pooling"))
    if why == "modifier":
        cn.transformer_options = {"controlnet_conditioning_modifiers": [lambda *a: a[1:]]}
    elif why == "wrapper":
        cn.transformer_options = {"controlnet_model_function_wrapper": lambda **kw: None}
    elif why == "weighting":
        cn.advanced_sigma_weighting = lambda s: s
    elif why == "foreign_model":
        cn.control_model = object()
    elif why == "subclass_get_control":
        class Custom(pc.ControlNet):
            def get_control(self, *a):
                return None
        c2 = Custom(cn.control_model)
        cn.copy_to(c2)
        c2.timestep_range, c2.model_sampling_current = cn.timestep_range, cn.model_sampling_current
        cn = c2
    assert cn.graph_entry(5.0, 4, 8, 8, ctx, None, 2) is None


def test_attached_groupnorm_statistics_die_with_an_unannounced_in_place_edit():
    """statistics a producer left on a tensor object are ignored once torch's version counter says the tensor was edited in place
    (a hook holding the NCHW view), instead of silently normalising with stale sums."""
    from forge_amd import hipops
    t = torch.zeros(2, 4, 4, 8)
    st = hipops.GnStats(torch.zeros(2, 1, 8, 2), 1)
    hipops.attach_stats(t, st)
    assert hipops._attached_stats(t) is st
    t.permute(0, 3, 1, 2).add_(1.0)          # the NCHW view a Python hook would hold
    assert hipops._attached_stats(t) is None and t._fmx_gn_stats is None
    hipops.attach_stats(t, st)
    hipops.clear_stats(t)
    assert hipops._attached_stats(t) is None
    hipops.attach_stats(t, None)
    assert hipops._attached_stats(t) is None


def test_arena_neighbours_do_not_invalidate_each_others_statistics():
    """all arena tensors are views of one uint8 buffer and share torch's version counter; an in-place op on one of them must not
    drop the GroupNorm statistics attached to another (it silently added a statistics pass); an announced write (clear_stats) still does."""
    from forge_amd import hipops
    buf = torch.zeros(4096, dtype=torch.uint8)
    a = buf[:1024].view(torch.float16).view(2, 4, 4, 16)
    b = buf[2048:3072].view(torch.float16).view(2, 4, 4, 16)
    st = hipops.GnStats(torch.zeros(2, 1, 16, 2), 1)
    hipops.attach_stats(a, st)
    b.zero_()
    buf[3072:].fill_(1)
    assert hipops._attached_stats(a) is st
    hipops.clear_stats(a)
    assert hipops._attached_stats(a) is None


def test_executor_serials_are_never_reused():
    """captured ControlNet graphs are keyed on the executors' serial numbers, which a later executor cannot inherit the
    way it can inherit a freed object's id()."""
    import itertools
    from forge_amd.backend.nn import unet
    a, b = next(unet._EXEC_SERIAL), next(unet._EXEC_SERIAL)
    assert b == a + 1 and isinstance(unet._EXEC_SERIAL, itertools.count)


def test_two_dimensional_bool_mask_keeps_the_reference_reading():
    """a 2-D bool mask is ALWAYS the per-batch key mask in the reference's attention_basic ('b ... -> b (...)',
    /root/reference/backend/attention.py:74-78), also when B == Nq makes it look like an SDPA [Nq, Nk] mask: warned about, not refused."""
    from forge_amd.backend import attention
    m = torch.ones(4, 4, dtype=torch.bool)
    with pytest.warns(UserWarning, match="key mask"):
        v = attention._mask_view(m, 4, 2, 4, 4)
    assert tuple(v.shape) == (4, 1, 1, 4)
    assert tuple(attention._mask_view(torch.ones(3, 5, dtype=torch.bool), 3, 2, 7, 5).shape) == (3, 1, 1, 5)      # [B, Nk], B != Nq: key mask
    assert tuple(attention._mask_view(torch.zeros(7, 5), 3, 2, 7, 5).shape) == (1, 1, 7, 5)                        # float [Nq, Nk]: additive per-query
    with pytest.raises(ValueError):
        attention._mask_view(torch.zeros(6, 5), 3, 2, 7, 5)
