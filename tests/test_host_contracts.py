"""Host-side contracts that need no GPU: a clone of the UNet patcher never leaks hooks into its parent (the reference
deep-copies model_options, backend/patcher/base.py:83), and the per-iteration conditioning slice tolerates the absent unconditional
batch of a cfg_scale 1 job (setup_conds leaves uc = None, as the reference does)."""
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
