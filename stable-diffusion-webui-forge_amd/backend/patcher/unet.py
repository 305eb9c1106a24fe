"""Minimal `UnetPatcher` (reference: backend/patcher/unet.py, backend/patcher/base.py) -- the object protocol the call
surface needs from `sd_model.forge_objects.unet`: `.model` (KModel), `.model_options`, clone(), and the
ControlNet / concat / LoRA attributes, which are present but empty on the native path."""


class UnetPatcher:
    def __init__(self, model, load_device=None, offload_device=None):
        self.model = model
        self.model_options = {"transformer_options": {}}
        self.controlnet_linked_list = None
        self.extra_concat_condition = None
        self.extra_preserved_memory_during_sampling = 0
        self.extra_model_patchers_during_sampling = []
        self.lora_patches = {}
        self.load_device = load_device
        self.offload_device = offload_device
        self.current_device = load_device

    @classmethod
    def from_model(cls, model, diffusers_scheduler=None, config=None, k_predictor=None):
        from ..modules.k_model import KModel
        return cls(KModel(model, k_predictor), load_device=model.device, offload_device=model.device)

    @staticmethod
    def _copy_containers(o):
        """Copy of the dict / list STRUCTURE of model_options; the leaves (hook callables, tensors, modules) stay shared.  The reference
        deep-copies model_options in ModelPatcher.clone (backend/patcher/base.py:83); a copy of the containers is what that buys on this
        path: `set_model_patch` / `append_transformer_option` on a clone append to the clone's own lists, never to its parent's."""
        if isinstance(o, dict):
            return {k: UnetPatcher._copy_containers(v) for k, v in o.items()}
        if isinstance(o, list):
            return [UnetPatcher._copy_containers(v) for v in o]
        return o

    def clone(self):
        n = UnetPatcher(self.model, self.load_device, self.offload_device)
        n.model_options = self._copy_containers(self.model_options)
        n.model_options.setdefault("transformer_options", {})
        n.controlnet_linked_list = self.controlnet_linked_list
        n.extra_concat_condition = self.extra_concat_condition
        return n

    def has_online_lora(self):
        return False

    def add_patched_controlnet(self, cnet):
        cnet.set_previous_controlnet(self.controlnet_linked_list)  # patcher/unet.py:68-71
        self.controlnet_linked_list = cnet

    def list_controlnets(self):
        results, pointer = [], self.controlnet_linked_list
        while pointer is not None:
            results.append(pointer)
            pointer = pointer.previous_controlnet
        return results

    def memory_required(self, input_shape):
        return self.model.memory_required(input_shape)

    # ---- hook setters (backend/patcher/base.py:146-200, backend/patcher/unet.py:91-172) ------------------------------------------------
    def append_transformer_option(self, k, v, ensure_uniqueness=False):
        to = self.model_options.setdefault("transformer_options", {})
        if k not in to:
            to[k] = []
        if ensure_uniqueness and v in to[k]:
            return
        to[k].append(v)

    def set_transformer_option(self, k, v):
        self.model_options.setdefault("transformer_options", {})[k] = v

    def set_model_patch(self, patch, name):
        to = self.model_options["transformer_options"]
        if "patches" not in to:
            to["patches"] = {}
        to["patches"][name] = to["patches"].get(name, []) + [patch]

    def set_model_patch_replace(self, patch, name, block_name, number, transformer_index=None):
        to = self.model_options["transformer_options"].copy()
        to["patches_replace"] = dict(to.get("patches_replace", {}))
        to["patches_replace"][name] = dict(to["patches_replace"].get(name, {}))
        block = (block_name, number, transformer_index) if transformer_index is not None else (block_name, number)
        to["patches_replace"][name][block] = patch
        self.model_options["transformer_options"] = to

    def set_model_attn1_patch(self, patch):
        self.set_model_patch(patch, "attn1_patch")

    def set_model_attn2_patch(self, patch):
        self.set_model_patch(patch, "attn2_patch")

    def set_model_attn1_replace(self, patch, block_name, number, transformer_index=None):
        self.set_model_patch_replace(patch, "attn1", block_name, number, transformer_index)

    def set_model_attn2_replace(self, patch, block_name, number, transformer_index=None):
        self.set_model_patch_replace(patch, "attn2", block_name, number, transformer_index)

    def set_model_attn1_output_patch(self, patch):
        self.set_model_patch(patch, "attn1_output_patch")

    def set_model_attn2_output_patch(self, patch):
        self.set_model_patch(patch, "attn2_output_patch")

    def set_model_input_block_patch(self, patch):
        self.set_model_patch(patch, "input_block_patch")

    def set_model_input_block_patch_after_skip(self, patch):
        self.set_model_patch(patch, "input_block_patch_after_skip")

    def set_model_output_block_patch(self, patch):
        self.set_model_patch(patch, "output_block_patch")

    def add_block_modifier(self, modifier, ensure_uniqueness=False):
        self.append_transformer_option("block_modifiers", modifier, ensure_uniqueness)

    def add_block_inner_modifier(self, modifier, ensure_uniqueness=False):
        self.append_transformer_option("block_inner_modifiers", modifier, ensure_uniqueness)

    def set_group_norm_wrapper(self, wrapper):
        self.set_transformer_option("group_norm_wrapper", wrapper)

    def set_model_replace_all(self, patch, target="attn1"):
        for block_name in ["input", "middle", "output"]:
            for number in range(16):
                for transformer_index in range(16):
                    self.set_model_patch_replace(patch, target, block_name, number, transformer_index)

    def set_model_unet_function_wrapper(self, wrapper):
        self.model_options["model_function_wrapper"] = wrapper  # rejected at sampling time (sampling_function.py)
