"""Minimal `UnetPatcher` (reference: backend/patcher/unet.py, backend/patcher/base.py) -- the object protocol the call
surface needs from `sd_model.forge_objects.unet`: `.model` (KModel), `.model_options`, clone(), and the
ControlNet / concat / LoRA attributes, which are present but empty on the native path."""
import copy


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

    def clone(self):
        n = UnetPatcher(self.model, self.load_device, self.offload_device)
        n.model_options = copy.deepcopy(self.model_options)
        return n

    def has_online_lora(self):
        return False

    def list_controlnets(self):
        return []

    def memory_required(self, input_shape):
        return self.model.memory_required(input_shape)

    def set_model_unet_function_wrapper(self, wrapper):
        self.model_options["model_function_wrapper"] = wrapper  # rejected at sampling time (sampling_function.py)
