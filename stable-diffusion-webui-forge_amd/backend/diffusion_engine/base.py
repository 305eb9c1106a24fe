"""`ForgeObjects` / engine object protocol -- mirror of backend/diffusion_engine/base.py:7-20 and the attributes of
sd15.py:19-84 / sdxl.py:22-138 the call surface touches: forge_objects{,_original,_after_applying_lora}, is_sdxl,
decode_first_stage, get_learned_conditioning (absent: text encoders are out of scope -> cond tensors are supplied)."""
import torch

from ..modules.k_prediction import Prediction
from ..nn.unet import IntegratedUNet2DConditionModel
from ..nn.vae import IntegratedAutoencoderKL
from ..patcher.unet import UnetPatcher
from ..patcher.vae import VAE


class ForgeObjects:
    def __init__(self, unet, clip, vae, clipvision=None):
        self.unet, self.clip, self.vae, self.clipvision = unet, clip, vae, clipvision

    def shallow_copy(self):
        return ForgeObjects(self.unet, self.clip, self.vae, self.clipvision)


class ForgeDiffusionEngine:
    def __init__(self, unet, vae, is_sdxl=False):
        predictor = Prediction(prediction_type="epsilon", beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
        self.forge_objects = ForgeObjects(unet=UnetPatcher.from_model(unet, k_predictor=predictor), clip=None,
                                          vae=VAE(vae) if vae is not None else None)
        self.forge_objects_original = self.forge_objects.shallow_copy()
        self.forge_objects_after_applying_lora = self.forge_objects.shallow_copy()
        self.is_sdxl = is_sdxl
        self.is_sd1 = not is_sdxl
        self.is_inpaint = False
        self.device = unet.device

    @torch.inference_mode()
    def decode_first_stage(self, x):
        """sd15.py:80-84: process_out -> vae.decode (NHWC [0,1]) -> NCHW [-1,1]."""
        vae = self.forge_objects.vae
        sample = vae.first_stage_model.process_out(x)
        sample = vae.decode(sample).movedim(-1, 1) * 2.0 - 1.0
        return sample.to(x)

    @torch.inference_mode()
    def encode_first_stage(self, x):
        """sd15.py:75-78: x NCHW in [-1, 1] -> process_in(vae.encode(NHWC in [0, 1]))."""
        vae = self.forge_objects.vae
        sample = vae.encode(x.movedim(1, -1) * 0.5 + 0.5)
        sample = vae.first_stage_model.process_in(sample)
        return sample.to(x)

    def get_learned_conditioning(self, prompt):
        raise NotImplementedError("text encoders are out of scope (SURVEY.md §2.2): pass cond tensors to the processing object")


def build_engine(unet_config, unet_state_dict, vae_config=None, vae_state_dict=None, device="cuda"):
    unet = IntegratedUNet2DConditionModel(unet_config, unet_state_dict, device=device)
    vae = IntegratedAutoencoderKL(vae_config, vae_state_dict, device=device) if vae_config is not None else None
    return ForgeDiffusionEngine(unet, vae, is_sdxl=unet_config.get("adm_in_channels") is not None)


class FluxEngine:
    """Object protocol of backend/diffusion_engine/flux.py:27-120 that the call surface touches (transformer only: text encoders
    and the 16-channel VAE are outside this round's scope)."""

    def __init__(self, transformer, seq_len):
        from ..modules.k_model import KModelFlux
        from ..modules.k_prediction import PredictionFlux
        from ..patcher.unet import UnetPatcher
        patcher = UnetPatcher(KModelFlux(transformer, PredictionFlux(seq_len=seq_len)), transformer.device, transformer.device)
        self.forge_objects = ForgeObjects(unet=patcher, clip=None, vae=None)
        self.forge_objects_original = self.forge_objects.shallow_copy()
        self.forge_objects_after_applying_lora = self.forge_objects.shallow_copy()
        self.is_sdxl = self.is_sd1 = self.is_inpaint = False
        self.is_flux = True
        self.use_distilled_cfg_scale = True
        self.latent_channels = transformer.in_channels // 4
        self.device = transformer.device


def build_flux_engine(flux_config, state_dict, width, height, device="cuda"):
    from ..nn.flux import IntegratedFluxTransformer2DModel
    net = IntegratedFluxTransformer2DModel(flux_config, state_dict, device=device)
    return FluxEngine(net, seq_len=(height // 16) * (width // 16))
