"""`AlterSampler` -- mirror of modules_forge/alter_samplers.py: the samplers of backend/modules/k_diffusion_extra.py (DDPM) behind
the KDiffusionSampler machinery."""
from ..backend.modules import k_diffusion_extra
from ..modules import sd_samplers_common, sd_samplers_kdiffusion


class AlterSampler(sd_samplers_kdiffusion.KDiffusionSampler):
    def __init__(self, sd_model, sampler_name):
        self.sampler_name = sampler_name
        self.unet = sd_model.forge_objects.unet
        super().__init__(getattr(k_diffusion_extra, f"sample_{sampler_name}"), sd_model, None)


def build_constructor(sampler_name):
    return lambda m: AlterSampler(m, sampler_name)


samplers_data_alter = [sd_samplers_common.SamplerData("DDPM", build_constructor(sampler_name="ddpm"), ["ddpm"], {})]
