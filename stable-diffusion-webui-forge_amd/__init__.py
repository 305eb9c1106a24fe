"""MI355X-native drop-in for Forge's txt2img denoising hot path (CFGDenoiser -> UNet -> sampler -> VAE decode).

Import as `forge_amd` (see /forge_amd.py at the repo root: the on-disk directory keeps the name the build
contract asks for, `stable-diffusion-webui-forge_amd`, which is not a valid Python identifier).
Sub-packages mirror the reference's module names for this path (`backend.attention`, `backend.nn.unet`,
`k_diffusion.sampling`, `modules.sd_samplers_kdiffusion`, ...).  Everything numeric runs through the
C-ABI library built from `csrc/` (see include/fmx.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
