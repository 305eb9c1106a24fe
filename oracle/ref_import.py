"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Imports the *real* reference (lllyasviel/stable-diffusion-webui-forge, mounted
read-only at /root/reference) on CPU so that (a) the torch-fp32 restatement in
this directory can be pinned against it and (b) golden fixtures can be generated
(`oracle/make_golden.py`).  /root/reference does not exist on the GPU box, so
nothing that runs there may call `load_reference()`; tests that need it are
skipped when `reference_available()` is False.

Recipe: consume argv with --always-cpu/--attention-pytorch
(read by backend/args.py:61), put the reference and its packages_3rdparty on
sys.path (modules_forge/initialization.py:36) and stub the few third-party
modules the hot-path files touch at import time (diffusers, torchsde,
torchdiffeq, torchvision) -- none of them contributes arithmetic to the path.
"""
import math
import os
import sys
import types
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("FORGE_REFERENCE_ROOT", "/root/reference")

_loaded = None


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "backend"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    class ConfigMixin:  # diffusers.configuration_utils.ConfigMixin stand-in (unet.py:6, vae.py:4)
        config_name = "config.json"

    class FlowMatchEulerDiscreteScheduler:  # k_prediction.py:5,300 (Flux only)
        def time_shift(self, mu, sigma, t):
            return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)

    def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
        m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
        return image_seq_len * m + (base_shift - m * base_seq_len)

    if "diffusers" not in sys.modules:
        d = _stub("diffusers", FlowMatchEulerDiscreteScheduler=FlowMatchEulerDiscreteScheduler)
        d.configuration_utils = _stub("diffusers.configuration_utils", ConfigMixin=ConfigMixin,
                                      register_to_config=lambda f: f)
        d.pipelines = _stub("diffusers.pipelines")
        d.pipelines.flux = _stub("diffusers.pipelines.flux")
        d.pipelines.flux.pipeline_flux = _stub("diffusers.pipelines.flux.pipeline_flux",
                                               calculate_shift=calculate_shift)
    if "torchsde" not in sys.modules:
        _stub("torchsde")
    if "torchdiffeq" not in sys.modules:
        _stub("torchdiffeq", odeint=None)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.transforms.functional = _stub("torchvision.transforms.functional")


def load_reference():
    """Import the reference's hot-path modules; returns a namespace of them."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    saved_argv = sys.argv
    sys.argv = ["oracle", "--always-cpu", "--attention-pytorch"]
    for p in (os.path.join(REFERENCE_ROOT, "packages_3rdparty"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    _install_stubs()
    try:
        import backend.args  # noqa: F401  (parses argv now)
        import backend.attention as attention
        import backend.nn.unet as nn_unet
        import backend.nn.vae as nn_vae
        import backend.nn.flux as nn_flux
        import backend.modules.k_prediction as k_prediction
        import backend.modules.k_model as k_model
        import backend.sampling.sampling_function as sampling_function
        import backend.sampling.condition as condition
        import backend.patcher.unet as patcher_unet
        import k_diffusion.sampling as kd_sampling
        import k_diffusion.external as kd_external
    finally:
        sys.argv = saved_argv
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_ref_rng_philox", os.path.join(REFERENCE_ROOT, "modules", "rng_philox.py"))
    rng_philox = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rng_philox)

    # modules/sd_schedulers.py:10-15 replaces k_diffusion.sampling.to_d at import; that module
    # needs gradio, so the same replacement is applied here by hand.
    def to_d(x, sigma, denoised):
        return (x - denoised) / sigma
    kd_sampling.to_d = to_d

    _loaded = SimpleNamespace(attention=attention, nn_unet=nn_unet, nn_vae=nn_vae, nn_flux=nn_flux,
                              k_prediction=k_prediction, k_model=k_model,
                              sampling_function=sampling_function, condition=condition,
                              patcher_unet=patcher_unet, kd_sampling=kd_sampling,
                              kd_external=kd_external, rng_philox=rng_philox)
    return _loaded


# ---------------------------------------------------------------------------------------------
# Builders for reference objects (construction follows backend/loader.py:117-173 by hand).
# ---------------------------------------------------------------------------------------------

def build_ref_unet(unet_config, state_dict=None):
    import torch
    ref = load_reference()
    cfg = dict(unet_config)
    # the reference ctor mutates list arguments (unet.py:510-511 copies, but be safe)
    for k in ("transformer_depth", "transformer_depth_output", "num_res_blocks"):
        if isinstance(cfg.get(k), (list, tuple)):
            cfg[k] = list(cfg[k])
    net = ref.nn_unet.IntegratedUNet2DConditionModel(**cfg)
    if state_dict is not None:
        missing, unexpected = net.load_state_dict(state_dict, strict=True)
    net.storage_dtype = net.computation_dtype = torch.float32
    net.load_device = net.offload_device = net.initial_device = torch.device("cpu")
    return net.eval()


def build_ref_vae(vae_config, state_dict=None):
    ref = load_reference()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        vae = ref.nn_vae.IntegratedAutoencoderKL(**vae_config)
    if state_dict is not None:
        vae.load_state_dict(state_dict, strict=True)
    return vae.eval()


def build_ref_predictor():
    ref = load_reference()
    return ref.k_prediction.Prediction(prediction_type="epsilon", beta_schedule="linear",
                                       linear_start=0.00085, linear_end=0.012, timesteps=1000)


class RefDenoiser:
    """Stand-in for modules/sd_samplers_cfg_denoiser.py:156-228 that calls the reference's own
    `sampling_function` (the original module needs gradio).  Only the arithmetic-relevant part of
    CFGDenoiser.forward is kept: cond_composition [[(i, 1.0)]...] (prompt_parser.py:337-365 for plain
    prompts), cond_scale, return of the CFG-combined denoised."""

    def __init__(self, unet, predictor, seeds=(0,)):
        import contextlib, io
        ref = load_reference()
        with contextlib.redirect_stdout(io.StringIO()):
            self.patcher = ref.patcher_unet.UnetPatcher.from_model(unet, None, config=None, k_predictor=predictor)
        linker = ref.kd_external.ForgeScheduleLinker(predictor)
        linker.inner_model = SimpleNamespace(forge_objects=SimpleNamespace(unet=self.patcher))
        self.inner_model = linker
        self.p = SimpleNamespace(seeds=list(seeds))
        self.step = 0
        self.last = None

    def __call__(self, x, sigma, uncond, cond, cond_scale, s_min_uncond=0.0, image_cond=None):
        ref = load_reference()
        params = SimpleNamespace(x=x, sigma=sigma, text_cond=cond, text_uncond=uncond, image_cond=image_cond)
        comp = [[(i, 1.0)] for i in range(x.shape[0])]
        denoised, cond_pred, uncond_pred = ref.sampling_function.sampling_function(
            self, denoiser_params=params, cond_scale=cond_scale, cond_composition=comp)
        self.last = (denoised, cond_pred, uncond_pred)
        self.step += 1
        return denoised


class SdxlCond(dict):
    """dict-like cond with `advanced_indexing` (prompt_parser.py:271-291 DictWithShape contract,
    consumed at condition.py:133-136)."""

    def advanced_indexing(self, item):
        return SdxlCond({k: v[item] for k, v in self.items()})


def build_ref_flux(flux_config, state_dict=None):
    """Reference IntegratedFluxTransformer2DModel (backend/nn/flux.py:310) on CPU fp32 with the attributes loader.py:169-173 sets."""
    import torch
    ref = load_reference()
    net = ref.nn_flux.IntegratedFluxTransformer2DModel(**flux_config)
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    net.storage_dtype = net.computation_dtype = torch.float32
    net.load_device = net.offload_device = net.initial_device = torch.device("cpu")
    return net.eval()
