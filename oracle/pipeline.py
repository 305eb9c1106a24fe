"""TEST INFRASTRUCTURE ONLY (CPU oracle).

End-to-end txt2img restatement tying the pieces together the way the reference's call surface does:
  modules/processing.py:852-1040 (seeds seed+i :894, ImageRNG :944, p.sample :990, decode :1010-1013),
  :1342-1391 (x = rng.next(); sampler.sample), modules/sd_samplers_kdiffusion.py:81-134 (sigma choice),
  :196-244 (x * sigma_0 via noise_scaling :207; launch the k-diffusion loop on CFGDenoiser).
"""
import torch

from . import sampling
from .cfg import cfg_denoise
from .k_prediction import Predictor, apply_model
from .rng import ImageRNG
from .unet import unet_forward
from .vae import decode_first_stage, to_uint8_images


def get_sigmas(predictor, sampler_name, steps):
    # sd_samplers_kdiffusion.py:81-134 with scheduler "Automatic": per-sampler default and discard_next_to_last_sigma (:14-34)
    if sampler_name in sampling.SAMPLERS:
        (_, sched), discard = sampling.SAMPLERS[sampler_name], False
    else:
        _, sched, _, _, discard = sampling.SAMPLERS_EXTRA[sampler_name]
    n = steps + (1 if discard else 0)
    if sched == "karras":
        sig = sampling.get_sigmas_karras(n, predictor.sigmas[0].item(), predictor.sigmas[-1].item())
    else:
        sig = sampling.get_sigmas_linker(predictor, n)
    return torch.cat([sig[:-2], sig[-1:]]) if discard else sig


@torch.no_grad()
def txt2img_latents_on_schedule(unet_sd, unet_cfg, cond, uncond, seeds, height, width, sigmas, sampler_name="Euler", cfg_scale=7.0):
    """txt2img_latents with an explicit sigma schedule (p.scheduler != Automatic, sd_samplers_kdiffusion.py:88-127)."""
    return txt2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, height, width, len(sigmas) - 1, sampler_name, cfg_scale, sigmas=sigmas)[0]


@torch.no_grad()
def txt2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, sampler_name="Euler",
                    cfg_scale=7.0, noise_source="CPU", trace=None, sigmas=None, predictor=None):
    pred = predictor if predictor is not None else Predictor()
    b = len(seeds)
    rng = ImageRNG((unet_cfg["in_channels"], height // 8, width // 8), seeds, noise_source)
    x = rng.next()

    def unet_fn(xc, t, ctx, y):
        return unet_forward(unet_sd, unet_cfg, xc, t, ctx, y)

    def denoiser(xx, sigma):
        den, _, _ = cfg_denoise(lambda a, s, c, y: apply_model(unet_fn, pred, a, s, c, y), xx, sigma, uncond, cond, cfg_scale)
        if trace is not None:
            trace.append(den.clone())
        return den

    sigmas = get_sigmas(pred, sampler_name, steps) if sigmas is None else sigmas
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x))
    if sampler_name in sampling.SAMPLERS_EXTRA:
        fn, _, draws, ancestral, _ = sampling.SAMPLERS_EXTRA[sampler_name]
        return (fn(denoiser, x, sigmas, noise_fn=rng.next) if (draws or ancestral) else fn(denoiser, x, sigmas)), sigmas
    fn, _ = sampling.SAMPLERS[sampler_name]
    if sampler_name in ("Euler", "Euler a"):
        return fn(denoiser, x, sigmas, noise_fn=rng.next), sigmas
    return fn(denoiser, x, sigmas), sigmas


@torch.no_grad()
def txt2img_latents_more(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, sampler_name, cfg_scale=7.0, eta=0.0):
    """DDIM / DDIM CFG++ / PLMS (modules/sd_samplers_timesteps.py:121-149: x = noise, timesteps :62), LCM (sd_samplers_lcm.py) and DDPM
    (modules_forge/alter_samplers.py) with the same denoiser underneath."""
    pred = Predictor()
    rng = ImageRNG((unet_cfg["in_channels"], height // 8, width // 8), seeds, "CPU")
    x = rng.next()

    def unet_fn(xc, t, ctx, y):
        return unet_forward(unet_sd, unet_cfg, xc, t, ctx, y)

    def denoiser_full(xx, sigma):
        return cfg_denoise(lambda a, s, c, y: apply_model(unet_fn, pred, a, s, c, y), xx, sigma, uncond, cond, cfg_scale)

    if sampler_name in ("DDIM", "DDIM CFG++", "PLMS", "UniPC"):
        eps_model = sampling.EpsFromDenoiser(denoiser_full, 1.0 / (pred.sigmas ** 2.0 + 1.0))
        ts = sampling.timesteps_for(steps)
        if sampler_name == "UniPC":
            from . import unipc
            return unipc.sample_unipc(eps_model, x, len(ts), eps_model.acd)
        if sampler_name == "PLMS":
            return sampling.sample_plms(eps_model, x, ts)
        return sampling.sample_ddim(eps_model, x, ts, rng.next, eta=eta, cfgpp=sampler_name == "DDIM CFG++")
    if sampler_name == "LCM":
        sigmas = sampling.LcmSchedule(pred).get_sigmas(steps)
        fn = sampling.sample_lcm
    else:
        sigmas = sampling.get_sigmas_linker(pred, steps)
        fn = sampling.sample_ddpm
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x))
    return fn(lambda xx, s: denoiser_full(xx, s)[0], x, sigmas, rng.next)


@torch.no_grad()
def txt2img(unet_sd, unet_cfg, vae_sd, vae_cfg, cond, uncond, seeds, height, width, steps, **kw):
    lat, _ = txt2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, **kw)
    dec = decode_first_stage(vae_sd, lat, vae_cfg.get("scaling_factor", 0.18215), vae_cfg.get("shift_factor", 0.0) or 0.0)
    return lat, dec, to_uint8_images(dec)


def setup_img2img_steps(steps, denoising_strength, fix_steps=False):
    """modules/sd_samplers_common.py:24-33 (opts.img2img_fix_steps False by default)."""
    if fix_steps:
        requested = steps
        steps = int(requested / min(denoising_strength, 0.999)) if denoising_strength > 0 else 0
        return steps, requested - 1
    return steps, int(min(denoising_strength, 0.999) * steps)


@torch.no_grad()
def img2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, init_latent, steps, denoising_strength, sampler_name="Euler",
                    cfg_scale=7.0, noise_source="CPU", mask=None, nmask=None, mask_noise=None, fix_steps=False):
    """modules/processing.py:1843-1875 (Img2Img.sample) + modules/sd_samplers_kdiffusion.py:136-194 (sample_img2img:
    t_enc, sigma_sched = sigmas[steps - t_enc - 1:], xi = init + noise * sigma_sched[0], loop over sigma_sched) +
    modules/sd_samplers_cfg_denoiser.py:178-181, 204-213 (inpaint mask: noised original under the mask before the model,
    original under the mask after it).  `mask_noise(step)` stands for the torch.randn_like of :180."""
    pred = Predictor()
    rng = ImageRNG(tuple(init_latent.shape[1:]), seeds, noise_source)
    noise = rng.next()
    steps, t_enc = setup_img2img_steps(steps, denoising_strength, fix_steps)
    sigmas = get_sigmas(pred, sampler_name, steps)
    sigma_sched = sigmas[steps - t_enc - 1:]
    xi = pred.noise_scaling(sigma_sched[0], noise, init_latent)

    def unet_fn(xc, t, ctx, y):
        return unet_forward(unet_sd, unet_cfg, xc, t, ctx, y)

    step = [0]

    def denoiser(xx, sigma):
        if mask is not None:
            noisy = pred.noise_scaling(sigma[:, None, None, None], mask_noise(step[0]), init_latent)
            xx = xx * nmask + noisy * mask
        den, _, _ = cfg_denoise(lambda a, s, c, y: apply_model(unet_fn, pred, a, s, c, y), xx, sigma, uncond, cond, cfg_scale)
        if mask is not None:
            den = den * nmask + init_latent * mask
        step[0] += 1
        return den

    fn, _ = sampling.SAMPLERS[sampler_name]
    if sampler_name in ("Euler", "Euler a"):
        out = fn(denoiser, xi, sigma_sched, noise_fn=rng.next)
    else:
        out = fn(denoiser, xi, sigma_sched)
    if mask is not None:
        out = out * nmask + init_latent * mask  # processing.py:1865-1866
    return out, sigma_sched


@torch.no_grad()
def hires_latents(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, hr_scale=2.0, mode="bilinear", antialias=False,
                  denoising_strength=0.75, hr_second_pass_steps=0, hr_cfg=1.0, sampler_name="Euler", cfg_scale=7.0, hr_sampler_name=None):
    """modules/processing.py:1342-1391 + :1430-1536 for a latent upscaler: first pass -> F.interpolate (:1459) -> noise from a NEW ImageRNG
    with the same seeds (:1498-1499) -> sample_img2img with explicit steps (the `steps is not None` branch of setup_img2img_steps,
    sd_samplers_common.py:25-29) at cond_scale = hr_cfg (sd_samplers_cfg_denoiser.py:189-190)."""
    import torch.nn.functional as F
    first, _ = txt2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, sampler_name=sampler_name, cfg_scale=cfg_scale)
    size = (int(height * hr_scale) // 8, int(width * hr_scale) // 8)
    kw = {"antialias": antialias} if mode in ("bilinear", "bicubic") else {}
    up = F.interpolate(first, size=size, mode=mode, **kw)
    out, _ = img2img_latents(unet_sd, unet_cfg, cond, uncond, seeds, up, hr_second_pass_steps or steps, denoising_strength,
                             sampler_name=hr_sampler_name or sampler_name, cfg_scale=hr_cfg, fix_steps=True)
    return first, up, out


@torch.no_grad()
def txt2img_latents_controlnet(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, chain, cfg_scale=7.0):
    """Euler txt2img with a ControlNet chain (oracle.controlnet.Control) in the loop, as backend/sampling/sampling_function.py:220-268 wires it:
    per call the stacked [uncond ; cond] batch goes through get_control, the residuals into the UNet forward."""
    from .cfg import _cat, _ctx_y
    pred = Predictor()
    b = len(seeds)
    rng = ImageRNG((unet_cfg["in_channels"], height // 8, width // 8), seeds, "CPU")
    x = rng.next()
    ctx, y = _ctx_y(_cat(uncond, cond))

    def denoiser(xx, sigma):
        x2, s2 = torch.cat([xx, xx]), torch.cat([sigma, sigma])
        to = {"cond_or_uncond": [1, 0], "sigmas": sigma, "cond_mark": torch.tensor([1.0] * b + [0.0] * b)}
        control = chain.get_control(pred, x2, s2, ctx, y, 2, to)
        out = apply_model(lambda xc, t, c, yy: unet_forward(unet_sd, unet_cfg, xc, t, c, yy, control=control), pred, x2, s2, ctx, y)
        un, co = out[:b], out[b:]
        return un + (co - un) * cfg_scale
    sigmas = get_sigmas(pred, "Euler", steps)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x))
    return sampling.sample_euler(denoiser, x, sigmas, noise_fn=rng.next)


@torch.no_grad()
def txt2img_latents_general_cfg(unet_sd, unet_cfg, cond, uncond, composition, seeds, height, width, steps, cfg_scale, options=None):
    """Euler txt2img through cfg_denoise_general (AND-composed prompts, cfg function hooks, model_function_wrapper)."""
    from .cfg import cfg_denoise_general
    pred = Predictor()
    rng = ImageRNG((unet_cfg["in_channels"], height // 8, width // 8), seeds, "CPU")
    x = rng.next()
    comp = composition or [[(i, 1.0)] for i in range(len(seeds))]

    def model_fn(xx, ss, ctx, y):
        return apply_model(lambda xc, t, c, yy: unet_forward(unet_sd, unet_cfg, xc, t, c, yy), pred, xx, ss, ctx, y)
    sigmas = get_sigmas(pred, "Euler", steps)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x))
    return sampling.sample_euler(lambda xx, s: cfg_denoise_general(model_fn, xx, s, uncond, cond, comp, cfg_scale, options), x, sigmas, noise_fn=rng.next)


@torch.no_grad()
def txt2img_latents_inpaint_model(unet_sd, unet_cfg, cond, uncond, seeds, height, width, steps, image_cond, cfg_scale=7.0):
    """Euler txt2img on an inpainting UNet (in_channels 9): image_cond [B, 5, h, w] goes to BOTH CFG halves as c_concat
    (sampling_function.py:342-350) and is concatenated to the scaled latent inside apply_model (k_model.py:38-39)."""
    pred = Predictor()
    b = len(seeds)
    rng = ImageRNG((4, height // 8, width // 8), seeds, "CPU")
    x = rng.next()

    def model(a, s, c, y):
        return apply_model(lambda xc, t, cc, yy: unet_forward(unet_sd, unet_cfg, xc, t, cc, yy), pred, a, s, c, y,
                           c_concat=torch.cat([image_cond] * (a.shape[0] // b)))
    sigmas = get_sigmas(pred, "Euler", steps)
    x = pred.noise_scaling(sigmas[0], x, torch.zeros_like(x))
    return sampling.sample_euler(lambda xx, s: cfg_denoise(model, xx, s, uncond, cond, cfg_scale)[0], x, sigmas, noise_fn=rng.next)
