"""txt2img / img2img job orchestration -- mirror of modules/processing.py for the hot path:
`StableDiffusionProcessingTxt2Img` (dataclass fields :123-216 that the path reads), `.sample` (:1342-1391),
`process_images` / `process_images_inner` (:815-1158: seeds seed+i :894, ImageRNG :944, p.sample :990,
decode_latent_batch :1010, clamp + *255 + uint8 truncation :1012-1040), `decode_latent_batch` (:628).
`StableDiffusionProcessingImg2Img` (:1637-1875): init images -> VAE encode -> `sample_img2img` on the tail of the sigma
schedule, optional latent inpaint mask (`mask` keeps the original, `nmask` = 1 - mask is repainted; :1822-1832, :1865-1866).
The PIL / cv2 mask preparation of the UI (:1692-1745) is outside the path: the latent-resolution mask tensor is an input.
Prompts are replaced by ready conditioning tensors (`p.c`, `p.uc`): the text encoders are out of scope.
"""
from dataclasses import dataclass, field
from typing import Any, List, Optional

import numpy as np
import torch

from . import rng, sd_samplers, shared


@dataclass
class Processed:
    images: List[Any]
    latents: torch.Tensor
    seeds: List[int]
    decoded: Optional[torch.Tensor] = None


def decode_first_stage(model, x):
    return model.decode_first_stage(x)


def decode_latent_batch(model, batch, target_device=None, check_for_nans=False):
    samples = decode_first_stage(model, batch)
    if target_device is not None:
        samples = samples.to(target_device)
    return [x for x in samples]


@dataclass
class StableDiffusionProcessingTxt2Img:
    sd_model: Any = None
    c: Any = None                 # cond:   tensor [B,T,D] or {"crossattn","vector"} (prompt_parser.DictWithShape), or schedule objects
    uc: Any = None                # uncond: same
    prompt: Any = None            # or prompt strings (str, or a list of n_iter * batch_size strings): encoded by setup_conds through the
    negative_prompt: Any = ""     #   engine's text encoders (needs attach_text_encoders(..., tokenizer_l=...))
    seed: int = -1
    subseed: int = -1             # variation seed (processing.py:132-135, rng.py:133-146): noise = slerp(subseed_strength, noise(seed), noise(subseed))
    subseed_strength: float = 0.0
    seed_resize_from_h: int = -1  # seed resize (rng.py:131,148-160): the noise of this (pixel) size is centred into the noise of the image size
    seed_resize_from_w: int = -1
    sampler_name: str = "Euler"
    scheduler: Optional[str] = None
    batch_size: int = 1
    n_iter: int = 1
    steps: int = 20
    cfg_scale: float = 7.0
    width: int = 512
    height: int = 512
    eta: Optional[float] = None
    s_min_uncond: float = 0.0
    s_churn: float = 0.0
    s_tmin: float = 0.0
    s_tmax: float = float("inf")
    s_noise: float = 1.0
    is_hr_pass: bool = False
    # hires fix (processing.py:1187-1204); only the latent upscalers of shared.latent_upscale_modes are built (image-space upscalers are
    # separate networks outside the path)
    enable_hr: bool = False
    denoising_strength: float = 0.75
    hr_scale: float = 2.0
    hr_upscaler: Optional[str] = None
    hr_second_pass_steps: int = 0
    hr_resize_x: int = 0
    hr_resize_y: int = 0
    hr_sampler_name: Optional[str] = None
    hr_scheduler: Optional[str] = None
    hr_cfg: float = 1.0
    hr_c: Any = None              # hires-pass conditioning (defaults to c / uc: `hr_prompt == ''` -> the first-pass prompt, :1554-1558)
    hr_uc: Any = None
    hr_upscale_to_x: int = 0
    hr_upscale_to_y: int = 0
    truncate_x: int = 0
    truncate_y: int = 0
    latent_scale_mode: Any = None
    do_decode: bool = True
    disable_progress: bool = True
    sampler_noise_scheduler_override: Any = None
    scripts: Any = None
    extra_generation_params: dict = field(default_factory=dict)
    # filled while running
    seeds: List[int] = field(default_factory=list)
    all_seeds: List[int] = field(default_factory=list)
    rng: Any = None
    sampler: Any = None
    iteration: int = 0

    def txt2img_image_conditioning(self, x, width=None, height=None):
        """processing.py:103-119: inpainting models get [all-ones mask | latent of an all-0.5 image]; others a dummy 1x1 zero tensor."""
        if getattr(self.sd_model, "is_inpaint", False):
            from .sd_samplers_common import images_tensor_to_samples
            gray = torch.ones(x.shape[0], 3, height or self.height, width or self.width, device=x.device) * 0.5
            lat = images_tensor_to_samples(gray, None, self.sd_model)
            return torch.nn.functional.pad(lat, (0, 0, 0, 0, 1, 0), value=1.0).to(x.dtype).contiguous()
        return x.new_zeros(x.shape[0], 5, 1, 1)

    def inpainting_image_conditioning(self, source_image, latent_image, image_mask=None, round_image_mask=True, inpainting_mask_weight=1.0):
        """processing.py:320-358: source_image [B, 3, H, W] in [-1, 1], image_mask tensor [1 or B, 1, H, W] in [0, 1] (1 = repaint) ->
        [mask at latent resolution | latent of the masked image] = the 5 concat channels of an inpainting UNet."""
        mask = image_mask if image_mask is not None else source_image.new_ones(1, 1, *source_image.shape[-2:])
        mask = mask.to(device=source_image.device, dtype=source_image.dtype)
        if round_image_mask and image_mask is not None:
            mask = torch.round(mask)
        cond_image = torch.lerp(source_image, source_image * (1.0 - mask), inpainting_mask_weight)
        cond_image = self.sd_model.encode_first_stage(cond_image)
        mask = torch.nn.functional.interpolate(mask, size=latent_image.shape[-2:]).expand(cond_image.shape[0], -1, -1, -1)
        return torch.cat([mask, cond_image], dim=1).contiguous()

    def img2img_image_conditioning(self, source_image, latent_image, image_mask=None, round_image_mask=True):
        if getattr(self.sd_model, "is_inpaint", False):  # processing.py:360-378
            return self.inpainting_image_conditioning(source_image.float(), latent_image, image_mask=image_mask, round_image_mask=round_image_mask)
        return latent_image.new_zeros(latent_image.shape[0], 5, 1, 1)

    def setup_conds(self):
        """processing.py:489-506: prompts -> p.c (MulticondLearnedConditioning: AND parts, prompt-editing schedules) and p.uc (schedules; None
        at cfg_scale == 1).  Only when no ready conditioning was supplied."""
        from . import prompt_parser
        if self.c is not None or self.prompt is None:
            return
        total = self.batch_size * self.n_iter
        prompts = list(self.prompt) if isinstance(self.prompt, (list, tuple)) else [self.prompt] * total
        negs = list(self.negative_prompt) if isinstance(self.negative_prompt, (list, tuple)) else [self.negative_prompt] * len(prompts)
        if len(prompts) != len(negs):
            raise RuntimeError(f"Received a different number of prompts ({len(prompts)}) and negative prompts ({len(negs)})")
        config = sd_samplers.find_sampler_config(self.sampler_name)
        total_steps = config.total_steps(self.steps) if config else self.steps
        mk = lambda texts, neg: prompt_parser.SdConditioning(texts, width=self.width, height=self.height, is_negative_prompt=neg)
        self.uc = None if self.cfg_scale == 1 else prompt_parser.get_learned_conditioning(self.sd_model, mk(negs, True), total_steps)
        self.c = prompt_parser.get_multicond_learned_conditioning(self.sd_model, mk(prompts, False), total_steps)

    def calculate_target_resolution(self):
        """processing.py:1246-1273."""
        if self.hr_resize_x == 0 and self.hr_resize_y == 0:
            self.hr_upscale_to_x, self.hr_upscale_to_y = int(self.width * self.hr_scale), int(self.height * self.hr_scale)
        elif self.hr_resize_y == 0:
            self.hr_upscale_to_x, self.hr_upscale_to_y = self.hr_resize_x, self.hr_resize_x * self.height // self.width
        elif self.hr_resize_x == 0:
            self.hr_upscale_to_x, self.hr_upscale_to_y = self.hr_resize_y * self.width // self.height, self.hr_resize_y
        else:
            target_w, target_h = self.hr_resize_x, self.hr_resize_y
            if self.width / self.height < self.hr_resize_x / self.hr_resize_y:
                self.hr_upscale_to_x, self.hr_upscale_to_y = self.hr_resize_x, self.hr_resize_x * self.height // self.width
            else:
                self.hr_upscale_to_x, self.hr_upscale_to_y = self.hr_resize_y * self.width // self.height, self.hr_resize_y
            self.truncate_x = (self.hr_upscale_to_x - target_w) // 8
            self.truncate_y = (self.hr_upscale_to_y - target_h) // 8

    def init_hr(self):
        """processing.py:1275-1341, the part that decides anything: scheduler default, upscaler -> latent mode, target size."""
        from . import latent_upscale
        if self.hr_scheduler is None:
            self.hr_scheduler = self.scheduler
        modes = latent_upscale.latent_upscale_modes
        self.latent_scale_mode = modes.get(self.hr_upscaler, None) if self.hr_upscaler is not None else modes[latent_upscale.latent_upscale_default_mode]
        if self.latent_scale_mode is None:
            raise NotImplementedError(f"hires upscaler '{self.hr_upscaler}': only the latent upscalers {list(modes)} are on the native path")
        self.calculate_target_resolution()

    def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
        self.sampler = sd_samplers.create_sampler(self.sampler_name, self.sd_model)
        x = self.rng.next()
        self.sd_model.forge_objects = self.sd_model.forge_objects_after_applying_lora.shallow_copy()
        samples = self.sampler.sample(self, x, conditioning, unconditional_conditioning,
                                      image_conditioning=self.txt2img_image_conditioning(x))
        if not self.enable_hr:
            return samples
        return self.sample_hr_pass(samples, None, seeds, subseeds, subseed_strength, prompts)

    def sample_hr_pass(self, samples, decoded_samples, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
        """processing.py:1430-1536 for the latent upscalers: resize the first-pass latent, fresh noise from a new ImageRNG with the same
        seeds, img2img pass with the hires sampler / scheduler / CFG / conds.  Returns the hires LATENT (the reference returns it decoded,
        :1531; here process_images_inner decodes it like any other sample)."""
        from . import latent_upscale
        if shared.state.interrupted:
            return samples
        self.is_hr_pass = True
        try:
            self.sampler = sd_samplers.create_sampler(self.hr_sampler_name or self.sampler_name, self.sd_model)
            samples = latent_upscale.interpolate(samples.contiguous(), (self.hr_upscale_to_y // 8, self.hr_upscale_to_x // 8),
                                                 mode=self.latent_scale_mode["mode"], antialias=self.latent_scale_mode["antialias"])
            image_conditioning = self.txt2img_image_conditioning(samples)
            ty, tx = self.truncate_y, self.truncate_x
            samples = samples[:, :, ty // 2:samples.shape[2] - (ty + 1) // 2, tx // 2:samples.shape[3] - (tx + 1) // 2].contiguous()
            self.rng = rng.ImageRNG(tuple(samples.shape[1:]), self.seeds, subseeds=getattr(self, "subseeds", None), subseed_strength=self.subseed_strength,
                                    seed_resize_from_h=self.seed_resize_from_h, seed_resize_from_w=self.seed_resize_from_w, device=samples.device)   # :1495
            noise = self.rng.next()
            lo = self.iteration * self.batch_size
            hr_c = _slice_cond(self.hr_c, lo, lo + self.batch_size) if self.hr_c is not None else self._first_pass_conds[0]
            hr_uc = _slice_cond(self.hr_uc, lo, lo + self.batch_size) if self.hr_uc is not None else self._first_pass_conds[1]
            if self.hr_cfg == 1:
                hr_uc = None  # :1586-1588
            self.sd_model.forge_objects = self.sd_model.forge_objects_after_applying_lora.shallow_copy()
            return self.sampler.sample_img2img(self, samples, noise, hr_c, hr_uc, steps=self.hr_second_pass_steps or self.steps,
                                               image_conditioning=image_conditioning)
        finally:
            self.is_hr_pass = False


@dataclass
class StableDiffusionProcessingImg2Img(StableDiffusionProcessingTxt2Img):
    init_images: Any = None            # tensor [B, 3, H, W] in [0, 1] (the reference converts PIL images to this, :1772-1777)
    init_latent: Any = None            # or an already encoded latent [B, lc, H/8, W/8]
    denoising_strength: float = 0.75
    latent_mask: Any = None            # [B or 1, 1 or lc, H/8, W/8] in [0, 1]: 1 = repaint (the reference's `latmask`, :1823-1831)
    inpainting_fill: int = 1           # "masked content" (:1781-1840): 0 = fill, 1 = original, 2 = latent noise, 3 = latent nothing
    initial_noise_multiplier: float = 1.0
    mask: Any = None
    nmask: Any = None
    image_conditioning: Any = None
    mask_noise_source: Any = None      # test hook, see CFGDenoiser.mask_noise_source
    image_mask: Any = None             # pixel-space mask tensor [1 or B, 1, H, W] for inpainting MODELS (their c_concat), 1 = repaint

    def init(self, all_seeds=None):
        """:1684-1842 for tensor inputs: VAE-encode the init images, build mask / nmask at latent resolution."""
        from .sd_samplers_common import images_tensor_to_samples
        dev = self.sd_model.device
        if self.inpainting_fill not in (0, 1, 2, 3):
            raise ValueError(f"inpainting_fill {self.inpainting_fill}: 0 = fill, 1 = original, 2 = latent noise, 3 = latent nothing (processing.py:1781-1840)")
        masked = self.latent_mask is not None
        if self.init_latent is None:
            if self.init_images is None:
                raise ValueError("img2img needs init_images or init_latent")
            if masked and self.inpainting_fill != 1:
                # :1781-1785 -- every "masked content" mode but `original` first bleeds the surroundings into the masked region of the IMAGE
                from . import masking
                pm = getattr(self, "image_mask", None)
                self.init_images = masking.fill_tensor(self.init_images, pm if pm is not None else self.latent_mask)
            self.init_latent = images_tensor_to_samples(self.init_images, None, self.sd_model)
        elif masked and self.inpainting_fill == 0:
            raise ValueError("inpainting_fill = 0 ('fill') works on the init IMAGE (masking.fill); this job handed over a ready init_latent")
        self.init_latent = self.init_latent.to(device=dev, dtype=torch.float32).contiguous()
        if masked and self.inpainting_fill == 2 and self.init_latent.shape[0] == 1 and self.batch_size > 1:
            # :1797-1799 -- the reference repeats a single init image to batch_size BEFORE the fill, so 'latent noise' draws one row per image from
            # all_seeds[0:batch_size]; a 1-row latent would hand every image of the batch image 0's noise (and, in a sharded job, each rank the noise
            # of ITS first image: a result that depends on the sharding)
            self.init_latent = self.init_latent.expand(self.batch_size, -1, -1, -1).contiguous()
        if self.latent_mask is not None:
            latmask = self.latent_mask.to(device=dev, dtype=torch.float32)
            if latmask.dim() == 3:
                latmask = latmask[:, None]
            latmask = latmask.expand(self.init_latent.shape[0], self.init_latent.shape[1], -1, -1).contiguous()
            self.mask = 1.0 - latmask       # :1830
            self.nmask = latmask            # :1831
            if self.inpainting_fill == 2:   # :1834-1836 'latent noise': the masked latent starts from the job's own seeded noise
                if all_seeds is None:
                    raise ValueError("inpainting_fill = 2 ('latent noise') draws create_random_tensors(shape, all_seeds[:B]): init() needs all_seeds")
                nb = self.init_latent.shape[0]
                noise = rng.ImageRNG(tuple(self.init_latent.shape[1:]), list(all_seeds[0:nb]), device=dev).next().to(device=dev, dtype=torch.float32)
                self.init_latent = (self.init_latent * self.mask + noise * self.nmask).contiguous()
            elif self.inpainting_fill == 3:  # :1838-1840 'latent nothing'
                self.init_latent = (self.init_latent * self.mask).contiguous()
        if getattr(self.sd_model, "is_inpaint", False):  # :1842: conditioning from the (pixel-space) source image and mask
            if self.init_images is None:
                raise ValueError("an inpainting model needs init_images (the masked image is VAE-encoded for its conditioning)")
            self.image_conditioning = self.img2img_image_conditioning(self.init_images.to(dev).float() * 2.0 - 1.0, self.init_latent,
                                                                      getattr(self, "image_mask", None))
        else:
            self.image_conditioning = self.init_latent.new_zeros(self.init_latent.shape[0], 5, 1, 1)  # non-inpaint models (:376-378)

    def sample(self, conditioning, unconditional_conditioning, seeds, subseeds=None, subseed_strength=0.0, prompts=None):
        from .. import hipops as ops
        self.sampler = sd_samplers.create_sampler(self.sampler_name, self.sd_model)
        x = self.rng.next()
        if self.initial_noise_multiplier != 1.0:
            x = x * self.initial_noise_multiplier
        self.sd_model.forge_objects = self.sd_model.forge_objects_after_applying_lora.shallow_copy()
        lo = self.iteration * self.batch_size
        init = self.init_latent[lo:lo + self.batch_size] if self.init_latent.shape[0] > self.batch_size else self.init_latent
        full_mask, full_nmask = self.mask, self.nmask
        if self.mask is not None and self.mask.shape[0] > self.batch_size:
            self.mask, self.nmask = full_mask[lo:lo + self.batch_size].contiguous(), full_nmask[lo:lo + self.batch_size].contiguous()
        try:
            samples = self.sampler.sample_img2img(self, init, x, conditioning, unconditional_conditioning, image_conditioning=self.image_conditioning)
            if self.mask is not None:
                samples = ops.blend_masked(samples.contiguous(), self.nmask, init.contiguous(), self.mask)  # :1865-1866
        finally:
            self.mask, self.nmask = full_mask, full_nmask
        return samples


def _slice_cond(c, a, b):
    from .prompt_parser import MulticondLearnedConditioning
    if c is None:   # no unconditional batch: setup_conds leaves uc = None at cfg_scale 1 (modules/processing.py:480-483 of the reference)
        return None
    if isinstance(c, dict):
        return type(c)({k: v[a:b] for k, v in c.items()})
    if isinstance(c, MulticondLearnedConditioning):
        return MulticondLearnedConditioning((b - a,), c.batch[a:b])
    return c[a:b]  # tensor, or list of per-image schedules


def process_images(p) -> Processed:
    if hasattr(p, "setup_conds"):
        p.setup_conds()
    shared.sd_model = p.sd_model  # the reference's p.sd_model IS shared.sd_model (processing.py:252-258); schedulers read is_sdxl from it
    return process_images_inner(p)


def _job_seeds(p, total):
    """processing.py:889-899: (all_seeds, all_subseeds) of a job of `total` images; -1 draws a random base."""
    seed = int(p.seed) if p.seed is not None and int(p.seed) != -1 else int(np.random.randint(0, 2 ** 31 - 1))
    subseed = int(p.subseed) if p.subseed is not None and int(p.subseed) != -1 else int(np.random.randint(0, 2 ** 31 - 1))
    return ([seed + (i if p.subseed_strength == 0 else 0) for i in range(total)],   # :894: a variation batch shares ONE seed ...
            [subseed + i for i in range(total)])                                      # ... and varies the subseed (:896-899)


@torch.inference_mode()
def process_images_inner(p, seed_plan=None) -> Processed:
    """seed_plan = (all_seeds, all_subseeds) fixes the per-image seeds from outside: a rank of a sharded job runs ITS images of the global
    job with the seeds the single-process job would have given them (process_images_sharded)."""
    total = p.batch_size * p.n_iter
    p.all_seeds, p.all_subseeds = seed_plan if seed_plan is not None else _job_seeds(p, total)
    assert len(p.all_seeds) == total and len(p.all_subseeds) == total
    dev = p.sd_model.device
    lc = p.sd_model.forge_objects.vae.latent_channels if p.sd_model.forge_objects.vae is not None else getattr(p.sd_model, "latent_channels", 4)
    images, lat_all, dec_all = [], [], []
    shared.state.interrupted = False
    if isinstance(p, StableDiffusionProcessingImg2Img):
        p.init(p.all_seeds)
    elif getattr(p, "enable_hr", False):
        p.init_hr()
    for n in range(p.n_iter):
        p.iteration = n
        lo, hi = n * p.batch_size, (n + 1) * p.batch_size
        p.seeds = p.all_seeds[lo:hi]
        p.subseeds = p.all_subseeds[lo:hi]
        p.rng = rng.ImageRNG((lc, p.height // 8, p.width // 8), p.seeds, subseeds=p.subseeds, subseed_strength=p.subseed_strength,
                             seed_resize_from_h=p.seed_resize_from_h, seed_resize_from_w=p.seed_resize_from_w, device=dev)   # :944
        c, uc = _slice_cond(p.c, lo, hi), _slice_cond(p.uc, lo, hi)
        p._first_pass_conds = (c, uc)
        samples = p.sample(conditioning=c, unconditional_conditioning=uc, seeds=p.seeds)
        lat_all.append(samples)
        if not p.do_decode or p.sd_model.forge_objects.vae is None:
            continue
        x = torch.stack(decode_latent_batch(p.sd_model, samples, target_device=None)).float()
        dec_all.append(x)
        x = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)
        for xs in x:
            arr = 255.0 * np.moveaxis(xs.cpu().numpy(), 0, 2)
            images.append(arr.astype(np.uint8))  # truncation, as the reference (:1039-1040)
    return Processed(images=images, latents=torch.cat(lat_all), seeds=p.all_seeds,
                     decoded=torch.cat(dec_all) if dec_all else None)


# ---- one job over the GPUs of a node ------------------------------------------------------------------------------------------------------
def _to_u8(images, device):
    """list of HxWx3 uint8 arrays -> [b, H, W, 3] uint8 tensor on `device` (empty list: a [0]-tensor the gather pads)"""
    if not images:
        return None
    return torch.from_numpy(np.stack(images)).to(device)


_I2I_FIELDS = ("init_images", "init_latent", "latent_mask", "image_mask")


def _check_per_image(name, cnd, total):
    from .prompt_parser import MulticondLearnedConditioning
    if cnd is None:
        return
    if torch.is_tensor(cnd):
        n = cnd.shape[0]
    elif isinstance(cnd, dict):
        n = next(iter(cnd.values())).shape[0]
    elif isinstance(cnd, MulticondLearnedConditioning):
        n = len(cnd.batch)
    else:
        n = len(cnd)
        if any(isinstance(e, str) for e in cnd):
            raise NotImplementedError(f"p.{name} holds strings: the sharded entry broadcasts ENCODED conditionings (tensors, dicts, prompt-editing schedules); "
                                      f"hand prompts over as p.prompt and they are encoded on the owner")
    if n != total:
        raise ValueError(f"p.{name} describes {n} images, the job has {total}")


def process_images_sharded(p, gather_images=True, dst=0) -> Processed:
    """`process_images` for a job whose images are spread over the ranks of the default process group (one process per GPU; backend "nccl" =
    RCCL over xGMI, "gloo" in the CPU tests).  It splits the reference's batch loop (modules/processing.py:924-1012) ACROSS ranks instead of
    walking it on one device: images are independent, so

      1. rank `dst` owns the job: its conditioning (p.c / p.uc for all batch_size * n_iter images: tensors, {"crossattn", "vector"} dicts, or the
         reference's schedule objects -- a MulticondLearnedConditioning / per-image ScheduledPromptConditioning lists, i.e. prompt editing and
         AND-composition, modules/prompt_parser.py:294-365; a tensor shared by several images travels once), for an img2img job its init images /
         init latents / masks, and its seed are broadcast -- the other ranks call this function with the same scalar parameters and those fields None;
      2. of every iteration's `batch_size` images rank r samples (and decodes) the contiguous share `shard_range(batch_size, r, world)` with the
         seeds / subseeds the single-process job gives those images (seed + global index), so the result does not depend on the sharding;
      3. latents (fp32) and, with `gather_images`, the decoded uint8 images are gathered on rank `dst` in global order (`dist.gather` of equal,
         padded shards: only `dst` receives).

    Nothing is exchanged inside the step loop.  Returns, on rank `dst`, the `Processed` of the whole job (`decoded` fp32 tensors stay local
    and are not gathered: 12 MB per 1024^2 image against 3 MB as uint8); on every other rank the `Processed` of its own share.  With one
    rank (no process group) it is `process_images`."""
    from .. import distributed as fdist
    import torch.distributed as dist
    rank, ws = fdist.world()
    if not fdist.group_active():
        return process_images(p)
    dev = p.sd_model.device
    B, total = p.batch_size, p.batch_size * p.n_iter
    # ---- 1. job header + conditioning from the owner.  Everything that can fail on the owner BEFORE the first collective (prompt encoding, the
    #      argument checks) runs inside the try: the header then carries the error and EVERY rank raises it -- an owner that raised on its own
    #      would leave the other ranks blocked in the broadcast for ever. --------------------------------------------------------------------------
    header = [None]
    is_i2i = isinstance(p, StableDiffusionProcessingImg2Img)
    flats = {}
    if rank == dst:
        try:
            if p.prompt is not None and p.c is None:
                p.setup_conds()      # prompts are encoded once, on the rank that owns the job
            # what travels: the conditionings -- tensors, {"crossattn", "vector"} dicts, or the reference's schedule objects (a MulticondLearnedConditioning
            # / per-image ScheduledPromptConditioning lists: prompt editing, AND-composition) -- and, for img2img, the per-image inputs of the job.
            # Flattened HERE so that a leaf the entry cannot broadcast is refused before the first collective.
            for name in ("c", "uc", "hr_c", "hr_uc"):
                try:
                    flats[name] = fdist.flatten_tree(getattr(p, name, None))
                except NotImplementedError as e:
                    raise NotImplementedError(f"p.{name}: {e}") from None
                _check_per_image(name, getattr(p, name, None), total)
            if is_i2i:
                if p.init_images is None and p.init_latent is None:
                    raise ValueError("img2img needs init_images or init_latent")
                for name in _I2I_FIELDS:
                    flats[name] = fdist.flatten_tree(getattr(p, name, None))
            header = [("ok", _job_seeds(p, total))]
        except Exception as e:   # noqa: BLE001 -- re-raised on every rank below
            header = [("error", type(e).__name__, str(e))]
    dist.broadcast_object_list(header, src=dst)
    if header[0][0] == "error":
        _, ename, emsg = header[0]
        raise (NotImplementedError if ename == "NotImplementedError" else RuntimeError)(f"process_images_sharded, owner rank {dst}: {ename}: {emsg}")
    _, (all_seeds, all_subseeds) = header[0]
    got = {name: fdist.broadcast_tree(getattr(p, name, None) if rank == dst else None, dev, src=dst, flat=flats.get(name))
           for name in (("c", "uc", "hr_c", "hr_uc") + (_I2I_FIELDS if is_i2i else ()))}
    c, uc, hr_c, hr_uc = got["c"], got["uc"], got["hr_c"], got["hr_uc"]
    lo, hi = fdist.shard_range(B, rank, ws)
    mine = [n * B + i for n in range(p.n_iter) for i in range(lo, hi)]          # global indices of this rank's images, iteration-major

    def take(cnd):
        return fdist.take_images(cnd, mine)

    def take_i2i(t):
        """a per-image input of an img2img job: one row per image of the job, one per image of an iteration (the same images every iteration), or one
        shared row"""
        if t is None or t.shape[0] == 1:
            return t
        if t.shape[0] == total:
            return take(t)
        if t.shape[0] == B:
            return t[lo:hi].contiguous()
        raise ValueError(f"an img2img input with {t.shape[0]} rows in a job of {p.n_iter} x {B} images")

    # ---- 2. this rank's share through the ordinary single-device job.  A rank whose job fails (out of memory, a kernel error) still takes part in
    #      the next collective and reports the failure there, so that every rank raises instead of the healthy ones waiting for ever ----------------
    local = None
    failure = None
    lat_local = u8_local = None
    want_images = bool(gather_images and p.do_decode and p.sd_model.forge_objects.vae is not None)
    if hi > lo:
        import copy
        try:
            q = copy.copy(p)
            q.batch_size = hi - lo
            q.c, q.uc, q.hr_c, q.hr_uc = take(c), take(uc), take(hr_c), take(hr_uc)
            q.prompt = None
            if is_i2i:   # this rank's init images / latents / masks: an img2img job is sharded by init image like a txt2img job by noise seed
                for name in _I2I_FIELDS:
                    setattr(q, name, take_i2i(got[name]))
            shared.sd_model = p.sd_model
            local = process_images_inner(q, seed_plan=([all_seeds[i] for i in mine], [all_subseeds[i] for i in mine]))
            p.sampler, p.rng = q.sampler, q.rng
            # what this rank contributes to the gathers is prepared INSIDE the try as well (an out-of-memory on the 8 x 1024^2 image
            # copy after the try would have left the other ranks in the collective)
            lat_local = local.latents.to(dev).float().contiguous()
            u8_local = _to_u8(local.images, dev) if want_images else None
        except Exception as e:   # noqa: BLE001
            failure = f"rank {rank}: {type(e).__name__}: {e}"
            local = lat_local = u8_local = None
    p.all_seeds, p.all_subseeds = all_seeds, all_subseeds
    # ---- 3. gather on the owner ---------------------------------------------------------------------------------------------------------
    # a rank with no image of this job (batch_size < world) still takes part in the gather: it learns the per-image shapes from the others
    shapes = [None] * ws
    mine_msg = ("error", failure) if failure else (None if local is None else (tuple(lat_local.shape[1:]), None if u8_local is None else tuple(u8_local.shape[1:])))
    dist.all_gather_object(shapes, mine_msg)
    errors = [sh[1] for sh in shapes if sh is not None and sh[0] == "error"]
    if errors:
        raise RuntimeError("process_images_sharded: " + "; ".join(errors))
    lat_shape, img_shape = next(sh for sh in shapes if sh is not None)
    if lat_local is None:
        lat_local = torch.zeros((0,) + lat_shape, device=dev)
    lat = fdist.gather_batch(lat_local, B, p.n_iter, dst=dst)
    images = None
    if want_images:
        if u8_local is None:
            u8_local = torch.zeros((0,) + img_shape, dtype=torch.uint8, device=dev)
        got = fdist.gather_batch(u8_local, B, p.n_iter, dst=dst)
        if rank == dst:
            images = [a for a in got.cpu().numpy()]
    if rank != dst:
        return local if local is not None else Processed(images=[], latents=lat_local, seeds=[])
    return Processed(images=images if images is not None else [], latents=lat, seeds=all_seeds, decoded=None)
