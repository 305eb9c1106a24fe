"""`ForgeObjects` / engine object protocol -- mirror of backend/diffusion_engine/base.py:7-20 and the attributes of
sd15.py:19-84 / sdxl.py:22-138 the call surface touches: forge_objects{,_original,_after_applying_lora}, is_sdxl,
decode_first_stage, encode_first_stage, get_learned_conditioning (from token batches; cond tensors may also be supplied directly)."""
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
    def __init__(self, unet, vae, is_sdxl=False, prediction_type="epsilon", ztsnr=False):
        predictor = Prediction(prediction_type=prediction_type, beta_schedule="linear", linear_start=0.00085, linear_end=0.012, timesteps=1000)
        if ztsnr:  # zero-terminal-SNR checkpoints carry a 'ztsnr' key (loader.py:462); their sigma table is rescaled (k_prediction.py:48-63)
            from ..modules.k_prediction import rescale_zero_terminal_snr_sigmas
            predictor.set_sigmas(rescale_zero_terminal_snr_sigmas(predictor.sigmas))
        self.forge_objects = ForgeObjects(unet=UnetPatcher.from_model(unet, k_predictor=predictor), clip=None,
                                          vae=VAE(vae) if vae is not None else None)
        self.forge_objects_original = self.forge_objects.shallow_copy()
        self.forge_objects_after_applying_lora = self.forge_objects.shallow_copy()
        self.is_sdxl = is_sdxl
        self.is_sd1 = not is_sdxl
        self.is_inpaint = getattr(unet, "concat_channels", 0) == 5  # huggingface_guess `inpaint_model()`: in_channels 9 (base.py:28)
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

    # ---- text conditioning (sd15.py:19-73, sdxl.py:22-120), from TOKEN batches: tokenisation is host-side string work ----------
    def attach_text_encoders(self, clip_l, clip_g=None, tokenizer_l=None, tokenizer_g=None, embeddings_l=None, embeddings_g=None):
        """clip_l / clip_g: forge_amd.backend.nn.clip.IntegratedCLIP.  Engine options as the reference constructs them.  With tokenizers
        (CLIPTokenizer objects of the user's install) `get_learned_conditioning` also takes prompt STRINGS (`SdConditioning([...])`)."""
        from ..text_processing.classic_engine import ClassicTextProcessingEngine as _Engine
        tl, tg = dict(tokenizer=tokenizer_l, embeddings=embeddings_l), dict(tokenizer=tokenizer_g, embeddings=embeddings_g)
        if self.is_sdxl:
            if clip_g is None:
                raise ValueError("SDXL needs both text encoders")
            self.text_processing_engine_l = _Engine(clip_l, embedding_key="clip_l", text_projection=False, minimal_clip_skip=2, clip_skip=2,
                                                    return_pooled=False, final_layer_norm=False, **tl)
            self.text_processing_engine_g = _Engine(clip_g, embedding_key="clip_g", text_projection=True, minimal_clip_skip=2, clip_skip=2,
                                                    return_pooled=True, final_layer_norm=False, **tg)
        else:
            self.text_processing_engine = _Engine(clip_l, embedding_key="clip_l", text_projection=False, minimal_clip_skip=1, clip_skip=1,
                                                  return_pooled=False, final_layer_norm=True, **tl)

    def set_clip_skip(self, clip_skip):
        for name in ("text_processing_engine", "text_processing_engine_l", "text_processing_engine_g"):
            if hasattr(self, name):
                getattr(self, name).clip_skip = clip_skip

    @torch.inference_mode()
    def get_learned_conditioning(self, prompt):
        """`prompt`: TokenizedPrompts (below).  SD1.x -> tensor [B, 77 n, 768]; SDXL -> {'crossattn': [B, 77 n, 2048], 'vector': [B, 2816]}
        (sdxl.py:76-117: penultimate CLIP-L | CLIP-G states, pooled-projected CLIP-G + six 256-wide size / crop embeddings)."""
        from ... import hipops as ops
        if not isinstance(prompt, TokenizedPrompts):
            # prompt strings (modules/prompt_parser.SdConditioning or a plain list), as Forge calls it (sd15.py:62-73, sdxl.py:76-117)
            texts = list(prompt)
            side = {"width": getattr(prompt, "width", None) or 1024, "height": getattr(prompt, "height", None) or 1024,
                    "is_negative_prompt": getattr(prompt, "is_negative_prompt", False), "all_empty": all(x == "" for x in texts)}
            if not self.is_sdxl:
                return self.text_processing_engine.encode_texts(texts)
            cond_l = self.text_processing_engine_l.encode_texts(texts)
            cond_g = self.text_processing_engine_g.encode_texts(texts)
            prompt = TokenizedPrompts(None, None, **side)
        elif not self.is_sdxl:
            if not hasattr(self, "text_processing_engine"):
                raise RuntimeError("no text encoder attached: attach_text_encoders() or pass cond tensors to the processing object")
            return self.text_processing_engine(prompt.tokens_l, prompt.multipliers_l)
        else:
            cond_l = self.text_processing_engine_l(prompt.tokens_l, prompt.multipliers_l)
            cond_g = self.text_processing_engine_g(prompt.tokens_g, prompt.multipliers_g)
        clip_pooled = cond_g.pooled
        vals = [prompt.height, prompt.width, prompt.crop_top, prompt.crop_left, prompt.height, prompt.width]  # sdxl.py:93-96
        t = torch.tensor([float(v) for v in vals], dtype=torch.float32, device=self.device)
        flat = ops.timestep_embedding(t, 256).float().flatten().unsqueeze(0).repeat(clip_pooled.shape[0], 1)
        if prompt.is_negative_prompt and prompt.all_empty:  # :100-105
            clip_pooled, cond_l, cond_g = torch.zeros_like(clip_pooled), torch.zeros_like(cond_l), torch.zeros_like(cond_g)
        from ...modules.prompt_parser import DictWithShape
        return DictWithShape({"crossattn": torch.cat([cond_l, cond_g], dim=2), "vector": torch.cat([clip_pooled, flat], dim=1)})


class TokenizedPrompts:
    """What the tokenizer side hands over for one batch of prompts: per text encoder [n_chunks][B][77] token ids and emphasis
    multipliers (classic_engine.py:150-261), plus the SDXL size conditioning inputs (sdxl.py:81-91)."""

    def __init__(self, tokens_l, multipliers_l, tokens_g=None, multipliers_g=None, width=1024, height=1024, crop_left=0, crop_top=0,
                 is_negative_prompt=False, all_empty=False):
        self.tokens_l, self.multipliers_l = tokens_l, multipliers_l
        self.tokens_g, self.multipliers_g = tokens_g, multipliers_g
        self.width, self.height, self.crop_left, self.crop_top = width, height, crop_left, crop_top
        self.is_negative_prompt, self.all_empty = is_negative_prompt, all_empty


def build_engine(unet_config, unet_state_dict, vae_config=None, vae_state_dict=None, device="cuda", prediction_type="epsilon", ztsnr=False,
                 vae_dtype=torch.float16):
    unet = IntegratedUNet2DConditionModel(unet_config, unet_state_dict, device=device)
    # vae_dtype: float16 (guarded: falls back to bfloat16 on overflow) or bfloat16; `memory_management.vae_dtype()` answers the reference's question
    vae = IntegratedAutoencoderKL(vae_config, vae_state_dict, device=device, dtype=vae_dtype) if vae_config is not None else None
    return ForgeDiffusionEngine(unet, vae, is_sdxl=unet_config.get("adm_in_channels") is not None, prediction_type=prediction_type, ztsnr=ztsnr)


class FluxEngine:
    """Object protocol of backend/diffusion_engine/flux.py:27-120 that the call surface touches: the transformer and, optionally, the
    16-channel VAE (decode_first_stage / encode_first_stage as flux.py:107-120); the T5 / CLIP text encoders are not built -- conditioning
    tensors are supplied."""

    def __init__(self, transformer, seq_len=4096, vae=None, schnell=None):
        """The predictor is chosen as flux.py:36-47 does: Flux-schnell gets mu = 1.0; everything else the time shift of a 4096-token image
        (mu = 1.15) -- a CONSTANT in the reference, whatever the resolution (its k_prediction.py:293 leaves binding the latent size to the
        sigmas as a TODO); `seq_len` is the hook for callers that want that binding.  The reference tells schnell from the repository name;
        here it is the model without a guidance embedding (the architectural difference between the two)."""
        from ..modules.k_model import KModelFlux
        from ..modules.k_prediction import PredictionFlux
        from ..patcher.unet import UnetPatcher
        if schnell is None:
            schnell = not transformer.guidance_embed
        predictor = PredictionFlux(mu=1.0) if schnell else PredictionFlux(seq_len=seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5,
                                                                         max_shift=1.15)
        patcher = UnetPatcher(KModelFlux(transformer, predictor), transformer.device, transformer.device)
        self.forge_objects = ForgeObjects(unet=patcher, clip=None, vae=VAE(vae) if vae is not None else None)
        self.forge_objects_original = self.forge_objects.shallow_copy()
        self.forge_objects_after_applying_lora = self.forge_objects.shallow_copy()
        self.is_sdxl = self.is_sd1 = self.is_inpaint = False
        self.is_flux = True
        self.use_distilled_cfg_scale = not schnell   # flux.py:47
        self.latent_channels = transformer.in_channels // 4
        self.device = transformer.device

    decode_first_stage = ForgeDiffusionEngine.decode_first_stage   # flux.py:113-117: same process_out -> decode -> [-1, 1] as sd15.py:80-84
    encode_first_stage = ForgeDiffusionEngine.encode_first_stage

    # ---- text conditioning (flux.py:52-73, :84-100): CLIP-L pooled vector + T5-XXL sequence --------------------------------------------------
    def attach_text_encoders(self, clip_l, t5xxl, tokenizer_l=None, tokenizer_t5=None, embeddings_l=None, emphasis_name="Original"):
        """clip_l: forge_amd.backend.nn.clip.IntegratedCLIP, t5xxl: forge_amd.backend.nn.t5.IntegratedT5; the tokenizers are the user's install's
        (CLIPTokenizer / T5TokenizerFast objects: their vocabularies are data files).  Options as flux.py:52-73 constructs the two engines."""
        from ..text_processing.classic_engine import ClassicTextProcessingEngine
        from ..text_processing.t5_engine import T5TextProcessingEngine
        self.text_processing_engine_l = ClassicTextProcessingEngine(clip_l, embedding_key="clip_l", text_projection=False, minimal_clip_skip=1, clip_skip=1,
                                                                    return_pooled=True, final_layer_norm=True, emphasis_name=emphasis_name,
                                                                    tokenizer=tokenizer_l, embeddings=embeddings_l)
        self.text_processing_engine_t5 = T5TextProcessingEngine(t5xxl, tokenizer_t5, emphasis_name=emphasis_name)

    def set_clip_skip(self, clip_skip):
        self.text_processing_engine_l.clip_skip = clip_skip          # flux.py:80-81

    @torch.inference_mode()
    def get_learned_conditioning(self, prompt):
        """flux.py:84-100: prompt strings (an SdConditioning or a list) -> {'crossattn': T5 [B, 256 n, 4096], 'vector': CLIP-L pooled [B, 768],
        'guidance': [B]} (the distilled guidance scale rides along only for the guidance-distilled model; schnell ignores it)."""
        if not hasattr(self, "text_processing_engine_t5"):
            raise RuntimeError("no text encoders attached: attach_text_encoders() or pass conditioning tensors to the processing object")
        from ...modules.prompt_parser import DictWithShape
        texts = list(prompt)
        cond_l = self.text_processing_engine_l.encode_texts(texts)
        cond = dict(crossattn=self.text_processing_engine_t5(texts), vector=cond_l.pooled)
        if self.use_distilled_cfg_scale:
            scale = getattr(prompt, "distilled_cfg_scale", 3.5) or 3.5
            cond["guidance"] = torch.FloatTensor([scale] * len(texts)).to(self.device)
        return DictWithShape(cond)

    def get_prompt_lengths_on_ui(self, prompt):
        n = len(self.text_processing_engine_t5.tokenize([prompt])[0])   # flux.py:102-105
        return n, max(255, n)


def build_flux_engine(flux_config, state_dict, device="cuda", vae_config=None, vae_state_dict=None, dtype=torch.float16, seq_len=4096, schnell=None,
                      vae_dtype=torch.float16):
    from ..nn.flux import IntegratedFluxTransformer2DModel
    net = IntegratedFluxTransformer2DModel(flux_config, state_dict, device=device, dtype=dtype)
    vae = IntegratedAutoencoderKL(vae_config, vae_state_dict, device=device, dtype=vae_dtype) if vae_config is not None else None
    return FluxEngine(net, seq_len=seq_len, vae=vae, schnell=schnell)
