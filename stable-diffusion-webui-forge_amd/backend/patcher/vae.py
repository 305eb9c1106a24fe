"""`VAE` patcher-level wrapper -- mirror of backend/patcher/vae.py:128-155 (`decode_inner`, `decode` with the optional
`model_vae_decode_wrapper` hook) and :157-187 (`encode_inner`, `encode` with `model_vae_encode_wrapper`).  No free-memory
chunking / tiled fallback: a batch of 1024^2 images fits HBM."""


class _Patcher:
    def __init__(self):
        self.model_options = {}


class VAE:
    def __init__(self, model):
        self.first_stage_model = model
        self.latent_channels = model.latent_channels
        self.downscale_ratio = model.up_factor
        self.patcher = _Patcher()
        self.device = model.device

    def decode_inner(self, samples_in):
        return self.first_stage_model.decode_inner(samples_in)

    def decode(self, samples_in):
        wrapper = self.patcher.model_options.get("model_vae_decode_wrapper", None)
        if wrapper is None:
            return self.decode_inner(samples_in)
        return wrapper(self.decode_inner, samples_in)

    def encode_inner(self, pixel_samples):
        """patcher/vae.py:157-180: pixel_samples NHWC in [0, 1] -> latent sample fp32 NCHW (not yet process_in'ed)."""
        regulation = self.patcher.model_options.get("model_vae_regulation", None)
        pixels_in = 2.0 * pixel_samples.movedim(-1, 1) - 1.0
        return self.first_stage_model.encode(pixels_in, regulation).float()

    def encode(self, pixel_samples):
        wrapper = self.patcher.model_options.get("model_vae_encode_wrapper", None)
        if wrapper is None:
            return self.encode_inner(pixel_samples)
        return wrapper(self.encode_inner, pixel_samples)
