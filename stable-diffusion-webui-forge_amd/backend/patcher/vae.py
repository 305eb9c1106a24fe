"""`VAE` patcher-level wrapper -- mirror of backend/patcher/vae.py:128-155 (`decode_inner`, `decode` with the optional
`model_vae_decode_wrapper` hook).  No free-memory chunking / tiled fallback: a batch of 1024^2 decodes fits HBM."""


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
