"""The one decision of the reference's backend/memory_management.py that the native path needs: which element type the VAE runs in.

Reference (:190-205, :840-855): `VAE_DTYPES = [float32]`, with bfloat16 put in front on parts that support it (NVIDIA >= sm80, Intel XPU);
`vae_dtype()` returns the type forced by `--vae-in-fp16 / --vae-in-bf16 / --vae-in-fp32`, else the first allowed type a caller listed, else
`VAE_DTYPES[0]`.  Model offloading, free-memory queries and the rest of that file are out of scope (weights are resident in 288 GB of HBM).

Native: the decoder / encoder kernels exist for float16 and bfloat16 (libfmx ABI 6).  MI355X is a bf16-capable part, so the preference list
reads [bfloat16, float16] for a caller that asks the reference's question ("what is safe for any checkpoint?"); the executors themselves
default to float16 -- 8x finer, the type of the parity fixtures and of BASELINE.json's configs -- and guard it: an fp16 decode that is not
finite is repeated in bfloat16 (backend/nn/vae.py `auto_bf16_fallback`).  float32 has no MFMA form on gfx950 and is refused.
"""
from types import SimpleNamespace

import torch

args = SimpleNamespace(vae_in_fp16=False, vae_in_bf16=False, vae_in_fp32=False)   # the reference's command-line flags (backend/args.py)

VAE_DTYPES = [torch.bfloat16, torch.float16]


def vae_dtype(device=None, allowed_dtypes=()):
    """backend/memory_management.py:840-855 with the native type list."""
    if args.vae_in_fp16:
        return torch.float16
    if args.vae_in_bf16:
        return torch.bfloat16
    if args.vae_in_fp32:
        raise NotImplementedError("--vae-in-fp32: the MI355X-native VAE runs in float16 or bfloat16 (bfloat16 has float32's range)")
    for d in allowed_dtypes:
        if d in VAE_DTYPES:
            return d
    return VAE_DTYPES[0]
