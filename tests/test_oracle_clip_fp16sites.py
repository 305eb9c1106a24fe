"""CPU checks of the CLIP rounding oracle (oracle/clip_fp16sites.py): rounding off = oracle/clip.py bit for bit (which the tiny_clip fixtures pin to
transformers); rounding on = an fp16 pipeline at the size of the reference's own fp16 floor."""
import pytest
import torch

from forge_amd import synth
from oracle import clip as oc
from oracle import clip_fp16sites as c16

import parity


def _ids(cfg, b=2, t=77, seed=3):
    g = torch.Generator("cpu").manual_seed(seed)
    ids = torch.randint(0, cfg["vocab_size"] - 2, (b, t), generator=g)
    ids[:, 0] = cfg["vocab_size"] - 2
    ids[0, 12:] = cfg["vocab_size"] - 1
    ids[1, 40:] = cfg["vocab_size"] - 1
    return ids


@pytest.mark.parametrize("cfg", [synth.TINY_CLIP_L_CONFIG, synth.TINY_CLIP_G_CONFIG], ids=["clip_l", "clip_g"])
def test_rounding_off_is_the_pinned_restatement_and_rounding_on_is_fp16_sized(cfg):
    sd = synth.synth_clip_state_dict(cfg)
    ids = _ids(cfg)
    a = oc.clip_hidden_states(sd, cfg, ids)
    b = c16.clip_hidden_states(sd, cfg, ids, rounding=False)
    assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
    r = c16.clip_hidden_states(sd, cfg, ids)
    m = parity.metrics(r[-1], a[-1])
    print(m)
    assert 2e-4 < m["rms_rel"] < 3e-3 and torch.equal(r[-1], r[-1].half().float())
    # teacher-forced with its own hidden states it reproduces them
    t = c16.clip_hidden_states(sd, cfg, ids, teacher=r)
    assert all(torch.equal(x, y) for x, y in zip(r, t))
