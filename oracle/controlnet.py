"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates the ControlNet side of the path in plain torch fp32:
  backend/nn/cnets/cldm.py:229-254 `ControlNet.forward` (hint block -> trunk = UNet encoder with the hint added after the first conv ->
  zero convs) on the LDM-keyed state dict, with oracle/unet.py's block walker;
  backend/patcher/controlnet.py:79-146 compute_controlnet_weighting, :149-168 broadcast_image_to, :223-272 control_merge, :284-338 get_control
  (previous-ControlNet chain, sigma range gate, hint resize 'nearest-exact' + centre crop, strength, global average pooling).
  backend/patcher/controlnet.py:360-457 Control-LoRA weight assembly (`control_lora_weights`).
Pinned against the reference's own classes in tests/golden/*_controlnet.pt, *_control_lora.pt (oracle/make_golden.py gen_controlnet,
gen_control_lora)."""
import torch
import torch.nn.functional as F

from .unet import _conv, _lin, _run_block, timestep_embedding


@torch.no_grad()
def controlnet_forward(sd, cfg, x, hint, timesteps, context, y=None):
    mc = cfg["model_channels"]
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, mc))))
    g = hint
    for i, stride in enumerate((1, 1, 2, 1, 2, 1, 2, 1)):  # cldm.py:109-125: conv, SiLU, conv, ... (no SiLU after the last)
        g = _conv(sd, f"input_hint_block.{2 * i}", g if i == 0 else F.silu(g), stride=stride)
    if "label_emb.0.0.weight" in sd:
        emb = emb + _lin(sd, "label_emb.0.2", F.silu(_lin(sd, "label_emb.0.0", y)))
    outs, h, i = [], x, 0
    while f"zero_convs.{i}.0.weight" in sd:
        h = _run_block(sd, cfg, f"input_blocks.{i}", h, emb, context)
        if i == 0:
            h = h + g
        outs.append(_conv(sd, f"zero_convs.{i}.0", h, padding=0))
        i += 1
    h = _run_block(sd, cfg, "middle_block", h, emb, context)
    outs.append(_conv(sd, "middle_block_out.0", h, padding=0))
    return outs


def control_lora_weights(unet_sd, control_weights):
    """patcher/controlnet.py:445-457 (+ ControlLoraOps.forward :371-417): the control model starts from the UNet's tensors (keys the ControlNet
    does not have are dropped by the try / except), the file's plain tensors replace them, and a module with `up` / `down` computes with
    weight + (up.flatten(1) @ down.flatten(1)).reshape(weight.shape).  Returned as an ordinary ControlNet state dict for `controlnet_forward`."""
    sd = {k: v.float() for k, v in unet_sd.items() if k.startswith(("input_blocks.", "middle_block.", "time_embed.", "label_emb."))}
    pairs = {}
    for k, v in control_weights.items():
        if k == "lora_controlnet":
            continue
        if k.endswith(".up") or k.endswith(".down"):
            pairs.setdefault(k.rsplit(".", 1)[0], {})[k.rsplit(".", 1)[1]] = v.float()
        else:
            sd[k] = v.float()
    for base, ud in pairs.items():
        w = sd[base + ".weight"]
        sd[base + ".weight"] = w + (ud["up"].flatten(1) @ ud["down"].flatten(1)).reshape(w.shape)
    return sd


def adaptive_resize_nearest_exact_center(samples, width, height):
    ow, oh = samples.shape[3], samples.shape[2]
    oa, na = ow / oh, width / height
    x = y = 0
    if oa > na:
        x = round((ow - ow * (na / oa)) / 2)
    elif oa < na:
        y = round((oh - oh * (oa / na)) / 2)
    return F.interpolate(samples[:, :, y:oh - y, x:ow - x], size=(height, width), mode="nearest-exact")


def broadcast_image_to(t, target, batched_number):
    if t.shape[0] == 1:
        return t
    per = target // batched_number
    t = t[:per]
    if per > t.shape[0]:
        t = torch.cat([t] * (per // t.shape[0]) + [t[:(per % t.shape[0])]], dim=0)
    return t if t.shape[0] == target else torch.cat([t] * batched_number, dim=0)


class Control:
    """One link of the ControlNet chain (patcher/controlnet.py ControlNet + ControlBase state)."""

    def __init__(self, sd, cfg, hint, strength=1.0, percent_range=(0.0, 1.0), global_average_pooling=False, previous=None, weighting=None):
        self.sd, self.cfg, self.hint, self.strength, self.percent_range = sd, cfg, hint, strength, percent_range
        self.gap, self.previous, self.weighting = global_average_pooling, previous, weighting or {}

    def get_control(self, predictor, x_noisy, t, context, y, batched_number, to):
        prev = self.previous.get_control(predictor, x_noisy, t, context, y, batched_number, to) if self.previous is not None else None
        lo, hi = predictor.percent_to_sigma(self.percent_range[0]), predictor.percent_to_sigma(self.percent_range[1])
        if t[0] > lo or t[0] < hi:
            return prev
        hint = adaptive_resize_nearest_exact_center(self.hint, x_noisy.shape[3] * 8, x_noisy.shape[2] * 8)
        if hint.shape[0] != x_noisy.shape[0]:
            hint = broadcast_image_to(hint, x_noisy.shape[0], batched_number)
        outs = controlnet_forward(self.sd, self.cfg, predictor.calculate_input(t, x_noisy), hint, predictor.timestep(t).float(), context, y)
        out = {"input": [], "middle": [], "output": []}
        for i, x in enumerate(outs):
            if self.gap:
                x = torch.mean(x, dim=(2, 3), keepdim=True).repeat(1, 1, x.shape[2], x.shape[3])
            out["middle" if i == len(outs) - 1 else "output"].append(x * self.strength)
        out = self._weight(out, to)
        if prev is not None:
            for k in out:
                for i, pv in enumerate(prev[k]):
                    if i >= len(out[k]):
                        out[k].append(pv)
                    elif pv is not None:
                        out[k][i] = pv if out[k][i] is None else out[k][i] + pv
        return out

    def _weight(self, control, to):
        w = self.weighting
        if not w:
            return control
        reps, sigmas, cond_mark = len(to["cond_or_uncond"]), to["sigmas"], to["cond_mark"]
        frame = torch.tensor(list(w["frame"]) * reps).to(sigmas) if "frame" in w else 1.0
        sig = torch.cat([w["sigma"](sigmas)] * reps) if "sigma" in w else 1.0
        for k, v in control.items():
            for i, s in enumerate(v):
                pw = (w.get("positive", {}).get(k, []) + [1.0] * 99)[i] if "positive" in w else 1.0
                nw = (w.get("negative", {}).get(k, []) + [1.0] * 99)[i] if "negative" in w else 1.0
                final = (pw * (1.0 - cond_mark) + nw * cond_mark) * sig * frame
                if "mask" in w:
                    m = w["mask"]
                    if m.shape[0] != 1 and s.shape[0] % m.shape[0] == 0:
                        m = m.repeat(s.shape[0] // m.shape[0], 1, 1, 1)
                    s = s * F.interpolate(m.to(s), size=s.shape[2:], mode="bilinear")
                control[k][i] = s * final[:, None, None, None]
        return control


# ---- T2I-Adapter (backend/nn/cnets/t2i_adapter.py:64-164, patcher/controlnet.py:477-545) ------------------------------------------------------
@torch.no_grad()
def adapter_forward(sd, x, channels, nums_rb=2, ksize=1, use_conv=False, xl=False):
    """Adapter.forward: pixel-unshuffle -> conv_in -> ResnetBlocks (sk=True layout: in_conv only where the width changes, identity skip) -> feature
    list with the None placeholders that align features with UNet input blocks."""
    r = 16 if xl else 8
    down_at = (2,) if xl else (3, 2, 1)
    pad = ksize // 2
    conv = lambda k, t, stride=1, p=1: F.conv2d(t, sd[k + ".weight"], sd[k + ".bias"], stride=stride, padding=p)
    h = conv("conv_in", F.pixel_unshuffle(x, r))
    feats = []
    for i in range(len(channels)):
        for j in range(nums_rb):
            k = f"body.{i * nums_rb + j}"
            if j == 0 and i in down_at:
                h = conv(k + ".down_opt.op", h, stride=2) if use_conv else F.avg_pool2d(h, 2, 2)
            if k + ".in_conv.weight" in sd:
                h = conv(k + ".in_conv", h, p=pad)
            y = conv(k + ".block2", F.relu(conv(k + ".block1", h)), p=pad)
            h = y + (conv(k + ".skep", h, p=pad) if k + ".skep.weight" in sd else h)
        if xl:
            feats.append(None)
            if i == 0:
                feats += [None, None]
            if i == 2:
                feats.append(None)
        else:
            feats += [None, None]
        feats.append(h)
    return feats


@torch.no_grad()
def adapter_light_forward(sd, x, channels, nums_rb=4):
    """t2i_adapter.py:226-293 `Adapter_light`: pixel-unshuffle x8; per stage [2x2 average pool from the second stage on] -> 1x1 in_conv -> nums_rb x
    (3x3, ReLU, 3x3, + input) -> 1x1 out_conv; the feature list has two None placeholders before every stage's map."""
    h = F.pixel_unshuffle(x, 8)
    feats = []
    for i in range(len(channels)):
        if i > 0:
            h = F.avg_pool2d(h, kernel_size=2, stride=2)
        h = _conv(sd, f"body.{i}.in_conv", h, padding=0)
        for j in range(nums_rb):
            t = F.relu(_conv(sd, f"body.{i}.body.{j}.block1", h))
            h = _conv(sd, f"body.{i}.body.{j}.block2", t) + h
        h = _conv(sd, f"body.{i}.out_conv", h, padding=0)
        feats += [None, None, h]
    return feats


class AdapterControl:
    """T2IAdapter.get_control + ControlBase.control_merge for the 'input' residual list (reversed: control_merge inserts at the front)."""

    def __init__(self, sd, hint, strength=1.0, percent_range=(0.0, 1.0), previous=None, **adapter_kw):
        self.sd, self.hint, self.strength, self.percent_range, self.previous, self.kw = sd, hint, strength, percent_range, previous, adapter_kw
        self.features = None

    def get_control(self, predictor, x_noisy, t, context, y, batched_number, to):
        prev = self.previous.get_control(predictor, x_noisy, t, context, y, batched_number, to) if self.previous is not None else None
        lo, hi = predictor.percent_to_sigma(self.percent_range[0]), predictor.percent_to_sigma(self.percent_range[1])
        if t[0] > lo or t[0] < hi:
            return prev
        if self.features is None:
            hint = adaptive_resize_nearest_exact_center(self.hint, x_noisy.shape[3] * 8, x_noisy.shape[2] * 8).float()
            if hint.shape[0] != x_noisy.shape[0]:
                hint = broadcast_image_to(hint, x_noisy.shape[0], batched_number)
            self.features = adapter_forward(self.sd, hint, **self.kw)
        feats = [None if f is None else f.clone() for f in self.features]
        mid = None
        if self.kw.get("xl"):
            mid, feats = feats[-1:], feats[:-1]
        out = {"input": [], "middle": [], "output": []}
        for f in feats:
            out["input"].insert(0, None if f is None else f * self.strength)
        for f in (mid or []):
            out["middle"].append(f * self.strength)
        if prev is not None:
            for k in out:
                for i, pv in enumerate(prev[k]):
                    if i >= len(out[k]):
                        out[k].append(pv)
                    elif pv is not None:
                        out[k][i] = pv if out[k][i] is None else out[k][i] + pv
        return out
