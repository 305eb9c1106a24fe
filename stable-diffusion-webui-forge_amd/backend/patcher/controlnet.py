"""ControlNet at the patcher level -- mirror of backend/patcher/controlnet.py: `apply_controlnet_advanced` (:11-76),
`compute_controlnet_weighting` (:79-146), `broadcast_image_to` (:149-168), `ControlBase` (:175-272: strength, start / end percent ->
sigma range, linked list of previous ControlNets, `control_merge`), `ControlNet.get_control` (:275-338).  The control model underneath is
the native backend/nn/cnets/cldm.ControlNet; `T2IAdapter` / `load_t2i_adapter` (:477-586) over the native backend/nn/cnets/t2i_adapter.Adapter.
`ControlLora` (:341-474): the control model is assembled in `pre_run` from the UNet's own trunk tensors plus the file's direct tensors and
`up @ down` pairs (summed once in fp32), and is an ordinary native ControlNet from there until `cleanup`.

Residuals stay fp16 and channels-last from the ControlNet's zero convs to the UNet's `h += ctrl` (the reference casts them to the
latent's fp32, :238-239; the native UNet consumes fp16, so the cast would only be undone)."""
import torch

from ..misc import image_resize


def apply_controlnet_advanced(unet, controlnet, image_bchw, strength, start_percent, end_percent, positive_advanced_weighting=None,
                              negative_advanced_weighting=None, advanced_frame_weighting=None, advanced_sigma_weighting=None,
                              advanced_mask_weighting=None):
    """-> a clone of `unet` (UnetPatcher) with the ControlNet appended to its linked list.  Weighting arguments as in the reference:
    per-injection-point weights for the cond / uncond halves, per-frame weights, a sigma -> weight function, a [B, 1, H, W] mask."""
    cnet = controlnet.copy().set_cond_hint(image_bchw, strength, (start_percent, end_percent))
    cnet.positive_advanced_weighting = positive_advanced_weighting
    cnet.negative_advanced_weighting = negative_advanced_weighting
    cnet.advanced_frame_weighting = advanced_frame_weighting
    cnet.advanced_sigma_weighting = advanced_sigma_weighting
    if advanced_mask_weighting is not None:
        assert isinstance(advanced_mask_weighting, torch.Tensor)
        B, C, H, W = advanced_mask_weighting.shape
        assert B > 0 and C == 1 and H > 0 and W > 0
    cnet.advanced_mask_weighting = advanced_mask_weighting
    m = unet.clone()
    m.add_patched_controlnet(cnet)
    return m


def get_at(array, index, default=None):
    return array[index] if 0 <= index < len(array) else default


def compute_controlnet_weighting(control, cnet):
    pos = getattr(cnet, "positive_advanced_weighting", None)
    neg = getattr(cnet, "negative_advanced_weighting", None)
    frame = getattr(cnet, "advanced_frame_weighting", None)
    sigma_fn = getattr(cnet, "advanced_sigma_weighting", None)
    mask = getattr(cnet, "advanced_mask_weighting", None)
    if pos is None and neg is None and frame is None and sigma_fn is None and mask is None:
        return control
    to = cnet.transformer_options
    cond_or_uncond, sigmas, cond_mark = to["cond_or_uncond"], to["sigmas"], to["cond_mark"]
    if frame is not None:
        frame = torch.Tensor(frame * len(cond_or_uncond)).to(sigmas)
        assert frame.shape[0] == cond_mark.shape[0], "Frame weighting list length is different from batch size!"
    sigma_w = torch.cat([sigma_fn(sigmas)] * len(cond_or_uncond)) if sigma_fn is not None else None
    for k, v in control.items():
        for i in range(len(v)):
            signal = control[k][i]
            if not isinstance(signal, torch.Tensor):
                continue
            B, C, H, W = signal.shape
            pw = get_at(pos.get(k, []), i, 1.0) if pos is not None else 1.0
            nw = get_at(neg.get(k, []), i, 1.0) if neg is not None else 1.0
            final = pw * (1.0 - cond_mark) + nw * cond_mark  # cond_mark is 1 on the uncond half
            if sigma_w is not None:
                final = final * sigma_w
            if frame is not None:
                final = final * frame
            if isinstance(mask, torch.Tensor):
                if mask.shape[0] != 1:
                    k_ = int(signal.shape[0] // mask.shape[0])
                    if signal.shape[0] == k_ * mask.shape[0]:
                        mask = mask.repeat(k_, 1, 1, 1)
                signal = signal * torch.nn.functional.interpolate(mask.to(signal), size=(H, W), mode="bilinear")
            control[k][i] = signal * final.to(signal)[:, None, None, None]
    return control


def broadcast_image_to(tensor, target_batch_size, batched_number):
    current = tensor.shape[0]
    if current == 1:
        return tensor
    per_batch = target_batch_size // batched_number
    tensor = tensor[:per_batch]
    if per_batch > tensor.shape[0]:
        tensor = torch.cat([tensor] * (per_batch // tensor.shape[0]) + [tensor[:(per_batch % tensor.shape[0])]], dim=0)
    if tensor.shape[0] == target_batch_size:
        return tensor
    return torch.cat([tensor] * batched_number, dim=0)


class ControlBase:
    """What every control object carries and how a chain of them behaves -- the attribute names and methods are the reference's public surface
    (backend/patcher/controlnet.py:11-45: extensions set `strength`, `timestep_percent_range`, `global_average_pooling`, chain with
    `set_previous_controlnet`); the arithmetic underneath is native.  A chain is a singly linked list through `previous_controlnet`; the walkers
    below visit it tail-first, as the reference's recursion does."""

    # (name, initial value) of the per-object state; the first four are what `copy_to` hands to a copy
    _COPIED = (("cond_hint_original", None), ("strength", 1.0), ("timestep_percent_range", (0.0, 1.0)), ("global_average_pooling", False))
    _STATE = _COPIED + (("cond_hint", None), ("timestep_range", None), ("previous_controlnet", None))

    def __init__(self, device=None):
        for name, initial in self._STATE:
            setattr(self, name, initial)
        self.transformer_options = {}
        self.device = device

    def _rest_of_chain(self):
        return self.previous_controlnet

    def set_cond_hint(self, cond_hint, strength=1.0, timestep_percent_range=(0.0, 1.0)):
        self.cond_hint_original, self.strength, self.timestep_percent_range = cond_hint, strength, timestep_percent_range
        return self

    def set_previous_controlnet(self, controlnet):
        self.previous_controlnet = controlnet
        return self

    def pre_run(self, model, percent_to_timestep_function):
        lo, hi = self.timestep_percent_range
        self.timestep_range = (percent_to_timestep_function(lo), percent_to_timestep_function(hi))
        rest = self._rest_of_chain()
        if rest is not None:
            rest.pre_run(model, percent_to_timestep_function)

    def cleanup(self):
        rest = self._rest_of_chain()
        if rest is not None:
            rest.cleanup()
        self.cond_hint = self.timestep_range = None

    def get_models(self):
        rest = self._rest_of_chain()
        return [] if rest is None else rest.get_models()

    def inference_memory_requirements(self, dtype):
        rest = self._rest_of_chain()
        return 0 if rest is None else rest.inference_memory_requirements(dtype)

    def copy_to(self, c):
        for name, _ in self._COPIED:
            setattr(c, name, getattr(self, name))

    def control_merge(self, control_input, control_output, control_prev, output_dtype):
        out = {"input": [], "middle": [], "output": []}
        if control_input is not None:
            for x in control_input:
                if x is not None and self.strength != 1.0:
                    x *= self.strength
                out["input"].insert(0, x)
        if control_output is not None:
            for i, x in enumerate(control_output):
                key = "middle" if i == len(control_output) - 1 else "output"
                if x is not None:
                    if self.global_average_pooling:
                        x = torch.mean(x, dim=(2, 3), keepdim=True).repeat(1, 1, x.shape[2], x.shape[3])
                    if self.strength != 1.0:
                        x *= self.strength
                out[key].append(x)
        out = compute_controlnet_weighting(out, self)
        if control_prev is not None:
            for name in ["input", "middle", "output"]:
                o = out[name]
                for i, prev_val in enumerate(control_prev[name]):
                    if i >= len(o):
                        o.append(prev_val)
                    elif prev_val is not None:
                        if o[i] is None:
                            o[i] = prev_val
                        elif o[i].shape[0] < prev_val.shape[0]:
                            o[i] = prev_val + o[i]
                        else:
                            o[i] += prev_val
        return out


class ControlNet(ControlBase):
    def __init__(self, control_model, global_average_pooling=False, device=None, load_device=None, manual_cast_dtype=None):
        super().__init__(device if device is not None else getattr(control_model, "device", None))
        self.control_model = control_model
        self.load_device = load_device
        self.global_average_pooling = global_average_pooling
        self.model_sampling_current = None
        self.manual_cast_dtype = manual_cast_dtype

    def get_control(self, x_noisy, t, cond, batched_number):
        """x_noisy fp32 [Bu, C, h, w] (NOT yet input-scaled), t = sigma [Bu], cond {'c_crossattn', 'y'} -> {'input', 'middle', 'output'}."""
        to = self.transformer_options
        for modifier in to.get("controlnet_conditioning_modifiers", []):
            x_noisy, t, cond, batched_number = modifier(self, x_noisy, t, cond, batched_number)
        control_prev = None
        if self.previous_controlnet is not None:
            control_prev = self.previous_controlnet.get_control(x_noisy, t, cond, batched_number)
        t0 = t.fmx_sigma.host[0] if hasattr(t, "fmx_sigma") else float(t[0])
        if self.timestep_range is not None:
            if t0 > self.timestep_range[0] or t0 < self.timestep_range[1]:
                return control_prev
        self._prepare_hint(x_noisy.shape[0], x_noisy.shape[2], x_noisy.shape[3], x_noisy.device, batched_number)
        context, y = cond["c_crossattn"], cond.get("y", None)
        predictor = self.model_sampling_current
        sig = t.fmx_sigma.host if hasattr(t, "fmx_sigma") else t.detach().float().cpu().tolist()
        sig = list(sig) * (x_noisy.shape[0] // len(sig))
        timestep = predictor.timestep(torch.tensor(sig, dtype=torch.float32)).float()
        scale = torch.tensor([1.0 / (s ** 2 + predictor.sigma_data ** 2) ** 0.5 for s in sig], dtype=torch.float32, device=x_noisy.device)
        x_in = x_noisy * scale[:, None, None, None]  # predictor.calculate_input (k_prediction.py:78-79)
        wrapper = to.get("controlnet_model_function_wrapper", None)
        if wrapper is not None:
            control = wrapper(x=x_in, hint=self.cond_hint, timesteps=timestep, context=context, y=y, model=self, inner_model=self.control_model)
        else:
            control = self.control_model(x=x_in, hint=self.cond_hint, timesteps=timestep, context=context, y=y)
        return self.control_merge(None, control, control_prev, x_noisy.dtype)

    def _prepare_hint(self, bu, hh, ww, device, batched_number):
        """The hint image at 8x the latent size, broadcast to the network batch (controlnet.py:get_control); kept across steps."""
        if self.cond_hint is None or hh * 8 != self.cond_hint.shape[2] or ww * 8 != self.cond_hint.shape[3] or bu != self.cond_hint.shape[0]:
            # `cleanup` drops cond_hint after every job; the executor keeps the last prepared image per source image, so that a second job
            # with the same hint finds the same tensor -- and with it the cached guided hint and a still-valid captured graph
            orig = self.cond_hint_original
            try:
                ver = orig._version
            except RuntimeError:
                ver = -1
            key = (orig.data_ptr(), tuple(orig.shape), orig.dtype, ver, hh, ww, bu, batched_number)
            holder = self.control_model if self.control_model is not None else self
            cached = getattr(holder, "_prepared_hint", None)
            if cached is not None and cached[0] == key and cached[1] is orig:
                self.cond_hint = cached[2]
                return self.cond_hint
            if self.cond_hint is None or hh * 8 != self.cond_hint.shape[2] or ww * 8 != self.cond_hint.shape[3]:
                self.cond_hint = image_resize.adaptive_resize(orig.to(device), ww * 8, hh * 8, "nearest-exact", "center")
            if bu != self.cond_hint.shape[0]:
                self.cond_hint = broadcast_image_to(self.cond_hint, bu, batched_number)
            try:
                holder._prepared_hint = (key, orig, self.cond_hint)
            except AttributeError:
                pass
        return self.cond_hint

    def graph_entry(self, sigma0, bu, hh, ww, context, y, batched_number):
        """What `KModel` needs to run this link of the chain inside its captured graph: the executor, its cached conditioning and guided hint,
        whether the link is active at this sigma (the timestep range is a host decision: the graph is keyed on the set of active links) and
        a callable naming everything of this link a captured graph points at.  None when the link needs Python per step -- conditioning
        modifiers, a model-function wrapper, advanced weighting, pooled residuals, a foreign control model -- and the chain runs eagerly."""
        to = self.transformer_options or {}
        cm = self.control_model
        if type(self).get_control is not ControlNet.get_control or not hasattr(cm, "forward_static"):
            return None
        if to.get("controlnet_conditioning_modifiers") or to.get("controlnet_model_function_wrapper") is not None or self.global_average_pooling:
            return None
        if any(getattr(self, a, None) is not None for a in ("positive_advanced_weighting", "negative_advanced_weighting", "advanced_frame_weighting",
                                                            "advanced_sigma_weighting", "advanced_mask_weighting")):
            return None
        if self.cond_hint_original is None or self.model_sampling_current is None:
            return None
        active = not (self.timestep_range is not None and (sigma0 > self.timestep_range[0] or sigma0 < self.timestep_range[1]))
        entry = {"cn": self, "cm": cm, "active": active, "ctxc": None, "gh": None, "valid": None}
        if active:
            hint = self._prepare_hint(bu, hh, ww, context.device, batched_number)
            ctxc = cm.prepare_context(context, y)
            gh = cm.hint_for_batch(hint, bu)
            if gh.shape[1] != hh or gh.shape[2] != ww:
                raise ValueError(f"hint {tuple(hint.shape)} is not 8x the latent {(hh, ww)}")
            entry.update(ctxc=ctxc, gh=gh)
            strength = float(self.strength)
            entry["valid"] = lambda: (ctxc.key, ctxc.serial, gh.data_ptr(), strength)
        return entry

    def copy(self):
        c = ControlNet(self.control_model, global_average_pooling=self.global_average_pooling, load_device=self.load_device,
                       manual_cast_dtype=self.manual_cast_dtype)
        self.copy_to(c)
        return c

    def get_models(self):
        return super().get_models() + [self.control_model]

    def pre_run(self, model, percent_to_timestep_function):
        super().pre_run(model, percent_to_timestep_function)
        self.model_sampling_current = model.predictor

    def cleanup(self):
        self.model_sampling_current = None
        super().cleanup()


def control_lora_state_dict(unet_sd, control_weights, device=None):
    """The weights ControlLora.pre_run assembles (patcher/controlnet.py:445-457 with ControlLoraOps :371-417): every trunk tensor of the UNet,
    overridden by the file's direct tensors; a module that carries an `up` / `down` pair computes with `weight + (up.flatten(1) @
    down.flatten(1)).reshape(weight.shape)`, where `weight` is the UNet's.  The sum is formed once here, in fp32 (the reference re-forms it in
    the storage dtype on every forward), so the control model is an ordinary ControlNet afterwards."""
    merged = dict(unet_sd)
    for k, v in control_weights.items():
        if k == "lora_controlnet" or k.endswith((".up", ".down")):
            continue
        merged[k] = v
    for k, up in control_weights.items():
        if not k.endswith(".up"):
            continue
        base = k[:-3]
        w = merged[base + ".weight"]
        dev = device if device is not None else w.device
        delta = torch.mm(up.to(dev, torch.float32).flatten(start_dim=1), control_weights[base + ".down"].to(dev, torch.float32).flatten(start_dim=1))
        merged[base + ".weight"] = w.to(dev, torch.float32) + delta.reshape(w.shape)
    return merged


class ControlLora(ControlNet):
    """patcher/controlnet.py:420-474: a ControlNet stored as low-rank differences to the UNet it is used with.  The control model exists only
    between pre_run and cleanup (it depends on the UNet, LoRAs included)."""

    def __init__(self, control_weights, global_average_pooling=False, device=None):
        ControlBase.__init__(self, device)
        self.control_weights = control_weights
        self.global_average_pooling = global_average_pooling
        self.control_model = None
        self.load_device = None
        self.model_sampling_current = None
        self.manual_cast_dtype = None

    def pre_run(self, model, percent_to_timestep_function):
        super().pre_run(model, percent_to_timestep_function)
        if self.control_model is not None:
            return  # sampling_prepare walks the chain AND every link recurses into its predecessors: build once per job (cleanup drops it)
        from ..nn.cnets import cldm
        net = model.diffusion_model
        config = dict(net.config)
        config.pop("out_channels", None)
        hint_channels = int(self.control_weights["input_hint_block.0.weight"].shape[1])
        self.device = net.device
        self.manual_cast_dtype = model.computation_dtype
        self.control_model = cldm.ControlNet(config, control_lora_state_dict(net.state_dict(), self.control_weights, device=net.device),
                                             device=net.device, hint_channels=hint_channels)

    def copy(self):
        c = ControlLora(self.control_weights, global_average_pooling=self.global_average_pooling)
        self.copy_to(c)
        return c

    def cleanup(self):
        self.control_model = None
        super().cleanup()

    def get_models(self):
        return ControlBase.get_models(self)

    def inference_memory_requirements(self, dtype):
        n = sum(int(v.numel()) for v in self.control_weights.values())
        return n * torch.empty(0, dtype=dtype).element_size() + ControlBase.inference_memory_requirements(self, dtype)


def load_controlnet(controlnet_data, unet_config=None, device="cuda"):
    """State dict -> patcher-level control object.  Control-LoRA files carry the `lora_controlnet` marker and need no config (it is the UNet's);
    a full ControlNet (LDM keys) is built on `unet_config`, the configuration of the UNet family it was trained for."""
    if "lora_controlnet" in controlnet_data:
        return ControlLora(controlnet_data, device=device)
    if "zero_convs.0.0.weight" not in controlnet_data:
        raise ValueError("not an LDM-keyed ControlNet state dict (diffusers-format files are converted by the loader of the host application)")
    if unet_config is None:
        raise ValueError("a full ControlNet needs the UNet configuration it belongs to")
    from ..nn.cnets import cldm
    config = {k: v for k, v in dict(unet_config).items() if k != "out_channels"}
    return ControlNet(cldm.ControlNet(config, controlnet_data, device=device, hint_channels=int(controlnet_data["input_hint_block.0.weight"].shape[1])))


class T2IAdapter(ControlBase):
    """patcher/controlnet.py:477-545: the adapter's features depend on the hint only -> computed once and cached (`control_input`), then
    injected as 'input' residuals (and, for SDXL adapters, the last one as 'middle') every step."""

    def __init__(self, t2i_model, channels_in, device=None):
        super().__init__(device if device is not None else getattr(t2i_model, "device", None))
        self.t2i_model = t2i_model
        self.channels_in = channels_in
        self.control_input = None

    def scale_image_to(self, width, height):
        import math
        r = self.t2i_model.unshuffle_amount
        return math.ceil(width / r) * r, math.ceil(height / r) * r

    def get_control(self, x_noisy, t, cond, batched_number):
        to = self.transformer_options
        for modifier in to.get("controlnet_conditioning_modifiers", []):
            x_noisy, t, cond, batched_number = modifier(self, x_noisy, t, cond, batched_number)
        control_prev = None
        if self.previous_controlnet is not None:
            control_prev = self.previous_controlnet.get_control(x_noisy, t, cond, batched_number)
        t0 = t.fmx_sigma.host[0] if hasattr(t, "fmx_sigma") else float(t[0])
        if self.timestep_range is not None:
            if t0 > self.timestep_range[0] or t0 < self.timestep_range[1]:
                return control_prev
        if self.cond_hint is None or x_noisy.shape[2] * 8 != self.cond_hint.shape[2] or x_noisy.shape[3] * 8 != self.cond_hint.shape[3]:
            self.control_input = None
            width, height = self.scale_image_to(x_noisy.shape[3] * 8, x_noisy.shape[2] * 8)
            self.cond_hint = image_resize.adaptive_resize(self.cond_hint_original.to(x_noisy.device), width, height, "nearest-exact", "center").float()
            if self.channels_in == 1 and self.cond_hint.shape[1] > 1:
                self.cond_hint = torch.mean(self.cond_hint, 1, keepdim=True)
        if x_noisy.shape[0] != self.cond_hint.shape[0]:
            self.cond_hint = broadcast_image_to(self.cond_hint, x_noisy.shape[0], batched_number)
            self.control_input = None
        if self.control_input is None:
            wrapper = to.get("controlnet_model_function_wrapper", None)
            if wrapper is not None:
                self.control_input = wrapper(hint=self.cond_hint, model=self, inner_model=self.t2i_model, inner_t2i_model=self.t2i_model)
            else:
                self.control_input = self.t2i_model(self.cond_hint)
        control_input = [None if a is None else a.clone() for a in self.control_input]  # control_merge scales in place
        mid = None
        if self.t2i_model.xl:
            mid, control_input = control_input[-1:], control_input[:-1]
        return self.control_merge(control_input, mid, control_prev, x_noisy.dtype)

    def copy(self):
        c = T2IAdapter(self.t2i_model, self.channels_in)
        self.copy_to(c)
        return c


def load_t2i_adapter(t2i_data, device="cuda"):
    """patcher/controlnet.py:548-586 for the `Adapter` family (TencentARC and diffusers key layouts)."""
    from ..nn.cnets import t2i_adapter
    if "adapter" in t2i_data:
        t2i_data = t2i_data["adapter"]
    if "adapter.body.0.resnets.0.block1.weight" in t2i_data:  # diffusers format
        repl = {}
        for i in range(4):
            for j in range(2):
                repl[f"adapter.body.{i}.resnets.{j}."] = f"body.{i * 2 + j}."
            repl[f"adapter.body.{i}."] = f"body.{i * 2}."
        repl["adapter."] = ""
        out = {}
        for k, v in t2i_data.items():
            for old in sorted(repl, key=len, reverse=True):
                if k.startswith(old):
                    k = repl[old] + k[len(old):]
                    break
            out[k] = v
        t2i_data = out
    keys = t2i_data.keys()
    if "body.0.in_conv.weight" in keys:
        # :561-563 hard-codes channels [320, 640, 1280, 1280] and nums_rb 4 -- what every Adapter_light checkpoint has; read off the tensors
        # here, which gives the same values for those files
        cin = t2i_data["body.0.in_conv.weight"].shape[1]
        channels = [t2i_data[f"body.{i}.out_conv.weight"].shape[0] for i in range(4)]
        nums_rb = sum(1 for k in keys if k.startswith("body.0.body.") and k.endswith(".block1.weight"))
        model = t2i_adapter.Adapter_light(t2i_data, channels=channels, nums_rb=nums_rb, cin=cin, device=device)
        return T2IAdapter(model, model.input_channels)
    if "conv_in.weight" not in keys:
        return None
    cin, channel = t2i_data["conv_in.weight"].shape[1], t2i_data["conv_in.weight"].shape[0]
    ksize = t2i_data["body.0.block2.weight"].shape[2]
    use_conv = any(k.endswith("down_opt.op.weight") for k in keys)
    xl = cin in (256, 768)
    model = t2i_adapter.Adapter(t2i_data, channels=[channel, channel * 2, channel * 4, channel * 4], nums_rb=2, cin=cin, ksize=ksize, sk=True,
                                use_conv=use_conv, xl=xl, device=device)
    return T2IAdapter(model, model.input_channels)
