"""`KModel` -- mirror of backend/modules/k_model.py:8-59 over the native UNet executor.

`apply_model(x, t=sigma, c_crossattn, y, ...)` keeps the reference contract (fp32 NCHW latents in, fp32 denoised
out) but executes `x / sqrt(sigma^2+1)` + im2col as ONE pack kernel, the UNet as a (graph-replayed) chain of gfx950
kernels and `x - eps*sigma` as ONE kernel.  The fully fused CFG path used by the sampler is `denoise_cfg` below:
[uncond ; cond] batch in one forward, CFG combine fused with calculate_denoised.
"""
import torch

from ... import hipops as ops
from ...runtime import HipGraph


class SigmaInfo:
    """Host knowledge about a device sigma vector (attached by our samplers to avoid a device sync per step)."""

    __slots__ = ("host",)

    def __init__(self, host):
        self.host = host  # list[float], one per sample


def tensor_version(t):
    """`_version` is unavailable on inference-mode tensors (processing runs under torch.inference_mode, as the reference
    does, processing.py:911); identity (data_ptr, shape) is then the cache key."""
    try:
        return t._version
    except RuntimeError:
        return -1


def host_sigmas(sigma):
    info = getattr(sigma, "fmx_sigma", None)
    if info is not None:
        return info.host
    return [float(v) for v in sigma.detach().float().cpu().tolist()]  # foreign caller: one sync


class KModel:
    MAX_CACHED_SHAPES = 16  # captured graphs + their static input buffers, one set per (batch, channels, h, w, CFG halves)

    def __init__(self, model, predictor, use_graph=True):
        self.diffusion_model = model
        self.predictor = predictor
        self.storage_dtype = model.storage_dtype
        self.computation_dtype = model.computation_dtype
        self.device = model.device
        self.use_graph = use_graph
        self._graphs = {}    # captured graphs, one per (shape key [, the active ControlNet executors])
        self._gstate = {}    # per graph key: warm-up count, what the graph points at (`validity`), its result view
        self._static = {}    # static input buffers (packed x, timesteps) per shape key
        self._stream = None

    def memory_required(self, input_shape):
        return 0  # weights and arena are resident (288 GB HBM); kept for interface parity (k_model.py:48-59)

    # --------------------------------------------------------------------------------------------------------
    def _timesteps(self, sig_host, reps):
        t = self.predictor.timestep(torch.tensor(sig_host, dtype=torch.float32)).float()
        return t.repeat(reps)

    def _drop_graphs(self):
        torch.cuda.synchronize(self.device)
        for g in self._graphs.values():
            g.destroy()
        self._graphs.clear()
        self._gstate.clear()

    def _control_plan(self, control_model, sig_host, bu, hh, ww, ctx, reps):
        """The ControlNet chain as a list of capturable entries (outermost first), or None when any link needs Python per step (conditioning
        modifiers, a model-function wrapper, advanced weighting, a T2I adapter, ...): patcher/controlnet.py `graph_entry`."""
        if not self.use_graph or len(set(sig_host)) != 1:
            return None
        entries, seen, p = [], set(), control_model
        while p is not None:
            e = p.graph_entry(sig_host[0], bu, hh, ww, ctx[0], ctx[1], reps) if hasattr(p, "graph_entry") else None
            if e is None or id(e["cm"]) in seen:   # one executor twice in a chain: its second forward would overwrite the first one's residuals
                return None
            seen.add(id(e["cm"]))
            entries.append(e)
            p = p.previous_controlnet
        return entries

    def _forward_static(self, key, x, sigma_dev, sig_host, reps, ctxc, control=None, transformer_options=None, c_concat=None, control_plan=None):
        """pack -> [ControlNets ->] UNet -> (returns eps view); static buffers per shape so the chain can be graph-replayed."""
        b, c, hh, ww = x.shape
        bu = reps * b
        st = self._static.get(key)
        if st is None:
            if len(self._static) >= self.MAX_CACHED_SHAPES:
                # a long-lived server sees many (batch, resolution) shapes: drop every cached graph and its static buffers rather than
                # grow without bound (the next call of each shape warms up and captures again)
                self._drop_graphs()
                self._static.clear()
            st = {"xcol": torch.empty(bu * hh * ww, 64, dtype=torch.float16, device=self.device),
                  "t": torch.empty(bu, dtype=torch.float32, device=self.device)}
            self._static[key] = st
        ops.unet_pack_input(x, sigma_dev, reps, self.predictor.sigma_data, out=st["xcol"])
        tvals = self._timesteps(sig_host, reps)
        if len(set(tvals.tolist())) == 1:
            st["t"].fill_(float(tvals[0]))
        else:
            st["t"].copy_(tvals, non_blocking=False)
        net = self.diffusion_model
        hooks = net._hooks(transformer_options)
        concat_term = net.prepare_concat(c_concat, bu) if c_concat is not None else None  # cached per c_concat tensor: once per job
        if not self.use_graph or control is not None or hooks is not None:
            # Python hooks cannot be captured, and a ControlNet chain that needs Python per step arrives here with its residuals: eager
            return net.forward_packed(st["xcol"], st["t"], ctxc, bu, hh, ww, control, hooks, concat_term)
        active = [e for e in control_plan if e["active"]] if control_plan else []
        gkey = key if not active else key + ("control",) + tuple(e["cm"].exec_serial for e in active)

        def run():
            # The ControlNet trunks read the SAME packed input and timestep buffers as the UNet (both are `calculate_input` of x and the
            # predictor's timestep of sigma: k_model.py:31-35, patcher/controlnet.py get_control), innermost link of the chain first,
            # merged exactly as get_control / control_merge do in the eager path.
            ctrl = None
            for e in reversed(active):
                outs = e["cm"].forward_static(st["xcol"], st["t"], e["ctxc"], bu, hh, ww, e["gh"])
                ctrl = e["cn"].control_merge(None, outs, ctrl, torch.float32)
            return net.forward_packed(st["xcol"], st["t"], ctxc, bu, hh, ww, control=ctrl, concat_term=concat_term)

        def validity():
            # What a captured graph points at: the executors' arenas (re-allocated when a larger shape comes through, e.g. the hires pass),
            # the cross-attention K / V^T buffers (re-allocated whenever the conditioning changes -- also when ANOTHER shape's job came in
            # between and this job's conditioning tensor then re-used the address, hence the key, of the earlier one: the serial number tells,
            # the key alone does not; that was a NaN: tools/soak.py, batch 8 -> batch 1 -> batch 8), the concat term, the guided hints.
            return ((net.arena_epoch,) + tuple(e["cm"].arena_epoch for e in active),
                    (ctxc.key, ctxc.serial), None if concat_term is None else concat_term.data_ptr(), tuple(e["valid"]() for e in active))

        gs = self._gstate.get(gkey)
        if gs is None:
            if len(self._graphs) >= 2 * self.MAX_CACHED_SHAPES:
                self._drop_graphs()
            gs = self._gstate[gkey] = {"warm": 0, "valid": None, "eps": None}
        g = self._graphs.get(gkey)
        if g is not None:
            now = validity()
            if now != gs["valid"]:
                g.destroy()
                del self._graphs[gkey]
                g = None
                # same arenas, other conditioning / hint (the usual case: a new job of a shape seen before): every buffer the chain needs
                # exists, capture again right away; a new arena is sized by an eager run first
                gs["warm"] = 2 if now[0] == gs["valid"][0] else 0
        if g is None:
            if gs["warm"] < 2:
                # eager warm-up (sizes the arenas, creates lazily-built buffers), then capture on a side stream
                eps = run()
                gs["warm"] += 1
                if gs["warm"] < 2:
                    return eps
            cur = torch.cuda.current_stream(self.device)
            if self._stream is None:
                self._stream = torch.cuda.Stream(self.device)
            s = self._stream
            s.wait_stream(cur)
            g = HipGraph()
            with torch.cuda.stream(s):
                gs["eps"] = g.capture(s, run)
                g.launch(s)
            cur.wait_stream(s)
            self._graphs[gkey] = g
            gs["valid"] = validity()
            return gs["eps"]
        cur = torch.cuda.current_stream(self.device)
        s = self._stream
        s.wait_stream(cur)
        g.launch(s)
        cur.wait_stream(s)
        return gs["eps"]

    def denoise_cfg(self, x, sigma, uncond_ctx, cond_ctx, cond_scale, want_parts=False, transformer_options=None, control_model=None,
                    c_concat=None):
        """Fused path: returns CFG-combined denoised (fp32 NCHW) [+ cond_pred, uncond_pred].
        `uncond_ctx`/`cond_ctx`: (context [B,T,Dc], y or None); uncond_ctx None => cond_scale == 1 shortcut.
        `transformer_options` with Python hooks: completed with the per-call keys of sampling_function.py:253-257 and run eagerly."""
        b, c, hh, ww = x.shape
        reps = 1 if uncond_ctx is None else 2
        if self.diffusion_model._hooks(transformer_options) is not None or control_model is not None:
            to = dict(transformer_options or {})
            cond_or_uncond = [1, 0] if reps == 2 else [0]          # batch order [uncond ; cond] (sampling_function.py:187-229)
            to["cond_or_uncond"] = cond_or_uncond[:]
            to["sigmas"] = sigma
            to["cond_mark"] = torch.tensor([float(cx) for cx in cond_or_uncond for _ in range(b)], dtype=sigma.dtype, device=sigma.device)
            to["cond_indices"] = [i * b + j for i, cx in enumerate(cond_or_uncond) if cx == 0 for j in range(b)]
            to["uncond_indices"] = [i * b + j for i, cx in enumerate(cond_or_uncond) if cx != 0 for j in range(b)]
            transformer_options = to
        else:
            transformer_options = None
        per_call_options = transformer_options
        sig_host = host_sigmas(sigma)
        if reps == 2:
            ctx = self._stack_ctx(uncond_ctx, cond_ctx)
        else:
            ctx = cond_ctx
        ctxc = self.diffusion_model.prepare_context(ctx[0], ctx[1])
        key = (b, c, hh, ww, reps)
        control, plan = None, None
        if control_model is not None:
            # sampling_function.py:261-268: every ControlNet of the chain sees the per-call options, then one get_control on the stacked batch
            p = control_model
            while p is not None:
                p.transformer_options = per_call_options
                p = p.previous_controlnet
            # a chain that needs no Python per step runs inside the captured graph; otherwise its residuals are computed here, eagerly
            if self.diffusion_model._hooks(transformer_options) is None:
                plan = self._control_plan(control_model, sig_host, reps * b, hh, ww, ctx, reps)
            if plan is None:
                t_all = torch.cat([sigma] * reps)
                t_all.fmx_sigma = SigmaInfo(list(sig_host) * reps)
                control = control_model.get_control(torch.cat([x] * reps), t_all, {"c_crossattn": ctx[0], "y": ctx[1]}, reps)
        eps = self._forward_static(key, x, sigma, sig_host, reps, ctxc, control=control, transformer_options=transformer_options, c_concat=c_concat,
                                   control_plan=plan)
        cond_pred = torch.empty_like(x) if want_parts else None
        uncond_pred = torch.empty_like(x) if want_parts else None
        den = ops.cfg_combine(eps, eps.shape[-1], x, sigma, reps, cond_scale, None, cond_pred, uncond_pred,
                              prediction_type=self.predictor.prediction_type, sigma_data=self.predictor.sigma_data)
        return (den, cond_pred, uncond_pred) if want_parts else den

    def _stack_ctx(self, uc, c):
        """[uncond ; cond] along batch (sampling_function.py:187-236 order); cached on the identity of the parts."""
        key = (uc[0].data_ptr(), c[0].data_ptr(), tensor_version(uc[0]), tensor_version(c[0]), tuple(uc[0].shape), tuple(c[0].shape))
        cached = getattr(self, "_stacked", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        tu, tc = uc[0].shape[1], c[0].shape[1]
        cu, cc = uc[0], c[0]
        if tu != tc:  # ConditionCrossAttn.concat (condition.py:56-70): repeat to the lcm of the token counts
            import math
            l = tu * tc // math.gcd(tu, tc)
            cu, cc = cu.repeat(1, l // tu, 1), cc.repeat(1, l // tc, 1)
        ctx = torch.cat([cu, cc]).contiguous()
        y = None if c[1] is None else torch.cat([uc[1], c[1]]).contiguous()
        self._stacked = (key, (ctx, y), (uc, c))
        return ctx, y

    def apply_model(self, x, t, c_concat=None, c_crossattn=None, control=None, transformer_options=None, y=None, **kwargs):
        """Reference signature (k_model.py:25): x fp32 [Bu,C,H,W], t = sigma [Bu] -> denoised fp32."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        sigma = t.to(device=self.device, dtype=torch.float32).contiguous()
        ctxc = self.diffusion_model.prepare_context(c_crossattn, y)
        b, c, hh, ww = x.shape
        eps = self._forward_static((b, c, hh, ww, 1, "apply"), x, sigma, host_sigmas(t), 1, ctxc, control, transformer_options, c_concat)
        return ops.cfg_combine(eps, eps.shape[-1], x, sigma, 1, 1.0, prediction_type=self.predictor.prediction_type,
                               sigma_data=self.predictor.sigma_data)


class KModelFlux:
    """`KModel` over the Flux executor (k_model.py:25-46 with prediction_type 'const': input = x, timestep = sigma, denoised =
    x - out * sigma).  Flux-dev is guidance-distilled: normally cond_scale == 1 and no uncond batch (diffusion_engine/flux.py:88-93); with a negative
    prompt and cond_scale != 1 the uncond batch is a second model call (denoise_cfg)."""

    def __init__(self, model, predictor):
        self.diffusion_model = model
        self.predictor = predictor
        self.storage_dtype = model.storage_dtype
        self.computation_dtype = model.computation_dtype
        self.device = model.device
        self.use_graph = False

    def memory_required(self, input_shape):
        return 0

    def apply_model(self, x, t, c_concat=None, c_crossattn=None, control=None, transformer_options=None, y=None, guidance=None, **kwargs):
        if c_concat is not None or control is not None:
            raise NotImplementedError("c_concat / control are outside the txt2img hot path")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        sigma = t.to(device=self.device, dtype=torch.float32).contiguous()
        out = self.diffusion_model.forward(x, sigma, c_crossattn, y, guidance).float()
        sig_host = host_sigmas(t)
        if len(set(sig_host)) == 1:
            return ops.lincomb3(x, out, None, 1.0, -float(sig_host[0]), 0.0)   # x - out * sigma in one kernel
        return x - out * sigma.view(-1, 1, 1, 1)

    def denoise_cfg(self, x, sigma, uncond_ctx, cond_ctx, cond_scale, want_parts=False, transformer_options=None, control_model=None):
        to = transformer_options or {}
        if control_model is not None or to.get("patches") or to.get("patches_replace") or to.get("block_modifiers"):
            raise NotImplementedError("ControlNet / per-block hooks are built for the LDM UNet executor, not for the Flux transformer")
        ctx, y, guidance = cond_ctx
        den = self.apply_model(x, sigma, c_crossattn=ctx, y=y, guidance=guidance)
        if uncond_ctx is None:          # cfg scale 1 (the guidance-distilled default, diffusion_engine/flux.py:88-93): one model call per step
            return (den, den, None) if want_parts else den
        # a negative prompt with cond_scale != 1: the reference's generic path (sampling_function.py:154-288, :292-312) runs the uncond batch through
        # the model as well and combines  uncond + (cond - uncond) * scale.  Two model calls (the batch a Flux transformer is tuned for is the job's own;
        # the uncond context may also differ in token count), one fused pass for the combination.
        uctx, uy, ug = uncond_ctx
        den_u = self.apply_model(x, sigma, c_crossattn=uctx, y=uy, guidance=ug if ug is not None else guidance)
        out = ops.lincomb3(den_u, den, None, 1.0 - float(cond_scale), float(cond_scale), 0.0)
        return (out, den, den_u) if want_parts else out
