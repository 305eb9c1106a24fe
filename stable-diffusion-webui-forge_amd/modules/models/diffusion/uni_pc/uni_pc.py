"""UniPC (Zhao et al. 2023) multistep predictor-corrector -- mirror of modules/models/diffusion/uni_pc/uni_pc.py:
`NoiseScheduleVP('discrete', alphas_cumprod=...)` (:6-174), `UniPC` (:372-808: `data_prediction_fn` :435, `get_time_steps` :459,
`multistep_uni_pc_vary_update` :522, `multistep_uni_pc_bh_update` :625, `sample(method='multistep')` :746) and `interpolate_fn` :811.

The schedule algebra (lambda, h, r_k, the B(h) / vary-coefficient linear systems of size <= order) is a handful of fp32 host
scalars per step; what touches the latents is, per step, two linear combinations of (x, previous model outputs, the new model
output), each ONE fused pass (fmx_sampler_lincomb).  Only what Forge's `unipc()` wrapper reaches is built: discrete schedule,
predict_x0, no thresholding, method 'multistep' (the reference raises NotImplementedError for anything else, :806)."""
import torch
import tqdm

from ..... import hipops as ops


def interpolate_fn(x, xp, yp):
    """Piecewise-linear y(x) through keypoints (xp ascending, 1-D), extended linearly beyond both ends."""
    k = xp.shape[0]
    i = torch.clamp(torch.searchsorted(xp, x.contiguous(), right=True) - 1, 0, k - 2)
    return yp[i] + (x - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])


class NoiseScheduleVP:
    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None):
        if schedule != "discrete":
            raise ValueError(f"only the 'discrete' schedule is on Forge's path, got {schedule}")
        log_alphas = 0.5 * torch.log(1 - betas).cumsum(dim=0) if betas is not None else 0.5 * torch.log(alphas_cumprod)
        self.schedule = schedule
        self.total_N = len(log_alphas)
        self.T = 1.0
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:]
        self.log_alpha_array = log_alphas.detach().float().cpu()

    def marginal_log_mean_coeff(self, t):
        return interpolate_fn(torch.as_tensor(t, dtype=torch.float32).reshape(-1), self.t_array, self.log_alpha_array)

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        log_mean = self.marginal_log_mean_coeff(t)
        return log_mean - 0.5 * torch.log(1.0 - torch.exp(2.0 * log_mean))

    def inverse_lambda(self, lamb):
        log_alpha = -0.5 * torch.logaddexp(torch.zeros((1,)), -2.0 * lamb)
        return interpolate_fn(log_alpha.reshape(-1), torch.flip(self.log_alpha_array, [0]), torch.flip(self.t_array, [0]))


def _combine(terms):
    """terms: [(coefficient, tensor)] -> one fused pass; repeated tensors are merged so each is read once."""
    srcs, coefs = [], []
    for c, t in terms:
        for j, s in enumerate(srcs):
            if s is t:
                coefs[j] += float(c)
                break
        else:
            srcs.append(t), coefs.append(float(c))
    return ops.lincomb(srcs, coefs)


class UniPC:
    def __init__(self, model_fn, noise_schedule, predict_x0=True, thresholding=False, max_val=1.0, variant="bh1", condition=None,
                 unconditional_condition=None, before_sample=None, after_sample=None, after_update=None):
        if not predict_x0 or thresholding:
            raise NotImplementedError("Forge's unipc() wrapper uses predict_x0=True, thresholding=False")
        self.model_fn_ = model_fn
        self.noise_schedule = noise_schedule
        self.variant = variant
        self.predict_x0 = predict_x0
        self.condition, self.unconditional_condition = condition, unconditional_condition
        self.before_sample, self.after_sample, self.after_update = before_sample, after_sample, after_update

    def model(self, x, t):
        """t: host float (continuous time); subclasses route it to the eps-mode CFGDenoiser (sd_samplers_timesteps_impl.UniPCCFG)."""
        return self.model_fn_(x, t, self.condition, self.unconditional_condition)

    def noise_prediction_fn(self, x, t):
        return self.model(x, t)

    def data_prediction_fn(self, x, t):
        noise = self.noise_prediction_fn(x, t)
        ns = self.noise_schedule
        alpha_t, sigma_t = float(ns.marginal_alpha(t)), float(ns.marginal_std(t))
        return ops.lincomb([x, noise], [1.0 / alpha_t, -sigma_t / alpha_t])  # x0 = (x - sigma_t eps) / alpha_t

    def model_fn(self, x, t):
        return self.data_prediction_fn(x, t)

    def get_time_steps(self, skip_type, t_T, t_0, N, device=None):
        ns = self.noise_schedule
        if skip_type == "logSNR":
            lambda_T, lambda_0 = ns.marginal_lambda(torch.tensor(t_T)), ns.marginal_lambda(torch.tensor(t_0))
            return ns.inverse_lambda(torch.linspace(lambda_T.item(), lambda_0.item(), N + 1))
        if skip_type == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        if skip_type == "time_quadratic":
            return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)
        raise ValueError(f"Unsupported skip_type {skip_type}, need to be 'logSNR' or 'time_uniform' or 'time_quadratic'")

    # -- one predictor(-corrector) step --------------------------------------------------------------------------------------------
    def multistep_uni_pc_update(self, x, model_prev_list, t_prev_list, t, order, **kwargs):
        if "bh" in self.variant:
            return self.multistep_uni_pc_bh_update(x, model_prev_list, t_prev_list, t, order, **kwargs)
        assert self.variant == "vary_coeff"
        return self.multistep_uni_pc_vary_update(x, model_prev_list, t_prev_list, t, order, **kwargs)

    def _step_scalars(self, t_prev_list, t, order):
        ns = self.noise_schedule
        t_prev_0 = t_prev_list[-1]
        lambda_prev_0, lambda_t = ns.marginal_lambda(t_prev_0), ns.marginal_lambda(t)
        h = (lambda_t - lambda_prev_0)[0]
        rks = [((ns.marginal_lambda(t_prev_list[-(i + 1)]) - lambda_prev_0) / h)[0] for i in range(1, order)]
        return h, rks, float(ns.marginal_std(t) / ns.marginal_std(t_prev_0)), float(ns.marginal_alpha(t))

    def multistep_uni_pc_bh_update(self, x, model_prev_list, t_prev_list, t, order, x_t=None, use_corrector=True):
        assert order <= len(model_prev_list)
        h, rk_list, sigma_ratio, alpha_t = self._step_scalars(t_prev_list, t, order)
        m0 = model_prev_list[-1]
        prev = [model_prev_list[-(i + 1)] for i in range(1, order)]
        rks = torch.tensor([float(r) for r in rk_list] + [1.0])
        hh = -h  # predict_x0
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        if self.variant == "bh1":
            B_h = hh
        elif self.variant == "bh2":
            B_h = torch.expm1(hh)
        else:
            raise NotImplementedError()
        R, b, factorial_i = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(float(h_phi_k * factorial_i / B_h))
            factorial_i *= (i + 1)
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        R, b = torch.stack(R), torch.tensor(b)
        # x_t_ = sigma_t / sigma_prev_0 * x - alpha_t * h_phi_1 * m0; residual terms are sums over D1_k = (m_k - m0) / r_k
        base = [(sigma_ratio, x), (-alpha_t * float(h_phi_1), m0)]
        scale = -alpha_t * float(B_h)

        def d1_terms(rhos):
            out = []
            for k, m in enumerate(prev):
                w = scale * float(rhos[k]) / float(rks[k])
                out += [(w, m), (-w, m0)]
            return out
        if x_t is None:
            if prev:
                rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
                x_t = _combine(base + d1_terms(rhos_p))
            else:
                x_t = _combine(base)
        model_t = None
        if use_corrector:
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            model_t = self.model_fn(x_t, t)
            w = scale * float(rhos_c[-1])
            x_t = _combine(base + d1_terms(rhos_c[:-1]) + [(w, model_t), (-w, m0)])
        return x_t, model_t

    def multistep_uni_pc_vary_update(self, x, model_prev_list, t_prev_list, t, order, use_corrector=True):
        assert order <= len(model_prev_list)
        h, rk_list, sigma_ratio, alpha_t = self._step_scalars(t_prev_list, t, order)
        m0 = model_prev_list[-1]
        prev = [model_prev_list[-(i + 1)] for i in range(1, order)]
        rks = torch.tensor([float(r) for r in rk_list] + [1.0])
        K = len(rks)
        C, col = [], torch.ones_like(rks)
        for k in range(1, K + 1):
            C.append(col)
            col = col * rks / (k + 1)
        C = torch.stack(C, dim=1)
        A_p = torch.linalg.inv(C[:-1, :-1]) if prev else None
        hh = -h
        h_phi_ks, factorial_k, h_phi_k = [], 1, torch.expm1(hh)
        for k in range(1, K + 2):
            h_phi_ks.append(float(h_phi_k))
            h_phi_k = h_phi_k / hh - 1 / factorial_k
            factorial_k *= (k + 1)
        base = [(sigma_ratio, x), (-alpha_t * h_phi_ks[0], m0)]

        def residual(A, cols):
            out = []
            for k in range(K - 1):
                for j, m in enumerate(prev):
                    w = -alpha_t * h_phi_ks[k + 1] * float(A[k][j]) / float(rks[j])
                    out += [(w, m), (-w, m0)]
            return out
        x_t = _combine(base + (residual(A_p, K - 1) if prev else []))
        model_t = None
        if use_corrector:
            A_c = torch.linalg.inv(C)
            model_t = self.model_fn(x_t, t)
            k_last = max(K - 2, 0)  # the reference indexes A_c with the loop variable left over from the residual loop (:589-593)
            w = -alpha_t * h_phi_ks[K] * float(A_c[k_last][-1])
            x_t = _combine(base + residual(A_c, K) + [(w, model_t), (-w, m0)])
        return x_t, model_t

    # -- driver ------------------------------------------------------------------------------------------------------------------------
    def sample(self, x, steps=20, t_start=None, t_end=None, order=3, skip_type="time_uniform", method="singlestep", lower_order_final=True,
               denoise_to_zero=False, solver_type="dpm_solver", atol=0.0078, rtol=0.05, corrector=False, disable=True):
        if method != "multistep":
            raise NotImplementedError()
        t_0 = 1.0 / self.noise_schedule.total_N if t_end is None else t_end
        t_T = self.noise_schedule.T if t_start is None else float(t_start)
        assert steps >= order, "UniPC order must be < sampling steps"
        timesteps = self.get_time_steps(skip_type=skip_type, t_T=t_T, t_0=t_0, N=steps)
        assert timesteps.shape[0] - 1 == steps
        t_of = lambda i: timesteps[i].reshape(1)
        model_prev_list, t_prev_list = [self.model_fn(x, t_of(0))], [t_of(0)]
        with tqdm.tqdm(total=steps, disable=disable) as pbar:
            for init_order in range(1, order):  # warm-up: lower-order steps until `order` model outputs exist
                x, model_x = self.multistep_uni_pc_update(x, model_prev_list, t_prev_list, t_of(init_order), init_order, use_corrector=True)
                if model_x is None:
                    model_x = self.model_fn(x, t_of(init_order))
                if self.after_update is not None:
                    self.after_update(x, model_x)
                model_prev_list.append(model_x)
                t_prev_list.append(t_of(init_order))
                pbar.update()
            for step in range(order, steps + 1):
                step_order = min(order, steps + 1 - step) if lower_order_final else order
                x, model_x = self.multistep_uni_pc_update(x, model_prev_list, t_prev_list, t_of(step), step_order, use_corrector=step != steps)
                if self.after_update is not None:
                    self.after_update(x, model_x)
                t_prev_list = t_prev_list[1:] + [t_of(step)] if order > 1 else [t_of(step)]
                if step < steps:  # the final model value is never needed
                    if model_x is None:
                        model_x = self.model_fn(x, t_of(step))
                    model_prev_list = model_prev_list[1:] + [model_x] if order > 1 else [model_x]
                elif order > 1:
                    model_prev_list = model_prev_list[1:] + [model_prev_list[-1]]
                pbar.update()
        return x
