"""TEST INFRASTRUCTURE ONLY (CPU oracle).

Restates UniPC as Forge uses it (modules/sd_samplers_timesteps_impl.py:145-181 -> modules/models/diffusion/uni_pc/uni_pc.py:
NoiseScheduleVP 'discrete' :100-174 with interpolate_fn :811-850, UniPC.data_prediction_fn :435-448, get_time_steps :459-474,
multistep_uni_pc_bh_update :625-743 / _vary_update :522-623 (predict_x0 branch), sample(method='multistep') :746-808), in plain
torch fp32 on whole tensors.  `eps_model(x, t_input_vec)` is the eps-mode denoiser (oracle.sampling.EpsFromDenoiser).
Pinned against the reference's own classes in tests/golden/tiny_sd15_samples_unipc.pt (oracle/make_golden.py gen_unipc).
"""
import torch


def _interp(x, xp, yp):
    """piecewise linear through (xp, yp), xp ascending; the outermost segments extend beyond the ends (:811-850)"""
    k = xp.shape[0]
    i = torch.clamp(torch.searchsorted(xp, x.contiguous(), right=True) - 1, 0, k - 2)
    return yp[i] + (x - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])


class Schedule:
    def __init__(self, alphas_cumprod):
        self.log_alpha = 0.5 * torch.log(alphas_cumprod)
        self.n = len(self.log_alpha)
        self.t = torch.linspace(0.0, 1.0, self.n + 1)[1:]

    def log_mean(self, t):
        return _interp(t.reshape(-1), self.t, self.log_alpha)

    def alpha(self, t):
        return torch.exp(self.log_mean(t))

    def std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.log_mean(t)))

    def lam(self, t):
        lm = self.log_mean(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))

    def inverse_lam(self, lamb):
        la = -0.5 * torch.logaddexp(torch.zeros((1,)), -2.0 * lamb)
        return _interp(la.reshape(-1), torch.flip(self.log_alpha, [0]), torch.flip(self.t, [0]))


def time_steps(ns, skip_type, t_T, t_0, n):
    if skip_type == "logSNR":
        return ns.inverse_lam(torch.linspace(ns.lam(torch.tensor(t_T)).item(), ns.lam(torch.tensor(t_0)).item(), n + 1))
    if skip_type == "time_uniform":
        return torch.linspace(t_T, t_0, n + 1)
    return torch.linspace(t_T ** 0.5, t_0 ** 0.5, n + 1).pow(2)  # time_quadratic


def sample_unipc(eps_model, x, steps, acd, variant="bh1", skip_type="time_uniform", order=3, lower_order_final=True, t_start=None):
    ns = Schedule(acd)
    b = x.shape[0]

    def x0_fn(xx, t):  # data prediction from the eps model at model time (t - 1/N) * 1000
        eps = eps_model(xx, ((t - 1.0 / ns.n) * 1000.0).expand(b))
        return (xx - ns.std(t) * eps) / ns.alpha(t)

    def update(xx, models, ts, t, p, use_corrector):
        m0, t0 = models[-1], ts[-1]
        h = (ns.lam(t) - ns.lam(t0))[0]
        rks = [((ns.lam(ts[-(i + 1)]) - ns.lam(t0)) / h)[0] for i in range(1, p)]
        d1 = [(models[-(i + 1)] - m0) / rks[i - 1] for i in range(1, p)]
        rks = torch.tensor([float(r) for r in rks] + [1.0])
        hh = -h
        alpha_t = ns.alpha(t)
        base = ns.std(t) / ns.std(t0) * xx - alpha_t * torch.expm1(hh) * m0
        if variant == "vary_coeff":
            k_n = len(rks)
            cols, col = [], torch.ones_like(rks)
            for k in range(1, k_n + 1):
                cols.append(col)
                col = col * rks / (k + 1)
            cmat = torch.stack(cols, dim=1)
            phis, fact, phi = [], 1, torch.expm1(hh)
            for k in range(1, k_n + 2):
                phis.append(phi)
                phi = phi / hh - 1 / fact
                fact *= (k + 1)
            x_t = base
            if d1:
                a_p = torch.linalg.inv(cmat[:-1, :-1])
                for k in range(k_n - 1):
                    x_t = x_t - alpha_t * phis[k + 1] * sum(a_p[k][j] * d1[j] for j in range(k_n - 1))
            model_t = None
            if use_corrector:
                a_c = torch.linalg.inv(cmat)
                model_t = x0_fn(x_t, t)
                x_t, k = base, 0
                for k in range(k_n - 1):
                    x_t = x_t - alpha_t * phis[k + 1] * sum(a_c[k][j] * d1[j] for j in range(k_n - 1))
                x_t = x_t - alpha_t * phis[k_n] * ((model_t - m0) * a_c[k][-1])
            return x_t, model_t
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        b_h = hh if variant == "bh1" else torch.expm1(hh)
        rmat, bvec, fact = [], [], 1
        for i in range(1, p + 1):
            rmat.append(torch.pow(rks, i - 1))
            bvec.append(h_phi_k * fact / b_h)
            fact *= (i + 1)
            h_phi_k = h_phi_k / hh - 1 / fact
        rmat, bvec = torch.stack(rmat), torch.tensor(bvec)
        pred = 0
        if d1:
            rhos_p = torch.tensor([0.5]) if p == 2 else torch.linalg.solve(rmat[:-1, :-1], bvec[:-1])
            pred = sum(rhos_p[k] * d1[k] for k in range(len(d1)))
        x_t = base - alpha_t * b_h * pred
        model_t = None
        if use_corrector:
            rhos_c = torch.tensor([0.5]) if p == 1 else torch.linalg.solve(rmat, bvec)
            model_t = x0_fn(x_t, t)
            corr = sum(rhos_c[k] * d1[k] for k in range(len(d1))) if d1 else 0
            x_t = base - alpha_t * b_h * (corr + rhos_c[-1] * (model_t - m0))
        return x_t, model_t

    t_0, t_T = 1.0 / ns.n, (1.0 if t_start is None else float(t_start))
    tsteps = time_steps(ns, skip_type, t_T, t_0, steps)
    t_of = lambda i: tsteps[i].reshape(1)
    models, ts = [x0_fn(x, t_of(0))], [t_of(0)]
    for p in range(1, order):
        x, m = update(x, models, ts, t_of(p), p, True)
        models.append(m if m is not None else x0_fn(x, t_of(p)))
        ts.append(t_of(p))
    for step in range(order, steps + 1):
        p = min(order, steps + 1 - step) if lower_order_final else order
        x, m = update(x, models, ts, t_of(step), p, step != steps)
        ts = (ts[1:] + [t_of(step)]) if order > 1 else [t_of(step)]
        if step < steps:
            m = m if m is not None else x0_fn(x, t_of(step))
            models = (models[1:] + [m]) if order > 1 else [m]
        elif order > 1:
            models = models[1:] + [models[-1]]
    return x
