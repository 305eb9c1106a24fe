"""DEIS coefficient tables -- mirror of k_diffusion/deis.py (`edm2t` :13-21, `get_deis_coeff_list` :59-121).  Host-side, O(steps * N)
scalar work done once per sampling run; the latent-sized updates it feeds are in sampling.py (`sample_deis`)."""
import torch

_EPS_S, _SIGMA_MIN, _SIGMA_MAX = 1e-3, 0.002, 80.0


def edm2t(edm_steps):
    """EDM sigma -> the VP-SDE time whose sigma(t) = sqrt(exp(beta_d t^2 / 2 + beta_min t) - 1) equals it; beta_d / beta_min are the
    VP parameters that put sigma_min at t = eps_s and sigma_max at t = 1.  Returns (t, beta_0, beta_1)."""
    lo, hi = torch.tensor(_SIGMA_MIN) ** 2 + 1, torch.tensor(_SIGMA_MAX) ** 2 + 1
    beta_d = 2 * (lo.log() / _EPS_S - hi.log()) / (_EPS_S - 1)
    beta_min = hi.log() - 0.5 * beta_d
    s = edm_steps.detach().clone().cpu()
    t = ((beta_min ** 2 + 2 * beta_d * (s ** 2 + 1).log()).sqrt() - beta_min) / beta_d
    return t, beta_min, beta_d + beta_min


def _integrand(beta_0, beta_1, taus):
    # alpha(t) = exp(-t^2 (b1 - b0)/2 - t b0);  -1/2 dlog(alpha)/dt / sqrt(alpha (1 - alpha)), derivative in closed form
    log_alpha = -0.5 * taus ** 2 * (beta_1 - beta_0) - taus * beta_0
    alpha = log_alpha.exp()
    return -0.5 * (-taus * (beta_1 - beta_0) - beta_0) / torch.sqrt(alpha * (1 - alpha))


def _lagrange(nodes, j, taus):
    p = 1
    for k in range(nodes.shape[0]):
        if k != j:
            p = p * (taus - nodes[k]) / (nodes[j] - nodes[k])
    return p


def _moment(lo, hi, roots, pivot):
    """integral over [lo, hi] of prod(tau - r for r in roots) / prod(pivot - r): 2 or 3 roots, closed form (rho-AB DEIS)."""
    if len(roots) == 2:
        a, b = roots
        v = (hi ** 3 - lo ** 3) / 3 - (hi ** 2 - lo ** 2) * (a + b) / 2 + (hi - lo) * a * b
        return v / ((pivot - a) * (pivot - b))
    a, b, c = roots
    v = (hi ** 4 - lo ** 4) / 4 - (hi ** 3 - lo ** 3) * (a + b + c) / 3 + (hi ** 2 - lo ** 2) * (a * b + a * c + b * c) / 2 - (hi - lo) * a * b * c
    return v / ((pivot - a) * (pivot - b) * (pivot - c))


def get_deis_coeff_list(t_steps, max_order, N=10000, deis_mode="tab"):
    out = []
    if deis_mode == "tab":
        t_steps, beta_0, beta_1 = edm2t(t_steps)
        for i in range(len(t_steps) - 1):
            order = min(i + 1, max_order)
            if order == 1:
                out.append([])
                continue
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            taus = torch.linspace(t_cur, t_next, N)
            dtau = (t_next - t_cur) / N
            nodes = t_steps[[i - k for k in range(order)]]
            f = _integrand(beta_0, beta_1, taus)
            out.append([torch.sum(f * _lagrange(nodes, j, taus)) * dtau for j in range(order)])
        return out
    if deis_mode != "rhoab":
        raise ValueError(deis_mode)
    for i in range(len(t_steps) - 1):
        order = min(i, max_order)
        if order == 0:
            out.append([])
            continue
        t_cur, t_next = t_steps[i], t_steps[i + 1]
        nodes = [t_steps[i - k] for k in range(order + 1)]  # nodes[0] == t_cur
        if order == 1:
            p1 = nodes[1]
            out.append([((t_next - p1) ** 2 - (t_cur - p1) ** 2) / (2 * (t_cur - p1)), (t_next - t_cur) ** 2 / (2 * (p1 - t_cur))])
        else:
            out.append([_moment(t_cur, t_next, [r for m, r in enumerate(nodes) if m != j], nodes[j]) for j in range(order + 1)])
    return out
