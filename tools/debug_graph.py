import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd
from forge_amd import synth
from forge_amd.backend.diffusion_engine.base import build_engine
cfg = synth.TINY_SDXL_UNET_CONFIG
eng = build_engine(cfg, synth.synth_unet_state_dict(cfg, seed=0), None, None, device="cuda")
km = eng.forge_objects.unet.model
c, uc = synth.synth_conditioning(2, cfg["context_dim"], cfg.get("adm_in_channels"), seed=1234)
cc = (c["crossattn"].cuda(), c["vector"].cuda()); uu = (uc["crossattn"].cuda(), uc["vector"].cuda())
g = torch.Generator().manual_seed(0)
xs = [torch.randn(2, 4, 16, 16, generator=g).cuda() * 10 for _ in range(6)]
def sig(v):
    from forge_amd.backend.modules.k_model import SigmaInfo
    s = torch.full((2,), v, device="cuda"); s.fmx_sigma = SigmaInfo([v, v]); return s
sigs = [14.6, 9.0, 5.0, 2.0, 1.0, 0.3]
def run(use_graph):
    km.use_graph = use_graph
    return [km.denoise_cfg(x, sig(s), uu, cc, 7.0).clone() for x, s in zip(xs, sigs)]
e1 = run(False); e2 = run(False); g1 = run(True); g2 = run(True)
for i in range(6):
    print(i, "eager-eager", float((e1[i]-e2[i]).abs().max()), "eager-graph", float((e1[i]-g1[i]).abs().max()), "graph-graph", float((g1[i]-g2[i]).abs().max()))
