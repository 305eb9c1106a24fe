"""Per-wave s_memtime timeline of one K-tile of the 256x256 GEMM (library built with EXTRA=-DFMX_ABLATE, FMX_ABL=128)."""
import os
import sys

os.environ["FMX_ABL"] = "128"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops

m = n = k = 4096
x = (torch.randn(m, k, device="cuda")).half()
w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
out = torch.zeros(m, n, dtype=torch.float16, device="cuda")
for _ in range(3):
    ops.conv_gemm(x, w, n, out=out, ld_out=n, force_tile=4)
torch.cuda.synchronize()
raw = out.view(-1)[:200 * 2].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
st = raw[:192].reshape(8, 24)
cyc, rt, kt = int(raw[192]), int(raw[193]), int(raw[194])
print(f"K loop of block 8: {cyc} shader cycles, {rt} realtime ticks (100 MHz) over {kt} K-tiles -> {cyc / max(kt, 1):.0f} cycles per K-tile, "
      f"effective clock {cyc / max(rt, 1) * 0.1:.3f} GHz")
t0 = st[:, 0].min()
names = ["start", "dma0_issued", "ds_issued", "barrier1_passed", "mfma+dma1_issued", "barrier2_passed"]
print("wave  phase | " + " ".join(f"{n_:>15s}" for n_ in names))
for wv in range(8):
    for ph in range(4):
        row = st[wv, ph * 6:(ph + 1) * 6] - t0
        order = [0, 6, 1, 2, 3, 4, 5]
        print(f"  w{wv} g{wv // 4} P{ph + 1} | " + " ".join(f"{int(row[i]):>15d}" for i in range(6)))
