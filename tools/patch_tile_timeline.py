"""Timing-only variant of csrc/fmx_gemm256p.hip (results WRONG in 28 bytes per tile): workgroup 8 stamps, per output tile it walks, the 100 MHz
real-time counter at the tile's top, in front of its K loop, behind the first K-tile's barrier, behind the K loop and behind the epilogue, into the
first words of that tile's output (tools/tile_timeline.py reads them).  The library sources stay untouched (bench.py keys the committed PMC traffic
on their hash): this script writes the patched copy to tools/_build/src/, tools/build_timeline.sh compiles it.

    python tools/patch_tile_timeline.py
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "stable-diffusion-webui-forge_amd", "csrc", "fmx_gemm256p.hip")
DST = os.path.join(ROOT, "tools", "_build", "src", "fmx_gemm256p.hip")

NOW = "(unsigned)(__builtin_amdgcn_s_memrealtime() - tl_entry)"
EDITS = [
    ("  const int tid = threadIdx.x;\n",
     "  const int tid = threadIdx.x;\n  const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime();\n"),
    ("  for (int lid = blockIdx.x; lid < nwg; lid += gridDim.x) {\n",
     "  for (int lid = blockIdx.x; lid < nwg; lid += gridDim.x) {\n  const unsigned tl_top = " + NOW + ";\n"),
    ("  if constexpr (MF == 16) {\n    // ---- 16x16x32 K loop",
     "  const unsigned tl_k0 = " + NOW + ";\n  unsigned tl_fb = 0;\n  if constexpr (MF == 16) {\n    // ---- 16x16x32 K loop"),
    ("      __builtin_amdgcn_s_barrier();\n      if constexpr (XT) {\n",
     "      __builtin_amdgcn_s_barrier();\n      { const unsigned now_ = " + NOW + "; tl_fb = t == 0 ? now_ : tl_fb; }\n      if constexpr (XT) {\n"),
    ("  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the tail's zero-fill pieces must land before the LDS is released\n",
     "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");  // the tail's zero-fill pieces must land before the LDS is released\n  const unsigned tl_k1 = " + NOW + ";\n"),
    ("  // every wave is done with its epilogue slice of the LDS before the next tile's LDS-DMA pieces (any wave's) land in it\n",
     "  {\n    const unsigned tl_e1 = " + NOW + ";\n    if (blockIdx.x == 8 && tid == 0) {\n      asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");   // (this wave stored these words itself, earlier)\n"
     "      unsigned* dbg = reinterpret_cast<unsigned*>(reinterpret_cast<f16*>(p.out) + (size_t)m0 * p.ld_out + (p.act == FMX_ACT_GEGLU ? n0 >> 1 : n0));\n"
     "      dbg[0] = tl_top; dbg[1] = tl_k0; dbg[2] = tl_fb; dbg[3] = tl_k1; dbg[4] = tl_e1; dbg[5] = 0x71e11e00u; dbg[6] = (unsigned)lid;\n    }\n  }\n"
     "  // every wave is done with its epilogue slice of the LDS before the next tile's LDS-DMA pieces (any wave's) land in it\n"),
]


def main():
    s = open(SRC).read()
    for a, b in EDITS:
        assert s.count(a) == 1, (s.count(a), a[:90])
        s = s.replace(a, b)
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    open(DST, "w").write(s)
    print(DST)


if __name__ == "__main__":
    main()
