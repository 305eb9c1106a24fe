"""Timing-only variants of csrc/fmx_gemm256p.hip, written to tools/_build/src/ (the library sources carry no such code since round 4):

  clock      workgroup 8 stamps s_memtime / s_memrealtime around its prologue, K loop and epilogue into the first words of its output tile
             (tools/clock_gemm.py reads them; results WRONG in 24 bytes per launch)
  dma        + only the first two K-tiles are fetched (what does the L2 -> LDS stream of the K loop cost?  results WRONG)
  lds_reads  + weight fragments read for k-step 0 only (32x32x16 loop; results WRONG)
  epi_lds    + half of the epilogue transpose's LDS writes (32x32x16 loop; results WRONG)

    python tools/patch_clock_stamps.py [clock|dma|lds_reads|epi_lds]   ->  tools/_build/src/clock_<variant>/fmx_gemm256p.hip
    tools/build_patched_file.sh <name> <patched file>                  ->  tools/_build/libfmx_<name>.so   (FMX_ALLOW_KNOBS=1 FMX_LIB=... selects it)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "stable-diffusion-webui-forge_amd", "csrc", "fmx_gemm256p.hip")

# (variant the edit belongs to, the line it follows, the lines inserted behind it)
EDITS = [
    ('clock', '  extern __shared__ __attribute__((aligned(16))) char smem[];',
     '  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();\n'),
    ('dma', '      char* sbase = smem + buf * STAGE_BYTES + wave * 1024;',
     "      if (c.t >= 2) return;   // their random data: a zero fill would change the operands' power) -- what does the L2 -> LDS stream of the K loop cost?\n"),
    ('lds_reads', '    for (int i = 0; i < MI; ++i) af[fb][i] = *reinterpret_cast<const f16x8*>(sa + lds_off(wm * (MI * 32) + i * 32 + li, ks * 2 + hi));',
     '    if (ks != 0) {\n#pragma unroll\n      for (int j = 0; j < NJ; ++j) wf[fb][j] = wf[fb ^ 1][j];\n      return;\n    }\n'),
    ('clock', '  //               wait + barrier                          k-step 3: + pieces 0-2 of tile t+2 (into the stage just released)',
     '  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();\n'),
    ('clock', '  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail\'s zero-fill pieces must land before the LDS is released',
     '  const unsigned dbg_cyc = (unsigned)(__builtin_amdgcn_s_memtime() - clk0), dbg_rt = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);\n'),
    ('epi_lds', '            const accv& a = acc[i][j];',
     '            if (q4 & 1) continue;\n'),
    ('clock', '  }',
     '  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n  __syncthreads();\n  if (wg == 8 && tid == 0) {  // overwrite the first words of this tile\'s output\n    unsigned* dbg = reinterpret_cast<unsigned*>(reinterpret_cast<f16*>(p.out) + (size_t)m0 * p.ld_out + (p.act == FMX_ACT_GEGLU ? n0 >> 1 : n0));\n    dbg[0] = dbg_cyc;\n    dbg[1] = dbg_rt;\n    dbg[2] = (unsigned)p.kt;\n    dbg[3] = 0x5eed5eedu;\n    dbg[4] = (unsigned)(rt0 - rt_entry);                                  // prologue, 10 ns ticks\n    dbg[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0) - dbg_rt;  // epilogue incl. store drain up to here\n  }\n'),
]


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else "clock"
    assert variant in ("clock", "dma", "lds_reads", "epi_lds"), variant
    s = open(SRC).read()
    for v, anchor, ins in EDITS:
        if v not in ("clock", variant):
            continue
        if anchor == "  }":   # the write-out goes in front of the tile loop's closing synchronisation
            anchor = "  // every wave is done with its epilogue slice of the LDS before the next tile's LDS-DMA pieces (any wave's) land in it"
            assert s.count(anchor) == 1
            s = s.replace(anchor, ins + anchor)
            continue
        assert s.count(anchor + "\n") == 1, (s.count(anchor + "\n"), anchor[:100])
        s = s.replace(anchor + "\n", anchor + "\n" + ins)
    dst = os.path.join(ROOT, "tools", "_build", "src", "clock_" + variant, "fmx_gemm256p.hip")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(s)
    print(dst)


if __name__ == "__main__":
    main()
