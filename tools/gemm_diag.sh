#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# s_memtime stamps + timing ablations of the 256x256 GEMM (needs the -DFMX_ABLATE build: libfmx_ablate_gfx950.so)
TAG=${1:-diag}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
export FMX_LIB=$R/stable-diffusion-webui-forge_amd/libfmx_ablate_gfx950.so
cd $R
timeout 300 python tools/stamp_gemm.py > $O/stamps.txt 2>&1
cat $O/stamps.txt | tail -40
ABLS=${ABLS:-0,1,2,4,6,8,10,12,14} timeout 600 python tools/ablate_gemm.py 2>&1 | grep abl > $O/ablation.jsonl
cat $O/ablation.jsonl
