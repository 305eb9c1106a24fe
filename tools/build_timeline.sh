#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# tools/_build/libfmx_timeline.so: the library with the tile-timeline stamps of tools/patch_tile_timeline.py in the 8-wave GEMM kernels
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stable-diffusion-webui-forge_amd/csrc
OUT=$ROOT/tools/_build/timeline
mkdir -p $OUT
python $ROOT/tools/patch_tile_timeline.py > /dev/null
make -C $CSRC -j8 > /dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CSRC -Wno-unused-value"
/opt/rocm/bin/hipcc $BASE -c $ROOT/tools/_build/src/fmx_gemm256p.hip -o $OUT/fmx_gemm256p.o &
/opt/rocm/bin/hipcc $BASE -DFMX_ELEM_BF16 -c $ROOT/tools/_build/src/fmx_gemm256p.hip -o $OUT/fmx_gemm256p_bf16.o &
wait
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o)
  if [[ $b == fmx_gemm256p* ]]; then OBJS="$OBJS $OUT/$b.o"; else [ -f $CSRC/${b%_bf16}.hip ] && OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/_build/libfmx_timeline.so
echo $ROOT/tools/_build/libfmx_timeline.so
