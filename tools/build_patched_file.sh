#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# Link tools/_build/libfmx_<name>.so from ONE already-patched copy of a kernel file (tools/patch_clock_stamps.py, tools/patch_tile_timeline.py, a hand edit)
# and the main build's other objects; csrc/ stays untouched.   usage: tools/build_patched_file.sh <name> <path/to/patched/fmx_xxx.hip>
set -e
NAME=$1; SRCFILE=$2; FILE=$(basename $SRCFILE); STEM=${FILE%.hip}
X=""; [[ $STEM == fmx_attention* ]] && X="-fno-slp-vectorize"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stable-diffusion-webui-forge_amd/csrc
OUT=$ROOT/tools/_build/$NAME
mkdir -p $OUT
make -C $CSRC -j8 > /dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CSRC -Wno-unused-value"
/opt/rocm/bin/hipcc $BASE $X -c $SRCFILE -o $OUT/$STEM.o &
/opt/rocm/bin/hipcc $BASE $X -DFMX_ELEM_BF16 -c $SRCFILE -o $OUT/${STEM}_bf16.o &
wait
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o)
  if [[ $b == $STEM || $b == ${STEM}_bf16 ]]; then OBJS="$OBJS $OUT/$b.o"; else [ -f $CSRC/${b%_bf16}.hip ] && OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/_build/libfmx_$NAME.so
echo $ROOT/tools/_build/libfmx_$NAME.so
