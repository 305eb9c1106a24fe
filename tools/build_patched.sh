#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# A/B builds that need a SOURCE edit of a kernel file without touching csrc/ (bench.py keys the committed PMC traffic on the hash of the
# library sources): sed the file into tools/_build/src/<name>/, compile both element-type builds, link with the main build's other objects.
#   usage: tools/build_patched.sh <name> '<sed expression>' [file.hip, default fmx_gemm256p.hip]        -> tools/_build/libfmx_<name>.so
set -e
NAME=$1; EXPR=$2; FILE=${3:-fmx_gemm256p.hip}; STEM=${FILE%.hip}
X=""; [[ $STEM == fmx_attention* ]] && X="-fno-slp-vectorize"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stable-diffusion-webui-forge_amd/csrc
OUT=$ROOT/tools/_build/$NAME
mkdir -p $OUT $ROOT/tools/_build/src/$NAME
sed "$EXPR" $CSRC/$FILE > $ROOT/tools/_build/src/$NAME/$FILE
if cmp -s $CSRC/$FILE $ROOT/tools/_build/src/$NAME/$FILE; then echo "the expression changed nothing" >&2; exit 1; fi
make -C $CSRC -j8 > /dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CSRC -Wno-unused-value"
/opt/rocm/bin/hipcc $BASE $X -c $ROOT/tools/_build/src/$NAME/$FILE -o $OUT/$STEM.o &
/opt/rocm/bin/hipcc $BASE $X -DFMX_ELEM_BF16 -c $ROOT/tools/_build/src/$NAME/$FILE -o $OUT/${STEM}_bf16.o &
wait
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o)
  if [[ $b == $STEM || $b == ${STEM}_bf16 ]]; then OBJS="$OBJS $OUT/$b.o"; else [ -f $CSRC/${b%_bf16}.hip ] && OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/_build/libfmx_$NAME.so
echo $ROOT/tools/_build/libfmx_$NAME.so
