#!/bin/bash
# A/B builds that need a SOURCE edit of the 8-wave GEMM kernel without touching csrc/ (bench.py keys the committed PMC traffic on the hash of the
# library sources): sed the file into tools/_build/src/<name>/, compile both element-type builds, link with the main build's other objects.
#   usage: tools/build_patched.sh <name> '<sed expression>'        -> tools/_build/libfmx_<name>.so
set -e
NAME=$1; EXPR=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stable-diffusion-webui-forge_amd/csrc
OUT=$ROOT/tools/_build/$NAME
mkdir -p $OUT $ROOT/tools/_build/src/$NAME
sed "$EXPR" $CSRC/fmx_gemm256p.hip > $ROOT/tools/_build/src/$NAME/fmx_gemm256p.hip
if cmp -s $CSRC/fmx_gemm256p.hip $ROOT/tools/_build/src/$NAME/fmx_gemm256p.hip; then echo "the expression changed nothing" >&2; exit 1; fi
make -C $CSRC -j8 > /dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$CSRC -Wno-unused-value"
/opt/rocm/bin/hipcc $BASE -c $ROOT/tools/_build/src/$NAME/fmx_gemm256p.hip -o $OUT/fmx_gemm256p.o &
/opt/rocm/bin/hipcc $BASE -DFMX_ELEM_BF16 -c $ROOT/tools/_build/src/$NAME/fmx_gemm256p.hip -o $OUT/fmx_gemm256p_bf16.o &
wait
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o)
  if [[ $b == fmx_gemm256p* ]]; then OBJS="$OBJS $OUT/$b.o"; else [ -f $CSRC/${b%_bf16}.hip ] && OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/_build/libfmx_$NAME.so
echo $ROOT/tools/_build/libfmx_$NAME.so
