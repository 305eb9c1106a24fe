#!/bin/bash
export FMX_ALLOW_KNOBS=1   # the A/B knobs below are development switches: the library ignores them without this
# A/B builds of libfmx_gfx950.so: recompile the named csrc files (both element-type builds) with extra flags, link them with the
# main build's other objects into tools/_build/libfmx_<name>.so.  Select with FMX_LIB=tools/_build/libfmx_<name>.so (forge_amd/_lib.py).
#   usage: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [...]
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/stable-diffusion-webui-forge_amd/csrc
OUT=$ROOT/tools/_build/$NAME
mkdir -p $OUT
make -C $CSRC -j8 > /dev/null
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -Wno-unused-value"
OBJS=""
for o in $CSRC/*.o; do
  b=$(basename $o .o); src=${b%_bf16}
  hit=0; for f in "$@"; do [ "$(basename $f .hip)" == "$src" ] && hit=1; done
  if [ $hit == 1 ]; then
    X=""; [[ $b == *_bf16 ]] && X="-DFMX_ELEM_BF16"
    [[ $src == fmx_attention* ]] && X="$X -fno-slp-vectorize"
    /opt/rocm/bin/hipcc $BASE $X $FLAGS -c $CSRC/$src.hip -o $OUT/$b.o &
    OBJS="$OBJS $OUT/$b.o"
  else
    [ -f $CSRC/$src.hip ] && OBJS="$OBJS $o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/tools/_build/libfmx_$NAME.so
echo $ROOT/tools/_build/libfmx_$NAME.so
