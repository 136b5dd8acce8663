#!/bin/bash
# Experiment build: tools/_ab/<name>.so = the product objects with ONE source file rebuilt under extra flags (git-ignored scratch).
#   tools/build_variant.sh <name> <file-without-.hip> <extra flags...>
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../ubdvss_amd/csrc"
mkdir -p ../../tools/_ab/_obj_var
extra=""
[ "$FILE" = "wino" ] && extra="-fno-slp-vectorize"
[ "$FILE" = "wino6" ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $extra "$@" -c $FILE.hip -o ../../tools/_ab/_obj_var/$NAME.o
objs=""
for f in api forward fwd16 wino wino6 postprocess loss backward train comm raster; do
  if [ "$f" = "$FILE" ]; then objs="$objs ../../tools/_ab/_obj_var/$NAME.o"; else objs="$objs _obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/$NAME.so $objs -ldl
echo "built tools/_ab/$NAME.so"
