#!/bin/bash
# Builds libubd_hip.so (gfx950 only) in-tree next to the Python host package.
set -e
cd "$(dirname "$0")"
OUT=../libubd_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
mkdir -p _obj
# fingerprint of the kernel sources, compiled into the library (ubd_build_id): the same digest bench.csrc_sha16() takes of the files
BUILD_ID=$(python3 - <<'PY'
import glob, hashlib, os
h = hashlib.sha256()
for f in sorted(glob.glob("*.hip") + glob.glob("*.h")):
    h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
print(h.hexdigest()[:16])
PY
)
pids=()
for f in api forward fwd16 wino wino6 postprocess loss backward train comm raster; do
  [ -f $f.hip ] || continue
  extra=""
  # OpenCV-exact float geometry: no FMA contraction in postprocess
  [ "$f" = "postprocess" ] && extra="-ffp-contract=off"
  # Pillow-exact float32 scan-line arithmetic: no FMA contraction
  [ "$f" = "raster" ] && extra="-ffp-contract=off"
  # no SLP packing of adjacent fp32 adds into v_pk_add_f32: beside MFMAs the packed form issues slower than two scalar adds
  [ "$f" = "wino" ] && extra="$extra -fno-slp-vectorize"
  [ "$f" = "wino6" ] && extra="$extra -fno-slp-vectorize"
  stale=0
  for dep in $f.hip *.h ../../include/ubd.h; do [ "$dep" -nt _obj/$f.o ] && stale=1; done
  if [ "$f" = "api" ]; then   # carries the fingerprint of ALL kernel sources
    extra="$extra -DUBD_BUILD_ID=\"$BUILD_ID\""
    [ "$(cat _obj/api.build_id 2>/dev/null)" != "$BUILD_ID" ] && stale=1
  fi
  if [ ! -f _obj/$f.o ] || [ $stale = 1 ]; then
    ( /opt/rocm/bin/hipcc $FLAGS $extra -c $f.hip -o _obj/$f.o ${UBD_SAVE_TEMPS:+-save-temps=obj} ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in api forward fwd16 wino wino6 postprocess loss backward train comm raster; do [ -f _obj/$f.o ] && objs="$objs _obj/$f.o"; done
echo "$BUILD_ID" > _obj/api.build_id
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs -ldl
echo "built $OUT"
