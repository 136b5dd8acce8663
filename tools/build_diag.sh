#!/bin/bash
# Diagnostic build: libubd_hip_diag.so = the product sources + -DUBD_STAMPS (in-kernel s_memtime stamps, an extra
# exported ubd_debug_set_stamps).  Written to tools/_ab/ (git-ignored scratch, emptied before the round ends). Never loaded by the package; tools/stamps_*.py load it explicitly.
set -e
cd "$(dirname "$0")/../ubdvss_amd/csrc"
OBJ=../../tools/_ab/_obj_diag
mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -DUBD_STAMPS $DIAG_FLAGS"   # DIAG_FLAGS / DIAG_OUT: experiment builds
pids=()
for f in ${DIAG_FILES:-api forward fwd16 wino wino6 postprocess loss backward train comm raster}; do
  extra=""
  [ "$f" = "postprocess" ] && extra="-ffp-contract=off"
  [ "$f" = "raster" ] && extra="-ffp-contract=off"
  [ "$f" = "wino" ] && extra="$extra -fno-slp-vectorize"
  [ "$f" = "wino6" ] && extra="$extra -fno-slp-vectorize"
  ( /opt/rocm/bin/hipcc $FLAGS $extra -c $f.hip -o $OBJ/$f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/${DIAG_OUT:-libubd_hip_diag.so} $OBJ/*.o -ldl
echo "built ${DIAG_OUT:-libubd_hip_diag.so}"
