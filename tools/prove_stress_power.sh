#!/bin/bash
# Shows that tests/test_gpu_postprocess.py::test_postprocess_stress_deterministic has the power to catch the round-2 flatten
# race: builds a copy of the library whose one-launch front end flattens with the compressing find again
# (-DUBD_PP_RACY_FLATTEN), runs the stress test against it (expected: FAILS) and against the product library (expected: passes).
# GPU box only.  Output: gpurun_out/stress_power.log
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/racy_obj
cd ubdvss_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -ffp-contract=off -DUBD_PP_RACY_FLATTEN -c postprocess.hip -o /tmp/racy_obj/postprocess.o || exit 1
objs=""
for f in api forward fwd16 wino loss backward train comm raster; do objs="$objs _obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/racy_obj/libubd_hip_racy.so $objs /tmp/racy_obj/postprocess.o -ldl || exit 1
cd ../..
{
  echo "== racy flatten (round-2 code): the stress test is expected to FAIL =="
  UBD_LIB_PATH=/tmp/racy_obj/libubd_hip_racy.so UBD_PP_STRESS_LAUNCHES=${UBD_PP_STRESS_LAUNCHES:-3000} python -m pytest tests/test_gpu_postprocess.py -m gpu -q -k "stress_deterministic and lds" 2>&1 | tail -15
  echo "== product library: expected to pass =="
  UBD_PP_STRESS_LAUNCHES=${UBD_PP_STRESS_LAUNCHES:-3000} python -m pytest tests/test_gpu_postprocess.py -m gpu -q -k "stress_deterministic and lds" 2>&1 | tail -5
} > gpurun_out/stress_power.log 2>&1
tail -30 gpurun_out/stress_power.log
