#!/bin/bash
# Shows that tests/test_gpu_comm_loopback.py::test_fused_step_ordering_* can catch a cross-stream ordering bug of the fused data-parallel
# step: builds two copies of the library with ONE event wait dropped in comm.hip (-DUBD_SABOTAGE_COMM=1: the communication stream
# does not wait for the gradients to be final; =2: the caller's stream does not wait for the reduced segment before Adam) and runs
# the ordering test against each (expected: FAILS) and against the product library (expected: passes).  GPU box only.
# Output: gpurun_out/comm_ordering_power.log
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/sab_obj
bash tests/loopback/build.sh > /dev/null
cd ubdvss_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function"
objs=""
for f in api forward fwd16 wino postprocess loss backward train raster; do objs="$objs _obj/$f.o"; done
for k in 1 2; do
  /opt/rocm/bin/hipcc $FLAGS -DUBD_SABOTAGE_COMM=$k -c comm.hip -o /tmp/sab_obj/comm$k.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/sab_obj/libubd_hip_sab$k.so $objs /tmp/sab_obj/comm$k.o -ldl || exit 1
done
cd ../..
{
  echo "== communication stream does not wait for the final gradients (UBD_SABOTAGE_COMM=1): the ordering test is expected to FAIL =="
  UBD_LIB_PATH=/tmp/sab_obj/libubd_hip_sab1.so python -m pytest tests/test_gpu_comm_loopback.py -m gpu -q -k "fused_step_ordering" 2>&1 | tail -12
  echo "== caller's stream does not wait for the reduced segment before Adam (UBD_SABOTAGE_COMM=2): expected to FAIL =="
  UBD_LIB_PATH=/tmp/sab_obj/libubd_hip_sab2.so python -m pytest tests/test_gpu_comm_loopback.py -m gpu -q -k "fused_step_ordering" 2>&1 | tail -12
  echo "== product library: expected to pass =="
  python -m pytest tests/test_gpu_comm_loopback.py -m gpu -q 2>&1 | tail -5
} > gpurun_out/comm_ordering_power.log 2>&1
cat gpurun_out/comm_ordering_power.log
