#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_forward.py tests/test_gpu_persistent.py tests/test_gpu_train.py tests/test_gpu_forward16.py -q -m gpu -x --timeout=900 2>&1 | tail -15 ) > gpurun_out/pytest_gpu_c.log 2>&1
tail -4 gpurun_out/pytest_gpu_c.log
timeout 300 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids
bash tools/gpu_prof_py.sh tools/bench_stem.py 2>&1 | grep -E "stem23|sepconv|Name" | cut -c1-200
for k in 1 2; do timeout 600 python tools/bench_train.py 64 bfloat16 2>&1 | tail -1; done
