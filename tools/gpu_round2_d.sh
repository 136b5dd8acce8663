#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -6 ) > gpurun_out/pytest_gpu_d.log 2>&1
tail -3 gpurun_out/pytest_gpu_d.log
timeout 300 python tools/bench_stem.py 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/gpu_prof_py.sh tools/bench_stem.py 2>&1 | grep -E "stem23|sepconv|Name" | cut -c1-200
timeout 600 python tools/bench_train.py 64 bfloat16 2>&1 | tail -1
LINES_OUT=16 bash tools/gpu_prof_py.sh tools/bench_train.py 64 bfloat16 2>&1 | cut -c1-175 | head -16
