#!/bin/bash
# round 2, second GPU session: full parity suite, then kernel stats of the forward bench (fused stem)
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
tail -6 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --no-cpu-baseline --no-train --steps 200 --warmup 20 ) > gpurun_out/bench_fwd.log 2>&1
tail -1 gpurun_out/bench_fwd.log | cut -c1-1800
bash tools/gpu_prof_py.sh tools/bench_warm.py
