#!/bin/bash
# One GPU-box session: parity tests, smoke, bench; everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -x --timeout=600 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
