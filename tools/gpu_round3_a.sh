#!/bin/bash
# round 3, first pass: full parity suite (incl. stress + loopback 2-rank tests), racy-flatten proof, postprocess stamps, bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -12 ) > gpurun_out/r3a_pytest_gpu.log 2>&1
tail -3 gpurun_out/r3a_pytest_gpu.log
timeout 900 bash tools/prove_stress_power.sh > /dev/null 2>&1; tail -12 gpurun_out/stress_power.log
( bash tools/build_diag.sh && timeout 300 python tools/stamps_pp.py ) > gpurun_out/r3a_stamps_pp.log 2>&1; tail -6 gpurun_out/r3a_stamps_pp.log
( timeout 900 python bench.py ) > gpurun_out/r3a_bench.log 2> gpurun_out/r3a_bench.err; tail -c 400 gpurun_out/r3a_bench.log; tail -2 gpurun_out/r3a_bench.err
