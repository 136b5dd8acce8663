#!/bin/bash
# round 4, first pass: the parity suite after the housekeeping changes + the driver's bench command with the reworked legs
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -12 ) > gpurun_out/r4a_pytest_gpu.log 2>&1
tail -3 gpurun_out/r4a_pytest_gpu.log
( timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4a_bench_driver.log 2> gpurun_out/r4a_bench_driver.err; tail -c 600 gpurun_out/r4a_bench_driver.log; tail -2 gpurun_out/r4a_bench_driver.err
( timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4a_bench_500.log 2> gpurun_out/r4a_bench_500.err; tail -c 300 gpurun_out/r4a_bench_500.log; tail -2 gpurun_out/r4a_bench_500.err
