#!/bin/bash
# round 2, first GPU session: parity tests, smoke, parity16 report, bench (new legs), baseline kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -40 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python tools/parity16_report.py ) > gpurun_out/parity16.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/bench.log 2> gpurun_out/bench.err
( timeout 300 python bench.py --gpus 2 --steps 5 --warmup 1 ) > gpurun_out/bench_gpus2.log 2>&1; echo "gpus2 rc=$?" >> gpurun_out/bench_gpus2.log
( UBD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_dist1.log 2>&1
tail -8 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -12 gpurun_out/parity16.log | cut -c1-400; tail -2 gpurun_out/bench.log | cut -c1-3000; tail -3 gpurun_out/bench.err; tail -3 gpurun_out/bench_gpus2.log; tail -1 gpurun_out/bench_dist1.log | cut -c1-600
