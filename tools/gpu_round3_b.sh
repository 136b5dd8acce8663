#!/bin/bash
# round 3: full parity suite, smoke, bench (all legs), rocprofv3 kernel stats of the forward and full legs
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -8 ) > gpurun_out/r3b_pytest_gpu.log 2>&1
tail -3 gpurun_out/r3b_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r3b_smoke.log 2>&1; tail -1 gpurun_out/r3b_smoke.log
( timeout 900 python bench.py ) > gpurun_out/r3b_bench.log 2> gpurun_out/r3b_bench.err; tail -c 300 gpurun_out/r3b_bench.log; tail -2 gpurun_out/r3b_bench.err
( UBD_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline ) > gpurun_out/r3b_bench_dist1.log 2>&1; grep -c metric gpurun_out/r3b_bench_dist1.log
cd /tmp
rm -rf /tmp/prof_fwd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fwd -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/gpurun_out/r3b_prof_bench_fwd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/full_size_avg.py $(find /tmp/prof_fwd -name "*kernel_trace.csv" | head -1) $(find /tmp/prof_fwd -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_bench_kernel_stats.csv
rm -rf /tmp/prof_full && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3b_prof_bench_full.log 2>&1
cp $(find /tmp/prof_full -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r03_bench_full_kernel_stats.csv
cut -c1-140 $GRAFT_REPO_ROOT/gpurun_out/r03_bench_kernel_stats.csv | head -14
