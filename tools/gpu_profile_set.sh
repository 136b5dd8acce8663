#!/bin/bash
# The committed profile set of a round, in one gpurun call: bench lines (three default runs + one with the driver's flags), rocprofv3 kernel
# stats of the bench (forward only, and with the train legs) with full-size averages, HBM traffic tables (separate --pmc passes).
# Everything lands in gpurun_out/; copy to profiles/<round>_* with tools/collect_profiles.py <round>.
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python3 bench.py > gpurun_out/bench_default_$i.json 2> gpurun_out/bench_default_$i.err; tail -c 300 gpurun_out/bench_default_$i.json; echo; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_flags.json 2> gpurun_out/bench_driver_flags.err
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1 )
python3 tools/full_size_avg.py "$(find /tmp/prof -name '*kernel_trace.csv' | head -1)" "$(find /tmp/prof -name '*kernel_stats.csv' | head -1)" gpurun_out/bench_kernel_stats.csv
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_full.log 2>&1 )
python3 tools/full_size_avg.py "$(find /tmp/prof2 -name '*kernel_trace.csv' | head -1)" "$(find /tmp/prof2 -name '*kernel_stats.csv' | head -1)" gpurun_out/bench_full_kernel_stats.csv
UBD_PMC_DTYPE=float32 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1
UBD_PMC_DTYPE=float16 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1
UBD_PMC_DTYPE=bfloat16 UBD_PMC_TRAIN=1 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1
UBD_PMC_DTYPE=float32 UBD_PMC_TRAIN=1 bash tools/gpu_pmc_traffic.sh > /dev/null 2>&1
head -12 gpurun_out/bench_kernel_stats.csv | cut -c1-160
ls -la gpurun_out | tail -20
