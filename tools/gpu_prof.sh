#!/bin/bash
# rocprofv3 kernel trace of the bench step; summary copied to gpurun_out/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline ${UBD_PROF_ARGS:---no-train} > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/kernel_stats.csv 2>/dev/null
cut -c1-200 gpurun_out/kernel_stats.csv | head -40
tail -1 gpurun_out/prof_bench.log | cut -c1-3000
