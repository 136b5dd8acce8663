#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script: tools/gpu_prof_py.sh <script.py> [args]; summary -> gpurun_out/kstats_<name>.csv
mkdir -p gpurun_out
S=$1; shift
NAME=$(basename $S .py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -- python3 $GRAFT_REPO_ROOT/$S "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/kstats_$NAME.csv 2>/dev/null
cut -c1-150 gpurun_out/kstats_$NAME.csv | head -${LINES_OUT:-30}
tail -4 gpurun_out/prof_$NAME.log | cut -c1-1000
