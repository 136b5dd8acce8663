#!/bin/bash
# per-kernel stats of the train step (rocprofv3 --kernel-trace --stats), top kernels; $1 = activation dtype (bfloat16 default, float32, float16)
DT=${1:-bfloat16}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py $DT > /tmp/ts.log 2>&1
grep "train step" /tmp/ts.log
f=$(find /tmp/ts -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out && cp $f $GRAFT_REPO_ROOT/gpurun_out/train_kernel_stats_$DT.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:24]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  per step {float(r['TotalDurationNs'])/360/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
