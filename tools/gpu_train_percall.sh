#!/bin/bash
# per-launch durations of one kernel (name pattern $2) inside the train step, by launch position within the step ($1 = dtype): which dilation /
# layer costs what.   tools/gpu_train_percall.sh float32 dil_wgrad_kernel
DT=${1:-float32}; PAT=${2:-dil_wgrad_kernel}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py $DT > /tmp/tp.log 2>&1
f=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$PAT" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = collections.OrderedDict()
per = None
# launches per step: find the period from the first kernel name sequence
seq = [r["Kernel_Name"] for r in rows]
for p in range(1, 64):
    if all(seq[i] == seq[i + p] for i in range(min(len(seq) - p, 200))): per = p; break
import os
if os.environ.get("PER"): per = int(os.environ["PER"])
print("launches per step:", per, "of", len(rows))
d = collections.defaultdict(list)
for i, r in enumerate(rows[len(rows) // 2 // per * per:]):
    d[i % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in range(per):
    v = sorted(d[k]); print(f"  launch {k}: median {v[len(v)//2]:8.1f} us   {seq[k][:70]}")
PY
