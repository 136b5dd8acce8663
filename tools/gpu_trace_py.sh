#!/bin/bash
# per-launch kernel trace of a python script: tools/gpu_trace_py.sh <script.py> <kernel-name-pattern>; prints the last launches
S=$1; PAT=$2; NAME=$(basename $S .py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$NAME && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$NAME -- python3 $GRAFT_REPO_ROOT/$S > /dev/null 2>&1
f=$(find /tmp/tr_$NAME -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$PAT" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
for r in rows[-int(len(rows) / 3):]:
    print(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, "us", "grid", r.get("Grid_Size_X", r.get("Grid_Size", "")))
PY
