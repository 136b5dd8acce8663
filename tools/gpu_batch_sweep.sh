#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bs && rocprofv3 --kernel-trace --output-format csv -d /tmp/bs -- python3 $GRAFT_REPO_ROOT/tools/batch_sweep.py
f=$(find /tmp/bs -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-40:]
    if "wino" not in n and "stem" not in n and "sepconv" not in n: continue
    agg[(n, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(agg.items()):
    v = v[len(v) // 3:]
    print(f"{n:42s} grid {g:8d}  n={len(v):5d}  mean {sum(v)/len(v):8.2f} us  min {min(v):8.2f}")
PY
