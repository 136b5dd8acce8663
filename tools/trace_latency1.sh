#!/bin/bash
# kernel timeline of one-image forward passes (512 x 512 x 1): rocprofv3 kernel trace of tools/bench_latency1.py, the last pass's kernels with start offsets
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt && rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 $GRAFT_REPO_ROOT/tools/bench_latency1.py > /tmp/lt.log 2>&1
python3 - "$(find /tmp/lt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last 9-kernel pass before the 1024 runs begin: find passes by the stem kernel name
idx = [i for i, r in enumerate(rows) if "stem123" in r["Kernel_Name"] or "sepconv_kernel<1" in r["Kernel_Name"]]
for start in (idx[40], idx[-3]):
    t0 = int(rows[start]["Start_Timestamp"])
    for r in rows[start:start + 9]:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.2f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.2f}  {r['Kernel_Name'][:70]}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
    print()
PY
