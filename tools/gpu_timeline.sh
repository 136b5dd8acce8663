#!/bin/bash
# kernel timeline of the pipelined fwd+CCL step: per kernel of the last steps, start offset, duration and which other kernels overlap it
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $GRAFT_REPO_ROOT/tools/trace_step.py > /dev/null 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-42:], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows]
ev.sort()
# last 3 steps: find sepconv_kernel<3 starts
starts = [i for i, e in enumerate(ev) if "sepconv_kernel<3" in e[2] or "stem123_kernel" in e[2]]
i0 = starts[-4]; i1 = starts[-1]
t0 = ev[i0][0]
print("step period (us):", [(ev[starts[k + 1]][0] - ev[starts[k]][0]) / 1e3 for k in range(len(starts) - 6, len(starts) - 1)])
for s, e, n, q in ev[i0:i1]:
    ov = [n2[:14] for s2, e2, n2, q2 in ev[i0 - 12:i1 + 12] if q2 != q and s2 < e and e2 > s]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us q{q} {n:44s} | overlaps: {','.join(ov)}")
# average durations by kernel over the last 200 steps, split by whether a pp kernel overlapped
agg = collections.defaultdict(lambda: [[], []])
pp = [(s, e) for s, e, n, q in ev if n.startswith("pp_") or "pp_" in n]
import bisect
pps = sorted(pp)
for s, e, n, q in ev[len(ev) // 2:]:
    if "pp_" in n: continue
    o = any(s2 < e and e2 > s for s2, e2 in pps[max(0, bisect.bisect_left(pps, (s - 400000, 0))):bisect.bisect_right(pps, (e, 1 << 62))])
    agg[n][1 if o else 0].append((e - s) / 1e3)
print("\nmean duration (us) without / with a postprocess kernel running beside it:")
for n, (a, b) in agg.items():
    if len(a) + len(b) > 50:
        print(f"  {n:44s} alone {sum(a)/max(len(a),1):7.1f} (n={len(a)})   beside pp {sum(b)/max(len(b),1):7.1f} (n={len(b)})")
PY
