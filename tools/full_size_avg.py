"""Adds a FullSizeAverageNs column to a rocprofv3 kernel_stats CSV: the mean duration of a kernel's FULL-SIZE launches only.
bench.py launches the dominant kernels on the headline batch (32 x 128 x 128 maps) and, in its batch-1 latency leg, on single
128 x 128 / 256 x 256 maps for a few microseconds each; the plain AverageNs mixes the two.  Full size = launches of the kernel's
largest grid (kernel trace, Grid_Size columns); a persistent kernel whose grid is the CU count at every size (round 6: the Winograd layer spreads
a single image's 256 groups over all CUs) is split by duration instead: of its largest-grid launches, those that take at least half of the
90th-percentile duration.  usage: full_size_avg.py <kernel_trace.csv> <kernel_stats.csv> <out.csv>"""
import collections, csv, sys
trace, stats, out = sys.argv[1:4]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    grid = 1
    for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Grid_Size"):
        if k in r and r[k]:
            grid *= int(r[k])
    dur[r["Kernel_Name"]].append((grid, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = list(csv.DictReader(open(stats)))
fields = list(rows[0].keys()) + ["FullSizeCalls", "FullSizeAverageNs"]
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=fields, quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    for r in rows:
        d = dur.get(r["Name"], [])
        if d:
            gmax = max(g for g, _ in d)
            full = sorted(t for g, t in d if g == gmax)
            p90 = full[min(len(full) - 1, int(0.9 * len(full)))]
            full = [t for t in full if t >= 0.5 * p90]
            r["FullSizeCalls"], r["FullSizeAverageNs"] = len(full), round(sum(full) / len(full), 3)
        else:
            r["FullSizeCalls"], r["FullSizeAverageNs"] = "", ""
        w.writerow(r)
