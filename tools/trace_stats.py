#!/usr/bin/env python
"""Per-kernel duration statistics of a rocprofv3 kernel trace with the cancelled launches of the
armed evaluation told apart (a timing-sampled step of bench.py cancels the evaluation
armed before it, whose three kernels then return at once -- 4 us entries that pull the plain
`--stats` average of the entropy kernel ~3 % below the duration of a launch that does its work).
    python tools/trace_stats.py <..._kernel_trace.csv> <out.csv>
A launch counts as cancelled when it lasts less than a quarter of its kernel's median."""
import csv
import statistics
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
dur = defaultdict(list)
with open(src) as f:
    for r in csv.DictReader(f):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for name, d in dur.items():
    med = statistics.median(d)
    done = [x for x in d if x >= 0.25 * med]
    rows.append((sum(d), name, len(d), sum(d) / len(d), len(d) - len(done), sum(done) / len(done), min(done), max(done), med))
rows.sort(reverse=True)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "AverageNs", "ReturnedAtOnce", "AverageNsOfTheOthers", "MinNsOfTheOthers", "MaxNs", "MedianNs"])
    for _, name, n, avg, canc, avg_done, mn, mx, med in rows:
        w.writerow([name, n, f"{avg:.1f}", canc, f"{avg_done:.1f}", mn, mx, f"{med:.1f}"])
