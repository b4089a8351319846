#!/usr/bin/env python
"""Condense the MFMA counter pass over tools/bench_rows.py (tools/collect_profile.sh) into
profiles/<tag>_rows_mfma_pmc.csv: per kernel the mean SQ_INSTS_VALU, SQ_INSTS_VALU_MFMA_MOPS_F64,
SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES per dispatch and MFMA-busy / (4 x CU-busy).

    python tools/summarize_rows_pmc.py gpurun_out/<tag> <tag>
"""
import collections
import csv
import shutil
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path(__file__).resolve().parent.parent / "profiles"
names = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
raw = src / "pmc_rows_mfma" / "p_counter_collection.csv"
if len(sys.argv) > 3 and sys.argv[3] == "--condense":
    # on the GPU box (tools/collect_profile.sh): the per-dispatch file of this pass outgrows what gpurun copies back;
    # keep per kernel and counter the dispatch count and the sum
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(raw)):
        t = tot[(r["Kernel_Name"], r["Counter_Name"])]
        t[0] += 1
        t[1] += float(r["Counter_Value"])
    with open(src / "rows_mfma_pmc_condensed.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Sum"])
        for (k, c), (n, v) in sorted(tot.items()):
            w.writerow([k, c, n, v])
    sys.exit(0)
if raw.exists():
    for r in csv.DictReader(open(raw)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
else:
    for r in csv.DictReader(open(src / "rows_mfma_pmc_condensed.csv")):
        n = int(r["Dispatches"])
        acc[r["Kernel_Name"]][r["Counter_Name"]] = [float(r["Sum"]) / n] * n


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    return k.split("(")[0]


with open(out / f"{tag}_rows_mfma_pmc.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches"] + [n + "_avg" for n in names] + ["mfma_busy_over_4xCU_busy"])
    for k, v in sorted(acc.items(), key=lambda kv: short(kv[0])):
        if "__amd" in k:
            continue
        av = [sum(v[c]) / len(v[c]) if v[c] else 0.0 for c in names]
        w.writerow([short(k), len(v[names[0]])] + [round(a, 1) for a in av] + [round(av[2] / (4 * av[3]), 3) if av[3] else 0])
f = src / "stats_rows" / "s_kernel_stats.csv"
if f.exists():
    shutil.copy(f, out / f"{tag}_rows_kernel_stats.csv")
print(open(out / f"{tag}_rows_mfma_pmc.csv").read())
