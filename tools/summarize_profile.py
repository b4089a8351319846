#!/usr/bin/env python
"""Condense a gpurun_out/<round>/ rocprofv3 collection into the tracked profiles/ files.

    python tools/summarize_profile.py gpurun_out/r01 r01

Expects the layout bench-profile runs produce (see profiles/README.md):
  <dir>/stats/s_kernel_stats.csv        rocprofv3 --kernel-trace --stats   (default bench, eps=philox)
  <dir>/stats_res/s_kernel_stats.csv    same, --rng resident
  <dir>/pmc_<NAME>/p_counter_collection.csv   one --pmc pass each
Writes profiles/<tag>_kernel_stats_{philox,resident}.csv, profiles/<tag>_pmc_summary.csv and
profiles/<tag>_pmc_summary.json (per-kernel means; HBM bytes corrected as MI355X_MICROARCH.md
prescribes: FETCH_SIZE is in KB and under-reports wide coalesced reads by 2x on gfx950).
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

src, tag = Path(sys.argv[1]), sys.argv[2]
out = Path(__file__).resolve().parent.parent / "profiles"
out.mkdir(exist_ok=True)
for mode, d in (("philox", "stats"), ("resident", "stats_res"), ("adam_loop", "stats_adam"), ("config2", "stats_c2"),
                ("config5", "stats_c5"), ("predict", "stats_predict")):
    f = src / d / "s_kernel_stats.csv"
    if f.exists():
        shutil.copy(f, out / f"{tag}_kernel_stats_{mode}.csv")
    f = src / d / "completed_stats.csv"  # tools/trace_stats.py: cancelled launches told apart
    if f.exists():
        shutil.copy(f, out / f"{tag}_kernel_stats_{mode}_completed.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(src.glob("pmc_*/p_counter_collection.csv")):
    if "rows_mfma" in p.parent.name:
        continue  # tools/summarize_rows_pmc.py
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for extra in ("timeline/timeline.txt", "gp_probe.txt", "adam_loop.txt", "rows.json"):
    f = src / extra
    if f.exists():
        shutil.copy(f, out / f"{tag}_{Path(extra).name}")
for f in sorted(src.glob("bench_*.json")):
    shutil.copy(f, out / f"{tag}_{f.name}")
rows = []
summary = {}
for k, v in acc.items():
    short = k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip() or k[:40]
    if "anonymous" in k and "::" in k:
        short = k.split("::")[1].split("(")[0].split("<")[0]
    if "entmc_ws_kernel<" in k:  # keep <DP, KTMAX, GRAD, EXACT, PHILOX>: the draw source changes the traffic
        short = "entmc_ws_kernel<" + k.split("entmc_ws_kernel<")[1].split(">")[0].replace(" ", "") + ">"
    summary.setdefault(short, {})
    for c, vals in v.items():
        m = sum(vals) / len(vals)
        rows.append((short, c, m, len(vals)))
        summary[short][c] = m
    s = summary[short]
    if "FETCH_SIZE" in s:
        s["hbm_read_bytes_corrected"] = s["FETCH_SIZE"] * 1024 * 2  # gfx950: x2 (MI355X_MICROARCH.md, HBM)
    if "WRITE_SIZE" in s:
        s["hbm_write_bytes_uncalibrated"] = s["WRITE_SIZE"] * 1024
with open(out / f"{tag}_pmc_summary.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
    w.writerows(rows)
json.dump(summary, open(out / f"{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("wrote", sorted(p.name for p in out.glob(f"{tag}_*")))
