#!/bin/bash
# Kernel / copy timeline of host-driven ELBO evaluations (bench.py's step), from rocprofv3.
#   tools/step_timeline.sh <outdir> [bench args...]
# Prints the last full evaluations' launches with start / duration / end in microseconds.
OUT=${1:-gpurun_out/tl}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_tl -o tl -- \
  python "$REPO/bench.py" --steps 60 --warmup 5 --no-cpu-baseline "$@" > "$OUT/bench_under_trace.log" 2>&1
tail -1 "$OUT/bench_under_trace.log" | cut -c1-300
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("::")[-1].split("(")[0][:34]))
for f in glob.glob("/tmp/prof_tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[:20] + " " + r.get("Size", "")))
rows.sort()
# the last 3 evaluations: find the last finish kernels
idx = [i for i, r in enumerate(rows) if "entmc_finish" in r[2]]
lo = idx[-4] + 1 if len(idx) >= 4 else 0
t0 = rows[lo][0]
lines = []
prev_end = None
for s, e, nm in rows[lo:]:
    gap = "" if prev_end is None else " gap=%6.2f" % ((s - prev_end) / 1e3)
    lines.append("%-40s start=%8.2f dur=%7.2f end=%8.2f%s" % (nm, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, gap))
    prev_end = max(prev_end or e, e)
open(out + "/timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
PY
