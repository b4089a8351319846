#!/bin/bash
# Per-kernel PMC averages for a command:  tools/pmc_kernels.sh <pattern> <counters...> -- <cmd...>
PAT=$1; shift
CTRS=()
while [ "$1" != "--" ]; do CTRS+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmck
rocprofv3 --pmc "${CTRS[@]}" --kernel-trace --output-format csv -d /tmp/pmck -o p -- "$@" > /tmp/pmck.log 2>&1
python - "$PAT" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmck/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
PY
