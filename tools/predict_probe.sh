#!/bin/bash
# Kernel durations of gp.predict at the acquisition batch size (bench.py's predict section) from rocprofv3.
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > /tmp/pp.log 2>&1
tail -1 /tmp/pp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['predict_roofline'])"
python - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "predict" in r["Name"] or "trinv" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us %.1f"%(float(r["AverageNs"])/1e3))
PY
