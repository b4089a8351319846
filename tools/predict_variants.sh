#!/bin/bash
# Kernel durations of gp.predict (config 3, M = 8192) for every library under variants/ (and the shipped one).
cd /tmp && export TMPDIR=/tmp
for lib in $GRAFT_REPO_ROOT/pyvbmc_amd/libvbmc_hip.so $GRAFT_REPO_ROOT/variants/libvbmc_*.so; do
  [ -f "$lib" ] || continue
  rm -rf /tmp/pv
  VBMC_HIP_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o p -- python $GRAFT_REPO_ROOT/tools/predict_loop.py ${1:-3} ${2:-8192} ${3:-1} 20 > /tmp/pv.log 2>&1
  python - "$lib" <<PY
import csv,glob,sys
f=glob.glob("/tmp/pv/**/*kernel_stats.csv",recursive=True)
out=[]
if f:
    for r in csv.DictReader(open(f[0])):
        if "predict_var" in r["Name"] or "kstar" in r["Name"]: out.append("%s %.1f"%(r["Name"].split("::")[-1][:22], float(r["AverageNs"])/1e3))
print(sys.argv[1].split("/")[-1], " | ".join(out))
PY
done
