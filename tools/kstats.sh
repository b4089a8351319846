#!/bin/bash
# Per-kernel duration statistics (rocprofv3 --kernel-trace, cancelled launches told apart) of a command:
#   tools/kstats.sh <outdir> <command ...>
OUT=$1; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ks -o ks -- "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
t=$(find /tmp/prof_ks -name "*_kernel_trace.csv" | head -1)
python "$REPO/tools/trace_stats.py" "$t" "$OUT/completed_stats.csv"
head -${KSTATS_ROWS:-12} "$OUT/completed_stats.csv" | cut -c1-220
