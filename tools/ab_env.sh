#!/bin/bash
# A/B of an environment switch on ONE box: tools/ab_env.sh out.log VAR v1 v2 ...   (CFG, REPS, MINT, BENCH_ARGS as ab_libs.sh)
out=$1; var=$2; shift 2
REPS=${REPS:-3}
MINT=${MINT:-2.0}
CFG=${CFG:-3}
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    env $var=$v python bench.py --config $CFG --no-secondary --no-cpu-baseline --min-timed-s $MINT $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s' % '$var=$v', 'evals/s %8.1f' % d['value'], 'step us %6.2f' % (1e3*d['ms_per_step']), 'kernel us %6.2f' % (1e3*d['roofline']['kernel_ms']), 'F %.12g' % d['F'])"
  done
done 2>&1 | tee $out
