#!/bin/bash
# A/B of library variants (tools/ws_variant.sh) on ONE box: the headline bench line, alternating, REPS rounds.
#   usage (on the GPU box): tools/ab_libs.sh out.log name1 name2 ...      ("base" = the shipped library)
# prints per run: evaluations/s, us per step, the main kernel's us (HIP events on its own dispatch), F
out=$1; shift
REPS=${REPS:-3}
MINT=${MINT:-2.0}
CFG=${CFG:-3}
for rep in $(seq 1 $REPS); do
  for n in "$@"; do
    if [ "$n" = base ]; then unset VBMC_HIP_LIB; else export VBMC_HIP_LIB=$PWD/variants/libvbmc_$n.so; fi
    python bench.py --config $CFG --no-secondary --no-cpu-baseline --min-timed-s $MINT $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s' % '$n', 'evals/s %8.1f' % d['value'], 'step us %6.2f' % (1e3*d['ms_per_step']), 'kernel us %6.2f' % (1e3*d['roofline']['kernel_ms']), 'F %.12g' % d['F'])"
  done
done 2>&1 | tee $out
