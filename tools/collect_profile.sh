#!/bin/bash
# Collect the rocprofv3 evidence profiles/README.md describes, on a GPU box:
#     gpurun -- 'bash tools/collect_profile.sh r01c'
# then locally:  python tools/summarize_profile.py gpurun_out/r01c r01c
#                python tools/summarize_rows_pmc.py gpurun_out/r01c r01c
set -u
TAG=${1:-prof}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats     -o s -- $B --steps 50 --warmup 5 > $OUT/bench_under_rocprof_philox.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_res -o s -- $B --steps 50 --warmup 5 --rng resident > $OUT/bench_under_rocprof_resident.json 2>/dev/null
for mode in philox resident; do
  extra=""; [ $mode = resident ] && extra="--rng resident"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${c}_$mode -o p -- $B --steps 10 --warmup 2 $extra > /dev/null 2>&1
  done
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_SQ1 -o p -- $B --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_SQ2 -o p -- $B --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_SQ3 -o p -- $B --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_adam -o s -- python $REPO/tools/adam_loop_profile.py > $OUT/adam_loop.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_rows -o s -- python $REPO/tools/bench_rows.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_rows_mfma -o p -- python $REPO/tools/bench_rows.py > /dev/null 2>&1
# un-profiled bench lines of the same build
cd $REPO
python bench.py --steps 200 --warmup 20 > $OUT/bench_philox.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --rng resident > $OUT/bench_resident.json 2>/dev/null
python tools/bench_rows.py > $OUT/rows.json 2>/dev/null
ls $OUT
