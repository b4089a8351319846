#!/bin/bash
# Collect the rocprofv3 evidence profiles/README.md describes, on a GPU box:
#     gpurun -- 'bash tools/collect_profile.sh r02'
# then locally:  python tools/summarize_profile.py gpurun_out/r02 r02
#                python tools/summarize_rows_pmc.py gpurun_out/r02 r02
# Every pass runs from /tmp with TMPDIR=/tmp; the PMC passes carry --kernel-trace only.
set -u
TAG=${1:-prof}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
B="python $REPO/bench.py --no-cpu-baseline --min-timed-s 0.5"
# kernel-trace summaries of the bench command: the headline config (both draw sources), configs 2 and 5 (per-GPU
# shapes) and the two JOB sizes on one GPU (config 4: Ns = 8e6; config 5 --job: Ns = 4e6)
# (the headline pass runs the headline's launches only: the secondary figures launch the same kernel instantiation at
# other shapes -- S = 4 / 8 with fewer filler parts, the optimiser loop's launches with the pre workgroup -- and would
# enter its per-kernel mean; they get a pass of their own)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats     -o s -- $B --steps 50 --warmup 5 --no-secondary > $OUT/bench_under_rocprof_c3_philox.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sec -o s -- $B --steps 50 --warmup 5 > $OUT/bench_under_rocprof_c3_philox_secondary.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_res -o s -- $B --steps 50 --warmup 5 --rng resident --no-secondary > $OUT/bench_under_rocprof_c3_resident.json 2>/dev/null
for c in 2 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$c -o s -- $B --config $c --steps 50 --warmup 5 > $OUT/bench_under_rocprof_c$c.json 2>/dev/null
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4job -o s -- $B --config 4 --steps 20 --warmup 3 --no-secondary > $OUT/bench_under_rocprof_c4job.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5job -o s -- $B --config 5 --job --steps 20 --warmup 3 --no-secondary > $OUT/bench_under_rocprof_c5job.json 2>/dev/null
# Counter passes run with VBMC_ELBO_ARM=0: an armed evaluation that the next call cancels leaves three launches that
# return at once, and a per-launch counter mean cannot tell them apart.
export VBMC_ELBO_ARM=0
# HBM traffic of the kernels, one counter per pass
for c in 3 5 2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_c$c -o p -- $B --config $c --steps 10 --warmup 2 --no-secondary > /dev/null 2>&1
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_c4job -o p -- $B --config 4 --steps 6 --warmup 2 --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_c5job -o p -- $B --config 5 --job --steps 6 --warmup 2 --no-secondary > /dev/null 2>&1
done
# issue / occupancy counters, configs 3 and 5
for c in 3 5; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_SQ1_c$c -o p -- $B --config $c --steps 10 --warmup 2 --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_SQ2_c$c -o p -- $B --config $c --steps 10 --warmup 2 --no-secondary > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_SQ3_c$c -o p -- $B --config $c --steps 10 --warmup 2 --no-secondary > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_MFMA_c5 -o p -- $B --config 5 --steps 10 --warmup 2 --no-secondary > /dev/null 2>&1
unset VBMC_ELBO_ARM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_adam -o s -- python $REPO/tools/adam_loop_profile.py > $OUT/adam_loop.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_adam_small -o s -- python $REPO/tools/adam_small_probe.py child > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_rows -o s -- python $REPO/tools/bench_rows.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_rows_mfma -o p -- python $REPO/tools/bench_rows.py > /dev/null 2>&1
python $REPO/tools/summarize_rows_pmc.py $OUT $TAG --condense
# gp.predict at the acquisition batch size (M = 8192, config 3): durations, HBM traffic, matrix-pipe and LDS counters
P="python $REPO/tools/predict_loop.py 3 8192 1 20"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_predict -o s -- $P > /dev/null 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_${ctr}_predict -o p -- $P > /dev/null 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_MFMA_predict -o p -- $P > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_LDS_predict -o p -- $P > /dev/null 2>&1
# the host-driven step's timeline
bash $REPO/tools/step_timeline.sh $OUT/timeline --no-secondary > $OUT/timeline_stdout.txt 2>&1
# the same statistics with the launches an armed evaluation cancelled told apart (tools/trace_stats.py)
for d in stats stats_sec stats_res stats_c2 stats_c5 stats_c4job stats_c5job stats_adam stats_adam_small stats_predict; do
  t=$(find $OUT/$d -name "*_kernel_trace.csv" | head -1)
  [ -n "$t" ] && python $REPO/tools/trace_stats.py "$t" $OUT/$d/completed_stats.csv
done
# keep only the small per-pass summaries (the raw traces exceed what gpurun copies back)
find $OUT -name "*_kernel_trace.csv" -delete
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*.db" -delete
find $OUT -type f -size +8M -delete
# un-profiled bench lines of the same build
cd $REPO
python bench.py --steps 200 --warmup 20 > $OUT/bench_c3_philox.json 2>/dev/null
python bench.py --config 4 --steps 50 --warmup 5 > $OUT/bench_c4job.json 2>/dev/null
python bench.py --config 5 --job --steps 50 --warmup 5 > $OUT/bench_c5job.json 2>/dev/null
python tools/ws_k_probe.py 10 > $OUT/ws_k_probe_d10.txt 2>/dev/null
python tools/mfma_probe.py 10 12 16 20 24 32 > $OUT/mfma_probe.txt 2>/dev/null
VBMC_MFMA_ANY=1 python tools/mfma_c3_probe.py > $OUT/mfma_c3_probe.txt 2>/dev/null
python tools/sieve_probe.py > $OUT/sieve_probe.txt 2>/dev/null
python tools/adam_small_probe.py > $OUT/adam_small_probe.txt 2>/dev/null
python tools/adam_shapes_probe.py > $OUT/adam_shapes_probe.txt 2>/dev/null
python tools/adam_batch_probe.py > $OUT/adam_batch_probe.txt 2>/dev/null
VBMC_FUSED_TIMES=1 python tools/adam_small_probe.py child 2>&1 | awk '/^fused/ {c[$3]++; if (c[$3] % 4 == 1) {print; getline; print}; next} /phase A of/ {next} {print}' > $OUT/fused_phase_times.txt
./tools/ubench_gen2 > $OUT/ubench_gen2.txt 2>/dev/null
./tools/ubench_mfma_entropy > $OUT/ubench_mfma_entropy.txt 2>/dev/null
python bench.py --steps 200 --warmup 20 --rng resident --no-secondary > $OUT/bench_c3_resident.json 2>/dev/null
python bench.py --config 5 --steps 100 --warmup 10 > $OUT/bench_c5.json 2>/dev/null
python bench.py --config 2 --steps 200 --warmup 20 > $OUT/bench_c2.json 2>/dev/null
python tools/bench_rows.py > $OUT/rows.json 2>/dev/null
python tools/gp_probe.py > $OUT/gp_probe.txt 2>/dev/null
du -sh $OUT; ls $OUT
