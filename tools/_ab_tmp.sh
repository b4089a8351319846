mkdir -p gpurun_out/s6
(timeout 1200 python -m pytest tests -m gpu -x -q -k "step or elbo or virtual or multibatch" ) 2>&1 | tail -4
for rep in 1 2 3; do for per in 0 -1 12; do
echo "== VBMC_MFMA_GP_PER=$per"
if [ $per = -1 ]; then unset VBMC_MFMA_GP_PER; else export VBMC_MFMA_GP_PER=$per; fi
python bench.py --config 5 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c5 step us', round(1e3*d['ms_per_step'],2), 'kernel us', round(1e3*d['roofline']['kernel_ms'],2), 'F', d['F'], d['host_us_per_step'])"
done; done 2>&1 | tee gpurun_out/s6/c5_ab1.log
