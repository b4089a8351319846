for rep in 1 2; do for n in old base; do if [ "$n" = base ]; then unset VBMC_HIP_LIB; else export VBMC_HIP_LIB=$PWD/variants/libvbmc_$n.so; fi; python bench.py --no-cpu-baseline --min-timed-s 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$n', 'full_elcbo ms %.4f' % d['full_elcbo']['ms_per_eval'], 'predict all %.4f' % d['predict_roofline']['all_launches_ms'], 'adam %.2f' % d['device_resident_adam_loop']['us_per_iteration'], 'refstream %.4f' % d['reference_stream']['ms_per_eval'], 'S4 %.4f S8 %.4f' % (d['gp_samples']['S4']['ms_per_step'], d['gp_samples']['S8']['ms_per_step']))"; done; done
