#!/usr/bin/env python
"""gp.predict timings between HIP events (all launches / the variance product alone) for A/B runs:
    VBMC_PREDICT_FUSED_FINISH=0|1 VBMC_KSTAR_ROWS_MIN=<M> python tools/predict_ab.py [config] [M ...]
Prints medians of 15 and a checksum of (fmu, fs2) so that variants can be compared."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Ms = [int(v) for v in sys.argv[2:]] or [8192]
ctx = _lib.Context(0)
_lib.set_default_context(ctx)
for S in (1, 4):
    wl = synthetic.make_workload(cfg, S=S)
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    for M in Ms:
        xs = np.random.default_rng(7).standard_normal((M, wl.D))
        fmu, fs2 = gp.predict(xs, separate_samples=True)
        pm, vm = [], []
        ctx.set_timing(1)
        for _ in range(15):
            gp.predict(xs, separate_samples=True)
            pm.append(ctx.last_kernel_ms(3))
        ctx.set_timing(2)
        for _ in range(15):
            gp.predict(xs, separate_samples=True)
            vm.append(ctx.last_kernel_ms(5))
        ctx.set_timing(False)
        print(f"cfg {cfg} S={S} M={M}: all launches {1e3 * np.median(pm):.2f} us (min {1e3 * min(pm):.2f}), "
              f"K* -> product events {1e3 * np.median(vm):.2f} us; sums {fmu.sum():.15e} {fs2.sum():.15e}", flush=True)
