#!/usr/bin/env python
"""gp.predict at the acquisition batch size in a loop (for rocprofv3 passes):
    python tools/predict_loop.py [config] [M] [S] [reps]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(cfg, S=S)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
xs = np.random.default_rng(7).standard_normal((M, wl.D))
for _ in range(reps):
    fmu, fs2 = gp.predict(xs, separate_samples=True)
print("done", float(fmu.sum()), float(fs2.sum()))
