#!/usr/bin/env python
"""Time vbmc_set_gp (upload + L^-1) and its kernels' share for a few GP sizes.
    python tools/gp_probe.py            (rocprofv3 --kernel-trace --stats gives the kernel split)
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
for N, S in ((400, 1), (400, 8), (800, 1), (800, 8), (1024, 1)):
    wl = synthetic.make_workload(3, S=S, N=N)
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    gpm.upload_gp(gp, ctx)
    ts = []
    for _ in range(10):
        gpm.invalidate_gp(ctx)
        t0 = time.perf_counter()
        gpm.upload_gp(gp, ctx)
        ts.append(time.perf_counter() - t0)
    xs = np.random.default_rng(1).standard_normal((256, wl.D))
    fmu, fs2 = gp.predict(xs, separate_samples=True)
    print(f"N={N} S={S}: upload_gp (host gather + vbmc_set_gp incl. L^-1) median {1e3 * np.median(ts):.3f} ms, "
          f"min {1e3 * min(ts):.3f} ms; fs2 range [{fs2.min():.3e}, {fs2.max():.3e}]", flush=True)
