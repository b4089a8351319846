"""The default draw source (rng="numpy": the reference's MT19937 stream generated on the device) at BASELINE config 3:
ms per evaluation; under tools/kstats.sh the generator's kernels.   python tools/randn_probe.py [evals]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyvbmc_amd import VariationalPosterior, _lib, synthetic
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.variational_optimization import _neg_elcbo

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(3)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
bnd = synthetic.default_theta_bnd(wl)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
np.random.seed(3)
_neg_elcbo(wl.theta.copy(), g, vp, 0.0, wl.NsK, True, False, bnd, rng="numpy")
ts = []
for _ in range(n):
    t0 = time.perf_counter()
    out = _neg_elcbo(wl.theta.copy(), g, vp, 0.0, wl.NsK, True, False, bnd, rng="numpy")
    ts.append(time.perf_counter() - t0)
print("rng=numpy: %.3f ms per evaluation (median of %d; p10 %.3f p90 %.3f)  F %.10f" % (
    1e3 * np.median(ts), n, 1e3 * np.percentile(ts, 10), 1e3 * np.percentile(ts, 90), out[0]))
# the entropy kernel's own duration in this path (HIP events on its dispatch), against the Philox path's
ctx.set_timing(True)
ms = {"numpy": [], "philox": []}
for i in range(12):
    _neg_elcbo(wl.theta.copy(), g, vp, 0.0, wl.NsK, True, False, bnd, rng="numpy")
    ms["numpy"].append(ctx.last_kernel_ms(0))
    _neg_elcbo(wl.theta.copy(), g, vp, 0.0, wl.NsK, True, False, bnd, rng="philox", seed=100 + i)
    ms["philox"].append(ctx.last_kernel_ms(0))
ctx.set_timing(False)
print("entropy kernel by events: behind the NumPy stream %.2f us, with Philox draws %.2f us (alternating, medians)" % (
    1e3 * np.median(ms["numpy"]), 1e3 * np.median(ms["philox"])))
