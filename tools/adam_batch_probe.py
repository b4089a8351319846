"""Per-iteration wall time of minimize_adam_elbo at the reference's default sample count (K = 50, NsK = 28) with and
without the stopping rule's batches of 20, four launches per iteration against one launch per batch, and where a
batch's host time goes (the C call alone, the stopping rule alone)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyvbmc_amd import VariationalPosterior, _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd import minimize_adam as ma  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=1)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
bnd = synthetic.default_theta_bnd(wl)
for fused in (0, 1):
    ctx.set_option("adam_fused", fused)
    for es in (False, True):
        best = 1e9
        for r in range(5):
            t0 = time.perf_counter()
            out = ma.minimize_adam_elbo(wl.theta.copy(), g, vp, 28, bnd, max_iter=400, use_early_stopping=es, tol_fun=1e-12,
                                        seed=11, rng="philox")
            best = min(best, (time.perf_counter() - t0) / out[4] * 1e6)
        print(f"fused={fused} early_stopping={es}: {best:.2f} us per iteration over {out[4]} iterations")
# the stopping rule alone
y = np.random.default_rng(0).standard_normal(20)
xr = np.random.default_rng(1).standard_normal((40, 610))
t0 = time.perf_counter()
for _ in range(2000):
    ma._window_stop(y, xr[:20].mean(axis=0), xr[20:].mean(axis=0), 1e-3)
print(f"stopping rule (fit + two means of 20 x 610): {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per batch")
# the C call alone: batches of 20 through the raw entry points
os.environ["ADAM_BATCH_PROBE"] = "1"
orig = ma._window_stop
ma._window_stop = lambda *a, **k: False
for fused in (0, 1):
    ctx.set_option("adam_fused", fused)
    best = 1e9
    for r in range(5):
        t0 = time.perf_counter()
        out = ma.minimize_adam_elbo(wl.theta.copy(), g, vp, 28, bnd, max_iter=400, use_early_stopping=True, tol_fun=1e-12, seed=11,
                                    rng="philox")
        best = min(best, (time.perf_counter() - t0) / out[4] * 1e6)
    print(f"fused={fused} batches of 20 without the stopping rule's arithmetic: {best:.2f} us per iteration")
ma._window_stop = orig
