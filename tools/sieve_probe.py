"""The sieve's batch of candidate evaluations (reference _sieve, vbmc/variational_optimization.py:775-787) at
BASELINE config 3's shape: milliseconds per batch of 256 and of 2 500 (= 50 K) candidates."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, synthetic
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.variational_optimization import _neg_elcbo_batch
ctx = _lib.Context(0); _lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=1)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
bnd = synthetic.default_theta_bnd(wl)
rng = np.random.default_rng(0)
th0 = vp.get_parameters()
for B in (256, 2500):
    cands = th0[None, :] + 0.3 * rng.standard_normal((B, th0.size))
    _neg_elcbo_batch(cands, g, vp, bnd)
    ts = []
    for r in range(7):
        t0 = time.perf_counter(); F = _neg_elcbo_batch(cands, g, vp, bnd); ts.append(time.perf_counter() - t0)
    print(f"B={B}: {np.median(ts)*1e3:.3f} ms per batch = {np.median(ts)/B*1e6:.2f} us per candidate; F[0]={F[0]:.6f}")
