"""Three runs of the device-resident Adam loop at BASELINE config 3 (200 iterations each); run under
rocprofv3 --kernel-trace --stats by tools/collect_profile.sh for the per-kernel breakdown."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyvbmc_amd import _lib, synthetic, VariationalPosterior
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.minimize_adam import minimize_adam_elbo
ctx = _lib.Context(0); _lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=1)
vp = VariationalPosterior(wl.D, wl.K); vp.mu = wl.mu.copy(); vp.sigma = wl.sigma.reshape(1,-1).copy(); vp.lambd = wl.lambd.reshape(-1,1).copy(); vp.w = wl.w.reshape(1,-1).copy(); vp.eta = wl.eta.reshape(1,-1).copy(); vp.ctx = ctx
g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True)); g.ctx = ctx
g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
bnd = synthetic.default_theta_bnd(wl)
NSK = int(sys.argv[1]) if len(sys.argv) > 1 else wl.NsK
kw = dict(max_iter=200, master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=False)
for r in range(3):
    t0 = time.perf_counter()
    out = minimize_adam_elbo(wl.theta.copy(), g, vp, NSK, bnd, seed=11, rng="philox", **kw)
    print("per-iter us", (time.perf_counter() - t0) / 200 * 1e6, out[3][0], out[3][-1])
