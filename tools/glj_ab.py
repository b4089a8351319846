import sys, time
sys.path.insert(0,'.')
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, synthetic
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.variational_optimization import _neg_elcbo_batch, _gp_log_joint
ctx=_lib.Context(0); _lib.set_default_context(ctx)
for cfg,N in ((3,400),(5,800)):
    wl=synthetic.make_workload(cfg, S=1)
    vp=VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1,-1), wl.lambd.reshape(-1,1)
    vp.w, vp.eta = wl.w.reshape(1,-1), wl.eta.reshape(1,-1)
    gp=gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd=synthetic.default_theta_bnd(wl)
    th=np.tile(wl.theta,(2500,1))+0.1*np.random.default_rng(0).standard_normal((2500,wl.theta.size))
    for _ in range(2): _neg_elcbo_batch(th, gp, vp, bnd)
    t0=time.perf_counter()
    for _ in range(5): _neg_elcbo_batch(th, gp, vp, bnd)
    t1=(time.perf_counter()-t0)/5
    for _ in range(50): _gp_log_joint(vp, gp, True)
    t0=time.perf_counter()
    for _ in range(500): _gp_log_joint(vp, gp, True)
    t2=(time.perf_counter()-t0)/500
    print(f"cfg {cfg}: sieve batch 2500: {1e3*t1:.3f} ms   _gp_log_joint grad: {1e6*t2:.1f} us")
