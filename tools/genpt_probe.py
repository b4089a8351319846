import sys, time, json
sys.path.insert(0,'.')
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, synthetic
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.variational_optimization import _neg_elcbo
ctx=_lib.Context(0); _lib.set_default_context(ctx)
for cfg, ns in ((3,1_000_000),(5,500_000)):
    wl=synthetic.make_workload(cfg, Ns_total=ns)
    vp=VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1,-1), wl.lambd.reshape(-1,1)
    vp.w, vp.eta = wl.w.reshape(1,-1), wl.eta.reshape(1,-1)
    gp=gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd=synthetic.default_theta_bnd(wl); theta=wl.theta.copy()
    for pt in (1,2,4,8,1):
        ctx.set_option("gen_pt", pt)
        n=[0]
        def step():
            n[0]+=1
            return _neg_elcbo(theta+1e-9*(n[0]%7), gp, vp, 0.0, wl.NsK, True, False, bnd, rng="philox", seed=n[0])
        for _ in range(300): step()
        ctx.synchronize(); t0=time.perf_counter()
        N=6000 if cfg==3 else 3000
        for _ in range(N): step()
        ctx.synchronize(); dt=time.perf_counter()-t0
        print(f"cfg {cfg} gen_pt={pt}: {1e6*dt/N:.2f} us/step", flush=True)
