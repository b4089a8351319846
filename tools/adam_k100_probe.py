"""The optimiser loop at config 5's own shape at the reference's ns_ent (D = 20, K = 100, N = 800, NsK = 22): four launches
per iteration (K > 64 has no one-launch form).  python tools/adam_k100_probe.py [iters]   (tools/kstats.sh for its kernels)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyvbmc_amd import VariationalPosterior, _lib, synthetic
from pyvbmc_amd import gp as gpm
from pyvbmc_amd.minimize_adam import minimize_adam_elbo

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for cfg, D, K, N, nsk in [(5, 20, 100, 800, 22), (5, 20, 100, 800, 256), (3, 10, 50, 400, 4096)]:
    wl = synthetic.make_workload(cfg, S=1, D=D, K=K, N=N, Ns_total=nsk * K)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
               gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    g.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    kw = dict(max_iter=n_it, use_early_stopping=False, seed=11, rng="philox")
    best = 1e9
    for r in range(3):
        t0 = time.perf_counter()
        out = minimize_adam_elbo(wl.theta.copy(), g, vp, nsk, bnd, **kw)
        best = min(best, (time.perf_counter() - t0) / n_it * 1e6)
    print(f"D={D:2d} K={K:3d} N={N:4d} NsK={nsk:4d}: {best:6.2f} us per iteration   F {out[3][0]:.8f} -> {out[3][-1]:.8f}  "
          f"plan {ctx.last_entmc_plan()}", flush=True)
