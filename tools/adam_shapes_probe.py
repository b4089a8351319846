"""The optimiser loop at shapes a VBMC run passes through (few components and many GP hyper-parameter samples early, many
components and few samples late), ns_ent = 100 K^(2/3) samples in total: microseconds per iteration, four launches per
iteration against the one-launch form (csrc/adam_fused.hip), with the stopping rule on (never met: tol_fun tiny)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pyvbmc_amd import VariationalPosterior, _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd.minimize_adam import minimize_adam_elbo  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
shapes = [(10, 2, 25, 20), (10, 5, 20, 50), (10, 10, 12, 100), (10, 20, 8, 150), (10, 30, 6, 250), (10, 40, 4, 350), (10, 50, 4, 400),
          (10, 50, 2, 600), (10, 50, 1, 800), (4, 20, 8, 100), (16, 40, 3, 300)]
for D, K, S, N in shapes:
    nsk = 2 * max(1, int(np.ceil(np.ceil(100.0 * K ** (2.0 / 3.0) / K) / 2.0)))
    wl = synthetic.make_workload(3, S=S, D=D, K=K, N=N, Ns_total=nsk * K)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    res = {}
    for fused in (0, 1):
        ctx.set_option("adam_fused", fused)
        best = 1e9
        for r in range(4):
            t0 = time.perf_counter()
            out = minimize_adam_elbo(wl.theta.copy(), g, vp, wl.NsK, bnd, max_iter=400, tol_fun=1e-12, seed=11, rng="philox")
            best = min(best, (time.perf_counter() - t0) / out[4] * 1e6)
        res[fused] = (best, ctx.last_entmc_plan()["kernel"], out[3][-1])
    ctx.set_option("adam_fused", 1)
    print(f"D={D:2d} K={K:2d} S={S:2d} N={N:3d} NsK={wl.NsK:3d}: four launches {res[0][0]:6.2f} us ({res[0][1]}), one launch {res[1][0]:6.2f} us "
          f"({res[1][1]})  x{res[0][0] / res[1][0]:.2f}   F {res[0][2]:.8f} / {res[1][2]:.8f}")
