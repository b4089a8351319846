"""The optimiser loop for shapes beyond D = 16 / N ~ 900 (round 5): one launch per batch against four launches per
iteration, microseconds per iteration:   python tools/adam_d20_probe.py"""
import os
import subprocess
import sys

SHAPES = [(5, 20, 50, 400, 22), (5, 20, 50, 800, 22), (5, 20, 64, 800, 24), (3, 24, 40, 300, 28), (3, 10, 50, 2000, 28),
          (3, 16, 40, 1200, 28), (3, 10, 50, 400, 28), (3, 10, 50, 400, 130), (3, 10, 50, 400, 256), (3, 10, 50, 400, 1024),
          (3, 10, 50, 400, 2048), (3, 10, 50, 400, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from pyvbmc_amd import VariationalPosterior, _lib, synthetic
    from pyvbmc_amd import gp as gpm
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    for cfg, D, K, N, nsk in SHAPES:
        wl = synthetic.make_workload(cfg, S=1, D=D, K=K, N=N, Ns_total=nsk * K)
        vp = VariationalPosterior(wl.D, wl.K)
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                   gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
        g.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
        bnd = synthetic.default_theta_bnd(wl)
        kw = dict(max_iter=400, use_early_stopping=False, seed=11, rng="philox")
        best = 1e9
        for r in range(4):
            t0 = time.perf_counter()
            out = minimize_adam_elbo(wl.theta.copy(), g, vp, nsk, bnd, **kw)
            best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
        print(f"D={D:2d} K={K:3d} N={N:4d} NsK={nsk:3d}: {best:6.2f} us per iteration   F {out[3][0]:.8f} -> {out[3][-1]:.8f}  "
              f"kernel {ctx.last_entmc_plan()['kernel']}", flush=True)
else:
    for tag, env in (("four launches per iteration (VBMC_ADAM_FUSED=0)", {"VBMC_ADAM_FUSED": "0"}),
                     ("one launch per batch of 20 iterations (adam_fused.hip)", {})):
        print(tag, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
