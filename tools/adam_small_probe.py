"""The device-resident optimiser loop at the reference's default sample counts (ns_ent = 100 K^(2/3) total:
NsK = 28 at K = 50; advanced_vbmc_options.ini:43) and at ns_ent_fine: microseconds per iteration.
(Round 3 used it to A/B a build whose step workgroup reduced the K partial rows itself instead of a finish
launch: 53.5 against 31.9 us at K = 50 -- one workgroup's chain of L2 reads for 500 x 50 terms costs far
more than a 153-workgroup launch and its boundary; dropped, DESIGN section 7.  Then to A/B the fused loop,
csrc/adam_fused.hip.)"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from pyvbmc_amd import VariationalPosterior, _lib, synthetic
    from pyvbmc_amd import gp as gpm
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    for cfg, nsk in ((3, 28), (3, 64), (2, 28), (3, 4096)):
        wl = synthetic.make_workload(cfg, S=1)
        vp = VariationalPosterior(wl.D, wl.K)
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
        g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
        bnd = synthetic.default_theta_bnd(wl)
        kw = dict(max_iter=400, use_early_stopping=False, seed=11, rng="philox")
        best = 1e9
        for r in range(4):
            t0 = time.perf_counter()
            out = minimize_adam_elbo(wl.theta.copy(), g, vp, nsk, bnd, **kw)
            best = min(best, (time.perf_counter() - t0) / 400 * 1e6)
        print(f"config {cfg} shape, NsK={nsk:5d}: {best:6.2f} us per iteration   F {out[3][0]:.10f} -> {out[3][-1]:.10f}  "
              f"|x| {np.linalg.norm(out[0]):.12f}  plan {ctx.last_entmc_plan()}")
else:
    for tag, env in (("four launches per iteration (VBMC_ADAM_FUSED=0)", {"VBMC_ADAM_FUSED": "0"}),
                     ("one launch per batch of 20 iterations (adam_fused.hip)", {}),
                     ("the same with release / acquire flags (VBMC_ADAM_FUSED=3)", {"VBMC_ADAM_FUSED": "3"})):
        print(tag)
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
