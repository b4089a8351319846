"""Soaks of the round's late additions (not part of the suite):  python tools/soak_r5b.py [seconds per part]
  1. gp.predict with the finish in the product's epilogue: repeated calls at changing batch sizes, every result compared
     bit for bit with the three-launch form's;
  2. the fused optimiser loop with the rows split over R workgroups per component, and its D = 20 build: repeated runs
     bit-identical."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd.minimize_adam import minimize_adam_elbo  # noqa: E402

T = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = _lib.Context(0)
_lib.set_default_context(ctx)


def mk(cfg, **kw):
    wl = synthetic.make_workload(cfg, **kw)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
               gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    g.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    return wl, vp, g


# ---- 1. predict
T1 = 0.0 if len(sys.argv) > 2 else T
wl, vp, g = mk(3, S=1)
rng = np.random.default_rng(0)
Ms = [8192, 100, 4097, 64, 8000, 1000, 33, 2048]
xs = {M: rng.standard_normal((M, wl.D)) for M in Ms}
ctx.set_option("predict_fused", 0)
want = {M: g.predict(xs[M], separate_samples=True) for M in Ms}
ctx.set_option("predict_fused", 2)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < T1:
    M = Ms[n % len(Ms)]
    mu, s2 = g.predict(xs[M], separate_samples=True)
    bad += not (np.array_equal(mu, want[M][0]) and np.array_equal(s2, want[M][1]))
    n += 1
ctx.set_option("predict_fused", 1)
print(f"predict, finish in the epilogue: {n} calls over {len(Ms)} batch sizes, {bad} differ from the three-launch form", flush=True)

# ---- 2. fused loop
CASES = (("D=10 K=50 NsK=320 (four slices)", dict(S=1), 320), ("D=20 K=50 N=800 NsK=22", dict(S=1, D=20, K=50, N=800), 22),
         ("D=6 K=64 S=2 NsK=302 (three slices)", dict(S=2, D=6, K=64, N=100), 302),
         ("D=16 K=40 N=1200 NsK=28 (X^T from memory)", dict(S=1, D=16, K=40, N=1200), 28),
         ("D=24 K=40 N=300 NsK=28", dict(S=1, D=24, K=40, N=300), 28), ("D=20 K=50 N=200 NsK=22 (X^T in LDS)", dict(S=1, D=20, K=50, N=200), 22))
only = [int(v) for v in sys.argv[2:]]
for ci, (tag, kw, nsk) in enumerate(CASES):
    if only and ci not in only:
        continue
    cfg = 5 if kw.get("D") == 20 else 3
    wl, vp, g = mk(cfg, Ns_total=nsk * kw.get("K", 50), **kw)
    bnd = synthetic.default_theta_bnd(wl)
    ref = None
    t0, n, bad, not_fused = time.time(), 0, 0, 0
    while time.time() - t0 < T / 2:
        t1 = time.time()
        out = minimize_adam_elbo(wl.theta.copy(), g, vp, nsk, bnd, max_iter=200, use_early_stopping=False, seed=5, rng="philox")
        if ctx.last_entmc_plan()["kernel"] != "adam_fused":
            not_fused += 1
            print(f"  run {n}: not the fused kernel ({ctx.last_entmc_plan()['kernel']}), {1e3 * (time.time() - t1):.1f} ms", flush=True)
        if ref is None:
            ref = (out[0].copy(), np.array(out[3]).copy())
        bad += not (np.array_equal(out[0], ref[0]) and np.array_equal(np.array(out[3]), ref[1]))
        n += 1
    print(f"fused loop {tag}: {n} runs of 200 iterations, {bad} differ from the first, {not_fused} gave the one-launch form up", flush=True)
