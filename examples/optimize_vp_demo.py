#!/usr/bin/env python
"""The core of PyVBMC's ``optimize_vp`` (reference vbmc/variational_optimization.py:24-330)
re-assembled from the accelerated pieces, on synthetic data:

  1. sieve   -- score a batch of perturbed starting points with the deterministic
                lower-bound objective  (``_sieve``, :775-787  ->  ``_neg_elcbo_batch``)
  2. Adam    -- stochastic optimisation of the best candidate with the Monte-Carlo entropy
                (``minimize_adam`` around ``_neg_elcbo``, :238-281  ->  ``minimize_adam_elbo``)
  3. report  -- ELBO, its variance and the per-component terms at the optimum
                (``_eval_full_elcbo``, :474-485  ->  ``_neg_elcbo(..., compute_var, separate_K)``)

    python examples/optimize_vp_demo.py [--config 2] [--candidates 256] [--iters 200]

Needs an MI355X and the built library (python -m pyvbmc_amd.build).
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from pyvbmc_amd import VariationalPosterior, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd.minimize_adam import minimize_adam_elbo  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo, _neg_elcbo_batch  # noqa: E402


def optimize(wl, n_candidates=256, n_iters=200, seed=0, verbose=True, reference_counts=True):
    """``reference_counts``: the sample counts ``optimize_vp`` uses -- ns_ent = 100 K^(2/3) in total for the
    optimiser, ns_ent_fine = 2^12 K for the report (option_configs/advanced_vbmc_options.ini:43-45,
    variational_optimization.py:467,728) -- instead of the workload's own NsK for both."""
    rng = np.random.default_rng(seed)
    K = wl.K
    ns_opt = int(np.ceil(100.0 * K ** (2.0 / 3.0) / K)) if reference_counts else wl.NsK
    ns_fine = 2**12 if reference_counts else wl.NsK
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1).copy(), wl.lambd.reshape(-1, 1).copy()
    vp.w, vp.eta = wl.w.reshape(1, -1).copy(), wl.eta.reshape(1, -1).copy()
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    theta0 = vp.get_parameters()
    cands = theta0[None, :] + 0.3 * rng.standard_normal((n_candidates, theta0.size))
    cands[0] = theta0
    _neg_elcbo_batch(cands, gp, vp, bnd)  # first device use: context, GP upload (L^-1), scratch buffers of this batch size
    t0 = time.perf_counter()
    F_sieve = _neg_elcbo_batch(cands, gp, vp, bnd)
    best = int(np.argmin(F_sieve))
    minimize_adam_elbo(cands[best], gp, vp, ns_opt, bnd, max_iter=20, seed=seed + 1)  # first use: kernel load, buffers
    t1 = time.perf_counter()
    x, y, x_tab, y_tab, iters = minimize_adam_elbo(cands[best], gp, vp, ns_opt, bnd, max_iter=n_iters,
                                                   tol_fun=0.01, seed=seed + 1)
    t2 = time.perf_counter()
    r = _neg_elcbo(x.copy(), gp, vp, 0.0, ns_fine, False, True, bnd, 0.0, True, rng="philox", seed=seed + 2)
    t3 = time.perf_counter()
    out = dict(theta=x, F_sieve=F_sieve, best=best, y_tab=y_tab, iters=iters, F=r[0], G=r[2], H=r[3],
               varF=np.ravel(r[4])[0], I_sk=r[9], J_sjk=r[10], seconds=(t1 - t0, t2 - t1, t3 - t2))
    if verbose:
        print(f"sieve : {n_candidates} candidates in {1e3 * (t1 - t0):.2f} ms; best #{best} F={F_sieve[best]:.4f} "
              f"(start point {F_sieve[0]:.4f})")
        print(f"adam  : {iters} iterations at {ns_opt} samples per component in {1e3 * (t2 - t1):.2f} ms "
              f"({1e6 * (t2 - t1) / iters:.1f} us each); "
              f"objective {y_tab[0]:.4f} -> {y_tab[-20:].mean():.4f}")
        print(f"report: ELBO={-r[0]:.5f}  (G={r[2]:.5f}, H={r[3]:.5f}), sd={np.sqrt(out['varF']):.2e}, "
              f"{1e3 * (t3 - t2):.2f} ms")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--candidates", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--workload-counts", action="store_true", help="the workload's NsK instead of the reference's ns_ent / ns_ent_fine")
    a = ap.parse_args()
    optimize(synthetic.make_workload(a.config), a.candidates, a.iters, reference_counts=not a.workload_counts)
