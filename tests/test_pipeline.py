"""The optimize_vp core (sieve -> Adam -> full-ELBO report) assembled from the accelerated
pieces (examples/optimize_vp_demo.py) against the same pipeline assembled from the oracle,
on identical candidates and identical Philox draws."""
import sys
from pathlib import Path

import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix, rel_err

from oracle import adam_ref, elbo_ref, philox_ref
from pyvbmc_amd import synthetic

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "examples"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("reference_counts", [False, True], ids=["workload-NsK", "ns_ent-and-ns_ent_fine"])
def test_pipeline_vs_oracle(reference_counts):
    """reference_counts: the optimiser at ns_ent = 100 K^(2/3) samples in total (the one-launch loop, csrc/adam_fused.hip)
    and the report at ns_ent_fine = 2^12 per component, as optimize_vp runs them."""
    from optimize_vp_demo import optimize

    from pyvbmc_amd import _lib

    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    try:
        wl = synthetic.make_workload(2, Ns_total=20 * 100)
        n_cand, n_it, seed = 24, 45, 3
        got = optimize(wl, n_cand, n_it, seed=seed, verbose=False, reference_counts=reference_counts)
        ns_opt = 2 * int(np.ceil(np.ceil(100.0 * wl.K ** (2.0 / 3.0) / wl.K) / 2.0)) if reference_counts else wl.NsK
        ns_fine = 2**12 if reference_counts else wl.NsK
        assert (ctx.last_entmc_plan()["kernel"] is not None)
        # ---- the oracle's run of the same three stages ----
        wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
                  y=wl.y, hyp=wl.hyp, s2=np.zeros(0))
        mix, gp = oracle_mix(wd), oracle_gp(wd)
        bnd = synthetic.default_theta_bnd(wl)
        from oracle import mixture_ref

        theta0 = mixture_ref.get_parameters(mix)
        rng = np.random.default_rng(seed)
        cands = theta0[None, :] + 0.3 * rng.standard_normal((n_cand, theta0.size))
        cands[0] = theta0
        Fs = np.array([elbo_ref.neg_elcbo(c.copy(), gp, mix.copy(), 0.0, 0, False, False, bnd)[0] for c in cands])
        assert rel_err(got["F_sieve"], Fs) < 1e-10 and got["best"] == int(np.argmin(Fs))
        it = [0]

        def f(t):
            eps = philox_ref.eps_half(wl.K, ns_opt // 2, wl.D, seed + 1 + it[0])
            it[0] += 1
            r = elbo_ref.neg_elcbo(t, gp, mix, 0.0, ns_opt, True, False, bnd, eps_half=eps)
            return r[0], r[1]

        x, y, xt, yt, iters = adam_ref.minimize_adam(f, cands[got["best"]].copy(), tol_fun=0.01, max_iter=n_it)
        assert got["iters"] == iters
        assert rel_err(got["y_tab"], yt) < 1e-7 and rel_err(got["theta"], x) < 1e-7
        eps = philox_ref.eps_half(wl.K, ns_fine // 2, wl.D, seed + 2)
        r = elbo_ref.neg_elcbo(x.copy(), gp, mix, 0.0, ns_fine, False, True, bnd, True, eps_half=eps)
        assert abs(got["F"] - r[0]) <= 1e-7 * max(1.0, abs(r[0]))
        assert abs(got["varF"] - np.ravel(r[4])[0]) <= 1e-6 * max(1e-12, abs(np.ravel(r[4])[0]))
        assert rel_err(got["I_sk"], r[9]) < 1e-7
        assert got["y_tab"][-1] < got["y_tab"][0]  # and the optimiser did its job
    finally:
        _lib.set_default_context(None)
        ctx.close()
