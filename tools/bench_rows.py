#!/usr/bin/env python
"""Secondary measurements: every SURVEY section-8 row other than the headline objective,
HIP path vs the oracle (CPU, NumPy port) on the same inputs, at BASELINE config 3 shapes.

    python tools/bench_rows.py [--config 3] > profiles/rNN_rows.json

One JSON object per row: device time (median of repeats, wall clock around the call incl.
host<->device copies of the boundary), CPU time of the oracle on a bounded sample (scaled
linearly, stated), max parity error observed on the sample.
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import elbo_ref, entropy_ref, gp_ref, mixture_ref  # noqa: E402
from pyvbmc_amd import VariationalPosterior, _lib, entlb_vbmc, entmc_vbmc, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd.variational_optimization import _gp_log_joint, _neg_elcbo, _neg_elcbo_batch  # noqa: E402


def med(fn, reps=7, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


def once(fn):
    t0 = time.perf_counter()
    out = fn()
    return time.perf_counter() - t0, out


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    a = ap.parse_args()
    ctx = _lib.Context(0)

    def kernel_ms(fn, which):
        """One extra call with the HIP event pair around the dominant kernel switched on (the rows'
        wall times are taken with it off: each record costs ~6 us)."""
        ctx.set_timing(True)
        try:
            fn()
            return ctx.last_kernel_ms(which)
        finally:
            ctx.set_timing(False)

    _lib.set_default_context(ctx)
    wl1 = synthetic.make_workload(a.config, S=1)
    wl8 = synthetic.make_workload(a.config, S=8)
    D, K, N = wl1.D, wl1.K, wl1.N

    def mkvp(wl):
        vp = VariationalPosterior(wl.D, wl.K)
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        return vp

    def mkgp(wl):
        g = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                   gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
        g.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
        return g

    mix = mixture_ref.Mixture.make(wl1.mu, wl1.sigma, wl1.lambd, wl1.w, wl1.eta)
    rows = []
    info = ctx.device_info()

    def emit(name, ref, shape, t_dev, t_cpu, cpu_note, err, extra=None):
        r = {"row": name, "reference": ref, "shape": shape, "device_ms": 1e3 * t_dev,
             "cpu_oracle_ms": 1e3 * t_cpu, "cpu_note": cpu_note, "speedup": t_cpu / t_dev,
             "parity_max_rel_err": err, "device": info["name"].strip()}
        if extra:
            r.update(extra)
        rows.append(r)
        print(json.dumps(r), flush=True)

    rng = np.random.default_rng(0)
    # ---- a3: mixture log-pdf --------------------------------------------------
    vp = mkvp(wl1)
    for n in (8192, 1_000_000):
        comp = rng.integers(0, K, size=n)
        x = wl1.mu.T[comp] + wl1.lambd * wl1.sigma[comp, None] * rng.standard_normal((n, D))
        t, y = med(lambda: vp.pdf(x, orig_flag=False, log_flag=True), reps=5)
        ns = min(n, 20000)
        tc, yo = once(lambda: mixture_ref.pdf(mix, x[:ns], log_flag=True))
        emit("a3 vp.pdf(log)", "variational_posterior.py:365-564", f"n={n} D={D} K={K}", t, tc * n / ns,
             f"oracle on {ns} points, scaled", rel(y[:ns], yo),
             {"kernel_ms": kernel_ms(lambda: vp.pdf(x, orig_flag=False, log_flag=True), 2),
              "algorithmic_bytes": 8.0 * n * (D + 1)})
    t, (y, dy) = med(lambda: vp.pdf(x[:8192], orig_flag=False, log_flag=True, grad_flag=True), reps=5)
    tc, (yo, dyo) = once(lambda: mixture_ref.pdf(mix, x[:8192], log_flag=True, grad_flag=True))
    emit("a3 vp.pdf(log,grad)", "variational_posterior.py:464-469,532", f"n=8192 D={D} K={K}", t, tc, "full", rel(dy, dyo))
    # ---- a7: entlb -------------------------------------------------------------
    t, (H, dH) = med(lambda: entlb_vbmc(mkvp(wl1)))
    tc, (Ho, dHo) = once(lambda: entropy_ref.entlb(mix))
    emit("a7 entlb_vbmc", "entropy/entlb_vbmc.py:6-180", f"D={D} K={K}", t, tc, "full", max(rel(H, Ho), rel(dH, dHo)))
    # ---- a6: entmc value-only / value+grad (Philox draws) ------------------------
    for gf, tag in (((False,) * 4, "value"), ((True,) * 4, "value+grad")):
        vpp = mkvp(wl1)
        t, _ = med(lambda: entmc_vbmc(vpp, wl1.NsK, gf, True, rng="philox", seed=5), reps=9)
        nsk = 2000
        eps = synthetic.draw_eps_half(K, D, nsk, 1)
        tc, (Ho, dHo) = once(lambda: entropy_ref.entmc(mix, nsk, gf, True, eps_half=eps))
        Hd, dHd = entmc_vbmc(mkvp(wl1), nsk, gf, True, eps_half=eps)
        err = rel(Hd, Ho) if not dHo.size else max(rel(Hd, Ho), rel(dHd, dHo))
        emit(f"a6 entmc_vbmc {tag}", "entropy/entmc_vbmc.py:6-134", f"D={D} K={K} NsK={wl1.NsK}", t,
             tc * wl1.NsK / nsk, f"oracle at NsK={nsk}, scaled", err,
             {"kernel_ms": kernel_ms(lambda: entmc_vbmc(vpp, wl1.NsK, gf, True, rng="philox", seed=5), 0)})
        tc_entmc_grad = tc * wl1.NsK / nsk  # the gradient pass (last) stands in for one Adam objective
    # ---- a8: _gp_log_joint -------------------------------------------------------
    for wl, tag in ((wl1, "S=1"), (wl8, "S=8")):
        g, ogp = mkgp(wl), gp_ref.make_gp(wl.X, wl.y, wl.hyp, s2=wl.s2, noise_user=wl.s2 is not None)
        v = mkvp(wl)
        t, r = med(lambda: _gp_log_joint(v, g, True, True, True, False, False))
        tc, ro = once(lambda: gp_ref.gp_log_joint(mix, ogp, True, True, True, False, False))
        emit(f"a8 _gp_log_joint grad {tag}", "vbmc/variational_optimization.py:1238-1606", f"N={N} K={K} {tag}",
             t, tc, "full", max(rel(r[0], ro[0]), rel(r[1], ro[1])))
        t, r = med(lambda: _gp_log_joint(v, g, False, True, True, True, True), reps=5)
        tc, ro = once(lambda: gp_ref.gp_log_joint(mix, ogp, False, True, True, True, True))
        emit(f"a8 _gp_log_joint var+separate_K {tag}", "vbmc/variational_optimization.py:1473-1514",
             f"N={N} K={K} {tag}", t, tc, "full", max(rel(np.ravel(r[2]), np.ravel(ro[2])), rel(r[6], ro[6])))
    # ---- a12: gp.predict -----------------------------------------------------------
    for wl, tag in ((wl1, "S=1"), (wl8, "S=8")):
        g, ogp = mkgp(wl), gp_ref.make_gp(wl.X, wl.y, wl.hyp, s2=wl.s2, noise_user=wl.s2 is not None)
        M = 8192
        xs = rng.standard_normal((M, D))
        t, (fmu, fs2) = med(lambda: g.predict(xs, separate_samples=True), reps=5)
        ms = 1024
        tc, (omu, os2) = once(lambda: gp_ref.predict(ogp, xs[:ms], separate_samples=True))
        sf2 = float(np.exp(2 * wl.hyp[0, D]))
        err = max(np.max(np.abs(fmu[:ms] - omu)) / max(1.0, np.max(np.abs(omu))), np.max(np.abs(fs2[:ms] - os2)) / max(1.0, sf2))
        emit(f"a12 gp.predict {tag}", "gpyreg GP.predict (third party; SURVEY App. A)", f"M={M} N={N} D={D} {tag}",
             t, tc * M / ms, f"oracle on {ms} points, scaled", float(err),
             {"kernel_ms_first_sample": kernel_ms(lambda: g.predict(xs, separate_samples=True), 3),
              "gemm_flops": 1.0 * M * N * N})
    # ---- a9 secondary: _eval_full_elcbo call (value, variance, separate_K) -------------
    g, ogp = mkgp(wl1), gp_ref.make_gp(wl1.X, wl1.y, wl1.hyp)
    nsk = 4096
    v = mkvp(wl1)
    t, r = med(lambda: _neg_elcbo(wl1.theta.copy(), g, v, 0.0, nsk, False, True, None, 0.0, True, rng="philox", seed=3), reps=5)
    eps = synthetic.draw_eps_half(K, D, 200, 2)
    tc, ro = once(lambda: elbo_ref.neg_elcbo(wl1.theta.copy(), ogp, mix.copy(), 0.0, 200, False, True, None, True, eps_half=eps))
    rd = _neg_elcbo(wl1.theta.copy(), g, mkvp(wl1), 0.0, 200, False, True, None, 0.0, True, eps_half=eps)
    emit("a9 _neg_elcbo(value,var,separate_K)", "vbmc/variational_optimization.py:474-485", f"NsK={nsk} N={N} K={K}",
         t, tc, "oracle at NsK=200 (variance part dominates CPU time; unscaled)", max(rel(rd[0], ro[0]), rel(rd[10], ro[10])))
    # ---- 8f row 1: the sieve's batch of candidates (Ns=0, no gradient) ------------------------
    B = 50 * K  # ns_elbo candidates of a full sieve (advanced_vbmc_options.ini:63)
    bnd = synthetic.default_theta_bnd(wl1)
    thetas = wl1.theta[None, :] + 0.3 * rng.standard_normal((B, wl1.theta.size))
    v = mkvp(wl1)
    t, Fb = med(lambda: _neg_elcbo_batch(thetas, g, v, bnd), reps=5)
    nseq = 200
    def seq():
        vv = mkvp(wl1)
        return [_neg_elcbo(thetas[b].copy(), g, vv, 0.0, 0, False, False, bnd)[0] for b in range(nseq)]
    tseq, Fs = med(seq, reps=3, warm=1)
    nb = 20
    tc, Fo = once(lambda: [elbo_ref.neg_elcbo(thetas[b].copy(), ogp, mix.copy(), 0.0, 0, False, False, bnd, False)[0] for b in range(nb)])
    tc_sieve_per_cand = tc / nb
    emit("8f-1 sieve batch: _neg_elcbo x B (Ns=0, no grad)", "vbmc/variational_optimization.py:775-787", f"B={B} K={K} N={N}",
         t, tc * B / nb, f"oracle on {nb} candidates, scaled", rel(Fb[:nb], Fo),
         {"per_candidate_device_calls_ms": 1e3 * tseq * B / nseq, "batch_vs_sequential_device": tseq * B / nseq / t})
    # ---- 8f row 2: the stochastic optimiser's loop, device resident --------------------------
    from pyvbmc_amd.minimize_adam import minimize_adam, minimize_adam_elbo
    n_it = 200
    kw = dict(max_iter=n_it, master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=False)
    v = mkvp(wl1)
    t, out = med(lambda: minimize_adam_elbo(wl1.theta.copy(), g, v, wl1.NsK, bnd, seed=11, rng="philox", **kw), reps=3, warm=1)
    def host_loop():
        vv, it = mkvp(wl1), [0]
        def f(th):
            r = _neg_elcbo(th, g, vv, 0.0, wl1.NsK, True, False, bnd, rng="philox", seed=11 + it[0])
            it[0] += 1
            return r[0], r[1]
        return minimize_adam(f, wl1.theta.copy(), **kw)
    th, oh = med(host_loop, reps=3, warm=1)
    emit("8f-2 minimize_adam, device-resident loop (per iteration)", "vbmc/minimize_adam.py:84-105 + variational_optimization.py:238-249",
         f"NsK={wl1.NsK} K={K} N={N} iters={n_it}", t / n_it, tc_entmc_grad, "oracle entropy value+grad of one evaluation, scaled (the GP part and the update are negligible)",
         rel(out[3], oh[3]), {"host_loop_device_objective_ms_per_iter": 1e3 * th / n_it, "device_vs_host_loop": th / t})
    # the same loop at the sample count optimize_vp really uses (ns_ent = 100 K^(2/3) in total, advanced_vbmc_options.ini:43:
    # NsK = 28 at K = 50): one launch per batch of iterations (csrc/adam_fused.hip) against the four-launch iteration
    nsk_ref = 2 * max(1, int(round(100.0 * K ** (2.0 / 3.0) / K / 2.0)))
    kw2 = dict(max_iter=400, master_min=0.001, master_max=0.1, master_decay=200, use_early_stopping=False)
    per = {}
    for fused in (0, 1):
        ctx.set_option("adam_fused", fused)
        vv = mkvp(wl1)
        tt, oo = med(lambda: minimize_adam_elbo(wl1.theta.copy(), g, vv, nsk_ref, bnd, seed=11, rng="philox", **kw2), reps=5, warm=1)
        per[fused] = (tt / 400, oo, ctx.last_entmc_plan()["kernel"])
    ctx.set_option("adam_fused", 1)
    emit("8f-2b minimize_adam at the reference's ns_ent (per iteration)", "vbmc/minimize_adam.py:84-137; option_configs/advanced_vbmc_options.ini:43",
         f"NsK={nsk_ref} K={K} N={N} iters=400", per[1][0], tc_entmc_grad * nsk_ref / wl1.NsK,
         "oracle entropy value+grad of one evaluation, scaled to this NsK", rel(per[1][1][3], per[0][1][3]),
         {"four_launch_iteration_ms": 1e3 * per[0][0], "fused_vs_four_launches": per[0][0] / per[1][0], "kernels": [per[0][2], per[1][2]]})
    # the same loop at BASELINE config 5's dimension and training-set size (D = 20, N = 800) with the reference's ns_ent: K = 50
    # and K = 64 run as one launch per batch since round 5 (csrc/adam_fused.hip: builds for D <= 24, X^T read from memory);
    # config 5's own K = 100 is beyond the fused kernel (lane = component) and keeps four launches -- both rows are here
    for Kx in (50, 64, 100):
        nskx = 2 * max(1, int(round(100.0 * Kx ** (2.0 / 3.0) / Kx / 2.0)))
        wlx = synthetic.make_workload(5, S=1, D=20, K=Kx, N=800, Ns_total=nskx * Kx)
        gx, bndx = mkgp(wlx), synthetic.default_theta_bnd(wlx)
        perx = {}
        for fused in (0, 1):
            ctx.set_option("adam_fused", fused)
            vx = mkvp(wlx)
            tt, oo = med(lambda: minimize_adam_elbo(wlx.theta.copy(), gx, vx, nskx, bndx, seed=11, rng="philox", **kw2), reps=3, warm=1)
            perx[fused] = (tt / 400, oo, ctx.last_entmc_plan()["kernel"])
        ctx.set_option("adam_fused", 1)
        emit("8f-2c minimize_adam at the reference's ns_ent, D=20 N=800 (per iteration)",
             "vbmc/minimize_adam.py:84-137; option_configs/advanced_vbmc_options.ini:43", f"NsK={nskx} D=20 K={Kx} N=800 iters=400",
             perx[1][0], float("nan"), "not timed (see 8f-2b)", rel(perx[1][1][3], perx[0][1][3]),
             {"four_launch_iteration_ms": 1e3 * perx[0][0], "fused_vs_four_launches": perx[0][0] / perx[1][0],
              "kernels": [perx[0][2], perx[1][2]]})
    # ---- the three stages of optimize_vp together, at the counts it uses (examples/optimize_vp_demo.py) ----
    nc = 50 * K
    cands = thetas  # the sieve row's candidates
    kw3 = dict(max_iter=1000, master_min=0.001, master_max=0.1, master_decay=200, tol_fun=1e-12)
    def vp_opt():
        vv = mkvp(wl1)
        Fs = _neg_elcbo_batch(cands, g, vv, bnd)
        out = minimize_adam_elbo(cands[int(np.argmin(Fs))], g, vv, nsk_ref, bnd, seed=3, rng="philox", **kw3)
        rr = _neg_elcbo(out[0].copy(), g, vv, 0.0, 4096, False, True, bnd, 0.0, True, rng="philox", seed=4)
        return Fs, out, rr
    t3, (Fs3, out3, rr3) = med(vp_opt, reps=5, warm=1)
    emit("optimize_vp's accelerated stages: sieve of 50 K candidates + Adam at ns_ent until the reference's stopping rule ends it (at most 1000 iterations) + full ELBO at ns_ent_fine",
         "vbmc/variational_optimization.py:90-391,428-500,660-810", f"K={K} N={N} candidates={nc} NsK={nsk_ref}/4096", t3,
         tc_sieve_per_cand * nc + int(out3[4]) * tc_entmc_grad * nsk_ref / wl1.NsK, "oracle: sieve scaled from 20 candidates + iterations x (entropy value+grad at ns_ent, scaled)",
         0.0, {"iterations": int(out3[4]), "F_report": float(rr3[0])})
    # ---- 8f row 3: acquisition evaluation on the cached search batch (2^13 points) ------------
    from types import SimpleNamespace
    from oracle import acq_ref
    from pyvbmc_amd import acquisition
    M = 8192
    g8, ogp8 = mkgp(wl8), gp_ref.make_gp(wl8.X, wl8.y, wl8.hyp, s2=wl8.s2, noise_user=wl8.s2 is not None)
    comp = rng.integers(0, K, size=M)
    Xs = wl8.mu.T[comp] + 1.5 * wl8.lambd * wl8.sigma[comp, None] * rng.standard_normal((M, D))
    length = np.exp(wl8.hyp[0, :D])
    state = dict(integer_vars=None, lb_eps_orig=wl8.X.min(0) - 2.0, ub_eps_orig=wl8.X.max(0) + 2.0,
                 gp_length_scale=length, variance_regularized_acq_fcn=True, tol_gp_var=1e-4)
    flog = SimpleNamespace(y_max=float(np.max(wl8.y)))
    v8 = mkvp(wl8)
    fn = acquisition.AcqFcnLog()
    t, av = med(lambda: fn(Xs.copy(), g8, v8, flog, state), reps=5)
    def composed():
        fm, fs = g8.predict(Xs, separate_samples=True)
        lp = v8.pdf(Xs, orig_flag=False, log_flag=True)
        return fm, fs, lp
    tcomp, _ = med(composed, reps=5)
    ms = 512
    with np.errstate(all="ignore"):
        tc, ao = once(lambda: acq_ref.acq_call(acq_ref.LOG, Xs[:ms].copy(), ogp8, mix, flog.y_max, state))
    fin = ~np.isinf(ao)
    emit("8f-3 AcqFcnLog.__call__ (predict S=8 + log pdf + formula)", "acquisition_functions/abstract_acq_fcn.py:68-147",
         f"M={M} N={N} D={D} K={K} S=8", t, tc * M / ms, f"oracle on {ms} points, scaled",
         float(np.max(np.abs(av[:ms][fin] - ao[fin]) / np.maximum(1.0, np.abs(ao[fin])))),
         {"separate_predict_plus_pdf_calls_ms": 1e3 * tcomp})
    # ---- 8f row 4: vp.sample and kl_div ---------------------------------------------------------
    from oracle import sample_ref
    vpa = mkvp(wl1)
    vpb = mkvp(wl1)
    vpb.mu = vpb.mu + 0.2 * rng.standard_normal(vpb.mu.shape)
    mix2 = mix.copy(); mix2.mu = vpb.mu.copy()
    Nsmp = 1_000_000
    t, (xs_, is_) = med(lambda: vpa.sample(Nsmp, orig_flag=False, balance_flag=True, rng="philox", seed=21, shuffle=False), reps=5)
    tc, (xo_, io_) = once(lambda: vpa.sample(Nsmp, orig_flag=False, balance_flag=True, rng="numpy"))
    nchk = 20000
    xr, ir = sample_ref.sample(mix, nchk, 21, False)
    xd, idd = vpa.sample(nchk, orig_flag=False, balance_flag=False, rng="philox", seed=21)
    emit("8f-4 vp.sample(1e6, balance) on the device generator", "variational_posterior.py:241-363", f"N={Nsmp} D={D} K={K}",
         t, tc, "the mirror's NumPy-stream path (reference arithmetic), full", float(np.max(np.abs(xd - xr)) / np.max(np.abs(xr))),
         {"labels_equal": bool(np.array_equal(idd, ir)), "note": "device time includes the 80 MB D2H of the samples"})
    Nkl = 100_000
    t, kld = med(lambda: vpa.kl_div(vpb, N=Nkl, rng="philox", seed=31), reps=5)
    tc, klo = once(lambda: sample_ref.kl_div_mc(mix, mix2, Nkl, 31))
    emit("8f-4 vp.kl_div(vp2, N=1e5), Monte-Carlo branch, one device call", "variational_posterior.py:1107-1126", f"N={Nkl} D={D} K={K}",
         t, tc, "oracle on the same draws, full", rel(kld, klo))
    ctx.close()


if __name__ == "__main__":
    main()
