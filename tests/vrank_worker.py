"""Virtual ranks on ONE GPU: what rank r of a W-rank sharded job computes, for every r.

BASELINE configs 4 and 5 are sharded jobs: rank r evaluates rows
[h*r/W, h*(r+1)/W) of every component's antithetic half, with Philox draws keyed by the GLOBAL
row index, and the ranks' raw entropy accumulators are summed by one all-reduce.  The C ABI takes
the slice explicitly (``row_begin/row_count`` of ``vbmc_entmc`` / ``vbmc_elbo_opts``), so every
rank's share can be computed on one device and added up here -- the code a rank > 0 runs
(row_begin > 0, rows < n_half: the generator's component-boundary walk, the kernels' row keys)
without an 8-GPU node.

Used in-process by tests/test_virtual_ranks.py and as a script (``python vrank_worker.py out.npz
cfg W seed what``) with ``VBMC_FORCE_COLLECTIVE=1``: a 1-rank RCCL communicator then puts the
device-raw -> ncclAllReduce -> publish branch of the fused step and of the optimiser loop in the
path (a 1-rank sum is the identity, so the results must be bit-identical).
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# BASELINE configs at JOB size (Ns of the whole job on one device)
JOB_NS = {2: 100_000, 3: 1_000_000, 4: 8_000_000, 5: 4_000_000}


def shards(h, W):
    return [(h * r // W, h * (r + 1) // W - h * r // W) for r in range(W)]


def problem(cfg, ctx):
    from pyvbmc_amd import VariationalPosterior, synthetic
    from pyvbmc_amd import gp as gpm

    wl = synthetic.make_workload(cfg, Ns_total=JOB_NS[cfg])

    def mk():
        vp = VariationalPosterior(wl.D, wl.K)
        vp.ctx = ctx
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        return vp

    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
                gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
    gp.ctx = ctx
    gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
    return wl, mk, gp, synthetic.default_theta_bnd(wl)


def run_entmc(ctx, cfg, W, seed, inline):
    """Raw accumulators of the stand-alone entropy: the un-sharded launch and every virtual rank's."""
    from pyvbmc_amd import entmc_vbmc

    wl, mk, _, _ = problem(cfg, ctx)
    h = wl.NsK // 2
    ctx.set_option("elbo_pregen", 0 if inline else 1)
    try:
        H, dH, raw = entmc_vbmc(mk(), wl.NsK, (True,) * 4, True, rng="philox", seed=seed, return_raw=True)
        plan = ctx.last_entmc_plan()
        parts, plans = [], []
        for (r0, n) in shards(h, W):
            parts.append(entmc_vbmc(mk(), wl.NsK, (True,) * 4, True, rng="philox", seed=seed, rows=(r0, n),
                                    return_raw=True)[2])
            plans.append(ctx.last_entmc_plan())
    finally:
        ctx.set_option("elbo_pregen", 1)
    return dict(H=H, dH=dH, raw=raw, parts=np.array(parts), plan=plan, plans=plans)


def run_elbo(ctx, cfg, W, seed):
    """The fused step (`vbmc_neg_elcbo`, value + gradient, bounds): un-sharded, then every virtual
    rank's slice -- each evaluated three times on consecutive seeds so that the second and third
    are armed evaluations reading draws generated ahead by the previous one's finish launch with
    row_begin > 0 -- and once more from cold on the last seed."""
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    wl, mk, gp, bnd = problem(cfg, ctx)
    D, K, h = wl.D, wl.K, wl.NsK // 2
    out = {}
    F, dF, G, H, _ = _neg_elcbo(wl.theta.copy(), gp, mk(), 0.0, wl.NsK, True, False, bnd, rng="philox", seed=seed + 2)
    out.update(F=F, dF=dF, G=G, H=H, raw=ctx.last_elbo_raw(D, K), plan=ctx.last_entmc_plan())
    parts, Hs, Fs, cold = [], [], [], []
    for (r0, n) in shards(h, W):
        vp = mk()
        for s in (seed, seed + 1, seed + 2):  # consecutive seeds: armed + ahead draws on this slice
            Fr, dFr, Gr, Hr, _ = _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, wl.NsK, True, False, bnd, rng="philox",
                                            seed=s, rows=(r0, n))
        parts.append(ctx.last_elbo_raw(D, K))
        Hs.append(Hr)
        Fs.append(Fr)
        ctx.synchronize()  # cancels what is armed; the next call plans and generates afresh
        ctx.set_option("elbo_ahead", 0)
        try:
            _neg_elcbo(wl.theta.copy(), gp, mk(), 0.0, wl.NsK, True, False, bnd, rng="philox", seed=seed + 2, rows=(r0, n))
            cold.append(ctx.last_elbo_raw(D, K))
        finally:
            ctx.set_option("elbo_ahead", 1)
    out.update(parts=np.array(parts), H_parts=np.array(Hs), F_parts=np.array(Fs), cold=np.array(cold))
    return out


def run_adam(ctx, cfg, W, seed, iters=6, ns_total=None):
    """The device-resident optimiser loop with a zero step size: every iteration evaluates the same
    mixture on the draws of seed + i, which come from the spare workgroups of the loop's short
    launches (GenSlice with this slice's row_begin / rows).  H_tab of the slices must add up."""
    from pyvbmc_amd import synthetic
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    wl, mk, gp, bnd = problem(cfg, ctx)
    nsk = wl.NsK if ns_total is None else synthetic.ns_per_component(ns_total, wl.K)
    h = nsk // 2
    kw = dict(max_iter=iters, master_min=0.0, master_max=0.0, use_early_stopping=False, seed=seed, rng="philox",
              return_parts=True)
    full = minimize_adam_elbo(wl.theta.copy(), gp, mk(), nsk, bnd, **kw)
    Hs = [minimize_adam_elbo(wl.theta.copy(), gp, mk(), nsk, bnd, rows=(r0, n), **kw)[6] for (r0, n) in shards(h, W)]
    return dict(H_full=full[6], G_full=full[5], y_full=full[3], H_parts=np.array(Hs), nsk=nsk)


def main():
    from pyvbmc_amd import _lib

    out, cfg, W, seed, what = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    ctx = _lib.Context(0)
    _lib.set_default_context(ctx)
    ctx.comm_init(_lib.comm_unique_id(), 0, 1)  # with VBMC_FORCE_COLLECTIVE=1: the multi-rank branches
    ctx.comm_barrier()
    if what == "elbo":
        r = run_elbo(ctx, cfg, W, seed)
        r.pop("plan")
    else:
        r = run_adam(ctx, cfg, W, seed, ns_total=int(sys.argv[6]) if len(sys.argv) > 6 else None)
    np.savez(out, **r)
    ctx.close()


if __name__ == "__main__":
    main()
