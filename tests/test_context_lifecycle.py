"""Context plumbing on the GPU: several contexts in one process, one context fed alternating
problem shapes (every scratch buffer regrows / is reused), objects pickled without their
handles, and results that do not depend on any of it."""
import pickle
from pathlib import Path

import numpy as np
import pytest
from helpers import oracle_gp, oracle_mix

from oracle import elbo_ref, philox_ref
from pyvbmc_amd import synthetic

ROOT = Path(__file__).resolve().parent.parent

pytestmark = pytest.mark.gpu


def make(wl, ctx):
    from test_gpu_parity import make_gp, make_vp

    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta, X=wl.X,
              y=wl.y, hyp=wl.hyp, s2=np.zeros(0))
    return wd, make_vp(wd, ctx), make_gp(wd, ctx)


def objective(wl, vp, gp, seed):
    from pyvbmc_amd.variational_optimization import _neg_elcbo

    bnd = synthetic.default_theta_bnd(wl)
    return _neg_elcbo(wl.theta.copy(), gp, vp, 0.0, wl.NsK, True, False, bnd, rng="philox", seed=seed)


def oracle_objective(wl, wd, seed):
    eps = philox_ref.eps_half(wl.K, wl.NsK // 2, wl.D, seed)
    bnd = synthetic.default_theta_bnd(wl)
    return elbo_ref.neg_elcbo(wl.theta.copy(), oracle_gp(wd), oracle_mix(wd), 0.0, wl.NsK, True, False, bnd,
                              eps_half=eps)


def test_alternating_shapes_two_contexts_and_pickling():
    from pyvbmc_amd import _lib
    from pyvbmc_amd.minimize_adam import minimize_adam_elbo

    a, b = _lib.Context(0), _lib.Context(0)
    try:
        shapes = [synthetic.make_workload(2, D=3, K=5, N=40, Ns_total=5 * 60),
                  synthetic.make_workload(2, D=7, K=21, N=130, Ns_total=21 * 200),
                  synthetic.make_workload(1),
                  synthetic.make_workload(2, D=3, K=5, N=40, Ns_total=5 * 60)]
        ref = [oracle_objective(wl, make(wl, a)[0], 11 + i) for i, wl in enumerate(shapes)]
        for rep in range(2):
            for i, wl in enumerate(shapes):
                for ctx in (a, b):
                    wd, vp, gp = make(wl, ctx)
                    F = objective(wl, vp, gp, 11 + i)
                    assert abs(F[0] - ref[i][0]) <= 1e-10 * max(1.0, abs(ref[i][0])), (rep, i)
                    assert np.max(np.abs(F[1] - ref[i][1])) <= 1e-8 * max(1.0, np.max(np.abs(ref[i][1])))
        # the device-resident loop leaves the context usable for a different shape
        wd, vp, gp = make(shapes[1], a)
        out = minimize_adam_elbo(shapes[1].theta.copy(), gp, vp, shapes[1].NsK, max_iter=20, seed=3)
        assert np.all(np.isfinite(out[3]))
        wd, vp, gp = make(shapes[0], a)
        F = objective(shapes[0], vp, gp, 11)
        assert abs(F[0] - ref[0][0]) <= 1e-10 * max(1.0, abs(ref[0][0]))
        # objects travel without their device handle and pick up the default context again
        vp2, gp2 = pickle.loads(pickle.dumps((vp, gp)))
        assert vp2._ctx is None and gp2._ctx is None
        _lib.set_default_context(b)
        F2 = objective(shapes[0], vp2, gp2, 11)
        assert F2[0] == F[0] and np.array_equal(F2[1], F[1])  # same kernels, same order: same bits
    finally:
        _lib.set_default_context(None)
        a.close()
        b.close()
    with pytest.raises(Exception):
        objective(shapes[0], *make(shapes[0], a)[1:], 1)  # a closed context refuses work


@pytest.mark.gpu
def test_kernel_timing_is_opt_in():
    """The HIP event pair around the dominant kernels is off by default (each record costs a
    barrier packet between dependent kernels); vbmc_set_timing switches it on per context."""
    from pyvbmc_amd import _lib, synthetic
    from pyvbmc_amd.entropy import entmc_vbmc
    from test_gpu_parity import make_vp

    ctx = _lib.Context(0)
    wl = synthetic.make_workload(1)
    wd = dict(D=wl.D, K=wl.K, mu=wl.mu, sigma=wl.sigma, lambd=wl.lambd, w=wl.w, eta=wl.eta)
    vp = make_vp(wd, ctx)
    entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=1, ctx=ctx)
    with pytest.raises(ValueError):
        ctx.last_kernel_ms(0)  # nothing was timed
    ctx.set_timing(True)
    H1 = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=1, ctx=ctx)[0]
    assert 0.0 < ctx.last_kernel_ms(0) < 100.0
    ctx.set_timing(False)
    H2 = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=1, ctx=ctx)[0]
    assert H1 == H2
    ctx.close()


@pytest.mark.gpu
def test_predict_timing_levels():
    """vbmc_set_timing(1) brackets predict's launches with one event pair and records nothing between them; level 2 adds the
    pair around the variance product (which = 5) -- two records between dependent launches, which is why a harness reads the
    whole interval at level 1 (round 5: rounds 2-4 read it with the inner pair on, ~5 us longer).  Results do not depend on
    the level (level 2 keeps the separate finish launch so that the product is timed alone)."""
    from pyvbmc_amd import _lib
    from pyvbmc_amd import gp as gpm

    ctx = _lib.Context(0)
    wl = synthetic.make_workload(3, S=1)
    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.ctx = ctx
    gp.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    xs = np.random.default_rng(3).standard_normal((4096, wl.D))
    ref = gp.predict(xs, separate_samples=True)
    with pytest.raises(ValueError):
        ctx.last_kernel_ms(3)  # nothing was timed
    ctx.set_timing(1)
    out1 = gp.predict(xs, separate_samples=True)
    t_all = ctx.last_kernel_ms(3)
    with pytest.raises(ValueError):
        ctx.last_kernel_ms(5)  # level 1 records no pair inside
    ctx.set_timing(2)
    out2 = gp.predict(xs, separate_samples=True)
    t_var, t_all2 = ctx.last_kernel_ms(5), ctx.last_kernel_ms(3)
    # a predict at a lower level invalidates the product's record (it used to leave the level-2 interval
    # of an EARLIER call to be read as this call's)
    ctx.set_timing(1)
    gp.predict(xs, separate_samples=True)
    with pytest.raises(ValueError, match="vbmc_set_timing"):
        ctx.last_kernel_ms(5)
    ctx.set_timing(0)
    assert 0.0 < t_var < t_all2 < 1.0 and 0.0 < t_all < 1.0
    for o in (out1, out2):
        assert np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1])
    ctx.close()


@pytest.mark.gpu
def test_context_binds_the_calling_thread_to_the_devices_numa_node():
    """Round 6: vbmc_ctx_create narrows the CALLING thread's CPU affinity to the CPUs local to its device (the polled step
    is a PCIe latency chain: 85.3 us from the GPU's node, 87.8 us from the other socket).  It only removes CPUs; a set
    that is already inside the node, or wholly outside it, is left alone; VBMC_HOST_AFFINITY=0 disables it.  Run in
    child processes: affinity is per thread and sticky."""
    import json
    import os
    import subprocess
    import sys

    child = r"""
import json, os, sys
sys.path.insert(0, %r)
from pyvbmc_amd import _lib
pre = sys.argv[1]
if pre != "all":
    os.sched_setaffinity(0, {int(c) for c in pre.split(",")})
else:  # (the parent -- pytest, after its first context -- may itself have been narrowed: children inherit that)
    try:
        os.sched_setaffinity(0, set(range(os.cpu_count())))
    except OSError:
        pass
before = sorted(os.sched_getaffinity(0))
ctx = _lib.Context(0)
bound, n = ctx.host_affinity()
after = sorted(os.sched_getaffinity(0))
ctx.close()
print(json.dumps({"before": before, "after": after, "bound": bound, "n": n}))
""" % str(ROOT)

    def run(pre, env=None):
        e = dict(os.environ)
        e.pop("VBMC_HOST_AFFINITY", None)
        e.update(env or {})
        out = subprocess.run([sys.executable, "-c", child, pre], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    free = run("all")
    assert set(free["after"]) <= set(free["before"])  # never adds CPUs
    if not free["bound"]:
        # (a single-node host, or a launcher that already confined us to the device's node: nothing to narrow)
        assert free["after"] == free["before"]
        pytest.skip("the calling thread is already inside the device's NUMA node")
    local = free["after"]
    assert 0 < len(local) < len(free["before"]) and free["n"] == len(local)
    # switched off
    off = run("all", {"VBMC_HOST_AFFINITY": "0"})
    assert not off["bound"] and off["after"] == off["before"]
    # a set already inside the node: untouched
    inside = run(",".join(str(c) for c in local[:4]))
    assert not inside["bound"] and inside["after"] == inside["before"]
    # a set wholly on the other node: the user's choice, untouched
    other = [c for c in free["before"] if c not in set(local)]
    far = run(",".join(str(c) for c in other[:4]))
    assert not far["bound"] and far["after"] == far["before"]
    # a set that straddles the nodes: narrowed to its local part
    mixed = run(",".join(str(c) for c in local[:3] + other[:3]))
    assert mixed["bound"] and mixed["after"] == sorted(local[:3])
