"""``pyvbmc_amd.patch(vo)`` and shapes the kernels do not cover (D > 32, ...).

The reference's loops take any D (/root/reference/pyvbmc/entropy/entmc_vbmc.py:64-112); the device path
answers VBMC_E_UNSUP (``_lib.UnsupportedShape``).  Under the drop-in such a call must go back to the REFERENCE
callable ``patch`` replaced -- never to an error, never to ``oracle/``.  CPU test: a stand-in module whose
"reference" callables record their calls, and mirrors that raise as the library would.
"""
import types

import numpy as np
import pytest

import pyvbmc_amd
from pyvbmc_amd import _lib, dropin
from pyvbmc_amd import entropy as aent
from pyvbmc_amd import minimize_adam as aadam
from pyvbmc_amd import variational_optimization as avo


def make_module(log):
    vo = types.ModuleType("standin_variational_optimization")

    def ref(name, ret):
        def f(*a, **kw):
            log.append((name, a, dict(kw)))
            return ret
        f.__name__ = name
        return f

    vo.entmc_vbmc = ref("entmc_vbmc", (1.0, np.zeros(3)))
    vo.entlb_vbmc = ref("entlb_vbmc", (2.0, np.zeros(3)))
    vo._gp_log_joint = ref("_gp_log_joint", (3.0, None, None, None, 0))
    vo._neg_elcbo = ref("_neg_elcbo", (4.0, np.ones(3), 3.0, 1.0, 0))

    def _sieve(options, optim_state, vp, gp, init_N=None, best_N=1, K=None):
        vps = np.empty(3, dtype=object)
        for i in range(3):
            v = types.SimpleNamespace(optimize_mu=True, optimize_sigma=True, optimize_lambd=True, optimize_weights=True,
                                      K=2, i=i, set_parameters=lambda th: None, get_parameters=lambda: np.zeros(3))
            vps[i] = v
        F = np.array([vo._neg_elcbo(np.full(3, float(i)), gp, vps[i], 0, 0, 0, False, None)[0] for i in range(3)])
        order = np.argsort(F)
        return vps[order], np.arange(3)[order]

    vo._sieve = _sieve

    def minimize_adam(f, x0, *a, **kw):
        log.append(("ref_minimize_adam", (), {}))
        return x0, 0.0, None, None, 0

    vo.minimize_adam = minimize_adam
    return vo


def raising(*a, **kw):
    raise _lib.UnsupportedShape("entmc: D=40 > 32 not supported")


def test_unsupported_shapes_go_back_to_the_reference_callables(monkeypatch):
    log = []
    vo = make_module(log)
    ref_neg = vo._neg_elcbo
    for mod, name in ((aent, "entmc_vbmc"), (aent, "entlb_vbmc"), (avo, "_gp_log_joint"), (avo, "_neg_elcbo"),
                      (avo, "_neg_elcbo_batch"), (aadam, "minimize_adam_elbo")):
        monkeypatch.setattr(mod, name, raising)
    pyvbmc_amd.patch(vo)
    try:
        theta = np.arange(3.0)
        # the four leaves: the reference's return values, its own arguments, none of the mirrors' keyword-only extras
        assert vo.entmc_vbmc("vp", 10, rng="philox", seed=3)[0] == 1.0
        assert vo.entlb_vbmc("vp")[0] == 2.0
        assert vo._gp_log_joint("vp", "gp", (True,) * 4)[0] == 3.0
        out = vo._neg_elcbo(theta, "gp", "vp", 0.0, 10, True, False, None, rng="philox", ctx=None)
        assert out[0] == 4.0
        assert [c[0] for c in log] == ["entmc_vbmc", "entlb_vbmc", "_gp_log_joint", "_neg_elcbo"]
        assert log[0][1] == ("vp", 10) and log[0][2] == {}
        assert log[3][1][1:] == ("gp", "vp", 0.0, 10, True, False, None) and log[3][2] == {}
        # the sieve: the batched call refuses the shape, every candidate goes through the per-call path to the reference
        log.clear()
        vps, _ = vo._sieve({}, {}, "vp", "gp", init_N=3)
        assert len(vps) == 3 and [c[0] for c in log] == ["_neg_elcbo"] * 3
        # the stochastic optimiser: the device loop refuses, the host loop runs around the caller's objective
        log.clear()

        def closure_factory():
            gp, vp0, elcbo_beta, ns_ent_K, compute_var, theta_bnd = "gp", "vp", 0.0, 10, False, None

            def vb_train_mc_fun(theta_):
                r = vo._neg_elcbo(theta_, gp, vp0, elcbo_beta, ns_ent_K, True, compute_var, theta_bnd)
                return r[0], r[1]

            return vb_train_mc_fun

        x, y, x_tab, y_tab, it = vo.minimize_adam(closure_factory(), np.zeros(3), max_iter=3, use_early_stopping=False)
        assert it == 3 and [c[0] for c in log] == ["_neg_elcbo"] * 3  # the mirror's host loop, objective by the reference
    finally:
        pyvbmc_amd.unpatch(vo)
    assert vo._neg_elcbo is ref_neg


def test_unsupported_shape_without_the_drop_in_is_a_not_implemented_error():
    """Outside ``patch`` there is nothing to route to: the package itself has no CPU path."""
    assert issubclass(_lib.UnsupportedShape, NotImplementedError)
    ctx = _lib.Context(-1)
    ctx._lib.vbmc_last_error.restype = __import__("ctypes").c_char_p
    with pytest.raises(NotImplementedError):
        ctx.check(_lib.E_UNSUP)
    ctx.close()
