"""Adam for the stochastic ELBO optimisation -- mirror of
``pyvbmc/vbmc/minimize_adam.py`` (reference :8-146) plus its device-resident
specialisation.

``minimize_adam(f, x0, ...)`` is the generic host loop with the reference's signature,
update rule, stopping rule and return tuple; ``f`` is any callable returning
``(value, gradient)``.

``minimize_adam_elbo(theta0, gp, vp, ...)`` runs the same loop for the one objective
PyVBMC gives it, ``lambda t: _neg_elcbo(t, gp, vp, beta, Ns, compute_grad=True,
theta_bnd=theta_bnd)[:2]`` (reference variational_optimization.py:238-249), with theta,
the Adam moments and the mixture kept on the GPU (csrc/adam.hip): the host only looks at
the ``y_tab`` / ``x_tab`` rows every ``batch_size`` iterations to apply the stopping rule,
which is when the reference applies it too.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._duck import ctx_of, optimize_mask, store_mixture, upload_vp
from .entropy import _even_ns, philox_seed, upload_reference_eps
from .gp import upload_gp

BATCH_SIZE = 20  # minimize_adam.py:66
_BETA_1, _BETA_2 = 0.9, 0.999


def _window_stop(y_win, x_prev_mean, x_last_mean, tol_fun, batch_size=BATCH_SIZE):
    """The reference's termination test at the end of a minibatch (:107-140): slope of a
    straight-line fit through the last ``batch_size`` objective values against its
    standard error, and the distance between the mean iterates of the last two batches."""
    tol_x, tol_x_max, tol_fun_max = 0.001, 0.1, tol_fun * 100
    half = (batch_size - 1) / 2
    t = np.linspace(-half, half, batch_size)
    coef, cov = np.polyfit(t, y_win, 1, cov=True)
    slope = coef[0]
    err = np.sqrt(cov[0, 0] + tol_fun**2)
    err_max = np.sqrt(cov[0, 0] + tol_fun_max**2)
    dx = np.sqrt(np.sum((x_last_mean - x_prev_mean) ** 2 / batch_size, axis=0))
    return bool((dx < tol_x and abs(slope) < err_max) or (abs(slope) < err and dx < tol_x_max))


def minimize_adam(f, x0, lb=None, ub=None, tol_fun=0.001, max_iter=10000, master_min=0.001,
                  master_max=0.1, master_decay=200, use_early_stopping=True):
    """Host Adam loop; returns ``(x, y, x_tab, y_tab, iterations)`` like the reference.

    As in the reference the first update is applied to ``x0`` in place (``x -= ...`` on the
    caller's array) and ``f`` sees the very array being iterated, so an objective that
    shifts its argument (``_neg_elcbo`` does, on the eta tail) shifts the iterate."""
    fudge = np.sqrt(np.spacing(1))
    n = np.size(x0)
    lb = np.full((n,), -np.inf) if lb is None else lb
    ub = np.full((n,), np.inf) if ub is None else ub
    x_tab = np.zeros((n, max_iter))
    y_tab = np.full((max_iter,), np.nan)
    m = v = 0
    x = x0
    b = BATCH_SIZE
    for i in range(max_iter):
        y_tab[i], g = f(x)
        m = _BETA_1 * m + (1 - _BETA_1) * g
        v = _BETA_2 * v + (1 - _BETA_2) * g**2
        m_hat = m / (1 - _BETA_1 ** (i + 1))
        v_hat = v / (1 - _BETA_2 ** (i + 1))
        step = master_min + (master_max - master_min) * np.exp(-(i + 1) / master_decay)
        x -= step * m_hat / (np.sqrt(v_hat) + fudge)
        x = np.minimum(ub, np.maximum(lb, x))
        x_tab[:, i] = x
        if use_early_stopping and (i + 1) % b == 0 and i + 1 >= 2 * b:
            if _window_stop(
                y_tab[i - b + 1 : i + 1],
                np.mean(x_tab[:, i - 2 * b + 1 : i + 1 - b], axis=1),
                np.mean(x_tab[:, i - b + 1 : i + 1], axis=1),
                tol_fun,
            ):
                break
    x = np.mean(x_tab[:, i - b + 1 : i + 1], axis=1)
    y = np.mean(y_tab[i - b + 1 : i + 1])
    return x, y, x_tab[:, : i + 1], y_tab[: i + 1], i + 1


def minimize_adam_elbo(theta0, gp, vp, Ns, theta_bnd=None, beta=0.0, lb=None, ub=None, tol_fun=0.001,
                       max_iter=10000, master_min=0.001, master_max=0.1, master_decay=200,
                       use_early_stopping=True, *, rng=None, seed=None, eps_half=None, ctx=None,
                       return_parts=False, rows=None, device_stop=True, _between_batches=None):
    """``minimize_adam(lambda t: _neg_elcbo(t, gp, vp, beta, Ns, True, theta_bnd=theta_bnd)[:2],
    theta0, lb, ub, ...)`` with the whole inner loop on the device.

    Same return tuple ``(x, y, x_tab, y_tab, iterations)``.  Differences from the host loop:
    ``theta0`` is not modified; the draws come from the in-kernel Philox generator (iteration
    ``i`` uses ``seed + i``) whatever ``VBMC_HIP_RNG`` says, because fresh NumPy draws would
    have to be uploaded every iteration; ``rng="numpy"`` / ``eps_half`` upload ONE set of
    draws from the NumPy stream that every iteration reuses (common random numbers).  On return ``vp``
    holds the parameters of the last iterate (the reference leaves those of the last
    *evaluated* iterate; its caller overwrites them right away, :283-300).  ``device_stop=False`` keeps the
    stopping rule on the host (batches of 20 iterations) also where the device can apply it itself.
    ``_between_batches(done)`` (tests): called after every batch of the host-driven form."""
    if beta != 0 and np.isfinite(beta):
        raise NotImplementedError("Computation of the gradient of ELBO with full variance not supported")
    ctx = ctx_of(vp, ctx)
    K, D = vp.K, vp.D
    theta0 = _lib.f64(np.ravel(theta0))
    n = theta0.size
    ns = _even_ns(Ns)
    if ns <= 0:
        raise ValueError("minimize_adam_elbo needs Ns > 0 (the stochastic entropy)")
    upload_vp(vp, ctx)
    upload_gp(gp, ctx)
    opts = _lib.ElboOpts()
    opts.ns_per_comp, opts.compute_grad, opts.optimize_mask = ns, 1, optimize_mask(vp)
    # rows=(begin, count): one (virtual) rank's slice of every component's antithetic-pair rows
    opts.row_begin, opts.row_count = (0, -1) if rows is None else (int(rows[0]), int(rows[1]))
    keep = []
    if theta_bnd is not None:
        blb, bub = _lib.f64(theta_bnd["lb"].ravel()), _lib.f64(theta_bnd["ub"].ravel())
        keep += [blb, bub]
        opts.bnd_lb, opts.bnd_ub, opts.n_bnd = _lib.ptr(blb), _lib.ptr(bub), blb.size
        opts.tol_con = float(theta_bnd["tol_con"])
        opts.weight_threshold = float(theta_bnd.get("weight_threshold", 0.0))
        opts.weight_penalty = float(theta_bnd.get("weight_penalty", 0.0))
    mode = "philox" if rng is None else rng  # fresh draws per iteration need the in-kernel generator
    if eps_half is not None or mode == "numpy":
        upload_reference_eps(ctx, K, D, ns, eps_half)
        opts.eps_mode, opts.seed = _lib.EPS_RESIDENT, 0
    elif mode == "philox":
        if seed is None:
            seed = philox_seed(ctx)
            # the loop consumes seed, seed+1, ... seed+max_iter-1: keep the context's sequence clear of them
            ctx.__dict__["_philox_seq"] = ((seed + int(max_iter)) & 0x3FFFFFFFFFFFFFFF, ctx.__dict__["_philox_seq"][1])
        opts.eps_mode, opts.seed = _lib.EPS_PHILOX, int(seed)
    else:
        raise ValueError(f"unknown rng {mode!r}")
    box_lb = None if lb is None else _lib.f64(np.broadcast_to(lb, (n,)))
    box_ub = None if ub is None else _lib.f64(np.broadcast_to(ub, (n,)))
    if (box_lb is None) != (box_ub is None):
        box_lb = np.full(n, -np.inf) if box_lb is None else box_lb
        box_ub = np.full(n, np.inf) if box_ub is None else box_ub
    lib, h_ = ctx._lib, ctx._h
    ctx.check(lib.vbmc_adam_begin(h_, _lib.ptr(theta0), n, C.byref(opts), _lib.ptr(box_lb), _lib.ptr(box_ub),
                                  int(max_iter), float(master_min), float(master_max), float(master_decay)))
    b = BATCH_SIZE
    x_rows = np.empty((max_iter, n))
    y_tab = np.full((max_iter,), np.nan)
    G_tab, H_tab = np.empty(max_iter), np.empty(max_iter)
    done, auto = 0, False
    try:
        if use_early_stopping and device_stop:
            # where the run has the one-launch form, its workgroups apply the stopping rule themselves: one call
            nd = C.c_int(0)
            rc = lib.vbmc_adam_run_auto(h_, int(max_iter), float(tol_fun), C.byref(nd), _lib.ptr(y_tab), _lib.ptr(x_rows),
                                        _lib.ptr(G_tab), _lib.ptr(H_tab))
            if rc != _lib.W_NOT_FUSED:  # (else: the batches below)
                ctx.check(rc)
                done, auto = nd.value, True
        while not auto and done < max_iter:
            # without early stopping nothing needs the host until the end
            step = min(b if use_early_stopping else max_iter, max_iter - done)
            ctx.check(lib.vbmc_adam_run(h_, step, _lib.ptr(y_tab[done:]), _lib.ptr(x_rows[done:]),
                                        _lib.ptr(G_tab[done:]), _lib.ptr(H_tab[done:])))
            done += step
            i = done - 1
            if _between_batches is not None:
                _between_batches(done)
            if use_early_stopping and done % b == 0 and done >= 2 * b:
                if _window_stop(y_tab[i - b + 1 : i + 1], x_rows[i - 2 * b + 1 : i + 1 - b].mean(axis=0),
                                x_rows[i - b + 1 : i + 1].mean(axis=0), tol_fun):
                    break
    finally:
        mu = np.empty((K, D))
        sg, lm, w, eta = np.empty(K), np.empty(D), np.empty(K), np.empty(K)
        th = np.empty(n)
        it = C.c_int()
        rc = lib.vbmc_adam_end(h_, _lib.ptr(th), _lib.ptr(mu), _lib.ptr(sg), _lib.ptr(lm), _lib.ptr(w),
                               _lib.ptr(eta), C.byref(it))
    ctx.check(rc)
    store_mixture(vp, mu, sg, lm, w, eta if vp.optimize_weights else None)
    i = done - 1
    lo = max(i - b + 1, 0)
    x_tab = x_rows[:done].T.copy()
    x = np.mean(x_tab[:, lo : i + 1], axis=1)
    y = np.mean(y_tab[lo : i + 1])
    out = (x, y, x_tab, y_tab[:done].copy(), done)
    if return_parts:
        return out + (G_tab[:done].copy(), H_tab[:done].copy())
    return out
