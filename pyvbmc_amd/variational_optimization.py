"""The ELBO objective on the MI355X: ``_gp_log_joint``, ``_neg_elcbo`` and the
soft-bound losses, with the reference's signatures and return conventions.

Mirrors /root/reference/pyvbmc/vbmc/variational_optimization.py:
``_soft_bound_loss`` (:609-657), ``_vp_bound_loss`` (:503-606),
``_neg_elcbo`` (:991-1235), ``_gp_log_joint`` (:1238-1606).  The optimiser loop
that calls them (``optimize_vp``, ``_sieve``, ``minimize_adam``) is the next row
of SURVEY.md section 8f and is not part of this module.

All ELBO arithmetic happens behind the C ABI; ``_neg_elcbo`` without
``separate_K`` is ONE library call (vbmc_neg_elcbo: theta -> mixture, G/dG
kernel, entropy kernels, optional all-reduce, bound losses) with a single
device round trip.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._duck import ctx_of, optimize_mask, upload_vp
from .entropy import DEFAULT_RNG, _even_ns, entlb_vbmc, entmc_vbmc, philox_seed, upload_reference_eps
from .gp import upload_gp


def _soft_bound_loss(x, slb, sub, tol_con=1e-3, compute_grad=False):
    """Quadratic penalty outside the soft bounds (host; O(len(x)), not a kernel)."""
    x = np.asarray(x, dtype=np.float64)
    ell = (sub - slb) * tol_con
    below = x < slb
    above = x > sub
    y = 0.0
    dy = np.zeros(x.shape)
    if np.any(below):
        y += 0.5 * np.sum(((slb[below] - x[below]) / ell[below]) ** 2)
        dy[below] = (x[below] - slb[below]) / ell[below] ** 2
    if np.any(above):
        y += 0.5 * np.sum(((x[above] - sub[above]) / ell[above]) ** 2)
        dy[above] = (x[above] - sub[above]) / ell[above] ** 2
    return (y, dy) if compute_grad else y


def _vp_bound_loss(vp, theta, theta_bnd, tol_con=1e-3, compute_grad=True):
    """Soft-bound loss on (mu, ln sigma*lambda, eta) and its gradient wrt theta (host)."""
    D, K = vp.D, vp.K
    pos = 0
    if vp.optimize_mu:
        mu = theta[: D * K]
        pos = D * K
    else:
        mu = vp.mu.ravel(order="F")
    if vp.optimize_sigma:
        ln_sigma = theta[pos : pos + K]
        pos += K
    else:
        ln_sigma = np.log(vp.sigma.ravel())
    ln_lambd = theta[pos : pos + D] if vp.optimize_lambd else np.log(vp.lambd.ravel())
    ln_scale = np.reshape(ln_lambd, (-1, 1)) + np.reshape(ln_sigma, (1, -1))
    parts = []
    if vp.optimize_mu:
        parts.append(np.ravel(mu))
    if vp.optimize_sigma or vp.optimize_lambd:
        parts.append(ln_scale.ravel(order="F"))
    if vp.optimize_weights:
        parts.append(np.ravel(theta[-K:]))
    ext = np.concatenate(parts)
    lb, ub = theta_bnd["lb"].ravel(), theta_bnd["ub"].ravel()
    if not compute_grad:
        return _soft_bound_loss(ext, lb, ub, tol_con)
    L, dL = _soft_bound_loss(ext, lb, ub, tol_con, compute_grad=True)
    out = []
    pos = 0
    if vp.optimize_mu:
        out.append(dL[: D * K])
        pos = D * K
    if vp.optimize_sigma or vp.optimize_lambd:
        block = np.reshape(dL[pos : pos + D * K], (D, K))  # C-order, as the reference (:585-587)
        if vp.optimize_sigma:
            out.append(block.sum(axis=0))
        if vp.optimize_lambd:
            out.append(block.sum(axis=1))
    if vp.optimize_weights:
        out.append(dL[-K:])
    return L, np.concatenate(out)


def _gp_log_joint(vp, gp, grad_flags, avg_flag=True, jacobian_flag=True, compute_var=False,
                  separate_K=False, *, ctx=None):
    """Expected variational log joint under the GP surrogate.

    Returns ``(G, dG, varG, dvarG, var_ss)`` or, with ``separate_K``,
    ``(G, dG, varG, dvarG, var_ss, I_sk, J_sjk)`` exactly like the reference
    (shapes: S==1 squeezes G/dG, varG stays an array, :1599-1602).
    """
    if np.isscalar(grad_flags):
        grad_flags = (bool(grad_flags),) * 4
    bits = _lib.flags_to_bits(grad_flags)
    ctx = ctx_of(vp, ctx)
    upload_vp(vp, ctx)
    upload_gp(gp, ctx)
    D, K, S = vp.D, vp.K, len(gp.posteriors)
    jac = bool(jacobian_flag)
    n_dG = (D * K * bool(bits & 1) + jac * (K * bool(bits & 2) + D * bool(bits & 4) + K * bool(bits & 8)))
    averaged = S > 1 and avg_flag
    G = np.empty(S)
    dG = np.empty((n_dG, S)) if bits else None
    varG = np.empty(S) if compute_var else None
    var_ss = C.c_double(0.0)
    I_sk = np.empty((S, K)) if separate_K else None
    J_sjk = np.empty((S, K, K)) if (separate_K and compute_var) else None
    ctx.check(
        ctx._lib.vbmc_gp_log_joint(
            ctx._h, bits, int(bool(avg_flag)), int(jac), int(compute_var), _lib.ptr(G),
            _lib.ptr(dG), _lib.ptr(varG), C.byref(var_ss), _lib.ptr(I_sk), _lib.ptr(J_sjk),
        )
    )
    if averaged:
        G_out = G[0]
        dG_out = dG.ravel()[:n_dG].copy() if bits else None
        varG_out = varG[0] if compute_var else None
    else:
        G_out = G[0] if S == 1 else G
        dG_out = (dG[:, 0].copy() if S == 1 else dG) if bits else None
        varG_out = varG if compute_var else None
    vss = var_ss.value if (averaged and compute_var) else 0
    if separate_K:
        return G_out, dG_out, varG_out, None, vss, I_sk, J_sjk
    return G_out, dG_out, varG_out, None, vss


def _neg_elcbo(theta, gp, vp, beta=0.0, Ns=0, compute_grad=True, compute_var=None, theta_bnd=None,
               _entropy_alpha=0.0, separate_K=False, *, rng=None, seed=None, eps_half=None, ctx=None,
               rows=None):
    """Negative evidence lower (confidence) bound and its gradient.

    Same positional signature, mutation of ``vp`` (and of the caller's ``theta``
    eta tail, :1082-1085) and return arity as the reference: ``(F, dF, G, H, varF)``
    or the 11-tuple when ``separate_K``.  Keyword-only extras select the source of
    the Monte-Carlo draws (see pyvbmc_amd.entropy); ``rows=(begin, count)`` (fused path, Philox
    draws) evaluates that slice of every component's antithetic-pair rows instead of the context's
    own share -- one rank's part of a sharded job (``ctx.last_elbo_raw`` then holds its additive
    entropy accumulator).
    """
    # ---- the optimiser's inner call, second evaluation onwards: same objects, same shape of request ----
    # (everything between two evaluations is on the step's critical path with the device idle; a record of the
    # last fused call -- the OBJECTS it was made with, held, so identities cannot be recycled -- lets a repeat go
    # straight to the C call.  Whatever may have changed behind an unchanged identity is still looked at:
    # the optimise flags, the GP records' identities, the bounds' arrays and scalars; contents of the GP arrays
    # are checksummed by the library while the device works, as on the general path.)
    fs = _fast_last[0]
    if (fs is not None and fs.vp is vp and fs.gp is gp and fs.bnd is theta_bnd and Ns == fs.Ns and beta == 0.0
            and compute_grad is fs.cg and not compute_var and not separate_K and ctx is fs.ctx_arg and rng == fs.rng
            and eps_half is None and rows is None and type(theta) is np.ndarray and theta.size == fs.n_theta):
        rc = fs.call(theta, seed)
        if rc is not None:
            return rc
    # (general path from here on: the repeat record is dropped FIRST -- _fused_call below re-binds the shared
    # argument block's bounds, and an exception between there and the C call must not leave a record behind that
    # still passes its identity checks against the old arguments; a new one is published only after a successful call)
    _fast_last[0] = None
    if not math.isfinite(beta):
        beta = 0
    if compute_var is None:
        compute_var = beta != 0
    if compute_grad and beta != 0 and compute_var != 2:
        raise NotImplementedError(
            "Computation of the gradient of ELBO with full variance not supported"
        )
    if separate_K and compute_grad:
        raise ValueError(
            "Computing the gradient of variational parameters and "
            "requesting per-component results at the same time."
        )
    ctx_arg = ctx
    ctx = ctx_of(vp, ctx)
    K, D = vp.K, vp.D

    if separate_K or compute_var or beta != 0:
        if rows is not None:
            raise ValueError("rows= is only supported by the fused evaluation")
        return _neg_elcbo_composed(theta, gp, vp, beta, Ns, compute_grad, compute_var, theta_bnd,
                                   separate_K, rng, seed, eps_half, ctx)

    # ---- fused path: one library call ---------------------------------------------------
    # (this sits between two evaluations of the optimiser -- on the step's critical path, with the
    # device idle: attribute reads and ctypes field writes are kept to what changes call to call)
    # theta overwrites every optimised block; the current attributes only need to be on
    # the device when some block is NOT optimised or the context holds another (D, K)
    mask = optimize_mask(vp)
    if mask != 15 or getattr(ctx, "D", None) != D or getattr(ctx, "K", None) != K:
        upload_vp(vp, ctx)
    upload_gp(gp, ctx, lazy=True)  # (content checksum: taken by the library while the device works)
    n_theta = theta.size if type(theta) is np.ndarray else np.size(theta)
    fc = _fused_call(ctx, D, K, n_theta, theta_bnd)
    fc.th[:] = theta
    opts = fc.opts
    ns = _even_ns(Ns) if Ns > 0 else 0
    mode = 0
    if ns > 0:
        mode = DEFAULT_RNG if rng is None else rng
        if rows is not None and (eps_half is not None or mode != "philox"):
            raise ValueError("rows= needs rng='philox' (uploaded draws follow the context's own share)")
        if eps_half is not None or mode == "numpy":
            upload_reference_eps(ctx, K, D, ns, eps_half)
            mode, seed = _lib.EPS_RESIDENT, 0
        elif mode == "philox":
            if seed is None:
                seed = philox_seed(ctx)
            mode = _lib.EPS_PHILOX
        else:
            raise ValueError(f"unknown rng {mode!r}")
    shape = (ns, 1 if compute_grad else 0, mask, mode, rows)
    if shape != fc.shape:  # the fields that stay the same through an optimisation
        opts.ns_per_comp, opts.compute_grad, opts.optimize_mask = ns, shape[1], mask
        opts.row_begin, opts.row_count = (0, -1) if rows is None else (int(rows[0]), int(rows[1]))
        opts.eps_mode = mode
        fc.shape = shape
    if ns > 0:
        opts.seed = seed
    # The reference's side effects on vp and on the caller's theta depend on theta alone: the library calls
    # fc.released() once its launches are out and the device is at work (vbmc_set_release_callback), so they cost
    # nothing between two evaluations; whatever path did not get there applies them after the call.
    fc.side = (vp, theta, mask, K)
    try:
        rc = fc.fn(*fc.args)
        if rc != 0:
            if rc == _lib.W_GP_CHANGED:
                # a GP array was edited in place since the upload: what came back was computed on the old GP
                upload_gp(gp, ctx)
                rc = fc.fn(*fc.args)  # (theta's tail is already shifted: shifting it again changes nothing)
            if rc != 0:
                ctx.check(rc)
        if fc.side is not None:
            fc.apply_side_effects()
    finally:
        fc.side = None
    if rows is None and eps_half is None and (ns == 0 or mode == _lib.EPS_PHILOX) and type(compute_grad) is bool:
        _fast_last[0] = _FastElbo(ctx_arg, ctx, fc, vp, gp, theta_bnd, Ns, compute_grad, rng, mask, D, K, n_theta, mode)
    return fc.F.value, (fc.dF.copy() if compute_grad else None), fc.G.value, fc.H.value, 0


def _neg_elcbo_batch(thetas, gp, vp, theta_bnd=None, *, ctx=None, return_parts=False):
    """The sieve's inner loop in one call (reference ``_sieve``,
    vbmc/variational_optimization.py:775-787): ``F[b] = _neg_elcbo(thetas[b], gp, vp, 0, 0,
    compute_grad=False, theta_bnd=theta_bnd)[0]`` for every row of ``thetas`` -- lower-bound
    entropy, no gradient.  ``vp`` supplies D, K, the optimise flags and the values of
    non-optimised blocks; unlike the per-candidate call it is NOT mutated, nor are the rows."""
    ctx = ctx_of(vp, ctx)
    thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
    B, n_theta = thetas.shape
    upload_vp(vp, ctx)
    upload_gp(gp, ctx)
    opts = _lib.ElboOpts()
    opts.ns_per_comp, opts.compute_grad, opts.optimize_mask = 0, 0, optimize_mask(vp)
    keep = []
    if theta_bnd is not None:
        lb, ub = _lib.f64(theta_bnd["lb"].ravel()), _lib.f64(theta_bnd["ub"].ravel())
        keep += [lb, ub]
        opts.bnd_lb, opts.bnd_ub, opts.n_bnd = _lib.ptr(lb), _lib.ptr(ub), lb.size
        opts.tol_con = float(theta_bnd["tol_con"])
        opts.weight_threshold = float(theta_bnd.get("weight_threshold", 0.0))
        opts.weight_penalty = float(theta_bnd.get("weight_penalty", 0.0))
    F, G, H = np.empty(B), np.empty(B), np.empty(B)
    ctx.check(
        ctx._lib.vbmc_neg_elcbo_batch(
            ctx._h, _lib.ptr(thetas), B, n_theta, C.byref(opts), _lib.ptr(F), _lib.ptr(G), _lib.ptr(H)
        )
    )
    return (F, G, H) if return_parts else F


_RELEASE_CB = os.environ.get("VBMC_RELEASE_CB", "1") != "0"  # measurement aid: 0 = side effects after the call
_FAST_PATH = os.environ.get("VBMC_FAST_PATH", "1") != "0"    # measurement aid: 0 = every call takes the general path
_fast_last = [None]  # the last fused Monte-Carlo / lower-bound call's record (_FastElbo), or None


def clear_fast_path():
    """Drop the repeat record of the last fused ``_neg_elcbo`` call (it holds that call's ``vp``, ``gp`` and context):
    called by ``unpatch``, ``invalidate_gp`` and ``Context.close``; the next call takes the general path."""
    _fast_last[0] = None


class _FastElbo:
    """What a repeat of the last fused ``_neg_elcbo`` call needs: the argument block, and the objects and values the
    general path derived it from.  ``call`` re-checks everything that can change behind an unchanged identity and
    returns None (the caller then takes the general path) unless all of it still holds."""

    __slots__ = ("ctx_arg", "ctx", "fc", "vp", "gp", "bnd", "Ns", "cg", "rng", "mask", "D", "K", "n_theta", "philox",
                 "lb", "ub", "tol", "wthr", "wpen", "fused_last")

    def __init__(self, ctx_arg, ctx, fc, vp, gp, bnd, Ns, cg, rng, mask, D, K, n_theta, mode):
        self.ctx_arg, self.ctx, self.fc, self.vp, self.gp, self.bnd = ctx_arg, ctx, fc, vp, gp, bnd
        self.Ns, self.cg, self.rng, self.mask, self.D, self.K, self.n_theta = Ns, cg, rng, mask, D, K, n_theta
        self.philox = Ns > 0 and mode == _lib.EPS_PHILOX
        self.fused_last = ctx.__dict__.get("_fused_last")
        if bnd is not None:
            self.lb, self.ub = bnd["lb"], bnd["ub"]
            self.tol, self.wthr, self.wpen = bnd["tol_con"], bnd.get("weight_threshold", 0.0), bnd.get("weight_penalty", 0.0)
            if not fc.direct:  # the bounds were copied (not float64 / not contiguous): the general path re-copies them
                self.vp = None

    def call(self, theta, seed):
        if not _FAST_PATH:
            return None
        vp, ctx, fc, bnd = self.vp, self.ctx, self.fc, self.bnd
        if (((1 if vp.optimize_mu else 0) | (2 if vp.optimize_sigma else 0) | (4 if vp.optimize_lambd else 0)
             | (8 if vp.optimize_weights else 0)) != self.mask or vp.D != self.D or vp.K != self.K
                or ctx.__dict__.get("_fused_last") is not self.fused_last or getattr(ctx, "D", None) != self.D
                or getattr(ctx, "K", None) != self.K or ctx_of(vp, self.ctx_arg) is not ctx):
            return None
        if self.mask != 15:
            return None  # (a block that is not optimised comes from vp's attributes: upload_vp on the general path)
        if bnd is not None and (bnd["lb"] is not self.lb or bnd["ub"] is not self.ub or bnd["tol_con"] != self.tol
                                or bnd.get("weight_threshold", 0.0) != self.wthr or bnd.get("weight_penalty", 0.0) != self.wpen):
            return None
        gp = self.gp
        ps, X = gp.posteriors, gp.X
        quick = [id(ps), id(X), X.shape[0]]
        for p in ps:
            quick += (id(p), id(p.alpha), id(p.hyp))
        if quick != ctx.__dict__.get("_gp_quick"):
            return None
        fc.th[:] = theta
        if self.philox:
            fc.opts.seed = philox_seed(ctx) if seed is None else seed
        fc.side = (vp, theta, self.mask, self.K)
        try:
            rc = fc.fn(*fc.args)
            if rc != 0:
                if rc == _lib.W_GP_CHANGED:
                    upload_gp(gp, ctx)
                    rc = fc.fn(*fc.args)
                if rc != 0:
                    ctx.check(rc)
            if fc.side is not None:
                fc.apply_side_effects()
        finally:
            fc.side = None
        return fc.F.value, (fc.dF.copy() if self.cg else None), fc.G.value, fc.H.value, 0


class _FusedCall:
    """Pre-built argument block of vbmc_neg_elcbo for one (ctx, D, K, n_theta): the ctypes
    pointers of the persistent buffers are created once, not per evaluation.  The soft
    bounds are re-bound on every call (``bind_bounds``), so a ``theta_bnd`` whose entries
    were reassigned or edited in place is honoured."""

    def __init__(self, ctx, D, K, n_theta):
        self.th = np.empty(n_theta)
        self.dF = np.empty(n_theta)
        self.mu = np.empty((K, D))
        self.sg, self.lm, self.w, self.eta = np.empty(K), np.empty(D), np.empty(K), np.empty(K)
        # the reference's attribute shapes as views of the buffers the library fills
        self.mu_T, self.sg_row, self.lm_col = self.mu.T, self.sg.reshape(1, -1), self.lm.reshape(-1, 1)
        self.w_row, self.eta_row = self.w.reshape(1, -1), self.eta.reshape(1, -1)
        self.F, self.G, self.H = C.c_double(), C.c_double(), C.c_double()
        o = self.opts = _lib.ElboOpts()
        o.row_begin, o.row_count, o.seed = 0, -1, 0
        self.shape = None  # (ns, compute_grad, mask, eps mode, rows) the option block currently holds
        self.lb_src = self.ub_src = self.lb = self.ub = None
        # the arguments in one block, filled once (vbmc_elbo_call): a foreign call converts every argument on every call --
        # 1.8 us for vbmc_neg_elcbo's thirteen against 0.3 us for these two, between two evaluations of the polled step
        self.fn = ctx._lib.vbmc_neg_elcbo_call
        self.side = None  # (vp, theta, mask, K) of the call in flight whose side effects are still to be applied
        self._cb = C.CFUNCTYPE(None, C.c_void_p)(self._released)  # held: the library keeps the raw pointer
        dp = C.POINTER(C.c_double)
        self.block = _lib.ElboCall(
            _lib.ptr(self.th), n_theta, C.pointer(o), C.cast(C.pointer(self.F), dp), _lib.ptr(self.dF),
            C.cast(C.pointer(self.G), dp), C.cast(C.pointer(self.H), dp), _lib.ptr(self.mu), _lib.ptr(self.sg),
            _lib.ptr(self.lm), _lib.ptr(self.w), _lib.ptr(self.eta),
        )
        self.args = (ctx._h, C.byref(self.block))
        if os.environ.get("VBMC_ELBO_CALL_BLOCK", "1") == "0":  # measurement aid: the thirteen-argument entry point
            self.fn = ctx._lib.vbmc_neg_elcbo
            self.args = (
                ctx._h, _lib.ptr(self.th), n_theta, C.byref(o), C.byref(self.F), _lib.ptr(self.dF),
                C.byref(self.G), C.byref(self.H), _lib.ptr(self.mu), _lib.ptr(self.sg), _lib.ptr(self.lm),
                _lib.ptr(self.w), _lib.ptr(self.eta),
            )

    def apply_side_effects(self):
        """vp.set_parameters(theta)'s effects from the arrays the library filled (store_mixture() through views shaped
        once), and the max-shifted eta tail of the caller's theta (variational_optimization.py:1082-1085)."""
        vp, theta, mask, K = self.side
        vp.mu, vp.sigma, vp.lambd, vp.w = self.mu_T.copy(), self.sg_row.copy(), self.lm_col.copy(), self.w_row.copy()
        if mask & 8:
            vp.eta = self.eta_row.copy()
            if type(theta) is np.ndarray and theta.dtype == self.th.dtype:
                theta[-K:] = self.th[-K:]
        if hasattr(vp, "_mode"):
            vp._mode = None  # set_parameters drops the cached mode (variational_posterior.py:759)
        self.side = None  # (only now: an assignment that raised leaves them to be applied -- and to raise -- after the call)

    def _released(self, _user):
        # called by the library from inside vbmc_neg_elcbo, launches released, device at work.  ctypes swallows an
        # exception raised in a callback: a failure here (a vp whose attribute refuses the assignment) leaves
        # self.side set, so the caller applies the side effects again after the C call returns and the error surfaces
        if self.side is not None:
            try:
                self.apply_side_effects()
            except Exception:
                pass

    def bind_bounds(self, theta_bnd):
        o = self.opts
        if theta_bnd is None:
            if self.lb is not None:
                o.bnd_lb, o.bnd_ub, o.n_bnd = None, None, 0
                self.lb_src = self.ub_src = self.lb = self.ub = None
            return
        lb, ub = theta_bnd["lb"], theta_bnd["ub"]
        if lb is not self.lb_src or ub is not self.ub_src:
            # f64() of a contiguous float64 array is a view: in-place edits stay visible
            # through the pointer; anything else is copied afresh on every call
            self.lb, self.ub = _lib.f64(np.ravel(lb)), _lib.f64(np.ravel(ub))
            self.direct = np.shares_memory(self.lb, lb) and np.shares_memory(self.ub, ub)
            self.lb_src, self.ub_src = (lb, ub) if self.direct else (None, None)
            o.bnd_lb, o.bnd_ub, o.n_bnd = _lib.ptr(self.lb), _lib.ptr(self.ub), self.lb.size
        o.tol_con = float(theta_bnd["tol_con"])
        o.weight_threshold = float(theta_bnd.get("weight_threshold", 0.0))
        o.weight_penalty = float(theta_bnd.get("weight_penalty", 0.0))


def _fused_call(ctx, D, K, n_theta, theta_bnd):
    last = ctx.__dict__.get("_fused_last")
    if last is not None and last[0] == D and last[1] == K and last[2] == n_theta:
        fc = last[3]
        fc.bind_bounds(theta_bnd)
        return fc
    # (another argument block takes over: the library's release callback follows it)
    cache = ctx.__dict__.setdefault("_fused_cache", {})
    key = (D, K, n_theta)
    fc = cache.get(key)
    if fc is None:
        if len(cache) > 16:
            cache.clear()
        fc = cache[key] = _FusedCall(ctx, D, K, n_theta)
    ctx.__dict__["_fused_last"] = (D, K, n_theta, fc)
    ctx.check(ctx._lib.vbmc_set_release_callback(ctx._h, C.cast(fc._cb, C.c_void_p) if _RELEASE_CB else None, None))
    fc.bind_bounds(theta_bnd)
    return fc


def _neg_elcbo_composed(theta, gp, vp, beta, Ns, compute_grad, compute_var, theta_bnd, separate_K,
                        rng, seed, eps_half, ctx):
    """Variance / per-component variants (used by ``_eval_full_elcbo``, not by the
    optimiser's inner loop): composed from the individual device calls, following
    the reference's control flow (:1080-1235)."""
    K = vp.K
    vp.set_parameters(theta)
    if vp.optimize_weights:
        # in-place on the caller's array, like the reference's view arithmetic
        tail = theta[-K:]
        tail -= np.amax(tail)
        vp.eta = np.reshape(tail, (1, -1))
    if compute_grad:
        grad_flags = (vp.optimize_mu, vp.optimize_sigma, vp.optimize_lambd, vp.optimize_weights)
    else:
        grad_flags = (False,) * 4
    I_sk = J_sjk = None
    dG = dvarG = None
    if separate_K:
        if compute_var:
            G, _, varG, _, varG_ss, I_sk, J_sjk = _gp_log_joint(vp, gp, grad_flags, 1, 1, compute_var, True, ctx=ctx)
        else:
            G, dG, _, _, _, I_sk, _ = _gp_log_joint(vp, gp, grad_flags, 1, 1, 0, True, ctx=ctx)
            varG = varG_ss = 0
    elif compute_var:
        G, dG, varG, dvarG, varG_ss = _gp_log_joint(vp, gp, grad_flags, 1, 1, compute_var, ctx=ctx)
    else:
        G, dG, _, _, _ = _gp_log_joint(vp, gp, grad_flags, 1, 1, 0, ctx=ctx)
        varG = varG_ss = 0
    if Ns > 0:
        H, dH = entmc_vbmc(vp, Ns, grad_flags, 1, rng=rng, seed=seed, eps_half=eps_half, ctx=ctx)
    else:
        H, dH = entlb_vbmc(vp, grad_flags, 1, ctx=ctx)
    F = -G - H
    if compute_grad:
        dF = -dG - dH
    else:
        dF = None
        dH = None
    varH = 0
    varF = varG + varH if compute_var else 0
    if beta != 0:
        F = F + beta * np.sqrt(varF)
        if compute_grad:
            dF = dF + 0.5 * beta * dvarG / np.sqrt(varF)
    if theta_bnd is not None:
        if compute_grad:
            L, dL = _vp_bound_loss(vp, theta, theta_bnd, tol_con=theta_bnd["tol_con"])
            dF = dF + dL
        else:
            L = _vp_bound_loss(vp, theta, theta_bnd, tol_con=theta_bnd["tol_con"], compute_grad=False)
        F = F + L
        if vp.optimize_weights:
            thresh = theta_bnd["weight_threshold"]
            small = vp.w < thresh
            F = F + np.sum(vp.w * small + thresh * (~small)) * theta_bnd["weight_penalty"]
            if compute_grad:
                ee = np.exp(vp.eta.ravel())
                es = ee.sum()
                wg = theta_bnd["weight_penalty"] * small.ravel()
                dL = np.zeros(dF.shape)
                dL[-K:] = -ee * (ee @ wg) / es**2 + ee * wg / es
                dF = dF + dL
    if separate_K:
        return F, dF, G, H, varF, dH, varG_ss, varG, varH, I_sk, J_sjk
    return F, dF, G, H, varF
