"""``pyvbmc_amd.patch(vo)`` (this module: pyvbmc_amd/dropin.py) -- the whole drop-in in one call.

Rebinds, on the reference's ``pyvbmc.vbmc.variational_optimization`` module object (or any module
with the same names):

* the four leaf names the ELBO path binds by value at import:
  ``entmc_vbmc``, ``entlb_vbmc``, ``_gp_log_joint``, ``_neg_elcbo``;
* ``_sieve`` (reference vbmc/variational_optimization.py:660-810): the reference's own function
  still runs -- candidate generation, ``np.random`` consumption, return tuple are its own -- but
  its per-candidate ``_neg_elcbo(theta, gp, vp0, 0, 0, 0, compute_var, theta_bnd)`` loop (:775-787)
  only *records* the candidates (and applies the side effects the real call has on ``vp0``); they
  are then evaluated in ONE device call (``_neg_elcbo_batch``) and the returned
  candidate arrays are sorted by the true values, as the reference sorts them (:789-792);
* ``minimize_adam`` (vbmc/minimize_adam.py:84-137): when it is handed ``optimize_vp``'s stochastic
  objective -- the closure ``vb_train_mc_fun`` over ``gp, vp0, elcbo_beta, ns_ent_K, compute_var,
  theta_bnd`` (:238-249) -- the whole loop runs on the device (``minimize_adam_elbo``, SURVEY 8f row
  2: Philox draws, iteration i uses seed + i); any other objective goes to the host loop with the
  reference's signature.

Everything falls back to the per-call path when a precondition of the batched / device-resident
form does not hold (Monte-Carlo entropy in the sieve, a variance term, candidates that differ in a
block that is not optimised, an objective that is not that closure).  A shape the kernels do not cover
at all (``_lib.UnsupportedShape``: D > 32 -- the reference's loops take any D,
entropy/entmc_vbmc.py:64-112 -- or a GP beyond the LDS plans) goes back to the REFERENCE callable
``patch`` replaced, with the reference's own arguments: the drop-in never turns a run the reference could do
into an error, and never routes anywhere but to the reference's own code.  ``unpatch(vo)`` restores the
module.  Nothing here imports the reference: ``vo`` is passed in (or imported on request).
"""
import numpy as np

from . import _lib
from . import entropy as _entropy
from . import minimize_adam as _adam
from . import variational_optimization as _avo

_SAVED = "_pyvbmc_amd_saved"
_LEAVES = ("entmc_vbmc", "entlb_vbmc", "_gp_log_joint", "_neg_elcbo")


_MIRROR_ONLY_KW = ("rng", "seed", "eps_half", "ctx", "rows", "return_raw")  # keyword-only extras of the mirrors


def _with_reference_fallback(fast, ref):
    """``fast`` with ``ref`` -- the reference callable of the same name -- behind it for shapes the device path
    does not cover.  The mirrors raise before they have touched ``vp`` (the C call fails while planning; the reference's
    in-place max-shift of theta's eta tail is idempotent), so the reference simply starts over."""

    def call(*a, **kw):
        try:
            return fast(*a, **kw)
        except _lib.UnsupportedShape:
            for k in _MIRROR_ONLY_KW:
                kw.pop(k, None)
            return ref(*a, **kw)

    call.__name__ = getattr(fast, "__name__", "call")
    call.__doc__ = getattr(fast, "__doc__", None)
    call.__wrapped__ = fast
    return call


def _apply_vp_side_effects(vp0, theta):
    """What the reference's ``_neg_elcbo`` does to the ``vp`` it is given before any arithmetic
    (:1080-1085): ``set_parameters(theta)`` and, with optimised weights, eta = theta's tail
    max-shifted (the tail of the caller's theta is shifted in place too: view arithmetic)."""
    vp0.set_parameters(theta)
    if vp0.optimize_weights:
        K = vp0.K
        tail = theta[-K:]
        tail -= np.amax(tail)
        vp0.eta = np.reshape(tail, (1, -1))


def _same_fixed_blocks(vps):
    """True when every candidate agrees with the first in the blocks that are NOT optimised (the
    batched call takes those from one ``vp``)."""
    a = vps[0]
    for b in vps[1:]:
        if not a.optimize_mu and not np.array_equal(a.mu, b.mu):
            return False
        if not a.optimize_sigma and not np.array_equal(a.sigma, b.sigma):
            return False
        if not a.optimize_lambd and not np.array_equal(a.lambd, b.lambd):
            return False
        if not a.optimize_weights and not (np.array_equal(a.w, b.w) and np.array_equal(a.eta, b.eta)):
            return False
    return True


def make_sieve(vo, ref_sieve, batch_eval=None):
    """The reference's ``_sieve`` with its candidate loop evaluated in one batched call.
    ``batch_eval(thetas, gp, vp, theta_bnd) -> F[B]`` defaults to the device call; the build
    container's check (tools/check_integration_patch.py) passes the oracle instead."""
    batch_eval = _avo._neg_elcbo_batch if batch_eval is None else batch_eval

    def _sieve(options, optim_state, vp, gp, init_N=None, best_N=1, K=None):
        rec = []          # (theta, vp0) of every deferred candidate, in call order
        state = {"bnd": None, "gp": None, "direct": False}
        real = vo._neg_elcbo

        def recorder(theta, gp_, vp0, beta=0.0, Ns=0, compute_grad=True, compute_var=None, theta_bnd=None,
                     *a, **kw):
            batchable = (Ns == 0 and not compute_grad and not compute_var and not beta and not a and not kw
                         and not state["direct"])
            if not batchable:
                state["direct"] = True  # (one form for the whole sieve: the values must be comparable)
                return real(theta, gp_, vp0, beta, Ns, compute_grad, compute_var, theta_bnd, *a, **kw)
            _apply_vp_side_effects(vp0, theta)
            rec.append((np.array(theta, dtype=np.float64), vp0))
            state["bnd"], state["gp"] = theta_bnd, gp_
            # a placeholder that keeps the reference's argsort an identity: the call index
            return float(len(rec) - 1), None, 0.0, 0.0, 0.0

        vo._neg_elcbo = recorder
        try:
            out = ref_sieve(options, optim_state, vp, gp, init_N, best_N, K)
        finally:
            vo._neg_elcbo = real
        if not rec:
            return out
        vp0_vec, vp0_type = out[0], out[1]
        vps = [v for _, v in rec]
        # the reference sorted by the placeholders: its arrays are still in candidate order
        if len(vp0_vec) != len(rec) or not all(a is b for a, b in zip(vp0_vec, vps)):
            # not the loop this wrapper was written for (another version of the caller): evaluate what it
            # returned, in the order it returned it
            F = np.array([real(v.get_parameters(), state["gp"], v, 0, 0, 0, False, state["bnd"])[0] for v in vp0_vec])
            order = np.argsort(F)
            return (vp0_vec[order], vp0_type[order]) + tuple(out[2:])
        thetas = np.stack([t for t, _ in rec])
        F = None
        if _same_fixed_blocks(vps):
            try:
                F = np.asarray(batch_eval(thetas, state["gp"], vps[0], state["bnd"]), dtype=np.float64)
            except _lib.UnsupportedShape:
                F = None  # a shape the batch kernels do not cover: one call each (which routes on to the reference)
        if F is None:  # candidates differ in a block the batched call would take from one vp: one call each
            F = np.array([real(t.copy(), state["gp"], v, 0, 0, 0, False, state["bnd"])[0] for t, v in rec])
        order = np.argsort(F)  # (:789-792)
        return (vp0_vec[order], vp0_type[order]) + tuple(out[2:])

    _sieve.__wrapped__ = ref_sieve
    return _sieve


_CLOSURE_VARS = ("gp", "vp0", "elcbo_beta", "ns_ent_K", "compute_var", "theta_bnd")


def _objective_parts(f):
    """(gp, vp0, beta, ns_ent_K, compute_var, theta_bnd) if ``f`` is ``optimize_vp``'s stochastic
    objective (:238-249: a closure named ``vb_train_mc_fun`` over exactly those names), else None."""
    code = getattr(f, "__code__", None)
    cells = getattr(f, "__closure__", None)
    if code is None or cells is None or code.co_name != "vb_train_mc_fun":
        return None
    env = dict(zip(code.co_freevars, cells))
    if not all(n in env for n in _CLOSURE_VARS):
        return None
    try:
        return tuple(env[n].cell_contents for n in _CLOSURE_VARS)
    except ValueError:  # an empty cell
        return None


def make_minimize_adam(vo, loop=None):
    """``minimize_adam`` that runs ``optimize_vp``'s stochastic objective on the device.
    ``loop(theta0, gp, vp0, ns, theta_bnd, beta, **kw)`` defaults to ``minimize_adam_elbo``."""
    loop = _adam.minimize_adam_elbo if loop is None else loop

    def minimize_adam(f, x0, lb=None, ub=None, tol_fun=0.001, max_iter=10000, master_min=0.001, master_max=0.1,
                      master_decay=200, use_early_stopping=True):
        parts = _objective_parts(f)
        # the device loop evaluates the MIRROR's objective: only when the module still routes there
        if parts is not None and getattr(vo._neg_elcbo, "__wrapped__", vo._neg_elcbo) is _avo._neg_elcbo:
            gp, vp0, beta, ns, compute_var, theta_bnd = parts
            if ns > 0 and not compute_var and (not beta or not np.isfinite(beta)):
                try:
                    return loop(np.array(x0, dtype=np.float64), gp, vp0, ns, theta_bnd, 0.0, lb, ub, tol_fun, max_iter,
                                master_min, master_max, master_decay, use_early_stopping)
                except _lib.UnsupportedShape:
                    pass  # the host loop below around f, whose _neg_elcbo routes on to the reference
        return _adam.minimize_adam(f, x0, lb, ub, tol_fun, max_iter, master_min, master_max, master_decay,
                                   use_early_stopping)

    return minimize_adam


def patch(vo=None, sieve=True, adam=True, _batch_eval=None, _loop=None):
    """Route ``vo`` (default: ``pyvbmc.vbmc.variational_optimization``) through the MI355X path.
    Idempotent; returns ``vo``.  ``sieve`` / ``adam`` switch the two caller-side rebinding steps
     off individually; the ``_``-prefixed arguments are the build-container
    check's injection points."""
    if vo is None:
        import importlib

        vo = importlib.import_module("pyvbmc.vbmc.variational_optimization")
    if hasattr(vo, _SAVED):
        unpatch(vo)
    saved = {n: getattr(vo, n) for n in _LEAVES + ("_sieve", "minimize_adam") if hasattr(vo, n)}
    setattr(vo, _SAVED, saved)
    mirrors = {"entmc_vbmc": _entropy.entmc_vbmc, "entlb_vbmc": _entropy.entlb_vbmc,
               "_gp_log_joint": _avo._gp_log_joint, "_neg_elcbo": _avo._neg_elcbo}
    for n, fast in mirrors.items():
        setattr(vo, n, _with_reference_fallback(fast, saved[n]) if callable(saved.get(n)) else fast)
    if sieve and "_sieve" in saved:
        vo._sieve = make_sieve(vo, saved["_sieve"], _batch_eval)
    if adam and "minimize_adam" in saved:
        vo.minimize_adam = make_minimize_adam(vo, _loop)
    return vo


def unpatch(vo):
    """Put back what ``patch`` replaced."""
    saved = getattr(vo, _SAVED, None)
    if saved is None:
        return vo
    for n, v in saved.items():
        setattr(vo, n, v)
    delattr(vo, _SAVED)
    _avo.clear_fast_path()  # (the repeat record holds the last call's vp, gp and context)
    return vo
