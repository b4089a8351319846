"""``entmc_vbmc`` / ``entlb_vbmc`` -- the reference's entropy estimators on the MI355X.

Same call signatures and return values as
/root/reference/pyvbmc/entropy/entmc_vbmc.py:6-134 and entlb_vbmc.py:6-180:
``(H: float, dH: ndarray)`` with ``dH = [mu 'F' | sigma | lambda | w]`` holding only
the enabled blocks.  All arithmetic runs in HIP kernels (vbmc_entmc / vbmc_entlb);
there is no NumPy fallback.

Random draws of ``entmc_vbmc``.  The reference consumes NumPy's global legacy
stream: for every component j ascending, ``np.random.randn(Ns//2, D)``
(entmc_vbmc.py:64-68).  ``rng="numpy"`` (default) does exactly that on the host and
ships the draws to HBM, so a seeded call returns the reference's value to
rounding.  ``rng="philox"`` generates the draws inside the kernel (Philox4x32-10 +
Box-Muller) from a 64-bit seed derived from ``np.random`` (see ``philox_seed``: runs
stay reproducible under ``np.random.seed``): statistically equivalent, not
stream-identical, and removes the host RNG + PCIe cost.

``vp`` may be any object with the reference's public mixture attributes (a real
``pyvbmc`` ``VariationalPosterior`` included): see pyvbmc_amd/_duck.py.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._duck import ctx_of, upload_vp

DEFAULT_RNG = os.environ.get("VBMC_HIP_RNG", "numpy")


def _even_ns(Ns):
    """The reference rounds Ns up to even (entmc_vbmc.py:61)."""
    return math.ceil(Ns / 2) * 2


def host_randn(n, threads=0):
    """The next ``n`` values of ``np.random.randn`` -- same values, same state left behind -- drawn
    by the library's multi-threaded restatement of NumPy's legacy stream (csrc/host_randn.hip);
    ``None`` when the global generator is not the MT19937 one that restatement covers."""
    out = np.empty(int(n))
    with _lib.NP_STREAM_LOCK:  # get_state -> draw -> set_state is one section (see _lib.NP_STREAM_LOCK)
        st = np.random.get_state(legacy=True)
        if not isinstance(st, tuple) or st[0] != "MT19937":  # (a replaced bit generator reports a dict)
            return None
        key = np.array(st[1], dtype=np.uint32)
        pos, has_gauss, gauss = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
        rc = _lib.load().vbmc_mt19937_randn(key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos), C.byref(has_gauss),
                                            C.byref(gauss), _lib.ptr(out), int(n), int(threads) or _lib.host_threads())
        if rc != 0:
            raise RuntimeError(f"vbmc_mt19937_randn failed ({rc})")
        np.random.set_state(("MT19937", key, pos.value, has_gauss.value, gauss.value))
    return out


_HOST_RANDN_MIN = 1 << 16  # below this np.random.randn itself is as fast as the call's fixed costs


def upload_reference_eps(ctx, K, D, ns, eps_half=None):
    """Make the reference's draws (or the caller's ``eps_half``) the context's resident eps: this
    rank's share of the antithetic-pair rows of every component.  Drawn inside the library --
    multi-threaded, straight into pinned memory -- when the job is large enough to pay for it."""
    h = ns // 2
    r0 = h * ctx.rank // ctx.world
    r1 = h * (ctx.rank + 1) // ctx.world
    if eps_half is None:
        if K * h * D >= _HOST_RANDN_MIN and ctx.set_eps_numpy(K, h, D, r0, r1 - r0, view=_np_fingerprint):
            return
        eps_half = draw_eps_half(K, D, ns)
    eps_half = np.ascontiguousarray(eps_half, dtype=np.float64)
    if eps_half.shape != (K, h, D):
        raise ValueError(f"eps_half must have shape {(K, h, D)}, got {eps_half.shape}")
    ctx.set_eps(eps_half, r0, r1 - r0)


def draw_eps_half(K, D, Ns):
    """The eps the reference draws (entmc_vbmc.py:67): ``np.random.randn(Ns/2, D)`` for j = 0..K-1, in
    that order, i.e. the next K*Ns/2*D values of the global stream in C order."""
    h = _even_ns(Ns) // 2
    n = K * h * D
    if n >= _HOST_RANDN_MIN:
        flat = host_randn(n)
        if flat is not None:
            return flat.reshape(K, h, D)
    eps = np.empty((K, h, D))
    for j in range(K):
        eps[j] = np.random.randn(h, D)
    return eps


class _NpFingerprint:
    """A cheap fingerprint of NumPy's global MT19937 state (its position and first two key words),
    read in place: ``np.random.get_state()`` copies 2.5 KB and costs ~35 us, more than the host has
    per evaluation.  The words are read through the bit generator's public ctypes interface
    (``BitGenerator.ctypes.state_address``); the struct layout behind it (``uint32 key[624]; int
    pos``) is checked against ``get_state`` whenever a generator is bound, and the generator object
    itself is HELD -- its memory cannot be freed under the view -- and compared by identity on every
    read: after ``np.random.set_bit_generator`` the reader re-binds to the new generator (or, if that
    is not an MT19937 / the layout check fails, reports no fingerprint and every call draws a fresh
    seed).  Any consumption of the stream moves ``pos`` or refills the key, so the cached second
    value of ``randn`` (``has_gauss``) cannot change without the fingerprint changing."""

    def __init__(self):
        self._get = getattr(np.random, "get_bit_generator", None) or (lambda: np.random.mtrand._rand._bit_generator)
        self._bg = self._words = None
        self._bind()

    def _bind(self):
        self._bg = self._words = None
        try:
            bg = self._get()
            if type(bg).__name__ != "MT19937":
                return
            words = (C.c_uint32 * 625).from_address(bg.ctypes.state_address)
            st = np.random.get_state(legacy=True)
            if (not isinstance(st, tuple) or st[0] != "MT19937"
                    or (words[0], words[1], words[624]) != (int(st[1][0]), int(st[1][1]), int(st[2]))):
                return
            self._bg, self._words = bg, words  # the generator is held as long as its memory is viewed
        except Exception:
            self._bg = self._words = None

    def __call__(self):
        try:
            cur = self._get()
        except Exception:
            return None
        if cur is not self._bg:
            self._bind()  # the global generator was replaced: view the new one (after the layout check)
        w = self._words
        return None if w is None else (w[0], w[1], w[624])


_np_fingerprint = _NpFingerprint()


def philox_seed(ctx):
    """The device generator's seed for a call that was not given one.

    Per context a sequence ``s0, s0+1, s0+2, ...``: ``s0`` is one ``np.random.randint`` draw,
    taken on first use and again whenever NumPy's global generator was touched since the last
    call (re-seeded or consumed by anyone) -- so scripts stay reproducible under
    ``np.random.seed`` and replaying from a re-seed replays the draws.  Consecutive seeds are
    what lets the library generate the next evaluation's draws while the host is busy with
    this one's result (``vbmc_neg_elcbo``, option ``elbo_ahead``)."""
    st = ctx.__dict__.get("_philox_seq")
    fp = _np_fingerprint()
    if st is not None and fp is not None and st[1] == fp:
        seed = (st[0] + 1) & 0x3FFFFFFFFFFFFFFF
    else:
        seed = int(np.random.randint(0, 2**62, dtype=np.int64))
        fp = _np_fingerprint()
    ctx.__dict__["_philox_seq"] = (seed, fp)
    return seed


def _n_grad(vp, bits):
    D, K = vp.D, vp.K
    return D * K * bool(bits & 1) + K * bool(bits & 2) + D * bool(bits & 4) + K * bool(bits & 8)


def entmc_vbmc(vp, Ns, grad_flags=tuple([True] * 4), jacobian_flag=True, *, rng=None, seed=None,
               eps_half=None, ctx=None, return_raw=False, rows=None):
    """Monte-Carlo entropy of the variational posterior and its gradient.

    ``rows=(begin, count)`` (Philox draws only) evaluates that slice of every component's
    antithetic-pair rows instead of the context's own share -- the contribution of one rank of a
    sharded job, so the slices of "virtual ranks" can be added up on one GPU."""
    ctx = ctx_of(vp, ctx)
    upload_vp(vp, ctx)
    D, K = vp.D, vp.K
    ns = _even_ns(Ns)
    h = ns // 2
    rng = DEFAULT_RNG if rng is None else rng
    # this context's share of the antithetic-pair rows (all of them on one GPU)
    r0 = h * ctx.rank // ctx.world
    r1 = h * (ctx.rank + 1) // ctx.world
    if rows is not None:
        if eps_half is not None or rng != "philox":
            raise ValueError("rows= needs rng='philox' (uploaded draws follow the context's own share)")
        r0, r1 = int(rows[0]), int(rows[0]) + int(rows[1])
    if eps_half is not None or rng == "numpy":
        upload_reference_eps(ctx, K, D, ns, eps_half)
        mode, seed = _lib.EPS_RESIDENT, 0
    elif rng == "philox":
        if seed is None:
            seed = philox_seed(ctx)
        mode = _lib.EPS_PHILOX
    else:
        raise ValueError(f"unknown rng {rng!r}")
    bits = _lib.flags_to_bits(grad_flags)
    H = C.c_double()
    dH = np.empty(_n_grad(vp, bits))
    raw = np.empty(1 + D * K + 2 * K + D) if return_raw else None
    ctx.check(
        ctx._lib.vbmc_entmc(
            ctx._h, ns, mode, C.c_uint64(seed), r0, r1 - r0, bits, int(bool(jacobian_flag)),
            C.byref(H), _lib.ptr(dH), _lib.ptr(raw),
        )
    )
    if return_raw:
        return H.value, dH, raw
    return H.value, dH


def entlb_vbmc(vp, grad_flags=tuple([True] * 4), jacobian_flag=True, *, ctx=None):
    """Entropy lower bound (Jensen) of the variational posterior and its gradient."""
    ctx = ctx_of(vp, ctx)
    upload_vp(vp, ctx)
    bits = _lib.flags_to_bits(grad_flags)
    H = C.c_double()
    dH = np.empty(_n_grad(vp, bits))
    ctx.check(ctx._lib.vbmc_entlb(ctx._h, bits, int(bool(jacobian_flag)), C.byref(H), _lib.ptr(dH)))
    return H.value, dH
