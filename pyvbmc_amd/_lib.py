"""ctypes binding of libvbmc_hip.so (the C ABI declared in include/vbmc_hip.h).

This is the only route from Python to the GPU: no PyTorch, no CPU fallback.  If
the shared library is missing, or no gfx950 device is visible when a context is
requested, the import / call raises -- it never silently computes elsewhere.
"""
import ctypes as C
import os
import sys
import threading
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("VBMC_HIP_LIB", _HERE / "libvbmc_hip.so"))

EPS_RESIDENT, EPS_PHILOX = 0, 1
MEAN_ZERO, MEAN_CONST, MEAN_NEGQUAD = 0, 1, 2
E_ARG, E_HIP, E_RCCL, E_NODEV, E_UNSUP, E_NONFINITE = -1, -2, -3, -4, -5, -6
W_GP_CHANGED = 1  # vbmc_neg_elcbo: the watched GP arrays changed under it (vbmc_set_gp_watch)
W_NOT_FUSED = 2   # vbmc_adam_run_auto: the one-launch loop does not apply (or gave up): run batches with vbmc_adam_run


class VbmcHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvbmc_hip error {code}: {msg}")
        self.code = code


class NoDeviceError(VbmcHipError):
    pass


class UnsupportedShape(NotImplementedError):
    """VBMC_E_UNSUP: a shape the kernels do not cover (D > 32, a GP too large for the LDS plans ...).  A
    ``NotImplementedError`` -- there is no CPU fallback inside this package -- that ``pyvbmc_amd.patch`` recognises:
    under the drop-in such a call goes back to the reference callable it replaced (pyvbmc_amd/dropin.py)."""


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p


class ElboOpts(C.Structure):
    _fields_ = [
        ("ns_per_comp", C.c_int64),
        ("eps_mode", C.c_int),
        ("seed", C.c_uint64),
        ("compute_grad", C.c_int),
        ("optimize_mask", C.c_int),
        ("row_begin", C.c_int64),
        ("row_count", C.c_int64),
        ("bnd_lb", _dp),
        ("bnd_ub", _dp),
        ("n_bnd", C.c_int),
        ("tol_con", C.c_double),
        ("weight_threshold", C.c_double),
        ("weight_penalty", C.c_double),
    ]


class ElboCall(C.Structure):
    """vbmc_elbo_call: vbmc_neg_elcbo's arguments in one block (a foreign call converts every argument on every call)."""

    _fields_ = [
        ("theta", _dp),
        ("n_theta", C.c_int),
        ("opts", C.POINTER(ElboOpts)),
        ("F", _dp),
        ("dF", _dp),
        ("G", _dp),
        ("H", _dp),
        ("mu_KxD", _dp),
        ("sigma_K", _dp),
        ("lambd_D", _dp),
        ("w_K", _dp),
        ("eta_K", _dp),
    ]


# name -> (restype, argtypes); every symbol include/vbmc_hip.h declares
SIGNATURES = {
    "vbmc_abi_version": (C.c_int, []),
    "vbmc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vbmc_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "vbmc_ctx_destroy": (None, [_vp]),
    "vbmc_last_error": (C.c_char_p, [_vp]),
    "vbmc_device_info": (
        C.c_int,
        [_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)],
    ),
    "vbmc_host_affinity": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vbmc_synchronize": (C.c_int, [_vp]),
    "vbmc_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "vbmc_last_entmc_plan": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "vbmc_ws_span_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vbmc_last_elbo_raw": (C.c_int, [_vp, _dp, C.c_int]),
    "vbmc_armed_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "vbmc_set_gp_watch": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.c_uint64]),
    "vbmc_host_checksum": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint64)]),
    "vbmc_set_timing": (C.c_int, [_vp, C.c_int]),
    "vbmc_last_kernel_ms": (C.c_int, [_vp, C.c_int, _dp]),
    "vbmc_last_host_us": (C.c_int, [_vp, _dp]),
    "vbmc_last_step_marks": (C.c_int, [_vp, _dp]),
    "vbmc_set_release_callback": (C.c_int, [_vp, C.c_void_p, C.c_void_p]),
    "vbmc_set_mixture": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "vbmc_set_mixture_dk": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "vbmc_theta_to_mixture": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "vbmc_mixture_pdf": (C.c_int, [_vp, C.c_int64, _dp, C.c_int, C.c_int, C.c_double, _dp, _dp]),
    "vbmc_set_eps": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, _dp, C.c_int64, C.c_int64]),
    "vbmc_mt19937_randn": (
        C.c_int,
        [C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp, C.c_int64, C.c_int],
    ),
    "vbmc_randn_dev_info": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "vbmc_mt19937_randn_dev": (
        C.c_int,
        [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, _dp, C.c_int64],
    ),
    "vbmc_mt_jump_host": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint64, C.POINTER(C.c_uint32)]),
    "vbmc_mt_jump_polys": (C.c_int, [C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]),
    "vbmc_set_eps_numpy": (
        C.c_int,
        [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_int, C.c_int64, C.c_int,
         C.c_int64, C.c_int64, C.c_int],
    ),
    "vbmc_entmc": (
        C.c_int,
        [_vp, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int, _dp, _dp, _dp],
    ),
    "vbmc_philox_normals": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int64, _dp]),
    "vbmc_entmc_finalize": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _dp, _dp]),
    "vbmc_entlb": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp]),
    "vbmc_set_gp": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp,
         C.POINTER(C.c_int32), _dp, _dp],
    ),
    "vbmc_gp_log_joint": (
        C.c_int,
        [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp],
    ),
    "vbmc_gp_predict": (C.c_int, [_vp, C.c_int64, _dp, C.c_int, C.c_int, _dp, _dp]),
    "vbmc_neg_elcbo": (
        C.c_int,
        [_vp, _dp, C.c_int, C.POINTER(ElboOpts), _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    ),
    "vbmc_neg_elcbo_call": (C.c_int, [_vp, C.POINTER(ElboCall)]),
    "vbmc_neg_elcbo_batch": (
        C.c_int,
        [_vp, _dp, C.c_int, C.c_int, C.POINTER(ElboOpts), _dp, _dp, _dp],
    ),
    "vbmc_adam_begin": (
        C.c_int,
        [_vp, _dp, C.c_int, C.POINTER(ElboOpts), _dp, _dp, C.c_int, C.c_double, C.c_double, C.c_double],
    ),
    "vbmc_adam_run": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "vbmc_adam_run_auto": (C.c_int, [_vp, C.c_int, C.c_double, C.POINTER(C.c_int), _dp, _dp, _dp, _dp]),
    "vbmc_adam_end": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "vbmc_acq_eval": (
        C.c_int,
        [_vp, C.c_int64, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp, _dp],
    ),
    "vbmc_acq_is_set": (C.c_int, [_vp, C.c_int64, _dp, C.c_int, _dp, _dp, _dp]),
    "vbmc_acq_is_eval": (C.c_int, [_vp, C.c_int64, _dp, _dp, C.c_double, _dp, _dp]),
    "vbmc_sq_dist": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int, _dp, _dp, _dp, C.POINTER(C.c_int64)]),
    "vbmc_mixture_sample": (C.c_int, [_vp, C.c_int64, C.c_uint64, C.c_int, _dp, C.POINTER(C.c_int32)]),
    "vbmc_mixture_sample_t": (C.c_int, [_vp, C.c_int64, C.c_uint64, C.c_int, C.c_double, _dp, C.POINTER(C.c_int32)]),
    "vbmc_kl_div_mc": (C.c_int, [_vp, C.c_int64, C.c_uint64, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "vbmc_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "vbmc_comm_init": (C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "vbmc_comm_destroy": (C.c_int, [_vp]),
    "vbmc_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vbmc_comm_allreduce_max": (C.c_int, [_vp, _dp]),
    "vbmc_comm_barrier": (C.c_int, [_vp]),
}

_lib = None
_lib_lock = threading.Lock()


def load():
    """Load (once) and return the shared library; raises if it is missing."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise ImportError(
                    f"{LIB_PATH} not found: build it with `python -m pyvbmc_amd.build` "
                    "(hipcc, gfx950).  pyvbmc_amd has no CPU fallback."
                )
            lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the symbol is missing
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def device_count():
    n = C.c_int(0)
    load().vbmc_device_count(C.byref(n))
    return n.value


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    """float64 pointer of a C-contiguous array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def flags_to_bits(flags):
    if np.isscalar(flags):
        return 15 if flags else 0
    return sum((1 << i) for i, f in enumerate(flags) if f)


class Context:
    """One vbmc_ctx: a device, its stream/scratch and optional RCCL communicator.
    Not thread-safe; create one per host thread."""

    def __init__(self, device=None):
        lib = load()
        if device is None:
            device = int(os.environ.get("VBMC_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        # device == -1: host-only context (mixture bookkeeping + host finalisation; every
        # kernel-launching call raises NoDeviceError) -- used by the CPU tests only
        h = _vp()
        rc = lib.vbmc_ctx_create(int(device), C.byref(h))
        if rc != 0:
            msg = (lib.vbmc_last_error(None) or b"").decode()
            raise (NoDeviceError if rc == E_NODEV else VbmcHipError)(rc, msg)
        self._h = h
        self._lib = lib
        self.device = int(device)
        self.rank, self.world = 0, 1

    def close(self):
        if getattr(self, "_h", None):
            # argument blocks built for this handle (variational_optimization._FusedCall) and the repeat record that
            # points at one of them must not outlive it
            self.__dict__.pop("_fused_last", None)
            self.__dict__.pop("_fused_cache", None)
            vo = sys.modules.get(__package__ + ".variational_optimization")
            if vo is not None:
                fs = vo._fast_last[0]
                if fs is not None and fs.ctx is self:
                    vo._fast_last[0] = None
            self._lib.vbmc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            msg = (self._lib.vbmc_last_error(self._h) or b"").decode()
            if rc == E_UNSUP:
                raise UnsupportedShape(msg)
            if rc == E_ARG:
                raise ValueError(msg)
            if rc == E_NODEV:
                raise NoDeviceError(rc, msg)
            raise VbmcHipError(rc, msg)

    # -- thin typed wrappers --------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        cu, clk, mem = C.c_int(), C.c_int(), C.c_uint64()
        self.check(self._lib.vbmc_device_info(self._h, name, 256, C.byref(cu), C.byref(clk), C.byref(mem)))
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_khz": clk.value,
                "hbm_bytes": mem.value}

    def host_affinity(self):
        """``(bound, n_cpus)``: whether creating this context narrowed the calling thread's CPU affinity to the device's
        NUMA node (vbmc_host_affinity; ``VBMC_HOST_AFFINITY=0`` disables it), and the CPUs in the thread's set afterwards."""
        b, n = C.c_int(), C.c_int()
        self.check(self._lib.vbmc_host_affinity(self._h, C.byref(b), C.byref(n)))
        return bool(b.value), n.value

    def synchronize(self):
        self.check(self._lib.vbmc_synchronize(self._h))

    def set_timing(self, on):
        """HIP event pair around the dominant kernels (off by default: each record costs ~6 us).  ``on=2`` also records
        the pair around predict's variance product (``last_kernel_ms(5)``), which sits between predict's launches."""
        self.check(self._lib.vbmc_set_timing(self._h, int(on) if on else 0))

    def set_option(self, key, value):
        """Per-context test / measurement switch (vbmc_set_option in include/vbmc_hip.h)."""
        self.check(self._lib.vbmc_set_option(self._h, key.encode(), int(value)))

    def last_entmc_plan(self):
        """Launch geometry of the most recent Monte-Carlo entropy: which kernel ran, how many
        64-row batches each workgroup looped over (span mode: the longest part), partial rows per
        component, draw source; ``span``: the wave-split kernel ran in span mode (front / filler parts
        sized to end together, csrc/entropy_args.h) rather than on equal chunks."""
        out = (C.c_int * 4)()
        self.check(self._lib.vbmc_last_entmc_plan(self._h, out))
        return {"kernel": ("valu", "ws", "small", "mfma", "adam_fused", "ws", "ws")[out[0]] if out[0] >= 0 else None, "rg": out[1],
                "chunks": out[2], "resident_draws": bool(out[3]), "span": out[0] in (5, 6), "adam_tail": out[0] == 6}

    def philox_normals(self, K, n_half, D, seed, row_begin=0, row_count=None):
        """[K][row_count][D] draws of the device generator (vbmc_philox_normals)."""
        row_count = n_half - row_begin if row_count is None else row_count
        out = np.empty((K, row_count, D))
        self.check(self._lib.vbmc_philox_normals(self._h, K, n_half, D, C.c_uint64(seed), row_begin, row_count, ptr(out)))
        return out

    def armed_stats(self):
        """Counters of the polled step: armed evaluations used / cancelled / recovered from a late go
        word, identity checks of result blocks done / failed, evaluations repeated because their
        completion word never came (vbmc_armed_stats)."""
        out = (C.c_uint64 * 6)()
        self.check(self._lib.vbmc_armed_stats(self._h, out))
        return dict(zip(("hits", "cancels", "late", "ident_checked", "ident_bad", "lost"), (int(v) for v in out)))

    def last_elbo_raw(self, D, K):
        """Raw entropy accumulator of the most recent Monte-Carlo ``vbmc_neg_elcbo`` (additive over
        row slices; see include/vbmc_hip.h)."""
        out = np.empty(1 + D * K + 2 * K + D)
        self.check(self._lib.vbmc_last_elbo_raw(self._h, ptr(out), out.size))
        return out

    def last_kernel_ms(self, which=0):
        v = C.c_double()
        self.check(self._lib.vbmc_last_kernel_ms(self._h, which, C.byref(v)))
        return v.value

    def last_host_us(self):
        out = np.zeros(5)
        self.check(self._lib.vbmc_last_host_us(self._h, ptr(out)))
        return out

    def last_step_marks(self):
        """us from the last fused evaluation's entry to its GP word / its entropy word, and where the GP
        sums ran (vbmc_last_step_marks)."""
        out = np.zeros(4)
        self.check(self._lib.vbmc_last_step_marks(self._h, ptr(out)))
        return {"gp_word_us": out[0], "entropy_word_us": out[1],
                "gp_sums_in": ("prep launch", "finish launch", "entropy launch")[int(out[2])]}

    def set_mixture(self, mu_DK, sigma, lambd, w, eta):
        mu_DK = np.asarray(mu_DK, dtype=np.float64)
        D, K = mu_DK.shape
        mu_kd = f64(mu_DK.T)
        s, l, ww = f64(np.ravel(sigma)), f64(np.ravel(lambd)), f64(np.ravel(w))
        e = f64(np.ravel(eta)) if eta is not None and np.size(eta) == K else None
        if s.size != K or l.size != D or ww.size != K:
            raise ValueError("mixture attribute shapes do not match (D, K)")
        self.check(self._lib.vbmc_set_mixture(self._h, D, K, ptr(mu_kd), ptr(s), ptr(l), ptr(ww), ptr(e)))
        self.D, self.K = D, K

    def set_eps(self, eps_half, row_begin=0, row_count=None):
        eps_half = f64(eps_half)
        K, h, D = eps_half.shape
        if row_count is None:
            row_count = h - row_begin
        self.check(self._lib.vbmc_set_eps(self._h, K, h, D, ptr(eps_half), row_begin, row_count))

    def set_eps_numpy(self, K, n_half, D, row_begin=0, row_count=None, threads=0, view=None):
        """Draw the reference's eps (the next K*n_half*D values of np.random.randn, NumPy's global
        state advanced accordingly) and make this context's rows of them the resident draws --
        without the values ever becoming a NumPy array.  False when NumPy's global generator is not
        MT19937 (the caller then draws with NumPy and uses set_eps).

        ``view`` (entropy._NpFingerprint): NumPy's MT19937 state viewed in place.  ``get_state`` +
        ``set_state`` copy 2.5 KB each way and cost ~37 us per call; when the state is exactly what the
        previous call here left behind (same fingerprint) AND that call left no cached second value
        (``has_gauss == 0`` -- the one part of the legacy state the view does not cover, and the one
        that can change without the fingerprint changing), the library reads the key and the
        position through the view and writes them back the same way, under the bit generator's own
        lock.  Anything else -- first call, a stream somebody consumed or re-seeded, an odd number of
        values -- takes get_state / set_state as before."""
        if row_count is None:
            row_count = n_half - row_begin
        with NP_STREAM_LOCK:
            left = self.__dict__.get("_np_left")
            fp = view() if view is not None and left is not None else None
            if fp is not None and fp == left and view._words is not None:
                bg, words = view._bg, view._words
                has_gauss, gauss = C.c_int(0), C.c_double(0.0)
                with bg.lock:
                    addr = C.addressof(words)
                    rc = self._lib.vbmc_set_eps_numpy(self._h, C.cast(addr, C.POINTER(C.c_uint32)),
                                                      C.cast(addr + 624 * 4, C.POINTER(C.c_int)), C.byref(has_gauss),
                                                      C.byref(gauss), K, n_half, D, row_begin, row_count,
                                                      threads or host_threads())
                    if has_gauss.value:  # (an odd number of values: the cached one goes in through set_state)
                        key = np.array(words[:624], dtype=np.uint32)
                        pos = int(words[624])
                if has_gauss.value:
                    np.random.set_state(("MT19937", key, pos, has_gauss.value, gauss.value))
                    self.__dict__["_np_left"] = None
                else:
                    self.__dict__["_np_left"] = view()
                self.check(rc)
                return True
            st = np.random.get_state(legacy=True)
            if not isinstance(st, tuple) or st[0] != "MT19937":  # (a replaced bit generator reports a dict)
                return False
            key = np.array(st[1], dtype=np.uint32)
            pos, has_gauss, gauss = C.c_int(int(st[2])), C.c_int(int(st[3])), C.c_double(float(st[4]))
            rc = self._lib.vbmc_set_eps_numpy(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                              C.byref(has_gauss), C.byref(gauss), K, n_half, D, row_begin, row_count,
                                              threads or host_threads())
            # (the state is written back even on an upload error: the values have been drawn)
            np.random.set_state(("MT19937", key, pos.value, has_gauss.value, gauss.value))
            self.__dict__["_np_left"] = view() if view is not None and has_gauss.value == 0 else None
        self.check(rc)
        return True

    def comm_init(self, uid_bytes, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid_bytes)
        self.check(self._lib.vbmc_comm_init(self._h, buf, rank, world))
        self.rank, self.world = rank, world

    def comm_info(self):
        """(rank, world) of the communicator as RCCL itself reports them; (0, 1) without one."""
        r, w = C.c_int(), C.c_int()
        self.check(self._lib.vbmc_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_barrier(self):
        self.check(self._lib.vbmc_comm_barrier(self._h))

    def comm_max(self, value):
        v = C.c_double(value)
        self.check(self._lib.vbmc_comm_allreduce_max(self._h, C.byref(v)))
        return v.value


def comm_unique_id():
    buf = (C.c_uint8 * 128)()
    rc = load().vbmc_comm_unique_id(buf)
    if rc != 0:
        raise VbmcHipError(rc, (load().vbmc_last_error(None) or b"").decode())
    return bytes(buf)


# The library's restatement of np.random.randn (vbmc_mt19937_randn / vbmc_set_eps_numpy) advances
# NumPy's global stream as get_state -> C call (GIL released) -> set_state.  The lock serialises
# those sections among this package's own callers; it cannot include other threads that call
# np.random.* directly in that window (NumPy's own lock is not reachable from here): such draws
# would be overwritten and replayed -- do not draw from the global generator concurrently.
NP_STREAM_LOCK = threading.Lock()


def host_threads():
    """Worker threads for the host draw stream: the cores this process may run on (cgroup /
    affinity aware, unlike hardware_concurrency()), at most 64."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(64, n))


_default_ctx = None


def default_context():
    """Process-wide context used when a call is not given one explicitly."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx
