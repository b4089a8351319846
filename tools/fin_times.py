#!/usr/bin/env python
"""When does the finish launch publish its completion word, and when does its last workgroup end?
(library built with SRC=entropy.hip tools/gp_variants.sh fin "-DFIN_TIMES")
    VBMC_HIP_LIB=variants/libvbmc_fin.so python tools/fin_times.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd import VariationalPosterior  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=1)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
theta = vp.get_parameters()
lib = C.CDLL(str(_lib.LIB_PATH))
buf = (C.c_ulonglong * (4 + 3 * 64))()
import time
rows = []
for i in range(40):
    t_a = time.perf_counter()
    _neg_elcbo(theta.copy(), gp, vp, 0.0, wl.NsK, True, False, None, 0.0, False, rng="philox", seed=100 + i)
    t_b = time.perf_counter()
    ctx.synchronize()
    assert lib.vbmc_debug_fin_times(buf) == 0
    if i >= 36:
        print("call %.1f us; raw" % ((t_b - t_a) * 1e6), list(buf))
    t = np.array(buf, dtype=np.int64)
    rows.append(((t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0))
r = np.array(rows[10:])
print("finish launch: completion word %.1f us after its first workgroup starts (median; p90 %.1f); last workgroup ends at %.1f us"
      % (np.median(r[:, 0]), np.percentile(r[:, 0], 90), np.median(r[:, 1])))
