#!/usr/bin/env python
"""Per-workgroup start/end of predict_var_dma_kernel (a library built with -DDMA_ABL_TIMES):
    VBMC_HIP_LIB=variants/libvbmc_times.so python tools/dma_times.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=1)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
xs = np.random.default_rng(7).standard_normal((8192, wl.D))
for _ in range(5):
    gp.predict(xs, separate_samples=True)
lib = C.CDLL(str(_lib.LIB_PATH))
n = 512
buf = (C.c_ulonglong * (4 * n))()
assert lib.vbmc_debug_dma_times(buf, 4 * n) == 0
t = np.array(buf, dtype=np.uint64).reshape(n, 4)
t0 = t[:, 0].min()
st = (t[:, 0] - t0).astype(float) / 100.0  # us (100 MHz)
en = (t[:, 1] - t0).astype(float) / 100.0
P = t[:, 3].astype(int)
hw = t[:, 2].astype(int)
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
xcc = (hw >> 16) & 0xF if False else (np.arange(n) % 8)
print("kernel span %.1f us; first-wave starts: median %.2f max %.2f" % (en.max(), np.median(st[:512]), st[:512].max()))
for p in sorted(set(P)):
    m = P == p
    print("P=%2d n=%3d start med %.1f  dur med %.2f min %.2f max %.2f  per-panel %.3f  end max %.1f" % (
        p, m.sum(), np.median(st[m]), np.median((en - st)[m]), (en - st)[m].min(), (en - st)[m].max(),
        np.median((en - st)[m]) / p, en[m].max()))
