#!/usr/bin/env python
"""Per-workgroup phases of predict_kstar_mfma_kernel (library built with -DKSTAR_ABL_TIMES)."""
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
n = 896
buf = (C.c_ulonglong * (4 * n))()
assert lib.vbmc_debug_dma_times(buf, 4 * n) == 0
t = np.array(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
t0 = t[:, 0].min()
st, en, t1, t2 = [(t[:, i] - t0) / 100.0 for i in (0, 1, 2, 3)]
print("span %.1f us" % en.max())
print("start: p10 %.2f med %.2f p90 %.2f max %.2f" % tuple(np.percentile(st, [10, 50, 90, 100])))
print("staging  (start->barrier): med %.2f p90 %.2f" % tuple(np.percentile(t1 - st, [50, 90])))
print("product  (barrier->mfma done): med %.2f p90 %.2f" % tuple(np.percentile(t2 - t1, [50, 90])))
print("epilogue (->end): med %.2f p90 %.2f" % tuple(np.percentile(en - t2, [50, 90])))
print("end: p10 %.2f med %.2f p90 %.2f" % tuple(np.percentile(en, [10, 50, 90])))
