#!/usr/bin/env python
"""Per-workgroup phases of entmc_mfma_kernel<20,7,20> at config 5's per-GPU share (entropy_mfma.hip built with -DMFMA_TIMES:
variants/libvbmc_mft.so)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import VariationalPosterior, entmc_vbmc  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
lib = C.CDLL(str(_lib.LIB_PATH))
wl = synthetic.make_workload(5, Ns_total=500_000)
vp = VariationalPosterior(wl.D, wl.K)
vp.ctx = ctx
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
eps = np.random.default_rng(1).standard_normal((wl.K, wl.NsK // 2, wl.D))
ctx.set_timing(True)
ms = []
for i in range(6):
    entmc_vbmc(vp, wl.NsK, (True,) * 4, True, eps_half=eps)
    ms.append(ctx.last_kernel_ms(0))
ctx.set_timing(False)
plan = ctx.last_entmc_plan()
n = plan["chunks"] * wl.K
buf = (C.c_ulonglong * (4 * n))()
assert lib.vbmc_debug_mfma_times(buf, 4 * n) == 0
t = np.array(buf, dtype=np.float64).reshape(n, 4) / 100.0
t0 = t[:, 0].min()
print(f"kernel {1e3 * np.median(ms):.2f} us by events; plan {plan}; {n} workgroups")
start = t[:, 0] - t0
print("workgroup starts: first round med %.2f, second round (start > 20 us) med %.2f us; %d in the second round" % (
    np.median(start[start < 20]), np.median(start[start >= 20]) if np.any(start >= 20) else float("nan"), int(np.sum(start >= 20))))
print("table row -> LDS          med %.2f us (min %.2f max %.2f)" % (np.median(t[:, 1] - t[:, 0]), (t[:, 1] - t[:, 0]).min(), (t[:, 1] - t[:, 0]).max()))
print("batch loop                med %.2f us (min %.2f max %.2f)" % (np.median(t[:, 2] - t[:, 1]), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
print("reduction                 med %.2f us" % np.median(t[:, 3] - t[:, 2]))
print("first start -> last stamp %.2f us" % (t[:, 3].max() - t0))
