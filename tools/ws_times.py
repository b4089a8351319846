#!/usr/bin/env python
"""Per-workgroup phases of entmc_ws_kernel<10,13> at config 3 (entropy_ws.hip built with -DWS_TIMES -DVBMC_DP=10):
start skew, prologue, batch loop, epilogue."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import VariationalPosterior, entmc_vbmc  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(3)
vp = VariationalPosterior(wl.D, wl.K)
vp.ctx = ctx
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
for i in range(6):
    entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=5 + i)
lib = C.CDLL(str(_lib.LIB_PATH))
plan = ctx.last_entmc_plan()
print(plan)
n = 256 + 246 if plan["span"] else 500  # span mode: 256 front parts + 246 fillers (csrc/entropy_args.h WsSpan)
buf = (C.c_ulonglong * (4 * n))()
assert lib.vbmc_debug_ws_times(buf, 4 * n) == 0
t = np.array(buf, dtype=np.float64).reshape(n, 4) / 100.0
t0 = t[:, 0].min()
print("start skew: med %.2f p90 %.2f max %.2f us" % tuple(np.percentile(t[:, 0] - t0, [50, 90, 100])))
print("prologue   %.2f us (med)" % np.median(t[:, 1] - t[:, 0]))
print("batch loop %.2f us (med), min %.2f max %.2f" % (np.median(t[:, 2] - t[:, 1]), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
loop = t[:, 2] - t[:, 1]
end = t[:, 3] - t0
first, second = np.arange(n) < 256, np.arange(n) >= 256  # workgroups b and b + 256 share a CU (tools/probes/placement.hip)
paired = np.arange(n) < n - 256
print("end of the front parts after the first start, by decile:", np.round(np.percentile(end[first], np.arange(0, 101, 10)), 1))
print("end of the filler parts, by decile:                      ", np.round(np.percentile(end[second], np.arange(0, 101, 10)), 1))
print("batch loop by role: first-dispatched of a pair med %.2f | alone on a CU med %.2f | later-dispatched med %.2f us" % (
    np.median(loop[first & paired]), np.median(loop[first & ~paired]) if np.any(first & ~paired) else float("nan"), np.median(loop[second])))
print("end of workgroup after the first start: first-dispatched med %.2f max %.2f | later-dispatched med %.2f max %.2f us" % (
    np.median(end[first]), end[first].max(), np.median(end[second]), end[second].max()))
print("epilogue   %.2f us (med) to the partial-row store" % np.median(t[:, 3] - t[:, 2]))
print("span: first start -> last stamp %.2f us" % (t[:, 3].max() - t0))
