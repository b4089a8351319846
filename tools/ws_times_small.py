#!/usr/bin/env python
"""Per-workgroup phases of entmc_ws_kernel<10,13> at config 3's shape with FEW rows per component (the fixed cost of a
launch): entropy_ws.hip built with -DWS_TIMES (tools/ws_variant.sh times "-DWS_TIMES").   NsK values on the command line."""
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
for nsk in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    wl = synthetic.make_workload(3, Ns_total=nsk * 50)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.ctx = ctx
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    eps = np.random.default_rng(1).standard_normal((wl.K, wl.NsK // 2, wl.D))
    ctx.set_timing(True)
    ms = []
    for i in range(8):
        entmc_vbmc(vp, wl.NsK, (True,) * 4, True, eps_half=eps)
        ms.append(ctx.last_kernel_ms(0))
    ctx.set_timing(False)
    plan = ctx.last_entmc_plan()
    n = plan["chunks"] * wl.K if not plan["span"] else 502
    buf = (C.c_ulonglong * (4 * n))()
    assert lib.vbmc_debug_ws_times(buf, 4 * n) == 0
    t = np.array(buf, dtype=np.float64).reshape(n, 4) / 100.0
    t0 = t[:, 0].min()
    print(f"NsK={nsk}: kernel {1e3 * np.median(ms):.2f} us by events; plan {plan}; {n} workgroups")
    print("  start skew after the first start: med %.2f p90 %.2f max %.2f us" % tuple(np.percentile(t[:, 0] - t0, [50, 90, 100])))
    print("  prologue (start -> exp2 coefficients loaded) med %.2f us" % np.median(t[:, 1] - t[:, 0]))
    print("  stamp 1 -> end of the batch loop med %.2f  min %.2f  max %.2f us" % (np.median(t[:, 2] - t[:, 1]), (t[:, 2] - t[:, 1]).min(), (t[:, 2] - t[:, 1]).max()))
    print("  epilogue (reduction -> stamp 3) med %.2f us" % np.median(t[:, 3] - t[:, 2]))
    print("  first start -> last stamp %.2f us (the rest of the kernel's duration: dispatch in front, partial-row stores and drain behind)" % (t[:, 3].max() - t0))
