"""Entropy main kernel at D = 20: the wave-split form against the matrix-pipe form (entropy_mfma.hip) over K.
    python tools/mfma_probe.py            -> kernel microseconds (HIP events on the kernel's own dispatch) per form"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
for K in (64, 72, 80, 88, 96, 100, 104, 112, 120, 128):
    wl = synthetic.make_workload(5, K=K, N=50, Ns_total=2500 * K)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    out = {}
    for form in (0, 1):
        ctx.set_option("entmc_mfma", form)
        ms = []
        for i in range(12):
            ctx.set_timing(i >= 2)
            H, dH = entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=5)
            if i >= 2:
                ms.append(ctx.last_kernel_ms(0))
        ctx.set_timing(False)
        out[form] = (float(np.median(ms)), ctx.last_entmc_plan()["kernel"], H, dH)
    d = float(np.max(np.abs(out[0][3] - out[1][3])) / np.max(np.abs(out[0][3])))
    print(f"K={K:4d} rows/comp={wl.NsK // 2}: {out[0][1]} {1e3 * out[0][0]:7.1f} us   {out[1][1]} {1e3 * out[1][0]:7.1f} us   "
          f"ratio {out[0][0] / out[1][0]:.3f}   |dH diff| {d:.1e}")
