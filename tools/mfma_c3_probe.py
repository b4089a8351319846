"""BASELINE config 3's shape (and neighbours) through the matrix-pipe form (VBMC_MFMA_ANY=1 lifts the
'only where the wave-split kernel runs one wave per SIMD' rule) against the wave-split kernel."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
for D, K, ns in ((10, 50, 1_000_000), (10, 64, 1_280_000), (10, 48, 960_000), (12, 50, 1_000_000), (10, 100, 2_000_000),
                 (12, 100, 2_000_000), (8, 48, 960_000)):
    wl = synthetic.make_workload(3, D=D, K=K, N=50, Ns_total=ns)
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
        out[form] = (float(np.median(ms)), ctx.last_entmc_plan(), H, dH)
    d = float(np.max(np.abs(out[0][3] - out[1][3])) / np.max(np.abs(out[0][3])))
    print(f"D={D:2d} K={K:4d} NsK={wl.NsK}: {out[0][1]['kernel']} {1e3 * out[0][0]:7.1f} us   {out[1][1]['kernel']} {1e3 * out[1][0]:7.1f} us "
          f"(rg {out[1][1]['rg']}, chunks {out[1][1]['chunks']})  ratio {out[0][0] / out[1][0]:.3f}   |dH diff| {d:.1e}")
