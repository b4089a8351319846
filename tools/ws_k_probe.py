"""Entropy main kernel (wave-split form) across K: kernel microseconds per (D, K), samples scaled so that
every case has the same number of (sample, component) pairs per component-row."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for K in (int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "24,28,32,36,40,44,48,50,52,56,60,64".split(","))):
    wl = synthetic.make_workload(3, D=D, K=K, N=50, Ns_total=20000 * K)
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    ms = []
    for i in range(10):
        ctx.set_timing(i >= 2)
        entmc_vbmc(vp, wl.NsK, (True,) * 4, True, rng="philox", seed=5)
        if i >= 2:
            ms.append(ctx.last_kernel_ms(0))
    ctx.set_timing(False)
    t = float(np.median(ms))
    pairs = wl.NsK * K * K
    print(f"D={D} K={K:4d} KT={-(-K // 4):3d}: {ctx.last_entmc_plan()} {1e3 * t:8.1f} us  {1e12 * t / pairs:7.2f} ps per (sample, component) pair")
