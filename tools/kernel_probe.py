#!/usr/bin/env python
"""Time the Monte-Carlo entropy's main kernel alone (HIP events of the dispatch itself) at a
BASELINE config's per-GPU shape, value-only and value+gradient, resident draws.
    python tools/kernel_probe.py [config ...]      (VBMC_HIP_LIB selects a library variant)
"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic  # noqa: E402

NS = {2: 100_000, 3: 1_000_000, 5: 500_000}
ctx = _lib.Context(0)
_lib.set_default_context(ctx)
tag = os.environ.get("VBMC_HIP_LIB", "default").split("/")[-1]
for cfg in [int(a) for a in sys.argv[1:]] or [3]:
    wl = synthetic.make_workload(cfg, Ns_total=NS[cfg])
    vp = VariationalPosterior(wl.D, wl.K)
    vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
    vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
    eps = np.random.default_rng(1).standard_normal((wl.K, wl.NsK // 2, wl.D))
    for gf in ((False,) * 4, (True,) * 4):
        H, dH = entmc_vbmc(vp, wl.NsK, gf, True, eps_half=eps)
        ctx.set_timing(True)
        ms = []
        for _ in range(30):
            entmc_vbmc(vp, wl.NsK, gf, True, eps_half=eps)  # re-uploads eps: fine, only the kernel is timed
            ms.append(ctx.last_kernel_ms(0))
        ctx.set_timing(False)
        print(f"{tag:28s} config {cfg} grad={int(gf[0])}: kernel {1e3 * np.median(ms):7.2f} us (min {1e3 * min(ms):.2f})  "
              f"H={H:.12g} plan={ctx.last_entmc_plan()}", flush=True)
