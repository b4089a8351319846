#!/usr/bin/env python
"""Span mode of the wave-split entropy kernel (csrc/entropy_args.h WsSpan): kernel time over the front workgroup's
share of a CU's batches, per padded D and register-array size, against the equal-chunk grid.  The table in
csrc/entropy.hip (ws_front_default) was read off this probe's output.
    python tools/ws_front_probe.py [grad=1]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import VariationalPosterior, _lib, entmc_vbmc, synthetic  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
grad = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fronts = [0, 500, 540, 580, 620, 660, 700, 740]
KS = {4: 16, 8: 32, 10: 40, 13: 50, 16: 64, 20: 80, 25: 100}
for D in (2, 4, 6, 8, 10, 12, 16):
    for KT, K in KS.items():
        if grad and 4 * D + 3 * KT > 95:
            continue  # one wave per SIMD: no filler parts
        wl = synthetic.make_workload(3, D=D, K=K, N=50, Ns_total=K * 128 * (8000 // K))  # ~8 000 batches in all
        vp = VariationalPosterior(wl.D, wl.K)
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        row = []
        for f in fronts:
            ctx.set_option("ws_span", 1 if f else 0)
            if f:
                ctx.set_option("ws_front", f)
            ms = []
            for i in range(9):
                ctx.set_timing(i >= 2)
                entmc_vbmc(vp, wl.NsK, (bool(grad),) * 4, True, rng="philox", seed=5)
                if i >= 2:
                    ms.append(ctx.last_kernel_ms(0))
            ctx.set_timing(False)
            assert ctx.last_entmc_plan()["span"] == bool(f), ctx.last_entmc_plan()
            row.append(1e3 * float(np.median(ms)))
        best = int(np.argmin(row[1:])) + 1
        print(f"D={D:2d} KT={KT:2d} K={K:3d} grad={grad}: chunks {row[0]:6.1f} | " + " ".join(f"{f}:{t:6.1f}" for f, t in zip(fronts[1:], row[1:]))
              + f" | best {fronts[best]} ({row[best] / row[0]:.3f} of chunks)", flush=True)
