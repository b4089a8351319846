#!/usr/bin/env python
"""GPU-clock timeline of consecutive host-driven evaluations without a tracer: prep start / end,
finish start / publish / end (library with prep.hip and entropy.hip built with -DFIN_TIMES), read
back after each call WITHOUT synchronising in between (the stamps of the call before last).
    tools/build_times_variant.sh && VBMC_HIP_LIB=$PWD/variants/libvbmc_st.so python tools/step_times.py [config [Ns_total]]
(Ns_total: the samples of ONE GPU's share -- bench.py's config 5 runs 4e6 / 8)"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd import VariationalPosterior  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo  # noqa: E402

ctx = _lib.Context(0)
_lib.set_default_context(ctx)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = synthetic.make_workload(cfg, S=1, Ns_total=int(float(sys.argv[2])) if len(sys.argv) > 2 else None)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
theta = vp.get_parameters()
lib = C.CDLL(str(_lib.LIB_PATH))
fb = (C.c_ulonglong * (4 + 3 * 64))()
pb = (C.c_ulonglong * 66)()
n = 300
for i in range(n):
    _neg_elcbo(theta.copy(), gp, vp, 0.0, wl.NsK, True, False, None, 0.0, False, rng="philox", seed=100 + i)
t0 = time.perf_counter()
for i in range(n):
    _neg_elcbo(theta.copy(), gp, vp, 0.0, wl.NsK, True, False, None, 0.0, False, rng="philox", seed=1000 + i)
dt = (time.perf_counter() - t0) / n * 1e6
ctx.synchronize()
lib.vbmc_debug_fin_times(fb)
lib.vbmc_debug_prep_times(pb)
hs = (C.c_int64 * (64 * 6))()
lib.vbmc_debug_host_stamps(hs)
# clock correlation: GPU ticks (100 MHz) <-> host ns; keep the tightest of 50 brackets
best = None
for _ in range(50):
    hb, ha, g = C.c_int64(), C.c_int64(), C.c_ulonglong()
    lib.vbmc_debug_clock_pair(ctx._h, C.byref(hb), C.byref(ha), C.byref(g))
    if best is None or ha.value - hb.value < best[1] - best[0]:
        best = (hb.value, ha.value, g.value)
# the stamp kernel starts ~L after the launch call and its store lands ~1 us before host_after
off_ns = best[1] - 1000 - best[2] * 10  # host_ns = gpu_ticks * 10 + off_ns   (+- ~1.5 us)
print("clock bracket %.1f us" % ((best[1] - best[0]) / 1e3))
f3 = np.array(fb, dtype=np.float64)[4:].reshape(64, 3) * 10 + off_ns   # host ns
f3 = f3[np.argsort(f3[:, 0])]
print("finish start -> GP word stored (GPU clock; taken from the next launch's record)  %.1f us" % np.median((f3[3:-2, 2] - f3[2:-3, 0]) / 1e3))
f = f3[:, :2]
p = np.array(pb, dtype=np.float64)[2:] * 10 + off_ns
h = np.array(hs, dtype=np.float64).reshape(64, 6)
f = f[np.argsort(f[:, 0])]
p = np.sort(p)
h = h[np.argsort(h[:, 0])]
# align: for every host call find its prep start (first prep after the call's entry) and its finish
rows = []
for c in h[2:-2]:
    pi = np.searchsorted(p, c[0])
    fi = np.searchsorted(f[:, 0], c[0])
    if pi >= len(p) or fi >= len(f):
        continue
    rows.append([c[1] - c[0], p[pi] - c[1], f[fi, 0] - p[pi], f[fi, 1] - f[fi, 0], c[3] - f[fi, 1], c[4] - f[fi, 1], c[5] - c[4]])
r = np.array(rows) / 1e3
names = ["entry -> prep launch issued (theta->pack, plan, CPU pack write)", "prep launch issued -> prep starts on the GPU",
         "prep start -> finish start (prep, entropy kernel)", "finish start -> completion word stored (GPU clock)",
         "word stored -> host has seen the GP word", "word stored -> host has seen the main word", "main word seen -> exit (finalise)"]
print("call period through the Python mirror %.1f us" % dt)
for i, nm in enumerate(names):
    print("%-70s %6.1f us" % (nm, np.median(r[:, i])))
print("finish start -> its last workgroup ends (last launch; generation of the next draws) %.1f us" % ((fb[2] - fb[0]) / 100.0))
print("exit -> next entry (Python mirror) %.1f us" % np.median((h[3:-1, 0] - h[2:-2, 5]) / 1e3))
