#!/usr/bin/env python
"""Phases of the finish launch between the entropy kernel and the completion word (in-kernel stamps).
    tools/build_times_variant.sh && VBMC_HIP_LIB=$PWD/variants/libvbmc_st.so python tools/fin_phases.py [config [Ns_total]]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd import VariationalPosterior  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _lib.Context(0)
_lib.set_default_context(ctx)
ctx.set_option("elbo_arm", 0)  # (an armed evaluation's launches would overwrite the stamps read here)
wl = synthetic.make_workload(cfg, S=1, Ns_total=int(float(sys.argv[2])) if len(sys.argv) > 2 else None)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
theta = vp.get_parameters()
bnd = synthetic.default_theta_bnd(wl)
lib = C.CDLL(str(_lib.LIB_PATH))
buf = (C.c_ulonglong * 8)()
rows = []
for i in range(40):
    lib.vbmc_debug_fin_phases(buf, 1)
    _neg_elcbo(theta.copy(), gp, vp, 0.0, wl.NsK, True, False, bnd, 0.0, False, rng="philox", seed=100 + i)
    ctx.synchronize()
    lib.vbmc_debug_fin_phases(buf, 0)
    t = np.array(buf, dtype=np.int64)
    rows.append((t[1:7] - t[0]) / 100.0)
r = np.median(np.array(rows[10:]), axis=0)
print("finish launch, us from its first block's start (median of 30): last sum formed %.2f | last result store acknowledged %.2f | "
      "the last block knows it %.2f | its copy acknowledged %.2f | flag stored %.2f | (last reduction block started %.2f)" % tuple(r))
