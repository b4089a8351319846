#!/usr/bin/env python
"""Where does the prep launch's time go?  Start and end of every workgroup of the last prep launch (table rows, GP
expected-log-joint blocks, the pack's copy block), from in-kernel stamps.
    SRC=prep.hip tools/gp_variants.sh pt "-DPREP_TIMES"      (or tools/unit_variant.sh)
    VBMC_ELBO_ARM=0 VBMC_HIP_LIB=$PWD/variants/libvbmc_pt.so python tools/prep_times.py [config]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from pyvbmc_amd import _lib, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd import VariationalPosterior  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = _lib.Context(0)
_lib.set_default_context(ctx)
wl = synthetic.make_workload(cfg, S=1)
gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(),
            gpm.GaussianNoise(constant_add=True, user_provided_add=wl.s2 is not None))
gp.update(X_new=wl.X, y_new=wl.y, s2_new=wl.s2, hyp=wl.hyp)
vp = VariationalPosterior(wl.D, wl.K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
theta = vp.get_parameters()
bnd = synthetic.default_theta_bnd(wl)
lib = C.CDLL(str(_lib.LIB_PATH))
buf = (C.c_ulonglong * 2048)()
K, S = wl.K, 1
for i in range(30):
    _neg_elcbo(theta.copy(), gp, vp, 0.0, wl.NsK, True, False, bnd, 0.0, False, rng="philox", seed=100 + i)
    ctx.synchronize()
    assert lib.vbmc_debug_prep_blocks(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(-1, 2)
    if i < 25:
        continue
    marks = ctx.last_step_marks()
    gp_in_prep = marks["gp_sums_in"] == "prep launch"
    n_gp = S * K if gp_in_prep else 0
    nb = K + n_gp + 1  # this launch's blocks (later entries are other launches' -- the stamps are not cleared)
    t0 = t[:nb, 0].min()

    def span(lo, hi, name):
        r = t[lo:hi]
        if len(r) == 0:
            return
        print("  %-10s %3d blocks: first start %5.2f, last start %5.2f, longest %5.2f, last end %5.2f us"
              % (name, len(r), (r[:, 0].min() - t0) / 100.0, (r[:, 0].max() - t0) / 100.0,
                 (r[:, 1] - r[:, 0]).max() / 100.0, (r[:, 1].max() - t0) / 100.0))
    print("evaluation %d, plan %s, GP sums in the %s" % (i, ctx.last_entmc_plan(), "prep launch" if gp_in_prep else "entropy launch"))
    span(0, K, "table")
    span(K, K + n_gp, "GP")
    span(nb - 1, nb, "copy")
    if not gp_in_prep:
        continue
    ph = (C.c_ulonglong * 8192)()
    assert lib.vbmc_debug_glj_phases(ph) == 0
    q = np.array(ph, dtype=np.int64).reshape(-1, 8)[K:2 * K]  # the GP blocks of an S = 1 launch
    d = np.diff(q[:, :7], axis=1) / 100.0
    print("  GP block phases (median over blocks, us): request+setup %.2f | rounds %.2f | dpp+lds %.2f | sums+put %.2f | drain %.2f | count %.2f ; whole %.2f"
          % (*np.median(d, axis=0), np.median(q[:, 6] - q[:, 0]) / 100.0))
