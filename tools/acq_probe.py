import sys, time
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, synthetic, acquisition
from pyvbmc_amd import gp as gpm
ctx = _lib.Context(0); _lib.set_default_context(ctx)
wl = synthetic.make_workload(3, S=8)
D, K = wl.D, wl.K
vp = VariationalPosterior(D, K)
vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
g = gpm.GP(D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
g.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
rng = np.random.default_rng(0)
state = dict(integer_vars=None, lb_eps_orig=wl.X.min(0) - 2.0, ub_eps_orig=wl.X.max(0) + 2.0,
             gp_length_scale=np.exp(wl.hyp[0, :D]), variance_regularized_acq_fcn=True, tol_gp_var=1e-4)
flog = SimpleNamespace(y_max=float(np.max(wl.y)))
fn = acquisition.AcqFcnLog()
for M in (1, 16, 64, 8192):
    comp = rng.integers(0, K, size=M)
    Xs = wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, D))
    for _ in range(5): fn(Xs.copy(), g, vp, flog, state)
    ts = []
    for _ in range(200 if M < 1000 else 20):
        t0 = time.perf_counter(); a = fn(Xs, g, vp, flog, state); ts.append(time.perf_counter() - t0)
    print(f"M={M}: {np.median(ts)*1e6:.1f} us per call")
# the C entry point alone (points already float64, no Python wrapper work)
import ctypes as C
for M in (1, 16, 64):
    comp = rng.integers(0, K, size=M)
    Xs = np.ascontiguousarray(wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, D)))
    acq = np.empty(M)
    args = (ctx._h, M, _lib.ptr(Xs), 1, float(flog.y_max), 1e-4, None, _lib.ptr(acq), None, None)
    for _ in range(5): ctx._lib.vbmc_acq_eval(*args)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); ctx._lib.vbmc_acq_eval(*args); ts.append(time.perf_counter() - t0)
    print(f"M={M}: vbmc_acq_eval alone {np.median(ts)*1e6:.1f} us per call")
