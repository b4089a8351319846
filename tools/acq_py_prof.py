import sys, time
sys.path.insert(0, "/root/repo")
from types import SimpleNamespace
import numpy as np
from pyvbmc_amd import VariationalPosterior, _lib, synthetic, acquisition
from pyvbmc_amd import gp as gpm
from pyvbmc_amd._duck import ctx_of, upload_vp
from pyvbmc_amd.gp import upload_gp
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
M = 16
comp = rng.integers(0, K, size=M)
Xs = wl.mu.T[comp] + 1.5 * wl.lambd * wl.sigma[comp, None] * rng.standard_normal((M, D))
fn(Xs, g, vp, flog, state)
def t(f, n=2000):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
print("upload_vp", t(lambda: upload_vp(vp, ctx)))
print("upload_gp", t(lambda: upload_gp(g, ctx)))
print("upload_gp lazy", t(lambda: upload_gp(g, ctx, lazy=True)))
print("real2int", t(lambda: fn._real2int(Xs, vp.parameter_transformer, None)))
print("inverse", t(lambda: vp.parameter_transformer.inverse(Xs)))
Xo = vp.parameter_transformer.inverse(Xs)
print("bounds mask", t(lambda: np.logical_or(np.any(Xo < state["lb_eps_orig"], axis=1), np.any(Xo > state["ub_eps_orig"], axis=1))))
print("asarray+f64", t(lambda: _lib.f64(np.asarray(Xs, dtype=np.float64))))
print("ctx_of", t(lambda: ctx_of(vp)))
