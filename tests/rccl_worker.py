"""Worker of tests/test_rccl_multi_gpu.py: one rank of a REAL multi-GPU run (one process per GPU,
RCCL inside libvbmc_hip.so).  Launched with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the
environment, like `python -m torch.distributed.run` does; torch is not imported.

Every rank evaluates the sharded objective (`_neg_elcbo`, value + gradient, Philox draws keyed by the
global row index, so the job's value does not depend on the sharding), the stand-alone entropy and
25 iterations of the device-resident Adam loop, and writes what it got; the test compares every rank
with the oracle on the whole job."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from pyvbmc_amd import VariationalPosterior, _lib, comm, entmc_vbmc, synthetic  # noqa: E402
from pyvbmc_amd import gp as gpm  # noqa: E402
from pyvbmc_amd.minimize_adam import minimize_adam_elbo  # noqa: E402
from pyvbmc_amd.variational_optimization import _neg_elcbo  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    rank, world, local_rank = comm.env_rank_world()
    ctx = _lib.Context(local_rank)
    _lib.set_default_context(ctx)
    comm.init_from_env(ctx)
    assert ctx.comm_info() == (rank, world), (ctx.comm_info(), rank, world)
    wl = synthetic.make_workload(2, Ns_total=20 * 4000)  # D=6, K=20, N=200, 4000 samples per component

    def mk():
        vp = VariationalPosterior(wl.D, wl.K)
        vp.mu, vp.sigma, vp.lambd = wl.mu.copy(), wl.sigma.reshape(1, -1), wl.lambd.reshape(-1, 1)
        vp.w, vp.eta = wl.w.reshape(1, -1), wl.eta.reshape(1, -1)
        return vp

    gp = gpm.GP(wl.D, gpm.SquaredExponential(), gpm.NegativeQuadratic(), gpm.GaussianNoise(constant_add=True))
    gp.update(X_new=wl.X, y_new=wl.y, hyp=wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    F, dF, G, H, _ = _neg_elcbo(wl.theta.copy(), gp, mk(), 0.0, wl.NsK, True, False, bnd, rng="philox", seed=4242)
    F2 = _neg_elcbo(wl.theta.copy(), gp, mk(), 0.0, wl.NsK, True, False, bnd, rng="philox", seed=4243)[0]  # consecutive seed
    He, dHe = entmc_vbmc(mk(), wl.NsK, rng="philox", seed=4242)
    ad = minimize_adam_elbo(wl.theta.copy(), gp, mk(), wl.NsK, bnd, max_iter=25, seed=5, rng="philox",
                            use_early_stopping=False)
    np.savez(out_dir / f"rank{rank}.npz", F=F, dF=dF, G=G, H=H, F2=F2, He=He, dHe=dHe, x_tab=ad[2], y_tab=ad[3])
    ctx.comm_barrier()
    ctx.close()


if __name__ == "__main__":
    main()
