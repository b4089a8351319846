"""A REAL multi-rank run of the HIP kernels + RCCL (one process per GPU).  Needs >= 2 visible
GPUs: skipped on the single-GPU boxes this build has had so far -- it is here so that the first
multi-GPU box that runs `pytest -m gpu` checks the sharded path end to end ("nothing in
the repo has ever run vbmc_neg_elcbo with world > 1")."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _ngpu():
    from pyvbmc_amd import _lib

    return _lib.device_count()


@pytest.mark.parametrize("world", [1, 2, 4, 8])  # world 1: the worker and the checks themselves, on any GPU box
def test_sharded_objective_on_real_gpus(tmp_path, world):
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs, {_ngpu()} visible")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               TORCHELASTIC_RUN_ID=f"pytest{os.getpid()}", HSA_ENABLE_IPC_MODE_LEGACY="0",
               VBMC_LAUNCH_NONCE=os.urandom(8).hex())
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "rccl_worker.py"), str(tmp_path)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    sys.path.insert(0, str(ROOT))
    from oracle import adam_ref, elbo_ref, entropy_ref, gp_ref, mixture_ref, philox_ref
    from pyvbmc_amd import synthetic

    wl = synthetic.make_workload(2, Ns_total=20 * 4000)
    mix = mixture_ref.Mixture.make(wl.mu, wl.sigma, wl.lambd, wl.w, wl.eta)
    ogp = gp_ref.make_gp(wl.X, wl.y, wl.hyp)
    bnd = synthetic.default_theta_bnd(wl)
    eps = philox_ref.eps_half(wl.K, wl.NsK // 2, wl.D, 4242)
    Fo, dFo, Go, Ho, _ = elbo_ref.neg_elcbo(wl.theta.copy(), ogp, mix, 0.0, wl.NsK, True, False, bnd, eps_half=eps)
    Heo, dHeo = entropy_ref.entmc(mix, wl.NsK, (True,) * 4, True, eps_half=eps)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for o in outs:
        assert abs(o["F"] - Fo) <= 1e-9 * abs(Fo) and abs(o["H"] - Ho) <= 1e-9 * abs(Ho) and abs(o["G"] - Go) <= 1e-9 * abs(Go)
        assert np.max(np.abs(o["dF"] - dFo)) <= 1e-8 * np.max(np.abs(dFo))
        assert abs(o["He"] - Heo) <= 1e-9 * abs(Heo) and np.max(np.abs(o["dHe"] - dHeo)) <= 1e-8 * np.max(np.abs(dHeo))
        # every rank holds the identical job value and applies the identical optimiser update
        assert o["F"] == outs[0]["F"] and np.array_equal(o["dF"], outs[0]["dF"]) and o["F2"] == outs[0]["F2"]
        assert np.array_equal(o["x_tab"], outs[0]["x_tab"]) and np.array_equal(o["y_tab"], outs[0]["y_tab"])
    assert outs[0]["F2"] != outs[0]["F"]
    # the loop's first objective value is the oracle's on iteration 0's draws
    eps0 = philox_ref.eps_half(wl.K, wl.NsK // 2, wl.D, 5)
    F0 = elbo_ref.neg_elcbo(wl.theta.copy(), ogp, mixture_ref.Mixture.make(wl.mu, wl.sigma, wl.lambd, wl.w, wl.eta),
                            0.0, wl.NsK, True, False, bnd, eps_half=eps0)[0]
    assert abs(outs[0]["y_tab"][0] - F0) <= 1e-9 * abs(F0)
    _ = adam_ref  # (trajectory parity of the loop itself: tests/test_adam.py on one GPU)
