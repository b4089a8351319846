"""Worker of tests/test_sharded_gloo.py: one rank of a gloo process group (CPU).

Run as: python tests/gloo_worker.py <rank> <world> <port> <out_dir>
torch is imported BEFORE the HIP library on purpose: both bring a ROCm runtime with the
same SONAMEs, and only this load order shuts down cleanly.  The product never imports torch.
"""
import sys
from pathlib import Path

import torch.distributed as dist  # noqa: E402  (must precede pyvbmc_amd._lib)

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
from helpers import oracle_mix  # noqa: E402

from oracle import entropy_ref  # noqa: E402
from pyvbmc_amd import _lib, comm, synthetic  # noqa: E402


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    grp = comm.GlooGroup()
    g = dict(np.load(ROOT / "tests" / "golden" / "c2s.npz"))
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    r0, r1 = comm.shard_rows(NsK // 2, grp.rank, grp.world)
    mix = oracle_mix(g)
    part = entropy_ref.pack_partial(entropy_ref.entmc_partial(mix, eps[:, r0:r1, :], NsK, (True,) * 4))
    total = grp.allreduce_sum(part)
    h = _lib.Context(-1)  # host-only context: the product's finalisation code, no kernels
    h.set_mixture(g["mu"], g["sigma"], g["lambd"], g["w"], g["eta"])
    H = C.c_double()
    dH = np.empty(D * K + 2 * K + D)
    raw = _lib.f64(total)
    h.check(h._lib.vbmc_entmc_finalize(h._h, _lib.ptr(raw), 15, 1, C.byref(H), _lib.ptr(dH)))
    np.savez(Path(out_dir) / f"rank{rank}.npz", H=H.value, dH=dH, rows=np.array([r0, r1]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
