"""World-size-2 CPU test of the sharded entropy path over gloo.

Each rank owns its even share of every component's antithetic-pair rows
(comm.shard_rows), produces the raw accumulator of ITS rows, the vectors are
summed with one all-reduce, and every rank finalises (vbmc_entmc_finalize, the
product's host code).  On the GPU the per-rank accumulator comes from the HIP
kernels and the all-reduce is RCCL inside libvbmc_hip.so; here, without GPUs,
the per-rank accumulator comes from the oracle and the all-reduce from gloo --
which checks the partition, the additivity and the finalisation, not the kernels."""
import ctypes as C
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist

    from helpers import oracle_mix
    from oracle import entropy_ref
    from pyvbmc_amd import _lib, comm, synthetic

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    grp = comm.GlooGroup()
    g = dict(np.load(ROOT / "tests" / "golden" / "c2s.npz"))
    K, D, NsK, seed = int(g["K"]), int(g["D"]), int(g["NsK"]), int(g["seed"])
    eps = synthetic.draw_eps_half(K, D, NsK, seed)
    r0, r1 = comm.shard_rows(NsK // 2, grp.rank, grp.world)
    mix = oracle_mix(g)
    part = entropy_ref.pack_partial(entropy_ref.entmc_partial(mix, eps[:, r0:r1, :], NsK, (True,) * 4))
    total = grp.allreduce_sum(part)
    h = _lib.Context(-1)
    h.set_mixture(g["mu"], g["sigma"], g["lambd"], g["w"], g["eta"])
    H = C.c_double()
    dH = np.empty(D * K + 2 * K + D)
    raw = _lib.f64(total)
    h.check(h._lib.vbmc_entmc_finalize(h._h, _lib.ptr(raw), 15, 1, C.byref(H), _lib.ptr(dH)))
    np.savez(Path(out_dir) / f"rank{rank}.npz", H=H.value, dH=dH, rows=np.array([r0, r1]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_entropy_matches_reference(tmp_path, world):
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = dict(np.load(ROOT / "tests" / "golden" / "c2s.npz"))
    rows = []
    for r in range(world):
        o = np.load(tmp_path / f"rank{r}.npz")
        # every rank ends with the full-job value: the reference's own output
        assert abs(o["H"] - g["entmc_H_1111_1"]) <= 1e-12 * abs(g["entmc_H_1111_1"])
        assert np.max(np.abs(o["dH"] - g["entmc_dH_1111_1"])) <= 1e-11 * np.max(np.abs(g["entmc_dH_1111_1"]))
        rows.append(tuple(o["rows"]))
    # the shards tile [0, n_half) exactly once
    assert rows[0][0] == 0 and rows[-1][1] == int(g["NsK"]) // 2
    assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))


def test_shard_rows_partition():
    from pyvbmc_amd import comm

    for n in (0, 1, 7, 10000, 80001):
        for w in (1, 2, 3, 8):
            cuts = [comm.shard_rows(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    from pyvbmc_amd import comm

    monkeypatch.setenv("VBMC_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29555")
    uid = comm.exchange_unique_id(0, 2, lambda: bytes(range(128)))
    assert comm.exchange_unique_id(1, 2, None, timeout=5) == uid == bytes(range(128))
