"""World-size-2 CPU test of the sharded entropy path over gloo.

Each rank owns its even share of every component's antithetic-pair rows
(comm.shard_rows), produces the raw accumulator of ITS rows, the vectors are
summed with one all-reduce, and every rank finalises (vbmc_entmc_finalize, the
product's host code).  On the GPU the per-rank accumulator comes from the HIP
kernels and the all-reduce is RCCL inside libvbmc_hip.so; here, without GPUs,
the per-rank accumulator comes from the oracle and the all-reduce from gloo --
which checks the partition, the additivity and the finalisation, not the kernels."""
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


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_entropy_matches_reference(tmp_path, world):
    import subprocess

    port = _free_port()
    worker = str(ROOT / "tests" / "gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
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


def test_stale_rendezvous_file_is_not_joined(tmp_path, monkeypatch):
    """A leftover id file of an earlier launch that reused the key (same port, run id, launcher pid)
    must not be joined: wrong nonce, or written long before this process started."""
    import os
    import time

    import pytest

    from pyvbmc_amd import comm

    monkeypatch.setenv("VBMC_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29556")
    monkeypatch.setenv("VBMC_LAUNCH_NONCE", "launch-A")
    comm.exchange_unique_id(0, 2, lambda: bytes(128))  # launch A's file stays behind (it "crashed")
    monkeypatch.setenv("VBMC_LAUNCH_NONCE", "launch-B")
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, None, timeout=0.3)
    uid = comm.exchange_unique_id(0, 2, lambda: bytes(range(128)))  # launch B's rank 0 replaces it
    assert comm.exchange_unique_id(1, 2, None, timeout=5) == uid
    # same nonce, but the file is far older than this process
    path = comm._rendezvous_path()
    old = time.time() - 3600
    os.utime(path, (old, old))
    with pytest.raises(TimeoutError):
        comm.exchange_unique_id(1, 2, None, timeout=0.3)
