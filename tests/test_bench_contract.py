"""The driver's bench contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line with the agreed keys.  Runs the real script on the GPU with a few steps and a small CPU
sample."""
import json
import pathlib
import subprocess
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2",
                        "--cpu-sample-nsk", "200", "--cpu-reps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["unit"] == "evals/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "D=10" in d["metric"] and "K=50" in d["metric"] and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]  # Ns_job = 1e6
    assert d["timed_steps"] % 8 == 0 and d["timed_region_s"] >= 5.0 and d["comm_world"] == 1
    assert 0 < d["roofline_e2e"]["frac"] < d["roofline"]["frac"] and d["roofline"]["traffic_from"]["file"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["unit"]
    rs = d["reference_stream"]
    assert rs["ms_per_eval_p10"] <= rs["ms_per_eval_p50"] <= rs["ms_per_eval_p90"]


def _run(*args, timeout=900):
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, cwd=str(ROOT))


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` outside a launcher starts the N ranks itself (--spawn forces that
    path at N = 1): still ONE JSON line, from rank 0, with the communicator's own world size."""
    p = _run("--gpus", "1", "--spawn", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-secondary",
             "--min-timed-s", "0.3")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["comm_world"] == 1 and d["value"] > 0 and "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,extra", [(4, ()), (5, ("--job",))])
def test_bench_job_size_lines(cfg, extra):
    """BASELINE configs 4 and 5 at job size on one GPU (secondary lines): several grid rounds of the
    entropy kernel, F finite, roofline fields filled."""
    p = _run("--gpus", "1", "--config", str(cfg), *extra, "--steps", "4", "--warmup", "1", "--no-cpu-baseline",
             "--no-secondary", "--min-timed-s", "0.2")
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    K = 50 if cfg == 4 else 100
    el = d["config"]["entropy_launch"]
    assert (el["span"] and el["rg"] > 16 or el["chunks"] * K > 512) and d["scaling"] == "strong"  # many batches per workgroup / several grid rounds
    assert f"Ns={'8e+06' if cfg == 4 else '4e+06'}" in d["metric"] and 0.3 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
@pytest.mark.parametrize("args", [("--spawn", "--scaling", "strong"), ("--spawn", "--config", "4"), ("--spawn",)],
                         ids=["strong", "config4-job", "weak"])
def test_bench_lines_through_the_collective_branch(args):
    """What the driver's multi-GPU runs execute, on the one GPU there is: the ranks started by bench.py itself, a real
    RCCL communicator of one rank and -- VBMC_FORCE_COLLECTIVE=1 -- the multi-rank step (finish -> all-reduce ->
    publish, never armed) for the weak line, the strong line and config 4's job: each must print ONE schema-valid
    JSON line, not a traceback."""
    import os

    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *args, "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-secondary", "--min-timed-s", "0.3"], capture_output=True, text=True,
                       timeout=900, cwd=str(ROOT), env=dict(os.environ, VBMC_FORCE_COLLECTIVE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["comm_world"] == 1 and d["value"] > 0 and np.isfinite(d["F"])
    assert d["scaling"] == ("strong" if "--config" in args else "weak")  # (one rank: the two readings coincide)
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] in ("hbm", "mfma")


def test_bench_more_gpus_than_visible_fails_cleanly():
    """`--gpus N` with fewer than N devices: one clear line on stderr, rc != 0, nothing on stdout."""
    sys.path.insert(0, str(ROOT))
    from pyvbmc_amd import _lib

    n = _lib.device_count() + 1
    if n == 1:
        n = 2  # (N = 1 does not go through the spawner)
    p = _run("--gpus", str(n), timeout=120)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert f"--gpus {n} needs {n} visible GPUs" in p.stderr and len(p.stderr.strip().splitlines()) == 1
