"""The driver's bench contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line with the agreed keys.  Runs the real script on the GPU with a few steps and a small CPU
sample."""
import json
import pathlib
import subprocess
import sys

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
    assert d["timed_steps"] % 8 == 0 and d["timed_region_s"] >= 0.4 and d["comm_world"] == 1
    assert 0 < d["roofline_e2e"]["frac"] < d["roofline"]["frac"] and d["roofline"]["traffic_from"]["file"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["kernel_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and c["sample"] and c["unit"]
