"""INTEGRATION.md section 2's run-time patch against the actual reference (build container only:
skipped where /root/reference does not exist, e.g. on the GPU box)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/pyvbmc").exists(), reason="needs the reference checkout")
def test_runtime_patch_resolves_and_accepts_reference_objects():
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "check_integration_patch.py")], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert p.stdout.strip().endswith("OK"), p.stdout[-2000:]
    assert "AttributeError" not in p.stdout + p.stderr
