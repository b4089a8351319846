"""The entropy kernels' exp2 polynomial (csrc/fastmath.h VBMC_ENT_EXP2_COEFFS) is what tools/fit_polys.py fits:
degree 8 on |f| <= 1/2, Chebyshev-node interpolation in 60-digit arithmetic, and its error in float64 Horner
evaluation is the 1.07e-12 the header's accuracy argument starts from."""
import re
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def header_coeffs():
    text = (ROOT / "pyvbmc_amd" / "csrc" / "fastmath.h").read_text()
    n = int(re.search(r"#define VBMC_ENT_EXP2_N (\d+)", text).group(1))
    body = text[text.index("#define VBMC_ENT_EXP2_COEFFS"):]
    body = body[: body.index("}") + 1]
    vals = [float.fromhex(h) for h in re.findall(r"0x[0-9a-fA-F.]+p[-+]?\d+", body)]
    assert len(vals) == n
    return n, vals


def test_entropy_exp2_polynomial_is_the_fitted_one():
    pytest.importorskip("mpmath")
    sys.path.insert(0, str(ROOT / "tools"))
    import mpmath as mp
    from fit_polys import cheb_fit, horner64

    n, vals = header_coeffs()
    half = mp.mpf(1) / 2
    c = cheb_fit(lambda x: mp.mpf(2) ** x, -half, half, n)
    assert abs(float(c[0]) - 1.0) < 1e-15  # the kernels use the constant 1 exactly
    assert [float(x).hex() for x in c[1:]] == [v.hex() for v in vals]
    f = np.linspace(-0.5, 0.5, 400001)
    err = np.max(np.abs(horner64([1.0] + vals, f) / np.exp2(f) - 1))
    assert 5e-13 < err < 1.1e-12, err
