"""BASELINE configs 3, 4 and 5 at JOB size on one GPU, through virtual ranks (tests/vrank_worker.py).

What runs here that ran nowhere before round 3: Philox draws with ``row_begin > 0`` and
``rows < n_half`` (the kernels' global row keys, csrc/entropy_ws.hip, and the generator's
component-boundary walk, csrc/philox.h gen_slice_block), entropy grids of several rounds
(config 4: 79 chunks x 50 components = 3 950 workgroups; config 5: the 1-wave/SIMD ``<20,25>``
build at 40 000 samples per component), the fused step's collective branch on a slice, and the
optimiser loop's slice generator.

Assertions: the slices' raw accumulators add up to the un-sharded launch (<= 1e-12 of each block's
scale: only the summation order differs), both draw forms agree, armed / ahead evaluations on a
slice equal a cold evaluation bit for bit, and the finalised values equal the oracle
(`entropy_ref.entmc` on `philox_ref.eps_half`, reference entropy/entmc_vbmc.py:61-112) at
<= 1e-10 (H) / 1e-9 (gradients; value-only at the two job sizes where the NumPy oracle with
gradients would take minutes).
"""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import vrank_worker as vw
from helpers import rel_err

from oracle import entropy_ref, mixture_ref, philox_ref
from pyvbmc_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SEED = 31337


@pytest.fixture(scope="module")
def ctx():
    from pyvbmc_amd import _lib

    c = _lib.Context(0)
    _lib.set_default_context(c)
    yield c
    _lib.set_default_context(None)
    c.close()


def blocks(D, K):
    return [(0, 1), (1, 1 + D * K), (1 + D * K, 1 + D * K + K), (1 + D * K + K, 1 + D * K + K + D),
            (1 + D * K + K + D, 1 + D * K + 2 * K + D)]


def additive_err(total, parts, D, K):
    """max over the blocks [H | mu | sigma | lambda | w] of |sum(parts) - total| / max|total| of the block"""
    s = np.sum(parts, axis=0)
    return max(float(np.max(np.abs(s[a:b] - total[a:b])) / max(np.max(np.abs(total[a:b])), 1e-300))
               for a, b in blocks(D, K))


_oracle_cache = {}


def oracle_entropy(cfg, seed, grad):
    """H (and dH at config 3) of the whole job on the restated generator's draws; per process cache."""
    key = (cfg, seed, grad)
    if not grad and (cfg, seed, True) in _oracle_cache:
        key = (cfg, seed, True)
    if key not in _oracle_cache:
        wl = synthetic.make_workload(cfg, Ns_total=vw.JOB_NS[cfg])
        mix = mixture_ref.Mixture.make(wl.mu, wl.sigma, wl.lambd, wl.w, wl.eta)
        h = wl.NsK // 2
        if grad:
            eps = philox_ref.eps_half(wl.K, h, wl.D, seed)
            _oracle_cache[key] = entropy_ref.entmc(mix, wl.NsK, (True,) * 4, True, eps_half=eps)
        else:
            # value only, in row blocks (the partial sums are additive): bounded memory at 4e8 pairs
            H = 0.0
            step = 10_000
            for r0 in range(0, h, step):
                n = min(step, h - r0)
                eps = philox_ref.eps_half(wl.K, h, wl.D, seed, r0, n)
                H += entropy_ref.entmc_partial(mix, eps, wl.NsK, (False,) * 4)["H"]
            _oracle_cache[key] = (H, None)
    return _oracle_cache[key]


@pytest.mark.parametrize("inline", [False, True], ids=["pregen", "inline"])
@pytest.mark.parametrize("cfg,W", [(3, 2), (3, 8), (4, 2), (4, 8), (5, 2), (5, 8)])
def test_entmc_virtual_ranks(ctx, cfg, W, inline):
    r = vw.run_entmc(ctx, cfg, W, SEED + cfg, inline)
    wl = synthetic.make_workload(cfg, Ns_total=vw.JOB_NS[cfg])
    D, K = wl.D, wl.K
    plan = r["plan"]
    ws_form = "mfma" if (cfg == 5 and not inline) else "ws"  # D = 20, K = 100 with resident draws: entropy_mfma.hip
    assert plan["kernel"] == ws_form and plan["resident_draws"] == (not inline), plan
    if cfg == 4:
        # the job size: 1 250 batches per component -- parts of well over a hundred batches in span mode (several
        # grid rounds of 16-batch workgroups on equal chunks)
        assert wl.NsK == 160_000 and (plan["rg"] > 100 if plan["span"] else (plan["chunks"] * K > 512 and plan["rg"] == 16)), plan
    if cfg == 5:
        assert (D, K, wl.NsK) == (20, 100, 40_000) and (plan["span"] or plan["chunks"] * K > 512), plan
    for p in r["plans"]:
        assert p["kernel"] == ws_form and p["resident_draws"] == (not inline), p
    err = additive_err(r["raw"], r["parts"], D, K)
    Ho, dHo = oracle_entropy(cfg, SEED + cfg, grad=(cfg == 3))
    print(f"cfg {cfg} W={W} {'inline' if inline else 'pregen'}: plan {plan}; |sum_r raw_r - raw| {err:.2e}; "
          f"H rel {abs(r['H'] - Ho) / abs(Ho):.2e}" + ("" if dHo is None else f"; dH rel {rel_err(r['dH'], dHo):.2e}"))
    assert err <= 1e-12
    assert abs(r["H"] - Ho) <= 1e-10 * abs(Ho)
    # the summed slices, finalised like the job's vector
    from pyvbmc_amd import _lib
    import ctypes as C

    tot = np.ascontiguousarray(np.sum(r["parts"], axis=0))
    Hs, dHs = C.c_double(), np.empty(r["dH"].size)
    ctx.check(ctx._lib.vbmc_entmc_finalize(ctx._h, _lib.ptr(tot), 15, 1, C.byref(Hs), _lib.ptr(dHs)))
    assert abs(Hs.value - Ho) <= 1e-10 * abs(Ho)
    assert rel_err(dHs, r["dH"]) <= 1e-11
    if dHo is not None:
        assert rel_err(r["dH"], dHo) < 1e-9 and rel_err(dHs, dHo) < 1e-9
        # the entropy kernels' exp2 is a degree-8 polynomial (1.07e-12 pointwise, csrc/fastmath.h): its error averages
        # out over the 5e7 densities of an evaluation -- far inside the parity bar (measured: H 1.2e-15, dH 1.6e-15)
        assert abs(r["H"] - Ho) <= 1e-12 * abs(Ho) and rel_err(r["dH"], dHo) < 1e-11


def test_draw_forms_agree_bitwise(ctx):
    """The same slice through the draws generated ahead into HBM and through the in-line generator:
    identical draws, identical kernels' arithmetic -> identical raw vectors (config 4, rank 5 of 8)."""
    a = vw.run_entmc(ctx, 4, 8, SEED, False)
    b = vw.run_entmc(ctx, 4, 8, SEED, True)
    assert np.array_equal(a["parts"], b["parts"]) and np.array_equal(a["raw"], b["raw"])


def _subprocess_worker(tmp_path, cfg, W, seed, what, extra=()):
    out = tmp_path / f"{what}_{cfg}_{W}.npz"
    env = dict(os.environ, VBMC_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "vrank_worker.py"), str(out), str(cfg), str(W), str(seed),
                        what, *map(str, extra)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return dict(np.load(out))


@pytest.mark.parametrize("cfg,W", [(3, 2), (4, 8), (5, 8), (5, 2)])
def test_fused_step_virtual_ranks(ctx, tmp_path, cfg, W):
    """`vbmc_neg_elcbo` on every virtual rank's slice: armed evaluations and draws generated ahead with
    row_begin > 0, against a cold evaluation (bit-identical), the un-sharded step (additive) and the
    oracle; then the same through the collective branch (1-rank communicator, VBMC_FORCE_COLLECTIVE)."""
    seed = SEED + cfg - 2  # its last seed (seed + 2) is the stand-alone test's: one oracle run serves both
    r = vw.run_elbo(ctx, cfg, W, seed)
    wl = synthetic.make_workload(cfg, Ns_total=vw.JOB_NS[cfg])
    D, K = wl.D, wl.K
    assert r["plan"]["kernel"] == ("mfma" if cfg == 5 else "ws")
    if cfg in (4, 5):
        assert r["plan"]["span"] or r["plan"]["chunks"] * K > 512, r["plan"]
    assert np.array_equal(r["parts"], r["cold"])  # armed + ahead-generated draws == cold evaluation
    err = additive_err(r["raw"], r["parts"], D, K)
    Ho, _ = oracle_entropy(cfg, seed + 2, grad=False)
    print(f"fused cfg {cfg} W={W}: |sum_r raw_r - raw| {err:.2e}; H rel {abs(r['H'] - Ho) / abs(Ho):.2e}; "
          f"sum H_r rel {abs(r['H_parts'].sum() - Ho) / abs(Ho):.2e}")
    assert err <= 1e-12
    assert abs(r["H"] - Ho) <= 1e-10 * abs(Ho) and abs(r["H_parts"].sum() - Ho) <= 1e-10 * abs(Ho)
    assert abs(r["raw"][0] - r["H"]) <= 1e-15 * abs(r["H"])
    # F_r = -G - H_r + bounds: the slices' objective values differ from the job's by the other slices' entropy
    assert np.allclose(r["F_parts"] + r["H_parts"], r["F"] + r["H"], rtol=1e-12, atol=0)
    if (cfg, W) in ((4, 8), (5, 2)):
        c = _subprocess_worker(tmp_path, cfg, W, seed, "elbo")
        assert np.array_equal(c["parts"], r["parts"]) and np.array_equal(c["raw"], r["raw"])
        assert np.array_equal(c["cold"], r["cold"]) and c["F"] == r["F"] and np.array_equal(c["dF"], r["dF"])


@pytest.mark.parametrize("cfg,W,ns_total", [(2, 2, None), (3, 8, None), (3, 2, 50 * 28)])
def test_adam_loop_virtual_ranks(ctx, tmp_path, cfg, W, ns_total):
    """The optimiser loop's draws come from spare workgroups of its short launches (GenSlice of the
    slice); with a zero step size every iteration evaluates the same mixture on seed + i: the slices'
    H_tab add up to the un-sharded loop's, iteration by iteration, and equal the oracle's."""
    seed = SEED + 100 * cfg
    r = vw.run_adam(ctx, cfg, W, seed, ns_total=ns_total)
    err = np.max(np.abs(r["H_parts"].sum(axis=0) - r["H_full"]) / np.abs(r["H_full"]))
    wl = synthetic.make_workload(cfg, Ns_total=vw.JOB_NS[cfg])
    mix = mixture_ref.Mixture.make(wl.mu, wl.sigma, wl.lambd, wl.w, wl.eta)
    nsk = int(r["nsk"])
    its = (0, len(r["H_full"]) - 1) if cfg == 3 and ns_total is None else range(len(r["H_full"]))
    for i in its:
        eps = philox_ref.eps_half(wl.K, nsk // 2, wl.D, seed + i)
        Ho = entropy_ref.entmc(mix, nsk, (False,) * 4, True, eps_half=eps)[0]
        assert abs(r["H_full"][i] - Ho) <= 1e-10 * abs(Ho), (i, r["H_full"][i], Ho)
    print(f"adam cfg {cfg} W={W} nsk={nsk}: max |sum_r H_r - H| / |H| over {len(r['H_full'])} iterations {err:.2e}")
    assert err <= 1e-12
    if cfg == 2:
        c = _subprocess_worker(tmp_path, cfg, W, seed, "adam")
        assert np.array_equal(c["H_parts"], r["H_parts"]) and np.array_equal(c["H_full"], r["H_full"])
