"""CPU oracle for the PyVBMC ELBO hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy (float64) restatement of the reference algorithm for
the one hot path this repository accelerates:

    mixture_ref.py   VariationalPosterior arithmetic: set/get_parameters,
                     pdf/log_pdf(+grad, t-tails), analytic moments
                     (reference: pyvbmc/variational_posterior/variational_posterior.py)
    entropy_ref.py   entmc_vbmc / entlb_vbmc
                     (reference: pyvbmc/entropy/entmc_vbmc.py, entlb_vbmc.py)
    gp_ref.py        the gpyreg boundary (SE-ARD kernel, mean functions,
                     posterior alpha/L/sW, predict) and _gp_log_joint
                     (reference: pyvbmc/vbmc/variational_optimization.py:1238-1606;
                     gpyreg itself is a third-party dependency that is NOT under
                     /root/reference: `gpyreg >= 0.1.0`, pyproject.toml:13 --
                     its arithmetic is restated from PyVBMC's call sites)
    elbo_ref.py      _neg_elcbo, _vp_bound_loss, _soft_bound_loss
                     (reference: pyvbmc/vbmc/variational_optimization.py:503-657,991-1235)
    philox_ref.py    counter-based normal generator the HIP kernels use in
                     device-RNG mode, restated so parity can be checked on
                     identical draws (no reference counterpart: the reference
                     uses NumPy's global MT19937 stream).

Pinning: the restatement is checked (tests/test_oracle_golden.py) against
golden vectors produced by importing the actual reference in the build
container (oracle/make_golden.py -> tests/golden/*.npz) and against the
MATLAB-derived known answers the reference's own tests assert.

Rules: only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg
may import this package.  Nothing under pyvbmc_amd/ imports it; the product
path fails loudly when the HIP library is missing instead of falling back here.
"""
