"""pyvbmc_amd -- MI355X-native (gfx950) implementation of PyVBMC's ELBO-evaluation
hot path, behind the reference's own Python names.

    from pyvbmc_amd import VariationalPosterior, entmc_vbmc, entlb_vbmc
    from pyvbmc_amd.variational_optimization import _neg_elcbo, _gp_log_joint
    pyvbmc_amd.patch()        # or: route the reference's own module through all of it (pyvbmc_amd/dropin.py)

Host code is plain Python calling hand-written HIP kernels through the C ABI of
libvbmc_hip.so (include/vbmc_hip.h) via ctypes.  No PyTorch, no CPU fallback.
"""
from .dropin import patch, unpatch  # noqa: F401  (pyvbmc_amd.patch(vo): the whole drop-in in one call)
from .entropy import entlb_vbmc, entmc_vbmc  # noqa: F401
from .variational_posterior import VariationalPosterior  # noqa: F401

__all__ = ["VariationalPosterior", "entmc_vbmc", "entlb_vbmc", "patch", "unpatch"]
