"""Build-authored stand-in for the third-party `gpyreg` package.  TEST INFRASTRUCTURE.

gpyreg is not installed in the build container and is not part of
/root/reference.  This stand-in exposes just enough of its class surface for
the reference's hot-path modules to import and run inside
oracle/make_golden.py; all arithmetic is delegated to oracle/gp_ref.py (the
restated gpyreg boundary).  It is never imported by the
product (pyvbmc_amd/) and never travels as part of a parity claim other than
through the golden vectors it helps generate.
"""
import numpy as np

from oracle import gp_ref

from . import covariance_functions, mean_functions, noise_functions, slice_sample  # noqa: F401


class GP:
    def __init__(self, D, covariance, mean, noise):
        self.D = D
        self.covariance = covariance
        self.mean = mean
        self.noise = noise
        self.X = None
        self.y = None
        self.s2 = None
        self.posteriors = None
        self.temporary_data = {}

    def _mean_kind(self):
        return self.mean.kind

    def update(self, X_new=None, y_new=None, s2_new=None, hyp=None, compute_posterior=True):
        if X_new is not None:
            self.X = np.atleast_2d(np.asarray(X_new, dtype=np.float64))
            self.y = np.asarray(y_new, dtype=np.float64).reshape(-1, 1)
            self.s2 = s2_new
        hyp = np.atleast_2d(np.asarray(hyp, dtype=np.float64))
        posts = [
            gp_ref.make_posterior(
                h, self.X, self.y, self._mean_kind(), self.s2, self.noise.user_provided_add
            )
            for h in hyp
        ]
        arr = np.empty(len(posts), dtype=object)
        for i, p in enumerate(posts):
            arr[i] = p
        self.posteriors = arr

    def _as_data(self):
        return gp_ref.GPData(
            self.D, self.X, self.y, self.s2, self._mean_kind(), list(self.posteriors),
            self.noise.user_provided_add,
        )

    def predict(self, x_star, y_star=None, s2_star=0, add_noise=False, separate_samples=False):
        return gp_ref.predict(self._as_data(), x_star, s2_star, add_noise, separate_samples)
