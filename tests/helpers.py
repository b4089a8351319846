"""Shared builders: golden arrays -> oracle objects (test side only)."""
import numpy as np

from oracle import gp_ref, mixture_ref


def oracle_mix(g):
    return mixture_ref.Mixture.make(g["mu"], g["sigma"], g["lambd"], g["w"], g["eta"])


def oracle_gp(g, hyp=None):
    s2 = g["s2"] if g["s2"].size else None
    hyp = g["hyp"] if hyp is None else hyp
    return gp_ref.make_gp(g["X"], g["y"], hyp, gp_ref.MEAN_NEGQUAD, s2=s2, noise_user=s2 is not None)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / den)
