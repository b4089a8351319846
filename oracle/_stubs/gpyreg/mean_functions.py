"""Stand-in mean-function classes (see package docstring)."""
from oracle import gp_ref


class ZeroMean:
    kind = gp_ref.MEAN_ZERO

    def hyperparameter_count(self, D):
        return 0


class ConstantMean:
    kind = gp_ref.MEAN_CONST

    def hyperparameter_count(self, D):
        return 1


class NegativeQuadratic:
    kind = gp_ref.MEAN_NEGQUAD

    def hyperparameter_count(self, D):
        return 1 + 2 * D
