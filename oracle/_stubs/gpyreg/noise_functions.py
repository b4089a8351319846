"""Stand-in noise class (see package docstring)."""


class GaussianNoise:
    def __init__(self, constant_add=False, user_provided_add=False,
                 scale_user_provided=False, rectified_linear_output_dependent_add=False):
        self.constant_add = constant_add
        self.user_provided_add = user_provided_add

    def hyperparameter_count(self):
        return 1
