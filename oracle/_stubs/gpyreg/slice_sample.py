"""Stand-in (import-only)."""


class SliceSampler:
    def __init__(self, *a, **k):
        raise NotImplementedError("slice sampling is outside the hot path")
