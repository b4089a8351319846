"""Empty import-only stand-in (plotting / CMA-ES are outside the hot path).  TEST INFRASTRUCTURE."""
