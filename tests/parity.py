"""Shared parity metric.

Tolerance (BASELINE.json north_star): 1e-3 per-channel L-inf.  Values above 1 are compared
relatively (|a-b| <= 1e-3*max(1,|b|)): K1's output is stored as binary16, whose ulp already
exceeds 1e-3 above 2.0, so an absolute 1e-3 is unattainable there for ANY two implementations
that are not bit-identical in their transcendental functions (SURVEY.md §7 "hard parts").

The path contains hard discontinuities (lobe selection `random.b < diffW`, ray hit tests,
nearest-texel boundaries of the rotated Poisson taps, `step(1e-4, w)`): a 1-ulp difference in
exp/log/sin between two correct implementations flips a branch for a few pixels and changes
them by O(1).  Those pixels are counted separately and bounded as a FRACTION of the frame.
"""
import numpy as np

ATOL = 1e-3


def compare(a, b):
    """returns (fraction of pixels with any channel out of tolerance, max error among in-tolerance channels)"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    tol = ATOL * np.maximum(1.0, np.abs(b))
    bad = (err > tol) | (np.isnan(a) != np.isnan(b))
    badpx = bad.any(axis=-1) if bad.ndim == 3 else bad
    ok = err[~bad]
    return float(badpx.mean()), float(ok.max()) if ok.size else 0.0


def assert_close(name, a, b, max_flip_fraction):
    frac, mx = compare(a, b)
    assert frac <= max_flip_fraction, "%s: %.4f%% of pixels outside 1e-3 (allowed %.4f%%), in-tolerance max err %.2e" % (
        name, 100 * frac, 100 * max_flip_fraction, mx)
    return frac, mx
