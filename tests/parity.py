"""Shared parity metric.

Bar (BASELINE.json north_star): 1e-3 per-channel L-inf against the reference GLSL on identical inputs.

Two facts shape how that bar can be stated honestly:

1. Storage formats.  K1's output and K3's targets are binary16.  Above 2.0 two ADJACENT half values are more than 1e-3
   apart, so any two implementations that differ by one rounding there differ by > 1e-3.  For half-stored outputs a channel
   is in tolerance when |a-b| <= 1e-3 OR a and b are the same or adjacent binary16 values.  For fp32 outputs (K2, K4) a
   channel is in tolerance when |a-b| <= 1e-3 OR |a-b| <= 1e-5*|b| (fp32 rounding noise on large radiances).  Both the
   strict absolute L-inf and the metric's verdict are REPORTED (`Report.linf_abs`).

2. Discontinuities.  The path branches on comparisons (lobe selection ssgi.frag:186, hit tests :463/:493, step(1e-4,w)
   poisson_denoise.frag:120, ...) and addresses NEAREST texels at computed coordinates.  Two correct implementations whose
   exp/log/sin differ in the last ulp take different sides in a few pixels and differ there by O(1).  Those pixels are not
   assumed, they are PROVEN, two independent ways (oracle/rfx_oracle.c): (a) the oracle records for every fragment how close
   its closest decision was ("discontinuity margins": margin < 1 = within reach of ulp-level operand differences); (b) the
   oracle re-evaluates the stage with its exp/log/pow/sqrt/sin/cos results perturbed within the reference GL's MEASURED error
   ("perturbed primitives") — a fragment whose output then moves by more than the tolerance is unstable, which also catches
   ill-conditioned expressions that are not branches.  An out-of-tolerance pixel is *explained* when (a) or (b) holds and
   *unexplained* otherwise.  Tests assert `unexplained == 0`, bound the explained flips, and print the at-risk population
   (explainable pixels among ALL pixels) so the reader can see the criteria discriminate.
"""
from dataclasses import dataclass

import numpy as np

ATOL = 1e-3
RTOL_F32 = 1e-5


def _half_ulp_distance(a, b):
    """ordered-integer distance between the binary16 values of two float arrays that hold half-representable numbers"""
    ha = np.asarray(a, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
    hb = np.asarray(b, np.float32).astype(np.float16).view(np.int16).astype(np.int32)
    ha = np.where(ha < 0, -32768 - ha, ha)
    hb = np.where(hb < 0, -32768 - hb, hb)
    return np.abs(ha - hb)


@dataclass
class Report:
    name: str
    pixels: int
    linf_abs: float          # true per-channel L-inf over ALL pixels, absolute
    linf_abs_ok: float       # the same over pixels that are in tolerance
    bad: int                 # pixels with a channel out of tolerance
    explained: int           # ... of which the oracle proves unstable (margin < 1, or output moves under perturbed primitives)
    unexplained: int
    at_risk: int             # explainable pixels among ALL pixels (estimated from a random sample; None without an oracle run)
    worst_unexplained: tuple

    def line(self):
        head = "%-22s px %9d  Linf(all) %.3e  Linf(in-tol) %.3e  out-of-tol %6d (%.4f%%)" % (
            self.name, self.pixels, self.linf_abs, self.linf_abs_ok, self.bad, 100.0 * self.bad / max(self.pixels, 1))
        if self.at_risk is None:
            return head + "  (bounded, not proven: no oracle run for this comparison)"
        return head + " = explained %d + UNEXPLAINED %d, at-risk %d (%.4f%%)" % (
            self.explained, self.unexplained, self.at_risk, 100.0 * self.at_risk / max(self.pixels, 1))


def _errors(a, b, half, tol_scale=1.0):
    """(per-channel abs error with NaN/inf conventions applied, (H, W) bool out-of-tolerance) — float32 throughout (8K frames).
    tol_scale < 1 tightens the absolute and relative bounds (binary16 neighbours still count as equal: a value cannot sit next to both of
    its rounding boundaries)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    if a.ndim == 2:
        a, b = a[..., None], b[..., None]
    with np.errstate(invalid="ignore", over="ignore"):
        err = np.abs(a - b)
        special = ~np.isfinite(err)
        if special.any():  # NaN == NaN and inf == inf of the same sign agree; NaN vs number / inf vs number do not
            na, nb = np.isnan(a), np.isnan(b)
            same = (na & nb) | (np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b)))
            err = np.where(same, np.float32(0.0), np.where(special, np.float32(np.inf), err))
        ok = err <= np.float32(ATOL * tol_scale)
        if half:
            rest = ~ok
            if rest.any():  # adjacent binary16 values: only looked at where the absolute rule failed
                ok[rest] = _half_ulp_distance(a[rest], b[rest]) <= 1
        else:
            ok |= err <= np.float32(RTOL_F32 * tol_scale) * np.abs(b)
    return err, ~ok.all(axis=-1)


def out_of_tolerance(a, b, half, tol_scale=1.0):
    """(H, W) bool: some channel of the pixel is outside the metric"""
    return _errors(a, b, half, tol_scale)[1]


# The instability test of the flip proofs asks whether perturbing the oracle's primitives within the reference GL's measured error moves a
# pixel by HALF the tolerance: the two implementations being compared each sit somewhere inside that error model, possibly on opposite
# sides of the unperturbed evaluation, so what separates them can be twice what separates either from it.
UNSTABLE_TOL_SCALE = 0.5


def strict(name, a, b, explainable=None, half=False, ignore=None):
    """a: implementation under test, b: reference; (H, W, C) float arrays.  explainable: (H, W) bool from the oracle (margin < 1 or
    unstable under perturbed primitives) or None.  ignore: optional (H, W) bool mask of pixels excluded from the comparison."""
    err, badpx = _errors(a, b, half)
    if ignore is not None:
        badpx &= ~ignore
        err = np.where(ignore[..., None], 0.0, err)
    perpx = err.max(axis=-1)
    n = int(badpx.size if ignore is None else (~ignore).sum())
    nbad = int(badpx.sum())
    if explainable is not None:
        expl = badpx & explainable
        unex = badpx & ~explainable
        at_risk = int((explainable & (True if ignore is None else ~ignore)).sum())
    else:
        expl = np.zeros_like(badpx)
        unex = badpx
        at_risk = None
    worst = ()
    if unex.any():
        idx = np.argwhere(unex)
        k = np.argmax(perpx[unex])
        y, x = idx[k]
        worst = (int(y), int(x), float(perpx[y, x]))
    finite = np.isfinite(perpx)
    return Report(name, n, float(perpx[finite].max()) if finite.any() else 0.0, float(perpx[~badpx].max()) if (~badpx).any() else 0.0, nbad,
                  int(expl.sum()), int(unex.sum()), at_risk, worst)


# ---- the round-1 metric (relative above 1.0, flips bounded as a fraction); kept for the many stage tests that use it
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
