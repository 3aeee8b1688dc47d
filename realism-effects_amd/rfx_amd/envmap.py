"""Host side of the environment map's importance sampling: the CPU pass of the reference's
src/ssgi/utils/EquirectHdrInfoUniform.js (`gatherData`, :148-245, run there in a Web Worker), which turns the equirect map into the
two inverse-CDF tables `sampleEquirectProbability` (ssgi_utils.frag:210-225) reads, plus the luminance sum.

JavaScript arithmetic is reproduced as it happens there: scalars are doubles, every store into a Float32Array rounds to float32."""
from __future__ import annotations

import numpy as np


def _closest_index(cdf: np.ndarray, targets: np.ndarray) -> np.ndarray:
    """binarySearchFindClosestIndexOf (:130-146): first index whose value is not below the target, capped at the last element."""
    return np.minimum(np.searchsorted(cdf.astype(np.float64), targets, side="left"), len(cdf) - 1)


def build_importance(data: np.ndarray, flip_y: bool = False):
    """data: (H, W, 4) float32 texels as stored (row 0 first).  Returns (marginalWeights float32[H], conditionalWeights float32[H, W],
    totalSum float) — `marginalWeights` becomes an H x 1 texture, `conditionalWeights` a W x H one (updateFrom :383-389)."""
    data = np.asarray(data, np.float32)
    if flip_y:  # :151-166 — the loop copies row y into row h-y while walking y upwards, so the upper half is overwritten before it
        h = data.shape[0] - 1  # is read: the result is the lower half mirrored, NOT a flipped image (reproduced, not corrected)
        data = data.copy()
        for y in range(h + 1):
            data[h - y] = data[y]
    H, W = data.shape[:2]
    d = data.astype(np.float64)
    weight = 0.2126 * d[..., 0] + 0.7152 * d[..., 1] + 0.0722 * d[..., 2]   # colorToLuminance :124-127, in doubles
    pdf_c = weight.astype(np.float32)                                         # pdfConditional[i] = weight
    run = np.cumsum(weight, axis=1)                                           # cumulativeRowWeight, double, left to right
    cdf_c = run.astype(np.float32)
    row = run[:, -1]
    nz = row != 0
    pdf_c[nz] = (pdf_c[nz].astype(np.float64) / row[nz, None]).astype(np.float32)  # /= cumulativeRowWeight (f32 read, double divide, f32 store)
    cdf_c[nz] = (cdf_c[nz].astype(np.float64) / row[nz, None]).astype(np.float32)
    # totalSumValue accumulates pixel by pixel in row-major order; cumulativeWeightMarginal row by row
    total = float(np.cumsum(weight.reshape(-1))[-1]) if weight.size else 0.0
    marg_run = np.cumsum(row)
    cdf_m = marg_run.astype(np.float32)
    if marg_run[-1] != 0:
        cdf_m = (cdf_m.astype(np.float64) / marg_run[-1]).astype(np.float32)
    # inverse CDFs, half-texel centred (:229-243)
    marginal = ((_closest_index(cdf_m, (np.arange(H) + 1) / H) + 0.5) / H).astype(np.float32)
    tx = (np.arange(W) + 1) / W
    conditional = np.empty((H, W), np.float32)
    for y in range(H):
        conditional[y] = ((_closest_index(cdf_c[y], tx) + 0.5) / W).astype(np.float32)
    return marginal, conditional, total
