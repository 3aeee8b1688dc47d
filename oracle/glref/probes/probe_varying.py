"""Probe (needs oracle/_ref/libglref.so): how far is the reference GL's interpolated vUv from (i + 0.5) / n?  The rasteriser evaluates the
varying from plane equations in fp32; every fetch the shaders make at vUv (LINEAR ones turn the offset into a weight error) inherits
the difference.  The perturbation model of the parity proofs (oracle/rfx_oracle.c pert_uv) uses the worst case measured here.

    python oracle/glref/probes/probe_varying.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chain import FMT_RGBA32F, GL, Program, Tex  # noqa: E402

SRC = "#version 300 es\nprecision highp float;\nin vec2 vUv;\nout vec4 o;\nvoid main(){ o = vec4(vUv, 0., 1.); }"

if __name__ == "__main__":
    print(GL.info())
    p = Program(SRC)
    for W, H in ((128, 72), (1920, 1080), (3840, 2160), (7680, 4320)):
        t = Tex(W, H, FMT_RGBA32F)
        p.draw([t])
        r = t.read()
        for axis, n, got in (("u", W, r[H // 3, :, 0]), ("v", H, r[:, W // 3, 1])):
            want = ((np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n)).astype(np.float32)
            ulp = np.spacing(want)
            d = (got.astype(np.float64) - want.astype(np.float64)) / ulp
            print("%5dx%-5d %s: max |vUv - (i+.5)/n| = %.1f ulp (%.2e texel), mean %.2f ulp, exact on %.0f %% of the positions" % (
                W, H, axis, np.abs(d).max(), (np.abs(got.astype(np.float64) - want) * n).max(), np.abs(d).mean(), 100 * (d == 0).mean()))
        t.free()
