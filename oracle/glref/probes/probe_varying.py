"""Probe (needs oracle/_ref/libglref.so): how far is the reference GL's interpolated vUv from (i + 0.5) / n?  The rasteriser evaluates the
varying from plane equations in fp32; every fetch the shaders make at vUv (LINEAR ones turn the offset into a weight error) inherits
the difference.  The perturbation model of the parity proofs (oracle/rfx_oracle.c pert_uv) uses the worst case measured here.

Second half: the planes themselves.  The full-screen triangle (-1,-1) (3,-1) (-1,3) is outside Mesa's guard band, the draw module clips it
to the viewport, and llvmpipe rasterises two triangles split along the frame diagonal, each with its own a0 / dadx / dady (lp_setup_coef)
evaluated by fma on the integer pixel position (lp_bld_interp).  rfx_oracle.frag_uv(W, H, "reference") restates them; this probe counts
the fragments where the restatement and the GL differ (0 on every size).

    python oracle/glref/probes/probe_varying.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import rfx_oracle as O  # noqa: E402
from chain import FMT_RGBA32F, GL, Program, Tex  # noqa: E402

SRC = "#version 300 es\nprecision highp float;\nin vec2 vUv;\nout vec4 o;\nvoid main(){ o = vec4(vUv, 0., 1.); }"

if __name__ == "__main__":
    print(GL.info())
    p = Program(SRC)
    for W, H in ((128, 72), (97, 55), (55, 97), (1920, 1080), (3840, 2160), (7680, 4320)):
        t = Tex(W, H, FMT_RGBA32F)
        p.draw([t])
        r = t.read()
        for axis, n, got in (("u", W, r[H // 3, :, 0]), ("v", H, r[:, W // 3, 1])):
            want = ((np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n)).astype(np.float32)
            ulp = np.spacing(want)
            d = (got.astype(np.float64) - want.astype(np.float64)) / ulp
            print("%5dx%-5d %s: max |vUv - (i+.5)/n| = %.1f ulp (%.2e texel), mean %.2f ulp, exact on %.0f %% of the positions" % (
                W, H, axis, np.abs(d).max(), (np.abs(got.astype(np.float64) - want) * n).max(), np.abs(d).mean(), 100 * (d == 0).mean()))
        mu, mv = O.frag_uv(W, H, "reference")
        print("%5dx%-5d plane-equation model vs the GL: %d u and %d v fragments differ" % (W, H, int((mu != r[..., 0]).sum()), int((mv != r[..., 1]).sum())))
        t.free()
