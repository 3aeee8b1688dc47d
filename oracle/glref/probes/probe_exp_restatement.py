"""Probe (needs oracle/_ref/libglref.so): is the reference GL's exp() a function we can restate BIT FOR BIT?

llvmpipe (Mesa 23.2 gallivm, lp_bld_arit.c — an absent third-party dependency of the reference's execution here; its published algorithm):
    exp(x)  = exp2(x * 1.4426950408889634)
    exp2(x) : x clamped to [-126.99999, 128]; ipart = floor(x), fpart = x - ipart; 2^ipart by exponent bits;
              2^fpart by the degree-5 polynomial {1, 0.693153073200168932794, 0.240153617044375388211, 0.0558263180532956664775,
              0.00898934009049466391101, 0.00187757667519147912699} evaluated as even(x^2) + x * odd(x^2) with fused multiply-adds.
If the restatement below equals the GL on every probed input, the oracle can evaluate `cs = 1 - exp(-t^2 / 4)` of the march (ssgi.frag:453-454)
exactly as the reference GL does — which is what tools/open_pixel.py uses to root-cause the one pinned "open pixel" of the 16-frame sequence.

    python oracle/glref/probes/probe_exp_restatement.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chain import FMT_RGBA32F, GL, Program, Tex  # noqa: E402

f32 = np.float32
C = [f32(c) for c in (1.000000000000000000000, 0.693153073200168932794, 0.240153617044375388211, 0.0558263180532956664775, 0.00898934009049466391101, 0.00187757667519147912699)]


def fma(a, b, c):  # one rounding: exact in float64 for float32 operands (24 + 24 bits of product, then one add: 53 bits suffice except for rare double roundings)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def gl_exp2(x):
    x = np.minimum(f32(128.0), x)
    x = np.maximum(f32(-126.99999), x)
    ip = np.floor(x)
    fp = (x - ip).astype(f32)
    x2 = (fp * fp).astype(f32)
    even = fma(x2, fma(x2, np.full_like(fp, C[4]), np.full_like(fp, C[2])), np.full_like(fp, C[0]))
    odd = fma(x2, fma(x2, np.full_like(fp, C[5]), np.full_like(fp, C[3])), np.full_like(fp, C[1]))
    poly = fma(odd, fp, even)
    scale = ((ip.astype(np.int32) + 127) << 23).view(f32)
    return (scale * poly).astype(f32)


def gl_exp(x):
    return gl_exp2((f32(1.4426950408889634074) * x.astype(f32)).astype(f32))


N = 4096
HEAD = "#version 300 es\nprecision highp float;\nprecision highp int;\nin vec2 vUv;\nout vec4 o;\nuniform vec2 range;\n"


def run(expr, lo, hi):
    p = Program(HEAD + "void main(){ float x = mix(range.x, range.y, vUv.x); o = vec4(x, %s, 0., 1.); }" % expr)
    t = Tex(N, 1, FMT_RGBA32F)
    p.set("range", [lo, hi])
    p.draw([t])
    r = t.read()[0]
    return r[:, 0].astype(f32), r[:, 1].astype(f32)


if __name__ == "__main__":
    print(GL.info())
    bad = 0
    for name, expr, lo, hi, fn in (("exp(x)", "exp(x)", -12.0, 3.0, gl_exp), ("exp2(x)", "exp2(x)", -20.0, 5.0, gl_exp2),
                                   ("exp(-0.25*x*x)", "exp(-0.25*x*x)", 0.5, 9.0, lambda x: gl_exp((f32(-0.25) * x * x).astype(f32)))):
        for k in range(8):  # eight interleaved grids of 4096 inputs
            x, y = run(expr, lo + k * 1e-3, hi + k * 1e-3)
            w = fn(x)
            n = int((y.view(np.uint32) != w.view(np.uint32)).sum())
            bad += n
            if n:
                i = np.flatnonzero(y.view(np.uint32) != w.view(np.uint32))[:3]
                print("  %s grid %d: %d of %d differ, e.g. x=%r gl=%r restated=%r" % (name, k, n, N, x[i].tolist(), y[i].tolist(), w[i].tolist()))
        print("%-18s [%g, %g]: restatement == GL on %d inputs%s" % (name, lo, hi, 8 * N, "" if not bad else "  (differences so far: %d)" % bad))
    print("BIT-IDENTICAL" if bad == 0 else "NOT identical: %d inputs differ" % bad)
