"""Probe (needs oracle/_ref/libglref.so): how accurate are the reference GL's (llvmpipe) exp / log / pow / sin / cos / sqrt / inversesqrt?
The answer sets the scales of the discontinuity margins in oracle/rfx_oracle.c (tests/parity.py): a decision whose operands went
through one of these on the reference side may flip when its relative gap is below the function's error.

    python oracle/glref/probes/probe_transcendentals.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chain import FMT_RGBA32F, GL, Program, Tex  # noqa: E402

N = 4096
HEAD = "#version 300 es\nprecision highp float;\nprecision highp int;\nin vec2 vUv;\nout vec4 o;\nuniform vec2 range;\n"


def run(expr, lo, hi):
    p = Program(HEAD + "void main(){ float x = mix(range.x, range.y, vUv.x); o = vec4(x, %s, 0., 1.); }" % expr)
    t = Tex(N, 1, FMT_RGBA32F)
    p.set("range", [lo, hi])
    p.draw([t])
    r = t.read()[0]
    return r[:, 0].astype(np.float64), r[:, 1].astype(np.float64)


def report(name, expr, lo, hi, fn, rel=True):
    x, y = run(expr, lo, hi)
    want = fn(x.astype(np.float32).astype(np.float64))
    err = np.abs(y - want)
    if rel:
        err = err / np.maximum(np.abs(want), 1e-30)
    print("%-28s x in [%g, %g]: max %s err %.3e, mean %.3e" % (name, lo, hi, "rel" if rel else "abs", err.max(), err.mean()))


if __name__ == "__main__":
    print(GL.info())
    report("exp(x)", "exp(x)", -12, 3, np.exp)
    report("exp(-0.25*x*x)", "exp(-0.25*x*x)", 0.5, 20, lambda x: np.exp(-0.25 * x * x))
    report("log(x+1)", "log(x + 1.)", 0, 50, lambda x: np.log(x + 1), rel=False)
    report("pow(x, 5)", "pow(x, 5.)", 1e-3, 1, lambda x: x ** 5)
    report("pow(x, 0.125)", "pow(x, 0.125)", 1e-3, 4, lambda x: x ** 0.125)
    report("pow(x, 0.1)", "pow(x, 0.1)", 1e-6, 1, lambda x: x ** 0.1)
    report("pow(x, 0.75)", "pow(x, 0.75)", 1e-3, 1, lambda x: x ** 0.75)
    report("sin(x)", "sin(x)", 0, 6.2832, np.sin, rel=False)
    report("cos(x)", "cos(x)", 0, 6.2832, np.cos, rel=False)
    report("sqrt(x)", "sqrt(x)", 1e-3, 100, np.sqrt)
    report("inversesqrt(x)", "inversesqrt(x)", 1e-3, 100, lambda x: 1 / np.sqrt(x))
    report("1/x", "1. / x", 1e-3, 100, lambda x: 1 / x)
    report("atan(x, 0.3)", "atan(x, 0.3)", -1, 1, lambda x: np.arctan2(x, 0.3), rel=False)
