#!/usr/bin/env python3
"""CPU (needs the reference GL, oracle/_ref/libglref.so): what the flips of K3's first pass that survive the reference vUv model are made of.

K3 rotates its eight Poisson taps by the per-pixel angle blueNoise.r * 2 pi (poisson_denoise.frag:183-189) and fetches depth / normal /
inputs NEAREST at the rotated positions: a tap within |offset| * eps of a texel boundary flips with an error eps of sin / cos.
(1) the C restatement against the golden K3 pass-0 targets under the reference vUv, its sin / cos / sqrt results perturbed by
    +-(2.4e-7 relative + `abs`), its exp / log results by +-`rel`: flips follow the ANGLE error and ignore the exp / log error entirely;
(2) the angle takes 256 values (an 8-bit blue-noise channel): the reference GL's sin / cos of those against libm's and against the
    correctly rounded values.

    python tools/k3_rotation_sensitivity.py        (output: profiles/r02_parity/k3_rotation_sensitivity.txt)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", os.path.join("oracle", "glref"), "realism-effects_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import golden_util as G  # noqa: E402
import rfx_oracle as O  # noqa: E402
import test_oracle_vs_golden as T  # noqa: E402
from chain import FMT_RGBA32F, Program, Tex  # noqa: E402
from parity import out_of_tolerance  # noqa: E402
from rfx_amd.context import load_blue_noise_table  # noqa: E402

blue = load_blue_noise_table()
g = G.load("chain_160x90_s20r5_it1")
W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
z16 = np.zeros((H, W, 4), np.uint16)


def h8(o):
    return O.half_bits_to_float(np.ascontiguousarray(o).view(np.uint16))


def flips(seed, rel, abs_):
    n = 0
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        dp = T.stage_params(g, fi, 0.0 if fi == 0 else 1.0)[2]
        A = [np.ascontiguousarray(g[kp + "A%d" % j]).copy() if fi else z16.copy() for j in range(2)]
        Tin = [np.ascontiguousarray(g[k + "temporal%d" % j]) for j in range(2)]
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][0]), 1, 0
        if seed:
            with O.perturbation(seed, rel, abs_):
                O.denoise(f.depth, f.gbuffer, Tin[0], Tin[1], blue, dp, A[0], A[1])
        else:
            O.denoise(f.depth, f.gbuffer, Tin[0], Tin[1], blue, dp, A[0], A[1])
        n += sum(int(out_of_tolerance(h8(A[j]), h8(g[k + "A%d" % j]), True).sum()) for j in range(2))
    return n


print("# (1) K3 pass 0 of tests/golden/chain_160x90_s20r5_it1 (%d pixel-textures), reference vUv: out-of-tolerance pixels of the C restatement" % (nf * 2 * W * H))
with O.uv_model("reference"):
    print("unperturbed                                   %d" % flips(0, 0, 0))
    for abs_ in (0.0, 2e-7, 1e-6, 4e-6, 1.6e-5):
        print("sin/cos/sqrt +-(2.4e-7 rel + %.1e abs), exp/log exact  seeds 1-3: %s" % (abs_, [flips(s, 0.0, abs_) for s in (1, 2, 3)]))
    for rel in (1e-6, 6e-6, 2.4e-5):
        print("exp/log +-%.1e rel, sin/cos/sqrt +-2.4e-7 rel           seeds 1-3: %s" % (rel, [flips(s, rel, 0.0) for s in (1, 2, 3)]))

print("# (2) sin / cos of the 256 possible rotation angles k / 255 * 2 pi on the reference GL")
p = Program("#version 300 es\nprecision highp float;\nin vec2 vUv;\nout vec4 o;\nvoid main(){ float k = floor(gl_FragCoord.x); float r = k / 255.;"
            " float a = r * 2. * 3.141592653589793; o = vec4(sin(a), cos(a), a, r); }")
t = Tex(256, 1, FMT_RGBA32F)
p.draw([t])
gl = t.read()[0]
f32 = np.float32
a = ((np.arange(256, dtype=f32) / f32(255)) * f32(2)) * f32(3.141592653589793)
assert (gl[:, 2] == a).all()
libm = ctypes.CDLL("libm.so.6")
libm.sinf.restype = libm.cosf.restype = ctypes.c_float
libm.sinf.argtypes = libm.cosf.argtypes = [ctypes.c_float]
ls, lc = np.array([libm.sinf(float(x)) for x in a], f32), np.array([libm.cosf(float(x)) for x in a], f32)
cs, cc = np.sin(a.astype(np.float64)).astype(f32), np.cos(a.astype(np.float64)).astype(f32)
for nm, s_, c_ in (("libm sinf / cosf", ls, lc), ("correctly rounded", cs, cc)):
    print("reference GL vs %-18s sin differs at %3d, cos at %3d of 256 angles, by at most %.2e / %.2e" % (
        nm + ":", int((gl[:, 0] != s_).sum()), int((gl[:, 1] != c_).sum()), np.abs(gl[:, 0] - s_).max(), np.abs(gl[:, 1] - c_).max()))
