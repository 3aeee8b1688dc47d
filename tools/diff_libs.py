#!/usr/bin/env python3
"""Development helper (GPU box): the same frames through TWO builds of the library, stage by stage on identical inputs — how many texels
of every target differ, and by how much.  Used to state what an optimisation did to the bits (an exact rewrite: 0 texels).

    python tools/diff_libs.py <lib_a.so> <lib_b.so> [WxH] [frames]

Each stage of lib B is fed lib A's previous-stage outputs, so a difference never compounds."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import stagewise as S  # noqa: E402  (paths)
from rfx_amd import abi  # noqa: E402
from rfx_amd.context import load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_frame_parallel  # noqa: E402

lib_a, lib_b = sys.argv[1], sys.argv[2]
W, H = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "3840x2160").split("x"))
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
blue = load_blue_noise_table()


def stages(path):
    abi.set_library_path(path)
    return S.HipStages(W, H, blue)


# two libraries in one process: both export the same symbols, each CDLL keeps its own (RTLD_LOCAL) — load A, build its stages, then B
a = stages(lib_a)
b = stages(lib_b)


def as_f(x):  # half-stored targets (RGBA16F texels, K1's eight packed halfs) decoded to float32
    x = np.ascontiguousarray(x)
    if x.dtype in (np.uint16, np.uint32):
        return x.view(np.float16).astype(np.float32).reshape(x.shape[0], x.shape[1], -1)
    return x


def report(name, xa, xb):
    xa, xb = (xa if isinstance(xa, (list, tuple)) else [xa]), (xb if isinstance(xb, (list, tuple)) else [xb])
    for j, (p, q) in enumerate(zip(xa, xb)):
        p, q = np.ascontiguousarray(p), np.ascontiguousarray(q)
        neq = (p.view(np.uint8).reshape(H, W, -1) != q.view(np.uint8).reshape(H, W, -1)).any(-1)
        n = int(neq.sum())
        line = "%-14s tex%d  differing texels %8d of %d (%.5f %%)" % (name, j, n, H * W, 100.0 * n / (H * W))
        if n:
            fa, fb = as_f(p)[neq].astype(np.float64), as_f(q)[neq].astype(np.float64)
            d = np.abs(fa - fb)
            rel = d / np.maximum(np.abs(fa), 1e-30)
            line += "   max |a-b| %.3e   max rel %.3e" % (np.nanmax(d), np.nanmax(rel))
        print(line, flush=True)


z16, zf = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.float32)
hist, B, T = zf.copy(), [z16.copy(), z16.copy()], [zf.copy(), zf.copy()]
prev_cam, keep = None, 0.0
for fi in range(frames):
    f = synthetic_frame_parallel(W, H, fi)
    a.frame(f)
    b.frame(f)
    sp, tp, dp, cp = S.stage_params(f.camera, prev_cam or f.camera, keep, 20, 5)
    sp.blueNoiseIndex = 1001 + fi
    k1a, k1b = a.ssgi(hist, sp), b.ssgi(hist, sp)
    report("f%d K1" % fi, k1a, k1b)
    ta, tb = a.temporal(k1a, B, T, tp), b.temporal(k1a, B, T, tp)
    report("f%d K2" % fi, ta, tb)
    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2001 + 2 * fi, 1, 0
    aa, ab = a.denoise(ta, [z16.copy(), z16.copy()], dp), b.denoise(ta, [z16.copy(), z16.copy()], dp)
    report("f%d K3 pass0" % fi, aa, ab)
    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2002 + 2 * fi, 0, 1
    ba, bb = a.denoise(aa, B, dp), b.denoise(aa, B, dp)
    report("f%d K3 pass1" % fi, ba, bb)
    ca, cb = a.compose(ba, hist, cp), b.compose(ba, hist, cp)
    report("f%d K4" % fi, ca, cb)
    hist, B, T = ca, ba, ta
    prev_cam, keep = f.camera, 1.0
a.close()
b.close()
