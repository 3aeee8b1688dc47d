#!/usr/bin/env python3
"""Build container (needs /root/reference + llvmpipe): root cause of the ONE pinned "open pixel" of the 16-frame sequence
(tests/test_gpu_baseline_configs.py: configs[4]'s options at 1920x1080, frame 10, K1, pixel (y 584, x 676) — the kernel equals the C restatement bit for
bit there, the reference GL reads the texel across a silhouette, 94 ulps of a coordinate away: more than any rounding model allowed).

The restatement runs the sequence as the implementation under test, stage-wise against the reference GLSL on llvmpipe, twice: with the true exp (the
GLSL's meaning; what the product computes) and with the reference GL's OWN exp restated bit for bit (rfxo_set_gl_exp: llvmpipe's degree-5 exp2
polynomial behind `cs = 1 - exp(-t^2/4)` of the march, oracle/glref/probes/probe_exp_restatement.py).  If the second run has no unexplained pixel at
frame 10 and reproduces the GL's texel at (584, 676), the open pixel is llvmpipe's exp approximation moving the ray, not an unknown.

    python tools/open_pixel.py [frame] [W H]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "realism-effects_amd", os.path.join("oracle", "glref")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import rfx_oracle as O
import stagewise as S
from rfx_amd.context import load_blue_noise_table
from rfx_amd.scene import synthetic_frame_parallel

FRAME = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
blue = load_blue_noise_table()
frames = {}


def frame_fn(i):
    if i not in frames:
        frames[i] = synthetic_frame_parallel(W, H, i)
    return frames[i]


captured = {}
_strict = S.strict


def spy(name, a, b, **kw):
    r = _strict(name, a, b, **kw)
    if name.endswith("K1 ssgi"):
        captured[name] = (np.array(a, copy=True), np.array(b, copy=True))
    return r


S.strict = spy
for mode in (0, 1):
    O.lib().rfxo_set_gl_exp(mode)
    lines = []
    reports = S.run(S.OracleStages, W, H, 40, 5, 3, FRAME + 1, blue, frame_fn, log=lines.append, n_perturb=16, compare_only={FRAME})
    k1 = [r for r in reports if r.name.endswith("K1 ssgi")][0]
    got, ref = captured[k1.name]
    px = (584, 676) if (W, H) == (1920, 1080) else None
    print("exp = %s:  %s" % ("the reference GL's polynomial" if mode else "true exp (the product's meaning)", k1.line()))
    if px:
        print("    pixel (y %d, x %d): restatement %s  reference GL %s  -> %s" % (px[0], px[1], np.array2string(got[px][:8], precision=5), np.array2string(ref[px][:8], precision=5),
                                                                                "EQUAL" if np.array_equal(got[px], ref[px]) else "differ"))
    others = [r for r in reports if not r.name.endswith("K1 ssgi")]
    print("    the other %d stage outputs of frame %d: unexplained %d" % (len(others), FRAME, sum(r.unexplained for r in others)))
O.lib().rfxo_set_gl_exp(0)
