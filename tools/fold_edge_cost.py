#!/usr/bin/env python3
"""Development helper (GPU box, round 6): what an EXACT fold of the compose draw into the last denoise launch could save at most.

The exact fold composes in-kernel only the pixels whose LINEAR footprint of target B lies inside the workgroup's 64 x 8 output tile and leaves the
tile-edge pixels (rows y % 8 in {0, 7}: 25 % of the rows; columns x % 64 in {0, 63} of the other rows: 2.3 % of the pixels) to a second, thin compose
launch.  Measured here with what exists: the two-launch pair, the folded pair of the opt-in (inexact) fold — a lower bound of the in-kernel part,
which would add an LDS exchange and a barrier to it — and the compose draw over 25 % of the rows in ONE contiguous window — a lower bound of the thin
launch, which has to walk every fourth row pair and the strided columns.

    python tools/fold_edge_cost.py [W H]

(Run in round 6 against ABI 18, which still had the inexact fold: profiles/r06_k4/exact_fold_bounds.txt.  On ABI 19 the folded-pair line prints nan.)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.scene import synthetic_frame

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
f = synthetic_frame(W, H, 1)
ctx = Context(W, H)
ctx.upload_frame(f)
cam = abi.Camera.from_scene(f.camera); pc = abi.Camera.from_scene(f.prev_camera)
sp = abi.SsgiParams(camera=cam, steps=20, refineSteps=5, mode=0, useDirectLight=1, rayDistance=10, thickness=10, envBlur=0.5, blueNoiseIndex=77)
tp = abi.TemporalParams(camera=cam, prevCamera=pc, textureCount=2, inputType=0, logTransform=1, fullAccumulate=0, confidencePower=0.75,
                        neighborhoodClampIntensity=0.5, maxBlend=1.0, keepData=1.0)
tp.reprojectSpecular[:] = [0, 1]; tp.neighborhoodClamp[:] = [0, 1]
dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=2, blueNoiseIndex=5,
                       inputIsTemporal=1, writeToB=0, halfStoreRTZ=1)
dp.isTextureSpecular[:] = [0, 1]
cp = abi.ComposeParams(camera=cam, inputType=0)
def d0(): dp.inputIsTemporal, dp.writeToB = 1, 0; ctx.poisson_denoise(dp)
def d1(): dp.inputIsTemporal, dp.writeToB = 0, 1; ctx.poisson_denoise(dp)
for _ in range(2):
    ctx.ssgi_march(sp); ctx.temporal_reproject(tp); d0(); d1(); ctx.compose(cp)
ctx.sync()

def timed(fn, n=40):
    for _ in range(3): fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.time_begin()
        for _ in range(n): fn()
        best = min(best, ctx.time_end() / n)
    return best

t_pass = timed(d1)
t_k4 = timed(lambda: ctx.compose(cp))
t_pair = timed(lambda: (d1(), ctx.compose(cp)))
rows = (H // 4) & ~1
def k4_quarter():
    ctx.set_row_window(0, rows); ctx.compose(cp); ctx.set_row_window()
t_quarter = timed(k4_quarter)
if hasattr(ctx, "set_compose_fold"):
    ctx.set_compose_fold(True)
    t_fold = timed(lambda: (d1(), ctx.compose(cp)))
    ctx.set_compose_fold(False)
else:
    t_fold = float("nan")
print("%dx%d: later denoise pass %.4f ms, compose %.4f ms, the two back to back %.4f ms" % (W, H, t_pass, t_k4, t_pair))
print("folded pair (inexact opt-in fold: every pixel composed in-kernel from its own texel) %.4f ms -> the in-kernel part costs >= %.4f ms" % (t_fold, t_fold - t_pass))
print("compose over %d contiguous rows (25 %% of the frame) %.4f ms -> the thin launch costs >= that" % (rows, t_quarter))
print("exact fold >= %.4f ms against %.4f ms for two launches: saves at most %.4f ms of a frame" % (t_fold + t_quarter, t_pair, t_pair - t_fold - t_quarter))
