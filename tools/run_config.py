#!/usr/bin/env python3
"""Run one of BASELINE.json's configs end-to-end through SSGIEffect on one GPU and report ms/frame.
   python tools/run_config.py <W> <H> <steps> <refineSteps> <denoiseIterations> <frames>"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
import numpy as np
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.effect import SSGIEffect
from rfx_amd.scene import AnalyticScene

W, H, steps, refine, it, nf = [int(a) for a in sys.argv[1:7]]
gen = AnalyticScene(1234)
t = time.time(); frames = [gen.render(W, H, i) for i in range(min(nf, 2))]
for _f in frames: _f.static = True  # resident dump: uploads stay out of the timed region
print("dump gen %.1fs" % (time.time() - t), flush=True)
ctx = Context(W, H)
scene = types.SimpleNamespace(frame=frames[0]); cam = types.SimpleNamespace(**vars(frames[0].camera))
fx = SSGIEffect(None, scene, cam, dict(width=W, height=H, steps=steps, refineSteps=refine, denoiseIterations=it), seeds=dict(ssgi=1, denoise=2))
def frame(i):
    f = frames[min(i, len(frames) - 1)]
    scene.frame = f
    for k, v in vars(f.camera).items(): setattr(cam, k, v)
    fx.update(ctx, None)
frame(0); frame(1); ctx.sync()
ctx.time_begin()
t = time.perf_counter()
for i in range(nf): frame(1)
host_ms = (time.perf_counter() - t) / nf * 1e3  # what the host needs to ISSUE a frame (the calls are asynchronous): above the device time = host-bound
ms = ctx.time_end() / nf
print("host issue time %.3f ms/frame" % host_ms)
out = ctx.download(abi.TEX_COMPOSE)
print("%dx%d steps %d/%d it %d: %.3f ms/frame  %.1f Mpix/s   compose finite=%s mean=%.4f  halo_violations=%d" % (
    W, H, steps, refine, it, ms, W * H / ms / 1e3, bool(np.isfinite(out).all()), float(out[..., :3].mean()), ctx.halo_violations()))
