#!/usr/bin/env python3
"""PCIe-inclusive frame rate: upload a fresh 4K dump (depth+gbuffer+velocity+direct = 431 MB) from host memory
every frame, then run the chain.  bench.py's `value` excludes the upload."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
import numpy as np
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.effect import SSGIEffect
from rfx_amd.scene import AnalyticScene
W, H = 3840, 2160
f = AnalyticScene(1234).render(W, H, 1)
ctx = Context(W, H)
scene = types.SimpleNamespace(frame=f); cam = types.SimpleNamespace(**vars(f.camera))
fx = SSGIEffect(None, scene, cam, dict(width=W, height=H), seeds=dict(ssgi=1, denoise=2))
fx.update(ctx, None); ctx.sync()
n = 10
t0 = time.perf_counter()
for i in range(n):
    # a NEW ndarray object per frame defeats the resident-plane cache: every plane crosses PCIe again
    scene.frame = types.SimpleNamespace(depth=f.depth.view(), gbuffer=f.gbuffer.view(), velocity=f.velocity.view(), direct=f.direct.view(), camera=f.camera)
    fx.update(ctx, None)
ctx.sync()
dt = (time.perf_counter() - t0) / n
mb = (f.depth.nbytes + f.gbuffer.nbytes + f.velocity.nbytes + f.direct.nbytes) / 1e6
print("PCIe-inclusive: %.2f ms/frame (%.0f Mpix/s); upload %.0f MB/frame from pageable host memory -> %.1f GB/s effective incl. compute" % (dt * 1e3, W * H / dt / 1e6, mb, mb / dt / 1e3))
