#!/usr/bin/env python3
"""Per-kernel timing of the HIP chain on one GPU (development helper; bench.py is the contract)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
import numpy as np
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.scene import synthetic_frame

if "--lib" in sys.argv:  # a tuning variant of the library (csrc/build_variants.sh) instead of the in-tree build
    i = sys.argv.index("--lib")
    abi.set_library_path(sys.argv[i + 1])
    del sys.argv[i:i + 2]
W, H = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
only = sys.argv[4] if len(sys.argv) > 4 else ""  # e.g. "K1": time only the stages whose name starts with this
import pickle
_cache = "/tmp/rfx_frame_%dx%d.pkl" % (W, H)
t = time.time()
if os.path.exists(_cache):
    f = pickle.load(open(_cache, "rb"))
else:
    f = synthetic_frame(W, H, 1)
    try:
        pickle.dump(f, open(_cache, "wb"), protocol=4)
    except Exception as e:  # cache is a convenience only
        print("no cache:", e)
print("scene %.1fs" % (time.time() - t), flush=True)
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
stages = [("K1 ssgi", lambda: ctx.ssgi_march(sp), 68), ("K1t trace", lambda: ctx.ssgi_trace(sp), 40), ("K1s shade", lambda: ctx.ssgi_shade(sp), 92), ("K2 temporal", lambda: ctx.temporal_reproject(tp), 80), ("K3 pass0", d0, 68), ("K3 pass1", d1, 52),
          ("K4 compose", lambda: ctx.compose(cp), 52)]
# two warm frames so the history textures are populated
for _ in range(2):
    for _, fn, _b in stages: fn()
ctx.sync()
tot = 0
for name, fn, bpp in stages:
    if not name.startswith(only): continue
    ctx.time_begin()
    for _ in range(iters):
        if name.startswith("K1s"):  # every shade needs its trace: time the pair, subtract the trace measured just before
            ctx.ssgi_trace(sp)
        fn()
    ms = ctx.time_end() / iters
    if name.startswith("K1t"): trace_ms = ms
    if name.startswith("K1s"): ms -= trace_ms
    gbs = bpp * W * H / (ms * 1e-3) / 1e9
    print("%-12s %8.3f ms  %8.1f Mpix/s  algorithmic %6.1f GB/s (%.1f%% of 8 TB/s)" % (name, ms, W * H / ms / 1e3, gbs, gbs / 80), flush=True)
    if not name.startswith("K4") and not name.startswith("K1t") and not name.startswith("K1s"): tot += ms
print("K1+K2+2xK3: %.3f ms  -> %.1f Mpix/s; 268 B/px -> %.1f GB/s (%.1f%% of 8 TB/s)" % (tot, W * H / tot / 1e3, 268 * W * H / tot / 1e6, 268 * W * H / tot / 1e6 / 80))
# the whole frame back to back, as a host issues it (launch gaps and the pre-pass overlap included): hipEvents around `iters` frames
if not only:
    def frame():
        ctx.ssgi_march(sp); ctx.temporal_reproject(tp); d0(); d1(); ctx.compose(cp)
    for _ in range(3):
        frame()
    ctx.sync()
    best = 1e9
    for _rep in range(3):
        ctx.time_begin()
        for _ in range(iters):
            frame()
        best = min(best, ctx.time_end() / iters)
    print("frame (K1+K2+2xK3+K4 back to back): %.4f ms  -> %.1f Mpix/s" % (best, W * H / best / 1e3), flush=True)
print("halo violations", ctx.halo_violations())
if only.startswith("K1") or not only:  # variants of the exact kernel must not change a single texel
    import hashlib
    ctx.ssgi_march(sp)
    print("ssgi sha1", hashlib.sha1(ctx.download(abi.TEX_SSGI).tobytes()).hexdigest()[:16])
if not only or only.startswith("K3") or only.startswith("K2"):  # (a K2 / K3 variant: the same bits as the in-tree build?)
    import hashlib
    for _, fn, _b in stages: fn()
    print("temporal0 sha1", hashlib.sha1(ctx.download(abi.TEX_TEMPORAL0).tobytes()).hexdigest()[:16], "a0 sha1", hashlib.sha1(ctx.download(abi.TEX_DENOISE_A0).tobytes()).hexdigest()[:16])
    print("b0 sha1", hashlib.sha1(ctx.download(abi.TEX_DENOISE_B0).tobytes()).hexdigest()[:16], "compose sha1", hashlib.sha1(ctx.download(abi.TEX_COMPOSE).tobytes()).hexdigest()[:16])
