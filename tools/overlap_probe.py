#!/usr/bin/env python3
"""Development probe (GPU box): does the NEXT frame's ray march (rfx_ssgi_trace: reads only that frame's depth / G-buffer) fill the issue slots the
current frame's K2-K4 leave idle?  Two contexts on one GPU, each with its own stream: A draws K2, K3 x 2, K4 of a frame, B traces.  Prints the wall
time of N rounds of each alone and of both enqueued together (host clock around enqueue + sync; N large enough that launch cost does not matter)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.scene import synthetic_frame

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    abi.set_library_path(sys.argv[i + 1])
    del sys.argv[i:i + 2]
W, H = int(sys.argv[1]), int(sys.argv[2])
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
import pickle
_cache = "/tmp/rfx_frame_%dx%d.pkl" % (W, H)
f = pickle.load(open(_cache, "rb")) if os.path.exists(_cache) else synthetic_frame(W, H, 1)
cam = abi.Camera.from_scene(f.camera); pc = abi.Camera.from_scene(f.prev_camera)
sp = abi.SsgiParams(camera=cam, steps=20, refineSteps=5, mode=0, useDirectLight=1, rayDistance=10, thickness=10, envBlur=0.5, blueNoiseIndex=77)
tp = abi.TemporalParams(camera=cam, prevCamera=pc, textureCount=2, inputType=0, logTransform=1, fullAccumulate=0, confidencePower=0.75,
                        neighborhoodClampIntensity=0.5, maxBlend=1.0, keepData=1.0)
tp.reprojectSpecular[:] = [0, 1]; tp.neighborhoodClamp[:] = [0, 1]
dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=2, blueNoiseIndex=5,
                       inputIsTemporal=1, writeToB=0, halfStoreRTZ=1)
dp.isTextureSpecular[:] = [0, 1]
cp = abi.ComposeParams(camera=cam, inputType=0)
A, B = Context(W, H), Context(W, H)
for c in (A, B):
    c.upload_frame(f)

def rest(c):  # K2, K3 x 2, K4
    c.temporal_reproject(tp)
    dp.inputIsTemporal, dp.writeToB = 1, 0; c.poisson_denoise(dp)
    dp.inputIsTemporal, dp.writeToB = 0, 1; c.poisson_denoise(dp)
    c.compose(cp)

for _ in range(3):
    A.ssgi_march(sp); rest(A)
    B.ssgi_trace(sp); B.ssgi_shade(sp)
A.sync(); B.sync()

def wall(fn, syncs):
    for s in syncs: s.sync()
    t = time.perf_counter()
    fn()
    for s in syncs: s.sync()
    return (time.perf_counter() - t) * 1e3 / N

def spin():  # the device at its sustained clock
    for _ in range(int(os.environ.get('RFX_SPIN', '100'))):
        A.ssgi_march(sp); rest(A)
    A.sync()

spin()
t_frame = wall(lambda: [(A.ssgi_march(sp), rest(A)) for _ in range(N)], [A])
t_rest = wall(lambda: [rest(A) for _ in range(N)], [A])
t_march = wall(lambda: [A.ssgi_march(sp) for _ in range(N)], [A])
t_trace = wall(lambda: [B.ssgi_trace(sp) for _ in range(N)], [B])
t_ts = wall(lambda: [(B.ssgi_trace(sp), B.ssgi_shade(sp)) for _ in range(N)], [B])
t_both = wall(lambda: [(rest(A), B.ssgi_trace(sp)) for _ in range(N)], [A, B])
t_both2 = wall(lambda: [(B.ssgi_trace(sp), rest(A)) for _ in range(N)], [A, B])
print("%dx%d, %d rounds, ms per round" % (W, H, N))
print("frame (march + K2-K4, one stream)      %.4f" % t_frame)
print("K2-K4 alone                            %.4f" % t_rest)
print("march alone (fused K1)                 %.4f" % t_march)
print("trace alone                            %.4f   trace + shade %.4f  (shade %.4f)" % (t_trace, t_ts, t_ts - t_trace))
print("K2-K4 on A || trace on B               %.4f / %.4f (enqueue order A,B / B,A)   sum of the two alone %.4f" % (t_both, t_both2, t_rest + t_trace))
print("pipelined frame estimate: max(overlapped) + shade = %.4f against %.4f today" % (min(t_both, t_both2) + (t_ts - t_trace), t_frame))
