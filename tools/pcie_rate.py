#!/usr/bin/env python3
"""PCIe-inclusive rate of the chain at 4K (GPU box): every frame's dump planes (431 MB: depth 33 + gbuffer / velocity / direct 3 x 133)
cross the host link.  Three ways:
  sync     : rfx_upload from pageable memory in front of the draws (round 1: 10.1 ms/frame)
  stream   : rfx_stage_upload from PINNED planes (rfx_host_alloc) on the upload stream + rfx_stage_flip: frame n+1 crosses PCIe while
             frame n is drawn (rfx.h "streaming dumps")
  resident : no upload at all (bench.py's `value`)
and the bare copy time of one frame's planes (the PCIe bound)."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
import numpy as np
from rfx_amd import abi
from rfx_amd.context import Context
from rfx_amd.effect import SSGIEffect
from rfx_amd.scene import synthetic_frame_parallel

W, H = 3840, 2160
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = synthetic_frame_parallel(W, H, 1)
mb = (f.depth.nbytes + f.gbuffer.nbytes + f.velocity.nbytes + f.direct.nbytes) / 1e6
PLANES = (("depth", abi.TEX_DEPTH), ("gbuffer", abi.TEX_GBUFFER), ("velocity", abi.TEX_VELOCITY), ("direct", abi.TEX_DIRECT_LIGHT))


def chain(ctx, frame):
    scene = types.SimpleNamespace(frame=frame)
    cam = types.SimpleNamespace(**vars(f.camera))
    return scene, SSGIEffect(None, scene, cam, dict(width=W, height=H), seeds=dict(ssgi=1, denoise=2), half_store_rtz=True)


def run(label, per_frame, ctx):
    per_frame(); per_frame(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        per_frame()
    ctx.sync()
    dt = (time.perf_counter() - t0) / n
    print("%-9s %7.2f ms/frame  %7.0f Mpix/s" % (label, dt * 1e3, W * H / dt / 1e6), flush=True)
    return dt


# ---- resident
ctx = Context(W, H)
fr = types.SimpleNamespace(depth=f.depth, gbuffer=f.gbuffer, velocity=f.velocity, direct=f.direct, camera=f.camera, static=True)
scene, fx = chain(ctx, fr)
t_res = run("resident", lambda: fx.update(ctx, None), ctx)
ctx.close()

# ---- synchronous uploads from pageable memory (every update() re-sends the dump: static is not set)
ctx = Context(W, H)
fr = types.SimpleNamespace(depth=f.depth, gbuffer=f.gbuffer, velocity=f.velocity, direct=f.direct, camera=f.camera)
scene, fx = chain(ctx, fr)
t_sync = run("sync", lambda: fx.update(ctx, None), ctx)
ctx.close()

# ---- streaming: two sets of pinned planes (a reader thread would fill them from disk), stage n+1, draw n, flip
ctx = Context(W, H)
sets = []
for _ in range(2):
    s = {}
    for name, tex in PLANES:
        src = getattr(f, name)
        s[name] = ctx.host_alloc(src.shape, src.dtype)
        s[name][...] = src
    sets.append(types.SimpleNamespace(camera=f.camera, static="resident", **s))
scene, fx = chain(ctx, sets[0])
# bare copy time of one frame (the PCIe bound of the streaming form)
ctx.stage_frame(sets[0]); ctx.stage_flip(); ctx.sync()
t0 = time.perf_counter()
for i in range(5):
    ctx.stage_frame(sets[i & 1]); ctx.stage_flip()
ctx.sync()
t_copy = (time.perf_counter() - t0) / 5
print("copy only %7.2f ms/frame  = %.1f GB/s pinned host -> device" % (t_copy * 1e3, mb / t_copy / 1e3), flush=True)
state = {"i": 0}


def stream_frame():
    i = state["i"]
    ctx.stage_frame(sets[(i + 1) & 1])     # frame n+1 starts crossing PCIe ...
    scene.frame = sets[i & 1]
    fx.update(ctx, None)                   # ... while frame n is drawn (its planes were published by the last flip)
    ctx.stage_flip()
    state["i"] = i + 1


t_stream = run("stream", stream_frame, ctx)
print("stream / PCIe bound = %.2f   (sync %.2fx slower than stream; resident %.2f ms)" % (t_copy / t_stream, t_sync / t_stream, t_res * 1e3))
# same pixels as the resident run
import hashlib
print("compose sha1", hashlib.sha1(ctx.download(abi.TEX_COMPOSE).tobytes()).hexdigest()[:16], "halo violations", ctx.halo_violations())
ctx.close()
