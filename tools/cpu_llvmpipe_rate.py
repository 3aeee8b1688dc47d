#!/usr/bin/env python3
"""CPU baseline of a BASELINE.json config: the reference's own GLSL (oracle/_ref/shaders) on llvmpipe on this box's host cores —
K1 + K2 + 2*it x K3 + K4 per frame, per-draw glFinish-fenced wall time, two warm-up frames (JIT + respecialisation), median of n.
    python tools/cpu_llvmpipe_rate.py <W> <H> <steps> <refineSteps> <denoiseIterations> [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("realism-effects_amd", os.path.join("oracle", "glref")):
    sys.path.insert(0, os.path.join(ROOT, p))
cores = len(os.sched_getaffinity(0))
os.environ.setdefault("LP_NUM_THREADS", str(min(cores, 32)))  # llvmpipe caps its rasteriser threads (LP_MAX_THREADS)
import chain  # noqa: E402
from rfx_amd.context import load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_frame_parallel  # noqa: E402

W, H, steps, refine, it = [int(a) for a in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
f = synthetic_frame_parallel(W, H, 1)
c = chain.GLRefChain(W, H, load_blue_noise_table(), steps=steps, refineSteps=refine, denoiseIterations=it)
c.upload_frame(f)


def one(i):
    c.ssgi(f.camera, 100 + i)
    c.temporal(f.camera)
    c.denoise(f.camera, [200 + 2 * it * i + k for k in range(2 * it)])
    c.compose(f.camera)
    return c.ms["ssgi"] + c.ms["temporal"] + sum(c.ms["denoise"]) + c.ms["compose"]


one(0)
one(1)
ms = sorted(one(2 + i) for i in range(n))
med = ms[len(ms) // 2]
print("%dx%d steps %d/%d it %d: llvmpipe %.1f ms/frame = %.2f Mpix/s (median of %d; %s; box has %d cores, LP_NUM_THREADS=%s)" % (
    W, H, steps, refine, it, med, W * H / med / 1e3, n, chain.GL.info(), cores, os.environ["LP_NUM_THREADS"]))
