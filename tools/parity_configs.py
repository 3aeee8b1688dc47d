#!/usr/bin/env python3
"""Strict stage-wise parity report against the reference GLSL on llvmpipe (tests/stagewise.py).

    python tools/parity_configs.py --impl hip|oracle --size 1920x1080 --steps 20 --refine 5 --it 1 --frames 2 [--out report.txt]

--impl hip    : the product (librfx_hip.so) on cuda:0 — run on the GPU box (uses oracle/_ref/shaders, no /root/reference needed)
--impl oracle : the C restatement (CPU) — pins the oracle itself
Prints one line per stage and frame: true L-inf, out-of-tolerance pixels, explained / UNEXPLAINED, at-risk population.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stagewise as S  # noqa: E402
from rfx_amd.context import load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_frame_parallel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", default="hip")
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--refine", type=int, default=5)
ap.add_argument("--it", type=int, default=1)
ap.add_argument("--frames", type=int, default=2)
ap.add_argument("--out", default=None)
ap.add_argument("--compare-from", type=int, default=0, help="frames before this one only advance the reference chain (no comparison)")
ap.add_argument("--perturb", type=int, default=16, help="perturbed oracle re-evaluations per stage (of the out-of-tolerance + sampled pixels)")
ap.add_argument("--uv-model", default="reference_gl", help="ideal | reference_gl (rfx_set_uv_model / rfxo_set_uv_model on the implementation and the proving oracle)")
a = ap.parse_args()
W, H = [int(v) for v in a.size.split("x")]
lines = []


def log(s):
    print(s, flush=True)
    lines.append(s)


t0 = time.time()
log("# %s vs reference GLSL on llvmpipe, stage-wise on identical inputs: %dx%d steps %d/%d denoiseIterations %d, %d frames, vUv model %s" % (
    a.impl, W, H, a.steps, a.refine, a.it, a.frames, a.uv_model))
frames = {}


def frame_fn(i):
    if i not in frames:
        frames.clear()  # one dump resident at a time (8K: 1.9 GB)
        frames[i] = synthetic_frame_parallel(W, H, i)
    return frames[i]


reports = S.run(S.HipStages if a.impl == "hip" else S.OracleStages, W, H, a.steps, a.refine, a.it, a.frames, load_blue_noise_table(), frame_fn, log=log, n_perturb=a.perturb, compare_from=a.compare_from, uv_model=a.uv_model)
log("# summary (all frames)   kind: pixels, Linf(all), Linf(in-tol), out-of-tol, explained, UNEXPLAINED, at-risk")
for kind, v in S.summarize(reports).items():
    log("#   %-18s %10d  %.3e  %.3e  %7d  %7d  %7d  %8d" % ((kind,) + tuple(v)))
import chain  # noqa: E402
log("# %s; LP_NUM_THREADS=%s; %.0f s" % (chain.GL.info(), os.environ.get("LP_NUM_THREADS"), time.time() - t0))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write("\n".join(lines) + "\n")
bad = sum(r.unexplained for r in reports)
sys.exit(1 if bad else 0)
