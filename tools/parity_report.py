#!/usr/bin/env python3
"""Per-stage parity report of the HIP path vs the C oracle (GPU box): fraction of pixels outside
1e-3 and the in-tolerance max error, each stage fed with the oracle's previous-stage output."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("realism-effects_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import rfx_oracle as O
from parity import compare
from rfx_amd import abi
from rfx_amd.context import Context, load_blue_noise_table
from rfx_amd.scene import synthetic_frame
from test_gpu_parity import _params

W, H = int(sys.argv[1]), int(sys.argv[2]); NF = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bn = load_blue_noise_table(); ctx = Context(W, H)
comp = np.zeros((H, W, 4), np.float32); A = [np.zeros((H, W, 4), np.uint16) for _ in range(2)]; B = [np.zeros((H, W, 4), np.uint16) for _ in range(2)]
T = [np.zeros((H, W, 4), np.float32) for _ in range(2)]; prev, keep = None, 0.0
def rep(name, a, b):
    f, m = compare(a, b); print("  %-12s %.4f%% outside 1e-3, in-tol max %.2e" % (name, 100 * f, m), flush=True)
for fi in range(NF):
    f = synthetic_frame(W, H, fi); sp, tp, dp, cp = _params(abi, f, prev or f.camera, keep); ctx.upload_frame(f); print("frame", fi)
    sp.blueNoiseIndex = 1000 + fi; ctx.upload(abi.TEX_COMPOSE, comp); ctx.ssgi_march(sp); g = ctx.download(abi.TEX_SSGI)
    o = O.ssgi(f.depth, f.gbuffer, f.direct, comp, bn, sp); ga, gb = O.unpack_ssgi(g); oa, ob = O.unpack_ssgi(o)
    rep("ssgi.diff", ga, oa); rep("ssgi.spec", gb, ob); print("  ssgi bit-identical texels: %.4f%%" % (100 * (g == o).all(axis=-1).mean()))
    ctx.upload(abi.TEX_SSGI, o); ctx.upload(abi.TEX_DENOISE_B0, B[0]); ctx.upload(abi.TEX_DENOISE_B1, B[1]); ctx.upload(abi.TEX_TEMPORAL0, T[0]); ctx.upload(abi.TEX_TEMPORAL1, T[1])
    ctx.temporal_reproject(tp); O.temporal(o, f.velocity, B[0], B[1], tp, T[0], T[1])
    rep("temporal0", ctx.download(abi.TEX_TEMPORAL0), T[0]); rep("temporal1", ctx.download(abi.TEX_TEMPORAL1), T[1]); keep, prev = 1.0, f.camera
    ctx.upload(abi.TEX_TEMPORAL0, T[0]); ctx.upload(abi.TEX_TEMPORAL1, T[1]); ctx.upload(abi.TEX_DENOISE_A0, A[0]); ctx.upload(abi.TEX_DENOISE_A1, A[1])
    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2000 + 2 * fi, 1, 0; ctx.poisson_denoise(dp); O.denoise(f.depth, f.gbuffer, T[0], T[1], bn, dp, A[0], A[1])
    rep("denoiseA0", O.half_bits_to_float(ctx.download(abi.TEX_DENOISE_A0)), O.half_bits_to_float(A[0])); rep("denoiseA1", O.half_bits_to_float(ctx.download(abi.TEX_DENOISE_A1)), O.half_bits_to_float(A[1]))
    ctx.upload(abi.TEX_DENOISE_A0, A[0]); ctx.upload(abi.TEX_DENOISE_A1, A[1])
    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2001 + 2 * fi, 0, 1; ctx.poisson_denoise(dp); O.denoise(f.depth, f.gbuffer, A[0], A[1], bn, dp, B[0], B[1])
    rep("denoiseB0", O.half_bits_to_float(ctx.download(abi.TEX_DENOISE_B0)), O.half_bits_to_float(B[0])); rep("denoiseB1", O.half_bits_to_float(ctx.download(abi.TEX_DENOISE_B1)), O.half_bits_to_float(B[1]))
    ctx.upload(abi.TEX_DENOISE_B0, B[0]); ctx.upload(abi.TEX_DENOISE_B1, B[1]); ctx.upload(abi.TEX_COMPOSE, comp); ctx.compose(cp); O.compose(f.depth, f.gbuffer, B[0], B[1], cp, comp)
    rep("compose", ctx.download(abi.TEX_COMPOSE), comp)
