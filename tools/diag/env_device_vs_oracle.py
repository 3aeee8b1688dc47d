#!/usr/bin/env python3
"""Diagnostic (round 6): K1 with scene.environment, the device against the C restatement on identical inputs, pixel by pixel — which pixels are
outside the tolerance, are they deterministic, what do they have in common.  TEST INFRASTRUCTURE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _p in ("realism-effects_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, _p))
import rfx_oracle as O  # noqa: E402
from parity import out_of_tolerance  # noqa: E402
from rfx_amd import abi  # noqa: E402
from rfx_amd.context import Context, load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_environment, synthetic_frame  # noqa: E402
import stagewise as S  # noqa: E402

blue = load_blue_noise_table()
env_img = synthetic_environment(64, 32)
h8 = lambda t: O.half_bits_to_float(np.ascontiguousarray(t).view(np.uint16))  # noqa: E731
for (W, H), uv, blur in (((64, 72), "reference_gl", 0.5), ((64, 72), "ideal", 0.5), ((127, 7), "reference_gl", 0.5), ((64, 72), "reference_gl", 0.0), ((320, 180), "reference_gl", 0.5)):
    env = O.EnvMap(env_img, half=True, rtz=True)
    ctx = Context(W, H)
    ctx.set_uv_model(uv)
    ctx.set_environment(env_img, half_float_type=True, half_store_rtz=True)
    for l in range(env.levels):
        same = np.array_equal(ctx.download_environment(l, (64, 32)).view(np.uint32), env.level(l).view(np.uint32))
        if not same:
            print("  MIP LEVEL %d differs between the device and the restatement" % l)
    nbad = ntot = 0
    with O.uv_model({"ideal": "ideal", "reference_gl": "reference"}[uv]):
        for fi in range(3):
            f = synthetic_frame(W, H, fi)
            comp = np.random.RandomState(fi).rand(H, W, 4).astype(np.float32)
            for idx in (11, 12, 13, 14):
                sp = S.stage_params(f.camera, f.camera, 1.0, 20, 5)[0]
                sp.useEnvMap, sp.envBlur, sp.blueNoiseIndex = 1, blur, 1000 * fi + idx
                ctx.upload_frame(f)
                ctx.upload(abi.TEX_COMPOSE, comp)
                ctx.ssgi_march(sp)
                g1 = ctx.download(abi.TEX_SSGI)
                ctx.ssgi_march(sp)
                g2 = ctx.download(abi.TEX_SSGI)
                o = O.ssgi(f.depth, f.gbuffer, f.direct, comp, blue, sp, env=env)
                bad = out_of_tolerance(h8(g1), h8(o), True)
                proven = S.prove_flips(lambda: O.ssgi(f.depth, f.gbuffer, f.direct, comp, blue, sp, env=env), h8, bad, True) if bad.any() else bad
                un = bad & ~proven
                nbad += int(un.sum())
                ntot += W * H
                if not np.array_equal(g1, g2):
                    print("  NOT DETERMINISTIC: %d texels differ between two launches" % int((g1 != g2).any(-1).sum()))
                for y, x in np.argwhere(un)[:3]:
                    a, b = h8(g1)[y, x], h8(o)[y, x]
                    print("  f%d idx %d (y %d, x %d): device diffuse %s spec %s | restatement diffuse %s spec %s" % (
                        fi, sp.blueNoiseIndex, y, x, np.array2string(a[:3], precision=4), np.array2string(a[4:], precision=4),
                        np.array2string(b[:3], precision=4), np.array2string(b[4:], precision=4)))
    print("%dx%d uv %s envBlur %g: %d unproven out-of-tolerance pixels of %d" % (W, H, uv, blur, nbad, ntot), flush=True)
    ctx.close()
