#!/usr/bin/env python3
"""The two side draws of the path at random sizes against the reference's own GLSL on llvmpipe (oracle/glref):
  * CubeToEquirectEnvPass (scene.environment as a CubeTexture): random face sizes (odd ones too), with and without the mip chain (which for a face that is not a power of two goes through odd-sized levels), at the pass's own
    target size — every texel inside the fp32 rule of tests/parity.py (|a - b| <= 1e-3 or <= 1e-5 |b|);
  * the raster passes' packers (packGBuffer / packNormal) as the importer of attribute planes: random frame sizes and scenes, HDR emissive
    values sprinkled in — bit for bit on covered texels (the emissive word of BLACK-emissive texels apart: encodeRGBE8 takes log2(0), and what
    -inf becomes as a uint is the platform's; tests/test_oracle_vs_golden.py test_pack_gbuffer_and_velocity_vs_golden).
The implementation is the C restatement, or with --device the library (rfx_cube_to_equirect, rfx_pack_gbuffer / _velocity) on the GPU — held
against the reference GLSL where its sources are (the build container), against the restatement on the GPU box.

    python tools/fuzz_aux_vs_reference_gl.py [--n 60] [--seed 1] [--device]

TEST INFRASTRUCTURE."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "glref"), os.path.join(ROOT, "tests"),
           os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, _p)
import rfx_oracle as O  # noqa: E402
from rfx_amd.scene import AnalyticScene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--device", action="store_true")
a = ap.parse_args()
os.environ.setdefault("LP_NUM_THREADS", str(os.cpu_count() or 8))
have_src = os.path.isdir("/root/reference/src")
if have_src:
    import chain
    from make_golden import synthetic_cube
else:
    assert a.device, "without the reference's sources only --device (against the restatement) has something to compare"


def within(x, y):
    e = np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64))
    return (e <= 1e-3) | (e <= 1e-5 * np.abs(y))


def cube_faces(S, seed):
    if have_src:
        return synthetic_cube(S, seed)
    rng = np.random.RandomState(seed)  # (the GPU box: any HDR cube with contrast on every face, edge and corner)
    f = (rng.rand(6, S, S, 4) * np.array([2.0, 1.5, 1.0, 0.0]) + np.array([0.1, 0.1, 0.1, 1.0])).astype(np.float32)
    f[2, S // 2, S // 2, :3] = (40.0, 36.0, 28.0)
    return f


rng = np.random.RandomState(a.seed)
fails = ncube = npack = 0
t0 = time.time()
for it in range(a.n):
    try:
        if it % 2 == 0:  # ---- cube -> equirect
            S = int(rng.choice([rng.randint(1, 9), rng.randint(9, 49), 16, 31, 32, 33]))
            mip = bool(rng.randint(2))
            faces = cube_faces(S, 100 + it)
            import math
            W, H = (chain.cube_equirect_size(S) if have_src else (int(2 ** math.ceil(math.log2(2 * S * 3 ** 0.5))), int(2 ** math.ceil(math.log2(S * 3 ** 0.5)))))
            want = chain.run_cube_to_equirect(faces, W, H, mip) if have_src else O.cube_to_equirect(faces, W, H, mipmaps=mip)
            if a.device:
                from rfx_amd.context import Context
                ctx = Context(8, 8)
                got = ctx.cube_to_equirect(faces, W, H, generate_mipmaps=mip)
                ctx.close()
            else:
                got = O.cube_to_equirect(faces, W, H, mipmaps=mip)
            ok = within(got, want)
            ncube += 1
            if not ok.all():
                fails += 1
                print("MISMATCH cube S %d -> %dx%d mipmaps %d: %d texels out of tolerance, max |err| %.3e" % (S, W, H, mip, int((~ok).any(-1).sum()), float(np.abs(got - want).max())), flush=True)
        else:  # ---- the packers
            W = int(rng.choice([rng.randint(1, 40), rng.randint(40, 200), 64, 65, 127]))
            H = int(rng.choice([rng.randint(1, 24), rng.randint(24, 100), 8, 9]))
            f = AnalyticScene(int(rng.randint(1, 10000))).render(W, H, int(rng.randint(4)), aov=True)
            aov = {k: v.copy() for k, v in f.aov.items()}
            m = rng.rand(H, W) < 0.3
            aov["emissive"][m] = (rng.rand(int(m.sum()), 3) * np.array([8, 2, 0.5])).astype(np.float32)
            if have_src:
                wg, wv = chain.run_pack(aov, f.depth)
            else:
                wg, wv = O.pack_gbuffer(aov, None), O.pack_velocity(aov, f.depth)
            if a.device:
                from rfx_amd import abi
                from rfx_amd.context import Context
                ctx = Context(W, H)
                ctx.pack_gbuffer(aov, None)
                ctx.pack_velocity(aov, f.depth)
                gg, gv = ctx.download(abi.TEX_GBUFFER), ctx.download(abi.TEX_VELOCITY)
                ctx.close()
            else:
                gg, gv = O.pack_gbuffer(aov, None), O.pack_velocity(aov, f.depth)
            cov, lit = f.depth < 1.0, aov["emissive"].max(-1) > 0
            bad = [ch for ch in range(3) if not np.array_equal(gg[..., ch][cov], wg[..., ch][cov])]
            if not np.array_equal(gg[..., 3][cov & lit], wg[..., 3][cov & lit]):
                bad.append(3)
            if not np.array_equal(gv[cov], wv[cov]):
                bad.append("velocity")
            npack += 1
            if bad:
                fails += 1
                print("MISMATCH pack %dx%d: words %s differ on covered texels" % (W, H, bad), flush=True)
    except Exception as e:  # noqa: BLE001
        fails += 1
        import traceback
        print("ERROR %r\n%s" % (e, traceback.format_exc(limit=3)), flush=True)
    if (it + 1) % 20 == 0:
        print("... %d / %d cases, %d problems, %.0f s" % (it + 1, a.n, fails, time.time() - t0), flush=True)
print("%d cube conversions, %d packed frames against %s: %d problems" % (ncube, npack, "the reference GLSL" if have_src else "the restatement", fails))
sys.exit(1 if fails else 0)
