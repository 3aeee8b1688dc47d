"""-m gpu: CubeToEquirectEnvPass on the device (rfx_cube_to_equirect, csrc/k0_import.hip) — scene.environment given as a CubeTexture
(SSGIEffect.js:316-321, src/ssgi/pass/CubeToEquirectEnvPass.js:21-42).  Written after the round's GPU budget was spent: this file sorts
last so that its first run on hardware cannot hide any other test's result."""
import types

import numpy as np
import pytest

import golden_util as G
import rfx_oracle as O
from rfx_amd import abi
from rfx_amd.context import Context

pytestmark = pytest.mark.gpu


def _within(a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return (e <= 1e-3) | (e <= 1e-5 * np.abs(b))


@pytest.mark.parametrize("mipmaps", [False, True])
def test_cube_to_equirect_vs_oracle_and_reference(mipmaps):
    """HIP against the C restatement and against the pass's own GLSL on llvmpipe (tests/golden/cube_32.npz), fp32-output tolerance
    (1e-3 absolute or 1e-5 relative) on every texel: LinearFilter cube and three's default mipmapped CubeTexture."""
    g = G.load("cube_32")
    W, H = int(g["width"]), int(g["height"])
    faces = np.ascontiguousarray(g["faces"])
    ctx = Context(64, 64)
    got = ctx.cube_to_equirect(faces, W, H, generate_mipmaps=mipmaps)
    want_o = O.cube_to_equirect(faces, W, H, mipmaps=mipmaps)
    want_g = g["equirect_mipmapped" if mipmaps else "equirect_linear"]
    eo, eg = np.abs(got - want_o), np.abs(got - want_g)
    print("mipmaps %d: vs oracle max %.3e (bit-identical %.3f), vs reference GLSL max %.3e" % (mipmaps, eo.max(), (eo == 0).mean(), eg.max()))
    assert _within(got, want_o).all() and _within(got, want_g).all()
    # other sizes, odd face sizes included (no chain there), against the oracle
    rng = np.random.RandomState(4)
    for S in ((5, 12, 64) if not mipmaps else (1, 2, 16, 64)):
        f2 = (rng.rand(6, S, S, 4) ** 3 * 20).astype(np.float32)
        w2, h2 = 4 * max(S, 8), 2 * max(S, 8)
        a, b = ctx.cube_to_equirect(f2, w2, h2, generate_mipmaps=mipmaps), O.cube_to_equirect(f2, w2, h2, mipmaps=mipmaps)
        assert _within(a, b).all(), (S, np.abs(a - b).max())
    # a constant cube converts to that constant exactly, whatever the face size (edges, corners and both mip levels blend weights that sum to one)
    for S in ((1, 7, 256) if not mipmaps else (2, 256)):
        c = np.empty((6, S, S, 4), np.float32)
        c[...] = np.array([0.3, 7.25, 1e-3, 1.0], np.float32)
        assert (ctx.cube_to_equirect(c, 64, 32, generate_mipmaps=mipmaps) == c[0, 0, 0]).all(), S
    ctx.close()


def test_cube_environment_through_the_effect_on_device(blue_noise):
    """SSGIEffect with a CubeTexture environment drives the device: converted once, stored FloatType, lit frame equals the run that is
    handed the converted equirectangular map directly."""
    from rfx_amd.effect import FloatType, SSGIEffect
    from rfx_amd.scene import synthetic_frame

    g = G.load("cube_32")
    W, H = 160, 90
    f = synthetic_frame(W, H, 0)
    faces = np.ascontiguousarray(g["faces"])
    outs = []
    for env in (dict(isCubeTexture=True, faces=faces), None):
        ctx = Context(W, H)
        if env is None:
            env = dict(data=ctx.cube_to_equirect(faces, int(g["width"]), int(g["height"]), generate_mipmaps=True), type=FloatType)
        scene = types.SimpleNamespace(frame=f, environment=env)
        fx = SSGIEffect(None, scene, f.camera, dict(width=W, height=H, steps=8, refineSteps=2), seeds=dict(ssgi=5, denoise=6))
        fx.update(ctx, None)
        outs.append((ctx.download(abi.TEX_SSGI), ctx.download(abi.TEX_COMPOSE)))
        assert ctx.environment_levels() == 8  # 128 x 64 -> 8 levels
        ctx.close()
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
