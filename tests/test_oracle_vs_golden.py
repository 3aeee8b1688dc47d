"""CPU: the C restatement (oracle/rfx_oracle.c) against the golden vectors produced by the
reference's own GLSL on llvmpipe (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

import golden_util as G
import rfx_oracle as O
from parity import assert_close, compare
from rfx_amd import abi

# allowed fraction of discontinuity-flipped pixels (see tests/parity.py); measured values are ~5x lower
FLIP = dict(ssgi=1.5e-3, temporal=1.5e-3, denoise0=8e-3, denoise=2e-3, compose=1e-3)


# ... and with both sides on the reference GL's own vUv (rfxo_set_uv_model(1): oracle/rfx_oracle.c frag_u / frag_v restate its rasteriser's
# plane equations bit for bit) the flips of K2 / K3 / K4 vanish: measured 0 on every stage of every golden file, K1 <= 1.9e-4 (4.3e-4 at
# resolutionScale 0.5) from transcendental rounding at its discontinuities.  Bounds: one pixel of the smallest frame / ~3x measured.
FLIP_REFERENCE_UV = dict(ssgi=6e-4, temporal=2e-4, denoise0=2e-4, denoise=2e-4, compose=3e-4)


@pytest.fixture(autouse=True, params=["ideal", "reference"])
def uv_model(request):
    """every test of this file runs twice: under the ideal vUv (what the HIP kernels compute by default) and under the reference GL's"""
    keep = dict(FLIP)
    if request.param == "reference":
        FLIP.update(FLIP_REFERENCE_UV)
    with O.uv_model(request.param):
        yield request.param
    FLIP.update(keep)


def stage_params(g, fi, keep):
    cam = abi.Camera.from_scene(G.camera(g, fi))
    prev = abi.Camera.from_scene(G.camera(g, fi - 1 if fi > 0 else 0))
    sp = abi.SsgiParams(camera=cam, steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), mode=0, useDirectLight=1, rayDistance=10,
                        thickness=10, envBlur=0.5, blueNoiseIndex=int(g["f%d_ssgi_index" % fi]),
                        missedRays=int(g["missedRays"]) if "missedRays" in g.files else 0)
    tp = abi.TemporalParams(camera=cam, prevCamera=prev, textureCount=2, inputType=0, logTransform=1, fullAccumulate=0, confidencePower=0.75,
                            neighborhoodClampIntensity=0.5, maxBlend=1.0, keepData=keep)
    tp.reprojectSpecular[:] = [0, 1]
    tp.neighborhoodClamp[:] = [0, 1]
    dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=2,
                           halfStoreRTZ=1)
    dp.isTextureSpecular[:] = [0, 1]
    cp = abi.ComposeParams(camera=cam, inputType=0)
    return sp, tp, dp, cp


def assert_strict(name, rerun, want, half, max_fraction=0.01):
    """Round-2 metric on one stage: `rerun()` evaluates the oracle stage (it is called again under the margin / perturbation machinery),
    `want` is the reference's output.  Every out-of-tolerance pixel must be proven unstable by the oracle (see _unstable below)."""
    from parity import strict

    def fl(outs):
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        return np.concatenate([O.half_bits_to_float(np.ascontiguousarray(o).view(np.uint16)) if o.dtype in (np.uint16, np.uint32) else o for o in outs], axis=-1)
    got, ref = fl(rerun()), fl(want)
    H, W = got.shape[:2]
    r = strict(name, got, ref, explainable=_unstable(rerun, half, H, W), half=half)
    print(r.line())
    assert r.unexplained == 0, r.line()
    assert r.bad <= max_fraction * r.pixels, r.line()
    return r


def test_k3_rotation_table_against_libm_sincos(blue_noise):
    """Round 3 moved K3's tap rotation — in the kernel AND in this restatement — from sinf / cosf of the fp32 angle to the correctly rounded
    (sin, cos) of the 256 possible angles.  The two forms, each against the reference's golden K3 outputs: both inside the stage's bound, and
    what the choice moves is a handful of texels (a radius-3 tap of a flat surface at 120 / 240 degrees sits on a texel boundary; the table's
    last bit of cos is the reference GL's there, libm's is not) — the kernel's choice is checked independently of the restatement's default."""
    g = G.load("chain_160x90_s20r5_it1")
    W, H = int(g["width"]), int(g["height"])
    f = G.frame(g, 1)
    _, _, dp, _ = stage_params(g, 1, 1.0)
    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g["f1_denoise_index"][0]), 1, 0

    def run():
        A = [np.ascontiguousarray(g["f0_A%d" % j]).copy() for j in range(2)]
        O.denoise(f.depth, f.gbuffer, np.ascontiguousarray(g["f1_temporal0"]), np.ascontiguousarray(g["f1_temporal1"]), blue_noise, dp, A[0], A[1])
        return [O.half_bits_to_float(a) for a in A]
    table = run()
    with O.k3_rotation_libm():
        libm = run()
    moved = 0
    for j in range(2):
        want = O.half_bits_to_float(g["f1_A%d" % j])
        ft, _ = assert_close("table A%d" % j, table[j], want, FLIP["denoise0"])
        fl, _ = assert_close("libm  A%d" % j, libm[j], want, FLIP["denoise0"])
        moved += int((table[j] != libm[j]).any(axis=-1).sum())
        print("K3 pass 0 tex%d vs golden: table %.4f %%, libm %.4f %% of the texels outside 1e-3" % (j, 100 * ft, 100 * fl))
    print("texels the rotation form moves at all: %d of %d" % (moved, 2 * W * H))
    assert moved <= 0.01 * 2 * W * H


@pytest.mark.parametrize("name", G.GOLDENS)
def test_stagewise(name, blue_noise):
    """Every pass fed with the GOLDEN outputs of the previous passes."""
    g = G.load(name)
    W, H, nf, it = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["denoiseIterations"])
    zero16 = np.zeros((H, W, 4), np.uint16)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, dp, cp = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        hist = g[kp + "compose"] if fi else np.zeros((H, W, 4), np.float32)
        o = O.ssgi(f.depth, f.gbuffer, f.direct, np.ascontiguousarray(hist), blue_noise, sp)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(o)
        assert_close(name + " ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"])
        assert_close(name + " ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"])
        assert (o == g[k + "ssgi"]).all(axis=-1).mean() > 0.995  # bit-identical packed texels for >99.5% of pixels

        B = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else zero16 for j in range(2)]
        T = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else np.zeros((H, W, 4), np.float32) for j in range(2)]
        O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B[0], B[1], tp, T[0], T[1])
        for j in range(2):
            assert_close(name + " temporal%d f%d" % (j, fi), T[j], g[k + "temporal%d" % j], FLIP["temporal"])

        # only the first and the last K3 pass outputs survive in A/B; check pass 0 when it == 1, and B always
        if it == 1:
            A = [np.ascontiguousarray(g[kp + "A%d" % j]) if fi else zero16.copy() for j in range(2)]
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][0]), 1, 0
            O.denoise(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "temporal0"]), np.ascontiguousarray(g[k + "temporal1"]), blue_noise, dp, A[0], A[1])
            for j in range(2):
                assert_close(name + " A%d f%d" % (j, fi), O.half_bits_to_float(A[j]), O.half_bits_to_float(g[k + "A%d" % j]), FLIP["denoise0"])
        Bn = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else zero16.copy() for j in range(2)]
        if it > 1:  # B of this frame was overwritten by pass 1 before the final pass: start from zero-history is not reproducible -> skip mask
            Bn = [np.ascontiguousarray(g[k + "B%d" % j]).copy() for j in range(2)]
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][-1]), 0, 1
        O.denoise(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "A0"]), np.ascontiguousarray(g[k + "A1"]), blue_noise, dp, Bn[0], Bn[1])
        for j in range(2):
            assert_close(name + " B%d f%d" % (j, fi), O.half_bits_to_float(Bn[j]), O.half_bits_to_float(g[k + "B%d" % j]), FLIP["denoise"])

        comp = np.ascontiguousarray(hist).copy()
        O.compose(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "B0"]), np.ascontiguousarray(g[k + "B1"]), cp, comp)
        assert_close(name + " compose f%d" % fi, comp, g[k + "compose"], FLIP["compose"])


@pytest.mark.parametrize("name", G.GOLDENS)
def test_full_chain_through_effect(name):
    """The host mirror (rfx_amd.effect.SSGIEffect: option plumbing, blue-noise recurrence, keepData,
    ping-pong, history wiring) driving the oracle renderer must reproduce the reference chain's
    per-frame outputs; flipped pixels compound through the feedback loop, hence looser bounds."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import SSGIEffect

    g = G.load(name)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    scene = types.SimpleNamespace(frame=None)
    cam = G.camera(g, 0)
    fx = SSGIEffect(None, scene, cam, dict(steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), denoiseIterations=int(g["denoiseIterations"]),
                                           missedRays=bool(int(g["missedRays"])) if "missedRays" in g.files else False, width=W, height=H),
                    seeds=dict(ssgi=int(g["ssgi_start"]), denoise=int(g["denoise_start"])))
    r = OracleRenderer(W, H)
    for fi in range(nf):
        scene.frame = G.frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        fx.update(r, None)
        k = "f%d_" % fi
        assert r.calls[0] == ("ssgi", int(g[k + "ssgi_index"]))
        # measured ~0.6% (2 passes) / ~2.2% (4 passes) at frame 0: a flipped K3 tap spreads to its ~9 neighbours in every later pass
        lim = 0.01 * 2 * int(g["denoiseIterations"]) * (fi + 1) + 0.005
        assert_close(name + " chain temporal0 f%d" % fi, r.tex[abi.TEX_TEMPORAL0], g[k + "temporal0"], lim)
        assert_close(name + " chain B1 f%d" % fi, O.half_bits_to_float(r.tex[abi.TEX_DENOISE_B1]), O.half_bits_to_float(g[k + "B1"]), lim)
        assert_close(name + " chain compose f%d" % fi, r.tex[abi.TEX_COMPOSE], g[k + "compose"], lim)
        assert np.abs(r.tex[abi.TEX_COMPOSE].astype(np.float64) - g[k + "compose"]).mean() < 2e-4
        di = [c[1] for c in r.calls if c[0] == "denoise"]
        assert di == [int(x) for x in g[k + "denoise_index"]]
        r.calls.clear()


def ssr_params(g, fi, keep):
    """mode "ssr" (SSGIEffect.js:70-73): inputType "specular", one texture, reprojectSpecular/neighborhoodClamp true."""
    cam = abi.Camera.from_scene(G.camera(g, fi))
    prev = abi.Camera.from_scene(G.camera(g, fi - 1 if fi > 0 else 0))
    sp = abi.SsgiParams(camera=cam, steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), mode=1, useDirectLight=1, rayDistance=10, thickness=10,
                        envBlur=0.5, blueNoiseIndex=int(g["f%d_ssgi_index" % fi]))
    tp = abi.TemporalParams(camera=cam, prevCamera=prev, textureCount=1, inputType=2, logTransform=1, fullAccumulate=0, confidencePower=0.75,
                            neighborhoodClampIntensity=0.5, maxBlend=1.0, keepData=keep)
    tp.reprojectSpecular[:] = [1, 1]
    tp.neighborhoodClamp[:] = [1, 1]
    dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=1, halfStoreRTZ=1)
    dp.isTextureSpecular[:] = [1, 1]
    cp = abi.ComposeParams(camera=cam, inputType=2)
    return sp, tp, dp, cp


def ssr_unpack(ssgi_bits):
    """MODE_SSR K1 texel: raw rgb floats + packHalf2x16(rayLength, roughness) in .a"""
    rgb = ssgi_bits[..., :3].view(np.float32)
    a = ssgi_bits[..., 3]
    ray = (a & np.uint32(0xffff)).astype(np.uint16).view(np.float16).astype(np.float32)
    rough = (a >> np.uint32(16)).astype(np.uint16).view(np.float16).astype(np.float32)
    return np.concatenate([rgb, ray[..., None], rough[..., None]], axis=-1)


def test_stagewise_ssr_mode(blue_noise):
    """mode "ssr": K1 MODE_SSR, K2 inputType SPECULAR, K3 with one specular texture, K4 TYPE_SPECULAR (sceneTexture)."""
    g = G.load(G.GOLDEN_SSR)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    z16 = np.zeros((H, W, 4), np.uint16)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, dp, cp = ssr_params(g, fi, 0.0 if fi == 0 else 1.0)
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else np.zeros((H, W, 4), np.float32)
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
        fg = f.depth < 1.0  # background texels carry packTwoVec4(direct, direct) in both modes (ssgi.frag:109-113)
        assert_close("ssr ssgi f%d" % fi, ssr_unpack(o)[fg][None], ssr_unpack(g[k + "ssgi"])[fg][None], FLIP["ssgi"])
        # (raw fp32 output here: no half rounding to hide last-ulp transcendental differences, so no bit-identity claim)
        B0 = np.ascontiguousarray(g[kp + "B0"]) if fi else z16
        T0 = np.ascontiguousarray(g[kp + "temporal0"]) if fi else np.zeros((H, W, 4), np.float32)
        T_init = T0.copy()
        O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B0, B0, tp, T0, None)
        assert_close("ssr temporal0 f%d" % fi, T0, g[k + "temporal0"], FLIP["temporal"])

        def k2(f=f, k=k, B0=B0, tp=tp, T_init=T_init):
            t = T_init.copy()
            O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B0, B0, tp, t, None)
            return t
        assert_strict("ssr K2 f%d" % fi, k2, g[k + "temporal0"], False)
        A0 = np.ascontiguousarray(g[kp + "A0"]) if fi else z16.copy()
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][0]), 1, 0
        t0 = np.ascontiguousarray(g[k + "temporal0"])
        O.denoise(f.depth, f.gbuffer, t0, t0, blue_noise, dp, A0, None)
        assert_close("ssr A0 f%d" % fi, O.half_bits_to_float(A0), O.half_bits_to_float(g[k + "A0"]), FLIP["denoise0"])
        Bn = np.ascontiguousarray(g[kp + "B0"]).copy() if fi else z16.copy()
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][1]), 0, 1
        a0 = np.ascontiguousarray(g[k + "A0"])
        O.denoise(f.depth, f.gbuffer, a0, a0, blue_noise, dp, Bn, None)
        assert_close("ssr B0 f%d" % fi, O.half_bits_to_float(Bn), O.half_bits_to_float(g[k + "B0"]), FLIP["denoise"])
        comp = hist.copy()
        O.compose(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "B0"]), None, cp, comp, scene=f.direct)
        assert_close("ssr compose f%d" % fi, comp, g[k + "compose"], FLIP["compose"])


def traa_params(g, fi):
    """TRAAEffect.js:21-33 over defaultTemporalReprojectPassOptions; the example's fullAccumulate option never fires while the camera orbits."""
    half = bool(int(g["half"]))
    cam = abi.Camera.from_scene(G.camera(g, fi))
    prev = abi.Camera.from_scene(G.camera(g, fi - 1 if fi > 0 else 0))
    tp = abi.TemporalParams(camera=cam, prevCamera=prev, textureCount=1, inputType=1, logTransform=1, fullAccumulate=0, confidencePower=4,
                            neighborhoodClampIntensity=1, maxBlend=0.9, keepData=1.0, historySource=1 if half else 2, targetHalf=1 if half else 0,
                            halfStoreRTZ=1)
    tp.reprojectSpecular[:] = [0, 0]
    tp.neighborhoodClamp[:] = [1, 1]
    return tp, half


def traa_input(g, fi, half):
    d = np.ascontiguousarray(g["f%d_direct" % fi])
    if half:  # a HalfFloatType composer buffer holds half texels
        d = d.astype(np.float16).astype(np.float32)
    return d.view(np.uint32)


@pytest.mark.parametrize("name", G.GOLDEN_TRAA)
def test_traa_stagewise(name):
    """TRAAEffect's TemporalReprojectPass fed with the GOLDEN framebuffer copy of the previous frame."""
    g = G.load(name)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    for fi in range(nf):
        tp, half = traa_params(g, fi)
        prev_out = np.ascontiguousarray(g["f%d_out" % (fi - 1)]) if fi else np.zeros((H, W, 4), np.float32)
        hist = prev_out.astype(np.float16).view(np.uint16) if half else prev_out
        out = np.zeros((H, W, 4), np.float32)
        O.temporal(traa_input(g, fi, half), np.ascontiguousarray(g["f%d_velocity" % fi]), hist, hist, tp, out, None)
        assert_close(name + " out f%d" % fi, out, g["f%d_out" % fi], FLIP["temporal"])
        if half:  # every stored channel is a half
            assert (out.astype(np.float16).astype(np.float32) == out).all()


@pytest.mark.parametrize("name", G.GOLDEN_TRAA)
def test_traa_through_effect(name):
    """rfx_amd.effect.TRAAEffect (target type from the input buffer, own framebuffer copy as history, prev-camera bookkeeping)
    driving the oracle renderer reproduces the reference sequence."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import FloatType, HalfFloatType, TRAAEffect, VelocityDepthNormalPass

    g = G.load(name)
    W, H, nf, half = int(g["width"]), int(g["height"]), int(g["frames"]), bool(int(g["half"]))
    scene = types.SimpleNamespace(frame=None)
    cam = G.camera(g, 0)
    fx = TRAAEffect(scene, cam, VelocityDepthNormalPass(scene, cam), dict(fullAccumulate=True))
    r = OracleRenderer(W, H)
    for fi in range(nf):
        scene.frame = G.traa_frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        buf = dict(texture=dict(type=HalfFloatType if half else FloatType), width=W, height=H, data=scene.frame.direct)
        fx.update(r, buf)
        assert [c[0] for c in r.calls] == ["temporal", "copy_framebuffer"]
        assert r.calls[1][1] == (abi.TEX_FBCOPY_F16 if half else abi.TEX_FBCOPY_F32)
        r.calls.clear()
        assert_close(name + " effect out f%d" % fi, r.tex[abi.TEX_TEMPORAL0], g["f%d_out" % fi], FLIP["temporal"] * (fi + 1))
        o = fx.output(r)
        assert (o[..., 3] == 1.0).all() and (o[..., :3] == r.tex[abi.TEX_TEMPORAL0][..., :3]).all()


def final_params(g, mode, debug=0):
    cam = abi.Camera(near_=float(g["near"]), far_=float(g["far"]), isPerspective=1)
    p = abi.FinalParams(camera=cam, isDebug=debug, fogMode=mode, fogNear=float(g["fogNear"]), fogFar=float(g["fogFar"]), fogDensity=float(g["fogDensity"]))
    p.fogColor[:] = [float(x) for x in g["fogColor"]]
    return p


def test_final_compose_vs_golden():
    """SSGIEffect's own fragment (ssgi_compose.frag): background -> scene colour, else composed GI, THREE.Fog / FogExp2, isDebug."""
    g = G.load(G.GOLDEN_FINAL)
    depth, gi, scene = (np.ascontiguousarray(g[k]) for k in ("depth", "gi", "scene"))
    for mode in (0, 1, 2):
        out = O.final(depth, gi, scene, final_params(g, mode))
        assert_close("final fog%d" % mode, out, g["final_fog%d" % mode], 0.0)
        if mode == 0:
            assert np.array_equal(out.view(np.uint32), g["final_fog0"].view(np.uint32))  # a select: bit-exact
    assert np.array_equal(O.final(depth, gi, scene, final_params(g, 0, 1)).view(np.uint32), g["final_debug"].view(np.uint32))
    assert (g["final_fog1"] != g["final_fog0"]).any() and (g["final_fog2"] != g["final_fog1"]).any()


@pytest.mark.parametrize("name", G.GOLDEN_MODES)
def test_denoise_modes_stagewise(name, blue_noise):
    """denoiseMode "full_temporal" / "temporal" / "denoised" (Denoiser.js:41-78): K2 on its own framebuffer copy (both textures
    read the copy of colour attachment 0), K4 fed by K2's targets, K1's history = K4's target / K2's texture[0] / three's empty
    texture, and the effect's own fragment on the mode's output texture — every pass fed with the GOLDEN previous outputs."""
    g = G.load(name)
    dm = str(g["denoiseMode"])
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    zf, z16 = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.uint16)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, dp, cp = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        # K1
        sp.historySource = {"full_temporal": 0, "temporal": 1, "denoised": 2}[dm]
        hist = zf if (fi == 0 or dm == "denoised") else np.ascontiguousarray(g[kp + ("compose" if dm == "full_temporal" else "temporal0")])
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
        oa, ob = O.unpack_ssgi(o)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        assert_close(name + " ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"])
        assert_close(name + " ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"])
        # K2
        T = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else zf.copy() for j in range(2)]
        if dm == "denoised":
            h = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else z16 for j in range(2)]
        else:
            tp.historySource = 2
            h = [np.ascontiguousarray(g[kp + "temporal0"]) if fi else zf] * 2  # ONE copy, of attachment 0, for both textures
        O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, h[0], h[1], tp, T[0], T[1])
        for j in range(2):
            assert_close(name + " temporal%d f%d" % (j, fi), T[j], g[k + "temporal%d" % j], FLIP["temporal"])
        # K3 ("denoised" only) is the pass test_stagewise pins; here: its last pass from the golden A
        if dm == "denoised":
            Bn = [np.ascontiguousarray(g[kp + "B%d" % j]).copy() if fi else z16.copy() for j in range(2)]
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][-1]), 0, 1
            O.denoise(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "A0"]), np.ascontiguousarray(g[k + "A1"]), blue_noise, dp, Bn[0], Bn[1])
            for j in range(2):
                assert_close(name + " B%d f%d" % (j, fi), O.half_bits_to_float(Bn[j]), O.half_bits_to_float(g[k + "B%d" % j]), FLIP["denoise"])
        # K4 ("full_temporal": composerInputTextures = K2's targets)
        if dm == "full_temporal":
            cp.giSource = 1
            comp = (np.ascontiguousarray(g[kp + "compose"]) if fi else zf).copy()
            O.compose(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "temporal0"]), np.ascontiguousarray(g[k + "temporal1"]), cp, comp)
            assert_close(name + " compose f%d" % fi, comp, g[k + "compose"], FLIP["compose"])
        # the effect's own fragment on the mode's output texture
        fp = abi.FinalParams(camera=abi.Camera.from_scene(f.camera), inputSource={"full_temporal": 0, "temporal": 1, "denoised": 2}[dm])
        src = {"full_temporal": "compose", "temporal": "temporal0", "denoised": "B0"}[dm]
        out = O.final(f.depth, np.ascontiguousarray(g[k + src]), f.direct, fp)
        assert np.array_equal(out.view(np.uint32), g[k + "final"].view(np.uint32)), "final f%d" % fi


@pytest.mark.parametrize("name", G.GOLDEN_MODES)
def test_denoise_modes_through_effect(name):
    """SSGIEffect(denoiseMode=...) on the oracle renderer: pass construction per mode, history wiring, K1's accumulatedTexture,
    the effect's inputTexture (rfx_amd.effect Denoiser / SSGIPass / SSGIEffect.update)."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import SSGIEffect

    g = G.load(name)
    dm = str(g["denoiseMode"])
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    scene = types.SimpleNamespace(frame=None)
    cam = G.camera(g, 0)
    fx = SSGIEffect(None, scene, cam, dict(steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), denoiseIterations=1, denoiseMode=dm, width=W, height=H),
                    seeds=dict(ssgi=int(g["ssgi_start"]), denoise=int(g["denoise_start"])))
    r = OracleRenderer(W, H)
    want_calls = {"full_temporal": ["ssgi", "temporal", "copy_framebuffer", "compose", "final"], "temporal": ["ssgi", "temporal", "copy_framebuffer", "final"],
                  "denoised": ["ssgi", "temporal", "denoise", "denoise", "final"]}[dm]
    for fi in range(nf):
        scene.frame = G.frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        fx.update(r, None)
        fx.mainImage(r)
        assert [c[0] for c in r.calls] == want_calls
        r.calls.clear()
        k = "f%d_" % fi
        # flipped K3 taps spread to their neighbours in every later pass (cf. test_full_chain_through_effect)
        lim = (0.02 if dm == "denoised" else 0.01) * (fi + 1) + 0.005
        assert_close(name + " chain temporal1 f%d" % fi, r.tex[abi.TEX_TEMPORAL1], g[k + "temporal1"], lim)
        assert_close(name + " chain final f%d" % fi, r.tex[abi.TEX_FINAL], g[k + "final"], lim)


@pytest.mark.parametrize("name", G.GOLDEN_ENV)
def test_env_map_stagewise(name, blue_noise):
    """scene.environment (USE_ENVMAP, SURVEY.md §8f-1 without MIS): missed rays and the screen-border fade take the equirect map's colour,
    sampled trilinearly at envBlur * maxEnvMapMipLevel (ssgi.frag:311-346).  The mip chain the effect asks the driver for is part of the
    statement: rfxo_env_build is pinned bit for bit against glGenerateMipmap on llvmpipe in make_golden's harness (oracle/glref)."""
    g = G.load(name)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    env = O.EnvMap(g["environment"], half=True, rtz=True)
    assert env.levels == int(np.log2(max(env.w, env.h))) + 1
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, _, _, _ = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        sp.useEnvMap, sp.envBlur = 1, float(g["envBlur"])
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else np.zeros((H, W, 4), np.float32)
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp, env=env)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(o)
        assert_close(name + " ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"])
        assert_close(name + " ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"])
        assert_strict(name + " K1 env f%d" % fi, lambda: O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp, env=env), g[k + "ssgi"], True)
        assert (o == g[k + "ssgi"]).all(axis=-1).mean() > 0.99
        # the environment really contributes: the same draw without it differs on a good part of the frame
        sp.useEnvMap = 0
        o0 = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
        assert (o0 != o).any(axis=-1).mean() > 0.2


@pytest.mark.parametrize("name", G.GOLDEN_ENV)
def test_env_map_through_effect(name):
    """SSGIEffect.keepEnvMapUpdated on the oracle renderer: scene.environment is handed to the device once, USE_ENVMAP switches on, the
    envBlur option reaches K1 (importanceSampling: false)."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import SSGIEffect

    g = G.load(name)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    scene = types.SimpleNamespace(frame=None, environment=dict(data=np.ascontiguousarray(g["environment"])))
    cam = G.camera(g, 0)
    opts = dict(steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), denoiseIterations=1, envBlur=float(g["envBlur"]), width=W, height=H)
    fx = SSGIEffect(None, scene, cam, dict(opts, importanceSampling=False), seeds=dict(ssgi=int(g["ssgi_start"]), denoise=int(g["denoise_start"])))
    r = OracleRenderer(W, H)
    for fi in range(nf):
        scene.frame = G.frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        fx.update(r, None)
        assert [c[0] for c in r.calls].count("set_environment") == (1 if fi == 0 else 0)
        r.calls.clear()
        k = "f%d_" % fi
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(r.tex[abi.TEX_SSGI])
        lim = 0.03 * (fi + 1)  # frame >= 1 marches against the chain's own (flip-compounded) composed history
        assert_close(name + " effect ssgi.diffuse f%d" % fi, oa, ga, lim)
        assert_close(name + " effect compose f%d" % fi, r.tex[abi.TEX_COMPOSE], g[k + "compose"], lim)
    scene.environment = None  # :361-366 the define goes away with the environment
    fx.update(r, None)
    assert fx.ssgiPass.uniforms.useEnvMap == 0 and r.env is None


@pytest.mark.parametrize("name", G.GOLDEN_RS)
def test_resolution_scale(name, blue_noise):
    """resolutionScale < 1 (SSGIPass.js:52-57): K1 renders a (W*s) x (H*s) target whose `resolution` drives vUv and the blue-noise pixel;
    K2 samples that smaller texture NEAREST at full-resolution vUv (its 5x5 neighbourhood then revisits texels).  Stage-wise, then the
    whole chain through SSGIEffect(resolutionScale=...)."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import SSGIEffect

    g = G.load(name)
    W, H, nf, rs = int(g["width"]), int(g["height"]), int(g["frames"]), float(g["resolutionScale"])
    oW, oH = int(W * rs), int(H * rs)
    assert g["f0_ssgi"].shape == (oH, oW, 4)
    z16, zf = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.float32)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, _, _ = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        sp.resolutionScale = rs
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else zf
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
        assert o.shape == (oH, oW, 4)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(o)
        assert_close(name + " ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"] * 2)
        assert_close(name + " ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"] * 2)
        tp.inputWidth, tp.inputHeight = oW, oH
        B = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else z16 for j in range(2)]
        T = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else zf.copy() for j in range(2)]
        O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B[0], B[1], tp, T[0], T[1])
        for j in range(2):
            assert_close(name + " temporal%d f%d" % (j, fi), T[j], g[k + "temporal%d" % j], FLIP["temporal"])
    # through the effect
    scene = types.SimpleNamespace(frame=None)
    cam = G.camera(g, 0)
    fx = SSGIEffect(None, scene, cam, dict(steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), denoiseIterations=1, resolutionScale=rs, width=W, height=H),
                    seeds=dict(ssgi=int(g["ssgi_start"]), denoise=int(g["denoise_start"])))
    assert (fx.denoiser.temporalReprojectPass.uniforms.inputWidth, fx.denoiser.temporalReprojectPass.uniforms.inputHeight) == (oW, oH)
    r = OracleRenderer(W, H)
    for fi in range(nf):
        scene.frame = G.frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        fx.update(r, None)
        assert_close(name + " chain compose f%d" % fi, r.tex[abi.TEX_COMPOSE], g["f%d_compose" % fi], 0.03 * (fi + 1))


def test_pack_gbuffer_and_velocity_vs_golden():
    """Encode side of the codec: packGBuffer / packNormal over attribute planes, bit for bit against the reference GLSL on
    llvmpipe — except the emissive word of BLACK-emissive texels: encodeRGBE8 takes log2(0) there (Appendix D-9) and what comes out depends
    on how the platform converts -inf to uint (llvmpipe: exponent byte 255, which DECODES to ~1e38; GPUs and this importer saturate to 0,
    which decodes to 0)."""
    g = G.load(G.GOLDEN_PACK)
    aov = {k[4:]: g[k] for k in g.files if k.startswith("aov_")}
    depth = np.ascontiguousarray(g["depth"])
    cov, lit = depth < 1.0, aov["emissive"].max(-1) > 0
    og = O.pack_gbuffer(aov, None)
    for ch in range(3):
        assert np.array_equal(og[..., ch][cov], g["gbuffer"][..., ch][cov]), ch
    assert np.array_equal(og[..., 3][cov & lit], g["gbuffer"][..., 3][cov & lit])
    assert set(np.unique(og[..., 3][cov & ~lit])) == {0x00fefefe} and set(np.unique(g["gbuffer"][..., 3][cov & ~lit])) == {0xfffefefe}
    ov = O.pack_velocity(aov, depth)
    assert np.array_equal(ov[cov], g["velocity"][cov])
    # uncovered texels keep the passes' clear colour (0, 0, 0, 1)
    og2 = O.pack_gbuffer(aov, depth)
    assert (og2[~cov] == np.array([0, 0, 0, 0x3f800000], np.uint32)).all() and (ov[~cov] == np.array([0, 0, 0, 0x3f800000], np.uint32)).all()
    # and the synthetic dumps' own (numpy) packer agrees with the importer wherever the emissive is lit or black alike decodes to 0
    from rfx_amd.scene import AnalyticScene
    f = AnalyticScene(1234).render(96, 54, 1, aov=True)
    p = O.pack_gbuffer(f.aov, f.depth)
    assert np.array_equal(p[..., :3], f.gbuffer[..., :3]) and np.array_equal(O.pack_velocity(f.aov, f.depth), f.velocity)


def test_orthographic_camera_stagewise(blue_noise):
    """An OrthographicCamera: every pass is compiled WITHOUT its PERSPECTIVE_CAMERA define (SSGIPass.js:38, TemporalReprojectPass.js:82,
    DenoiserComposePass.js:110, SSGIEffect.js:63) — depth -> view-Z switches to the orthographic formula in K1 (the march), K2 (the
    disocclusion distance factor), K4 and the effect's fog; the projection is the general matrix path."""
    g = G.load(G.GOLDEN_ORTHO)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    assert not G.camera(g, 0).isPerspectiveCamera
    z16, zf = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.float32)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, dp, cp = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        assert sp.camera.isPerspective == 0
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else zf
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(o)
        assert_close("ortho ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"])
        assert_close("ortho ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"])
        assert_strict("ortho K1 f%d" % fi, lambda: O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp), g[k + "ssgi"], True)
        assert (o == g[k + "ssgi"]).all(axis=-1).mean() > 0.99
        B = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else z16 for j in range(2)]
        T = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else zf.copy() for j in range(2)]
        O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B[0], B[1], tp, T[0], T[1])
        for j in range(2):
            assert_close("ortho temporal%d f%d" % (j, fi), T[j], g[k + "temporal%d" % j], FLIP["temporal"])
        T0 = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else zf.copy() for j in range(2)]

        def k2():
            Tn = [t.copy() for t in T0]
            O.temporal(np.ascontiguousarray(g[k + "ssgi"]), f.velocity, B[0], B[1], tp, Tn[0], Tn[1])
            return Tn
        assert_strict("ortho K2 f%d" % fi, k2, [g[k + "temporal0"], g[k + "temporal1"]], False)
        comp = hist.copy()
        O.compose(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "B0"]), np.ascontiguousarray(g[k + "B1"]), cp, comp)
        assert_close("ortho compose f%d" % fi, comp, g[k + "compose"], FLIP["compose"])

        def k4():
            c4 = hist.copy()
            O.compose(f.depth, f.gbuffer, np.ascontiguousarray(g[k + "B0"]), np.ascontiguousarray(g[k + "B1"]), cp, c4)
            return c4
        assert_strict("ortho K4 f%d" % fi, k4, g[k + "compose"], False)
        fp = abi.FinalParams(camera=abi.Camera.from_scene(f.camera), fogMode=2, fogDensity=0.05)
        fp.fogColor[:] = [0.5, 0.6, 0.7]
        assert_close("ortho final fog f%d" % fi, O.final(f.depth, np.ascontiguousarray(g[k + "compose"]), f.direct, fp), g[k + "final_fog2"], 0.0)


def test_env_map_importance_sampling(blue_noise):
    """USE_ENVMAP + importanceSampling — the reference's DEFAULT once the scene has an environment (ssgi.frag:197-216, sampleEquirectProbability
    ssgi_utils.frag:210-225, misHeuristic).  Three things are pinned: (1) the host's CPU pass (rfx_amd.envmap.build_importance) equals the
    tables the reference's OWN worker code produced (run by node in make_golden); (2) the implicit-LOD `texture(info.map, uv)` at an unrelated
    uv per pixel, as llvmpipe resolves it (one lod per quad, from its top-left pixel, linear-mantissa log2 of rho^2); (3) K1's MIS arithmetic."""
    from rfx_amd.envmap import build_importance
    g = G.load(G.GOLDEN_ENVMIS)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    envimg = np.ascontiguousarray(g["environment"])
    env = O.EnvMap(envimg, half=True, rtz=True)
    mw, cw, tot = build_importance(env.level(0))  # the half-float texels, as the worker sees them after fromHalfFloat
    assert np.array_equal(mw, g["marginalWeights"]) and np.array_equal(cw, g["conditionalWeights"]) and tot == float(g["totalSumValue"])
    env.set_importance(mw, cw, tot)
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, _, _, _ = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        sp.useEnvMap, sp.importanceSampling, sp.envBlur = 1, 1, float(g["envBlur"])
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else np.zeros((H, W, 4), np.float32)
        o = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp, env=env)
        ga, gb = O.unpack_ssgi(g[k + "ssgi"])
        oa, ob = O.unpack_ssgi(o)
        assert_close("envmis ssgi.diffuse f%d" % fi, oa, ga, FLIP["ssgi"] * 2)
        assert_close("envmis ssgi.specular f%d" % fi, ob, gb, FLIP["ssgi"] * 2)
        sp.importanceSampling = 1
        assert_strict("envmis K1 f%d" % fi, lambda: O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp, env=env), g[k + "ssgi"], True)
        assert (o == g[k + "ssgi"]).all(axis=-1).mean() > 0.98
        # MIS changes a good part of the frame with respect to plain environment lighting
        sp.importanceSampling = 0
        assert (O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp, env=env) != o).any(axis=-1).mean() > 0.2


def test_env_map_importance_sampling_through_effect():
    """The DEFAULT options with an environment (importanceSampling: true): keepEnvMapUpdated builds the importance tables on the host
    (EquirectHdrInfoUniform.updateFrom) and switches the define on; the chain reproduces the reference's."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import SSGIEffect

    g = G.load(G.GOLDEN_ENVMIS)
    W, H, nf = int(g["width"]), int(g["height"]), int(g["frames"])
    scene = types.SimpleNamespace(frame=None, environment=dict(data=np.ascontiguousarray(g["environment"])))
    cam = G.camera(g, 0)
    fx = SSGIEffect(None, scene, cam, dict(steps=int(g["steps"]), refineSteps=int(g["refineSteps"]), denoiseIterations=1, width=W, height=H),
                    seeds=dict(ssgi=int(g["ssgi_start"]), denoise=int(g["denoise_start"])))
    r = OracleRenderer(W, H)
    for fi in range(nf):
        scene.frame = G.frame(g, fi)
        for kk, vv in vars(G.camera(g, fi)).items():
            setattr(cam, kk, vv)
        fx.update(r, None)
        names = [c[0] for c in r.calls]
        assert (names.count("set_environment"), names.count("set_environment_importance")) == ((1, 1) if fi == 0 else (0, 0))
        r.calls.clear()
        assert fx.ssgiPass.uniforms.importanceSampling == 1
        ga, gb = O.unpack_ssgi(g["f%d_ssgi" % fi])
        oa, ob = O.unpack_ssgi(r.tex[abi.TEX_SSGI])
        lim = 0.03 * (fi + 1)
        assert_close("envmis effect ssgi.specular f%d" % fi, ob, gb, lim)
        assert_close("envmis effect compose f%d" % fi, r.tex[abi.TEX_COMPOSE], g["f%d_compose" % fi], lim)
    fx.importanceSampling = False  # the reactive option: the environment is re-examined, the define goes away
    fx.update(r, None)
    assert fx.ssgiPass.uniforms.importanceSampling == 0 and fx.ssgiPass.uniforms.useEnvMap == 1


def _unstable(fn, half, H, W, seeds=24):
    """(H, W) bool: the oracle proves the pixel may flip — discontinuity margin < 1, or its output leaves the tolerance when the
    oracle's primitives are perturbed within the reference GL's measured error (tests/parity.py, oracle/rfx_oracle.c)."""
    from parity import UNSTABLE_TOL_SCALE, out_of_tolerance

    def flat(outs):
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        parts = []
        for o in outs:
            o = np.ascontiguousarray(o)
            parts.append(O.half_bits_to_float(o.view(np.uint16)) if o.dtype in (np.uint16, np.uint32) else o)
        return np.concatenate(parts, axis=-1)

    with O.margins(H, W) as mm:
        base = flat(fn())
    u = mm.plane < 1.0
    for seed in range(1, seeds + 1):
        with O.perturbation(seed):
            u |= out_of_tolerance(flat(fn()), base, half, UNSTABLE_TOL_SCALE)
    return u


@pytest.mark.parametrize("name", G.GOLDENS)
def test_stagewise_strict_metric_every_flip_proven(name, blue_noise):
    """The golden vectors again, under the round-2 metric (tests/parity.py `strict`): absolute 1e-3 — or adjacent binary16 values for the
    half-stored targets, 1e-5 relative for fp32 ones — and EVERY out-of-tolerance pixel proven unstable by the oracle itself.  No flip
    fraction is taken on trust here: `unexplained == 0` on every stage of every frame."""
    from parity import strict
    g = G.load(name)
    W, H, nf, it = int(g["width"]), int(g["height"]), int(g["frames"]), int(g["denoiseIterations"])
    zero16 = np.zeros((H, W, 4), np.uint16)
    h8 = lambda o: O.half_bits_to_float(np.ascontiguousarray(o).view(np.uint16))  # noqa: E731
    reports = []
    for fi in range(nf):
        f = G.frame(g, fi)
        k, kp = "f%d_" % fi, "f%d_" % (fi - 1)
        sp, tp, dp, cp = stage_params(g, fi, 0.0 if fi == 0 else 1.0)
        hist = np.ascontiguousarray(g[kp + "compose"]) if fi else np.zeros((H, W, 4), np.float32)
        k1 = lambda: O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)  # noqa: E731
        reports.append(strict(name + " K1 f%d" % fi, h8(k1()), h8(g[k + "ssgi"]), explainable=_unstable(k1, True, H, W), half=True))

        B = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else zero16 for j in range(2)]
        T0 = [np.ascontiguousarray(g[kp + "temporal%d" % j]) if fi else np.zeros((H, W, 4), np.float32) for j in range(2)]
        ssgi_tex = np.ascontiguousarray(g[k + "ssgi"])

        def k2():
            T = [t.copy() for t in T0]
            O.temporal(ssgi_tex, f.velocity, B[0], B[1], tp, T[0], T[1])
            return T
        u2, T = _unstable(k2, False, H, W), k2()
        for j in range(2):
            reports.append(strict(name + " K2.%d f%d" % (j, fi), T[j], g[k + "temporal%d" % j], explainable=u2, half=False))

        if it == 1:  # (with more iterations only the last pass's targets survive in the golden file)
            A0 = [np.ascontiguousarray(g[kp + "A%d" % j]) if fi else zero16.copy() for j in range(2)]
            Tin = [np.ascontiguousarray(g[k + "temporal%d" % j]) for j in range(2)]
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][0]), 1, 0

            def k3a():
                A = [a.copy() for a in A0]
                O.denoise(f.depth, f.gbuffer, Tin[0], Tin[1], blue_noise, dp, A[0], A[1])
                return A
            ua, A = _unstable(k3a, True, H, W), k3a()
            for j in range(2):
                reports.append(strict(name + " K3p0.%d f%d" % (j, fi), h8(A[j]), h8(g[k + "A%d" % j]), explainable=ua, half=True))
            Ain = [np.ascontiguousarray(g[k + "A%d" % j]) for j in range(2)]
            B0 = [np.ascontiguousarray(g[kp + "B%d" % j]) if fi else zero16.copy() for j in range(2)]
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = int(g[k + "denoise_index"][1]), 0, 1

            def k3b():
                Bn = [b.copy() for b in B0]
                O.denoise(f.depth, f.gbuffer, Ain[0], Ain[1], blue_noise, dp, Bn[0], Bn[1])
                return Bn
            ub, Bn = _unstable(k3b, True, H, W), k3b()
            for j in range(2):
                reports.append(strict(name + " K3p1.%d f%d" % (j, fi), h8(Bn[j]), h8(g[k + "B%d" % j]), explainable=ub, half=True))

        Bc = [np.ascontiguousarray(g[k + "B%d" % j]) for j in range(2)]

        def k4():
            comp = hist.copy()
            O.compose(f.depth, f.gbuffer, Bc[0], Bc[1], cp, comp)
            return comp
        reports.append(strict(name + " K4 f%d" % fi, k4(), g[k + "compose"], explainable=_unstable(k4, False, H, W), half=False))
    for r in reports:
        print(r.line())
    assert all(r.unexplained == 0 for r in reports), "\n".join(r.line() for r in reports if r.unexplained)
    assert all(r.bad <= 0.01 * r.pixels for r in reports)


def _within(a, b):
    """the fp32-output rule of tests/parity.py on raw arrays: |a - b| <= 1e-3 or <= 1e-5 |b|"""
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return (e <= 1e-3) | (e <= 1e-5 * np.abs(b))


@pytest.mark.parametrize("mipmaps", [False, True])
def test_cube_to_equirect_vs_golden(mipmaps):
    """CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42) — scene.environment given as a CubeTexture — against the pass's
    own GLSL on llvmpipe: seamless bilinear cube lookups (edges from the neighbouring face, corners as the average of the three texels that
    exist), and for three's default CubeTexture (LinearMipmapLinearFilter) glGenerateMipmap's chain and the GL's per-pixel cube level of
    detail.  No out-of-tolerance texel on an HDR cube with a sun (values to 40) that spills over three faces."""
    g = G.load("cube_32")
    S, W, H = int(g["size"]), int(g["width"]), int(g["height"])
    want = g["equirect_mipmapped" if mipmaps else "equirect_linear"]
    got = O.cube_to_equirect(g["faces"], W, H, mipmaps=mipmaps)
    ok = _within(got, want)
    e = np.abs(got - want)
    print("cube %d -> %dx%d mipmaps %d: max |err| %.3e, bit-identical %.3f" % (S, W, H, mipmaps, e.max(), (e == 0).mean()))
    assert ok.all(), "%d texels out of tolerance, max %.3e" % (int((~ok).any(-1).sum()), e.max())
    if mipmaps:  # the chain matters: the level-0-only lookup is NOT the mipmapped result near the face edges
        assert not _within(O.cube_to_equirect(g["faces"], W, H, mipmaps=False), want).all()


def test_cube_environment_through_the_effect():
    """SSGIEffect.keepEnvMapUpdated with a CubeTexture (SSGIEffect.js:316-321): converted once through CubeToEquirectEnvPass at
    generateEquirectEnvMap's size, continues as a FloatType equirectangular DataTexture (mip chain + importance tables built from it)."""
    import types
    from oracle_renderer import OracleRenderer
    from rfx_amd.effect import FloatType, LinearFilter, NearestFilter, SSGIEffect

    gc = G.load("cube_32")
    g = G.load(G.GOLDEN_ENVMIS)
    W, H = int(g["width"]), int(g["height"])
    cube = dict(isCubeTexture=True, faces=np.ascontiguousarray(gc["faces"]))  # three's defaults: LinearMipmapLinearFilter + generateMipmaps
    scene = types.SimpleNamespace(frame=G.frame(g, 0), environment=cube)
    cam = G.camera(g, 0)
    fx = SSGIEffect(None, scene, cam, dict(steps=4, refineSteps=1, denoiseIterations=1, width=W, height=H), seeds=dict(ssgi=1, denoise=2))
    r = OracleRenderer(W, H)
    fx.update(r, None)
    calls = dict((c[0], c[1]) for c in r.calls if c[0] in ("cube_to_equirect", "set_environment", "set_environment_importance"))
    assert calls["cube_to_equirect"] == ((6, 32, 32, 4), int(gc["width"]), int(gc["height"]), True)
    assert calls["set_environment"] == (int(gc["height"]), int(gc["width"]), 4) and "set_environment_importance" in calls
    assert fx.ssgiPass.uniforms.useEnvMap == 1 and fx.ssgiPass.uniforms.importanceSampling == 1
    # the environment the device holds is the pass's output, stored FloatType (no half rounding of level 0)
    assert _within(r.env.level(0).reshape(int(gc["height"]), int(gc["width"]), 4), gc["equirect_mipmapped"]).all()
    r.calls.clear()
    fx.update(r, None)  # same texture object: nothing is converted or uploaded again
    assert not [c for c in r.calls if c[0] in ("cube_to_equirect", "set_environment")]
    # HDRCubeTextureLoader's set-up: LinearFilter, no chain
    scene.environment = dict(isCubeTexture=True, faces=cube["faces"], minFilter=LinearFilter, generateMipmaps=False)
    fx.update(r, None)
    assert [c[1][3] for c in r.calls if c[0] == "cube_to_equirect"] == [False]
    scene.environment = dict(isCubeTexture=True, faces=cube["faces"], minFilter=NearestFilter)
    with pytest.raises(NotImplementedError):
        fx.update(r, None)


@pytest.mark.parametrize("mipmaps", [False, True])
def test_cube_to_equirect_properties(mipmaps):
    """Size-independent properties of the cube lookup: a constant cube converts to that constant EXACTLY (the seamless weights — edges,
    corners, two mip levels — always sum to one in fp32 lerp form), per-face constants come back as values inside the hull of the faces a
    footprint can touch, and the six axis directions return the centre of their faces."""
    for S, W, H in ((1, 8, 4), (4, 32, 16), (64, 256, 128), (6, 50, 26)):
        if mipmaps and S & (S - 1):
            continue
        c = np.empty((6, S, S, 4), np.float32)
        c[...] = np.array([0.3, 7.25, 1e-3, 1.0], np.float32)
        out = O.cube_to_equirect(c, W, H, mipmaps=mipmaps)
        assert (out == c[0, 0, 0]).all(), (S, np.abs(out - c[0, 0, 0]).max())
        ids = np.zeros((6, S, S, 4), np.float32)
        ids[..., 0] = np.arange(6, dtype=np.float32)[:, None, None]
        out = O.cube_to_equirect(ids, W, H, mipmaps=mipmaps)[..., 0]
        assert out.min() >= 0.0 and out.max() <= 5.0
        # the pass's direction at vUv: the centre row looks at the horizon, column u = 0.25 / 0.5 / 0.75 / 0 at -Z... check through +Y / -Y rows
        assert np.allclose(out[-1], 2.0, atol=0.51) and np.allclose(out[0], 3.0, atol=0.51)  # top rows see +Y (dir.y = -cos(lat) -> +1 at v = 1), bottom rows -Y
