"""Stage-wise parity of an implementation against the REFERENCE GLSL running live on llvmpipe (oracle/glref), with the
strict metric of tests/parity.py.

The reference chain (GLRefChain: the reference's own fragment shaders under the uniform values of its JS drivers) renders
frame after frame; before every pass its input textures are read back and handed to the implementation under test — the
HIP path through the C ABI (`HipStages`) or the C restatement (`OracleStages`) — so every stage is compared on IDENTICAL
inputs and a flipped pixel of one stage never compounds into the next.  The oracle additionally yields, per stage and
pixel, the discontinuity margin that proves (or refuses) every out-of-tolerance pixel (rfx_oracle.c).

Test infrastructure: imports oracle/; never imported by the product.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "glref"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import rfx_oracle as O  # noqa: E402
from parity import UNSTABLE_TOL_SCALE, out_of_tolerance, strict  # noqa: E402
from rfx_amd import abi  # noqa: E402

M31 = 0x7FFFFFFF


def stage_params(cam, prev_cam, keep, steps, refine):
    """The uniform blocks the reference's drivers set for defaultSSGIOptions (SSGIOptions.js:26-48), camera_moved = True."""
    c, pc = abi.Camera.from_scene(cam), abi.Camera.from_scene(prev_cam)
    sp = abi.SsgiParams(camera=c, steps=steps, refineSteps=refine, mode=0, useDirectLight=1, rayDistance=10, thickness=10, envBlur=0.5, blueNoiseIndex=0)
    tp = abi.TemporalParams(camera=c, prevCamera=pc, textureCount=2, inputType=0, logTransform=1, fullAccumulate=0, confidencePower=0.75,
                            neighborhoodClampIntensity=0.5, maxBlend=1.0, keepData=keep)
    tp.reprojectSpecular[:] = [0, 1]
    tp.neighborhoodClamp[:] = [0, 1]
    dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=2, halfStoreRTZ=1)
    dp.isTextureSpecular[:] = [0, 1]
    cp = abi.ComposeParams(camera=c, inputType=0)
    return sp, tp, dp, cp


class OracleStages:
    """the C restatement as the implementation under test (CPU: pins the oracle itself against the reference)"""
    name = "oracle"

    def __init__(self, W, H, blue):
        self.W, self.H, self.blue = W, H, blue

    def frame(self, f):
        self.f = f

    def ssgi(self, history, sp):
        return O.ssgi(self.f.depth, self.f.gbuffer, self.f.direct, history, self.blue, sp)

    def temporal(self, ssgi, B, T_init, tp):
        T = [t.copy() for t in T_init]
        O.temporal(ssgi, self.f.velocity, B[0], B[1], tp, T[0], T[1])
        return T

    def denoise(self, ins, outs_init, dp):
        outs = [t.copy() for t in outs_init]
        O.denoise(self.f.depth, self.f.gbuffer, ins[0], ins[1], self.blue, dp, outs[0], outs[1])
        return outs

    def compose(self, B, comp_init, cp):
        comp = comp_init.copy()
        O.compose(self.f.depth, self.f.gbuffer, B[0], B[1], cp, comp)
        return comp

    def close(self):
        pass


class HipStages:
    """the product: librfx_hip.so through the C ABI (rfx_amd.context.Context)"""
    name = "hip"

    def __init__(self, W, H, blue=None):
        from rfx_amd.context import Context
        self.ctx = Context(W, H)

    def set_uv_model(self, model):
        self.ctx.set_uv_model(model)

    def frame(self, f):
        self.ctx.upload_frame(f)

    def ssgi(self, history, sp):
        self.ctx.upload(abi.TEX_COMPOSE, history)
        self.ctx.ssgi_march(sp)
        return self.ctx.download(abi.TEX_SSGI)

    def temporal(self, ssgi, B, T_init, tp):
        c = self.ctx
        c.upload(abi.TEX_SSGI, ssgi)
        c.upload(abi.TEX_DENOISE_B0, B[0]); c.upload(abi.TEX_DENOISE_B1, B[1])
        c.upload(abi.TEX_TEMPORAL0, T_init[0]); c.upload(abi.TEX_TEMPORAL1, T_init[1])
        c.temporal_reproject(tp)
        return [c.download(abi.TEX_TEMPORAL0), c.download(abi.TEX_TEMPORAL1)]

    def denoise(self, ins, outs_init, dp):
        c = self.ctx
        if dp.inputIsTemporal:
            c.upload(abi.TEX_TEMPORAL0, ins[0]); c.upload(abi.TEX_TEMPORAL1, ins[1])
        else:
            i0, i1 = (abi.TEX_DENOISE_A0, abi.TEX_DENOISE_A1) if dp.writeToB else (abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1)
            c.upload(i0, ins[0]); c.upload(i1, ins[1])
        o0, o1 = (abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1) if dp.writeToB else (abi.TEX_DENOISE_A0, abi.TEX_DENOISE_A1)
        c.upload(o0, outs_init[0]); c.upload(o1, outs_init[1])
        c.poisson_denoise(dp)
        return [c.download(o0), c.download(o1)]

    def compose(self, B, comp_init, cp):
        c = self.ctx
        c.upload(abi.TEX_DENOISE_B0, B[0]); c.upload(abi.TEX_DENOISE_B1, B[1])
        c.upload(abi.TEX_COMPOSE, comp_init)
        c.compose(cp)
        return c.download(abi.TEX_COMPOSE)

    def close(self):
        assert self.ctx.halo_violations() == 0
        self.ctx.close()


def prove_flips(fn, outs_to_float, bad, half, n_perturb=8, extra_perturb=48):
    """The flip proof of the strict metric for a handful of pixels, without a reference chain: `fn()` re-evaluates a stage on the ORACLE
    (OracleStages), `outs_to_float(outs)` turns its outputs into one (H, W, C) float array, `bad` (H, W) marks the out-of-tolerance pixels.
    Returns the (H, W) bool of pixels the oracle proves unstable: a decision margin < 1, or the output moves by half the tolerance when
    the oracle's primitives are perturbed within the reference GL's measured error (seeded runs).  Only the pixels of `bad` are evaluated."""
    H, W = bad.shape
    unstable = np.zeros((H, W), bool)
    if not bad.any():
        return unstable
    with O.pixel_mask(bad):
        with O.margins(H, W) as mm:
            base = outs_to_float(fn())
        unstable |= bad & (mm.plane < 1.0)
        seed = 0
        while seed < n_perturb + extra_perturb and (bad & ~unstable).any():
            seed += 1
            with O.perturbation(seed):
                unstable |= bad & out_of_tolerance(outs_to_float(fn()), base, half, UNSTABLE_TOL_SCALE)
            if seed >= n_perturb and not (bad & ~unstable).any():
                break
    return unstable


def _h(t):  # RGBA16F target read back as float32 -> the half bit patterns (exactly representable)
    return np.ascontiguousarray(t.read().astype(np.float16).view(np.uint16))


def run(impl_cls, W, H, steps, refine, iterations, frames, blue, frame_fn, ssgi_start=1000, denoise_start=2000, shader_dir=None, log=print,
        with_margins=True, n_perturb=6, sample_every=64, extra_perturb=96, compare_from=0, uv_model="reference_gl", rows=None, compare_only=None, options=None):
    """Returns the list of parity.Report (one per stage output and frame).  frame_fn(i) -> dump frame i (rfx_amd.scene Frame).
    uv_model "reference_gl" (the default of the library and of this harness): the implementation (rfx_set_uv_model / rfxo_set_uv_model)
    evaluates the reference GL's own vUv planes, and the proving oracle then carries no vUv uncertainty at all; "ideal": (i + 0.5) / n on the
    implementation's side, the vUv uncertainty in the proofs.
    rows (y0, y1): every draw still covers the whole frame on both sides, but only that band of rows is compared and proven — what makes an
    8K frame affordable in the default suite (the numpy side of a whole 33 Mpixel stage output costs minutes).
    compare_only: a set of frame numbers — the other frames only advance the reference chain (a long sequence compared at a few ages).
    options: SSGIEffect options other than the defaults, set on the reference chain AND in the implementation's uniform blocks — any of
    distance, thickness, missedRays, radius, phi, lumaPhi, depthPhi, normalPhi, roughnessPhi, specularPhi (tools/fuzz_oracle_vs_gl.py)."""
    with O.uv_model({"ideal": "ideal", "reference_gl": "reference"}[uv_model]):
        return _run(impl_cls, W, H, steps, refine, iterations, frames, blue, frame_fn, ssgi_start, denoise_start, shader_dir, log,
                    with_margins, n_perturb, sample_every, extra_perturb, compare_from, uv_model, rows, compare_only, options)


def _run(impl_cls, W, H, steps, refine, iterations, frames, blue, frame_fn, ssgi_start, denoise_start, shader_dir, log,
         with_margins, n_perturb, sample_every, extra_perturb, compare_from, uv_model, rows=None, compare_only=None, options=None):
    import chain
    options = dict(options or {})
    unknown = set(options) - {"distance", "thickness", "missedRays", "radius", "phi", "lumaPhi", "depthPhi", "normalPhi", "roughnessPhi", "specularPhi"}
    assert not unknown, unknown
    y0, y1 = (0, H) if rows is None else (max(0, int(rows[0])), min(H, int(rows[1])))
    ref = chain.GLRefChain(W, H, blue, shader_dir=shader_dir, steps=steps, refineSteps=refine, denoiseIterations=iterations, **options)
    impl = impl_cls(W, H, blue)
    if hasattr(impl, "set_uv_model"):
        impl.set_uv_model(uv_model)
    ora = OracleStages(W, H, blue) if with_margins else None
    reports = []
    si = di = 0
    prev_cam, keep = None, 0.0

    def as_float(outs):  # a stage's outputs as one (H, W, C) float array (halfs decoded, K1's packed texel as its 8 halfs)
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        parts = []
        for o in outs:
            if o.dtype == np.uint16:
                parts.append(O.half_bits_to_float(o))
            elif o.dtype == np.uint32:
                parts.append(O.half_bits_to_float(np.ascontiguousarray(o).view(np.uint16)))
            else:
                parts.append(o)
        return np.concatenate(parts, axis=-1)

    # the oracle re-evaluates only the out-of-tolerance pixels and a fixed random sample (1 pixel in `sample_every`) of the frame:
    # the sample estimates the at-risk population without 1 + n_perturb whole-frame oracle runs per stage (minutes at 8K)
    sample = np.random.RandomState(12345).rand(H, W) < 1.0 / sample_every
    sample[:y0] = False
    sample[y1:] = False

    def margins_of(fn, half, bad):
        """(explainable (H, W) bool, estimated at-risk pixel count): the oracle proves a pixel unstable — discontinuity margin < 1, or
        its output moves out of tolerance when the oracle's transcendental results are perturbed within the reference GL's measured
        error (n_perturb seeded runs).  Only the pixels of `bad | sample` are evaluated AND compared (an 8K stage output is 0.5-1 GB)."""
        if ora is None:
            return None, None
        diag.clear()
        mask = bad | sample
        sel = np.flatnonzero(mask)

        def picked(outs):  # the selected pixels of a stage's outputs as one (n, C) float32 array
            outs = outs if isinstance(outs, (list, tuple)) else [outs]
            return np.concatenate([as_float(np.ascontiguousarray(o.reshape(H * W, -1)[sel])[None])[0] for o in outs], axis=-1)

        unstable_sel = np.zeros(sel.size, bool)
        with O.pixel_mask(mask):
            with O.margins(H, W) as mm:
                base = picked(fn())
            unstable_sel |= mm.plane.reshape(-1)[sel] < 1.0
            # ... of which: recorded ONLY by the texel-boundary reach of a march / refine tap (rfx_oracle.c margin_tap — the predicate round 4 widened)
            tap_only_sel = unstable_sel & ~(mm.plane_notap.reshape(-1)[sel] < 1.0)
            for seed in range(1, n_perturb + 1):
                with O.perturbation(seed):
                    moved = out_of_tolerance(picked(fn())[None], base[None], half, UNSTABLE_TOL_SCALE)[0]
                    unstable_sel |= moved
                    tap_only_sel &= ~moved  # a perturbed re-evaluation moves it too: not "only" the reach
        # an out-of-tolerance pixel the first seeds did not move gets more draws (random signs per call: a flip that needs one particular
        # combination of signs is found with probability < 1 per seed) — only those few pixels are re-evaluated
        bad_sel = bad.reshape(-1)[sel]
        seed = n_perturb
        while (bad_sel & ~unstable_sel).any() and seed < n_perturb + extra_perturb:
            rest = np.zeros(H * W, bool)
            rest[sel[bad_sel & ~unstable_sel]] = True
            with O.pixel_mask(rest.reshape(H, W)):
                for _ in range(8):
                    seed += 1
                    with O.perturbation(seed):
                        moved = out_of_tolerance(picked(fn())[None], base[None], half, UNSTABLE_TOL_SCALE)[0]
                    unstable_sel |= moved & bad_sel  # (pixels outside `rest` were not re-evaluated: they compare equal)
        unstable = np.zeros(H * W, bool)
        unstable[sel] = unstable_sel
        unstable = unstable.reshape(H, W)
        diag["explained_only_by_tap_reach"] = int((tap_only_sel & bad.reshape(-1)[sel]).sum())
        left = bad & ~unstable
        if left.any():  # diagnostics for the test log: what the oracle itself computes at the pixels nobody could explain
            li = np.flatnonzero(left.reshape(-1))[:6]
            pos = np.searchsorted(sel, li)
            diag["oracle_at_unexplained"] = (np.stack([li // W, li % W], 1), base[pos], mm.plane.reshape(-1)[li])
        n_sample = int(sample.sum())
        return unstable, int(round(float(unstable[sample].sum()) / max(n_sample, 1) * (y1 - y0) * W))

    def bad_of(gots, wants, half):
        gots = gots if isinstance(gots, (list, tuple)) else [gots]
        wants = wants if isinstance(wants, (list, tuple)) else [wants]
        b = np.zeros((H, W), bool)
        for g, w in zip(gots, wants):
            b[y0:y1] |= out_of_tolerance(as_float(g[y0:y1]), as_float(w[y0:y1]), half)
        return b

    diag = {}

    def check(name, got, want, mr, half):
        m, at_risk = mr
        got, want, m = got[y0:y1], want[y0:y1], (None if m is None else m[y0:y1])  # the compared band (the whole frame by default)
        if isinstance(got, np.ndarray) and got.dtype == np.uint16:
            got, want = O.half_bits_to_float(got), O.half_bits_to_float(want)
        r = strict(name, got, want, explainable=m, half=half)
        r.at_risk = at_risk
        # of the unexplained pixels: how many does the implementation compute EXACTLY as the C restatement does (then the open question is
        # the reference GL against the restatement at that pixel — the proof's reach — not the kernel)
        r.unexplained_equal_to_restatement = 0
        r.unexplained_at = []  # (y, x) of the unexplained pixels the oracle was re-run at
        if r.unexplained and "oracle_at_unexplained" in diag:  # (kept for the stage's other outputs: margins_of clears it)
            idx, obase, omargin = diag["oracle_at_unexplained"]
            r.unexplained_at = [(int(y), int(x)) for y, x in idx]
            g, w = as_float(got), as_float(want)
            for k, (y, x) in enumerate(idx):
                c = slice(0, g.shape[-1]) if obase.shape[-1] == g.shape[-1] else slice(0, 0)
                if obase.shape[-1] == g.shape[-1] and np.array_equal(g[y - y0, x], obase[k]):
                    r.unexplained_equal_to_restatement += 1
                log("    unexplained (y %d, x %d) margin %.3g\n      impl   %s\n      ref    %s\n      oracle %s" % (
                    y, x, omargin[k], np.array2string(g[y - y0, x], precision=6), np.array2string(w[y - y0, x], precision=6),
                    np.array2string(obase[k], precision=6)))
        #  how many of the explained out-of-tolerance pixels rest on the tap-reach margin alone: tracked per stage output, printed when not 0
        r.explained_only_by_tap_reach = diag.get("explained_only_by_tap_reach", 0)
        reports.append(r)
        log(r.line() + ("   [%d of the explained only by a march / refine tap's texel-boundary reach]" % r.explained_only_by_tap_reach if r.explained_only_by_tap_reach else ""))
        return r

    for fi in range(frames):
        f = frame_fn(fi)
        ref.upload_frame(f)
        impl.frame(f)
        if ora is not None:
            ora.frame(f)
        sp, tp, dp, cp = stage_params(f.camera, prev_cam or f.camera, keep, steps, refine)
        for k, v in options.items():
            if k in ("distance", "thickness", "missedRays"):
                setattr(sp, {"distance": "rayDistance", "thickness": "thickness", "missedRays": "missedRays"}[k], v)
            else:
                setattr(dp, k, v)
        tag = "f%d " % fi
        if fi < compare_from or (compare_only is not None and fi not in compare_only):  # only advance the reference chain (its state after frame fi is what frame fi + 1 is compared on)
            si = (ssgi_start + si + 1) % M31
            ref.ssgi(f.camera, si)
            ref.temporal(f.camera, camera_moved=True)
            for pi in range(2 * iterations):
                di = (denoise_start + di + 1) % M31
                _one_denoise_pass(ref, f.camera, pi, di)
            ref.compose(f.camera)
            keep, prev_cam = 1.0, f.camera
            continue
        # ---- K1 (history = the reference's composed GI of the previous frame)
        hist = ref.t_compose.read()
        si = (ssgi_start + si + 1) % M31
        sp.blueNoiseIndex = si
        ref.ssgi(f.camera, si)
        R = np.ascontiguousarray(ref.t_ssgi.read().view(np.uint32))
        I = impl.ssgi(hist, sp)
        # the packed texel's 8 halfs as stored (unpackTwoVec4 subtracts the same 1e-4 from both sides)
        I = np.ascontiguousarray(I)
        if rows is None:
            got, want = as_float(I), as_float(R)
        else:  # decode the band only, at its place in a frame-sized array of zeros
            got, want = np.zeros((H, W, 8), np.float32), np.zeros((H, W, 8), np.float32)
            got[y0:y1], want[y0:y1] = as_float(I[y0:y1]), as_float(R[y0:y1])
        r = check(tag + "K1 ssgi", got, want, margins_of(lambda: ora.ssgi(hist, sp), True, bad_of(got, want, True)), half=True)
        r.bit_identical = float((I[y0:y1] == R[y0:y1]).all(axis=-1).mean())
        # ---- K2 (input: the reference's K1 output; history: its K3 target B of the previous frame; targets keep discarded texels)
        B_prev = [_h(t) for t in ref.t_B]
        T_prev = [np.ascontiguousarray(t.read()) for t in ref.t_temporal]
        ref.temporal(f.camera, camera_moved=True)
        RT = [np.ascontiguousarray(t.read()) for t in ref.t_temporal]
        IT = impl.temporal(R, B_prev, T_prev, tp)
        m = margins_of(lambda: ora.temporal(R, B_prev, T_prev, tp), False, bad_of(IT, RT, False))
        for j in range(2):
            check(tag + "K2 temporal%d" % j, IT[j], RT[j], m, half=False)
        keep, prev_cam = 1.0, f.camera
        # ---- K3 passes (PoissonDenoisePass.js:135-149: 2 * iterations draws, ping-pong A/B; pass 0 reads K2's targets)
        idx = []
        for _ in range(2 * iterations):
            di = (denoise_start + di + 1) % M31
            idx.append(di)
        for pi in range(2 * iterations):
            horizontal = pi % 2 == 0
            ins = RT if pi == 0 else ([_h(t) for t in (ref.t_B if horizontal else ref.t_A)])
            outs_init = [_h(t) for t in (ref.t_A if horizontal else ref.t_B)]
            ref.o["denoiseIterations"] = iterations
            _one_denoise_pass(ref, f.camera, pi, idx[pi])
            RO = [_h(t) for t in (ref.t_A if horizontal else ref.t_B)]
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = idx[pi], int(pi == 0), int(not horizontal)
            IO = impl.denoise(ins, outs_init, dp)
            m = margins_of(lambda: ora.denoise(ins, outs_init, dp), True, bad_of(IO, RO, True))
            for j in range(2):
                check(tag + "K3 pass%d tex%d" % (pi, j), IO[j], RO[j], m, half=True)
        # ---- K4 (reads target B — never written when denoiseIterations == 0)
        Bc = [_h(t) for t in ref.t_B]
        comp_prev = np.ascontiguousarray(ref.t_compose.read())
        ref.compose(f.camera)
        RC = np.ascontiguousarray(ref.t_compose.read())
        IC = impl.compose(Bc, comp_prev, cp)
        check(tag + "K4 compose", IC, RC, margins_of(lambda: ora.compose(Bc, comp_prev, cp), False, bad_of(IC, RC, False)), half=False)
    impl.close()
    return reports


def _one_denoise_pass(ref, cam, i, blue_noise_index):
    """draw i of PoissonDenoisePass.render on the reference chain (GLRefChain.denoise runs all of them at once)"""
    p, o = ref.p_denoise, ref.o
    p.sampler("depthTexture", ref.t_depth)
    p.sampler("gBufferTexture", ref.t_gbuffer)
    p.sampler("blueNoiseTexture", ref.t_blue)
    for k in ("radius", "phi", "lumaPhi", "depthPhi", "normalPhi", "roughnessPhi", "specularPhi"):
        p.set(k, float(o[k]))
    p.set("projectionMatrix", cam.projectionMatrix)
    p.set("projectionMatrixInverse", cam.projectionMatrixInverse)
    p.set("cameraMatrixWorld", cam.matrixWorld)
    p.set("viewMatrix", cam.matrixWorldInverse)
    p.set("resolution", [float(ref.W), float(ref.H)])
    p.set("blueNoiseSize", [128.0, 128.0])
    horizontal = i % 2 == 0
    src = ref.t_temporal if i == 0 else (ref.t_B if horizontal else ref.t_A)
    dst = ref.t_A if horizontal else ref.t_B
    p.sampler("inputTexture", src[0])
    p.sampler("inputTexture2", src[1])
    p.set("blueNoiseIndex", int(blue_noise_index))
    p.draw(dst[:2])


def summarize(reports):
    """aggregate per stage kind over frames -> dict kind -> (pixels, linf_all, linf_in_tol, bad, explained, unexplained, at_risk)"""
    agg = {}
    for r in reports:
        kind = r.name.split(" ", 1)[1]
        a = agg.setdefault(kind, [0, 0.0, 0.0, 0, 0, 0, 0])
        a[0] += r.pixels
        a[1] = max(a[1], r.linf_abs)
        a[2] = max(a[2], r.linf_abs_ok)
        a[3] += r.bad
        a[4] += r.explained
        a[5] += r.unexplained
        a[6] += r.at_risk or 0
    return agg
