#!/usr/bin/env python3
"""Differential fuzzing of the VARIANTS through the host's option surface: random SSGIEffect / SSREffect options (mode, denoiseMode, denoise
iterations and radius, the phi's, steps, missedRays, importance sampling, envBlur, resolutionScale), random frame sizes, perspective /
orthographic cameras, with / without scene.environment (HalfFloatType / FloatType) and fog — the effect drives the library (tests/hostsim's build
of the kernel sources, or with --device the product on an MI355X) and the C restatement IN LOCK STEP: before every draw the library's
textures are set to the restatement's (identical inputs, stage-wise), after it the target the draw wrote is compared.  tools/fuzz_hostsim.py
covers the default chain and its row tilings; this one the other specialisations of the same kernels, at the sizes no test pins.

    make -C tests/hostsim && python tools/fuzz_effects.py --lib tests/hostsim/_build/librfx_hostsim.so [--n 100] [--seed 1]
    python tools/fuzz_effects.py --device [--n 300]            # on an MI355X

TEST INFRASTRUCTURE (imports oracle/rfx_oracle.py, the checker)."""
import argparse
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "glref"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, _p)
import rfx_oracle as O  # noqa: E402
from oracle_renderer import OracleRenderer  # noqa: E402
from parity import compare  # noqa: E402
from test_oracle_vs_golden import ssr_unpack  # noqa: E402
from rfx_amd import abi, effect  # noqa: E402
from rfx_amd.context import Context  # noqa: E402
from rfx_amd.scene import synthetic_environment, synthetic_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--frames", type=int, default=3)
ap.add_argument("--self-test", action="store_true", help="hand the LIBRARY a perturbed parameter per draw (K1 rayDistance x 0.5, K2 maxBlend x 0.8, K3 "
                "depthPhi x 2): the run must report mismatches on all three kernels — that the limits can see a wrong kernel at all")
ap.add_argument("--device", action="store_true", help="run against librfx_hip.so on the GPU instead of the simulator")
ap.add_argument("--lib", default=None, help="tests/hostsim/_build/librfx_hostsim.so")
a = ap.parse_args()
if a.lib:
    abi.set_library_path(a.lib)
assert a.device != hasattr(abi.load_library(), "rfx_hostsim_build"), "run with --device on a GPU box, or with --lib tests/hostsim/_build/librfx_hostsim.so"

SYNCED = [t for t in abi.TEX_FORMAT if t != abi.TEX_BLUE_NOISE]


def h8(t):
    return O.half_bits_to_float(np.ascontiguousarray(t).view(np.uint16))


class Lockstep:
    """The renderer the effect sees: every call goes to the library AND to the restatement; a draw runs on identical inputs on both."""

    def __init__(self, W, H, uv, report):
        self.W, self.H = W, H
        self.dev, self.ora = Context(W, H), OracleRenderer(W, H)
        self.dev.set_uv_model(uv)
        self.report = report
        self.draws = 0

    def close(self):
        assert self.dev.halo_violations() == 0
        self.dev.close()

    def held_rows(self, tex):
        return self.dev.held_rows(tex)

    def upload(self, tex, array, row0=None, rows=None):
        self.dev.upload(tex, array, row0, rows)
        self.ora.upload(tex, array, row0, rows)

    def download(self, tex, row0=None, rows=None):
        return self.ora.download(tex, row0, rows)

    def set_environment(self, rgba, half_float_type=True, half_store_rtz=True):
        self.dev.set_environment(rgba, half_float_type=half_float_type, half_store_rtz=half_store_rtz)
        self.ora.set_environment(rgba, half_float_type=half_float_type, half_store_rtz=half_store_rtz)

    def set_environment_importance(self, marginal, conditional, total_sum):
        self.dev.set_environment_importance(marginal, conditional, total_sum)
        self.ora.set_environment_importance(marginal, conditional, total_sum)

    def sync(self):
        self.dev.sync()

    def _both(self, name, call, targets, lim, p=None, mutate=None):
        for t in SYNCED:  # identical inputs — and identical texels where the draw writes nothing (background, outside its target)
            self.dev.upload(t, self.ora.tex[t])
        if a.self_test and mutate:
            q = type(p).from_buffer_copy(p)
            setattr(q, mutate[0], getattr(q, mutate[0]) * mutate[1])
            call(self.dev, q)
        else:
            call(self.dev, p)
        call(self.ora, p)
        self.draws += 1
        for label, tex, view in targets:
            g, w = view(self.dev.download(tex)), view(self.ora.tex[tex])
            frac, mx = compare(g, w)
            npx = max(1, int(np.prod(g.shape[:-1])))
            self.report(name + label, frac, mx, frac > (lim + 2.0 / npx if lim else 0.0))  # (lim 0: a copy, exact)

    def ssgi_march(self, p):
        rs = p.resolutionScale or 1.0
        oH, oW = (int(self.H * rs), int(self.W * rs)) if rs != 1.0 else (self.H, self.W)
        cut = lambda t: np.ascontiguousarray(t).reshape(-1)[:oH * oW * 4].reshape(oH, oW, 4)  # noqa: E731
        view = (lambda t: ssr_unpack(cut(t))) if p.mode == 1 else (lambda t: h8(cut(t)))
        # the environment's contribution is a product of more primitives per texel: the bound the tests use for it is 6x the plain one
        self._both("K1", lambda r, q: r.ssgi_march(q), [("", abi.TEX_SSGI, view)], 2e-2 if p.useEnvMap else 5e-3, p, ("rayDistance", 0.5))

    def temporal_reproject(self, p):
        t = [(".0", abi.TEX_TEMPORAL0, lambda x: x)] + ([(".1", abi.TEX_TEMPORAL1, lambda x: x)] if p.textureCount == 2 else [])
        self._both("K2", lambda r, q: r.temporal_reproject(q), t, 5e-3, p, ("maxBlend", 0.8))

    def poisson_denoise(self, p):
        o = (abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1) if p.writeToB else (abi.TEX_DENOISE_A0, abi.TEX_DENOISE_A1)
        t = [(".0", o[0], h8)] + ([(".1", o[1], h8)] if p.textureCount == 2 else [])
        self._both("K3" + ("t" if p.inputIsTemporal else "n"), lambda r, q: r.poisson_denoise(q), t, 1e-2, p, ("depthPhi", 2.0))

    def copy_framebuffer(self, dst):
        view = h8 if dst == abi.TEX_FBCOPY_F16 else (lambda x: x)
        self._both("copy", lambda r, q: r.copy_framebuffer(dst), [("", dst, view)], 0.0)

    def compose(self, p):
        self._both("K4", lambda r, q: r.compose(q), [("", abi.TEX_COMPOSE, lambda x: x)], 5e-3, p)

    def final_compose(self, p):
        self._both("final", lambda r, q: r.final_compose(q), [("", abi.TEX_FINAL, lambda x: x)], 5e-3, p)


rng = np.random.RandomState(a.seed)
fails = nchecks = ndraws = ntraa = 0
seen, caught = {}, {}
t0 = time.time()
for it in range(a.n):
    W = int(rng.choice([rng.randint(2, 40), rng.randint(40, 200), 64, 65, 127, 192]))
    H = int(rng.choice([rng.randint(2, 24), rng.randint(24, 120), 8, 9, 72]))
    rs = float(rng.choice([1, 1, 0.5, 0.25]))
    if (W * rs) % 1 or (H * rs) % 1:  # the library takes whole W*s x H*s targets only
        rs = 1.0
    opt = dict(mode=str(rng.choice(["ssgi", "ssgi", "ssr"])), denoiseMode=str(rng.choice(["full", "full", "full_temporal", "denoised", "temporal"])),
               denoiseIterations=int(rng.choice([0, 1, 1, 2])), radius=float(rng.choice([0.0, 1.0, 3.0, 3.0, 5.0])),
               phi=float(rng.choice([0.1, 0.5, 2.0])), lumaPhi=float(rng.choice([0.5, 5.0, 20.0])), depthPhi=float(rng.choice([0.5, 2.0, 10.0])),
               normalPhi=float(rng.choice([5.0, 50.0])), roughnessPhi=float(rng.choice([1.0, 50.0])), specularPhi=float(rng.choice([1.0, 50.0])),
               steps=int(rng.randint(1, 25)), refineSteps=int(rng.randint(0, 7)), distance=float(rng.choice([0.5, 3.0, 10.0, 40.0])),
               thickness=float(rng.choice([0.1, 1.0, 10.0])), missedRays=bool(rng.randint(2)), importanceSampling=bool(rng.randint(2)),
               envBlur=float(rng.choice([0.0, 0.1, 0.5, 1.0])), resolutionScale=rs)
    ortho = float(rng.choice([0, 0, 0, 3.2]))
    envkind = str(rng.choice(["none", "none", "half", "float"]))
    fog = str(rng.choice(["none", "none", "linear", "exp2"]))
    uv = str(rng.choice(["ideal", "reference_gl"]))
    rtz = bool(rng.randint(2))
    traa = str(rng.choice(["no", "no", "no", "no", "half", "float"]))  # one case in three: TRAAEffect instead (K2 alone on the composer's buffer, its own framebuffer copy)
    traa_full = bool(rng.randint(2))
    cfg = dict(W=W, H=H, ortho=ortho, env=envkind, fog=fog, uv=uv, rtz=rtz, **opt) if traa == "no" else dict(W=W, H=H, ortho=ortho, uv=uv, rtz=rtz, traa=traa, fullAccumulate=traa_full)
    case_fail = []

    def report(name, frac, mx, bad):
        global nchecks, fails
        nchecks += 1
        seen[name.split(".")[0]] = max(seen.get(name.split(".")[0], 0.0), frac)
        if bad:
            fails += 1
            case_fail.append(name)
            caught[name[:2]] = caught.get(name[:2], 0) + 1
            if a.self_test:
                return
            print("MISMATCH %s: %.3f%% of pixels (in-tolerance max %.2e)  cfg %s" % (name, 100 * frac, mx, cfg), flush=True)
    try:
        kw = dict(ortho_half_height=ortho) if ortho else {}
        frames = [synthetic_frame(W, H, i, **kw) for i in range(a.frames)]
        scene = types.SimpleNamespace(frame=frames[0])
        if envkind != "none":
            scene.environment = dict(data=synthetic_environment(64, 32), type=effect.HalfFloatType if envkind == "half" else effect.FloatType)
        if fog == "linear":
            scene.fog = types.SimpleNamespace(color=(0.3, 0.5, 0.7), near=1.0, far=6.0)
        elif fog == "exp2":
            scene.fog = types.SimpleNamespace(color=(0.6, 0.5, 0.4), density=0.15, isFogExp2=True)
        cam = types.SimpleNamespace(**vars(frames[0].camera))
        R = Lockstep(W, H, uv, report)
        with O.uv_model({"ideal": "ideal", "reference_gl": "reference"}[uv]):
            if traa != "no":
                tx = effect.TRAAEffect(scene, cam, effect.VelocityDepthNormalPass(scene, cam), dict(fullAccumulate=traa_full), half_store_rtz=rtz)
                for f in frames:
                    scene.frame = f
                    for k, v in vars(f.camera).items():
                        setattr(cam, k, v)
                    tx.update(R, dict(texture=dict(type=effect.HalfFloatType if traa == "half" else effect.FloatType), width=W, height=H, data=f.direct))
                frames = []
                ntraa += 1
            fx = effect.SSGIEffect(None, scene, cam, dict(width=W, height=H, **opt), seeds=dict(ssgi=10 + it, denoise=500 + it), half_store_rtz=rtz)
            for f in frames:
                scene.frame = f
                for k, v in vars(f.camera).items():
                    setattr(cam, k, v)
                fx.update(R, None)
                fx.mainImage(R)
        ndraws += R.draws
        R.close()
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR %r cfg %s" % (e, cfg), flush=True)
    if (it + 1) % 20 == 0:
        print("... %d / %d cases, %d draws, %d problems, %.0f s" % (it + 1, a.n, ndraws, fails, time.time() - t0), flush=True)
print("largest out-of-tolerance fraction seen per stage: " + ", ".join("%s %.3f%%" % (k, 100 * v) for k, v in sorted(seen.items())))
print("%d cases (%d of them TRAAEffect), %d draws in lock step, %d target comparisons, %d problems" % (a.n, ntraa, ndraws, nchecks, fails))
if a.self_test:
    print("self-test: draws caught per kernel with the library's parameter perturbed: %s" % caught)
    sys.exit(0 if all(caught.get(k, 0) > 0 for k in ("K1", "K2", "K3")) else 1)
sys.exit(1 if fails else 0)
