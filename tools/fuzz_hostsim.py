#!/usr/bin/env python3
"""Differential fuzzing of the kernels' logic in the build container: random frame sizes (odd, tiny, portrait), option values and row tilings; every stage of the chain on the host simulator (tests/hostsim: the kernel sources compiled for x86) against the C restatement
on the same inputs.  Both sides use libm-grade primitives, so they agree to a handful of flipped pixels: a stage that differs on more
than a fraction of a percent of its pixels is a logic bug (apron, halo, launch shape, an option the kernel and the oracle read differently).

    make -C tests/hostsim && python tools/fuzz_hostsim.py --lib tests/hostsim/_build/librfx_hostsim.so [--n 200] [--seed 1]

--device: the same cases against the PRODUCT library on an MI355X (the kernels as compiled for gfx950 at the odd sizes no test pins: 1-pixel
frames, portrait, tiles as thin as the halo; the device's v_exp / v_log / v_rcp against libm flip a few more pixels than the simulator does —
same limits, a fraction of a percent plus two pixels).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stagewise as S  # noqa: E402
import rfx_oracle as O  # noqa: E402
from parity import compare  # noqa: E402
from rfx_amd import abi  # noqa: E402
from rfx_amd.context import Context, load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_frame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--device", action="store_true", help="run against librfx_hip.so on the GPU instead of the simulator")
ap.add_argument("--lib", default=None, help="tests/hostsim/_build/librfx_hostsim.so (not needed in a child of pytest --hostsim)")
a = ap.parse_args()
if a.lib:
    abi.set_library_path(a.lib)
assert a.device != hasattr(abi.load_library(), "rfx_hostsim_build"), "run with --device on a GPU box, or with --lib tests/hostsim/_build/librfx_hostsim.so (or under pytest --hostsim's child environment)"
rng = np.random.RandomState(a.seed)
blue = load_blue_noise_table()
fails = nchecks = 0
t0 = time.time()
for it in range(a.n):
    W = int(rng.choice([rng.randint(1, 40), rng.randint(40, 200), 64, 128, 65, 127, 191]))
    H = int(rng.choice([rng.randint(1, 24), rng.randint(24, 120), 8, 9, 72]))
    steps, refine = int(rng.randint(1, 25)), int(rng.randint(0, 7))
    radius = float(rng.choice([0.0, 1.0, 2.5, 3.0, 5.0, 9.0]))
    uvm = str(rng.choice(["ideal", "reference_gl"]))
    opt = dict(missed=int(rng.randint(2)), direct=int(rng.randint(2)), dist=float(rng.choice([0.5, 3.0, 10.0, 40.0])), thick=float(rng.choice([0.1, 1.0, 10.0])),
               nci=float(rng.choice([0.0, 0.5, 1.0])), conf=float(rng.choice([0.5, 0.75, 4.0])), maxBlend=float(rng.choice([0.5, 0.9, 1.0])), full=int(rng.randint(2)),
               phi=float(rng.choice([0.1, 0.5, 2.0])), lumaPhi=float(rng.choice([0.5, 5.0, 20.0])), depthPhi=float(rng.choice([0.5, 2.0, 10.0])),
               normalPhi=float(rng.choice([5.0, 50.0])), rtz=int(rng.randint(2)))
    cfg = dict(W=W, H=H, steps=steps, refine=refine, radius=radius, uv=uvm, **opt)
    try:
        f0, f1 = synthetic_frame(W, H, 0), synthetic_frame(W, H, 1)
        ora = S.OracleStages(W, H, blue)
        z16, zf = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.float32)
        with O.uv_model({"ideal": "ideal", "reference_gl": "reference"}[uvm]):
            # oracle: two frames of the chain (each stage fed the oracle's own previous outputs)
            state = dict(hist=zf.copy(), B=[z16.copy(), z16.copy()], T=[zf.copy(), zf.copy()])
            outs = []
            prev_cam, keep = None, 0.0
            for fi, f in enumerate((f0, f1)):
                ora.frame(f)
                sp, tp, dp, cp = S.stage_params(f.camera, prev_cam or f.camera, keep, steps, refine)
                sp.blueNoiseIndex = 100 + fi
                sp.missedRays, sp.useDirectLight, sp.rayDistance, sp.thickness = opt["missed"], opt["direct"], opt["dist"], opt["thick"]
                tp.neighborhoodClampIntensity, tp.confidencePower, tp.maxBlend, tp.fullAccumulate = opt["nci"], opt["conf"], opt["maxBlend"], opt["full"]
                dp.phi, dp.lumaPhi, dp.depthPhi, dp.normalPhi, dp.halfStoreRTZ = opt["phi"], opt["lumaPhi"], opt["depthPhi"], opt["normalPhi"], opt["rtz"]
                dp.radius = radius
                k1 = ora.ssgi(state["hist"], sp)
                T = ora.temporal(k1, state["B"], state["T"], tp)
                dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 200 + 2 * fi, 1, 0
                A = ora.denoise(T, [z16.copy(), z16.copy()], dp)
                dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 201 + 2 * fi, 0, 1
                Bn = ora.denoise(A, state["B"], dp)
                comp = ora.compose(Bn, state["hist"], cp)
                outs.append(dict(sp=abi.SsgiParams.from_buffer_copy(sp), tp=abi.TemporalParams.from_buffer_copy(tp), dp_r=radius, cp=abi.ComposeParams.from_buffer_copy(cp),
                                 hist=state["hist"], B=state["B"], T=state["T"], k1=k1, Tn=T, A=A, Bn=Bn, comp=comp, f=f, fi=fi))
                state = dict(hist=comp, B=Bn, T=T)
                prev_cam, keep = f.camera, 1.0
        # simulator: the same stages on the same inputs
        ctx = Context(W, H)
        ctx.set_uv_model(uvm)
        hip = S.HipStages.__new__(S.HipStages)
        hip.ctx = ctx
        for o in outs:
            f = o["f"]
            ctx.upload_frame(f)
            dp = S.stage_params(f.camera, f.camera, 1.0, steps, refine)[2]
            dp.phi, dp.lumaPhi, dp.depthPhi, dp.normalPhi, dp.halfStoreRTZ = opt["phi"], opt["lumaPhi"], opt["depthPhi"], opt["normalPhi"], opt["rtz"]
            dp.radius = radius
            k1 = hip.ssgi(o["hist"], o["sp"])
            T = hip.temporal(o["k1"], o["B"], o["T"], o["tp"])
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 200 + 2 * o["fi"], 1, 0
            A = hip.denoise(o["Tn"], [z16.copy(), z16.copy()], dp)
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 201 + 2 * o["fi"], 0, 1
            Bn = hip.denoise(o["A"], o["B"], dp)
            comp = hip.compose(o["Bn"], o["hist"], o["cp"])
            h8 = lambda t: O.half_bits_to_float(np.ascontiguousarray(t).view(np.uint16))  # noqa: E731
            for name, g, w, lim in (("K1", h8(k1), h8(o["k1"]), 5e-3), ("K2.0", T[0], o["Tn"][0], 5e-3), ("K2.1", T[1], o["Tn"][1], 5e-3),
                                    ("K3a.0", h8(A[0]), h8(o["A"][0]), 1e-2), ("K3a.1", h8(A[1]), h8(o["A"][1]), 1e-2),
                                    ("K3b.0", h8(Bn[0]), h8(o["Bn"][0]), 1e-2), ("K3b.1", h8(Bn[1]), h8(o["Bn"][1]), 1e-2), ("K4", comp, o["comp"], 5e-3)):
                frac, mx = compare(g, w)
                nchecks += 1
                if frac > lim + 2.0 / (W * H):
                    fails += 1
                    print("MISMATCH %s frame %d: %.3f%% of pixels (in-tolerance max %.2e)  cfg %s" % (name, o["fi"], 100 * frac, mx, cfg), flush=True)
        assert ctx.halo_violations() == 0, "halo violations"
        ctx.close()
        # ... and cut into row tiles (ragged, as thin as the halo allows): every tile's own rows, each stage fed the oracle's whole-frame
        # outputs of the previous stage — what a row-tiled run holds after its exchanges
        from rfx_amd import tiling
        ntiles = int(rng.choice([2, 3, 5]))
        vmax = max(float(np.abs(o["f"].velocity[..., 1].view(np.float32)).max()) for o in outs)
        halo = tiling.required_halo(radius, vmax, H, W)
        split = tiling.split_rows(H, ntiles) if H >= 2 * ntiles else []
        if split and min(r for _, r in split) >= max(halo, 1):
            for (y0, rows) in split:
                c = Context(W, H, tile_y0=y0, tile_rows=rows, halo_rows=halo)
                c.set_uv_model(uvm)

                def up(tex, full):
                    h0, hn = c.held_rows(tex)
                    c.upload(tex, full[h0:h0 + hn], h0, hn)

                def own(tex):
                    return c.download(tex, y0, rows)
                for o in outs:
                    f = o["f"]
                    c.upload_frame(f)
                    dp = S.stage_params(f.camera, f.camera, 1.0, steps, refine)[2]
                    dp.phi, dp.lumaPhi, dp.depthPhi, dp.normalPhi, dp.halfStoreRTZ = opt["phi"], opt["lumaPhi"], opt["depthPhi"], opt["normalPhi"], opt["rtz"]
                    dp.radius = radius
                    up(abi.TEX_COMPOSE, o["hist"])
                    c.ssgi_march(o["sp"])
                    k1 = own(abi.TEX_SSGI)
                    up(abi.TEX_SSGI, o["k1"]); up(abi.TEX_DENOISE_B0, o["B"][0]); up(abi.TEX_DENOISE_B1, o["B"][1])
                    up(abi.TEX_TEMPORAL0, o["T"][0]); up(abi.TEX_TEMPORAL1, o["T"][1])
                    c.temporal_reproject(o["tp"])
                    T = [own(abi.TEX_TEMPORAL0), own(abi.TEX_TEMPORAL1)]
                    up(abi.TEX_TEMPORAL0, o["Tn"][0]); up(abi.TEX_TEMPORAL1, o["Tn"][1]); up(abi.TEX_DENOISE_A0, z16); up(abi.TEX_DENOISE_A1, z16)
                    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 200 + 2 * o["fi"], 1, 0
                    c.poisson_denoise(dp)
                    A = [own(abi.TEX_DENOISE_A0), own(abi.TEX_DENOISE_A1)]
                    up(abi.TEX_DENOISE_A0, o["A"][0]); up(abi.TEX_DENOISE_A1, o["A"][1]); up(abi.TEX_DENOISE_B0, o["B"][0]); up(abi.TEX_DENOISE_B1, o["B"][1])
                    dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 201 + 2 * o["fi"], 0, 1
                    c.poisson_denoise(dp)
                    Bn = [own(abi.TEX_DENOISE_B0), own(abi.TEX_DENOISE_B1)]
                    up(abi.TEX_DENOISE_B0, o["Bn"][0]); up(abi.TEX_DENOISE_B1, o["Bn"][1]); up(abi.TEX_COMPOSE, o["hist"])
                    c.compose(o["cp"])
                    comp = own(abi.TEX_COMPOSE)
                    sl = slice(y0, y0 + rows)
                    for name, g, w, lim in (("K1", h8(k1), h8(o["k1"][sl]), 5e-3), ("K2.0", T[0], o["Tn"][0][sl], 5e-3), ("K2.1", T[1], o["Tn"][1][sl], 5e-3),
                                            ("K3a.0", h8(A[0]), h8(o["A"][0][sl]), 1e-2), ("K3b.0", h8(Bn[0]), h8(o["Bn"][0][sl]), 1e-2),
                                            ("K3b.1", h8(Bn[1]), h8(o["Bn"][1][sl]), 1e-2), ("K4", comp, o["comp"][sl], 5e-3)):
                        frac, mx = compare(g, w)
                        nchecks += 1
                        if frac > lim + 2.0 / (W * rows):
                            fails += 1
                            print("MISMATCH tile [%d,+%d) halo %d %s frame %d: %.3f%% of pixels  cfg %s" % (y0, rows, halo, name, o["fi"], 100 * frac, cfg), flush=True)
                if c.halo_violations():
                    fails += 1
                    print("HALO VIOLATIONS %d tile [%d,+%d) halo %d cfg %s" % (c.halo_violations(), y0, rows, halo, cfg), flush=True)
                c.close()
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("ERROR %r cfg %s" % (e, cfg), flush=True)
    if (it + 1) % 20 == 0:
        print("... %d / %d cases, %d problems, %.0f s" % (it + 1, a.n, fails, time.time() - t0), flush=True)
print("%d cases, %d stage comparisons, %d problems" % (a.n, nchecks, fails))
sys.exit(1 if fails else 0)
