"""GPU: rfx_set_uv_model(RFX_UV_REFERENCE_GL) on the device.

  (1) HIP against the reference GLSL live on llvmpipe, stage by stage on identical inputs (tests/stagewise.py), under both vUv models:
      out-of-tolerance pixels per stage (each proven by the oracle or reported UNEXPLAINED);
  (2) HIP against the C restatement under the same model: share of bit-identical texels per stage;
  (3) the chain's kernel times under both models (one warm frame, hipEvent per draw).

    python tools/gpu_runs/uv_model_check.py [--size 480x270] [--frames 3] [--impl hip|oracle] [--time-size 1920x1080]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import stagewise as S  # noqa: E402  (puts realism-effects_amd, oracle, oracle/glref on the path)
import rfx_oracle as O  # noqa: E402
from rfx_amd import abi  # noqa: E402
from rfx_amd.context import load_blue_noise_table  # noqa: E402
from rfx_amd.scene import synthetic_frame, synthetic_frame_parallel  # noqa: E402


def bits_equal(a, b):
    a = [np.ascontiguousarray(x) for x in (a if isinstance(a, (list, tuple)) else [a])]
    b = [np.ascontiguousarray(x) for x in (b if isinstance(b, (list, tuple)) else [b])]
    eq = [(x.view(np.uint8).reshape(x.shape[0], x.shape[1], -1) == y.view(np.uint8).reshape(y.shape[0], y.shape[1], -1)).all(-1) for x, y in zip(a, b)]
    return float(np.mean([e.mean() for e in eq]))


def vs_oracle(W, H, blue, model, frames=2):
    """HIP and the C restatement on the same inputs (each stage fed the ORACLE's previous outputs): bit-identical share per stage."""
    hip, ora = S.HipStages(W, H, blue), S.OracleStages(W, H, blue)
    hip.set_uv_model(model)
    z16, zf = np.zeros((H, W, 4), np.uint16), np.zeros((H, W, 4), np.float32)
    hist, B, T = zf.copy(), [z16.copy(), z16.copy()], [zf.copy(), zf.copy()]
    prev_cam, keep = None, 0.0
    out = []
    with O.uv_model({"ideal": "ideal", "reference_gl": "reference"}[model]):
        for fi in range(frames):
            f = synthetic_frame(W, H, fi)
            hip.frame(f); ora.frame(f)
            sp, tp, dp, cp = S.stage_params(f.camera, prev_cam or f.camera, keep, 20, 5)
            sp.blueNoiseIndex = 1001 + fi
            k1h, k1o = hip.ssgi(hist, sp), ora.ssgi(hist, sp)
            t_h, t_o = hip.temporal(k1o, B, T, tp), ora.temporal(k1o, B, T, tp)
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2001 + 2 * fi, 1, 0
            a_h, a_o = hip.denoise(t_o, [z16.copy(), z16.copy()], dp), ora.denoise(t_o, [z16.copy(), z16.copy()], dp)
            dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2002 + 2 * fi, 0, 1
            b_h, b_o = hip.denoise(a_o, B, dp), ora.denoise(a_o, B, dp)
            c_h, c_o = hip.compose(b_o, hist, cp), ora.compose(b_o, hist, cp)
            out.append((fi, bits_equal(k1h, k1o), bits_equal(t_h, t_o), bits_equal(a_h, a_o), bits_equal(b_h, b_o), bits_equal(c_h, c_o)))
            hist, B, T = c_o, b_o, t_o
            prev_cam, keep = f.camera, 1.0
    hip.close()
    return out


def kernel_times(W, H, f, model, reps=5):
    from rfx_amd.context import Context
    ctx = Context(W, H)
    ctx.set_uv_model(model)
    ctx.upload_frame(f)
    sp, tp, dp, cp = S.stage_params(f.camera, f.camera, 1.0, 20, 5)
    sp.blueNoiseIndex = 1001

    def frame():
        t = {}
        for name, fn in (("K1", lambda: ctx.ssgi_march(sp)), ("K2", lambda: ctx.temporal_reproject(tp))):
            ctx.time_begin(); fn(); t[name] = ctx.time_end()
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2001, 1, 0
        ctx.time_begin(); ctx.poisson_denoise(dp); t["K3p0"] = ctx.time_end()
        dp.blueNoiseIndex, dp.inputIsTemporal, dp.writeToB = 2002, 0, 1
        ctx.time_begin(); ctx.poisson_denoise(dp); t["K3p1"] = ctx.time_end()
        ctx.time_begin(); ctx.compose(cp); t["K4"] = ctx.time_end()
        return t
    frame(); frame()
    ts = [frame() for _ in range(reps)]
    ctx.close()
    return {k: float(np.median([t[k] for t in ts])) for k in ts[0]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="480x270")
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--impl", default="hip")
    ap.add_argument("--time-size", default="1920x1080")
    ap.add_argument("--perturb", type=int, default=8)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    blue = load_blue_noise_table()
    impl = S.HipStages if a.impl == "hip" else S.OracleStages
    for model in ("ideal", "reference_gl"):
        t0 = time.time()
        reports = S.run(impl, W, H, 20, 5, 1, a.frames, blue, lambda i: synthetic_frame(W, H, i), n_perturb=a.perturb, sample_every=16,
                        uv_model=model, log=lambda *_: None)
        print("== %s vs reference GLSL, vUv model %s, %dx%d, %d frames (%.1f s)" % (impl.name, model, W, H, a.frames, time.time() - t0))
        by = {}
        for r in reports:
            k = r.name.split()[1] + (" " + r.name.split()[2] if r.name.split()[1] == "K3" else "")
            e = by.setdefault(k, [0, 0, 0, 0.0])
            e[0] += r.bad; e[1] += r.unexplained; e[2] += r.pixels; e[3] = max(e[3], r.linf_abs_ok)
        for k, (bad, unexp, px, linf) in by.items():
            print("   %-9s out-of-tol %5d of %8d (%.4f %%)  UNEXPLAINED %d  Linf(in-tol) %.3e" % (k, bad, px, 100.0 * bad / px, unexp, linf))
        sys.stdout.flush()
    if a.impl == "hip":
        for model in ("ideal", "reference_gl"):
            print("== hip vs C oracle, both on vUv model %s: bit-identical texels  K1 / K2 / K3p0 / K3p1 / K4" % model)
            for row in vs_oracle(W, H, blue, model):
                print("   f%d  %.6f  %.6f  %.6f  %.6f  %.6f" % row)
            sys.stdout.flush()
        tw, th = (int(v) for v in a.time_size.split("x"))
        tf = synthetic_frame_parallel(tw, th, 0)
        for model in ("ideal", "reference_gl", "ideal", "reference_gl"):
            t = kernel_times(tw, th, tf, model)
            print("== kernel ms at %dx%d, vUv model %-12s " % (tw, th, model) + "  ".join("%s %.4f" % kv for kv in t.items()) + "  sum %.4f" % sum(t.values()))
            sys.stdout.flush()
