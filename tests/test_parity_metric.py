"""CPU: the strict parity metric and the oracle's two flip-proof mechanisms (tests/parity.py, oracle/rfx_oracle.c)."""
import numpy as np

import rfx_oracle as O
import stagewise as S
from parity import out_of_tolerance, strict
from rfx_amd.scene import synthetic_frame, synthetic_frame_parallel


def test_metric_half_and_float_rules():
    b = np.array([[[0.5, 3.0, 100.0, 0.0]]], np.float32)
    a = b.copy()
    a[0, 0, 1] = np.float32(np.float16(3.0)) + np.float32(2 ** -9)  # the adjacent binary16 above 3.0 is 2^-9 away (> 1e-3)
    assert not out_of_tolerance(a, b, half=True).any()
    assert out_of_tolerance(a, b, half=False).any()
    a = b.copy()
    a[0, 0, 2] = 100.0008  # 8e-6 relative on a large radiance: fp32 noise, accepted for float outputs only up to 1e-5 relative
    assert not out_of_tolerance(a, b, half=False).any()
    a[0, 0, 2] = 100.01
    assert out_of_tolerance(a, b, half=False).any()
    a = b.copy()
    a[0, 0, 0] = 0.5 + 2e-3
    r = strict("t", a, b, explainable=np.zeros((1, 1), bool))
    assert (r.bad, r.explained, r.unexplained) == (1, 0, 1) and abs(r.linf_abs - 2e-3) < 1e-6
    r = strict("t", a, b, explainable=np.ones((1, 1), bool))
    assert (r.bad, r.explained, r.unexplained, r.at_risk) == (1, 1, 0, 1)


def test_parallel_dump_equals_serial():
    a, b = synthetic_frame(160, 90, 1), synthetic_frame_parallel(160, 90, 1, workers=4)
    for k in ("depth", "gbuffer", "velocity", "direct"):
        assert (getattr(a, k) == getattr(b, k)).all(), k


def test_oracle_margins_and_perturbation(blue_noise):
    """Unperturbed runs are reproducible; a perturbed run moves only a small unstable population; rim samples of the GGX VNDF sampler
    (blueNoise.r == 1: sqrt(1 - t1^2 - t2^2) cancels, h = normalize(v + l) with v + l -> 0) are flagged by the margin."""
    W, H = 160, 90
    f = synthetic_frame(W, H, 0)
    sp, tp, dp, cp = S.stage_params(f.camera, f.camera, 0.0, 12, 3)
    sp.blueNoiseIndex = 7
    hist = np.zeros((H, W, 4), np.float32)
    with O.margins(H, W) as m:
        base = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
    again = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
    assert (base == again).all()
    h8 = lambda o: O.half_bits_to_float(np.ascontiguousarray(o).view(np.uint16))  # noqa: E731
    with O.perturbation(3):
        p = O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp)
    moved = out_of_tolerance(h8(p), h8(base), True)
    assert 0 < moved.mean() < 0.08
    assert (O.ssgi(f.depth, f.gbuffer, f.direct, hist, blue_noise, sp) == base).all()  # the perturbation is disarmed again
    # blue-noise red == 255 on a geometry pixel -> margin < 1
    import ctypes as C
    rim = 0
    for y in range(H):
        for x in range(W):
            if f.depth[y, x] == 1.0:
                continue
            out = (C.c_float * 4)()
            O.lib().rfxo_blue_noise(blue_noise.ctypes.data_as(C.c_void_p), x, y, 7, out)
            if out[0] == 1.0:
                rim += 1
                assert m.plane[y, x] < 1.0, (y, x, m.plane[y, x])
    assert rim > 0
