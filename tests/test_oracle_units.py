"""CPU unit/property tests of the oracle's building blocks (codec round trips, half rounding
modes, clamp-to-edge indexing, blue-noise hash) — SURVEY.md §4 item 5."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import rfx_oracle as O
from rfx_amd import scene


@settings(max_examples=300, deadline=None)
@given(st.floats(width=32, allow_nan=False, allow_infinity=False, min_value=-70000, max_value=70000))
def test_half_rne_matches_numpy(x):
    L = O.lib()
    want = np.float32(x).astype(np.float16).view(np.uint16)
    assert L.rfxo_f2h_rne(C.c_float(x)) == int(want)


def test_half_rounding_modes_known_answers():
    """SURVEY.md Appendix C-2/C-3 probe values."""
    L = O.lib()
    f = lambda v: C.c_float(v)
    assert L.rfxo_f2h_rne(f(1 + 1.5 * 2 ** -11)) == 0x3C01 and L.rfxo_f2h_rne(f(1 + 2 ** -11)) == 0x3C00 and L.rfxo_f2h_rne(f(70000.0)) == 0x7C00
    assert L.rfxo_f2h_rtz(f(1 + 1.5 * 2 ** -11)) == 0x3C00 and L.rfxo_f2h_rtz(f(1 + 3 * 2 ** -11)) == 0x3C01
    assert L.rfxo_f2h_rtz(f(65520.0)) == 0x7BFF and L.rfxo_f2h_rtz(f(65536.0)) == 0x7BFF  # saturates at 65504
    for h in (0x0001, 0x03FF, 0x0400, 0x3C00, 0x7BFF, 0x8001, 0xFBFF):
        v = L.rfxo_h2f(h)
        assert L.rfxo_f2h_rne(f(v)) == h and L.rfxo_f2h_rtz(f(v)) == h


def test_nearest_clamp_to_edge_index():
    """Appendix C-4: NaN, +-1e30 and |u*W| >= 2^31 give texel 0; u = 5.0 gives the last texel."""
    L = O.lib()
    f = lambda v: C.c_float(v)
    W = 8
    for u in (1e30, -1e30, float("nan"), 3e9 / 8):
        assert L.rfxo_nearest_idx(f(u), W) == 0
    assert L.rfxo_nearest_idx(f(5.0), W) == W - 1
    assert L.rfxo_nearest_idx(f(-0.01), W) == 0
    assert [L.rfxo_nearest_idx(f((i + 0.5) / W), W) for i in range(W)] == list(range(W))


@settings(max_examples=200, deadline=None)
@given(st.floats(0, 1), st.floats(0, 1), st.floats(0, 1), st.floats(0, 0.99), st.floats(0, 1),
       st.floats(-1, 1), st.floats(-1, 1), st.floats(-1, 1))
def test_gbuffer_codec_round_trip(r, g, b, rough, metal, nx, ny, nz):
    """encode side (scene.py, follows packGBuffer) -> decode side (oracle getMaterial): 8-bit colour,
    1/256 roughness, half-float octahedral normal."""
    n = np.array([nx, ny, nz])
    if np.linalg.norm(n) < 1e-3:
        n = np.array([0.0, 1.0, 0.0])
    n = n / np.linalg.norm(n)
    metal = float(metal > 0.5)
    word = np.zeros(4, np.uint32)
    word[0] = scene.vec4_to_float_bits(np.array([[r, g, b, 1.0]], np.float32))[0]
    word[1] = scene.pack_normal(n[None, :])[0]
    word[2] = scene.color2float(np.array([rough], np.float32), np.array([metal], np.float32)).view(np.uint32)[0]
    word[3] = 0
    out = np.zeros(12, np.float32)
    O.lib().rfxo_get_material(word.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert np.abs(out[0:3] - [r, g, b]).max() <= 1 / 255 + 2e-4
    assert np.abs(out[4:7] - n).max() < 4e-3 and abs(np.linalg.norm(out[4:7]) - 1) < 1e-5
    # the reference codes roughness + 257^2 * metalness in ONE float32 (color2float): with metalness = 1 the value exceeds
    # 2^24, odd roughness codes are rounded to even ones (and roughness -> 1 wraps to 0) — reference behaviour, not ours
    rtol = (1 if metal == 0 else 2) / 256 + 2e-4
    assert abs(out[7] - rough) <= rtol and abs(out[8] - metal) <= 1 / 256 + 2e-4
    assert (out[9:12] == 0).all()


def test_rgbe_emissive_round_trip():
    for e in ([2.0, 1.5, 0.5], [0.3, 0.3, 0.9], [7.5, 0.1, 0.0]):
        word = np.zeros(4, np.uint32)
        word[3] = scene.encode_rgbe8_bits(np.array([e], np.float32))[0]
        out = np.zeros(12, np.float32)
        O.lib().rfxo_get_material(word.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        # the codec's -1e-4 NON_ZERO_OFFSET also lands on the EXPONENT byte: exp2(fExp - 0.0255) = -1.75% (reference behaviour)
        assert np.abs(out[9:12] - e).max() <= max(e) * 0.03 + 2e-3


def test_pack_two_vec4_round_trip():
    a = np.array([0.25, 1.5, 3.0, 0.7], np.float32)
    b = np.array([-1.0, -1.0, -1.0, 12.5], np.float32)
    out = np.zeros(4, np.uint32)
    O.lib().rfxo_pack_two_vec4(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    ua, ub = O.unpack_ssgi(out[None, None, :])
    assert np.abs(ua[0, 0] - a).max() < 2e-3 and np.abs(ub[0, 0] - b).max() < 8e-3
    assert ub[0, 0, 0] < 0  # the "not sampled" marker survives the +-1e-4 bias


def test_blue_noise_is_a_toroidal_shift_of_the_table(blue_noise):
    """blue_noise.glsl:31-43: texel (pixel + s.xy % 0x0fffffff) % 128 — one shift per draw index."""
    L = O.lib()
    out = np.zeros(4, np.float32)
    idx = 424242
    L.rfxo_blue_noise(blue_noise.ctypes.data_as(C.c_void_p), 0, 0, idx, out.ctypes.data_as(C.c_void_p))
    base = out.copy()
    hits = np.argwhere((np.abs(blue_noise.astype(np.float32) * np.float32(1 / 255) - base) < 1e-7).all(axis=-1))
    assert len(hits) >= 1
    sy, sx = hits[0]
    for (px, py) in ((5, 9), (127, 127), (300, 200)):
        L.rfxo_blue_noise(blue_noise.ctypes.data_as(C.c_void_p), px, py, idx, out.ctypes.data_as(C.c_void_p))
        want = blue_noise[(sy + py) % 128, (sx + px) % 128].astype(np.float32) * np.float32(1 / 255)
        if len(hits) == 1:
            assert np.abs(out - want).max() < 1e-7
