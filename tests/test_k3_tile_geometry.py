"""CPU: the geometry behind K3's staged tile (realism-effects_amd/csrc/k3_denoise.hip, rfx_launch_k3).  The launcher sizes the apron of the
LDS tile from the taps' reach and — pass 0 — does not hold the corners of the staged rectangle no tap can address (three workgroups per CU
instead of two at 4K: profiles/r05_k3).  Both are claims about where `rm * (offset / resolution)` (poisson_denoise.frag:183-189)
can land; this test enumerates the taps — every rotation, every Poisson sample, the whole flatness range — and holds the launcher's formulas,
restated here line by line, against what it finds."""
import math

import numpy as np
import pytest

SLACK = 4e-3  # k3_denoise.hip rfx_launch_k3: the rounding of the tap coordinate itself
SQ = 0.25 * 1.41421356237
POISSON = [(-1.0, 0.0), (0.0, -1.0), (1.0, 0.0), (0.0, 1.0), (-SQ, -SQ), (SQ, -SQ), (SQ, SQ), (-SQ, SQ)]  # poisson_denoise.frag:91-92


def launcher_tile(W, H, radius, temporal):
    """(Rx, Ry, skip) as rfx_launch_k3 computes them (k3_apron, the corner shave)."""
    aspect = W / H
    rx, ry = radius * max(1.0, aspect), radius * max(1.0, 1.0 / aspect)

    def apron(r):
        a = int(math.floor(r + 0.5 + SLACK)) if temporal else int(math.floor(r + SLACK)) + 1
        return max(a, 1)

    Rx, Ry = apron(rx), apron(ry)
    skip = 0
    if temporal:
        t = (Ry - 0.5 - SLACK) / ry
        X = int(math.floor(0.5 + rx * math.sqrt(max(0.0, 1.0 - t * t)) + SLACK))
        skip = min(max(Rx - X, 0), 4)
    return Rx, Ry, skip, rx, ry


def tap_texel_offsets(W, H, radius):
    """(column, row) offset of the NEAREST texel of every tap from its pixel: the tap's texture coordinate is vUv + rm * (POISSON[k] / resolution)
    with rm = radius * flatness * mat2(c, -s, s, c) (k3_tiled_body: m00 = rf co, m01 = rf -sn, m10 = rf sn, m11 = rf co;
    nu = u + (m00 ox + m10 oy), nv = v + (m01 ox + m11 oy)); times the resolution, from the pixel centre x + 0.5: texel floor(x + 0.5 + d)."""
    ang = np.linspace(0.0, 2.0 * np.pi, 4096, endpoint=False)[:, None, None]
    flat = np.linspace(0.25, 1.0, 49)[None, :, None]  # flatness = (1 - min(|fwidth n|, 1))^2 * 0.75 + 0.25 (:172-173)
    px = np.array([p[0] for p in POISSON])[None, None, :]
    py = np.array([p[1] for p in POISSON])[None, None, :]
    rf = radius * flat
    co, sn = np.cos(ang), np.sin(ang)
    dx = W * ((rf * co) * (px / W) + (rf * sn) * (py / H))
    dy = H * ((rf * -sn) * (px / W) + (rf * co) * (py / H))
    return dx, dy


@pytest.mark.parametrize("W,H", [(3840, 2160), (1920, 1080), (7680, 4320), (1024, 1024), (1080, 1920), (2560, 1080), (333, 187)])
@pytest.mark.parametrize("radius", [1.0, 1.7, 2.5, 3.0])
def test_pass0_apron_holds_every_tap_and_the_shaved_corners_hold_none(W, H, radius):
    Rx, Ry, skip, rx, ry = launcher_tile(W, H, radius, temporal=True)
    dx, dy = tap_texel_offsets(W, H, radius)
    # the taps lie in the ellipse the launcher's comment claims
    assert float(((dx / rx) ** 2 + (dy / ry) ** 2).max()) <= 1.0 + 1e-9
    # +- SLACK: the rounding of the tap coordinate in fp32 (one ulp of vUv * size is 1e-3 pixel on a 16K frame)
    for s in (-SLACK, 0.0, SLACK):
        cx, cy = np.floor(0.5 + dx + s).astype(int), np.floor(0.5 + dy + s).astype(int)
        assert np.abs(cx).max() <= Rx and np.abs(cy).max() <= Ry, "a NEAREST tap leaves the staged apron"
        # the rectangle's first row is read only by the tile's first row of pixels, with a row offset of -Ry: a pixel at tile column c then reads
        # column c + cx; the first `skip` texels of that row are columns -Rx .. -Rx + skip - 1 relative to the tile, so a read needs cx < -Rx + skip
        # from the tile's first pixel (c = 0) at the latest.  Mirrored for the last row.
        if skip:
            first_row = cy == -Ry
            assert not np.any(first_row & (cx < -Rx + skip)), "a tap reaches a shaved texel of the first staged row"
            last_row = cy == Ry
            assert not np.any(last_row & (cx > Rx - skip)), "a tap reaches a shaved texel of the last staged row"


def test_the_4k_frame_is_the_case_the_shave_was_made_for():
    """4K, radius 3: 74 x 14 staged texels, two texels shaved at each end -> 53 744 B of LDS, under the 53 760 B that fit three times into a
    CU's 160 KiB of 1 280-byte granules (profiles/r05_microbench/lds_occupancy.txt)."""
    Rx, Ry, skip, _, _ = launcher_tile(3840, 2160, 3.0, temporal=True)
    assert (Rx, Ry, skip) == (5, 3, 2)
    ntex = (64 + 2 * Rx) * (8 + 2 * Ry)
    lds = 16 + (ntex - 2 * skip) * (4 + 16 + 32) + skip * 32
    assert ntex == 1036 and lds == 53744 and lds <= 42 * 1280 < 1036 * 52


@pytest.mark.parametrize("W,H", [(3840, 2160), (1024, 1024), (1080, 1920)])
@pytest.mark.parametrize("radius", [1.0, 2.5, 3.0])
def test_later_passes_apron_holds_every_bilinear_footprint(W, H, radius):
    Rx, Ry, _, _, _ = launcher_tile(W, H, radius, temporal=False)
    dx, dy = tap_texel_offsets(W, H, radius)
    for s in (-SLACK, 0.0, SLACK):
        # LINEAR: the footprint's lower texel is floor(x + 0.5 + d - 0.5) = x + floor(d), its upper one + 1
        lx, ly = np.floor(dx + s).astype(int), np.floor(dy + s).astype(int)
        assert lx.min() >= -Rx and lx.max() + 1 <= Rx and ly.min() >= -Ry and ly.max() + 1 <= Ry
