"""CPU: the file formats either side of the path: EXR / PFM / PNG round trips, the AOV-EXR importer feeding the
device-side packer's oracle, tone mapping, and — where /root/reference exists — the reference example's own Radiance .hdr environment and
its blue-noise PNG."""
import os

import numpy as np
import pytest

import rfx_oracle as O
from rfx_amd import dump, imageio
from rfx_amd.scene import AnalyticScene


def test_exr_round_trip_all_compressions(tmp_path):
    rng = np.random.RandomState(1)
    H, W = 37, 53
    ch = {"diffuse.R": rng.rand(H, W).astype(np.float32), "normal.X": rng.rand(H, W).astype(np.float32) * 2 - 1,
          "depth.Z": rng.rand(H, W).astype(np.float32) * 1e3}
    ch["normal.X"][4:9, 10:40] = 0.25  # constant stretches: the run-length paths
    for comp in ("none", "rle", "zips", "zip", "piz", "pxr24"):
        for half in (False, True):
            p = str(tmp_path / "t.exr")
            imageio.write_exr(p, ch, comp, half)
            r = imageio.read_exr(p)
            assert sorted(r) == sorted(ch)
            for k in ch:
                want = ch[k].astype(np.float16).astype(np.float32) if half else ch[k]
                if comp == "pxr24" and not half:  # the one lossy case: a float32 sample keeps 24 bits (15 of mantissa), rounded
                    want = (imageio._float_to_f24(ch[k]) << np.uint32(8)).view(np.float32)
                    assert (np.abs(want - ch[k]) <= np.abs(ch[k]) * 2.0 ** -15).all()
                assert np.array_equal(r[k], want), (comp, half, k)


def test_exr_tiled_round_trip(tmp_path):
    """single-level tiled files (some renderers' default): ragged last tiles in both directions, every codec, tile chunks in file order"""
    rng = np.random.RandomState(2)
    H, W = 37, 53
    ch = {"depth.Z": rng.rand(H, W).astype(np.float32), "id.Y": np.repeat(rng.rand(H, 1).astype(np.float32), W, axis=1)}
    for comp in ("none", "rle", "zips", "zip", "piz", "pxr24"):
        for tiles in ((16, 8), (64, 64), (5, 37)):
            p = str(tmp_path / "t.exr")
            imageio.write_exr(p, ch, comp, half=True, tiles=tiles)
            r = imageio.read_exr(p)
            for k in ch:
                assert np.array_equal(r[k], ch[k].astype(np.float16).astype(np.float32)), (comp, tiles, k)
    imageio.write_exr(p, ch, "zip", tiles=(16, 16))
    assert np.array_equal(imageio.read_exr(p)["depth.Z"], ch["depth.Z"])


def test_exr_multipart_round_trip(tmp_path):
    """one AOV per part: the parts' channels come back under "<part>.<channel>" (or as they are when they already carry a layer)"""
    rng = np.random.RandomState(4)
    H, W = 40, 33
    parts = {"normal": {c: rng.rand(H, W).astype(np.float32) for c in "XYZ"}, "depth": {"Z": rng.rand(H, W).astype(np.float32)},
             "extra": {"diffuse.R": rng.rand(H, W).astype(np.float32)}}
    for comp in ("none", "zip", "piz"):
        p = str(tmp_path / "m.exr")
        imageio.write_exr_multipart(p, parts, comp)
        r = imageio.read_exr(p)
        assert sorted(r) == ["depth.Z", "diffuse.R", "normal.X", "normal.Y", "normal.Z"]
        assert np.array_equal(r["normal.Y"], parts["normal"]["Y"]) and np.array_equal(r["depth.Z"], parts["depth"]["Z"])
        assert np.array_equal(r["diffuse.R"], parts["extra"]["diffuse.R"])
    with open(p, "r+b") as f:  # deep data stays refused
        f.seek(4)
        f.write((2 | 0x1800).to_bytes(4, "little"))
    with pytest.raises(ValueError, match="deep"):
        imageio.read_exr(p)


def test_exr_rle_and_pxr24_known_answers():
    """The two simple codecs against hand-made streams (the formats' definitions, not this module's encoders)."""
    # RLE: a count byte n >= 0 repeats the next byte n + 1 times, n < 0 copies -n literal bytes
    stream = bytes([2, 7, 0xFD, 1, 2, 3, 0, 9, 127, 5])
    assert imageio._rle_uncompress(stream, 3 + 3 + 1 + 128) == bytes([7, 7, 7, 1, 2, 3, 9] + [5] * 128)
    with pytest.raises(ValueError):
        imageio._rle_uncompress(stream, 10)
    data = bytes([1, 1, 1, 1, 2, 3, 3, 4, 4, 4] + [8] * 300 + [1, 2])
    assert imageio._rle_uncompress(imageio._rle_compress(data), len(data)) == data
    # float -> 24 bits: sign, exponent, 15 mantissa bits rounded half up; never rounds a finite value into infinity; NaN stays NaN
    f = np.array([1.0, -2.0, np.inf, -np.inf, np.nan, 3.4028234e38, 1.0 + 2.0 ** -16, 1.0 + 2.0 ** -17], np.float32)
    assert [hex(v) for v in imageio._float_to_f24(f)] == ["0x3f8000", "0xc00000", "0x7f8000", "0xff8000", "0x7fc000", "0x7f7fff", "0x3f8001", "0x3f8000"]
    # one scanline, one HALF channel of 4 samples 1, 3, 3, 0x0102: differences 1, 2, 0, 0x00ff as a high-byte plane then a low-byte plane
    import zlib
    blk = zlib.compress(bytes([0, 0, 0, 0, 1, 2, 0, 0xFF]))
    raw = imageio._pxr24_uncompress_block(blk, [("c", imageio._PT_HALF)], 4, 1)
    assert np.array_equal(np.frombuffer(raw, np.uint16), [1, 3, 3, 0x0102])


def test_exr_piz_codec(tmp_path):
    """PIZ (the OpenEXR default: value bitmap + lookup table, 2-D wavelet in 14-bit or modulo-16-bit arithmetic, canonical Huffman with a
    run-length escape): every stage on its own and the block codec on data that exercises each branch — smooth renders (compressible, the
    14-bit wavelet), float noise (> 16383 distinct words in a block: the 16-bit wavelet), constant planes (runs), ragged and odd sizes,
    a single-value block.  No externally produced PIZ file exists in this container; the externally produced uncompressed EXR of the
    CPython test suite (when present) pins the container format itself."""
    rng = np.random.RandomState(5)
    # the wavelet is its own inverse pair, both arithmetics, every odd / even size
    for ny, nx in ((1, 1), (2, 2), (3, 5), (8, 8), (7, 13), (32, 53), (33, 64)):
        for mx in (100, 16383, 16384, 65535):
            a = rng.randint(0, mx + 1, (ny, nx)).astype(np.uint16)
            b = a.copy()
            imageio._wav2(b, mx, False)
            imageio._wav2(b, mx, True)
            assert np.array_equal(a, b), (ny, nx, mx)
    # Huffman: skewed, uniform, single-symbol and run-heavy inputs
    for words in (rng.randint(0, 50, 5000), rng.randint(0, 65536, 70000), np.full(1000, 7), np.repeat(rng.randint(0, 9, 40), rng.randint(1, 900, 40)),
                  np.array([65535]), np.concatenate([np.zeros(300, int), [65535, 0, 65535], np.zeros(600, int)])):
        w = words.astype(np.uint16)
        blk = imageio._huf_compress(w)
        assert np.array_equal(imageio._huf_uncompress(blk, w.size), w)
    assert len(imageio._huf_compress(np.zeros(100000, np.uint16))) < 600  # runs collapse: 10 bits per 256 words
    # whole blocks, compressed path forced (write_exr stores a block raw when PIZ does not shrink it)
    H, W = 32, 200
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    smooth = (np.sin(xx * 0.05) * np.cos(yy * 0.1) + 1.5).astype(np.float32)
    for chans, planes in (([("A", 1), ("B", 1)], [smooth.astype("<f2"), (smooth * 0.5).astype("<f2")]),
                          ([("Z", 2)], [smooth.astype("<f4")]),                                   # float: two interleaved word planes
                          ([("N", 2), ("h", 1)], [rng.rand(H, W).astype("<f4"), smooth.astype("<f2")]),  # > 16383 distinct words: 16-bit wavelet
                          ([("id", 0)], [np.full((H, W), 7, "<u4")])):
        raw = b"".join(pl[y].tobytes() for y in range(H) for pl in planes)
        blk = imageio._piz_compress_block(raw, chans, W, H)
        assert imageio._piz_uncompress_block(blk, chans, W, H) == raw, chans
    assert len(imageio._piz_compress_block(smooth.astype("<f2").tobytes(), [("A", 1)], W, H)) < 0.6 * smooth.size * 2  # it does compress
    # files: ragged last block (70 rows = 32 + 32 + 6), odd width, half and float, PIZ blocks actually present
    H, W = 70, 129
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    ch = {"R": (np.sin(xx * 0.03) + yy * 0.01).astype(np.float32), "G": np.floor(xx / 16).astype(np.float32), "depth.Z": np.ones((H, W), np.float32)}
    for half in (True, False):
        p = str(tmp_path / "piz.exr")
        imageio.write_exr(p, ch, "piz", half)
        assert os.path.getsize(p) < 0.7 * H * W * 3 * (2 if half else 4)
        r = imageio.read_exr(p)
        for k in ch:
            assert np.array_equal(r[k], ch[k].astype(np.float16).astype(np.float32) if half else ch[k]), (half, k)
    ext = "/mnt/sandboxing/model_tools_env/v1/python/install/lib/python3.11/test/imghdrdata/python.exr"
    if os.path.exists(ext):  # written by a real OpenEXR library: RGBA half 16 x 16, uncompressed
        r = imageio.read_exr(ext)
        assert sorted(r) == ["A", "B", "G", "R"] and r["R"].shape == (16, 16) and np.isfinite(r["R"]).all()


def test_pfm_png_round_trip_and_tonemap(tmp_path):
    rng = np.random.RandomState(2)
    img = rng.rand(20, 30, 3).astype(np.float32)
    imageio.write_pfm(str(tmp_path / "a.pfm"), img)
    assert np.array_equal(imageio.read_pfm(str(tmp_path / "a.pfm")), img)
    png = (rng.rand(20, 30, 4) * 255).astype(np.uint8)
    imageio.write_png(str(tmp_path / "a.png"), png)
    assert np.array_equal(imageio.read_png(str(tmp_path / "a.png")), png)
    t = imageio.tonemap(np.array([[[0.0, 0.0, 0.0], [0.18, 0.18, 0.18], [1e9, np.nan, np.inf]]], np.float32))
    assert t.dtype == np.uint8 and (t[0, 0] == 0).all() and 100 < t[0, 1, 0] < 150 and t[0, 2, 0] == 255
    lin = imageio.tonemap(np.array([[[0.5, 0.5, 0.5]]], np.float32), "linear")
    assert abs(int(lin[0, 0, 0]) - 188) <= 1  # sRGB OETF of 0.5


def test_aov_exr_import_feeds_the_packer(tmp_path):
    """A frame exported as ONE multi-layer AOV EXR (what a renderer writes) comes back as the attribute planes the importer packs; packing
    them reproduces the packed dump's texels (oracle of the device-side packer; the GPU test does the same through rfx_pack_gbuffer)."""
    f = AnalyticScene(1234).render(96, 54, 1, aov=True)
    p = str(tmp_path / "frame.exr")
    dump.write_exr_dump(p, f)
    g = dump.read_exr_dump(p)
    assert (g.width, g.height) == (96, 54) and np.array_equal(g.depth, f.depth) and np.array_equal(g.direct, f.direct)
    for k in ("diffuse", "normal", "roughness", "metalness", "emissive", "velocity"):
        assert np.array_equal(g.aov[k], np.asarray(f.aov[k], np.float32)), k
    gb = O.pack_gbuffer(g.aov, g.depth)
    cov = f.depth != 1.0
    assert np.array_equal(gb[cov][:, :3], f.gbuffer[cov][:, :3])  # diffuse, normal, roughness/metalness words (emissive: see D-9, log2(0))
    assert np.array_equal(O.pack_velocity(g.aov, g.depth)[cov], f.velocity[cov])
    # and through the dump directory both hosts read
    d = str(tmp_path / "dumpdir")
    dump.write_dump(d, g, packed=False)
    h = dump.read_dump(d)
    assert h.gbuffer is None and np.array_equal(h.aov["normal"], g.aov["normal"])
    np.testing.assert_allclose(h.camera.projectionMatrix, f.camera.projectionMatrix)


@pytest.mark.reference
@pytest.mark.skipif(not os.path.isdir("/root/reference/example/public/hdr"), reason="/root/reference absent")
def test_reference_example_assets_import(blue_noise):
    """Non-synthetic inputs: the reference example's environment map (example/public/hdr/*.hdr, what its RGBELoader loads into
    scene.environment) through read_hdr -> the environment chain + the importance tables; and the reference's blue-noise PNG decodes
    to exactly the table this package ships (src/utils/blue_noise_rgba.png, flipY as BlueNoiseUtils.js:9-15)."""
    from rfx_amd import envmap
    env = imageio.environment_from_hdr("/root/reference/example/public/hdr/spree_bank_1k.hdr")
    assert env.shape == (512, 1024, 4) and np.isfinite(env).all() and env[..., :3].max() > 4.0 and (env[..., :3] >= 0).all()
    e = O.EnvMap(env, half=True)
    assert e.levels == 11 and np.isfinite(e.level(10)).all()
    # the sky half of an outdoor panorama is brighter than the ground half
    assert env[256:, :, :3].mean() > env[:256, :, :3].mean()
    m, c, tot = envmap.build_importance(env[::4, ::4].astype(np.float16).astype(np.float32))
    assert m.shape == (128,) and c.shape == (128, 256) and tot > 0 and (np.diff(m) >= 0).all()
    assert np.array_equal(imageio.read_png("/root/reference/src/utils/blue_noise_rgba.png"), blue_noise)


def test_node_image_writers_match_python(tmp_path):
    """js/imageio.js (what run_dump.js --png / --exr / --pfm write) against the Python twin: the EXR and PFM carry the exact float32
    texels, the tone-mapped PNG equals imageio.tonemap to 1 LSB."""
    import subprocess
    W, H = 23, 11
    rng = np.random.RandomState(5)
    img = (rng.rand(H, W, 4).astype(np.float32) * np.array([3, 2, 1, 1], np.float32))
    img[0, 0, :3] = [np.nan, np.inf, 1e-9]
    src = tmp_path / "img.bin"
    img.tofile(str(src))
    js = ("const io=require('%s');const fs=require('fs');const b=fs.readFileSync('%s');"
          "const a=new Float32Array(b.buffer.slice(b.byteOffset,b.byteOffset+b.length));"
          "io.writeEXR('%s/o.exr',a,%d,%d);io.writePFM('%s/o.pfm',a,%d,%d);io.writePNG('%s/o.png',io.tonemap(a,%d,%d),%d,%d,3)") % (
        os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "realism-effects_amd", "js", "imageio.js"), src,
        tmp_path, W, H, tmp_path, W, H, tmp_path, W, H, W, H)
    subprocess.check_call(["node", "-e", js])
    e = imageio.read_exr(str(tmp_path / "o.exr"))
    for i, c in enumerate("RGBA"):
        assert np.array_equal(e[c], img[..., i], equal_nan=True), c
    assert np.array_equal(imageio.read_pfm(str(tmp_path / "o.pfm")), img[..., :3], equal_nan=True)
    png = imageio.read_png(str(tmp_path / "o.png")).astype(np.int32)
    assert np.abs(png - imageio.tonemap(img).astype(np.int32)).max() <= 1
