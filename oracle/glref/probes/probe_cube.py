"""Probe (this container only — needs oracle/_ref/libglref.so): how does llvmpipe answer `textureCube` lookups, the one operation of
CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42)?  What rfx_oracle.c cube_face / cube_texel / cube_linear / cube_lod restate:

  * face selection and (s, t) follow the GL table; (s, t) = (sc * (1 / ma)) * 0.5 + 0.5 reproduces the filter position of LINEAR lookups
    BIT-EXACTLY on every interior position (this probe, "LINEAR");
  * seamless edges: a footprint texel beyond the face edge is the neighbouring face's texel; beyond a corner: the average of the three texels
    that exist; blend = fused lerp in x, then in y (tests/test_oracle_vs_golden.py::test_cube_to_equirect_vs_golden pins the whole pass);
  * a mipmapped cube (three's CubeTexture default): the level of detail is PER PIXEL, from the direction differences within the pixel's own
    row / own column of its 2x2 quad, through the quotient rule on the pixel's own face, then the linear-mantissa log2 — reproduced to < 1e-6
    in lod on a chain whose level l holds the constant l (this probe, "LOD").

Round 1's version of this probe concluded "8-bit weights and an approximate reciprocal: no restatement exists".  That was the probe, not the
GL: its shaders lacked `precision highp samplerCube;`, GLSL ES defaults samplers to lowp, and Mesa lowers a lowp fetch to fp16 — every
fetched value was rounded to 11 bits.  No GPU runs the pass that way; with the qualifier the lookup is plain fp32.

    python oracle/glref/probes/probe_cube.py
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from chain import GL, Program, Tex  # noqa: E402

f32 = np.float32
HEAD = "#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp samplerCube;\nprecision highp sampler2D;\nin vec2 vUv;\nout vec4 o;\nuniform samplerCube cubeMap;\n"
DIRS = """
#define M_PI 3.1415926535897932384626433832795
vec3 direction() {  // the pass's own arithmetic
    float longitude = vUv.x * 2. * M_PI - M_PI + M_PI / 2.;
    float latitude = vUv.y * M_PI;
    vec3 dir = vec3(-sin(longitude) * sin(latitude), cos(latitude), -cos(longitude) * sin(latitude));
    dir.y = -dir.y;
    return dir;
}
"""


class CubeTex:
    def __init__(self, size, fmt, filt, data):
        self.w = self.h = size
        data = np.ascontiguousarray(data)
        self.id = GL.lib().glref_cube_texture(size, fmt, filt, data.ctypes.data_as(ctypes.c_void_p))
        assert self.id > 0, self.id


def face_coords(d):
    ax = np.abs(d)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    face = np.where((ax[..., 0] >= ax[..., 1]) & (ax[..., 0] >= ax[..., 2]), np.where(x >= 0, 0, 1),
                    np.where(ax[..., 1] >= ax[..., 2], np.where(y >= 0, 2, 3), np.where(z >= 0, 4, 5)))
    sc = np.choose(face, [-z, z, x, x, x, -x])
    tc = np.choose(face, [-y, -y, z, -z, -y, -y])
    ma = np.choose(face, [ax[..., 0], ax[..., 0], ax[..., 1], ax[..., 1], ax[..., 2], ax[..., 2]])
    ima = (f32(1) / ma).astype(f32)
    return face, ((sc * ima) * f32(0.5) + f32(0.5)).astype(f32), ((tc * ima) * f32(0.5) + f32(0.5)).astype(f32)


def main():
    W, H, S = 128, 64, 16
    pd = Program(HEAD + DIRS + "void main() { o = vec4(direction(), 1.); }")
    dirs = Tex(W, H, 0)
    pd.draw([dirs])
    d = dirs.read()[..., :3]
    face, s, t = face_coords(d)
    # 1. NEAREST lookups of an index cube
    idx = np.zeros((6, S, S, 4), f32)
    idx[..., 0] = np.arange(6)[:, None, None]
    idx[..., 1] = np.arange(S)[None, None, :]
    idx[..., 2] = np.arange(S)[None, :, None]
    pl = Program(HEAD + "uniform sampler2D dirs;\nvoid main() { o = texture(cubeMap, texture(dirs, vUv).xyz); }")
    pl.sampler("dirs", dirs)
    out = Tex(W, H, 0)
    pl.sampler("cubeMap", CubeTex(S, 0, 0, idx))
    pl.draw([out])
    r = out.read()
    xi, yi = np.clip(np.floor(s * S), 0, S - 1), np.clip(np.floor(t * S), 0, S - 1)
    print("NEAREST: face %.4f x %.4f y %.4f of the lookups as modelled" % ((r[..., 0] == face).mean(), (r[..., 1] == xi).mean(), (r[..., 2] == yi).mean()))
    # 2. LINEAR lookups of a ramp cube return the filter position llvmpipe used
    ramp = np.zeros((6, S, S, 4), f32)
    ramp[..., 0] = np.arange(S)[None, None, :]
    ramp[..., 1] = np.arange(S)[None, :, None]
    pl.sampler("cubeMap", CubeTex(S, 0, 1, ramp))
    pl.draw([out])
    r = out.read()
    u, v = s * f32(S) - f32(0.5), t * f32(S) - f32(0.5)
    inside = (np.floor(u) >= 0) & (np.floor(u) + 1 < S) & (np.floor(v) >= 0) & (np.floor(v) + 1 < S)
    du, dv = (r[..., 0] - u)[inside], (r[..., 1] - v)[inside]
    q = r[..., 0][inside] * 256
    print("LINEAR: filter position off by up to %.2e / %.2e texel (u / v), bit-exact on %.2f %% of the interior positions" % (
        np.abs(du).max(), np.abs(dv).max(), 100 * ((du == 0) & (dv == 0)).mean()))


def lod_probe():
    W, H, S = 256, 128, 64
    pd = Program(HEAD + DIRS + "void main() { o = vec4(direction(), 1.); }")
    dirs = Tex(W, H, 0)
    pd.draw([dirs])
    d = dirs.read()[..., :3].astype(np.float64)
    pl = Program(HEAD + DIRS + "void main() { o = texture(cubeMap, direction()); }")
    out = Tex(W, H, 0)
    ct = CubeTex(S, 0, 3, np.zeros((6, S, S, 4), f32))  # mipmap filters, no generation: level l is uploaded as the constant l
    lv, s_ = 0, S
    while True:
        for f in range(6):
            data = np.full((s_, s_, 4), float(lv), f32)
            GL.lib().glref_cube_upload_level(ct.id, f, lv, data.ctypes.data_as(ctypes.c_void_p))
        if s_ == 1:
            break
        s_, lv = max(s_ >> 1, 1), lv + 1
    pl.sampler("cubeMap", ct)
    pl.draw([out])
    lod = out.read()[..., 0]
    face, _, _ = face_coords(d.astype(f32))

    def comps(f, p):
        x, y, z = p
        return [(-z, -y, x), (z, -y, -x), (x, z, y), (x, -z, -y), (x, -y, z), (-x, -y, -z)][f]
    model = np.zeros((H, W))
    for y in range(H):
        for x in range(W):
            p = d[y, x]
            ddx = d[y, x | 1] - d[y, x & ~1]
            ddy = d[y | 1, x] - d[y & ~1, x]
            f = int(face[y, x])
            sc, tc, ma = comps(f, p)
            xsc, xtc, xma = comps(f, ddx)
            ysc, ytc, yma = comps(f, ddy)
            k = 0.5 / (ma * ma)
            r2 = max(((xsc * ma - sc * xma) * k) ** 2 + ((xtc * ma - tc * xma) * k) ** 2, ((ysc * ma - sc * yma) * k) ** 2 + ((ytc * ma - tc * yma) * k) ** 2) * S * S
            m, e = np.frexp(r2)
            model[y, x] = min(max(0.5 * ((e - 1) + (2 * m - 1)), 0.0), 6.0)
    print("LOD: range %.3f .. %.3f over a %dx%d target of a %d^2 cube; model vs GL max |diff| %.2e" % (lod.min(), lod.max(), W, H, S, np.abs(model - lod).max()))


if __name__ == "__main__":
    main()
    lod_probe()
