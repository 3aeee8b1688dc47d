"""Probe (this container only — needs oracle/_ref/libglref.so): how does llvmpipe filter `textureCube` lookups, the one operation of
CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42)?  Findings that DESIGN.md §6 cites:

  * face selection and (s, t) follow the GL table exactly (NEAREST lookups of an index cube: 100 % as modelled);
  * LINEAR lookups fed with llvmpipe's own direction vectors differ from the exact fp32 bilinear blend of the (seamlessly wrapped) texels:
    the filter position is off by up to ~0.011 texel and the weights sit on a 1/256 grid (8-bit weights + an approximate reciprocal of the
    major axis) -> a value error of up to ~0.011 x the local texel contrast, unbounded for an HDR cube; no bit-level restatement exists
    short of emulating the host CPU's rcpps.

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
    return face, (f32(0.5) * (sc / ma + f32(1))).astype(f32), (f32(0.5) * (tc / ma + f32(1))).astype(f32)


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
    print("LINEAR: filter position off by up to %.4f / %.4f texel (u / v); %.0f %% of the positions sit on the 1/256 grid" % (
        np.abs(du).max(), np.abs(dv).max(), 100 * (np.abs(q - np.round(q)) < 1e-3).mean()))


if __name__ == "__main__":
    main()
