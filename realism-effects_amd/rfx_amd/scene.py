"""Synthetic G-buffer dumps (the "pre-dumped Float32/Uint8 arrays" of the north star).

The reference produces its G-buffers by rasterising a three.js scene
(src/gbuffer/GBufferPass.js:100-119, src/temporal-reproject/pass/VelocityDepthNormalPass.js);
that rasteriser is out of scope.  What IS in scope is the
*texel format* those passes emit, because K1..K4 decode it.  This module ray-casts an analytic
scene (ground plane + spheres + boxes, orbiting perspective camera) and encodes it with the
reference's encode-side codec so that the dumps are byte-compatible with what the reference
shaders expect:

  depth     R32F     gl_FragCoord.z of the G-buffer pass (1.0 = background clear value)
  gbuffer   RGBA32F  packGBuffer(): {RGBA8 diffuse bits, oct half2 normal bits,
                     float-coded roughness/metalness, RGBE8 emissive bits}
                     (src/gbuffer/shader/gbuffer_packing.glsl:166-178)
  velocity  RGBA32F  (vel.x, vel.y, packNormal(worldNormal), fragCoordZ)
                     (src/temporal-reproject/material/VelocityDepthNormalMaterial.js:76-83,186-188)
  direct    RGBA32F  the composer's input buffer (direct lighting), SSGIEffect.js:396

All planes are row-major with row 0 = BOTTOM (GL convention), matching `vUv`.
Matrices are column-major float32[16] exactly like three.js `Matrix4.elements`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# ----------------------------------------------------------------------------- camera


def make_perspective(fov_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    """three.js PerspectiveCamera.updateProjectionMatrix + Matrix4.makePerspective (r151)."""
    top = near * math.tan(math.radians(fov_deg) * 0.5)
    height = 2.0 * top
    width = aspect * height
    left = -0.5 * width
    right, bottom = left + width, top - height
    x = 2 * near / (right - left)
    y = 2 * near / (top - bottom)
    a = (right + left) / (right - left)
    b = (top + bottom) / (top - bottom)
    c = -(far + near) / (far - near)
    d = -2 * far * near / (far - near)
    m = np.zeros((4, 4), np.float64)  # m[row, col]
    m[0, 0] = x
    m[0, 2] = a
    m[1, 1] = y
    m[1, 2] = b
    m[2, 2] = c
    m[2, 3] = d
    m[3, 2] = -1
    return m


def make_orthographic(half_height: float, aspect: float, near: float, far: float) -> np.ndarray:
    """three.js OrthographicCamera.updateProjectionMatrix + Matrix4.makeOrthographic (r151), symmetric frustum."""
    top, right = half_height, half_height * aspect
    m = np.zeros((4, 4), np.float64)
    m[0, 0] = 1.0 / right
    m[1, 1] = 1.0 / top
    m[2, 2] = -2.0 / (far - near)
    m[2, 3] = -(far + near) / (far - near)
    m[3, 3] = 1.0
    return m


def look_at_world(eye, target, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """camera.matrixWorld for camera.lookAt(target) (camera looks down -Z)."""
    eye = np.asarray(eye, np.float64)
    z = eye - np.asarray(target, np.float64)
    z /= np.linalg.norm(z)
    x = np.cross(np.asarray(up, np.float64), z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, eye
    return m


def col_major32(m: np.ndarray) -> np.ndarray:
    """row/col ndarray -> three.js `elements` (column-major) float32[16]."""
    return np.ascontiguousarray(m.T.reshape(16)).astype(np.float32)


@dataclass
class Camera:
    """Dumped camera state: what the reference passes read from `camera.*` each frame."""

    near: float
    far: float
    position: np.ndarray  # float32[3]
    projectionMatrix: np.ndarray  # float32[16] column-major
    projectionMatrixInverse: np.ndarray
    matrixWorld: np.ndarray
    matrixWorldInverse: np.ndarray
    isPerspectiveCamera: bool = True
    quaternion: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1], np.float64))

    @staticmethod
    def orbit(frame: int, aspect: float, fov=40.0, near=0.01, far=250.0, deg_per_frame=0.5,
              radius=8.5, height=3.2, target=(0.0, 0.9, 0.0), start_deg=30.0, ortho_half_height: float | None = None) -> "Camera":
        """`ortho_half_height`: an OrthographicCamera with that half extent (near 0.1, far 16: the far plane cuts the ground, leaving
        background texels) instead of the PerspectiveCamera."""
        ang = math.radians(start_deg + deg_per_frame * frame)
        eye = np.array([radius * math.cos(ang), height, radius * math.sin(ang)])
        mw = look_at_world(eye, target)
        if ortho_half_height is not None:
            near, far = 0.1, 16.0
            p = make_orthographic(ortho_half_height, aspect, near, far)
        else:
            p = make_perspective(fov, aspect, near, far)
        # quaternion from rotation matrix (for didCameraMove, src/utils/SceneUtils.js:17-27)
        r = mw[:3, :3]
        qw = math.sqrt(max(0.0, 1 + r[0, 0] + r[1, 1] + r[2, 2])) / 2
        qx = math.copysign(math.sqrt(max(0.0, 1 + r[0, 0] - r[1, 1] - r[2, 2])) / 2, r[2, 1] - r[1, 2])
        qy = math.copysign(math.sqrt(max(0.0, 1 - r[0, 0] + r[1, 1] - r[2, 2])) / 2, r[0, 2] - r[2, 0])
        qz = math.copysign(math.sqrt(max(0.0, 1 - r[0, 0] - r[1, 1] + r[2, 2])) / 2, r[1, 0] - r[0, 1])
        return Camera(near=near, far=far, position=eye.astype(np.float32),
                      projectionMatrix=col_major32(p), projectionMatrixInverse=col_major32(np.linalg.inv(p)),
                      matrixWorld=col_major32(mw), matrixWorldInverse=col_major32(np.linalg.inv(mw)),
                      isPerspectiveCamera=ortho_half_height is None, quaternion=np.array([qx, qy, qz, qw]))

    def _m(self, name) -> np.ndarray:
        return getattr(self, name).astype(np.float64).reshape(4, 4).T  # back to [row, col]


# ----------------------------------------------------------------------------- encode-side codec


def pack_half2x16(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """GLSL packHalf2x16(vec2(a, b)): RNE to binary16, a in the low 16 bits."""
    lo = a.astype(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
    hi = b.astype(np.float32).astype(np.float16).view(np.uint16).astype(np.uint32)
    return lo | (hi << np.uint32(16))


def encode_oct_wrap(n: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """gbuffer_packing.glsl:36-50 (float32 arithmetic)."""
    n = n.astype(np.float32)
    s = np.abs(n[..., 0]) + np.abs(n[..., 1]) + np.abs(n[..., 2])
    n = n / s[..., None]
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    wx = np.float32(1.0) - np.abs(y)
    wy = np.float32(1.0) - np.abs(x)
    wx = np.where(x < 0, -wx, wx)
    wy = np.where(y < 0, -wy, wy)
    ox = np.where(z > 0, x, wx)
    oy = np.where(z > 0, y, wy)
    return (ox * np.float32(0.5) + np.float32(0.5)).astype(np.float32), (oy * np.float32(0.5) + np.float32(0.5)).astype(np.float32)


def pack_normal(n: np.ndarray) -> np.ndarray:
    """packNormal(): uint32 bit pattern of the float the shader stores."""
    ox, oy = encode_oct_wrap(n)
    return pack_half2x16(ox, oy)


def vec4_to_float_bits(v: np.ndarray) -> np.ndarray:
    """vec4ToFloat() gbuffer_packing.glsl:143-149 -> uint32 bits."""
    v = np.minimum(v.astype(np.float32) + np.float32(1e-4), np.float32(0.999999))
    b = (v * np.float32(255.0)).astype(np.uint32)  # uvec4(): truncation
    return (b[..., 3] << np.uint32(24)) | (b[..., 2] << np.uint32(16)) | (b[..., 1] << np.uint32(8)) | b[..., 0]


def color2float(rough: np.ndarray, metal: np.ndarray) -> np.ndarray:
    """color2float(vec3(roughness, metalness, 0.)) gbuffer_packing.glsl:17-22 -> float32 value."""
    one_safe, off = np.float32(0.999999), np.float32(1e-4)
    r = np.minimum(rough.astype(np.float32) + off, one_safe)
    g = np.minimum(metal.astype(np.float32) + off, one_safe)
    b = np.minimum(np.zeros_like(r) + off, one_safe)
    p, p1 = np.float32(256.0), np.float32(257.0)
    fr = np.floor(r * p + np.float32(0.5))
    fb = np.floor(b * p + np.float32(0.5))
    fg = np.floor(g * p + np.float32(0.5))
    return (fr + fb * p1 + fg * p1 * p1).astype(np.float32)


def encode_rgbe8_bits(rgb: np.ndarray) -> np.ndarray:
    """vec4ToFloat(encodeRGBE8(emissive)); black -> word 0 (the GLSL takes log2(0), Appendix D-9)."""
    rgb = rgb.astype(np.float32)
    mx = rgb.max(axis=-1)
    safe = np.where(mx > 0, mx, np.float32(1.0))
    fexp = np.ceil(np.log2(safe)).astype(np.float32)
    enc = np.empty(rgb.shape[:-1] + (4,), np.float32)
    enc[..., :3] = rgb / np.exp2(fexp)[..., None]
    enc[..., 3] = (fexp + np.float32(128.0)) / np.float32(255.0)
    bits = vec4_to_float_bits(enc)
    return np.where(mx > 0, bits, np.uint32(0)).astype(np.uint32)


# ----------------------------------------------------------------------------- scene


@dataclass
class Frame:
    """One dumped frame: planes + camera (cur & prev) — the input contract of the hot path."""

    width: int
    height: int
    depth: np.ndarray  # (H, W) float32
    gbuffer: np.ndarray  # (H, W, 4) uint32 (bit patterns of the RGBA32F texels)
    velocity: np.ndarray  # (H, W, 4) uint32 bit patterns (xy float, z packed normal, w depth)
    direct: np.ndarray  # (H, W, 4) float32
    camera: Camera
    prev_camera: Camera
    frame_index: int = 0
    aov: dict | None = None  # unpacked attribute planes (render(..., aov=True)): what an engine exports, input of the device importer


class AnalyticScene:
    """Ground plane y=0 + spheres + axis-aligned boxes, static; camera orbits (0.5 deg/frame)."""

    def __init__(self, seed: int = 1234, n_spheres: int = 20, n_boxes: int = 12):
        rng = np.random.RandomState(seed)
        self.spheres = []
        self.boxes = []
        mats = []
        # material 0 = ground (two-tone checker handled at shading time)
        mats.append(dict(diffuse=(0.55, 0.55, 0.5), rough=0.6, metal=0.0, emissive=(0, 0, 0)))
        for _ in range(n_spheres):
            r = rng.uniform(0.3, 1.0)
            c = np.array([rng.uniform(-6, 6), r, rng.uniform(-6, 6)])
            self.spheres.append((c, r, len(mats)))
            mats.append(self._rand_mat(rng))
        for _ in range(n_boxes):
            h = rng.uniform(0.3, 0.9, size=3)
            c = np.array([rng.uniform(-6, 6), h[1], rng.uniform(-6, 6)])
            self.boxes.append((c - h, c + h, len(mats)))
            mats.append(self._rand_mat(rng))
        self.mat_diffuse = np.array([m["diffuse"] for m in mats], np.float32)
        self.mat_rough = np.array([m["rough"] for m in mats], np.float32)
        self.mat_metal = np.array([m["metal"] for m in mats], np.float32)
        self.mat_emissive = np.array([m["emissive"] for m in mats], np.float32)
        self.light_dir = np.array([0.45, 0.8, 0.35])
        self.light_dir /= np.linalg.norm(self.light_dir)

    @staticmethod
    def _rand_mat(rng):
        emissive = (0, 0, 0)
        if rng.uniform() < 0.12:
            emissive = tuple(rng.uniform(0.3, 3.0, size=3))
        return dict(diffuse=tuple(rng.uniform(0.2, 0.9, size=3)), rough=float(rng.uniform(0, 1)),
                    metal=float(rng.uniform() < 0.3), emissive=emissive)

    # ---- ray casting (float64, vectorised over pixels, loop over objects)
    def _trace(self, o: np.ndarray, d: np.ndarray):
        """`o`: one origin (3,) — perspective — or one per ray (n, 3) — orthographic."""
        n_px = d.shape[0]
        if o.ndim == 2:
            return self._trace_many(o, d)
        t_best = np.full(n_px, np.inf)
        mat = np.full(n_px, -1, np.int32)
        nrm = np.zeros((n_px, 3))
        # ground plane
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -o[1] / d[:, 1]
        hit = (d[:, 1] < 0) & (t > 0)
        t_best = np.where(hit, t, t_best)
        mat[hit] = 0
        nrm[hit] = (0, 1, 0)
        for c, r, m in self.spheres:
            oc = o - c
            b = d @ oc
            cc = oc @ oc - r * r
            disc = b * b - cc
            ok = disc > 0
            sq = np.sqrt(np.where(ok, disc, 0))
            t = -b - sq
            hit = ok & (t > 1e-6) & (t < t_best)
            t_best = np.where(hit, t, t_best)
            mat[hit] = m
            p = o + d[hit] * t[hit, None]
            nrm[hit] = (p - c) / r
        for lo, hi, m in self.boxes:
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / d
                t0 = (lo - o) * inv
                t1 = (hi - o) * inv
            tmin = np.minimum(t0, t1)
            tmax = np.maximum(t0, t1)
            tn = tmin.max(axis=1)
            tf = tmax.min(axis=1)
            hit = (tn < tf) & (tn > 1e-6) & (tn < t_best)
            t_best = np.where(hit, tn, t_best)
            mat[hit] = m
            ax = tmin[hit].argmax(axis=1)
            nn = np.zeros((int(hit.sum()), 3))
            nn[np.arange(nn.shape[0]), ax] = -np.sign(d[hit][np.arange(nn.shape[0]), ax])
            nrm[hit] = nn
        return t_best, mat, nrm

    def _trace_many(self, o: np.ndarray, d: np.ndarray):
        n_px = d.shape[0]
        t_best = np.full(n_px, np.inf)
        mat = np.full(n_px, -1, np.int32)
        nrm = np.zeros((n_px, 3))
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -o[:, 1] / d[:, 1]
        hit = (d[:, 1] < 0) & (t > 0)
        t_best = np.where(hit, t, t_best)
        mat[hit] = 0
        nrm[hit] = (0, 1, 0)
        for c, r, m in self.spheres:
            oc = o - c
            b = (d * oc).sum(1)
            cc = (oc * oc).sum(1) - r * r
            disc = b * b - cc
            ok = disc > 0
            t = -b - np.sqrt(np.where(ok, disc, 0))
            hit = ok & (t > 1e-6) & (t < t_best)
            t_best = np.where(hit, t, t_best)
            mat[hit] = m
            nrm[hit] = (o[hit] + d[hit] * t[hit, None] - c) / r
        for lo, hi, m in self.boxes:
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / d
                t0, t1 = (lo - o) * inv, (hi - o) * inv
            tmin, tmax = np.minimum(t0, t1), np.maximum(t0, t1)
            tn, tf = tmin.max(axis=1), tmax.min(axis=1)
            hit = (tn < tf) & (tn > 1e-6) & (tn < t_best)
            t_best = np.where(hit, tn, t_best)
            mat[hit] = m
            ax = tmin[hit].argmax(axis=1)
            nn = np.zeros((int(hit.sum()), 3))
            nn[np.arange(nn.shape[0]), ax] = -np.sign(d[hit][np.arange(nn.shape[0]), ax])
            nrm[hit] = nn
        return t_best, mat, nrm

    def render(self, width: int, height: int, frame_index: int = 0, row0: int = 0, rows: int | None = None,
               frame_height: int | None = None, vfov_rows: int | None = None, aov: bool = False, ortho_half_height: float | None = None) -> Frame:
        """Dump frame `frame_index`.  (row0, rows, frame_height) select a horizontal band of a
        taller frame — used by the row-tiled multi-GPU path; default = the whole frame.
        `vfov_rows`: number of rows that span the nominal 40 deg vertical fov; a taller frame
        (weak scaling: N stacked 4K tiles) widens the vertical fov instead of squeezing the
        horizontal one.  `aov=True` also keeps the UNPACKED attribute planes an engine would export (frame.aov: diffuse RGBA,
        world normal, roughness, metalness, emissive, uv-space velocity) — the input of the device-side importer (rfx_pack_gbuffer /
        rfx_pack_velocity), which must reproduce `gbuffer` / `velocity` from them."""
        fh = frame_height or height
        rows = rows if rows is not None else height
        aspect = width / fh
        fov = 40.0
        if vfov_rows and vfov_rows != fh:
            fov = min(160.0, 2.0 * math.degrees(math.atan(math.tan(math.radians(20.0)) * fh / vfov_rows)))
        cam = Camera.orbit(frame_index, aspect, fov=fov, ortho_half_height=ortho_half_height)
        prev = Camera.orbit(frame_index - 1, aspect, fov=fov, ortho_half_height=ortho_half_height) if frame_index > 0 else cam
        P, Pi, C, V = cam._m("projectionMatrix"), cam._m("projectionMatrixInverse"), cam._m("matrixWorld"), cam._m("matrixWorldInverse")
        Pp, Vp = prev._m("projectionMatrix"), prev._m("matrixWorldInverse")

        depth = np.ones((rows, width), np.float32)
        gb = np.zeros((rows, width, 4), np.uint32)
        vel = np.zeros((rows, width, 4), np.uint32)
        direct = np.zeros((rows, width, 4), np.float32)
        # raster clear colour (0,0,0,1) for every colour target
        one_bits = np.float32(1.0).view(np.uint32)
        gb[..., 3] = one_bits
        vel[..., 3] = one_bits
        direct[..., 3] = 1.0

        planes = None
        if aov:
            planes = dict(diffuse=np.zeros((rows, width, 4), np.float32), normal=np.zeros((rows, width, 3), np.float32),
                          roughness=np.zeros((rows, width), np.float32), metalness=np.zeros((rows, width), np.float32),
                          emissive=np.zeros((rows, width, 3), np.float32), velocity=np.zeros((rows, width, 2), np.float32))
        xs = (np.arange(width) + 0.5) / width * 2 - 1
        chunk = max(1, (1 << 20) // width)
        o = C[:3, 3]
        for y0 in range(0, rows, chunk):
            y1 = min(rows, y0 + chunk)
            ys = (np.arange(row0 + y0, row0 + y1) + 0.5) / fh * 2 - 1
            gx, gy = np.meshgrid(xs, ys)
            n_px = gx.size
            # view-space ray through the pixel centre: unproject ndc (x, y, -1, 1) and (x, y, 1, 1)
            ndc = np.stack([gx.ravel(), gy.ravel(), -np.ones(n_px), np.ones(n_px)], axis=1)
            pv = ndc @ Pi.T
            pv = pv[:, :3] / pv[:, 3:4]
            if ortho_half_height is None:
                dv = pv / np.linalg.norm(pv, axis=1, keepdims=True)
                d = dv @ C[:3, :3].T
                ro = o
            else:  # orthographic: parallel rays down the camera's -Z from the pixel's point on the near plane
                d = np.tile(-C[:3, 2], (n_px, 1))
                ro = pv @ C[:3, :3].T + o
            t, mat, nrm = self._trace(ro, d)
            hit = mat >= 0
            wp = ro + d * np.where(hit, t, 0)[:, None]
            wp4 = np.concatenate([wp, np.ones((n_px, 1))], axis=1)
            clip = wp4 @ (P @ V).T
            with np.errstate(divide="ignore", invalid="ignore"):
                z = 0.5 * clip[:, 2] / clip[:, 3] + 0.5
                pos1 = clip[:, :2] / clip[:, 3:4] * 0.5 + 0.5
                clip0 = wp4 @ (Pp @ Vp).T
                pos0 = clip0[:, :2] / clip0[:, 3:4] * 0.5 + 0.5
            z32 = z.astype(np.float32)
            hit &= z32 < np.float32(1.0)  # beyond the far plane -> clipped -> background
            m = np.where(hit, mat, 0)
            diffuse = self.mat_diffuse[m].copy()
            # ground checker
            ground = hit & (mat == 0)
            chk = ((np.floor(wp[:, 0] * 0.5) + np.floor(wp[:, 2] * 0.5)) % 2) == 0
            diffuse[ground & chk] *= np.float32(0.45)
            diff4 = np.concatenate([diffuse, np.ones((n_px, 1), np.float32)], axis=1)
            g = np.zeros((n_px, 4), np.uint32)
            g[:, 0] = vec4_to_float_bits(diff4)
            nsafe = np.where(hit[:, None], nrm, (0, 1, 0))
            nbits = pack_normal(nsafe)
            g[:, 1] = nbits
            g[:, 2] = color2float(self.mat_rough[m], self.mat_metal[m]).view(np.uint32)
            g[:, 3] = encode_rgbe8_bits(self.mat_emissive[m])
            v4 = np.zeros((n_px, 4), np.uint32)
            velxy = (pos1 - pos0).astype(np.float32)
            v4[:, 0] = velxy[:, 0].view(np.uint32)
            v4[:, 1] = velxy[:, 1].view(np.uint32)
            v4[:, 2] = nbits
            v4[:, 3] = z32.view(np.uint32)
            ndl = np.clip(nsafe @ self.light_dir, 0, None)
            dl = np.zeros((n_px, 4), np.float32)
            dl[:, :3] = diffuse * (0.9 * ndl + 0.05)[:, None].astype(np.float32) * (1 - self.mat_metal[m])[:, None] + self.mat_emissive[m]
            dl[:, 3] = 1.0
            sl = np.s_[y0:y1]
            hm = hit.reshape(y1 - y0, width)
            depth[sl] = np.where(hm, z32.reshape(hm.shape), np.float32(1.0))
            gb[sl] = np.where(hm[..., None], g.reshape(hm.shape + (4,)), gb[sl])
            vel[sl] = np.where(hm[..., None], v4.reshape(hm.shape + (4,)), vel[sl])
            direct[sl] = np.where(hm[..., None], dl.reshape(hm.shape + (4,)), direct[sl])
            if planes is not None:
                shp = hm.shape
                planes["diffuse"][sl] = diff4.reshape(shp + (4,))
                planes["normal"][sl] = nsafe.astype(np.float32).reshape(shp + (3,))
                planes["roughness"][sl] = self.mat_rough[m].astype(np.float32).reshape(shp)
                planes["metalness"][sl] = self.mat_metal[m].astype(np.float32).reshape(shp)
                planes["emissive"][sl] = self.mat_emissive[m].astype(np.float32).reshape(shp + (3,))
                planes["velocity"][sl] = velxy.reshape(shp + (2,))
        fr = Frame(width, rows, depth, gb, vel, direct, cam, prev, frame_index)
        fr.aov = planes
        return fr


_default_scene = None


def synthetic_frame(width: int, height: int, frame_index: int = 0, seed: int = 1234, **kw) -> Frame:
    """Convenience: frame `frame_index` of the default scene (seed 1234)."""
    global _default_scene
    if _default_scene is None or _default_scene[0] != seed:
        _default_scene = (seed, AnalyticScene(seed))
    return _default_scene[1].render(width, height, frame_index, **kw)


def _render_band(args):
    seed, width, height, frame_index, row0, rows = args
    f = AnalyticScene(seed).render(width, rows, frame_index, row0=row0, rows=rows, frame_height=height)
    return f.depth, f.gbuffer, f.velocity, f.direct, f.camera, f.prev_camera


def _pool_worker_init():
    # a tool preloaded into the parent (rocprofv3 --pmc) leaves its SIGTERM handler in the forked workers: Pool.terminate() then waits
    # forever for a profiler finalisation in processes that never touched the device (round 3: bench.py "hung" under --pmc right here)
    import signal
    signal.signal(signal.SIGTERM, signal.SIG_DFL)


def synthetic_band_parallel(width: int, height: int, frame_index: int, row0: int, rows: int, seed: int = 1234, workers: int | None = None) -> Frame:
    """Rows [row0, row0 + rows) of frame `frame_index` of a width x height frame, ray-cast by a pool of processes (one slice each) —
    the same texels as AnalyticScene.render(..., row0, rows, frame_height=height) in a fraction of the wall time on a many-core host
    (a 4K dump takes ~13 s single-threaded).  Returns a Frame whose planes hold just those rows."""
    import multiprocessing as mp
    import os
    workers = workers or max(1, min(len(os.sched_getaffinity(0)), 32))
    workers = max(1, min(workers, rows // 4))
    if workers == 1:
        return AnalyticScene(seed).render(width, rows, frame_index, row0=row0, rows=rows, frame_height=height)
    edges = [row0 + rows * i // workers for i in range(workers + 1)]
    jobs = [(seed, width, height, frame_index, edges[i], edges[i + 1] - edges[i]) for i in range(workers) if edges[i + 1] > edges[i]]
    # close() + join(): the workers leave through os._exit after their last task.  Pool.__exit__ would terminate() them with SIGTERM, and a
    # profiler preloaded into the parent (rocprofv3 --pmc) keeps ITS handler for that signal in the forked workers (it also wraps
    # sigaction, so the initializer's reset is best effort): the workers then "finalise" a profiler session forever and join() never returns.
    pool = mp.get_context("fork").Pool(len(jobs), initializer=_pool_worker_init)
    try:
        parts = pool.map(_render_band, jobs)
        pool.close()
        pool.join()
    except BaseException:
        pool.terminate()
        raise
    cat = lambda k: np.ascontiguousarray(np.concatenate([p[k] for p in parts], axis=0))  # noqa: E731
    return Frame(width, rows, cat(0), cat(1), cat(2), cat(3), parts[0][4], parts[0][5], frame_index)


def synthetic_frame_parallel(width: int, height: int, frame_index: int = 0, seed: int = 1234, workers: int | None = None) -> Frame:
    """synthetic_frame() through synthetic_band_parallel: the same texels, ~10x sooner."""
    if (workers or 2) == 1 or height < 16:
        return synthetic_frame(width, height, frame_index, seed)
    return synthetic_band_parallel(width, height, frame_index, 0, height, seed, workers)


def synthetic_environment(width: int = 256, height: int = 128, seed: int = 1234) -> np.ndarray:
    """A synthetic equirectangular HDR environment (`scene.environment`), (H, W, 4) float32, row 0 = bottom (v = 0 = straight
    down): ground colour below the horizon, a sky gradient above it, a warm sun whose core exceeds the shader's luminance
    clamp (ssgi.frag:330-340) and a few soft cloud lobes so that neighbouring texels differ at every mip level."""
    rng = np.random.RandomState(seed)
    v = (np.arange(height, dtype=np.float64) + 0.5) / height
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    phi = (1.0 - v)[:, None] * np.pi          # equirectUvToDirection, ssgi_utils.frag:77-86
    theta = (u - 0.5)[None, :] * 2.0 * np.pi
    d = np.stack([np.sin(phi) * np.cos(theta), np.cos(phi) * np.ones_like(theta), np.sin(phi) * np.sin(theta)], -1)
    up = d[..., 1]
    sky = np.array([0.25, 0.45, 0.9]) * (0.35 + 0.65 * np.clip(up, 0, 1)[..., None]) + np.array([0.9, 0.8, 0.7]) * (np.exp(-6.0 * np.abs(up))[..., None] * 0.6)
    ground = np.array([0.18, 0.16, 0.13]) * (0.6 + 0.4 * np.clip(-up, 0, 1)[..., None])
    img = np.where((up > 0)[..., None], sky, ground)
    sun = np.array([0.45, 0.55, -0.7]); sun /= np.linalg.norm(sun)
    c = np.clip((d * sun).sum(-1), -1, 1)
    img += np.array([1.0, 0.85, 0.6]) * (60.0 * np.exp(-(1 - c) * 900.0) + 2.5 * np.exp(-(1 - c) * 25.0))[..., None]
    for _ in range(6):
        a = rng.normal(size=3); a[1] = abs(a[1]) + 0.2; a /= np.linalg.norm(a)
        cc = np.clip((d * a).sum(-1), -1, 1)
        img += np.array([0.8, 0.8, 0.85]) * (rng.uniform(0.3, 0.9) * np.exp(-(1 - cc) * rng.uniform(20, 80)))[..., None]
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out
