"""On-disk dump format shared by the Python and Node hosts: `frame.json` + raw little-endian
planes, row 0 = bottom (the "pre-dumped Float32/Uint8 arrays" of the north star).

  frame.json    {width, height, camera{...}, prevCamera{...}}  (matrices = 16 numbers, column-major)
  depth.bin     float32 W*H          gbuffer.bin   uint32 W*H*4 (bit patterns of the RGBA32F texels)
  velocity.bin  uint32  W*H*4        direct.bin    float32 W*H*4

An engine that exports UNPACKED attribute planes writes, instead of gbuffer.bin and velocity.bin (write_dump(..., packed=False)):
  aov_diffuse.bin float32 W*H*4   aov_normal.bin float32 W*H*3 (world space)   aov_roughness.bin / aov_metalness.bin float32 W*H
  aov_emissive.bin float32 W*H*3  aov_velocity.bin float32 W*H*2 (uv units)
and the device packs them (rfx_pack_gbuffer / rfx_pack_velocity).

A renderer's usual AOV container is one multi-layer OpenEXR per frame: read_exr_dump() / write_exr_dump() map it onto the same frame
object (layer names: rfx_amd.imageio.AOV_LAYOUT; the cameras travel in the EXR's side-car `<name>.json`, same fields as frame.json).
"""
from __future__ import annotations

import json
import os
import types

import numpy as np

_CAM_FIELDS = ("projectionMatrix", "projectionMatrixInverse", "matrixWorld", "matrixWorldInverse", "position", "quaternion")


def _cam_to_json(cam):
    d = {k: [float(x) for x in np.asarray(getattr(cam, k)).ravel()] for k in _CAM_FIELDS if hasattr(cam, k)}
    d.update(near=float(cam.near), far=float(cam.far), isPerspectiveCamera=bool(getattr(cam, "isPerspectiveCamera", True)))
    return d


def _cam_from_json(d):
    ns = types.SimpleNamespace(**{k: (np.asarray(v, np.float64) if k == "quaternion" else np.asarray(v, np.float32)) for k, v in d.items()
                                  if k in _CAM_FIELDS})
    ns.near, ns.far, ns.isPerspectiveCamera = d["near"], d["far"], d.get("isPerspectiveCamera", True)
    return ns


_AOV = (("diffuse", 4), ("normal", 3), ("roughness", 1), ("metalness", 1), ("emissive", 3), ("velocity", 2))


def write_dump(dirname: str, frame, packed: bool = True) -> None:
    os.makedirs(dirname, exist_ok=True)
    meta = dict(width=int(frame.width), height=int(frame.height), camera=_cam_to_json(frame.camera),
                prevCamera=_cam_to_json(getattr(frame, "prev_camera", frame.camera)))
    with open(os.path.join(dirname, "frame.json"), "w") as f:
        json.dump(meta, f)
    np.ascontiguousarray(frame.depth, np.float32).tofile(os.path.join(dirname, "depth.bin"))
    if packed:
        np.ascontiguousarray(frame.gbuffer).view(np.uint32).tofile(os.path.join(dirname, "gbuffer.bin"))
        np.ascontiguousarray(frame.velocity).view(np.uint32).tofile(os.path.join(dirname, "velocity.bin"))
    else:
        for k, _ in _AOV:
            np.ascontiguousarray(frame.aov[k], np.float32).tofile(os.path.join(dirname, "aov_%s.bin" % k))
    np.ascontiguousarray(frame.direct, np.float32).tofile(os.path.join(dirname, "direct.bin"))


def read_dump(dirname: str):
    with open(os.path.join(dirname, "frame.json")) as f:
        meta = json.load(f)
    W, H = meta["width"], meta["height"]
    rd = lambda n, dt, shape: np.fromfile(os.path.join(dirname, n), dt).reshape(shape)  # noqa: E731
    fr = types.SimpleNamespace(width=W, height=H, camera=_cam_from_json(meta["camera"]), prev_camera=_cam_from_json(meta["prevCamera"]),
                               depth=rd("depth.bin", np.float32, (H, W)), direct=rd("direct.bin", np.float32, (H, W, 4)), gbuffer=None, velocity=None,
                               aov=None)
    if os.path.exists(os.path.join(dirname, "gbuffer.bin")):
        fr.gbuffer, fr.velocity = rd("gbuffer.bin", np.uint32, (H, W, 4)), rd("velocity.bin", np.uint32, (H, W, 4))
    else:
        fr.aov = {k: rd("aov_%s.bin" % k, np.float32, (H, W, ch) if ch > 1 else (H, W)) for k, ch in _AOV}
    return fr


def write_exr_dump(path: str, frame, compression: str = "zip") -> None:
    """One multi-layer EXR (`path`) + side-car `path`.json (cameras) from a frame that carries unpacked attribute planes (frame.aov)."""
    from . import imageio
    ch = {}
    for key, names in imageio.AOV_LAYOUT.items():
        a = frame.depth if key == "depth" else (frame.direct if key == "direct" else frame.aov[key])
        a = np.asarray(a, np.float32)
        a = a[..., None] if a.ndim == 2 else a
        for i, n in enumerate(names):
            ch[n] = a[..., i]
    imageio.write_exr(path, ch, compression)
    meta = dict(width=int(frame.width), height=int(frame.height), camera=_cam_to_json(frame.camera),
                prevCamera=_cam_to_json(getattr(frame, "prev_camera", frame.camera)))
    with open(path + ".json", "w") as f:
        json.dump(meta, f)


def read_exr_dump(path: str, names: dict | None = None):
    """The inverse: a frame whose G-buffer / velocity come as unpacked attribute planes (frame.aov) for the device-side importer."""
    from . import imageio
    planes = imageio.exr_to_dump_planes(path, names)
    with open(path + ".json") as f:
        meta = json.load(f)
    H, W = planes["depth"].shape
    if (W, H) != (meta["width"], meta["height"]):
        raise ValueError("%s: image is %dx%d, side-car says %dx%d" % (path, W, H, meta["width"], meta["height"]))
    return types.SimpleNamespace(width=W, height=H, camera=_cam_from_json(meta["camera"]), prev_camera=_cam_from_json(meta["prevCamera"]),
                                 depth=planes["depth"], direct=planes["direct"], gbuffer=None, velocity=None, aov=planes["aov"])
