"""On-disk dump format shared by the Python and Node hosts: `frame.json` + raw little-endian
planes, row 0 = bottom (the "pre-dumped Float32/Uint8 arrays" of the north star).

  frame.json    {width, height, camera{...}, prevCamera{...}}  (matrices = 16 numbers, column-major)
  depth.bin     float32 W*H          gbuffer.bin   uint32 W*H*4 (bit patterns of the RGBA32F texels)
  velocity.bin  uint32  W*H*4        direct.bin    float32 W*H*4

An engine that exports UNPACKED attribute planes writes, instead of gbuffer.bin and velocity.bin (write_dump(..., packed=False)):
  aov_diffuse.bin float32 W*H*4   aov_normal.bin float32 W*H*3 (world space)   aov_roughness.bin / aov_metalness.bin float32 W*H
  aov_emissive.bin float32 W*H*3  aov_velocity.bin float32 W*H*2 (uv units)
and the device packs them (rfx_pack_gbuffer / rfx_pack_velocity).
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
