#!/usr/bin/env python3
"""Convert a renderer's AOV EXR (+ side-car <file>.json with the cameras) into the dump directory both hosts read:
    python tools/exr_to_dump.py frame0001.exr dumps/f0001 [--map normal=N.X,N.Y,N.Z --map depth=Z.Z ...]
Layer names expected by default: rfx_amd.imageio.AOV_LAYOUT (diffuse.RGBA, normal.XYZ [world], roughness.Y, metalness.Y, emissive.RGB,
velocity.XY [uv units], depth.Z [gl_FragCoord.z], direct.RGBA).  The dump holds UNPACKED planes; the device packs them (rfx_pack_gbuffer)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
from rfx_amd import dump  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("exr")
ap.add_argument("out")
ap.add_argument("--map", action="append", default=[], help="aov=chan0,chan1,... overrides a default layer mapping")
a = ap.parse_args()
names = {m.split("=")[0]: tuple(m.split("=")[1].split(",")) for m in a.map}
f = dump.read_exr_dump(a.exr, names or None)
dump.write_dump(a.out, f, packed=False)
print("wrote", a.out, "%dx%d" % (f.width, f.height))
