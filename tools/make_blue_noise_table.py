#!/usr/bin/env python3
"""Decode the reference's blue-noise PNG once into the 64 KiB RGBA8 table the kernels consume.

Run in the build container only (needs /root/reference):
    python tools/make_blue_noise_table.py

Source asset: src/utils/blue_noise_rgba.png (128x128 RGBA8, loaded by
src/utils/BlueNoiseUtils.js:9-15 through three's TextureLoader, i.e. flipY=true, nearest,
repeat).  The table is stored ALREADY FLIPPED so that table[row][col] is texel (col,row) in GL
texture space (row 0 = bottom), which is what `texelFetch(blueNoiseTexture, ivec2, 0)` in
src/utils/shader/blue_noise.glsl:42 addresses.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

SRC = "/root/reference/src/utils/blue_noise_rgba.png"
DST = os.path.join(os.path.dirname(__file__), "..", "realism-effects_amd", "data", "blue_noise_128_rgba8.bin")


def main():
    img = Image.open(SRC)
    assert img.size == (128, 128) and img.mode == "RGBA", (img.size, img.mode)
    a = np.asarray(img, np.uint8)
    print("sha256(decoded, PNG row order) =", hashlib.sha256(a.tobytes()).hexdigest())
    flipped = np.ascontiguousarray(a[::-1])
    with open(DST, "wb") as f:
        f.write(flipped.tobytes())
    print("wrote", os.path.abspath(DST), flipped.nbytes, "bytes; sha256 =", hashlib.sha256(flipped.tobytes()).hexdigest())


if __name__ == "__main__":
    sys.exit(main())
