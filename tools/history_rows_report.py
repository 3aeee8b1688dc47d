#!/usr/bin/env python3
"""Development helper (GPU box, one GPU): what the bounded gather of the composed GI (rfx_gather_history_rows) would move in a row-tiled
run of BASELINE configs[3] — the 4K frame cut into N row tiles — on the synthetic orbit.  For every tile of an N-way split: trace the
tile's rows (rfx_set_row_window + rfx_ssgi_trace), reduce the history rows its rays will read (rfx_ssgi_hit_rows), and count the rows it
would RECEIVE from the other tiles' owners, next to the whole-frame all-gather of round 2 ((N - 1) / N of the frame to every rank).

    python tools/history_rows_report.py [WxH] [frames]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
import numpy as np  # noqa: E402

from rfx_amd import abi, tiling  # noqa: E402
from rfx_amd.context import Context  # noqa: E402
from rfx_amd.scene import synthetic_frame_parallel  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3840x2160").split("x"))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = Context(W, H)
print("# frame %dx%d, steps 20 / refineSteps 5, RFX_TEX_COMPOSE_RGB rows of %d bytes" % (W, H, W * 12))
for fi in range(1, 1 + frames):
    f = synthetic_frame_parallel(W, H, fi)
    ctx.upload_frame(f)
    cam = abi.Camera.from_scene(f.camera)
    sp = abi.SsgiParams(camera=cam, steps=20, refineSteps=5, mode=0, useDirectLight=1, rayDistance=10, thickness=10, envBlur=0.5, blueNoiseIndex=100 + fi)
    for n in (2, 4, 8):
        tiles = tiling.split_rows(H, n)
        worst = 0
        rows_txt = []
        for r, (y0, rows) in enumerate(tiles):
            ctx.set_row_window(y0, y0 + rows)
            ctx.ssgi_trace(sp)
            lo, hi = ctx.ssgi_hit_rows()
            ctx.ssgi_shade(sp)
            got = 0
            if hi >= lo:
                for q, (qy0, qrows) in enumerate(tiles):
                    if q != r:
                        got += max(0, min(hi + 1, qy0 + qrows) - max(lo, qy0))
            worst = max(worst, got)
            rows_txt.append("%d:[%d,%d]->%d" % (r, lo, hi, got))
        ctx.set_row_window(0, 0)
        allg = (H - min(t[1] for t in tiles))
        print("frame %d N=%d  max rows received per rank %4d = %6.2f MB (all-gather: %4d rows = %6.2f MB; %.0f %%)   tile:[lo,hi]->rows received  %s" % (
            fi, n, worst, worst * W * 12 / 1e6, allg, allg * W * 12 / 1e6, 100.0 * worst / allg, "  ".join(rows_txt)), flush=True)
ctx.close()
