#!/usr/bin/env python3
"""Development helper (GPU box, one GPU): what the bounded gather of the composed GI (rfx_gather_history_rows) would move in a row-tiled
run of BASELINE configs[3] — the 4K frame cut into N row tiles — on the synthetic orbit.  For every tile of an N-way split: trace the
tile's rows (rfx_set_row_window + rfx_ssgi_trace), reduce the history texels its rays will read (rfx_ssgi_hit_rows: the (min, max) row
interval of ABI 15; rfx_ssgi_hit_mask: the per-row column-block mask of ABI 16), and count what it would RECEIVE from the other tiles'
owners three ways — the interval (round 3's plan), the whole rows the mask uses, and the mask's column blocks (what rfx_gather_history_rows
moves since round 4: packed, one message per peer) — next to the whole-frame all-gather ((N - 1) / N of the frame to every rank).

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
        worst_m = worst_b = 0
        for r, (y0, rows) in enumerate(tiles):
            ctx.set_row_window(y0, y0 + rows)
            ctx.ssgi_trace(sp)
            lo, hi = ctx.ssgi_hit_rows()
            mask = ctx.ssgi_hit_mask()
            ctx.ssgi_shade(sp)
            got = 0
            if hi >= lo:
                for q, (qy0, qrows) in enumerate(tiles):
                    if q != r:
                        got += max(0, min(hi + 1, qy0 + qrows) - max(lo, qy0))
            other = np.ones(H, bool)
            other[y0:y0 + rows] = False
            got_m = int((mask[other] != 0).sum())                                        # whole rows the mask uses, owned by others
            got_b = int(sum(bin(int(w)).count("1") for w in mask[other])) / 32.0         # ... in column blocks, as row equivalents
            worst, worst_m, worst_b = max(worst, got), max(worst_m, got_m), max(worst_b, got_b)
            rows_txt.append("%d:[%d,%d]->%d/%d/%.0f" % (r, lo, hi, got, got_m, got_b))
        ctx.set_row_window(0, 0)
        allg = (H - min(t[1] for t in tiles))
        mb = lambda n_rows: n_rows * W * 12 / 1e6  # noqa: E731
        print("frame %d N=%d  busiest rank receives: interval %4d rows = %6.2f MB | whole rows of the mask %4d = %6.2f MB | column blocks (moved) = %6.2f MB   (all-gather: %4d rows = %6.2f MB)   tile:[lo,hi]->interval/mask/blocks  %s" % (
            fi, n, worst, mb(worst), worst_m, mb(worst_m), mb(worst_b), allg, mb(allg), "  ".join(rows_txt)), flush=True)
ctx.close()
