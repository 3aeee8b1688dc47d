"""One tile of a row-tiled run with the exchanges behind the C ABI (test_tiling_gloo.py: under pytest --hostsim).  No torch here."""
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "realism-effects_amd"))

from rfx_amd import abi, tiling  # noqa: E402
from rfx_amd.context import Context  # noqa: E402
from rfx_amd.effect import SSGIEffect  # noqa: E402
from rfx_amd.scene import synthetic_frame  # noqa: E402

rank, world, outdir, W, H, FRAMES = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
MODE = sys.argv[7] if len(sys.argv) > 7 else "all"  # CommTiledRenderer history_gather
frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
vmax = max(float(np.abs(f.velocity[..., 1].view(np.float32)).max()) for f in frames)
halo = (int(sys.argv[8]) if len(sys.argv) > 8 else 0) or tiling.required_halo(3.0, vmax, H, W)  # an override taller than the tiles: multi-hop exchange
y0, rows = tiling.split_rows(H, world)[rank]
ctx = Context(W, H, tile_y0=y0, tile_rows=rows, halo_rows=halo)
idf = os.path.join(outdir, "nccl_id")
if rank == 0:
    with open(idf + ".tmp", "wb") as f:
        f.write(Context.comm_unique_id())
    os.rename(idf + ".tmp", idf)
while not os.path.exists(idf):
    time.sleep(0.01)
with open(idf, "rb") as f:
    uid = f.read()
r = tiling.CommTiledRenderer(ctx, rank, world, uid, history_gather=MODE)
scene, cam = types.SimpleNamespace(frame=None), frames[0].camera
fx = SSGIEffect(None, scene, cam, dict(width=W, height=H, denoiseIterations=1), seeds=dict(ssgi=5, denoise=9))
for f in frames:
    scene.frame = f
    for k, v in vars(f.camera).items():
        setattr(cam, k, v)
    fx.update(r, None)
r.finish_pending()
r.finish_halo()
if MODE == "bounded":  # every rank holds only the rows its own rays needed: complete the frame for the whole-frame comparison below
    assert r.history_gather == "bounded" and len(r.history_bytes_received) == FRAMES
    r.gather_whole_history()
assert ctx.halo_violations() == 0 and r.exchange_count == FRAMES * 3, (ctx.halo_violations(), r.exchange_count)
np.savez(os.path.join(outdir, "c%d.npz" % rank), y0=y0, rows=rows,
         **{abi.TEX_NAMES[t]: ctx.download(t, y0, rows) for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1, abi.TEX_COMPOSE)},
         compose_rgb_full=ctx.download(abi.TEX_COMPOSE_RGB), history_bytes=np.array(r.history_bytes_received, np.int64))
ctx.comm_destroy()
ctx.close()
assert "torch" not in sys.modules
