"""Row tiling of a frame over the GPUs of one node, one process per GPU.

Rank r owns frame rows [r*H/N, (r+1)*H/N).  Pixels are independent inside every kernel; the
coupling is only through gathers from previous-stage textures, so the tile holds `halo` extra
rows above and below and three exchange steps keep them current:

  after K2 and after every K3 pass : Send/Recv of the `halo` rows around each tile (the two row neighbours' boundary rows; with a halo
                                     taller than the tiles also rows of tiles further away: halo_plan) of the textures just
                                     written (RCCL over xGMI; message = halo*W*texel bytes per
                                     texture and direction — latency-bound).
                                     Issued asynchronously: the next draw (K3 pass, K4) first
                                     produces the INTERIOR of the tile — rows whose taps stay
                                     inside the tile's own rows — through rfx_set_row_window,
                                     waits for the halo rows, then draws the two boundary strips
  after K4                         : all-gather of the composed GI tile rows (next frame's K1
                                     gathers it anywhere on screen) — of its .rgb, which is all K1
                                     reads, kept by K4 as 12-byte texels in RFX_TEX_COMPOSE_RGB.  It is the one big message
                                     (tile bytes x (N-1) per rank) and only the SHADING half of K1
                                     reads it, so it is started asynchronously and waited for
                                     between rfx_ssgi_trace and rfx_ssgi_shade: it overlaps the next
                                     frame's depth pre-pass and ray march
  after a framebuffer copy (TRAA)  : the same neighbour Send/Recv for the pass's own history

K1 needs no exchange: it recomputes the +-2 rows K2's neighbourhood clamp reads, from the
read-only dump planes every rank holds for its band, and reads depth / last frame's composed GI
whole-frame.  The tiled result is bit-identical to the single-GPU result as long as the halo
covers the gather footprints: `required_halo()`.

The exchanger is written against torch tensors so the same code runs over RCCL on device
memory (the rfx contexts' buffers, bound with rfx_bind_external) and over gloo on CPU tensors
(tests/test_tiling_gloo.py).
"""
from __future__ import annotations

import math

import numpy as np

from . import abi

EXCHANGED = (abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_A0, abi.TEX_DENOISE_A1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1)


def split_rows(height: int, world: int):
    """Even split; tile boundaries on EVEN rows so the 2x2 derivative quads never straddle tiles."""
    base = (height // world) & ~1
    if world < 1 or base <= 0:
        raise ValueError("split_rows: %d rows cannot be cut into %d tiles of at least 2 rows" % (height, world))
    starts = [r * base for r in range(world)]
    rows = [base] * world
    rows[-1] = height - starts[-1]
    return list(zip(starts, rows))


def required_halo(radius: float, max_abs_velocity_y: float, frame_height: int, frame_width: int | None = None) -> int:
    """Rows of halo that make the tiled result exact:
       K3: the reference rotates the Poisson offsets in UV space (`rm * (offset / resolution)`,
           poisson_denoise.frag:183-189), so the tap footprint is radius*max(1, H/W) ROWS high
           (and radius*max(1, W/H) columns wide), +1 for the bilinear footprint, +1 for rounding
       K2: history bicubic at vUv - velocity: |v_y|*H rows + 2 texels of Catmull-Rom + 1 bilinear;
           K1 output neighbourhood +-2 rows (recomputed locally)."""
    aspect_rows = max(1.0, frame_height / frame_width) if frame_width else 1.0
    k3 = int(math.ceil(radius * aspect_rows)) + 2
    k2 = int(math.ceil(abs(max_abs_velocity_y) * frame_height)) + 4
    return max(k3, k2, 2)


def halo_plan(height: int, world: int, rank: int, halo: int):
    """One halo exchange as row intervals of the frame: [(peer, send, recv)] with send = the rows of THIS tile that lie inside the peer's
    band (tile +- halo), recv = the peer's rows inside this tile's band; None where empty.  With halo <= the tiles' height that is the two
    neighbours' boundary rows; a taller halo (many ranks, a fast camera) reaches past them and more peers appear.  Every rank derives
    the same intervals from split_rows, so every send has its receive (rfx_halo_exchange in csrc/rfx_comm.hip follows the same rule)."""
    tiles = split_rows(height, world)
    y0, n = tiles[rank]
    y1 = y0 + n
    plan = []
    for p, (py0, pn) in enumerate(tiles):
        if p == rank:
            continue
        py1 = py0 + pn
        send = (max(y0, py0 - halo), min(y1, py1 + halo))
        recv = (max(py0, y0 - halo), min(py1, y1 + halo))
        send = send if send[1] > send[0] else None
        recv = recv if recv[1] > recv[0] else None
        if send or recv:
            plan.append((p, send, recv))
    return plan


class TiledRenderer:
    """Wraps the per-tile renderer (an rfx Context, or the oracle double in tests) and performs the
    exchange steps through torch.distributed.  `tensors[tex]` is a torch tensor over the rows the
    tile holds of that texture (first dim = held rows)."""

    def __init__(self, inner, tensors: dict, rank: int, world: int, group=None):
        import torch.distributed as dist
        self._dist = dist
        self._setup(inner, tensors, rank, world, group)

    def _setup(self, inner, tensors, rank, world, group=None):
        self.inner, self.tensors, self.rank, self.world, self.group = inner, tensors, rank, world, group
        self.W, self.H = inner.W, inner.H
        self.tile_y0, self.tile_rows, self.halo = inner.tile_y0, inner.tile_rows, inner.halo
        if world > 1:
            # every rank derives the exchange's row intervals from the even split (halo_plan): the tile must be the rank's share of it
            tiles = split_rows(self.H, world)
            if (self.tile_y0, self.tile_rows) != tiles[rank]:
                raise ValueError("TiledRenderer: rank %d of %d holds rows [%d, %d), split_rows() assigns [%d, %d)" % (
                    rank, world, self.tile_y0, self.tile_y0 + self.tile_rows, tiles[rank][0], tiles[rank][0] + tiles[rank][1]))
        self.exchange_count = 0
        self._pending = []  # (works, tensor) of the composed-GI all-gather in flight
        self._halo_pending = []  # (works, tensor) of halo Send/Recvs in flight
        # draws overlap the halo exchange of their input (interior first) when the tile renderer can window its launches
        self.overlap_halo_exchange = world > 1 and hasattr(inner, "set_row_window")
        # effect.SSGIPass splits K1 into trace + shade around before_ssgi_shade() when this is set
        self.overlap_history_gather = world > 1 and hasattr(inner, "ssgi_trace")
        # K1 reads only .rgb of the composed GI: when the RGB twin is bound, K4 keeps it and IT is gathered (12 B/px instead of 16)
        self.gather_history_rgb = world > 1 and abi.TEX_COMPOSE_RGB in tensors
        self.history_gather = "all"
        self.history_bytes_received = []  # per frame, "bounded" / "peer" modes (what "all" receives: the other tiles, every frame)

    def use_peer_history(self, all_gather_object):
        """Switch the composed-GI exchange to the DEVICE-DRIVEN pull (include/rfx.h rfx_peer_*): between a frame's trace and its shade this
        rank's own kernel loads the column blocks its rays will read straight out of their owners' planes, through IPC mappings — no
        collective, no host wait.  The RGB twin must be the library's own plane (not bound to a torch tensor).
        `all_gather_object(obj) -> [every rank's obj in rank order]`: how the ranks' export blobs travel, once (any host channel)."""
        if self.world == 1:
            return
        if not self.overlap_history_gather:
            raise RuntimeError("use_peer_history: the tile renderer must split K1 into ssgi_trace / ssgi_shade (the pull sits between them)")
        tex = abi.TEX_COMPOSE_RGB
        blobs = all_gather_object(self.inner.peer_export(tex))
        self.inner.peer_open(tex, blobs, self.rank, self.world)
        self.gather_history_rgb = True
        self.history_gather = "peer"
        self._all_gather_object = all_gather_object

    def _peer_before_shade(self):
        self.finish_pending()
        self.history_bytes_received.append(self.inner.peer_gather_history(abi.TEX_COMPOSE_RGB))
        self.inner.comm_wait()  # orders the shade after the pull (stream order: nothing waits on the host)

    def gather_whole_history(self):
        """every rank's rows of the composed GI on every rank, now (a host that wants the whole frame on one rank — bench.py's checksum)"""
        if self.world > 1 and self.history_gather == "peer":
            tex = abi.TEX_COMPOSE_RGB
            self.inner.sync()
            parts = self._all_gather_object(np.ascontiguousarray(self.inner.download(tex)[self.tile_y0:self.tile_y0 + self.tile_rows]))
            self.inner.upload(tex, np.concatenate(parts, axis=0))

    def __getattr__(self, name):  # everything else (upload, the four draws, ...) goes to the tile's renderer
        return getattr(self.inner, name)

    # anything that reads the whole composed GI, or hands control back to the caller, first lets the all-gather land
    def before_ssgi_shade(self):
        if self.history_gather == "peer":
            return self._peer_before_shade()
        self.finish_pending()

    def ssgi_march(self, p):
        if self.world > 1 and self.history_gather == "peer":
            raise RuntimeError("history_gather \"peer\": K1 must run as ssgi_trace / ssgi_shade (the pull sits between them)")
        self.finish_pending()
        return self.inner.ssgi_march(p)

    # K2 gathers its history (exchanged at the end of the previous frame) at vUv - velocity: anywhere within the halo
    def temporal_reproject(self, p):
        self.finish_halo()
        return self.inner.temporal_reproject(p)

    def copy_framebuffer(self, dst):
        self.finish_halo()
        return self.inner.copy_framebuffer(dst)

    # K3 passes and K4 read the textures whose halo rows may still be in flight: interior first
    def poisson_denoise(self, p):
        return self._interior_first(lambda: self.inner.poisson_denoise(p))

    def compose(self, p):
        return self._interior_first(lambda: self.inner.compose(p))

    def final_compose(self, p):
        self.finish_halo()
        return self.inner.final_compose(p)

    def _interior_first(self, draw):
        if not self._halo_pending:
            return draw()
        y0, y1, h = self.tile_y0, self.tile_y0 + self.tile_rows, self.halo
        lo = y0 + (h if self.rank > 0 else 0)               # rows below `lo` / from `hi` on may read the halo rows
        hi = y1 - (h if self.rank < self.world - 1 else 0)
        if not self.overlap_halo_exchange or hi <= lo:
            self.finish_halo()
            return draw()
        try:
            self.inner.set_row_window(lo, hi)
            draw()
            self.finish_halo()
            if lo > y0:
                self.inner.set_row_window(y0, lo)
                draw()
            if hi < y1:
                self.inner.set_row_window(hi, y1)
                draw()
        finally:
            self.inner.set_row_window(0, 0)

    def download(self, *a, **k):
        self.finish_pending()
        self.finish_halo()
        return self.inner.download(*a, **k)

    def sync(self):
        self.finish_pending()
        self.finish_halo()
        return self.inner.sync()

    def finish_halo(self):
        pending, self._halo_pending = self._halo_pending, []
        for works, tensor in pending:
            for w in works:
                w.wait()
            self._sync_after_comm(tensor)

    def finish_pending(self):
        pending, self._pending = self._pending, []
        for works, tensor in pending:
            for w in works:
                w.wait()
            self._sync_after_comm(tensor)

    # ---- hooks called by rfx_amd.effect
    def after_temporal_pass(self):
        self.exchange((abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1))

    def after_denoise_pass(self, i, uniforms):
        self.exchange((abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1) if uniforms.writeToB else (abi.TEX_DENOISE_A0, abi.TEX_DENOISE_A1))

    def after_compose_pass(self):
        self.allgather_compose()

    def after_copy_framebuffer(self, tex):
        self.exchange((tex,))  # TRAA: the pass's own history (TemporalReprojectPass.js:197-201) is gathered at vUv - velocity next frame

    # ---- halo Send/Recv with the row neighbours
    def exchange(self, texs):
        if self.world == 1 or self.halo == 0:
            return
        dist = self._dist
        ops = []
        plan = halo_plan(self.H, self.world, self.rank, self.halo)
        for tex in texs:
            if tex not in self.tensors:
                raise KeyError("TiledRenderer.exchange: texture %s is not bound for exchange — pass it to bind_torch_buffers(texs=...) "
                               "(exchanged_textures(denoise_mode) lists what a configuration needs)" % abi.TEX_NAMES[tex])
            t = self.tensors[tex]
            b0, bn = self.inner.held_rows(tex)
            if b0 > max(0, self.tile_y0 - self.halo) or b0 + bn < min(self.H, self.tile_y0 + self.tile_rows + self.halo):
                raise ValueError("TiledRenderer.exchange: the held band of %s [%d, %d) does not contain %d halo rows around the tile" % (
                    abi.TEX_NAMES[tex], b0, b0 + bn, self.halo))
            for peer, send, recv in plan:
                if send:
                    ops.append(dist.P2POp(dist.isend, t[send[0] - b0:send[1] - b0], peer, self.group))
                if recv:
                    ops.append(dist.P2POp(dist.irecv, t[recv[0] - b0:recv[1] - b0], peer, self.group))
        self.finish_halo()  # one exchange in flight at a time (and the same order on every rank)
        self._sync_before_comm()
        self._halo_pending.append((dist.batch_isend_irecv(ops), self.tensors[texs[0]]))
        self.exchange_count += 1
        if not self.overlap_halo_exchange:
            self.finish_halo()

    def allgather_compose(self):
        if self.world == 1 or self.history_gather == "peer":
            return  # peer: the column blocks are pulled on demand, in before_ssgi_shade
        dist = self._dist
        full = self.tensors[abi.TEX_COMPOSE_RGB if self.gather_history_rgb else abi.TEX_COMPOSE]  # whole frame
        mine = full[self.tile_y0:self.tile_y0 + self.tile_rows]
        self.finish_pending()
        self._sync_before_comm()
        parts = [full[y0:y0 + n] for (y0, n) in split_rows(self.H, self.world)]
        if all(p.shape == mine.shape for p in parts) and full.is_cuda:
            works = [dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group, async_op=True)]
        else:  # ragged last tile, or a backend without all_gather_into_tensor: one broadcast per owner
            works = [dist.broadcast(p, src=r, group=self.group, async_op=True) for r, p in enumerate(parts)]
        self._pending.append((works, full))
        if not self.overlap_history_gather:
            self.finish_pending()

    def _sync_after_comm(self, tensor):
        # on device tensors `wait()` only makes torch's current stream wait for the collective; kernels of a context that
        # runs on ANOTHER stream must not start before the rows have landed
        if getattr(self.inner, "uses_torch_stream", False) or not tensor.is_cuda:
            return
        import torch
        torch.cuda.current_stream(tensor.device).synchronize()

    def _sync_before_comm(self):
        # kernels run on the context's stream; when that is torch's CURRENT stream (bench.py creates one, binds it with
        # rfx_set_stream and marks the context) the collectives are ordered after them automatically.  Anything else —
        # a context on its own stream — drains that stream here before its rows are sent.
        if getattr(self.inner, "uses_torch_stream", False):
            return
        self.inner.sync()


class CommTiledRenderer(TiledRenderer):
    """The same protocol with the exchanges BEHIND THE C ABI (include/rfx.h "row-tiled runs"): rfx_halo_exchange /
    rfx_allgather_history enqueue RCCL Send/Recv / all-gather on the context's exchange stream, ordered after the draws issued so
    far; rfx_comm_wait orders the following draws after them.  No torch, no bound external buffers: the textures stay the
    context's own.  `unique_id`: the 128 bytes of Context.comm_unique_id() made by rank 0 and handed to every rank."""

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes, denoise_mode: str = "full", history_gather: str = "all"):
        """history_gather "all" (default): the whole-frame all-gather of the composed GI after K4, overlapped with the next frame's trace.
        "bounded": no all-gather; between a frame's trace and its shade rfx_gather_history_rows moves only the rows the tiles' rays will
        read (include/rfx.h).  Same pixels either way.  Which one is faster depends on the scene: the bounded form moves fewer bytes
        (measured on the synthetic orbit at 4K: 67 % / 71 % of the all-gather's at N = 4 / 8, 100 % at N = 2 — reflections reach most of
        the frame below the horizon, profiles/r03_multigpu/history_rows_4k.txt) but sits on the critical path between trace and shade
        and needs one host-side wait for 2 N integers, while the all-gather hides under the next frame's trace."""
        if history_gather not in ("bounded", "all"):
            raise ValueError("history_gather: \"bounded\" or \"all\"")
        self._dist = None
        self._setup(ctx, {t: None for t in (exchanged_textures(denoise_mode) if world > 1 else ())}, rank, world)
        ctx.comm_init(unique_id, rank, world)
        self._comm_pending = False
        self.history_gather = history_gather if (world > 1 and self.overlap_history_gather) else "all"

    def exchange(self, texs):
        if self.world == 1 or self.halo == 0:
            return
        up = self.rank + 1 if self.rank + 1 < self.world else -1
        down = self.rank - 1
        for tex in texs:
            if tex not in self.tensors:
                raise KeyError("CommTiledRenderer.exchange: texture %s is not part of this run's exchange set" % abi.TEX_NAMES[tex])
            self.inner.halo_exchange(tex, up, down)
        self._halo_pending = [True]
        self.exchange_count += 1

    def _history_tex(self):
        return abi.TEX_COMPOSE_RGB if self.gather_history_rgb else abi.TEX_COMPOSE

    def allgather_compose(self):
        if self.world == 1 or self.history_gather in ("bounded", "peer"):
            return  # bounded / peer: the rows travel on demand, in before_ssgi_shade
        self.inner.allgather_history(self._history_tex())
        self._pending = [True]

    def before_ssgi_shade(self):
        if self.world > 1 and self.history_gather == "peer":
            return self._peer_before_shade()
        if self.world > 1 and self.history_gather == "bounded":
            self.history_bytes_received.append(self.inner.gather_history_rows(self._history_tex()))
            self._pending = [True]
        self.finish_pending()

    def ssgi_march(self, p):
        if self.world > 1 and self.history_gather in ("bounded", "peer"):
            raise RuntimeError("CommTiledRenderer(history_gather=\"bounded\"): K1 must run as ssgi_trace / ssgi_shade (the gather sits between them)")
        return super().ssgi_march(p)

    def gather_whole_history(self):
        """every rank's rows of the composed GI to every rank, now (a host that wants the whole frame on one rank — bench.py's checksum)"""
        if self.world > 1 and self.history_gather == "peer":
            return TiledRenderer.gather_whole_history(self)
        if self.world > 1:
            self.inner.allgather_history(self._history_tex())
            self._pending = [True]
            self.finish_pending()

    def finish_halo(self):
        if self._halo_pending or self._pending:
            self.inner.comm_wait()  # one event covers everything issued so far on the exchange stream
        self._halo_pending, self._pending = [], []

    finish_pending = finish_halo


def exchanged_textures(denoise_mode: str = "full"):
    """The textures a row-tiled SSGIEffect run exchanges, by Denoiser mode (Denoiser.js:41-61): modes without a denoise pass
    ("full_temporal", preset "low"; "temporal") keep K2's history in the pass's own RGBA32F framebuffer copy, which is then the
    texture whose halo rows travel after every frame."""
    if denoise_mode in ("full", "denoised"):
        return EXCHANGED + (abi.TEX_COMPOSE_RGB,)
    if denoise_mode == "full_temporal":
        return (abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_FBCOPY_F32, abi.TEX_COMPOSE_RGB)
    if denoise_mode == "temporal":  # K1's history is K2's texture[0], gathered anywhere: whole-frame contexts only (rfx.h historySource 1)
        raise ValueError("denoiseMode \"temporal\" cannot be row-tiled: K1 gathers K2's target anywhere on screen")
    raise ValueError("unknown denoiseMode %r" % (denoise_mode,))


def bind_torch_buffers(ctx, device, texs=None, denoise_mode: str = "full"):
    """Allocate the exchanged textures as torch tensors on `device` and bind them into the rfx
    context (rfx_bind_external), so torch.distributed can send/receive their rows in place.
    `texs` defaults to what the SSGI chain exchanges in `denoise_mode`; a TRAA run binds (TEX_FBCOPY_F16,) or (TEX_FBCOPY_F32,)."""
    import torch
    tensors = {}
    for tex in (exchanged_textures(denoise_mode) if texs is None else tuple(texs)):
        r0, n = ctx.held_rows(tex)
        dtype, ch = abi.TEX_FORMAT[tex]
        nbytes = np.dtype(dtype).itemsize * ch * ctx.W
        t = torch.zeros((n, nbytes), dtype=torch.uint8, device=device)
        ctx.bind_external(tex, t.data_ptr())
        tensors[tex] = t
    return tensors
