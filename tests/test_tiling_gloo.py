"""CPU, world_size 2 and 3 over gloo (3: a middle rank with two neighbours, ragged tiles -> the per-owner broadcast form of the gather): the row-tiled chain (halo Send/Recv after K2 and every K3 pass,
all-gather of the composed GI; for TRAAEffect the Send/Recv of its framebuffer copy) must be BIT-IDENTICAL to the single-tile chain.  The per-tile
compute is the oracle double (tests only); the exchange code is the product's (rfx_amd.tiling)."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
W, H, FRAMES = 96, 64, 2


def _chain(renderer, scene, cam, frames, **extra):
    from rfx_amd.effect import SSGIEffect
    fx = SSGIEffect(None, scene, cam, dict(width=W, height=H, denoiseIterations=1, **extra), seeds=dict(ssgi=5, denoise=9))
    for f in frames:
        scene.frame = f
        for k, v in vars(f.camera).items():
            setattr(cam, k, v)
        fx.update(renderer, None)


def _traa(renderer, scene, cam, frames):
    from rfx_amd.effect import HalfFloatType, TRAAEffect, VelocityDepthNormalPass
    fx = TRAAEffect(scene, cam, VelocityDepthNormalPass(scene, cam), dict(fullAccumulate=True))
    for f in frames:
        scene.frame = f
        for k, v in vars(f.camera).items():
            setattr(cam, k, v)
        fx.update(renderer, dict(texture=dict(type=HalfFloatType), width=W, height=H, data=f.direct))


def _worker(rank, world, port, outdir, halo_override=0):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401  (sys.path setup)
    from oracle_renderer import OracleRenderer
    from rfx_amd import abi, tiling
    from rfx_amd.scene import synthetic_frame

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    vmax = max(float(np.abs(f.velocity[..., 1].view(np.float32)).max()) for f in frames)
    halo = halo_override or tiling.required_halo(3.0, vmax, H, W)
    y0, rows = tiling.split_rows(H, world)[rank]
    inner = OracleRenderer(W, H, y0, rows, halo)
    tensors = {}
    for tex in tiling.EXCHANGED + (abi.TEX_COMPOSE_RGB,):
        b0, n = inner.held_rows(tex)
        tensors[tex] = torch.from_numpy(inner.tex[tex][b0:b0 + n])  # shares memory with the renderer's planes
    r = tiling.TiledRenderer(inner, tensors, rank, world)
    scene = types.SimpleNamespace(frame=None)
    _chain(r, scene, frames[0].camera, frames)
    assert r.exchange_count == FRAMES * 3  # after K2, after K3 pass 0, after K3 pass 1
    # K1 ran as trace + shade with the (asynchronous) composed-GI all-gather waited for in between; the last one is still pending
    assert r.gather_history_rgb and r.overlap_history_gather and sum(1 for c in inner.calls if c[0] == "ssgi_trace") == FRAMES
    assert len(r._pending) == 1
    r.finish_pending()
    # the halo exchanges are asynchronous too: K3 passes and K4 drew their interior first (windowed launches), then the boundary strips
    # (a halo of half the tile or more leaves no interior: the draw then simply waits first)
    assert r.overlap_halo_exchange and (2 * halo >= rows or sum(1 for c in inner.calls if c[0] == "set_row_window") >= FRAMES * 3 * 3)
    r.finish_halo()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), y0=y0, rows=rows, halo=halo,
             **{abi.TEX_NAMES[t]: inner.tex[t][y0:y0 + rows] for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0,
                                                                     abi.TEX_DENOISE_B1)},
             compose_tile=inner.tex[abi.TEX_COMPOSE][y0:y0 + rows], compose_rgb_full=inner.tex[abi.TEX_COMPOSE_RGB])
    # TRAAEffect on the same tiles: one exchange per frame, of the pass's own history
    inner2 = OracleRenderer(W, H, y0, rows, tiling.required_halo(0.0, vmax, H, W))
    b0, n = inner2.held_rows(abi.TEX_FBCOPY_F16)
    r2 = tiling.TiledRenderer(inner2, {abi.TEX_FBCOPY_F16: torch.from_numpy(inner2.tex[abi.TEX_FBCOPY_F16][b0:b0 + n])}, rank, world)
    _traa(r2, types.SimpleNamespace(frame=None), types.SimpleNamespace(**vars(frames[0].camera)), frames)
    assert r2.exchange_count == FRAMES
    r2.finish_halo()
    np.save(os.path.join(outdir, "traa%d.npy" % rank), inner2.tex[abi.TEX_TEMPORAL0][y0:y0 + rows])
    # denoiseMode "full_temporal" (preset "low": no denoise pass): K2's history is its own RGBA32F framebuffer copy, whose halo rows
    # travel after every frame — bound through tiling.exchanged_textures("full_temporal")
    inner3 = OracleRenderer(W, H, y0, rows, halo)
    t3 = {}
    for tex in tiling.exchanged_textures("full_temporal"):
        b0, n = inner3.held_rows(tex)
        t3[tex] = torch.from_numpy(inner3.tex[tex][b0:b0 + n])
    r3 = tiling.TiledRenderer(inner3, t3, rank, world)
    _chain(r3, types.SimpleNamespace(frame=None), types.SimpleNamespace(**vars(frames[0].camera)), frames, denoiseMode="full_temporal")
    r3.finish_pending()
    r3.finish_halo()
    np.savez(os.path.join(outdir, "ft%d.npz" % rank), t0=inner3.tex[abi.TEX_TEMPORAL0][y0:y0 + rows], t1=inner3.tex[abi.TEX_TEMPORAL1][y0:y0 + rows],
             compose=inner3.tex[abi.TEX_COMPOSE][y0:y0 + rows])
    # an exchange of a texture that was not bound names the slot instead of raising a bare KeyError
    try:
        r2.exchange((abi.TEX_FBCOPY_F32,))
        raise AssertionError("exchange of an unbound texture went through")
    except KeyError as e:
        assert "fbcopy_f32" in str(e)
    dist.barrier()
    dist.destroy_process_group()


def _comm_worker(rank, world, port, outdir, halo_override=0):
    """CommTiledRenderer (the protocol over the C ABI's exchange entry points) on a stand-in context whose halo_exchange /
    allgather_history are executed LAZILY, at comm_wait, over gloo: the adversarial schedule — rows arrive as late as the protocol
    allows and are read from the sender's buffers as late as it allows.  A draw that touched halo rows before its wait, or rewrote
    rows that were still to be sent, would change the result."""
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    from oracle_renderer import OracleRenderer
    from rfx_amd import abi, tiling
    from rfx_amd.scene import synthetic_frame

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class LazyCommCtx(OracleRenderer):
        def __init__(self, *a):
            super().__init__(*a)
            self.queue, self.waits = [], 0

        def comm_init(self, uid, r, n):
            assert tiling.split_rows(self.H, n)[r] == (self.tile_y0, self.tile_rows)

        def halo_exchange(self, tex, up, down):
            self.queue.append(("halo", tex, up, down))

        def allgather_history(self, tex):
            self.queue.append(("gather", tex))

        def comm_wait(self):
            self.waits += 1
            q, self.queue = self.queue, []
            for op in q:  # same order on every rank
                if op[0] == "halo":
                    _, tex, up, down = op
                    t = torch.from_numpy(self.tex[tex])
                    ops = []  # what rfx_halo_exchange does (csrc/rfx_comm.hip): the rows of every tile inside this tile's band, both ways
                    for peer, send, recv in tiling.halo_plan(self.H, dist.get_world_size(), dist.get_rank(), self.halo):
                        if (peer > dist.get_rank() and up < 0) or (peer < dist.get_rank() and down < 0):
                            continue
                        if send:
                            ops.append(dist.P2POp(dist.isend, t[send[0]:send[1]].contiguous(), peer))
                        if recv:
                            ops.append(dist.P2POp(dist.irecv, t[recv[0]:recv[1]], peer))
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                else:
                    t = torch.from_numpy(self.tex[op[1]])
                    for r, (ty0, tn) in enumerate(tiling.split_rows(self.H, dist.get_world_size())):
                        dist.broadcast(t[ty0:ty0 + tn], src=r)

    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    vmax = max(float(np.abs(f.velocity[..., 1].view(np.float32)).max()) for f in frames)
    halo = halo_override or tiling.required_halo(3.0, vmax, H, W)
    y0, rows = tiling.split_rows(H, world)[rank]
    inner = LazyCommCtx(W, H, y0, rows, halo)
    r = tiling.CommTiledRenderer(inner, rank, world, b"\0" * 128)
    _chain(r, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    r.sync()
    assert r.exchange_count == FRAMES * 3 and inner.waits >= FRAMES * 4 and not inner.queue
    np.savez(os.path.join(outdir, "comm%d.npz" % rank), y0=y0, rows=rows,
             **{abi.TEX_NAMES[t]: inner.tex[t][y0:y0 + rows] for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0,
                                                                     abi.TEX_DENOISE_B1)},
             compose_tile=inner.tex[abi.TEX_COMPOSE][y0:y0 + rows], compose_rgb_full=inner.tex[abi.TEX_COMPOSE_RGB])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,halo", [(2, 0), (3, 0), (4, 20)])  # (4, 20): 16-row tiles under a 20-row halo — rows from two tiles away
def test_comm_tiled_protocol_is_bit_identical_under_the_latest_possible_delivery(tmp_path, world, halo):
    import socket
    from oracle_renderer import OracleRenderer
    from rfx_amd import abi
    from rfx_amd.scene import synthetic_frame
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_comm_worker, args=(world, port, str(tmp_path), halo), nprocs=world, join=True)
    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    ref = OracleRenderer(W, H)
    _chain(ref, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "comm%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1):
            assert np.array_equal(z[abi.TEX_NAMES[t]], ref.tex[t][y0:y0 + rows]), "rank %d %s differs" % (rank, abi.TEX_NAMES[t])
        assert np.array_equal(z["compose_rgb_full"], ref.tex[abi.TEX_COMPOSE][..., :3]), "rank %d gathered composed GI differs" % rank
        assert np.array_equal(z["compose_tile"], ref.tex[abi.TEX_COMPOSE][y0:y0 + rows])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,halo", [(2, 0), (3, 0), (4, 20)])  # 3: a middle rank with two neighbours, a ragged last tile; (4, 20): a halo taller than the 16-row tiles
def test_tiled_chain_is_bit_identical(tmp_path, world, halo):
    import socket
    from oracle_renderer import OracleRenderer
    from rfx_amd import abi
    from rfx_amd.scene import synthetic_frame

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), halo), nprocs=world, join=True)

    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    ref = OracleRenderer(W, H)
    _chain(ref, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1):
            assert np.array_equal(z[abi.TEX_NAMES[t]], ref.tex[t][y0:y0 + rows]), "rank %d %s differs" % (rank, abi.TEX_NAMES[t])
        # every rank ends with .rgb of the WHOLE composed frame (next frame's K1 gathers it anywhere), and its own tile of the target
        assert np.array_equal(z["compose_rgb_full"], ref.tex[abi.TEX_COMPOSE][..., :3]), "rank %d gathered composed GI differs" % rank
        assert np.array_equal(z["compose_tile"], ref.tex[abi.TEX_COMPOSE][y0:y0 + rows]), "rank %d compose tile differs" % rank
    ref2 = OracleRenderer(W, H)
    _traa(ref2, types.SimpleNamespace(frame=None), types.SimpleNamespace(**vars(frames[0].camera)), frames)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "traa%d.npy" % rank)), ref2.tex[abi.TEX_TEMPORAL0][y0:y0 + rows]), "rank %d TRAA differs" % rank


    ref3 = OracleRenderer(W, H)
    _chain(ref3, types.SimpleNamespace(frame=None), types.SimpleNamespace(**vars(frames[0].camera)), frames, denoiseMode="full_temporal")
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        ft = np.load(os.path.join(str(tmp_path), "ft%d.npz" % rank))
        assert np.array_equal(ft["t0"], ref3.tex[abi.TEX_TEMPORAL0][y0:y0 + rows]) and np.array_equal(ft["t1"], ref3.tex[abi.TEX_TEMPORAL1][y0:y0 + rows])
        assert np.array_equal(ft["compose"], ref3.tex[abi.TEX_COMPOSE][y0:y0 + rows]), "rank %d full_temporal compose differs" % rank


def test_halo_plan_covers_every_band_and_pairs_every_send():
    """tiling.halo_plan (and rfx_halo_exchange, which follows the same rule): for any split and halo, what a rank receives is exactly its
    band minus its tile, from the owners; every send has the matching receive on the peer; halo <= tile height = the two neighbours only."""
    from rfx_amd import tiling
    for H, world, halo in [(64, 4, 5), (64, 4, 16), (64, 4, 20), (64, 4, 40), (70, 3, 30), (128, 8, 50), (64, 2, 100)]:
        tiles = tiling.split_rows(H, world)
        plans = [tiling.halo_plan(H, world, r, halo) for r in range(world)]
        for r, (y0, n) in enumerate(tiles):
            got = np.zeros(H, int)
            for peer, send, recv in plans[r]:
                if recv:
                    assert tiles[peer][0] <= recv[0] and recv[1] <= tiles[peer][0] + tiles[peer][1]  # the owner's rows
                    got[recv[0]:recv[1]] += 1
                    assert (r, recv, None) in [(q, s, None) for q, s, _ in plans[peer]], "no send for a receive"
                if send:
                    assert y0 <= send[0] and send[1] <= y0 + n
                    assert any(q == r and rc == send for q, _, rc in plans[peer]), "no receive for a send"
            want = np.zeros(H, int)
            want[max(0, y0 - halo):min(H, y0 + n + halo)] = 1
            want[y0:y0 + n] = 0
            assert np.array_equal(got, want), (H, world, halo, r)
            if halo <= min(m for _, m in tiles):
                assert {p for p, _, _ in plans[r]} <= {r - 1, r + 1}


def test_tiled_renderer_rejects_geometry_it_cannot_exchange():
    """a tile that is not the rank's share of the split must be refused, not silently wrong (a halo taller than a tile is
    exchanged since round 3: test_tiled_chain_is_bit_identical[4-20])."""
    from rfx_amd import tiling
    inner = types.SimpleNamespace(W=64, H=64, tile_y0=10, tile_rows=16, halo=2)
    with pytest.raises(ValueError, match="split_rows"):
        tiling.TiledRenderer(inner, {}, 1, 4)
    with pytest.raises(ValueError):
        tiling.split_rows(6, 8)
    with pytest.raises(ValueError, match="temporal"):
        tiling.exchanged_textures("temporal")


def test_split_rows_is_one_rule_in_c_python_and_node():
    """rfx_split_rows (the C ABI's tile rule, used by rfx_comm_init and the Node host) == tiling.split_rows (Python) == js splitRows."""
    import json
    import shutil
    import subprocess
    from rfx_amd import tiling
    from rfx_amd.context import Context
    cases = [(2160, 8), (2160, 3), (1080, 7), (4320, 8), (90, 4), (64, 1)]
    for Hh, n in cases:
        assert [Context.split_rows(Hh, n, r) for r in range(n)] == tiling.split_rows(Hh, n)
    with pytest.raises(ValueError):
        Context.split_rows(6, 8, 0)
    node = shutil.which("node")
    addon = os.path.join(HERE, "..", "realism-effects_amd", "napi", "rfx_napi.node")
    if node and os.path.exists(addon):
        js = "const a=require(%r);console.log(JSON.stringify(%s.map(c=>Array.from({length:c[1]},(_,r)=>a.splitRows(c[0],c[1],r)))))" % (
            os.path.abspath(addon), json.dumps(cases))
        got = json.loads(subprocess.check_output([node, "-e", js], text=True))
        assert got == [[list(t) for t in tiling.split_rows(Hh, n)] for Hh, n in cases]


def test_split_rows_even_boundaries_and_halo():
    from rfx_amd import tiling
    for Hh, n in ((2160, 8), (2160, 4), (1080, 8), (90, 4), (4320, 8)):
        tiles = tiling.split_rows(Hh, n)
        assert sum(r for _, r in tiles) == Hh and all(y % 2 == 0 for y, _ in tiles)
        assert tiles[0][0] == 0 and all(tiles[i][0] + tiles[i][1] == tiles[i + 1][0] for i in range(n - 1))
    assert tiling.required_halo(3.0, 0.0, 2160, 3840) == 5
    assert tiling.required_halo(3.0, 0.005, 2160, 3840) == 15
    assert tiling.required_halo(3.0, 0.0, 3840, 2160) == 8  # portrait: the UV-space rotation stretches the tap footprint vertically


# ---------------------------------------------------------------- the same, with the product's KERNELS under the tiles (pytest --hostsim)
def _kernel_worker(rank, world, port, outdir):
    """one process per tile, each with its own rfx context (tests/hostsim: the kernel sources on the CPU), exchanging through the product's
    rfx_amd.tiling over gloo — the device memory the tensors are bound to IS host memory there"""
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    from rfx_amd import abi, tiling
    from rfx_amd.context import Context
    from rfx_amd.scene import synthetic_frame

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    vmax = max(float(np.abs(f.velocity[..., 1].view(np.float32)).max()) for f in frames)
    halo = tiling.required_halo(3.0, vmax, H, W)
    y0, rows = tiling.split_rows(H, world)[rank]
    ctx = Context(W, H, tile_y0=y0, tile_rows=rows, halo_rows=halo)
    r = tiling.TiledRenderer(ctx, tiling.bind_torch_buffers(ctx, "cpu"), rank, world)
    _chain(r, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    r.finish_pending()
    r.finish_halo()
    assert ctx.halo_violations() == 0
    np.savez(os.path.join(outdir, "k%d.npz" % rank), y0=y0, rows=rows,
             **{abi.TEX_NAMES[t]: ctx.download(t, y0, rows) for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1, abi.TEX_COMPOSE)},
             compose_rgb_full=ctx.download(abi.TEX_COMPOSE_RGB))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("RFX_HOSTSIM") != "1", reason="the kernels under gloo tiles: pytest --hostsim (on a GPU box the multi-rank flow test of test_gpu_parity.py does this)")
@pytest.mark.parametrize("world", [2, 3])
def test_tiled_kernels_are_bit_identical_to_one_context(tmp_path, world):
    import socket
    from rfx_amd import abi
    from rfx_amd.context import Context
    from rfx_amd.scene import synthetic_frame

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_kernel_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    ref = Context(W, H)
    _chain(ref, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "k%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1, abi.TEX_COMPOSE):
            assert np.array_equal(z[abi.TEX_NAMES[t]], ref.download(t, y0, rows)), "rank %d %s differs" % (rank, abi.TEX_NAMES[t])
        assert np.array_equal(z["compose_rgb_full"], ref.download(abi.TEX_COMPOSE)[..., :3]), "rank %d gathered composed GI differs" % rank
    ref.close()



@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("RFX_HOSTSIM") != "1", reason="the C ABI's exchanges between processes without RCCL: pytest --hostsim")
@pytest.mark.parametrize("world,mode,halo", [(2, "bounded", 0), (3, "bounded", 0), (4, "bounded", 0), (8, "bounded", 0), (3, "all", 0), (4, "all", 0), (4, "all", 20), (3, "bounded", 30)])
def test_tiled_kernels_with_c_abi_exchanges_are_bit_identical_to_one_context(tmp_path, world, mode, halo):
    """The same tiles with the exchanges BEHIND THE C ABI (rfx_comm_init / rfx_halo_exchange / rfx_allgather_history / rfx_comm_wait), one
    process per tile (tests/comm_tile_worker.py — no torch in them: a torch process maps the real librccl.so.1, which rfx_comm.hip would
    rightly reuse).  Under --hostsim the librccl.so.1 that rfx_comm.hip binds is tests/hostsim/fakerccl.c: unix sockets between these
    processes.  Ragged tiles at 3 ranks (the grouped-broadcast form of the gather), even ones at 2 and 4 (the in-place all-gather).
    mode "bounded" (the default): no all-gather of the composed GI; between a frame's trace and its shade rfx_gather_history_rows moves only
    the rows the tiles' rays will read — same pixels, fewer bytes; "all": the whole-frame all-gather after K4.
    halo > 0: a halo taller than the tiles (16 / 20 rows here): rfx_halo_exchange then moves rows between tiles that are not neighbours."""
    import subprocess
    from rfx_amd import abi
    from rfx_amd.context import Context
    from rfx_amd.scene import synthetic_frame

    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "comm_tile_worker.py"), str(r), str(world), str(tmp_path), str(W), str(H), str(FRAMES), mode, str(halo)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
    frames = [synthetic_frame(W, H, i) for i in range(FRAMES)]
    ref = Context(W, H)
    _chain(ref, types.SimpleNamespace(frame=None), frames[0].camera, frames)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "c%d.npz" % rank))
        y0, rows = int(z["y0"]), int(z["rows"])
        for t in (abi.TEX_SSGI, abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1, abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1, abi.TEX_COMPOSE):
            assert np.array_equal(z[abi.TEX_NAMES[t]], ref.download(t, y0, rows)), "rank %d %s differs" % (rank, abi.TEX_NAMES[t])
        assert np.array_equal(z["compose_rgb_full"], ref.download(abi.TEX_COMPOSE)[..., :3]), "rank %d gathered composed GI differs" % rank
        if mode == "bounded":  # never more than the all-gather would deliver (the other tiles' rows), and recorded once per frame
            assert len(z["history_bytes"]) == FRAMES and (z["history_bytes"] <= (H - rows) * W * 12).all(), z["history_bytes"]
            print("rank %d of %d receives %s bytes of composed GI per frame (all-gather: %d)" % (rank, world, list(z["history_bytes"]), (H - rows) * W * 12))
    ref.close()
