#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X: Mpixels/s of the SSGI chain at 4K.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no RANK in the environment the command launches itself, one rank per GPU, as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(which is also how the driver may launch it directly).

One "step" = one frame through SSGIEffect.update(): K1 SSGI march (steps 20 / refineSteps 5) ->
K2 temporal reprojection -> 2 x K3 Poisson denoise (denoiseIterations 1) -> K4 compose, over a
3840x2160 synthetic G-buffer dump (seed 1234) that is ALREADY RESIDENT in HBM when the timed
region starts.
  N = 1: BASELINE.json configs[2], the 4K frame on one GPU.
  N > 1: configs[3] — THE SAME 4K frame cut into N row tiles (STRONG scaling: 1080 / 540 / 270 rows per GPU), which exchange halo rows
         with their neighbours over RCCL after K2 and after every K3 pass, plus the gather of the composed GI that the next frame's K1
         shading reads (asynchronous, under the next frame's depth pre-pass + ray march: K1 runs as rfx_ssgi_trace / rfx_ssgi_shade).
         The round-1 weak-scaling case and configs[4] (8K, steps 40, denoiseIterations 3) ride along as extra keys.

Timing: `--spinup` untimed frames first (default 200 = 0.3 s: the device sat idle through the ~20 s of dump generation, and its first tens of
milliseconds run below the sustained clock — with --steps 20 --warmup 5 alone the step measured 1.556 ms against 1.51 sustained), then the W
untimed warm-up steps, then EXACTLY K timed steps between barriers and device synchronisation, max over ranks.  The line says `spinup_frames`.

Prints ONE JSON line (rank 0).  `kernel_ms` are the per-draw durations INSIDE the frame loop (rfx_profile: hipEvents around every draw's
launches on the stream they run on, over K more frames after the timed region — they sum to `ms_per_step`; K1's depth pre-pass runs on
its own stream under the previous frame's later draws and is listed beside them); `roofline` is for the dominant kernel — K1, whose 68 B/px
include the depth plane its pre-pass reads, so its duration there is march + pre-pass — next to a device-to-device stream-copy rate measured
in the same process; `issue_model` is, per kernel, the
absolute time its measured instruction mix costs to issue next to the time it took (rocprofv3 counters of this same command, committed under
PROFILE_DIR; tools/issue_model.py) — the bound that actually binds; `ms_per_step_cold` is the same W + K protocol before the spin-up;
`cpu_baseline` is the REFERENCE's own GLSL on Mesa llvmpipe on this box's host cores over the same frame (kind "reference"), with the C
restatement under OpenMP on all cores as `cpu_baseline.port` (SURVEY.md §8d's second baseline; alone when no GL is available).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))

import numpy as np  # noqa: E402

from rfx_amd import abi, tiling  # noqa: E402
from rfx_amd.context import Context  # noqa: E402
from rfx_amd.effect import SSGIEffect  # noqa: E402
from rfx_amd.scene import synthetic_band_parallel  # noqa: E402

W4K, H4K = 3840, 2160
HISTORY_GATHER = ["all"]  # --history-gather
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# algorithmic bytes per pixel per launch, reference texel formats (SURVEY.md §8d / DESIGN.md)
BYTES_PER_PX = {"k1_ssgi_march": 68, "k2_temporal_reproject": 80, "k3_poisson_denoise_pass0": 68, "k3_poisson_denoise_pass1": 52, "k4_compose": 52}


# rocprofv3 kernel-name fragments of bench.py's kernel keys (profiles/*/pmc_hbm.csv)
# (every fragment of a tuple must occur in the name; k3_tiled<IN_TEMPORAL, textures, LDS pitch, WHOLE>)
PMC_KERNEL = {"k1_ssgi_march": ("false, 0>(K1Args)",), "k2_temporal_reproject": ("k2_temporal_reproject",), "k3_poisson_denoise_pass0": ("k3_tiled<true",),
              "k3_poisson_denoise_pass1": ("k3_tiled<false",), "k4_compose": ("k4_compose",), "k1_prepass": ("k1_prepare",)}


def _is_kernel(key, name):
    return all(f in name for f in PMC_KERNEL[key])


TRAFFIC_NOTE = ("fabric-side bytes per launch (L2 misses: Infinity-Cache hits included) from %s/pmc_hbm.csv — separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                "passes over this bench command at 4K, tools/collect_profiles.sh, collected at git %s; FETCH_SIZE calibrated on known request counts "
                "(profiles/r05_microbench): see traffic_calibration")
PROFILE_DIR = "profiles/r06_final"  # the committed rocprofv3 collection (tools/collect_profiles.sh) the counter-derived figures are read from


def profile_meta():
    """which collection (directory, git commit, date) the PMC-derived figures of the line come from"""
    try:
        return json.load(open(os.path.join(ROOT, PROFILE_DIR, "meta.json")))
    except Exception:  # noqa: BLE001
        return {}


# What FETCH_SIZE counts on this part (profiles/r05_microbench/: known request counts under rocprofv3 --pmc, kernel durations beside them): every
# L2-to-fabric read request as 64 bytes.  Wide coalesced reads (4 or 16 B per lane over consecutive lanes: 1 GiB read once reports 512 MiB) travel
# as 128-byte requests -> the counter shows HALF their bytes (the guide's x2); a 4-byte gather that misses travels as ONE 64-byte request (2^24
# of them report 2^24 x 64 B and take 303 us = 3.5 TB/s; 128-byte requests would be 7.1 TB/s, above what the part sustains) -> the counter shows
# its bytes.  So: traffic = streamed bytes + (FETCH_SIZE - streamed bytes / 2) + WRITE_SIZE, the kernel's STREAMED reads being known exactly
# (every plane it reads once, row by row).  K1: view-Z 4 + depth 4 + G-buffer 16 + direct light 16 = 40 B/px streamed; its march taps and the
# history fetch at the hit point are the gathers.  The LDS-tiled kernels read rows of their tile and apron (wide loads): x2 throughout — an upper
# bound for K2, whose history taps are 8-byte gathers.
STREAMED_READ_BPP = {"k1_ssgi_march": 40}


def pmc_traffic(kernel_key, pixels=W4K * H4K):
    """Fabric-side bytes per launch of a kernel from the committed PMC summary of this same command (PROFILE_DIR/pmc_hbm.csv: separate
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, values in KiB), calibrated as above.  -> (bytes, how) or (None, None) when the
    summary is missing."""
    import csv
    path = os.path.join(ROOT, PROFILE_DIR, "pmc_hbm.csv")
    if not os.path.exists(path):
        return None, None
    vals = {}
    for r in csv.DictReader(open(path)):
        if _is_kernel(kernel_key, r["kernel"]):
            vals[r["counter"]] = float(r["mean_value_KB"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None
    fetch, write = vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    if kernel_key in STREAMED_READ_BPP:
        streamed = STREAMED_READ_BPP[kernel_key] * pixels
        gathers = max(fetch - streamed / 2.0, 0.0)
        return int(streamed + gathers + write), ("streamed reads %.0f MB (counted at half) + gather requests %.0f MB (64 B each, counted in full) + writes %.0f MB; "
                                                 "the uniform x2 convention of rounds 2-4 would say %.0f MB" % (streamed / 1e6, gathers / 1e6, write / 1e6, (2 * fetch + write) / 1e6))
    return int(2.0 * fetch + write), "2 x FETCH_SIZE + WRITE_SIZE (wide tile-row loads; an upper bound where the kernel also gathers)"


def issue_model():
    """kernel key -> tools/issue_model.py's figures from the committed collection (None when the class counters are not there)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import issue_model as IM
        rows = IM.table(os.path.join(ROOT, PROFILE_DIR))
    except Exception as e:  # noqa: BLE001
        log("issue model not available: %r" % (e,))
        return None
    out = {}
    for key in PMC_KERNEL:
        for name, m in rows.items():
            if _is_kernel(key, name):
                out[key] = {k: m[k] for k in ("valu_per_px", "predicted_issue_ms", "measured_ms", "issue_share_of_measured", "clock_GHz", "rate_int", "rate_other",
                                              "lds_busy_ms", "lds_busy_share_of_measured", "lds_conflict_share") if k in m}
    return out or None


def stream_copy_gbs(dev, nbytes=1 << 30, iters=10):
    """Device-to-device copy rate measured here and now (read + write bytes / time): the practical HBM ceiling next to the 8 TB/s spec."""
    import torch
    try:
        a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        del a, b
        torch.cuda.empty_cache()
        return round(2.0 * nbytes / (ms * 1e-3) / 1e9, 1)
    except Exception as e:  # noqa: BLE001  (the host simulator has no device allocator)
        log("stream copy not measured: %r" % (e,))
        return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def ctl_device(dist, dev):
    """control-plane tensors (barrier payloads, the max-over-ranks reduction, set-up gathers) live where the process group's backend wants them"""
    return dev if (dist is not None and dist.get_backend() == "nccl") else "cpu"


def build_case(world, rank, local_rank, dev, dist, one_gpu, W, H, tiles, steps, refine, iterations, use_c=False, group=None):
    """Dump + context + effect of one benchmark case on this rank: frame W x H cut into `tiles` (one per rank).  Returns a dict with
    the step function, the context and what the JSON line reports about the case."""
    import torch
    y0, rows = tiles[rank]
    t0 = time.time()
    nproc = max(1, min(32, len(os.sched_getaffinity(0)) // max(world, 1)))  # the dump is ray-cast on the host cores, split over the ranks
    opts = dict(width=W, height=H, steps=steps, refineSteps=refine, denoiseIterations=iterations)
    # frame 1 of the orbit: non-zero velocity (camera moved 0.5 deg since frame 0).  The tile is dumped first, the
    # velocity bound over ALL tiles fixes the halo width, then the halo rows are dumped and attached.
    tile = synthetic_band_parallel(W, H, 1, y0, rows, workers=nproc)
    vmax = float(np.abs(tile.velocity[..., 1].view(np.float32)).max())
    if dist is not None:  # every rank must use the SAME halo: the neighbours' send/recv sizes have to match
        t = torch.tensor([vmax], dtype=torch.float64, device=ctl_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vmax = float(t.item())
    halo = 0 if world == 1 else tiling.required_halo(3.0, vmax, H, W)
    b0, b1 = max(0, y0 - halo), min(H, y0 + rows + halo)
    parts = [tile]
    if b0 < y0:
        parts.insert(0, synthetic_band_parallel(W, H, 1, b0, y0 - b0, workers=nproc))
    if b1 > y0 + rows:
        parts.append(synthetic_band_parallel(W, H, 1, y0 + rows, b1 - y0 - rows, workers=nproc))
    band = types.SimpleNamespace(camera=tile.camera, **{k: np.concatenate([getattr(q, k) for q in parts], axis=0)
                                                         for k in ("depth", "gbuffer", "velocity", "direct")})
    log("[rank %d] dump band rows [%d,%d) of %dx%d generated in %.1fs (halo %d)" % (rank, b0, b1, W, H, time.time() - t0, halo))

    ctx = Context(W, H, device=local_rank, tile_y0=y0, tile_rows=rows, halo_rows=halo)
    renderer, exchange = ctx, "none"
    depth_full = band.depth
    if world > 1:
        # set-up only: the dump's depth plane is held whole by every rank.  Every rank contributes its rows of a zero
        # frame and the sum is the frame (works for ragged tiles on every backend)
        full = torch.zeros((H, W), dtype=torch.float32, device=ctl_device(dist, dev))
        full[y0:y0 + rows] = torch.from_numpy(np.ascontiguousarray(band.depth[y0 - b0:y0 - b0 + rows]))
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
        depth_full = full.cpu().numpy()
        if use_c:
            box = [Context.comm_unique_id() if rank == 0 else None]  # one ncclUniqueId per communicator
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
            # the exchanges behind the C ABI: RCCL Send/Recv + all-gather on the context's own exchange stream (rfx.h "row-tiled runs")
            renderer = tiling.CommTiledRenderer(ctx, rank, world, uid, history_gather="all" if HISTORY_GATHER[0] == "peer" else HISTORY_GATHER[0])
            exchange = "C ABI: rfx_halo_exchange / rfx_allgather_history (RCCL, own stream, overlapped)"
            bad = [verify_exchange(ctx, rank, world)]
            flags = [None] * world
            dist.all_gather_object(flags, bad[0])  # every rank must take the same path
            if any(f is not None for f in flags):
                ctx.close()
                raise RuntimeError("exchange pre-flight check failed: %s" % [f for f in flags if f is not None][:2])
        else:
            # torch.distributed transport (gloo in the one-GPU functional mode, or the NCCL backend): kernels, collectives and torch's
            # copies share ONE created stream — handle 0 would mean "the context's own stream" to rfx_set_stream
            stream = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(stream)
            ctx.set_stream(stream.cuda_stream)
            ctx.uses_torch_stream = not one_gpu  # gloo stages device tensors through the host: drain the stream around every exchange there
            # (the peer pull reads the library's own RGB twin through IPC mappings: not bound to a torch tensor then)
            texs = tiling.EXCHANGED if HISTORY_GATHER[0] == "peer" else None
            renderer = tiling.TiledRenderer(ctx, tiling.bind_torch_buffers(ctx, dev, texs=texs), rank, world, group=group)
            exchange = "torch.distributed (%s)" % ("gloo, one-GPU functional mode" if one_gpu else "nccl = RCCL")
        if HISTORY_GATHER[0] == "peer":
            def all_gather_object(obj):
                out = [None] * world
                dist.all_gather_object(out, obj, group=group)
                return out
            renderer.use_peer_history(all_gather_object)
            exchange += "; composed GI: rfx_peer_gather_history (peer loads through IPC mappings, device-driven)"
    # static: the same dump every step, uploaded once before the timed region (the metric is quoted with inputs resident in HBM)
    frame = types.SimpleNamespace(depth=depth_full, gbuffer=band.gbuffer, velocity=band.velocity, direct=band.direct, camera=band.camera, static=True)
    scene = types.SimpleNamespace(frame=frame)
    fx = SSGIEffect(None, scene, band.camera, opts, seeds=dict(ssgi=1, denoise=2), half_store_rtz=True)
    return dict(ctx=ctx, renderer=renderer, fx=fx, frame=frame, halo=halo, rows=rows, W=W, H=H, exchange=exchange, steps=steps, refine=refine, it=iterations)


def verify_exchange(ctx, rank, world):
    """Pre-flight check of the C ABI's RCCL exchanges on this communicator, with known patterns (nothing the frame loop needs survives
    it): every rank fills ITS tile rows of a K3 target with rank + 1, exchanges halos, and must find every halo row holding its owner's
    value; then the same for the all-gather of the composed-GI twin.  Returns None, or a description of what is wrong."""
    from rfx_amd import abi as A
    y0, rows, h, H, W = ctx.tile_y0, ctx.tile_rows, ctx.halo, ctx.H, ctx.W
    up, down = (rank + 1 if rank + 1 < world else -1), rank - 1
    r0, n = ctx.held_rows(A.TEX_DENOISE_A0)
    band = np.zeros((n, W, 4), np.uint16)
    band[y0 - r0:y0 - r0 + rows] = rank + 1
    ctx.upload(A.TEX_DENOISE_A0, band, r0, n)
    ctx.halo_exchange(A.TEX_DENOISE_A0, up, down)
    ctx.comm_wait()
    ctx.sync()
    got = ctx.download(A.TEX_DENOISE_A0, r0, n)
    err = None
    # every held row must hold its OWNER's value (the neighbours', or — under a halo taller than the tiles — tiles further away)
    for k, (ky0, kn) in enumerate(tiling.split_rows(H, world)):
        a, b = max(ky0, r0), min(ky0 + kn, r0 + n)
        if k != rank and b > a and not (got[a - r0:b - r0] == k + 1).all():
            err = "halo rows %s the tile do not hold rank %d's rows" % ("above" if k > rank else "below", k)
    if not (got[y0 - r0:y0 - r0 + rows] == rank + 1).all():
        err = "the tile's own rows were overwritten"
    full = np.zeros((H, W, 3), np.float32)
    full[y0:y0 + rows] = rank + 1
    ctx.upload(A.TEX_COMPOSE_RGB, full, 0, H)
    ctx.allgather_history(A.TEX_COMPOSE_RGB)
    ctx.comm_wait()
    ctx.sync()
    g = ctx.download(A.TEX_COMPOSE_RGB, 0, H)
    for k, (ky0, kn) in enumerate(tiling.split_rows(H, world)):
        if not (g[ky0:ky0 + kn] == k + 1).all():
            err = "gathered rows of rank %d are wrong" % k
    ctx.clear(A.TEX_DENOISE_A0)
    ctx.clear(A.TEX_COMPOSE_RGB)
    ctx.sync()
    return err


def time_case(case, dist, n_steps, n_warmup, dev="cpu", spinup=0, cold=False, first_frame=True):
    """W untimed steps, then exactly K timed steps between barriers + device synchronisation; max over ranks.
    `spinup`: untimed frames BEFORE the W warm-up steps — the device sat idle through ~20 s of dump generation and set-up, and the
    first tens of milliseconds after that run below the sustained clock (same frames, same work; measured with the driver's --steps 20
    --warmup 5: 1.556 ms per step without them against 1.51 sustained)."""
    import torch
    ctx, renderer, fx = case["ctx"], case["renderer"], case["fx"]

    def barrier():
        for name in ("finish_pending", "finish_halo"):  # the exchanges of the last frame are asynchronous: they belong to it
            fin = getattr(renderer, name, None)
            if fin:
                fin()
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(n):
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fx.update(renderer, None)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=ctl_device(dist, dev))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    if first_frame:
        fx.update(renderer, None)  # first frame: uploads the dump (not timed), keepData = 0
    dt_cold = None
    if cold:  # the driver's own W + K protocol straight after the idle set-up phase, BEFORE any spin-up: reported beside the sustained figure
        for _ in range(n_warmup):
            fx.update(renderer, None)
        dt_cold = timed(n_steps)
    for _ in range(spinup):
        fx.update(renderer, None)
    for _ in range(n_warmup):
        fx.update(renderer, None)
    return timed(n_steps), dt_cold


def kernel_times(case, iters):
    """per-kernel durations (hipEvents on the kernels' stream), this rank's tile"""
    ctx, fx = case["ctx"], case["fx"]
    sp, tp = fx.ssgiPass.uniforms, fx.denoiser.temporalReprojectPass.uniforms
    dp, cp = fx.denoiser.denoisePass.uniforms, fx.denoiser.denoiserComposePass.uniforms

    def k3(pass_i):
        dp.inputIsTemporal, dp.writeToB = (1, 0) if pass_i == 0 else (0, 1)
        ctx.poisson_denoise(dp)

    kernels = [("k1_ssgi_march", lambda: ctx.ssgi_march(sp)), ("k2_temporal_reproject", lambda: ctx.temporal_reproject(tp)),
               ("k3_poisson_denoise_pass0", lambda: k3(0)), ("k3_poisson_denoise_pass1", lambda: k3(1)), ("k4_compose", lambda: ctx.compose(cp))]
    kms = {}
    for name, fn in kernels:
        fn()
        ctx.time_begin()
        for _ in range(iters):
            fn()
        kms[name] = ctx.time_end() / iters
    ctx.sync()
    return kms


def kernel_times_in_frame(case, n_frames):
    """Per-draw durations INSIDE the frame loop, as a frame executes them: rfx_profile brackets every draw's launches with hipEvents on the stream
    they run on.  -> ({kernel key: ms per launch}, K1's depth pre-pass ms per frame — its own stream, under the previous frame's later draws)"""
    ctx, renderer, fx = case["ctx"], case["renderer"], case["fx"]
    for _ in range(3):
        fx.update(renderer, None)
    ctx.profile(True)
    for _ in range(n_frames):
        fx.update(renderer, None)
    got = ctx.profile_read()
    ctx.profile(False)
    names = {"k1_ssgi_march": "k1_ssgi_march", "k2_temporal_reproject": "k2_temporal_reproject", "k3_poisson_denoise_pass0": "k3_poisson_denoise_pass0",
             "k3_poisson_denoise_passN": "k3_poisson_denoise_pass1", "k4_compose": "k4_compose"}
    kms = {names[k]: ms / n for k, (ms, n) in got.items() if k in names}
    pre = got.get("k1_prepass")
    return kms, (pre[0] / pre[1] if pre else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 frames = 0.16 s of device time: the fixed costs of the timed region's
    ap.add_argument("--warmup", type=int, default=10)   # barriers and the clock ramp after the idle set-up phase amortise to < 0.5 %
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K, help="frame rows (N = 1) / rows of the 4K frame that is cut into N tiles")
    ap.add_argument("--spinup", type=int, default=-1, help="untimed frames before the W warm-up steps that bring the idle device to its sustained clock (default: 200; 0 on the host simulator)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-copy", action="store_true", help="skip the device-to-device copy measurement (profiling passes: only the path's own kernels)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the frame the CPU port baseline processes (0 = auto)")
    ap.add_argument("--cpu-port", action="store_true", help="(default since round 4) also time the C restatement (OpenMP) as a second, non-GL CPU line")
    ap.add_argument("--no-cpu-port", action="store_true", help="skip the C restatement's CPU line (~10 s)")
    ap.add_argument("--no-cold", action="store_true", help="skip the un-spun-up measurement (ms_per_step_cold)")
    ap.add_argument("--checksum", action="store_true", help="add sha1 of the final whole-frame composed GI to the JSON line (tiled == single check)")
    ap.add_argument("--no-compose-fold", action="store_true", help="(accepted for old command lines: the compose fold left the library in ABI 19)")
    ap.add_argument("--exchange", choices=("c", "torch"), default="c", help="N > 1: exchanges through the C ABI's RCCL entry points (default) or torch.distributed")
    ap.add_argument("--extras-timeout", type=int, default=480, help="seconds the N > 1 extras (weak scaling, configs[4]) may take before the headline line is printed without them")
    ap.add_argument("--configs4-size", default="7680x4320", help="N > 1 extras: frame of the BASELINE configs[4] case (tests shrink it)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: only the headline case (skip the weak-scaling and configs[4] extras)")
    ap.add_argument("--no-kernel-loops", action="store_true", help="skip kernel_ms_solo (back-to-back launches of one entry point): a profile taken over this command then "
                    "holds in-frame launches only (tools/collect_profiles.sh)")
    ap.add_argument("--no-configs4", action="store_true", help="N = 1: skip the BASELINE configs[4] extra (8K, steps 40, six K3 draws: ~1-2 min of host-side dump generation)")
    ap.add_argument("--history-gather", choices=("all", "bounded", "peer"), default="all",
                    help="N > 1: the composed GI as a whole-frame all-gather under the next frame's trace (default); only the column blocks the traced rays read, between trace and "
                         "shade, as packed RCCL messages (bounded: rfx_gather_history_rows) or pulled by the consumer's own kernel through IPC mappings (peer: rfx_peer_gather_history)")
    args = ap.parse_args()
    HISTORY_GATHER[0] = args.history_gather

    if args.gpus > 1 and "RANK" not in os.environ:
        # launched plainly (the way the driver launches N = 1): become the launcher — one rank per GPU of this node under torch.distributed.run
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
        log("bench.py: --gpus %d without RANK in the environment -> %s" % (args.gpus, " ".join(cmd[1:9])))
        raise SystemExit(subprocess.call(cmd))

    import torch
    from rfx_amd import abi as _abi
    # (only an INJECTED library is looked at this early: the in-tree one loads with the first context, after torch has initialised the device)
    hostsim = bool(_abi.injected_library_path() and hasattr(_abi.load_library(), "rfx_hostsim_build"))
    if hostsim:  # tests/hostsim (the kernel sources on the CPU): no device to select or drain, no clock to ramp
        torch.cuda.set_device = torch.cuda.synchronize = lambda *a, **k: None
    if args.spinup < 0:
        args.spinup = 0 if hostsim else 200
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world))
    # functional test hook (tests/test_gpu_parity.py): RFX_BENCH_ONE_GPU=1 puts every rank on device 0 and moves the exchanges over
    # gloo — RCCL refuses two ranks on one device.  Everything but the transport is the multi-GPU path.
    one_gpu = os.environ.get("RFX_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist, use_c = None, False
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_c = args.exchange == "c" and not one_gpu
        if use_c:
            # control plane (barriers, the max-over-ranks reduction, set-up gathers) over gloo; the DATA path is RCCL through the C ABI.
            # If the communicator cannot be created on this node the run falls back to the torch.distributed NCCL (= RCCL) transport.
            dist.init_process_group("gloo", rank=rank, world_size=world)
            box = [None]
            if rank == 0:
                try:
                    box[0] = Context.comm_unique_id()  # probe: is RCCL loadable here?
                except Exception as e:  # noqa: BLE001
                    log("[rank 0] rfx_comm_unique_id failed (%s): falling back to torch.distributed NCCL" % e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                use_c = False
                dist.destroy_process_group()
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        elif one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W1, H1 = args.width, args.height
    # ---- headline: N = 1 -> BASELINE configs[2] (the 4K frame on one GPU); N > 1 -> configs[3]: THE SAME 4K frame cut into N row
    # tiles (strong scaling: 1080 / 540 / 270 rows per GPU at N = 2 / 4 / 8), RCCL halo exchange + composed-GI all-gather
    tiles = [(0, H1)] if world == 1 else tiling.split_rows(H1, world)
    group, fallback_note = None, None
    try:
        case = build_case(world, rank, local_rank, dev, dist, one_gpu, W1, H1, tiles, 20, 5, 1, use_c=use_c)
    except RuntimeError as e:
        if not (use_c and "pre-flight" in str(e)):
            raise
        # every rank saw the same verdict (all_gather_object): continue on the torch.distributed NCCL (= RCCL) transport, and say so
        fallback_note = str(e)[:200]
        log("[rank %d] %s -> torch.distributed NCCL transport" % (rank, fallback_note))
        use_c = False
        group = dist.new_group(backend="nccl")
        case = build_case(world, rank, local_rank, dev, dist, one_gpu, W1, H1, tiles, 20, 5, 1, use_c=False, group=group)
    dt, dt_cold = time_case(case, dist, args.steps, args.warmup, dev, spinup=args.spinup, cold=bool(args.spinup) and not args.no_cold)
    ctx = case["ctx"]
    ms_per_step = dt / args.steps * 1e3
    value = W1 * H1 * args.steps / dt / 1e6  # Mpixels/s, whole job
    viol = ctx.halo_violations()
    compose_sha1 = None
    if args.checksum:  # before the per-kernel timing below re-runs kernels on this rank's tile only
        import hashlib
        renderer = case["renderer"]
        if getattr(renderer, "history_gather", "all") in ("bounded", "peer"):  # the ranks hold only the rows their own rays needed: complete the frame
            renderer.gather_whole_history()
        # .rgb of the whole composed frame: what a tiled run gathers (RFX_TEX_COMPOSE_RGB) and what the next frame's K1 reads
        rgb = ctx.download(abi.TEX_COMPOSE_RGB) if getattr(renderer, "gather_history_rgb", False) else ctx.download(abi.TEX_COMPOSE)[..., :3]
        compose_sha1 = hashlib.sha1(np.ascontiguousarray(rgb).tobytes()).hexdigest()
    # the composed-GI exchange of a row-tiled run: what every rank RECEIVES per frame (bounded gather: rfx_gather_history_rows reports it;
    # the whole-frame all-gather: the other tiles' rows, always), max over ranks
    history = None
    if world > 1:
        got = getattr(case["renderer"], "history_bytes_received", None)
        mode = getattr(case["renderer"], "history_gather", "all")
        mine = float(np.mean(got[-args.steps:])) if (mode in ("bounded", "peer") and got) else float((H1 - case["rows"]) * W1 * 12)
        t = torch.tensor([mine], dtype=torch.float64, device=ctl_device(dist, dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        history = {"mode": mode, "MB_received_per_frame_max_over_ranks": round(float(t.item()) / 1e6, 3),
                   "whole_frame_allgather_MB": round((H1 - min(n for _, n in tiles)) * W1 * 12 / 1e6, 3)}
    kms_solo = {} if args.no_kernel_loops else kernel_times(case, max(5, min(args.steps, 20)))
    kms, prepass_ms = kernel_times_in_frame(case, args.steps)
    rows, halo = case["rows"], case["halo"]
    copy_gbs = stream_copy_gbs(dev) if (rank == 0 and not args.no_stream_copy) else None  # measured here, after the timed region

    extras = {}
    if world == 1 and not args.no_configs4 and not args.checksum and (W1, H1) == (W4K, H4K):
        # BASELINE configs[4] on this one GPU, beside the headline: 8K, steps 40, denoiseIterations 3 (six K3 draws), 16 frames — the configuration
        # north_star assigns to eight GPUs, and the one whose dominant kernel is NOT the 4K line's (five of its nine launches are K3's later pass)
        try:
            W8, H8 = (int(v) for v in args.configs4_size.split("x"))
            c4 = build_case(1, 0, local_rank, dev, None, one_gpu, W8, H8, [(0, H8)], 40, 5, 3)
            d4, _ = time_case(c4, None, 16, 2, dev, spinup=0, cold=False)
            k4ms, pre4 = kernel_times_in_frame(c4, 16)
            px8 = W8 * H8
            per_frame = {"k1_ssgi_march": 1, "k2_temporal_reproject": 1, "k3_poisson_denoise_pass0": 1, "k3_poisson_denoise_pass1": 5, "k4_compose": 1}
            frame_bytes = sum(BYTES_PER_PX[k] * n for k, n in per_frame.items())  # 68 + 80 + 68 + 5 x 52 + 52 = 528 B/px
            sum_ms = sum(k4ms.get(k, 0.0) * n for k, n in per_frame.items())
            extras["configs4_8k"] = {
                "workload": "configs[4]: %dx%d (%.1f Mpixel) steps=40 refineSteps=5 denoiseIterations=3, K1+K2+6xK3+K4 per frame, 16 frames, one GPU" % (W8, H8, px8 / 1e6),
                "ms_per_frame": round(d4 / 16 * 1e3, 4), "value": round(px8 * 16 / d4 / 1e6, 2), "unit": "Mpixels/s",
                "kernel_ms_per_launch": {k: round(v, 4) for k, v in k4ms.items()}, "launches_per_frame": per_frame,
                "k1_prepass_ms": round(pre4, 4) if pre4 is not None else None,
                "roofline_frac_per_kernel": {k: round(BYTES_PER_PX[k] * px8 / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) for k, v in k4ms.items() if k in BYTES_PER_PX and v > 0},
                "dominant_kernel_by_frame_time": max(per_frame, key=lambda k: k4ms.get(k, 0.0) * per_frame[k]),
                "chain": {"algorithmic_bytes_per_px": frame_bytes, "sum_kernel_ms": round(sum_ms, 4),
                          "frac_of_peak": round(frame_bytes * px8 / (sum_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sum_ms else None},
                "halo_violations": c4["ctx"].halo_violations()}
            c4["ctx"].close()
            del c4
        except Exception as e:  # noqa: BLE001  the headline is measured: report it, and what stopped the extra
            extras["configs4_8k"] = {"error": repr(e)[:300]}

    def emit():
        if rank != 0:
            return
        px_tile = W1 * rows
        bpp = BYTES_PER_PX
        # K1's 68 B/px include the depth plane, which its PRE-PASS reads (k1_prepare + k1_pack_cells, own stream): its roofline duration is both
        dur = dict(kms)
        if prepass_ms is not None and "k1_ssgi_march" in dur:
            dur["k1_ssgi_march"] += prepass_ms
        dom = max(dur, key=dur.get)
        achieved = bpp[dom] * px_tile / (dur[dom] * 1e-3) / 1e9
        chain_bytes = sum(bpp[k] for k in kms)
        chain_ms = sum(kms.values())
        k4_ms = kms.get("k4_compose", 0.0)
        ns_ms = chain_ms - k4_ms if "k4_compose" in kms else None  # the north-star quantity: K1 + K2 + 2 x K3 (268 B/px)
        ns_bytes = chain_bytes - BYTES_PER_PX["k4_compose"]
        prof = profile_meta()
        workload = ("configs[2]: %dx%d (%.2f Mpixel) on one GPU" % (W1, H1, W1 * H1 / 1e6) if world == 1 else
                    "configs[3]: the %dx%d frame cut into %d row tiles of %d rows" % (W1, H1, world, rows))
        out = {
            "metric": "Mpixels/s SSGI+denoise @4K steps=20; achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_frames": args.spinup,
            "ms_per_step": round(ms_per_step, 4),
            # the same W + K protocol run once BEFORE the spin-up frames, straight after the ~20 s of host-side set-up (device below its sustained clock)
            "ms_per_step_cold": round(dt_cold / args.steps * 1e3, 4) if dt_cold else None,
            # the job is the same 4K frame at every N (N > 1 cuts it into N row tiles): total work fixed
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload + ", steps=20 refineSteps=5 denoiseIterations=1, K1+K2+2xK3+K4 per step",
                       "frame": "%dx%d" % (W1, H1), "tile_rows": rows, "halo_rows": halo, "direct_light": True, "half_store": "rtz", "uv_model": "reference_gl",
                       "parallelism": "row-tiles x%d, RCCL halo send/recv after K2 and every K3 pass + composed GI (see history_exchange); exchange: %s" % (
                           world, case_exchange(use_c, one_gpu, args)) if world > 1 else "single GPU"},
            # per-draw durations INSIDE the frame loop (rfx_profile: events around every draw on its stream, K more frames after the timed region)
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "k1_prepass_ms": round(prepass_ms, 4) if prepass_ms is not None else None,
            # ... and every kernel timed on its own (back-to-back launches of the same entry point), as rounds 1-4 reported them
            "kernel_ms_solo": {k: round(v, 4) for k, v in kms_solo.items()},
            "chain": {"algorithmic_bytes_per_px": chain_bytes, "sum_kernel_ms": round(chain_ms, 4),
                      "achieved_GBs": round(chain_bytes * px_tile / (chain_ms * 1e-3) / 1e9, 1),
                      "frac_of_peak": round(chain_bytes * px_tile / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "north_star_chain": ({"kernels": "K1+K2+2xK3", "algorithmic_bytes_per_px": ns_bytes, "sum_kernel_ms": round(ns_ms, 4),
                                  "achieved_GBs": round(ns_bytes * px_tile / (ns_ms * 1e-3) / 1e9, 1),
                                  "frac_of_peak": round(ns_bytes * px_tile / (ns_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "target_frac": 0.70} if ns_ms else None),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "stream_copy_GBs": copy_gbs,
                         "frac_of_stream_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
                         "traffic": pmc_traffic(dom)[0] if (W1, rows) == (W4K, H4K) else None,
                         "traffic_calibration": pmc_traffic(dom)[1] if (W1, rows) == (W4K, H4K) else None,
                         "traffic_note": TRAFFIC_NOTE % (PROFILE_DIR, prof.get("git_commit", "?")),
                         "algorithmic_bytes_per_launch": bpp[dom] * px_tile, "avg_launch_ms": round(dur[dom], 4),
                         "avg_launch_ms_note": ("in-frame duration, rfx_profile" + (
                             ": march %.4f + depth pre-pass %.4f ms (the pre-pass reads the depth plane that is part of K1's 68 B/px; it runs on its own "
                             "stream under the previous frame's later draws)" % (kms["k1_ssgi_march"], prepass_ms) if dom == "k1_ssgi_march" and prepass_ms is not None else ""))},
            "halo_violations": viol,
        }
        # what binds: instruction issue.  Per kernel, the ABSOLUTE time its measured dynamic instruction mix costs to issue
        # (tools/issue_model.py: class counts from rocprofv3 --pmc x the per-class issue cost measured on this part) next to the time the kernel took
        # in the same collection.  Only quoted for the frame the collection was taken on (the whole 4K frame on one GPU).
        if (W1, rows) == (W4K, H4K):
            im = issue_model()
            if im:
                out["issue_model"] = dict(im, note="per kernel: predicted_issue_ms = sum(class count x measured issue cycles) x waves per SIMD / clock, the classes the SQ "
                                                   "counters do not name priced at the kernel's own ISA mix (rate_int / rate_other, tools/isa_mix.py); measured_ms = the kernel's "
                                                   "average duration in the same rocprofv3 collection; lds_busy_ms = SQ_LDS_IDX_ACTIVE / 256 CUs / clock — "
                                                   "%s/{pmc_sq_l2.csv, kernel_stats.csv, isa_other_mix.json, issue_model.txt} (collected at git %s).  A linear model: interleaved "
                                                   "instruction streams issue up to ~25 %% below the sum of their isolated costs (valu_rates2.txt, k_mix_*)" % (
                                                       PROFILE_DIR, prof.get("git_commit", "?")))
        out["kernel_ms_note"] = ("kernel_ms: per-draw durations inside the frame loop (hipEvents around every draw's launches, rfx_profile) — they sum to ms_per_step up to the "
                                 "events' own few microseconds; K1's depth pre-pass (k1_prepass_ms) runs on its own stream under the previous frame's K2-K4 and is not "
                                 "in that sum.  kernel_ms_solo: every kernel timed on its own, back-to-back launches of one entry point (K1 then serialises with its own "
                                 "pre-pass: longer than in a frame; K2-K4 run without a pre-pass beside them: shorter)")
        if world > 1:
            out["config"]["exchange_verified"] = bool(use_c)  # the C-ABI exchanges passed their pre-flight pattern check on every rank
            out["config"]["history_exchange"] = history
            if fallback_note:
                out["config"]["exchange_fallback"] = fallback_note
        out.update(extras)
        if args.checksum:
            out["compose_sha1"], out["frame_rows"] = compose_sha1, H1
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(case["frame"], case["fx"], W1, H1, args.cpu_sample_rows, not args.no_cpu_port)
        print(json.dumps(out), flush=True)

    if world > 1 and not args.no_extras:
        ctx.close()
        case = None
        # the headline is measured: a collective that hangs in the extras (their first run on a real multi-GPU node is the driver's) must
        # not cost the line.  After --extras-timeout seconds rank 0 prints what it has and every rank leaves.
        import threading

        def give_up():
            extras["extras_error"] = "timed out after %d s (a hung collective?); done so far: %s" % (args.extras_timeout, sorted(extras))
            emit()
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(args.extras_timeout + (0 if rank == 0 else 5), give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            # (a) weak scaling, the round-1 headline: the frame grows with N at constant aspect, every rank owns 8.29 Mpixel
            Ww = int(round(W1 * world ** 0.5 / 64.0)) * 64
            Ht = int(W1 * H1 / Ww) & ~1
            wtiles = [(r * Ht, Ht) for r in range(world)]
            if wtiles == tiling.split_rows(Ht * world, world):
                wcase = build_case(world, rank, local_rank, dev, dist, one_gpu, Ww, Ht * world, wtiles, 20, 5, 1, use_c=use_c, group=group)
                wdt, _ = time_case(wcase, dist, args.steps, args.warmup, dev)
                extras["weak_scaling"] = {"frame": "%dx%d" % (Ww, Ht * world), "tile_rows": Ht, "halo_rows": wcase["halo"], "ms_per_step": round(wdt / args.steps * 1e3, 4),
                                          "value": round(Ww * Ht * world * args.steps / wdt / 1e6, 2), "unit": "Mpixels/s",
                                          "halo_violations": wcase["ctx"].halo_violations()}
                wcase["ctx"].close()
            # (b) BASELINE configs[4]: 8K, steps 40, denoiseIterations 3, row-tiled (the 16-frame sequence re-renders one dumped frame)
            W8, H8 = (int(v) for v in args.configs4_size.split("x"))
            c4 = build_case(world, rank, local_rank, dev, dist, one_gpu, W8, H8, tiling.split_rows(H8, world), 40, 5, 3, use_c=use_c, group=group)
            n4 = max(4, min(args.steps, 16))
            d4, _ = time_case(c4, dist, n4, 2, dev)
            extras["configs4_8k"] = {"frame": "%dx%d" % (W8, H8), "steps": 40, "refineSteps": 5, "denoiseIterations": 3, "frames_timed": n4, "halo_rows": c4["halo"],
                                     "ms_per_frame": round(d4 / n4 * 1e3, 4), "value": round(W8 * H8 * n4 / d4 / 1e6, 2), "unit": "Mpixels/s",
                                     "halo_violations": c4["ctx"].halo_violations()}
            c4["ctx"].close()
        except Exception as e:  # noqa: BLE001  the headline case above is already measured: report it, and what stopped the extras
            extras["extras_error"] = repr(e)[:300]
        watchdog.cancel()

    emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if case is not None:
        ctx.close()


def case_exchange(use_c, one_gpu, args):
    if use_c:
        return "C ABI rfx_halo_exchange / rfx_allgather_history (RCCL on the context's exchange stream)"
    return "torch.distributed (%s)" % ("gloo, one-GPU functional mode" if one_gpu else "nccl = RCCL")


def cpu_baseline_llvmpipe(frame, W, H):
    """kind "reference": the reference's OWN fragment shaders (assembled by `make -C oracle ref` into
    oracle/_ref/shaders/, build products) executed by Mesa llvmpipe on this box's host cores through
    oracle/glref — K1+K2+2xK3+K4 on the same 4K frame, per-draw glFinish-fenced wall time, after two
    warm-up frames (llvmpipe JIT-compiles on the first draw and re-specialises on the second)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("LP_NUM_THREADS", str(min(cores, 32)))  # llvmpipe caps its rasteriser threads (LP_MAX_THREADS)
    import chain
    from rfx_amd.context import load_blue_noise_table
    c = chain.GLRefChain(W, H, load_blue_noise_table(), steps=20, refineSteps=5, denoiseIterations=1)
    c.upload_frame(frame)

    def one(i):
        c.ssgi(frame.camera, 100 + i)
        c.temporal(frame.camera)
        c.denoise(frame.camera, [200 + 2 * i, 201 + 2 * i])
        c.compose(frame.camera)
        return c.ms["ssgi"] + c.ms["temporal"] + sum(c.ms["denoise"]) + c.ms["compose"]

    one(0)
    one(1)
    ms, n, t0 = [], 0, time.perf_counter()
    while n < 5 and (time.perf_counter() - t0 < 20.0 or n == 0):
        ms.append(one(2 + n))
        n += 1
    med = sorted(ms)[len(ms) // 2]
    return {"value": round(W * H / med / 1e3, 3), "unit": "Mpixels/s", "cores": int(os.environ["LP_NUM_THREADS"]), "kind": "reference",
            "sample": "median of %d frames of the reference GLSL (ssgi/temporal/2x denoise/compose) on llvmpipe over the same %dx%d frame, "
                      "%.0f ms per frame; %s; box has %d cores, LP_NUM_THREADS=%s" % (n, W, H, med, chain.GL.info(), cores, os.environ["LP_NUM_THREADS"])}


def cpu_baseline(frame, fx, W, H, sample_rows, with_port=False):
    """The reference GLSL on llvmpipe (kind "reference"): the reference's own code on this box's host cores.  The C restatement
    (oracle/rfx_oracle.c, kind "port": scalar, OpenMP over rows) is slower than llvmpipe's vectorised JIT even on 8x the threads
    (2.0-2.9 vs 10-12 Mpix/s), so it is only the fallback when the GL harness is unavailable, or an extra line on request."""
    try:
        ref = cpu_baseline_llvmpipe(frame, W, H)
        if with_port:
            port = cpu_baseline_port(frame, fx, W, H, sample_rows)
            ref["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}
        return ref
    except Exception as e:  # no swrast_dri.so / no prebuilt shaders on this box
        port = cpu_baseline_port(frame, fx, W, H, sample_rows)
        port["llvmpipe_unavailable"] = repr(e)[:200]
        return port


def cpu_baseline_port(frame, fx, W, H, sample_rows):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rfx_oracle as O
    from rfx_amd.context import load_blue_noise_table
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    blue = load_blue_noise_table()
    rows = sample_rows or min(H, 540)  # a quarter of the 4K frame through its middle: ~1 s per pass of the chain on this box's cores
    y0 = max(0, (H - rows) // 2) & ~1
    band = (y0, y0 + rows)
    z16 = lambda: np.zeros((H, W, 4), np.uint16)  # noqa: E731
    z32 = lambda: np.zeros((H, W, 4), np.float32)  # noqa: E731
    sp, tp = fx.ssgiPass.uniforms, fx.denoiser.temporalReprojectPass.uniforms
    dp, cp = fx.denoiser.denoisePass.uniforms, fx.denoiser.denoiserComposePass.uniforms
    hist, ssgi, T0, T1, A0, A1, B0, B1, comp = z32(), np.zeros((H, W, 4), np.uint32), z32(), z32(), z16(), z16(), z16(), z16(), z32()

    def chain():
        O.ssgi(frame.depth, frame.gbuffer, frame.direct, hist, blue, sp, out=ssgi, rows=band)
        O.temporal(ssgi, frame.velocity, B0, B1, tp, T0, T1, rows=band)
        dp.inputIsTemporal, dp.writeToB = 1, 0
        O.denoise(frame.depth, frame.gbuffer, T0, T1, blue, dp, A0, A1, rows=band)
        dp.inputIsTemporal, dp.writeToB = 0, 1
        O.denoise(frame.depth, frame.gbuffer, A0, A1, blue, dp, B0, B1, rows=band)
        O.compose(frame.depth, frame.gbuffer, B0, B1, cp, out=comp, rows=band)

    chain()  # warm (page faults, OpenMP team)
    t0 = time.perf_counter()
    n = 0
    while True:
        chain()
        n += 1
        dt = time.perf_counter() - t0
        if dt > 8.0 or n >= 20:
            break
    return {"value": round(W * rows * n / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d x (K1+K2+2xK3+K4) over rows [%d,%d) of the same %dx%d frame, %.1f s of wall time, OMP threads=%s" % (
                n, band[0], band[1], W, H, dt, os.environ.get("OMP_NUM_THREADS"))}


if __name__ == "__main__":
    main()
