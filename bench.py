#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X: Mpixels/s of the SSGI chain at 4K.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one frame through SSGIEffect.update(): K1 SSGI march (steps 20 / refineSteps 5) ->
K2 temporal reprojection -> 2 x K3 Poisson denoise (denoiseIterations 1) -> K4 compose, over a
3840x2160 synthetic G-buffer dump (seed 1234) that is ALREADY RESIDENT in HBM when the timed
region starts.  N > 1: weak scaling — the frame keeps its 16:9 aspect and grows to N x 8.29 Mpixel
(e.g. 7680x4320 for N = 4), is cut into N row tiles (one per GPU, 8.29 Mpixel each) which exchange
halo rows with their neighbours over RCCL after K2 and after every K3 pass, plus an all-gather of
the composed GI that runs asynchronously under the next frame's depth pre-pass + ray march (K1 is
split into rfx_ssgi_trace / rfx_ssgi_shade for that; rfx_amd/tiling.py).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (largest share of the
step), measured live with hipEvents on the stream the kernels run on; `cpu_baseline` is the
oracle (the C restatement, OpenMP) timed on this box's host cores on the same frame.
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
from rfx_amd.scene import AnalyticScene  # noqa: E402

W4K, H4K = 3840, 2160
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# algorithmic bytes per pixel per launch, reference texel formats (SURVEY.md §8d / DESIGN.md)
BYTES_PER_PX = {"k1_ssgi_march": 68, "k2_temporal_reproject": 80, "k3_poisson_denoise_pass0": 68, "k3_poisson_denoise_pass1": 52, "k4_compose": 52}


# rocprofv3 kernel-name fragments of bench.py's kernel keys (profiles/*/pmc_hbm.csv)
PMC_KERNEL = {"k1_ssgi_march": "k1_ssgi_march", "k2_temporal_reproject": "k2_temporal_reproject", "k3_poisson_denoise_pass0": "k3_tiled<true",
              "k3_poisson_denoise_pass1": "k3_tiled<false", "k4_compose": "k4_compose"}


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of a kernel from the committed PMC summary of this same command
    (profiles/r01_final/pmc_hbm.csv: separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, values in KiB).
    gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of wide coalesced reads -> x2.
    Returns None when the summary is missing or was taken at another frame size."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_final", "pmc_hbm.csv")
    if not os.path.exists(path):
        return None
    vals = {}
    for r in csv.DictReader(open(path)):
        if PMC_KERNEL[kernel_key] in r["kernel"]:
            vals[r["counter"]] = float(r["mean_value_KB"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)


# measured on MI355X by tools/microbench/valu_rates.hip (profiles/r01_final/valu_rates.txt): a wave64 fp32 VALU instruction issues every
# 1.026 ns per SIMD at 8 waves/SIMD (2 cycles at the ~1.95 GHz the part sustains), a transcendental (v_exp/v_log/v_rcp/v_sqrt) every 4.1 ns
VALU_NS, TRANS_NS, N_SIMD = 1.0 / 0.975, 4.1, 256 * 4


def valu_floor_ms(kernel_key, pixels):
    """Issue-bound time of a kernel: its VALU instruction counts per wavefront (committed PMC summary, profiles/r01_final/pmc_sq_l2.csv:
    SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32, SQ_WAVES — a property of the code, not of the run) priced at the measured issue rates."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_final", "pmc_sq_l2.csv")
    if not os.path.exists(path):
        return None
    c = {}
    for r in csv.DictReader(open(path)):
        if PMC_KERNEL[kernel_key] in r["kernel"]:
            c[r["counter"]] = float(r["mean_value"])
    if not all(k in c for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAVES")):
        return None
    valu, trans = c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_WAVES"]
    waves_per_simd = pixels / 64.0 / N_SIMD
    return ((valu - trans) * VALU_NS + trans * TRANS_NS) * waves_per_simd * 1e-6


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=W4K)
    ap.add_argument("--height", type=int, default=H4K, help="rows per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=0, help="rows of the frame the CPU baseline processes (0 = auto)")
    ap.add_argument("--checksum", action="store_true", help="add sha1 of the final whole-frame composed GI to the JSON line (tiled == single check)")
    args = ap.parse_args()

    import torch
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
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # weak scaling: the frame grows with the number of GPUs at constant aspect, so that the per-pixel work (tap
    # footprints, ray lengths in pixels) stays what it is on one GPU; every rank owns W*Ht = const pixels
    W1, H1 = args.width, args.height
    if world == 1:
        W, H, Ht = W1, H1, H1
    else:
        W = int(round(W1 * world ** 0.5 / 64.0)) * 64
        Ht = int(W1 * H1 / W) & ~1
        H = Ht * world
    tiles = [(r * Ht, Ht) for r in range(world)]
    y0, rows = tiles[rank]

    # ---- synthetic dump: every rank ray-casts the band it holds (+ halo); depth is gathered whole
    t0 = time.time()
    scene_gen = AnalyticScene(1234)
    opts = dict(width=W, height=H, steps=20, refineSteps=5, denoiseIterations=1)
    # frame 1 of the orbit: non-zero velocity (camera moved 0.5 deg since frame 0).  The tile is dumped first, the
    # velocity bound over ALL tiles fixes the halo width, then the halo rows are dumped and attached.
    tile = scene_gen.render(W, rows, 1, row0=y0, rows=rows, frame_height=H)
    vmax = float(np.abs(tile.velocity[..., 1].view(np.float32)).max())
    if dist is not None:  # every rank must use the SAME halo: the neighbours' send/recv sizes have to match
        t = torch.tensor([vmax], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vmax = float(t.item())
    halo = 0 if world == 1 else tiling.required_halo(3.0, vmax, H, W)
    b0, b1 = max(0, y0 - halo), min(H, y0 + rows + halo)
    parts = [tile]
    if b0 < y0:
        parts.insert(0, scene_gen.render(W, y0 - b0, 1, row0=b0, rows=y0 - b0, frame_height=H))
    if b1 > y0 + rows:
        parts.append(scene_gen.render(W, b1 - y0 - rows, 1, row0=y0 + rows, rows=b1 - y0 - rows, frame_height=H))
    band = types.SimpleNamespace(camera=tile.camera, **{k: np.concatenate([getattr(q, k) for q in parts], axis=0)
                                                         for k in ("depth", "gbuffer", "velocity", "direct")})
    log("[rank %d] dump band rows [%d,%d) of %dx%d generated in %.1fs (halo %d)" % (rank, b0, b1, W, H, time.time() - t0, halo))

    ctx = Context(W, H, device=local_rank, tile_y0=y0, tile_rows=rows, halo_rows=halo)
    # kernels, RCCL ops and torch's copies share ONE created stream: torch's legacy default stream has handle 0, which
    # rfx_set_stream reads as "use the context's own stream" — the collectives would then not be ordered against the kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    # gloo stages device tensors through the host on its own schedule: drain the stream around every exchange there
    ctx.uses_torch_stream = not one_gpu
    renderer = ctx
    depth_full = band.depth
    if world > 1:
        tensors = tiling.bind_torch_buffers(ctx, dev)
        renderer = tiling.TiledRenderer(ctx, tensors, rank, world)
        mine = torch.from_numpy(np.ascontiguousarray(band.depth[y0 - b0:y0 - b0 + rows])).to(dev)
        full = torch.empty((H, W), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(full, mine)
        depth_full = full.cpu().numpy()
    # static: the same dump every step, uploaded once before the timed region (the metric is quoted with inputs resident in HBM)
    frame = types.SimpleNamespace(depth=depth_full, gbuffer=band.gbuffer, velocity=band.velocity, direct=band.direct, camera=band.camera, static=True)
    scene = types.SimpleNamespace(frame=frame)
    cam = band.camera
    fx = SSGIEffect(None, scene, cam, opts, seeds=dict(ssgi=1, denoise=2), half_store_rtz=True)

    def step():
        fx.update(renderer, None)

    def barrier():
        for name in ("finish_pending", "finish_halo"):  # the exchanges of the last frame are asynchronous (tiling.py): they belong to it
            fin = getattr(renderer, name, None)
            if fin:
                fin()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    step()  # first frame: uploads the dump (not timed), keepData = 0
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = W * H * args.steps / dt / 1e6  # Mpixels/s, whole job
    viol = ctx.halo_violations()
    compose_sha1 = None
    if args.checksum:  # before the per-kernel timing below re-runs kernels on this rank's tile only
        import hashlib
        # .rgb of the whole composed frame: what a tiled run gathers (RFX_TEX_COMPOSE_RGB) and what the next frame's K1 reads
        rgb = ctx.download(abi.TEX_COMPOSE_RGB) if getattr(renderer, "gather_history_rgb", False) else ctx.download(abi.TEX_COMPOSE)[..., :3]
        compose_sha1 = hashlib.sha1(np.ascontiguousarray(rgb).tobytes()).hexdigest()

    # ---- per-kernel durations (hipEvents on the kernels' stream), this rank's tile
    sp, tp = fx.ssgiPass.uniforms, fx.denoiser.temporalReprojectPass.uniforms
    dp, cp = fx.denoiser.denoisePass.uniforms, fx.denoiser.denoiserComposePass.uniforms

    def k3(pass_i):
        dp.inputIsTemporal, dp.writeToB = (1, 0) if pass_i == 0 else (0, 1)
        ctx.poisson_denoise(dp)

    kernels = [("k1_ssgi_march", lambda: ctx.ssgi_march(sp)), ("k2_temporal_reproject", lambda: ctx.temporal_reproject(tp)),
               ("k3_poisson_denoise_pass0", lambda: k3(0)), ("k3_poisson_denoise_pass1", lambda: k3(1)), ("k4_compose", lambda: ctx.compose(cp))]
    iters = max(5, min(args.steps, 20))
    kms = {}
    for name, fn in kernels:
        fn()
        ctx.time_begin()
        for _ in range(iters):
            fn()
        kms[name] = ctx.time_end() / iters
    barrier()

    if rank == 0:
        px_tile = W * rows
        dom = max(kms, key=kms.get)
        achieved = BYTES_PER_PX[dom] * px_tile / (kms[dom] * 1e-3) / 1e9
        chain_bytes = sum(BYTES_PER_PX[k] for k in kms)
        chain_ms = sum(kms.values())
        out = {
            "metric": "Mpixels/s SSGI+denoise @4K steps=20; achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: %dx%d (%.2f Mpixel) per GPU, steps=20 refineSteps=5 denoiseIterations=1, K1+K2+2xK3+K4 per step" % (W, Ht, W * Ht / 1e6),
                       "frame": "%dx%d" % (W, H), "tile_rows": rows, "halo_rows": halo, "direct_light": True, "half_store": "rtz",
                       "parallelism": "row-tiles x%d, RCCL halo send/recv + compose all-gather (async, overlapped with the next frame's K1 trace)" % world if world > 1 else "single GPU"},
            "kernel_ms": {k: round(v, 4) for k, v in kms.items()},
            "chain": {"algorithmic_bytes_per_px": chain_bytes, "sum_kernel_ms": round(chain_ms, 4),
                      "achieved_GBs": round(chain_bytes * px_tile / (chain_ms * 1e-3) / 1e9, 1),
                      "frac_of_peak": round(chain_bytes * px_tile / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": pmc_traffic(dom) if (W, rows) == (W4K, H4K) else None,
                         "traffic_note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/r01_final/pmc_hbm.csv (rocprofv3 --pmc, same command, 4K)",
                         "algorithmic_bytes_per_launch": BYTES_PER_PX[dom] * px_tile, "avg_launch_ms": round(kms[dom], 4)},
            "halo_violations": viol,
        }
        # the bound that actually binds: VALU issue — reported next to the HBM roofline the metric asks for
        floors = {k: valu_floor_ms(k, px_tile) for k in kms}
        if all(v is not None for v in floors.values()):
            out["valu_issue_roofline"] = {"floor_ms": {k: round(v, 4) for k, v in floors.items()}, "sum_floor_ms": round(sum(floors.values()), 4),
                                          "frac": round(sum(floors.values()) / chain_ms, 4),
                                          "note": "VALU + transcendental instructions per wave (profiles/r01_final PMC) at the issue rates measured by tools/microbench/valu_rates.hip"}
        if args.checksum:
            out["compose_sha1"], out["frame_rows"] = compose_sha1, H
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frame, fx, W, H, args.cpu_sample_rows)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


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


def cpu_baseline(frame, fx, W, H, sample_rows):
    """Preferred: the reference GLSL on llvmpipe (kind "reference").  Fallback / second line: the oracle
    (oracle/rfx_oracle.c, kind "port": scalar C restatement, OpenMP over rows) on the host cores of
    this box: the same chain on a bounded band of the same 4K frame."""
    port = cpu_baseline_port(frame, fx, W, H, sample_rows)
    try:
        ref = cpu_baseline_llvmpipe(frame, W, H)
        ref["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}
        return ref
    except Exception as e:  # no swrast_dri.so / no prebuilt shaders on this box
        port["llvmpipe_unavailable"] = repr(e)[:200]
        return port


def cpu_baseline_port(frame, fx, W, H, sample_rows):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rfx_oracle as O
    from rfx_amd.context import load_blue_noise_table
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    blue = load_blue_noise_table()
    rows = sample_rows or H
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
        if dt > 10.0 or n >= 20:
            break
    return {"value": round(W * rows * n / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "%d x (K1+K2+2xK3+K4) over rows [%d,%d) of the same %dx%d frame, %.1f s of wall time, OMP threads=%s" % (
                n, band[0], band[1], W, H, dt, os.environ.get("OMP_NUM_THREADS"))}


if __name__ == "__main__":
    main()
