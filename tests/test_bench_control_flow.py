"""CPU: bench.py's control flow with the device work mocked — the one JSON line for N = 1, for N = 2 with its extras, and the watchdog that
prints the headline when a collective in the extras hangs (their first run on a real multi-GPU node is the driver's)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MOCK = textwrap.dedent('''
    import sys, os, json, time
    sys.path.insert(0, %r)
    import bench, torch
    import torch.distributed as dist
    torch.cuda.set_device = lambda *_: None
    dist.init_process_group = lambda *a, **k: None
    dist.barrier = lambda *a, **k: None
    dist.destroy_process_group = lambda *a, **k: None
    dist.all_reduce = lambda *a, **k: None
    dist.get_backend = lambda *a, **k: "gloo"
    class FakeCtx:
        def halo_violations(self): return 0
        def close(self): pass
    kms = {"k1_ssgi_march": 0.61, "k2_temporal_reproject": 0.45, "k3_poisson_denoise_pass0": 0.24, "k3_poisson_denoise_pass1": 0.38, "k4_compose": 0.11}
    bench.build_case = lambda world, rank, lr, dev, d, one, W, H, tiles, *a, **k: dict(ctx=FakeCtx(), rows=tiles[rank][1], halo=0 if world == 1 else 12,
                                                                                       frame=None, fx=None, renderer=None)
    calls = [0]
    def time_case(*a, **k):
        calls[0] += 1
        if os.environ.get("HANG_AT") and calls[0] >= int(os.environ["HANG_AT"]):
            time.sleep(3600)
        return 0.0366, (0.0380 if k.get("cold") else None)
    bench.time_case = time_case
    bench.kernel_times = lambda *a, **k: dict(kms)
    bench.kernel_times_in_frame = lambda *a, **k: (dict(kms, k1_ssgi_march=0.55), 0.11)
    bench.cpu_baseline = lambda *a, **k: {"value": 10.4, "unit": "Mpixels/s", "cores": 32, "kind": "reference", "sample": "mock"}
    bench.main()
''') % ROOT


def _run(argv, env=None, timeout=120):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RFX_BENCH_ONE_GPU", "HANG_AT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, "-c", "import sys; sys.argv = %r\n%s" % (["bench.py"] + argv, MOCK)], capture_output=True, text=True, timeout=timeout, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    j = _run(["--steps", "20", "--warmup", "5"])
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and j["vs_baseline"] is None and j["unit"] == "Mpixels/s"  # the same 4K frame at every N
    assert abs(j["ms_per_step_cold"] - 1.9) < 1e-6  # the W + K protocol before the spin-up, reported beside the sustained figure
    assert abs(j["value"] - 3840 * 2160 * 20 / 0.0366 / 1e6) < 0.01 and abs(j["ms_per_step"] - 1.83) < 1e-6
    assert j["roofline"]["bound"] == "hbm" and j["roofline"]["kernel"] == "k1_ssgi_march" and 0 < j["roofline"]["frac"] < 1
    # the dominant kernel's duration is its in-frame launch PLUS its depth pre-pass (whose input is part of K1's algorithmic bytes)
    assert abs(j["roofline"]["avg_launch_ms"] - 0.66) < 1e-6 and abs(j["kernel_ms"]["k1_ssgi_march"] - 0.55) < 1e-6 and abs(j["k1_prepass_ms"] - 0.11) < 1e-6
    assert abs(j["roofline"]["achieved"] - 68 * 3840 * 2160 / 0.66e-3 / 1e9) < 0.1
    assert "compose_fold" not in j["config"]  # one launch per draw is the only path since ABI 19
    assert j["cpu_baseline"]["kind"] == "reference" and j["config"]["workload"].startswith("configs[2]")


def test_bench_line_row_tiled_with_extras_and_watchdog():
    env = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", RFX_BENCH_ONE_GPU="1")
    j = _run(["--gpus", "2", "--steps", "4", "--warmup", "1"], env)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["tile_rows"] == 1080 and j["config"]["workload"].startswith("configs[3]")
    assert "weak_scaling" in j and "configs4_8k" in j and "cpu_baseline" not in j
    assert j["config"]["history_exchange"]["mode"] == "all" and j["config"]["history_exchange"]["whole_frame_allgather_MB"] == round(1080 * 3840 * 12 / 1e6, 3)
    # the second extra hangs: after --extras-timeout the headline is printed with what was done, and the process leaves
    j = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--extras-timeout", "2"], dict(env, HANG_AT="3"))
    assert j["n_gpus"] == 2 and "weak_scaling" in j and "configs4_8k" not in j and "timed out" in j["extras_error"]


def test_bench_row_tiled_for_real_under_the_host_simulator(tmp_path):
    """bench.py --gpus 2 launched plainly (it re-launches itself under torch.distributed.run) and --gpus 3 as the contract launches it
    (torch.distributed.run, one process per rank), nothing mocked: the dump bands, the halo
    from the velocity bound, the depth all-reduce, the C-ABI communicator with its pre-flight pattern check (verify_exchange), the timed
    steps through CommTiledRenderer, the per-kernel timing, the JSON line — on a small frame, with tests/hostsim under the C ABI (the kernel
    sources on the CPU, a socket stand-in for the RCCL slice).  N = 1 the same way."""
    import shutil
    import socket
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or shutil.which("make") is None:
        import pytest
        pytest.skip("no host clang++ / make")
    sim = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", sim])
    from conftest import hostsim_child_env
    env = dict(os.environ, **hostsim_child_env(sim))
    env.update(LD_LIBRARY_PATH=os.path.join(sim, "_build", "fakerccl") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "RFX_BENCH_ONE_GPU"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--width", "192", "--height", "128", "--no-cpu-baseline", "--checksum"]
    p1 = subprocess.run([sys.executable, "bench.py"] + common, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p1.returncode == 0, p1.stderr[-2000:]
    one = json.loads([ln for ln in p1.stdout.splitlines() if ln.startswith("{")][-1])
    for n in (2, 3):  # 3: ragged tiles (128 rows over three ranks), a middle rank with two neighbours
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        # n = 2: launched PLAINLY, `python bench.py --gpus 2`, the way the driver launches N = 1 — bench.py becomes the launcher itself;
        # n = 3: under torch.distributed.run, the way the contract spells the N > 1 launch
        launch = ([sys.executable, "bench.py", "--gpus", "2"] if n == 2 else
                  [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
                   "bench.py", "--gpus", str(n)])
        p2 = subprocess.run(launch + (["--no-extras", "--history-gather", "bounded"] if n == 3 else ["--configs4-size", "160x96"]) + common,
                            cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert p2.returncode == 0, (p2.stdout + p2.stderr)[-3000:]
        many = json.loads([ln for ln in p2.stdout.splitlines() if ln.startswith("{")][-1])
        assert many["n_gpus"] == n and many["config"]["tile_rows"] in ((64,) if n == 2 else (42, 43)) and many["halo_violations"] == 0 and many["scaling"] == "strong"
        assert many["config"]["exchange_verified"] is True and "exchange_fallback" not in many["config"]  # the C-ABI exchanges passed their pre-flight check
        assert many["compose_sha1"] == one["compose_sha1"]  # the tiled run's composed frame == the single-context run's, bit for bit
        hx = many["config"]["history_exchange"]  # the composed GI travelled through the bounded gather (rfx_gather_history_rows)
        assert hx["mode"] == ("bounded" if n == 3 else "all") and 0 < hx["MB_received_per_frame_max_over_ranks"] <= hx["whole_frame_allgather_MB"], hx
        if n == 2:  # the extras ran too: the weak-scaling frame and the configs[4] options (steps 40, six K3 passes), both row-tiled
            assert "extras_error" not in many, many.get("extras_error")
            assert many["weak_scaling"]["halo_violations"] == 0 and many["configs4_8k"]["halo_violations"] == 0 and many["configs4_8k"]["frame"] == "160x96"
