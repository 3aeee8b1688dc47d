"""CPU: the kernels' LOGIC, executed.  tests/hostsim compiles the product's own kernel sources (realism-effects_amd/csrc/*.hip) for x86
against a stand-in <hip/hip_runtime.h> and runs them thread by thread; a subset of the `-m gpu` tests is run against that library here
(the whole `-m gpu` suite runs the same way with `pytest tests -m gpu --hostsim`, a few minutes).  What this proves: indexing, tiles and
aprons, launch shapes, the C ABI's state handling, the host drivers on top.  What it does not: anything about the device's bits (the
hardware transcendentals are libm here) or speed.  The library under test is test infrastructure: nothing in the product can load it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("make") is None, reason="no host clang++ / make")
def test_gpu_tests_subset_passes_under_host_simulation():
    sel = ("cube or chain_stagewise_vs_oracle or row_tiled_chain_is_bit_identical or denoise_variants or traa_end_to_end or ssgi_trace_plus_shade "
           "or row_windowed_draws or per_draw_profile or peer_history_gather_between or reference_vuv_model_on_device or resolution_scale or node_host_drives or streamed_dumps or tiled_kernels or node_row_tiled_run or single_rank_ring or node_cube_environment")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_zz_gpu_cube_environment.py", "tests/test_gpu_parity.py", "tests/test_gpu_baseline_configs.py", "tests/test_node_host.py", "tests/test_tiling_gloo.py",
                        "-m", "gpu", "--hostsim", "-q", "-x", "-k", sel, "-p", "no:cacheprovider"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0, tail
    assert " passed" in p.stdout and "failed" not in p.stdout, tail
    print(tail.splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("make") is None, reason="no host clang++ / make")
def test_differential_fuzz_of_the_kernels_against_the_oracle():
    """tools/fuzz_hostsim.py: random frame sizes (odd, tiny, portrait), option values and vUv models — every stage of two frames on the
    simulated kernels against the C restatement (80 cases here; 500 under AddressSanitizer were clean when this was written)."""
    sim = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", sim])
    from conftest import hostsim_child_env
    env = dict(os.environ, **hostsim_child_env(sim))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_hostsim.py"), "--n", "80", "--seed", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "0 problems" in p.stdout.splitlines()[-1]


@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("make") is None, reason="no host clang++ / make")
def test_differential_fuzz_of_the_variants_in_lock_step_with_the_oracle():
    """tools/fuzz_effects.py: random SSGIEffect / SSREffect options (mode, denoiseMode, iterations, radius, the phi's, resolutionScale ...), frame
    sizes, cameras, environments and fog — the effect drives the simulated kernels and the C restatement in lock step, every draw on identical
    inputs (60 cases here; 500 plain, 120 under AddressSanitizer and 600 on the device were clean when this was written) — and its self-test:
    with one parameter of the LIBRARY's draw perturbed the same limits must flag K1, K2 and K3."""
    sim = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", sim])
    from conftest import hostsim_child_env
    env = dict(os.environ, **hostsim_child_env(sim))
    tool = [sys.executable, os.path.join(ROOT, "tools", "fuzz_effects.py"), "--lib", os.path.join(sim, "_build", "librfx_hostsim.so")]
    p = subprocess.run(tool + ["--n", "60", "--seed", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert " 0 problems" in p.stdout.splitlines()[-1]
    p = subprocess.run(tool + ["--n", "12", "--seed", "5", "--self-test"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "self-test" in p.stdout.splitlines()[-1], (p.stdout + p.stderr)[-3000:]


@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("make") is None, reason="no host clang++ / make")
def test_k3_pass0_asks_for_an_lds_size_that_fits_three_times_into_a_cu():
    """The launcher's own arithmetic, executed: on a 16:9 frame at radius 3 the first denoise pass of two textures must ask for no more dynamic LDS
    than fits THREE times into a CU's 160 KiB handed out in 1 280-byte granules (measured: profiles/r05_microbench/lds_occupancy.txt) — 53 744 B since
    the staged rectangle's unreachable corners are not held, 53 872 B (two workgroups per CU, pass 0 13 % slower) before.  The later passes: 43 776 B."""
    sim = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", sim])
    code = (
        "import sys, ctypes\n"
        "sys.path[:0] = [%r, %r]\n"
        "import numpy as np\n"
        "from rfx_amd import abi\n"
        "from rfx_amd.context import Context\n"
        "from rfx_amd.scene import synthetic_frame\n"
        "W, H = 256, 144\n"
        "f = synthetic_frame(W, H, 0)\n"
        "ctx = Context(W, H)\n"
        "ctx.upload_frame(f)\n"
        "ctx.upload(abi.TEX_TEMPORAL0, np.ones((H, W, 4), np.float32)); ctx.upload(abi.TEX_TEMPORAL1, np.ones((H, W, 4), np.float32))\n"
        "dp = abi.DenoiseParams(radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, textureCount=2, blueNoiseIndex=5,\n"
        "                       inputIsTemporal=1, writeToB=0, halfStoreRTZ=1)\n"
        "dp.isTextureSpecular[:] = [0, 1]\n"
        "lib = abi.load_library()\n"
        "lib.rfx_hostsim_last_dynamic_lds.restype = ctypes.c_ulonglong\n"
        "ctx.poisson_denoise(dp); ctx.sync(); a = lib.rfx_hostsim_last_dynamic_lds()\n"
        "dp.inputIsTemporal, dp.writeToB = 0, 1\n"
        "ctx.poisson_denoise(dp); ctx.sync(); b = lib.rfx_hostsim_last_dynamic_lds()\n"
        "print(a, b)\n"
    ) % (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "tests"))
    from conftest import hostsim_child_env
    env = dict(os.environ, **hostsim_child_env(sim))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    pass0, later = (int(v) for v in p.stdout.split()[-2:])
    granule, cu = 1280, 160 * 1024
    fits = lambda nbytes: cu // (-(-nbytes // granule) * granule)  # noqa: E731  workgroups of that LDS size per CU
    assert pass0 == 16 + (74 * 14 - 4) * 52 + 64 == 53744 and fits(pass0) == 3, (pass0, fits(pass0))
    assert later == 76 * 16 * 36 == 43776 and fits(later) == 3, (later, fits(later))
    assert fits(74 * 14 * 52) == 2  # (what the whole rectangle would ask for)


def test_comm_entry_points_without_a_loadable_rccl_report_unsupported():
    """rfx_comm.hip binds RCCL with dlopen at first use; a host without it must get RFX_EUNSUPPORTED from every exchange entry point, not a
    crash (round 2's loader built its message from two dlerror() calls: the second returns NULL -> std::string + nullptr).  The simulator
    build looks for librccl_hostsim.so.1 only; this process does not put tests/hostsim/_build/fakerccl on its library path."""
    sim = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", sim])
    code = (
        "import ctypes, sys\n"
        "lib = ctypes.CDLL(sys.argv[1])\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = lib.rfx_comm_unique_id(buf)\n"
        "lib.rfx_create.restype = ctypes.c_void_p\n"
        "lib.rfx_create.argtypes = [ctypes.c_int] * 6\n"
        "ctx = lib.rfx_create(0, 64, 32, 0, 32, 0)\n"
        "lib.rfx_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]\n"
        "rc2 = lib.rfx_comm_init(ctx, buf, 0, 1)\n"
        "lib.rfx_last_error.restype = ctypes.c_char_p\n"
        "lib.rfx_last_error.argtypes = [ctypes.c_void_p]\n"
        "print(rc, rc2, lib.rfx_last_error(ctx).decode())\n"
    )
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "LD_PRELOAD")}
    p = subprocess.run([sys.executable, "-c", code, os.path.join(sim, "_build", "librfx_hostsim.so")], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    rc, rc2, msg = p.stdout.strip().split(" ", 2)
    assert int(rc) == -5 and int(rc2) == -5, p.stdout  # RFX_EUNSUPPORTED
    assert "RCCL" in msg
