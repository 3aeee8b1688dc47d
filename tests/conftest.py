import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# -m gpu tests that need the real device (RCCL, torch.cuda): not runnable under --hostsim
HOSTSIM_NEEDS_HARDWARE = ("test_bench_multi_rank_flow_on_one_gpu",)
# ... and the ones whose frame is too large for kernels executed thread by thread on the CPU (minutes per draw)
HOSTSIM_TOO_LARGE = ("configs[4] 8K", "streamed_dumps_equal_uploaded_dumps[3840-2160") + (
    () if os.environ.get("RFX_TEST_SEQ_SIZE") else ("_16_frames",))  # (1080p x 16 frames: 9 minutes on the simulator; RFX_TEST_SEQ_SIZE=160x90 runs its logic)


# `-m "gpu and quick"`: the mid-round check — one BASELINE configuration against the reference GLSL, the 4K band, one variant of every widened
# row, the hosts and the exchanges; measured on MI355X: see tools/gpu_runs (the full `-m gpu` suite stays the round's last word)
QUICK = ("stagewise_vs_reference_glsl[configs[1]-", "full_size_4k_band", "chain_stagewise_vs_oracle[size2", "env_map_vs_oracle[0.5", "env_map_importance_sampling_vs_oracle",
         "ssr_mode_chain", "single_texture_variants", "traa_end_to_end_vs_oracle[True", "denoise_modes_vs_oracle[full_temporal", "final_compose_vs_oracle",
         "resolution_scale_vs_oracle[0.5", "import_attribute_planes", "orthographic_camera", "cube_to_equirect", "node_host_drives", "per_draw_profile",
         "row_windowed_draws", "row_tiled_chain_is_bit_identical_to_single_context[2", "trace_plus_shade_is_bit_identical_to_march[plain", "hit_rows_bound",
         "comm_entry_points_on_a_single_rank_ring", "multi_rank_flow_on_one_gpu[2-540-peer", "peer_history_gather_between", "rgb_history_twin", "nan_texels",
         "tiled_kernels_with_c_abi_exchanges", "config0_through_the_effect")


def hostsim_child_env(sim, build="_build"):
    """Environment of a python child process that must load the host simulator: tests/hostsim/inject/sitecustomize.py (found through
    PYTHONPATH at interpreter start) calls rfx_amd.abi.set_library_path(RFX_TEST_LIB)."""
    return {"RFX_TEST_LIB": os.path.join(sim, build, "librfx_hostsim.so"),
            "PYTHONPATH": os.path.join(sim, "inject") + os.pathsep + os.environ.get("PYTHONPATH", "")}


def pytest_addoption(parser):
    parser.addoption("--hostsim", action="store_true", default=False,
                     help="run the -m gpu tests against tests/hostsim (the product's kernel sources compiled for x86: kernel LOGIC on the CPU, "
                          "no statement about the device's bits) instead of librfx_hip.so on a GPU")
    parser.addoption("--hostsim-build", default="_build",
                     help="with --hostsim: the build of tests/hostsim to load (_build; _asan / _ubsan after `make -C tests/hostsim asan|ubsan`, with the "
                          "sanitizer runtime in LD_PRELOAD)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference + llvmpipe (build container only)")
    config.addinivalue_line("markers", "quick: the few-minute subset of the gpu tests (-m \"gpu and quick\"; the list: QUICK in tests/conftest.py)")
    if config.getoption("--hostsim"):
        import subprocess
        sim = os.path.join(ROOT, "tests", "hostsim")
        build = config.getoption("--hostsim-build")
        if build == "_build":
            subprocess.check_call(["make", "-s", "-C", sim])
        lib = os.path.join(sim, build, "librfx_hostsim.so")
        from rfx_amd import abi
        abi.set_library_path(lib)  # this process: explicit injection (the product loader reads no environment variable)
        for k, v in hostsim_child_env(sim, build).items():  # the processes the tests spawn
            os.environ[k] = v
        os.environ["RFX_HOSTSIM"] = "1"  # read by tests only (skip conditions)
        # child processes (node + the N-API addon, which links librfx_hip.so by rpath): the simulator's rfx_* symbols interpose
        os.environ["LD_PRELOAD"] = (os.environ["LD_PRELOAD"] + os.pathsep if os.environ.get("LD_PRELOAD") else "") + lib
        # rfx_comm.hip binds RCCL at run time (dlopen; the simulator build looks for librccl_hostsim.so.1): tests/hostsim/fakerccl.c moves the
        # bytes over unix sockets instead
        import ctypes
        fake = os.path.join(sim, "_build", "fakerccl")
        ctypes.CDLL(os.path.join(fake, "librccl_hostsim.so.1"), mode=ctypes.RTLD_GLOBAL)  # this process: found by soname (RTLD_NOLOAD)
        os.environ["LD_LIBRARY_PATH"] = fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")  # child processes


def pytest_collection_modifyitems(config, items):
    for it in items:
        if it.get_closest_marker("gpu") and any(n in it.nodeid for n in QUICK):
            it.add_marker(pytest.mark.quick)
    if not config.getoption("--hostsim"):
        return
    skip = pytest.mark.skip(reason="--hostsim: needs the device (RCCL / torch.cuda)")
    big = pytest.mark.skip(reason="--hostsim: an 8K frame on the simulator takes minutes per draw")
    for it in items:
        if any(n in it.nodeid for n in HOSTSIM_NEEDS_HARDWARE):
            it.add_marker(skip)
        if any(n in it.nodeid for n in HOSTSIM_TOO_LARGE):
            it.add_marker(big)


@pytest.fixture(scope="session")
def blue_noise():
    from rfx_amd.context import load_blue_noise_table
    return load_blue_noise_table()
