import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "realism-effects_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# -m gpu tests that need the real device (RCCL, torch.cuda): not runnable under --hostsim
HOSTSIM_NEEDS_HARDWARE = ("test_bench_multi_rank_flow_on_one_gpu",)


def pytest_addoption(parser):
    parser.addoption("--hostsim", action="store_true", default=False,
                     help="run the -m gpu tests against tests/hostsim (the product's kernel sources compiled for x86: kernel LOGIC on the CPU, "
                          "no statement about the device's bits) instead of librfx_hip.so on a GPU")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference + llvmpipe (build container only)")
    if config.getoption("--hostsim"):
        import subprocess
        sim = os.path.join(ROOT, "tests", "hostsim")
        subprocess.check_call(["make", "-s", "-C", sim])
        os.environ["RFX_HIP_LIB"] = os.path.join(sim, "_build", "librfx_hostsim.so")
        os.environ["RFX_HOSTSIM"] = "1"
        # child processes (node + the N-API addon, which links librfx_hip.so by rpath): the simulator's rfx_* symbols interpose
        os.environ["LD_PRELOAD"] = os.environ["RFX_HIP_LIB"]
        # rfx_comm.hip binds RCCL at run time (dlopen; the simulator build looks for librccl_hostsim.so.1): tests/hostsim/fakerccl.c moves the
        # bytes over unix sockets instead
        import ctypes
        fake = os.path.join(sim, "_build", "fakerccl")
        ctypes.CDLL(os.path.join(fake, "librccl_hostsim.so.1"), mode=ctypes.RTLD_GLOBAL)  # this process: found by soname (RTLD_NOLOAD)
        os.environ["LD_LIBRARY_PATH"] = fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")  # child processes


def pytest_collection_modifyitems(config, items):
    if not config.getoption("--hostsim"):
        return
    skip = pytest.mark.skip(reason="--hostsim: needs the device (RCCL / torch.cuda)")
    for it in items:
        if any(n in it.nodeid for n in HOSTSIM_NEEDS_HARDWARE):
            it.add_marker(skip)


@pytest.fixture(scope="session")
def blue_noise():
    from rfx_amd.context import load_blue_noise_table
    return load_blue_noise_table()
