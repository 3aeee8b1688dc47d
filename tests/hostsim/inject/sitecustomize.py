"""TEST INFRASTRUCTURE.  `pytest --hostsim` (tests/conftest.py) puts this directory on the PYTHONPATH of the processes it spawns (workers of
the multi-process tests, torch.distributed.run ranks of bench.py, tools/fuzz_hostsim.py): every such interpreter then points rfx_amd.abi at the
host simulator named by RFX_TEST_LIB before anything else runs.  The product package itself reads no library path from the environment."""
import os
import sys

_lib = os.environ.get("RFX_TEST_LIB")
if _lib:
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    _pkg = os.path.join(_root, "realism-effects_amd")
    if _pkg not in sys.path:
        sys.path.insert(0, _pkg)
    from rfx_amd import abi as _abi

    _abi.set_library_path(_lib)
