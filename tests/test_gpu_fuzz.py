"""GPU: the differential fuzzers as part of the suite — small batches of what tools/gpu_runs/gpu_r06_{w,x,y,z}.sh ran in the hundreds.
Each tool is its own process (its own random sizes, contexts and, for the two *_vs_reference_gl ones, GL context); under `pytest --hostsim` the
children load the simulator (tests/conftest.py hostsim_child_env)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOSTSIM = os.environ.get("RFX_HOSTSIM") == "1"


def _run(tool, *args, timeout=900):
    # the checkers' OpenMP teams over frames of a few hundred pixels: 256 threads cost two orders of magnitude (call Y against call Z)
    env = dict(os.environ, OMP_NUM_THREADS="32", LP_NUM_THREADS="32")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + list(args), capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    return p.stdout.splitlines()[-1]


def _gl_or_skip():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
    try:
        import chain
        chain.GL.info()
    except Exception as e:  # noqa: BLE001
        pytest.skip("reference GL unavailable: %s" % e)
    if not os.path.isdir("/root/reference/src") and not os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "shaders")):
        pytest.skip("no assembled reference shaders (make -C oracle ref)")


@pytest.mark.gpu
def test_fuzz_default_chain_and_row_tilings_against_the_restatement():
    """tools/fuzz_hostsim.py: random odd / tiny / portrait frame sizes, option values, vUv models and ragged row tilings as thin as the halo — every
    stage of two frames against the C restatement on identical inputs."""
    last = _run("fuzz_hostsim.py", *([] if HOSTSIM else ["--device"]), "--n", "40", "--seed", "101")
    assert last.endswith(" 0 problems"), last


@pytest.mark.gpu
def test_fuzz_effect_options_in_lock_step_with_the_restatement():
    """tools/fuzz_effects.py: random SSGIEffect / SSREffect / TRAAEffect options, cameras, environments and fog; the library and the restatement
    driven in lock step, every draw on identical inputs."""
    last = _run("fuzz_effects.py", *(["--lib", os.environ["RFX_TEST_LIB"]] if HOSTSIM else ["--device"]), "--n", "60", "--seed", "102")
    assert last.endswith(" 0 problems"), last


@pytest.mark.gpu
def test_fuzz_kernels_against_the_reference_glsl_default_chain():
    """tools/fuzz_vs_reference_gl.py --device: the kernels against the reference's own GLSL on llvmpipe at random sizes, step counts and option
    values — strict metric, every out-of-tolerance pixel proven by the restatement."""
    _gl_or_skip()
    last = _run("fuzz_vs_reference_gl.py", "--device", "--n", "24", "--seed", "103")
    assert " 0 unexplained; 0 errors" in last, last


@pytest.mark.gpu
def test_fuzz_kernels_against_the_reference_chain_over_the_variants():
    """tools/fuzz_variants_vs_reference_gl.py --device: mode ssgi / ssr, the four denoiseModes, resolutionScale, environment with / without importance
    sampling, fog — the effect host drives the kernels while the reference chain on llvmpipe makes the same draws in lock step; strict metric with
    proofs.  The odd-sized importance-sampling batch is the regression test of round 6's two findings (the quad partner outside the target; the
    equirect pole)."""
    _gl_or_skip()
    last = _run("fuzz_variants_vs_reference_gl.py", "--device", "--n", "30", "--seed", "104")
    assert " 0 unexplained; 0 errors" in last, last
    last = _run("fuzz_variants_vs_reference_gl.py", "--device", "--n", "12", "--seed", "105", "--only-envmis")
    assert " 0 unexplained; 0 errors" in last, last


@pytest.mark.gpu
def test_fuzz_cube_conversion_and_packers():
    """tools/fuzz_aux_vs_reference_gl.py --device: rfx_cube_to_equirect at random face sizes (odd ones, with and without the chain) and the importer's
    packers at random frame sizes — against the reference GLSL where its sources are, against the restatement on the GPU box (which the same tool
    holds against the reference GLSL in the build container: tests/test_parity_metric.py)."""
    if os.path.isdir("/root/reference/src"):
        _gl_or_skip()
    last = _run("fuzz_aux_vs_reference_gl.py", "--device", "--n", "80", "--seed", "106")
    assert last.endswith(" 0 problems"), last
