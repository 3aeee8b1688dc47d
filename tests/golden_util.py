"""Helpers to read tests/golden/*.npz (reference GLSL on llvmpipe, see make_golden.py)."""
import os
import types

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDENS = ["chain_160x90_s20r5_it1", "chain_97x55_s8r2_it2", "chain_missed_96x54_s12r3_it1"]
GOLDEN_SSR = "chain_ssr_128x72_s20r5_it1"
GOLDEN_ENV = ["chain_env_128x72_s12r3_it1", "chain_envsharp_96x54_s12r3_it1"]  # scene.environment (USE_ENVMAP), envBlur 0.5 / 0.1
GOLDEN_ORTHO = "chain_ortho_120x68_s12r3_it1"  # OrthographicCamera: every pass without its PERSPECTIVE_CAMERA define
GOLDEN_RS = ["chain_rs050_128x72_s12r3_it1"]  # resolutionScale 0.5 (SSGIPass.js:52-57); see make_golden.py on other scales
GOLDEN_ENVMIS = "chain_envmis_128x72_s12r3_it1"  # USE_ENVMAP + importanceSampling (the default with an environment)
GOLDEN_MODES = ["chain_full_temporal_104x58_s10r2", "chain_temporal_104x58_s10r2", "chain_denoised_104x58_s10r2"]  # Denoiser.js:7 denoiseMode
GOLDEN_PACK = "pack_96x54"  # packGBuffer / packNormal (the raster passes' fragment epilogues) over attribute planes
GOLDEN_FINAL = "final_112x63"  # SSGIEffect's own fragment: no fog / Fog / FogExp2 / isDebug
GOLDEN_TRAA = ["traa_half_128x72", "traa_float_96x54"]  # TRAAEffect: composer buffers HalfFloatType / FloatType


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def camera(g, fi):
    k = "f%d_cam_" % fi
    return types.SimpleNamespace(projectionMatrix=g[k + "projectionMatrix"], projectionMatrixInverse=g[k + "projectionMatrixInverse"],
                                 matrixWorld=g[k + "matrixWorld"], matrixWorldInverse=g[k + "matrixWorldInverse"], position=g[k + "position"],
                                 quaternion=g[k + "quaternion"], near=float(g["f%d_near" % fi]), far=float(g["f%d_far" % fi]),
                                 isPerspectiveCamera=not (("orthographic" in g.files) and int(g["orthographic"])))


def frame(g, fi):
    k = "f%d_" % fi
    return types.SimpleNamespace(depth=np.ascontiguousarray(g[k + "depth"]), gbuffer=np.ascontiguousarray(g[k + "gbuffer"]),
                                 velocity=np.ascontiguousarray(g[k + "velocity"]), direct=np.ascontiguousarray(g[k + "direct"]),
                                 camera=camera(g, fi), width=int(g["width"]), height=int(g["height"]))


def traa_frame(g, fi):
    """TRAA goldens carry the composer input buffer (`direct`), the velocity/normal/depth plane and the camera."""
    k = "f%d_" % fi
    return types.SimpleNamespace(velocity=np.ascontiguousarray(g[k + "velocity"]), direct=np.ascontiguousarray(g[k + "direct"]),
                                 camera=camera(g, fi), width=int(g["width"]), height=int(g["height"]))
