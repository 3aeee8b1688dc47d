"""ctypes mirror of include/rfx.h (the C ABI of librfx_hip.so) and the library loader.

There is deliberately NO fallback here: if the HIP library is missing or fails to load the
import raises — the product path never silently runs on a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "..", "csrc", "librfx_hip.so")

RFX_ABI_VERSION = 19
PEER_BLOB_BYTES = 192  # include/rfx.h RFX_PEER_BLOB_BYTES
# rfx_profile_read's kinds (include/rfx.h RFX_PROF_*)
PROF_KINDS = ("k1_prepass", "k1_ssgi_march", "k2_temporal_reproject", "k3_poisson_denoise_pass0", "k3_poisson_denoise_passN", "k4_compose", "k5_final_compose")
RFX_UV_IDEAL, RFX_UV_REFERENCE_GL = 0, 1  # rfx_set_uv_model
RFX_OK, RFX_EINVAL, RFX_ENOMEM, RFX_EDEVICE, RFX_ESTATE, RFX_EUNSUPPORTED = 0, -1, -2, -3, -4, -5

(TEX_DEPTH, TEX_GBUFFER, TEX_VELOCITY, TEX_DIRECT_LIGHT, TEX_BLUE_NOISE, TEX_SSGI, TEX_TEMPORAL0, TEX_TEMPORAL1,
 TEX_DENOISE_A0, TEX_DENOISE_A1, TEX_DENOISE_B0, TEX_DENOISE_B1, TEX_COMPOSE, TEX_FBCOPY_F16, TEX_FBCOPY_F32, TEX_FINAL, TEX_COMPOSE_RGB, TEX_COUNT) = range(18)

TEX_NAMES = ["depth", "gbuffer", "velocity", "direct_light", "blue_noise", "ssgi", "temporal0", "temporal1",
             "denoise_a0", "denoise_a1", "denoise_b0", "denoise_b1", "compose", "fbcopy_f16", "fbcopy_f32", "final", "compose_rgb"]
# (numpy dtype, channels) per slot, matching rfx_tex_texel_bytes()
TEX_FORMAT = {
    TEX_DEPTH: (np.float32, 1), TEX_GBUFFER: (np.uint32, 4), TEX_VELOCITY: (np.uint32, 4), TEX_DIRECT_LIGHT: (np.float32, 4),
    TEX_BLUE_NOISE: (np.uint8, 4), TEX_SSGI: (np.uint32, 4), TEX_TEMPORAL0: (np.float32, 4), TEX_TEMPORAL1: (np.float32, 4),
    TEX_DENOISE_A0: (np.uint16, 4), TEX_DENOISE_A1: (np.uint16, 4), TEX_DENOISE_B0: (np.uint16, 4), TEX_DENOISE_B1: (np.uint16, 4),
    TEX_COMPOSE: (np.float32, 4), TEX_FBCOPY_F16: (np.uint16, 4), TEX_FBCOPY_F32: (np.float32, 4), TEX_FINAL: (np.float32, 4),
    TEX_COMPOSE_RGB: (np.float32, 3),
}

M16 = C.c_float * 16


class Camera(C.Structure):
    _fields_ = [("projectionMatrix", M16), ("projectionMatrixInverse", M16), ("matrixWorld", M16), ("matrixWorldInverse", M16),
                ("position", C.c_float * 3), ("near_", C.c_float), ("far_", C.c_float), ("isPerspective", C.c_int32)]

    @staticmethod
    def from_scene(cam) -> "Camera":
        """From a dumped camera object (rfx_amd.scene.Camera or anything with the same fields)."""
        c = Camera()
        for name in ("projectionMatrix", "projectionMatrixInverse", "matrixWorld", "matrixWorldInverse"):
            getattr(c, name)[:] = [float(x) for x in np.asarray(getattr(cam, name), np.float32).ravel()]
        c.position[:] = [float(x) for x in np.asarray(cam.position, np.float32).ravel()]
        c.near_, c.far_ = float(cam.near), float(cam.far)
        c.isPerspective = 1 if getattr(cam, "isPerspectiveCamera", True) else 0
        return c


class SsgiParams(C.Structure):
    _fields_ = [("camera", Camera), ("steps", C.c_int32), ("refineSteps", C.c_int32), ("mode", C.c_int32),
                ("useDirectLight", C.c_int32), ("missedRays", C.c_int32), ("importanceSampling", C.c_int32), ("useEnvMap", C.c_int32),
                ("rayDistance", C.c_float), ("thickness", C.c_float), ("envBlur", C.c_float), ("blueNoiseIndex", C.c_int32),
                ("resolutionScale", C.c_float), ("historySource", C.c_int32)]


class TemporalParams(C.Structure):
    _fields_ = [("camera", Camera), ("prevCamera", Camera), ("textureCount", C.c_int32), ("inputType", C.c_int32),
                ("reprojectSpecular", C.c_int32 * 2), ("neighborhoodClamp", C.c_int32 * 2), ("logTransform", C.c_int32),
                ("fullAccumulate", C.c_int32), ("confidencePower", C.c_float), ("neighborhoodClampIntensity", C.c_float),
                ("maxBlend", C.c_float), ("keepData", C.c_float), ("historySource", C.c_int32), ("targetHalf", C.c_int32),
                ("halfStoreRTZ", C.c_int32), ("inputWidth", C.c_int32), ("inputHeight", C.c_int32)]


class DenoiseParams(C.Structure):
    _fields_ = [("radius", C.c_float), ("phi", C.c_float), ("lumaPhi", C.c_float), ("depthPhi", C.c_float), ("normalPhi", C.c_float),
                ("roughnessPhi", C.c_float), ("specularPhi", C.c_float), ("textureCount", C.c_int32),
                ("isTextureSpecular", C.c_int32 * 2), ("blueNoiseIndex", C.c_int32), ("inputIsTemporal", C.c_int32),
                ("writeToB", C.c_int32), ("halfStoreRTZ", C.c_int32)]


class ComposeParams(C.Structure):
    _fields_ = [("camera", Camera), ("inputType", C.c_int32), ("giSource", C.c_int32), ("writeHistoryRGB", C.c_int32)]


FP = C.POINTER(C.c_float)


class AovGBuffer(C.Structure):
    _fields_ = [("diffuse", FP), ("normal", FP), ("roughness", FP), ("metalness", FP), ("emissive", FP), ("depth", FP)]


class AovVelocity(C.Structure):
    _fields_ = [("velocity", FP), ("normal", FP), ("depth", FP)]


class FinalParams(C.Structure):
    _fields_ = [("camera", Camera), ("isDebug", C.c_int32), ("inputSource", C.c_int32), ("fogMode", C.c_int32), ("fogColor", C.c_float * 3), ("fogNear", C.c_float),
                ("fogFar", C.c_float), ("fogDensity", C.c_float)]


EXPORTS = [
    "rfx_abi_version", "rfx_create", "rfx_destroy", "rfx_last_error", "rfx_get_geometry", "rfx_set_stream", "rfx_tex_texel_bytes", "rfx_tex_held_rows",
    "rfx_upload", "rfx_download", "rfx_clear", "rfx_tex_device_ptr", "rfx_bind_external", "rfx_pack_gbuffer", "rfx_pack_velocity", "rfx_set_environment", "rfx_set_environment_importance", "rfx_download_environment", "rfx_cube_to_equirect", "rfx_set_row_window", "rfx_set_uv_model", "rfx_ssgi_march", "rfx_ssgi_trace", "rfx_ssgi_shade", "rfx_temporal_reproject",
    "rfx_copy_framebuffer", "rfx_poisson_denoise", "rfx_compose", "rfx_final_compose", "rfx_sync", "rfx_halo_violations", "rfx_time_begin", "rfx_time_end", "rfx_profile", "rfx_profile_read",
    "rfx_host_alloc", "rfx_host_free", "rfx_stage_upload", "rfx_stage_flip", "rfx_split_rows", "rfx_comm_unique_id", "rfx_comm_init", "rfx_comm_destroy", "rfx_halo_exchange", "rfx_allgather_history", "rfx_gather_history_rows", "rfx_peer_export", "rfx_peer_open", "rfx_peer_gather_history", "rfx_peer_close", "rfx_ssgi_hit_rows", "rfx_ssgi_hit_mask", "rfx_comm_wait",
]

_lib = None
_lib_path = None  # set_library_path(): an explicit path instead of the in-tree librfx_hip.so


def set_library_path(path: str | None) -> None:
    """Load `path` instead of the in-tree csrc/librfx_hip.so from now on (development: a tuning variant of the library; tests: the host
    simulator, injected by tests/conftest.py).  Explicit on purpose — nothing in this package reads a library path from the environment."""
    global _lib, _lib_path
    _lib, _lib_path = None, (os.path.abspath(path) if path else None)


def injected_library_path() -> str | None:
    """the path set_library_path() installed, or None when the in-tree library is in use"""
    return _lib_path


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen librfx_hip.so and declare prototypes.  Raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = os.path.abspath(path or _lib_path or LIB_PATH)
    if not os.path.exists(p):
        raise ImportError(
            "librfx_hip.so not found at %s — build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the product path." % p)
    lib = C.CDLL(p)
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    lib.rfx_abi_version.restype = i
    lib.rfx_create.restype = vp
    lib.rfx_create.argtypes = [i, i, i, i, i, i]
    lib.rfx_destroy.argtypes = [vp]
    lib.rfx_destroy.restype = None
    lib.rfx_last_error.argtypes = [vp]
    lib.rfx_last_error.restype = C.c_char_p
    lib.rfx_get_geometry.argtypes = [vp] + [C.POINTER(i)] * 5
    lib.rfx_set_stream.argtypes = [vp, vp]
    lib.rfx_tex_texel_bytes.argtypes = [i]
    lib.rfx_tex_texel_bytes.restype = C.c_size_t
    lib.rfx_tex_held_rows.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
    lib.rfx_upload.argtypes = [vp, i, vp, i, i]
    lib.rfx_download.argtypes = [vp, i, vp, i, i]
    lib.rfx_clear.argtypes = [vp, i]
    lib.rfx_tex_device_ptr.argtypes = [vp, i]
    lib.rfx_tex_device_ptr.restype = vp
    lib.rfx_bind_external.argtypes = [vp, i, vp]
    lib.rfx_pack_gbuffer.argtypes = [vp, C.POINTER(AovGBuffer), i, i]
    lib.rfx_pack_velocity.argtypes = [vp, C.POINTER(AovVelocity), i, i]
    lib.rfx_set_environment.argtypes = [vp, vp, i, i, i, i]
    lib.rfx_set_environment_importance.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, f, f]
    lib.rfx_download_environment.argtypes = [vp, i, vp, C.POINTER(i)]
    lib.rfx_set_row_window.argtypes = [vp, C.c_int, C.c_int]
    lib.rfx_set_uv_model.argtypes = [vp, C.c_int]
    lib.rfx_cube_to_equirect.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    lib.rfx_ssgi_march.argtypes = [vp, C.POINTER(SsgiParams)]
    lib.rfx_ssgi_trace.argtypes = [vp, C.POINTER(SsgiParams)]
    lib.rfx_ssgi_shade.argtypes = [vp, C.POINTER(SsgiParams)]
    lib.rfx_temporal_reproject.argtypes = [vp, C.POINTER(TemporalParams)]
    lib.rfx_copy_framebuffer.argtypes = [vp, i]
    lib.rfx_poisson_denoise.argtypes = [vp, C.POINTER(DenoiseParams)]
    lib.rfx_compose.argtypes = [vp, C.POINTER(ComposeParams)]
    lib.rfx_final_compose.argtypes = [vp, C.POINTER(FinalParams)]
    lib.rfx_sync.argtypes = [vp]
    lib.rfx_halo_violations.argtypes = [vp]
    lib.rfx_halo_violations.restype = C.c_uint
    lib.rfx_time_begin.argtypes = [vp]
    lib.rfx_time_end.argtypes = [vp, C.POINTER(f)]
    lib.rfx_profile.argtypes = [vp, i]
    lib.rfx_profile_read.argtypes = [vp, C.POINTER(f), C.POINTER(i)]
    lib.rfx_host_alloc.argtypes = [C.c_size_t]
    lib.rfx_host_alloc.restype = vp
    lib.rfx_host_free.argtypes = [vp]
    lib.rfx_host_free.restype = None
    lib.rfx_stage_upload.argtypes = [vp, i, vp, i, i]
    lib.rfx_stage_flip.argtypes = [vp]
    lib.rfx_split_rows.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
    lib.rfx_comm_unique_id.argtypes = [vp]
    lib.rfx_comm_init.argtypes = [vp, vp, i, i]
    lib.rfx_comm_destroy.argtypes = [vp]
    lib.rfx_halo_exchange.argtypes = [vp, i, vp, i, i]
    lib.rfx_allgather_history.argtypes = [vp, i, vp]
    lib.rfx_gather_history_rows.argtypes = [vp, i, vp, C.POINTER(C.c_size_t)]
    lib.rfx_peer_export.argtypes = [vp, i, vp]
    lib.rfx_peer_open.argtypes = [vp, i, vp, i, i]
    lib.rfx_peer_gather_history.argtypes = [vp, i, C.POINTER(C.c_size_t)]
    lib.rfx_peer_close.argtypes = [vp]
    lib.rfx_ssgi_hit_rows.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    lib.rfx_ssgi_hit_mask.argtypes = [vp, C.POINTER(C.c_uint32), i]
    lib.rfx_comm_wait.argtypes = [vp]
    if lib.rfx_abi_version() != RFX_ABI_VERSION:
        raise ImportError("librfx_hip.so ABI version %d != %d" % (lib.rfx_abi_version(), RFX_ABI_VERSION))
    if path is None:
        _lib = lib
    return lib
