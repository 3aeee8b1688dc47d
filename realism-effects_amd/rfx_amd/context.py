"""Thin object wrapper over the C ABI (include/rfx.h): one `Context` = one `rfx_ctx` =
the device-side state of the pass chain on one GPU (or one row tile of it)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data")


class RfxError(RuntimeError):
    pass


def load_blue_noise_table() -> np.ndarray:
    """128x128 RGBA8 table, decoded once from the reference's PNG asset, already flipY'd
    (tools/make_blue_noise_table.py; src/utils/BlueNoiseUtils.js:9-15)."""
    t = np.fromfile(os.path.join(DATA_DIR, "blue_noise_128_rgba8.bin"), np.uint8)
    assert t.size == 128 * 128 * 4
    return t.reshape(128, 128, 4)


class Context:
    def __init__(self, width: int, height: int, device: int = 0, tile_y0: int = 0, tile_rows: int | None = None, halo_rows: int = 0):
        self.lib = abi.load_library()
        self.W, self.H = int(width), int(height)
        self.tile_y0 = int(tile_y0)
        self.tile_rows = int(tile_rows if tile_rows is not None else height - tile_y0)
        self.halo = int(halo_rows)
        self._h = self.lib.rfx_create(int(device), self.W, self.H, self.tile_y0, self.tile_rows, self.halo)
        if not self._h:
            raise RfxError("rfx_create failed: %s" % self.lib.rfx_last_error(None).decode())
        self.upload(abi.TEX_BLUE_NOISE, load_blue_noise_table(), 0, 128)

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.lib.rfx_destroy(self._h)
            self._h = None
            self.__dict__.pop("_staged_keepalive", None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RfxError("%s failed (%d): %s" % (what, rc, self.lib.rfx_last_error(self._h).decode()))

    # -- textures
    def held_rows(self, tex: int):
        r0, n = C.c_int(), C.c_int()
        self._chk(self.lib.rfx_tex_held_rows(self._h, tex, C.byref(r0), C.byref(n)), "rfx_tex_held_rows")
        return r0.value, n.value

    def upload(self, tex: int, array: np.ndarray, row0: int | None = None, rows: int | None = None):
        """Upload rows [row0, row0+rows) (FRAME rows; default = everything the context holds).
        `array` holds exactly those rows."""
        dtype, ch = abi.TEX_FORMAT[tex]
        h0, hn = self.held_rows(tex)
        row0 = h0 if row0 is None else row0
        rows = hn if rows is None else rows
        a = np.ascontiguousarray(array)
        if a.dtype != dtype:
            if a.dtype.itemsize == np.dtype(dtype).itemsize:
                a = a.view(dtype)
            else:
                raise TypeError("texture %s wants %s, got %s" % (abi.TEX_NAMES[tex], np.dtype(dtype), a.dtype))
        width = 128 if tex == abi.TEX_BLUE_NOISE else self.W
        if a.size != rows * width * ch:
            raise ValueError("texture %s: expected %d x %d x %d elements, got %s" % (abi.TEX_NAMES[tex], rows, width, ch, a.shape))
        self._chk(self.lib.rfx_upload(self._h, tex, a.ctypes.data_as(C.c_void_p), row0, rows), "rfx_upload")

    def download(self, tex: int, row0: int | None = None, rows: int | None = None) -> np.ndarray:
        dtype, ch = abi.TEX_FORMAT[tex]
        h0, hn = self.held_rows(tex)
        row0 = h0 if row0 is None else row0
        rows = hn if rows is None else rows
        width = 128 if tex == abi.TEX_BLUE_NOISE else self.W
        out = np.empty((rows, width, ch) if ch > 1 else (rows, width), dtype)
        self._chk(self.lib.rfx_download(self._h, tex, out.ctypes.data_as(C.c_void_p), row0, rows), "rfx_download")
        return out

    # -- streaming dumps (rfx.h "streaming dumps"): the next frame's planes cross PCIe while the current frame is drawn
    def host_alloc(self, shape, dtype) -> np.ndarray:
        """A pinned (hipHostMalloc) numpy array: what makes rfx_stage_upload asynchronous.  The memory lives as long as the array (or any
        view of it) does: it is freed by a finalizer of the buffer object the array is built on, not with the context."""
        import weakref
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.rfx_host_alloc(n)
        if not p:
            raise RfxError("rfx_host_alloc(%d bytes) failed" % n)
        buf = (C.c_char * n).from_address(p)
        weakref.finalize(buf, self.lib.rfx_host_free, p)  # numpy keeps `buf` alive as the base of every view
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def stage_upload(self, tex: int, array: np.ndarray, row0: int | None = None, rows: int | None = None):
        """Asynchronous upload of rows [row0, row0+rows) into the slot's BACK buffer; published by stage_flip()."""
        h0, hn = self.held_rows(tex)
        row0 = h0 if row0 is None else row0
        rows = hn if rows is None else rows
        dtype, ch = abi.TEX_FORMAT[tex]
        a = array if array.flags["C_CONTIGUOUS"] else np.ascontiguousarray(array)
        if a.nbytes != rows * self.W * ch * np.dtype(dtype).itemsize:
            raise ValueError("texture %s: %d bytes do not cover %d rows" % (abi.TEX_NAMES[tex], a.nbytes, rows))
        self.__dict__.setdefault("_staged_now", []).append(a)  # kept alive until the copies of this batch have executed
        self._chk(self.lib.rfx_stage_upload(self._h, tex, a.ctypes.data_as(C.c_void_p), row0, rows), "rfx_stage_upload")

    def stage_frame(self, frame):
        for tex, plane in ((abi.TEX_DEPTH, frame.depth), (abi.TEX_GBUFFER, frame.gbuffer), (abi.TEX_VELOCITY, frame.velocity),
                           (abi.TEX_DIRECT_LIGHT, frame.direct)):
            r0, n = self.held_rows(tex)
            self.stage_upload(tex, plane[r0:r0 + n] if plane.shape[0] != n else plane, r0, n)

    def stage_flip(self):
        self._chk(self.lib.rfx_stage_flip(self._h), "rfx_stage_flip")
        # rfx_stage_flip returns when the copies published by the PREVIOUS flip have executed: keep this batch's planes and the one before
        gens = self.__dict__.setdefault("_staged_keepalive", [])
        gens.append(self.__dict__.pop("_staged_now", []))
        del gens[:-2]

    def clear(self, tex: int):
        self._chk(self.lib.rfx_clear(self._h, tex), "rfx_clear")

    def device_ptr(self, tex: int) -> int:
        p = self.lib.rfx_tex_device_ptr(self._h, tex)
        if not p:
            raise RfxError("rfx_tex_device_ptr: %s" % self.lib.rfx_last_error(self._h).decode())
        return p

    def bind_external(self, tex: int, device_ptr: int):
        self._chk(self.lib.rfx_bind_external(self._h, tex, C.c_void_p(device_ptr)), "rfx_bind_external")

    def set_stream(self, hip_stream: int | None):
        self._chk(self.lib.rfx_set_stream(self._h, C.c_void_p(hip_stream or 0)), "rfx_set_stream")

    def upload_frame(self, frame):
        """Upload a dumped frame (rfx_amd.scene.Frame or any object with depth/gbuffer/velocity/direct)
        whose planes cover the FULL frame; each slot takes the band it holds."""
        for tex, plane in ((abi.TEX_DEPTH, frame.depth), (abi.TEX_GBUFFER, frame.gbuffer), (abi.TEX_VELOCITY, frame.velocity),
                           (abi.TEX_DIRECT_LIGHT, frame.direct)):
            r0, n = self.held_rows(tex)
            self.upload(tex, plane[r0:r0 + n], r0, n)

    # -- importer: engine-side attribute planes -> packed render targets, on the device
    @staticmethod
    def _plane(a, ch, rows, width):
        a = np.ascontiguousarray(a, np.float32)
        if a.size != rows * width * ch:
            raise ValueError("AOV plane: expected %d x %d x %d floats, got %s" % (rows, width, ch, a.shape))
        return a

    def pack_gbuffer(self, aov: dict, depth=None, row0: int | None = None, rows: int | None = None):
        """rfx_pack_gbuffer: `aov` holds diffuse (RGBA), normal (xyz, world), roughness, metalness, emissive (rgb) planes of the band
        [row0, row0+rows) (default: everything the context holds); `depth` (1.0 = not covered) may be None."""
        h0, hn = self.held_rows(abi.TEX_GBUFFER)
        row0, rows = (h0 if row0 is None else row0), (hn if rows is None else rows)
        keep = [self._plane(aov[k], ch, rows, self.W) for k, ch in (("diffuse", 4), ("normal", 3), ("roughness", 1), ("metalness", 1), ("emissive", 3))]
        d = self._plane(depth, 1, rows, self.W) if depth is not None else None
        ptr = lambda a: a.ctypes.data_as(abi.FP) if a is not None else None
        s = abi.AovGBuffer(ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), ptr(keep[3]), ptr(keep[4]), ptr(d))
        self._chk(self.lib.rfx_pack_gbuffer(self._h, C.byref(s), row0, rows), "rfx_pack_gbuffer")

    def pack_velocity(self, aov: dict, depth, row0: int | None = None, rows: int | None = None):
        h0, hn = self.held_rows(abi.TEX_VELOCITY)
        row0, rows = (h0 if row0 is None else row0), (hn if rows is None else rows)
        v, n, d = self._plane(aov["velocity"], 2, rows, self.W), self._plane(aov["normal"], 3, rows, self.W), self._plane(depth, 1, rows, self.W)
        s = abi.AovVelocity(v.ctypes.data_as(abi.FP), n.ctypes.data_as(abi.FP), d.ctypes.data_as(abi.FP))
        self._chk(self.lib.rfx_pack_velocity(self._h, C.byref(s), row0, rows), "rfx_pack_velocity")

    def set_environment(self, rgba, half_float_type=True, half_store_rtz=True):
        """scene.environment: an (H, W, 4) float32 equirectangular map (row 0 = bottom), or None to remove it."""
        if rgba is None:
            self._chk(self.lib.rfx_set_environment(self._h, None, 0, 0, 0, 0), "rfx_set_environment")
            return
        a = np.ascontiguousarray(rgba, np.float32)
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("environment map must be (H, W, 4)")
        self._chk(self.lib.rfx_set_environment(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], 1 if half_float_type else 0,
                                               1 if half_store_rtz else 0), "rfx_set_environment")

    def set_environment_importance(self, marginal, conditional, total_sum: float):
        """The tables of EquirectHdrInfoUniform.updateFrom (rfx_amd.envmap.build_importance) for importanceSampling; totalSumValue is split the
        way the reference splits it for its uniform struct (`~~total` and the rest, :391-394)."""
        m = np.ascontiguousarray(marginal, np.float32)
        c = np.ascontiguousarray(conditional, np.float32)
        whole = float(int(total_sum))
        # the library checks the counts against the environment's size (height / width*height floats)
        self._chk(self.lib.rfx_set_environment_importance(self._h, m.ctypes.data_as(C.c_void_p), m.size, c.ctypes.data_as(C.c_void_p), c.size, whole,
                                                          float(total_sum - whole)), "rfx_set_environment_importance")

    def download_environment(self, level: int, size) -> np.ndarray:
        """Mip level `level` of the environment; `size` = (width, height) of the base level."""
        w, h = max(size[0] >> level, 1), max(size[1] >> level, 1)
        out = np.empty((h, w, 4), np.float32)
        self._chk(self.lib.rfx_download_environment(self._h, level, out.ctypes.data_as(C.c_void_p), None), "rfx_download_environment")
        return out

    def cube_to_equirect(self, faces, width: int, height: int, generate_mipmaps: bool = False) -> np.ndarray:
        """CubeToEquirectEnvPass's draw + read-back (rfx_cube_to_equirect): faces (6, S, S, 4) float32 (+X -X +Y -Y +Z -Z, row j = t as
        uploaded) -> (height, width, 4) float32, row 0 = bottom."""
        faces = np.ascontiguousarray(faces, np.float32)
        if faces.ndim != 4 or faces.shape[0] != 6 or faces.shape[1] != faces.shape[2] or faces.shape[3] != 4:
            raise ValueError("cube_to_equirect: faces must be (6, S, S, 4) float32")
        out = np.empty((int(height), int(width), 4), np.float32)
        self._chk(self.lib.rfx_cube_to_equirect(self._h, faces.ctypes.data_as(C.c_void_p), faces.shape[1], 1 if generate_mipmaps else 0,
                                                out.ctypes.data_as(C.c_void_p), int(width), int(height)), "rfx_cube_to_equirect")
        return out

    def environment_levels(self) -> int:
        n = C.c_int()
        self._chk(self.lib.rfx_download_environment(self._h, 0, None, C.byref(n)), "rfx_download_environment")
        return n.value

    def set_row_window(self, y0: int = 0, y1: int = 0):
        """Restrict the rows the following draws produce to [y0, y1); no arguments (or y1 <= y0) resets (rfx_set_row_window)."""
        self._chk(self.lib.rfx_set_row_window(self._h, int(y0), int(y1)), "rfx_set_row_window")

    def set_uv_model(self, model="reference_gl"):
        """Which vUv the following draws' fragments see: "ideal" = (i + 0.5) / n; "reference_gl" = the reference GL's rasteriser value, bit
        for bit (include/rfx.h rfx_set_uv_model; what the parity tests against the reference GLSL on llvmpipe are tightest under)."""
        m = {"ideal": abi.RFX_UV_IDEAL, "reference_gl": abi.RFX_UV_REFERENCE_GL}.get(model, model)
        self._chk(self.lib.rfx_set_uv_model(self._h, int(m)), "rfx_set_uv_model")

    def ssgi_march(self, p: abi.SsgiParams):
        self._chk(self.lib.rfx_ssgi_march(self._h, C.byref(p)), "rfx_ssgi_march")

    def ssgi_trace(self, p: abi.SsgiParams):
        """First half of ssgi_march (up to the end of the ray march); see rfx.h."""
        self._chk(self.lib.rfx_ssgi_trace(self._h, C.byref(p)), "rfx_ssgi_trace")

    def ssgi_shade(self, p: abi.SsgiParams):
        """Second half: shades the traced rays; the only part that reads last frame's composed GI."""
        self._chk(self.lib.rfx_ssgi_shade(self._h, C.byref(p)), "rfx_ssgi_shade")

    def temporal_reproject(self, p: abi.TemporalParams):
        self._chk(self.lib.rfx_temporal_reproject(self._h, C.byref(p)), "rfx_temporal_reproject")

    def copy_framebuffer(self, dst: int):
        self._chk(self.lib.rfx_copy_framebuffer(self._h, dst), "rfx_copy_framebuffer")

    def poisson_denoise(self, p: abi.DenoiseParams):
        self._chk(self.lib.rfx_poisson_denoise(self._h, C.byref(p)), "rfx_poisson_denoise")

    def compose(self, p: abi.ComposeParams):
        self._chk(self.lib.rfx_compose(self._h, C.byref(p)), "rfx_compose")

    def final_compose(self, p: abi.FinalParams):
        self._chk(self.lib.rfx_final_compose(self._h, C.byref(p)), "rfx_final_compose")

    def sync(self):
        self._chk(self.lib.rfx_sync(self._h), "rfx_sync")

    # -- row-tiled runs: RCCL exchanges behind the C ABI (rfx.h "row-tiled runs")
    @staticmethod
    def split_rows(height: int, nranks: int, rank: int):
        y0, n = C.c_int(), C.c_int()
        if abi.load_library().rfx_split_rows(int(height), int(nranks), int(rank), C.byref(y0), C.byref(n)) != 0:
            raise ValueError("rfx_split_rows(%d, %d, %d)" % (height, nranks, rank))
        return y0.value, n.value

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = abi.load_library().rfx_comm_unique_id(buf)
        if rc != 0:
            raise RfxError("rfx_comm_unique_id failed (%d): RCCL not loadable on this host?" % rc)
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, nranks: int):
        assert len(unique_id) == 128
        self._chk(self.lib.rfx_comm_init(self._h, C.c_char_p(unique_id), int(rank), int(nranks)), "rfx_comm_init")
        self.comm_rank, self.comm_nranks = int(rank), int(nranks)

    def comm_destroy(self):
        self._chk(self.lib.rfx_comm_destroy(self._h), "rfx_comm_destroy")

    def halo_exchange(self, tex: int, up_rank: int, down_rank: int):
        self._chk(self.lib.rfx_halo_exchange(self._h, tex, None, int(up_rank), int(down_rank)), "rfx_halo_exchange")

    def allgather_history(self, tex: int):
        self._chk(self.lib.rfx_allgather_history(self._h, tex, None), "rfx_allgather_history")

    def ssgi_hit_rows(self):
        """rfx_ssgi_hit_rows: after ssgi_trace, the inclusive (lo, hi) range of history rows the shade will read (hi < lo: none)."""
        lo, hi = C.c_int(0), C.c_int(0)
        self._chk(self.lib.rfx_ssgi_hit_rows(self._h, C.byref(lo), C.byref(hi)), "rfx_ssgi_hit_rows")
        return int(lo.value), int(hi.value)

    def ssgi_hit_mask(self) -> np.ndarray:
        """rfx_ssgi_hit_mask: after ssgi_trace, one uint32 per frame row — bit b set when the shade reads a history texel of that row in column
        block b (32 blocks across the frame); 0 = the row is not read."""
        m = np.zeros(self.H, np.uint32)
        self._chk(self.lib.rfx_ssgi_hit_mask(self._h, m.ctypes.data_as(C.POINTER(C.c_uint32)), self.H), "rfx_ssgi_hit_mask")
        return m

    def gather_history_rows(self, tex: int) -> int:
        """rfx_gather_history_rows (between ssgi_trace and ssgi_shade): only the rows of last frame's composed GI that some tile's rays
        will read travel, from their owners.  Returns the bytes this rank receives."""
        n = C.c_size_t(0)
        self._chk(self.lib.rfx_gather_history_rows(self._h, tex, None, C.byref(n)), "rfx_gather_history_rows")
        return int(n.value)

    # -- the device-driven history gather (rfx_peer_*: peer loads through IPC mappings, no RCCL)
    def peer_export(self, tex: int) -> bytes:
        """rfx_peer_export: the blob (IPC handles of the plane and of this rank's flag block) every other rank needs"""
        b = C.create_string_buffer(abi.PEER_BLOB_BYTES)
        self._chk(self.lib.rfx_peer_export(self._h, tex, b), "rfx_peer_export")
        return b.raw

    def peer_open(self, tex: int, blobs, rank: int, nranks: int):
        """rfx_peer_open: `blobs` = every rank's peer_export() in rank order"""
        raw = b"".join(blobs)
        assert len(raw) == nranks * abi.PEER_BLOB_BYTES
        self._chk(self.lib.rfx_peer_open(self._h, tex, raw, rank, nranks), "rfx_peer_open")

    def peer_gather_history(self, tex: int) -> int:
        """rfx_peer_gather_history (between ssgi_trace and ssgi_shade, on every rank): this rank's kernel pulls the column blocks its rays will
        read out of their owners' planes.  Returns the bytes the PREVIOUS call's kernel moved (nothing waits on the host)."""
        n = C.c_size_t(0)
        self._chk(self.lib.rfx_peer_gather_history(self._h, tex, C.byref(n)), "rfx_peer_gather_history")
        return int(n.value)

    def peer_close(self):
        self._chk(self.lib.rfx_peer_close(self._h), "rfx_peer_close")

    def comm_wait(self):
        self._chk(self.lib.rfx_comm_wait(self._h), "rfx_comm_wait")

    def time_begin(self):
        self._chk(self.lib.rfx_time_begin(self._h), "rfx_time_begin")

    def time_end(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.rfx_time_end(self._h, C.byref(ms)), "rfx_time_end")
        return ms.value


    def profile(self, enable=True):
        """Per-draw device timing inside a frame loop (include/rfx.h rfx_profile): True resets the sums and starts, False stops."""
        self._chk(self.lib.rfx_profile(self._h, 1 if enable else 0), "rfx_profile")

    def profile_read(self):
        """-> {kind: (summed ms, launches)} since the last profile(True), for the kinds that launched (abi.PROF_KINDS)"""
        n = len(abi.PROF_KINDS)
        ms, cnt = (C.c_float * n)(), (C.c_int * n)()
        self._chk(self.lib.rfx_profile_read(self._h, ms, cnt), "rfx_profile_read")
        return {abi.PROF_KINDS[k]: (float(ms[k]), int(cnt[k])) for k in range(n) if cnt[k]}

    def halo_violations(self) -> int:
        return int(self.lib.rfx_halo_violations(self._h))
