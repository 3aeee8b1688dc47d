"""Test double: the `Context` interface (rfx_amd.context.Context) implemented on the CPU oracle.

TESTS ONLY.  It lets the host-side logic (rfx_amd.effect: option plumbing, ping-pong, history
wiring, blue-noise recurrence, keepData, tiling/halo hooks) be exercised without a GPU, and it is
the checker the -m gpu tests compare the HIP path against.  It is never importable from the
product package."""
import numpy as np

import rfx_oracle as O
from rfx_amd import abi
from rfx_amd.context import load_blue_noise_table


class OracleRenderer:
    def __init__(self, width, height, tile_y0=0, tile_rows=None, halo_rows=0):
        self.W, self.H = width, height
        self.tile_y0 = tile_y0
        self.tile_rows = tile_rows if tile_rows is not None else height - tile_y0
        self.halo = halo_rows
        self.tex = {}
        for t, (dtype, ch) in abi.TEX_FORMAT.items():
            if t == abi.TEX_BLUE_NOISE:
                self.tex[t] = load_blue_noise_table().copy()
            else:
                # the oracle addresses FULL frames; rows outside the held band simply stay zero
                self.tex[t] = np.zeros((height, width, ch) if ch > 1 else (height, width), dtype)
        self.calls = []

    def held_rows(self, tex):
        if tex in (abi.TEX_DEPTH, abi.TEX_COMPOSE, abi.TEX_COMPOSE_RGB):
            return 0, self.H
        if tex == abi.TEX_BLUE_NOISE:
            return 0, 128
        b0 = max(0, self.tile_y0 - self.halo)
        b1 = min(self.H, self.tile_y0 + self.tile_rows + self.halo)
        return b0, b1 - b0

    def upload(self, tex, array, row0=None, rows=None):
        h0, hn = self.held_rows(tex)
        row0 = h0 if row0 is None else row0
        rows = hn if rows is None else rows
        assert h0 <= row0 and row0 + rows <= h0 + hn, "band outside held rows"
        dtype, ch = abi.TEX_FORMAT[tex]
        a = np.ascontiguousarray(array)
        if a.dtype != dtype:
            a = a.view(dtype)
        self.tex[tex][row0:row0 + rows] = a.reshape(self.tex[tex][row0:row0 + rows].shape)

    def download(self, tex, row0=None, rows=None):
        h0, hn = self.held_rows(tex)
        row0 = h0 if row0 is None else row0
        rows = hn if rows is None else rows
        return self.tex[tex][row0:row0 + rows].copy()

    def _rows(self, extra=0):
        b0, n = self.held_rows(abi.TEX_SSGI)
        w0, w1 = getattr(self, "_window", (0, 1 << 30))
        return max(b0, self.tile_y0 - extra, w0), min(b0 + n, self.tile_y0 + self.tile_rows + extra, w1)

    def set_row_window(self, y0=0, y1=0):
        self.calls.append(("set_row_window", y0, y1))
        self._window = (y0, y1) if y1 > y0 else (0, 1 << 30)

    def pack_gbuffer(self, aov, depth=None, row0=None, rows=None):
        h0, hn = self.held_rows(abi.TEX_GBUFFER)
        row0, rows = (h0 if row0 is None else row0), (hn if rows is None else rows)
        self.tex[abi.TEX_GBUFFER][row0:row0 + rows] = O.pack_gbuffer(aov, depth)

    def pack_velocity(self, aov, depth, row0=None, rows=None):
        h0, hn = self.held_rows(abi.TEX_VELOCITY)
        row0, rows = (h0 if row0 is None else row0), (hn if rows is None else rows)
        self.tex[abi.TEX_VELOCITY][row0:row0 + rows] = O.pack_velocity(aov, depth)

    def set_environment(self, rgba, half_float_type=True, half_store_rtz=True):
        self.calls.append(("set_environment", None if rgba is None else tuple(rgba.shape)))
        self.env = None if rgba is None else O.EnvMap(rgba, half=half_float_type, rtz=half_store_rtz)

    def cube_to_equirect(self, faces, width, height, generate_mipmaps=False):
        self.calls.append(("cube_to_equirect", (tuple(np.shape(faces)), width, height, bool(generate_mipmaps))))
        return O.cube_to_equirect(faces, width, height, mipmaps=generate_mipmaps)

    def set_environment_importance(self, marginal, conditional, total_sum):
        self.calls.append(("set_environment_importance", float(total_sum)))
        self.env.set_importance(marginal, conditional, total_sum)

    def ssgi_march(self, p):
        self.calls.append(("ssgi", p.blueNoiseIndex))
        t = self.tex
        hist = t[abi.TEX_TEMPORAL0] if p.historySource == 1 else t[abi.TEX_COMPOSE]
        if p.historySource == 3:  # the RGB twin of the composed GI: same texels, so the oracle reads them as source 0
            hist = np.concatenate([t[abi.TEX_COMPOSE_RGB], np.zeros((self.H, self.W, 1), np.float32)], axis=-1)
            p = abi.SsgiParams.from_buffer_copy(p)
            p.historySource = 0
        rs = p.resolutionScale or 1.0
        if rs != 1.0:  # the smaller render target lives at the start of the slot, pitch W*s (as on the device)
            oH, oW = int(self.H * rs), int(self.W * rs)
            out = t[abi.TEX_SSGI].reshape(-1)[:oH * oW * 4].reshape(oH, oW, 4)
            O.ssgi(t[abi.TEX_DEPTH], t[abi.TEX_GBUFFER], t[abi.TEX_DIRECT_LIGHT], hist, t[abi.TEX_BLUE_NOISE], p, out=out, env=getattr(self, "env", None))
            return
        O.ssgi(t[abi.TEX_DEPTH], t[abi.TEX_GBUFFER], t[abi.TEX_DIRECT_LIGHT], hist, t[abi.TEX_BLUE_NOISE], p,
               out=t[abi.TEX_SSGI], rows=self._rows(min(2, self.halo)), env=getattr(self, "env", None))

    # the split draw: the double has nothing to overlap, so the trace is a no-op and the shade runs the whole fragment —
    # which makes a shade that starts before the composed-GI all-gather has landed visible as a wrong frame
    def ssgi_trace(self, p):
        self.calls.append(("ssgi_trace", p.blueNoiseIndex))
        self._traced = True

    def ssgi_shade(self, p):
        assert getattr(self, "_traced", False), "ssgi_shade without ssgi_trace"
        self._traced = False
        self.ssgi_march(p)

    def temporal_reproject(self, p):
        self.calls.append(("temporal", p.keepData, p.fullAccumulate))
        t = self.tex
        if p.historySource == 0:
            h0 = t[abi.TEX_DENOISE_B0]
            h1 = t[abi.TEX_DENOISE_B1] if p.textureCount == 2 else h0
        else:
            h0 = h1 = t[abi.TEX_FBCOPY_F16 if p.historySource == 1 else abi.TEX_FBCOPY_F32]
        src = t[abi.TEX_SSGI]
        if p.inputWidth:
            src = src.reshape(-1)[:p.inputHeight * p.inputWidth * 4].reshape(p.inputHeight, p.inputWidth, 4)
        if self._rows()[1] <= self._rows()[0]:
            return
        O.temporal(src, t[abi.TEX_VELOCITY], h0, h1, p, t[abi.TEX_TEMPORAL0], t[abi.TEX_TEMPORAL1], rows=self._rows())

    def copy_framebuffer(self, dst):
        self.calls.append(("copy_framebuffer", dst))
        y0, y1 = self._rows()
        src = self.tex[abi.TEX_TEMPORAL0][y0:y1]
        if dst == abi.TEX_FBCOPY_F16:
            self.tex[dst][y0:y1] = src.astype(np.float16).view(np.uint16)  # exact: the target was drawn with targetHalf
        else:
            self.tex[dst][y0:y1] = src

    def poisson_denoise(self, p):
        self.calls.append(("denoise", p.blueNoiseIndex, p.inputIsTemporal, p.writeToB))
        t = self.tex
        if p.inputIsTemporal:
            i0, i1 = t[abi.TEX_TEMPORAL0], t[abi.TEX_TEMPORAL1]
        elif p.writeToB:
            i0, i1 = t[abi.TEX_DENOISE_A0], t[abi.TEX_DENOISE_A1]
        else:
            i0, i1 = t[abi.TEX_DENOISE_B0], t[abi.TEX_DENOISE_B1]
        o0, o1 = (t[abi.TEX_DENOISE_B0], t[abi.TEX_DENOISE_B1]) if p.writeToB else (t[abi.TEX_DENOISE_A0], t[abi.TEX_DENOISE_A1])
        if p.textureCount == 1:
            i1 = i0
        if self._rows()[1] <= self._rows()[0]:
            return
        O.denoise(t[abi.TEX_DEPTH], t[abi.TEX_GBUFFER], i0, i1, t[abi.TEX_BLUE_NOISE], p, o0, o1, rows=self._rows())

    def compose(self, p):
        self.calls.append(("compose",))
        t = self.tex
        g0, g1 = (abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1) if p.giSource else (abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1)
        if self._rows()[1] <= self._rows()[0]:
            return
        O.compose(t[abi.TEX_DEPTH], t[abi.TEX_GBUFFER], t[g0], t[g1], p, out=t[abi.TEX_COMPOSE], rows=self._rows(),
                  scene=t[abi.TEX_DIRECT_LIGHT])
        if p.writeHistoryRGB:
            y0, y1 = self._rows()
            t[abi.TEX_COMPOSE_RGB][y0:y1] = t[abi.TEX_COMPOSE][y0:y1, :, :3]

    def final_compose(self, p):
        self.calls.append(("final", p.fogMode, p.isDebug))
        t = self.tex
        src = (abi.TEX_COMPOSE, abi.TEX_TEMPORAL0, abi.TEX_DENOISE_B0)[p.inputSource]
        O.final(t[abi.TEX_DEPTH], t[src], t[abi.TEX_DIRECT_LIGHT], p, out=t[abi.TEX_FINAL], rows=self._rows())

    def sync(self):
        pass

    def halo_violations(self):
        return 0
