"""Host-side mirror of the reference's operator interface for the hot path (Python flavour; the
Node flavour with the same surface lives in ../js).

Same class names, constructor signatures, option names, defaults and frame logic as the
reference's JS drivers; the device work each `render()` did through
`renderer.setRenderTarget(rt); renderer.render(scene, camera)` is one C-ABI call on a
`Context` (librfx_hip.so).  `scene` / `camera` are plain dumped-state objects (no three.js):

  camera  : rfx_amd.scene.Camera-like (projectionMatrix[Inverse], matrixWorld[Inverse], position,
            quaternion, near, far, isPerspectiveCamera) — updated by the caller every frame
  scene   : anything; only `scene.frame` (a dumped rfx_amd.scene.Frame) is read, by the
            raster shims (GBufferPass / VelocityDepthNormalPass) that stand where the
            reference rasterises
  renderer: an rfx_amd.context.Context (the device), passed to update()/render() like
            three's WebGLRenderer is.

Reference files mirrored (file:line in each class):
  src/ssgi/SSGIEffect.js, src/ssgi/SSGIOptions.js, src/ssgi/pass/SSGIPass.js,
  src/denoise/Denoiser.js, src/denoise/pass/PoissonDenoisePass.js,
  src/denoise/pass/DenoiserComposePass.js, src/temporal-reproject/TemporalReprojectPass.js,
  src/temporal-reproject/pass/VelocityDepthNormalPass.js, src/traa/TRAAEffect.js,
  src/utils/BlueNoiseUtils.js, src/utils/SceneUtils.js
"""
from __future__ import annotations

import math
import random

import numpy as np

from . import abi

# src/ssgi/SSGIOptions.js:26-48
defaultSSGIOptions = dict(
    mode="ssgi", distance=10, thickness=10, denoiseIterations=1, denoiseKernel=2, denoiseDiffuse=10, denoiseSpecular=10, radius=3,
    phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=50, roughnessPhi=50, specularPhi=50, envBlur=0.5, importanceSampling=True, steps=20,
    refineSteps=5, resolutionScale=1, missedRays=False, outputTexture=None)

# src/temporal-reproject/TemporalReprojectPass.js:17-32
defaultTemporalReprojectPassOptions = dict(
    dilation=False, fullAccumulate=False, neighborhoodClamp=False, neighborhoodClampRadius=1, neighborhoodClampIntensity=1, maxBlend=1,
    logTransform=False, depthDistance=2, worldDistance=4, reprojectSpecular=False, renderTarget=None, copyTextures=True,
    confidencePower=0.75, inputType="diffuse")

# src/denoise/pass/PoissonDenoisePass.js:16-24
defaultPoissonBlurOptions = dict(iterations=1, radius=3, phi=0.5, lumaPhi=5, depthPhi=2, normalPhi=3.25, inputType="diffuseSpecular")

# src/denoise/Denoiser.js:6-11
defaultDenoiserOptions = dict(denoiseMode="full", inputType="diffuseSpecular", gBufferPass=None, velocityDepthNormalPass=None)

_INPUT_TYPES = ["diffuseSpecular", "diffuse", "specular"]
# three.js texture `type` constants the path distinguishes (three/src/constants.js)
UnsignedByteType, FloatType, HalfFloatType = 1009, 1015, 1016


def _texture_type(texture) -> int:
    """`texture.type` of an input texture: a three-style object/dict carrying `type`, or a bare slot id / None for the
    FloatType targets the SSGI chain passes around (SSGIPass.js:24)."""
    t = getattr(texture, "type", None)
    if t is None and isinstance(texture, dict):
        t = texture.get("type")
    return FloatType if t is None else int(t)


# src/taa/TAAUtils.js:3 + src/temporal-reproject/utils/QuasirandomGenerator.js:12-27 (JS doubles)
def _generate_r2(count):
    g = 1.32471795724474602596090885447809
    a1, a2, base = 1.0 / g, 1.0 / (g * g), 1.1127756842787055
    return [[math.fmod(base + a1 * n, 1.0), math.fmod(base + a2 * n, 1.0)] for n in range(count)]


r2Sequence = [[a - 0.5, b - 0.5] for a, b in _generate_r2(256)]


def jitter(width, height, camera, frame, jitterScale=1):
    """src/taa/TAAUtils.js:5-11.  `camera.setViewOffset` is three's PerspectiveCamera method; the dumped-state camera of
    this host records the offset for the raster side (which renders the next dump with it) when it has one."""
    x, y = r2Sequence[frame % len(r2Sequence)]
    if hasattr(camera, "setViewOffset"):
        camera.setViewOffset(width, height, x * jitterScale, y * jitterScale, width, height)
    return x * jitterScale, y * jitterScale
HIGHEST_SIGNED_INT = 0x7FFFFFFF


class BlueNoiseIndex:
    """The `blueNoiseIndex` uniform getter of src/utils/BlueNoiseUtils.js:17-33: every READ (one
    per draw of the material) advances `(startIndex + idx + 1) % 0x7fffffff`; `startIndex` is
    random per material unless pinned for reproducible runs."""

    def __init__(self, start_index: int | None = None):
        self.start_index = int(start_index) if start_index is not None else int(math.floor(random.random() * HIGHEST_SIGNED_INT))
        self._idx = 0

    @property
    def value(self) -> int:
        self._idx = (self.start_index + self._idx + 1) % HIGHEST_SIGNED_INT
        return self._idx

    @value.setter
    def value(self, v: int):
        self._idx = int(v)


def didCameraMove(camera, last_position, last_quaternion) -> bool:
    """src/utils/SceneUtils.js:17-27."""
    p = np.asarray(camera.position, np.float64)
    if float(((p - last_position) ** 2).sum()) > 0.000001:
        return True
    q = np.asarray(getattr(camera, "quaternion", (0, 0, 0, 1)), np.float64)
    d = abs(float(np.clip(np.dot(q, last_quaternion), -1, 1)))
    return 2 * math.acos(d) > 0.001  # Quaternion.angleTo


def _upload_plane(renderer, tex, plane, static=False):
    """Hand one dumped full-frame plane to the device (the slot takes the band it holds).  The reference re-renders its raster passes
    every frame; so does this — every call uploads — unless the dump DECLARES itself unchanged (`scene.frame.static = True`): then a
    plane object that is already resident (same ndarray as last time) is not sent again.  Opt-in because identity says nothing about
    contents: a caller that refills a preallocated buffer in place (the normal per-frame dump loop) hands over the same object every
    frame with new texels in it."""
    cache = renderer.__dict__.setdefault("_resident_planes", {})
    if static == "resident" or (static and cache.get(tex) is plane):
        return  # "resident": the caller streamed the planes itself (Context.stage_frame / stage_flip)
    r0, n = renderer.held_rows(tex)
    if plane.shape[0] == n:  # the caller dumped exactly the band this tile holds
        renderer.upload(tex, plane, r0, n)
    else:
        renderer.upload(tex, plane[r0:r0 + n], r0, n)
    cache[tex] = plane


def _pack_planes(renderer, tex, aov, depth, pack, static=False):
    """Device-side packing of a dumped frame's attribute planes into `tex` (skipped only for a `static` frame's resident aov, like _upload_plane)."""
    cache = renderer.__dict__.setdefault("_resident_planes", {})
    if static and cache.get(tex) is aov:
        return
    r0, n = renderer.held_rows(tex)
    full = depth.shape[0] != n  # the caller dumped the whole frame: hand over the band this tile holds
    band = {k: (v[r0:r0 + n] if full else v) for k, v in aov.items()}
    pack(band, depth[r0:r0 + n] if full else depth, r0, n)
    cache[tex] = aov


class GBufferPass:
    """Stand-in for src/gbuffer/GBufferPass.js: the rasteriser is out of scope;
    `render` uploads the pre-dumped packed G-buffer + depth planes of `scene.frame`."""

    def __init__(self, scene, camera):
        self._scene, self._camera = scene, camera
        self.texture, self.depthTexture = abi.TEX_GBUFFER, abi.TEX_DEPTH

    def setSize(self, width, height):
        self.width, self.height = width, height

    def render(self, renderer):
        f = self._scene.frame
        _upload_plane(renderer, abi.TEX_DEPTH, f.depth, getattr(f, "static", False))
        if getattr(f, "gbuffer", None) is not None:
            _upload_plane(renderer, abi.TEX_GBUFFER, f.gbuffer, getattr(f, "static", False))
        else:  # an engine dump of UNPACKED attribute planes: the device packs them (rfx_pack_gbuffer = the pass's fragment epilogue)
            _pack_planes(renderer, abi.TEX_GBUFFER, f.aov, f.depth, renderer.pack_gbuffer, getattr(f, "static", False))

    def dispose(self):
        pass


class VelocityDepthNormalPass:
    """src/temporal-reproject/pass/VelocityDepthNormalPass.js:66 — kept as a loader shim: the
    class, its (scene, camera) signature and `texture`/`renderTarget` accessors survive, the
    raster work is replaced by uploading the dumped RGBA32F plane (format :186-188)."""

    def __init__(self, scene, camera):
        self._scene, self._camera = scene, camera
        self.renderTarget = self
        self.texture = abi.TEX_VELOCITY
        self.depthTexture = abi.TEX_VELOCITY
        self.lastVelocityTexture = None  # declared by the reference, never read by K2 (Appendix D-6)

    def setSize(self, width, height):
        self.width, self.height = width, height

    def render(self, renderer):
        f = self._scene.frame
        if getattr(f, "velocity", None) is not None:
            _upload_plane(renderer, abi.TEX_VELOCITY, f.velocity, getattr(f, "static", False))
        else:  # unpacked planes (uv-space velocity, world normal, depth): rfx_pack_velocity
            _pack_planes(renderer, abi.TEX_VELOCITY, f.aov, f.depth, renderer.pack_velocity, getattr(f, "static", False))

    def dispose(self):
        pass


class TemporalReprojectPass:
    """src/temporal-reproject/TemporalReprojectPass.js:38-225."""

    def __init__(self, scene, camera, velocityDepthNormalPass, texture, textureCount, options=None, half_store_rtz=True):
        self._scene, self._camera = scene, camera
        self.textureCount = textureCount
        o = dict(defaultTemporalReprojectPassOptions)
        o.update(options or {})
        self.options = o
        self.velocityDepthNormalPass = velocityDepthNormalPass
        self.frame = 0
        self.overrideAccumulatedTextures = []
        self.lastCameraTransform = dict(position=np.zeros(3), quaternion=np.array([0.0, 0, 0, 1]))
        # :63-68 the render target takes the TYPE of the input texture; :137-142 so does the framebuffer copy
        self.targetType = _texture_type(texture)
        if self.targetType not in (FloatType, HalfFloatType):
            raise ValueError("TemporalReprojectPass: input texture type %r — only FloatType / HalfFloatType targets are built" % (self.targetType,))
        it = _INPUT_TYPES.index(o["inputType"]) if o["inputType"] in _INPUT_TYPES else 1
        p = abi.TemporalParams(textureCount=textureCount, inputType=it, logTransform=1 if o["logTransform"] else 0,
                               confidencePower=float(o["confidencePower"]), neighborhoodClampIntensity=float(o["neighborhoodClampIntensity"]),
                               maxBlend=float(o["maxBlend"]), keepData=1.0, targetHalf=1 if self.targetType == HalfFloatType else 0,
                               halfStoreRTZ=1 if half_store_rtz else 0)
        for name in ("reprojectSpecular", "neighborhoodClamp"):  # :109-116 — arrays of (arrays of) bools; indices 0,1 matter
            v = o[name]
            v = list(v) if isinstance(v, (list, tuple)) else [v] * 2
            getattr(p, name)[:] = [1 if x else 0 for x in (v + v)[:2]]
        self.uniforms = p
        # :95-104 the ctor clones the current camera state into the prev* uniforms
        self._prev = abi.Camera.from_scene(camera)

    def setSize(self, width, height):
        self.width, self.height = width, height

    @property
    def texture(self):
        return abi.TEX_TEMPORAL0

    @property
    def framebufferTexture(self):
        """:137-142 — the slot the pass copies its target into when nothing overrides its history."""
        return abi.TEX_FBCOPY_F16 if self.targetType == HalfFloatType else abi.TEX_FBCOPY_F32

    def reset(self):
        self.uniforms.keepData = 0.0  # :158-160

    def render(self, renderer):
        self.frame = (self.frame + 1) % 4096
        cam = self._camera
        # :168-172,185-187 the pass draws with the UNJITTERED projection (view offset disabled while the uniforms are read)
        self.uniforms.camera = abi.Camera.from_scene(getattr(cam, "unjittered", cam))
        self.uniforms.prevCamera = self._prev
        moved = didCameraMove(cam, self.lastCameraTransform["position"], self.lastCameraTransform["quaternion"])
        self.uniforms.fullAccumulate = 1 if (self.options["fullAccumulate"] and not moved) else 0  # :178-180
        self.lastCameraTransform["position"] = np.asarray(cam.position, np.float64).copy()
        self.lastCameraTransform["quaternion"] = np.asarray(getattr(cam, "quaternion", (0, 0, 0, 1)), np.float64).copy()
        own_history = len(self.overrideAccumulatedTextures) == 0  # :148-151
        self.uniforms.historySource = 0 if not own_history else (1 if self.targetType == HalfFloatType else 2)
        renderer.temporal_reproject(self.uniforms)  # :192-193
        self.uniforms.keepData = 1.0  # :195
        if own_history:  # :197-201
            renderer.copy_framebuffer(self.framebufferTexture)
            hook = getattr(renderer, "after_copy_framebuffer", None)
            if hook:
                hook(self.framebufferTexture)  # multi-GPU: halo exchange of the history rows
        self._prev = abi.Camera.from_scene(getattr(cam, "unjittered", cam))  # :203-213

    def jitter(self, jitterScale=1):  # :216-220
        self.unjitter()
        return jitter(self.width, self.height, self._camera, self.frame, jitterScale)

    def unjitter(self):  # :222-224
        if hasattr(self._camera, "clearViewOffset"):
            self._camera.clearViewOffset()

    def dispose(self):
        pass


class PoissonDenoisePass:
    """src/denoise/pass/PoissonDenoisePass.js:26-152."""

    DefaultOptions = defaultPoissonBlurOptions

    def __init__(self, camera, textures, options=None, blue_noise_start=None, half_store_rtz=True):
        o = dict(defaultPoissonBlurOptions)
        o.update(options or {})
        self.iterations = defaultPoissonBlurOptions["iterations"]
        self.textures = textures
        spec = [0, 1]
        if o["inputType"] == "diffuse":
            spec = [0, 0]
        if o["inputType"] == "specular":
            spec = [1, 1]
        tc = 2 if o["inputType"] == "diffuseSpecular" else 1
        d = defaultPoissonBlurOptions
        # :48-66 — roughnessPhi/specularPhi start `undefined` until SSGIEffect's setters write them
        self.uniforms = abi.DenoiseParams(radius=d["radius"], phi=d["phi"], lumaPhi=d["lumaPhi"], depthPhi=float(o["depthPhi"]),
                                          normalPhi=float(o["normalPhi"]), roughnessPhi=float("nan"), specularPhi=float("nan"),
                                          textureCount=tc, halfStoreRTZ=1 if half_store_rtz else 0)
        self.uniforms.isTextureSpecular[:] = spec
        self.blueNoiseIndex = BlueNoiseIndex(blue_noise_start)

    def setSize(self, width, height):
        self.width, self.height = width, height

    @property
    def texture(self):
        return (abi.TEX_DENOISE_B0, abi.TEX_DENOISE_B1)

    def render(self, renderer):
        for i in range(2 * int(self.iterations)):  # :135-149
            horizontal = i % 2 == 0
            self.uniforms.inputIsTemporal = 1 if i == 0 else 0
            self.uniforms.writeToB = 0 if horizontal else 1
            self.uniforms.blueNoiseIndex = self.blueNoiseIndex.value
            renderer.poisson_denoise(self.uniforms)
            hook = getattr(renderer, "after_denoise_pass", None)
            if hook:
                hook(i, self.uniforms)  # multi-GPU: halo exchange of the target just written

    def dispose(self):
        pass


class DenoiserComposePass:
    """src/denoise/pass/DenoiserComposePass.js:8-136."""

    def __init__(self, camera, textures, gBufferTexture, depthTexture, options=None):
        self._camera = camera
        o = options or {}
        it = _INPUT_TYPES.index(o.get("inputType", "diffuseSpecular")) if o.get("inputType", "diffuseSpecular") in _INPUT_TYPES else 0
        # Denoiser.js:53 composerInputTextures: the denoise pass's targets, or (denoiseMode "full_temporal") K2's own
        src = 1 if tuple(textures)[0] == abi.TEX_TEMPORAL0 else 0
        self.uniforms = abi.ComposeParams(inputType=it, giSource=src)

    def setSize(self, width, height):
        self.width, self.height = width, height

    @property
    def texture(self):
        return abi.TEX_COMPOSE

    def render(self, renderer):
        self.uniforms.camera = abi.Camera.from_scene(self._camera)
        # a row-tiled renderer all-gathers the part of this target K1 reads next frame (.rgb) as 12-byte texels
        self.uniforms.writeHistoryRGB = 1 if getattr(renderer, "gather_history_rgb", False) else 0
        renderer.compose(self.uniforms)

    def dispose(self):
        pass


class Denoiser:
    """src/denoise/Denoiser.js:16-108 — temporal + spatial + compose chain."""

    def __init__(self, scene, camera, texture, options=None, blue_noise_start=None, half_store_rtz=True):
        o = dict(defaultDenoiserOptions)
        o.update(options or {})
        self.options = o
        self.velocityDepthNormalPass = o["velocityDepthNormalPass"] or VelocityDepthNormalPass(scene, camera)
        self.isOwnVelocityDepthNormalPass = not o["velocityDepthNormalPass"]
        textureCount = 2 if o["inputType"] == "diffuseSpecular" else 1
        topt = dict(fullAccumulate=True, logTransform=True, copyTextures=not o.get("denoise"), reprojectSpecular=[False, True],
                    neighborhoodClamp=[True, True], neighborhoodClampRadius=2, neighborhoodClampIntensity=0.5)
        topt.update({k: v for k, v in o.items() if k in defaultTemporalReprojectPassOptions})
        self.temporalReprojectPass = TemporalReprojectPass(scene, camera, self.velocityDepthNormalPass, texture, textureCount, topt)
        self.denoisePass = None
        self.denoiserComposePass = None
        if o["denoiseMode"] in ("full", "denoised"):
            popt = {k: v for k, v in o.items() if k in defaultPoissonBlurOptions}
            self.denoisePass = PoissonDenoisePass(camera, (abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1), popt, blue_noise_start, half_store_rtz)
            self.temporalReprojectPass.overrideAccumulatedTextures = list(self.denoisePass.texture)
        if o["denoiseMode"] not in ("full", "full_temporal", "denoised", "temporal"):
            raise ValueError("denoiseMode %r" % (o["denoiseMode"],))
        textures = (abi.TEX_TEMPORAL0, abi.TEX_TEMPORAL1)[:textureCount]
        composerInputTextures = self.denoisePass.texture if self.denoisePass else textures  # :53
        if o["denoiseMode"].startswith("full"):
            self.denoiserComposePass = DenoiserComposePass(camera, composerInputTextures, abi.TEX_GBUFFER, abi.TEX_DEPTH, o)

    @property
    def texture(self):
        m = self.options["denoiseMode"]
        if m in ("full", "full_temporal"):
            return self.denoiserComposePass.texture
        if m == "denoised":
            return self.denoisePass.texture
        return self.temporalReprojectPass.texture

    def reset(self):
        self.temporalReprojectPass.reset()

    def setSize(self, width, height):
        for p in (self.velocityDepthNormalPass, self.temporalReprojectPass, self.denoisePass, self.denoiserComposePass):
            if p:
                p.setSize(width, height)

    def dispose(self):
        pass

    def render(self, renderer, inputBuffer=None):
        if self.isOwnVelocityDepthNormalPass:
            self.velocityDepthNormalPass.render(renderer)
        self.temporalReprojectPass.render(renderer)
        hook = getattr(renderer, "after_temporal_pass", None)
        if hook:
            hook()
        if self.denoisePass:
            self.denoisePass.render(renderer)
        if self.denoiserComposePass:
            self.denoiserComposePass.render(renderer)
            hook = getattr(renderer, "after_compose_pass", None)
            if hook:
                hook()


class SSGIPass:
    """src/ssgi/pass/SSGIPass.js:7-96."""

    def __init__(self, ssgiEffect, options, blue_noise_start=None):
        self.ssgiEffect = ssgiEffect
        self._scene, self._camera = ssgiEffect._scene, ssgiEffect._camera
        self.frame = 21483
        mode = ["ssgi", "ssr"].index(options["mode"])
        self.uniforms = abi.SsgiParams(steps=20, refineSteps=5, mode=mode, useDirectLight=0, missedRays=0, importanceSampling=0)
        self.blueNoiseIndex = BlueNoiseIndex(blue_noise_start)
        self.gBufferPass = GBufferPass(self._scene, self._camera)

    @property
    def texture(self):
        return abi.TEX_SSGI

    def setSize(self, width, height):
        # :52-57 the pass's render target (and its `resolution` uniform) is width*resolutionScale x height*resolutionScale
        s = float(self.ssgiEffect._options["resolutionScale"])
        self.renderTargetSize = (width * s, height * s) if width is not None else (None, None)
        self.uniforms.resolutionScale = s
        self.gBufferPass.setSize(width, height)

    def render(self, renderer):
        self.frame = (self.frame + 1) % 4096
        self.gBufferPass.render(renderer)
        self.uniforms.camera = abi.Camera.from_scene(self._camera)
        self.uniforms.blueNoiseIndex = self.blueNoiseIndex.value
        # :89 accumulatedTexture = ssgiEffect.denoiser.texture: K4's target, K2's texture[0] ("temporal"), or — "denoised", where the
        # getter returns the ARRAY of K3's targets — what three binds for a non-texture value: its empty texture (zeros)
        t = self.ssgiEffect.denoiser.texture
        self.uniforms.historySource = 2 if isinstance(t, (tuple, list)) else (1 if t == abi.TEX_TEMPORAL0 else 0)
        if self.uniforms.historySource == 0 and getattr(renderer, "gather_history_rgb", False):
            self.uniforms.historySource = 3  # the same values from RFX_TEX_COMPOSE_RGB (tiling.py)
        if getattr(renderer, "overlap_history_gather", False):
            # row-tiled run: last frame's composed GI is still being all-gathered; only the shading half of the draw reads it
            renderer.ssgi_trace(self.uniforms)
            renderer.before_ssgi_shade()
            renderer.ssgi_shade(self.uniforms)
        else:
            renderer.ssgi_march(self.uniforms)  # :93-94

    def dispose(self):
        pass


NearestFilter, LinearFilter, LinearMipmapLinearFilter = 1003, 1006, 1008


class CubeToEquirectEnvPass:
    """src/ssgi/pass/CubeToEquirectEnvPass.js: `scene.environment` given as a CubeTexture is rendered into an equirectangular FloatType
    target (one textureCube lookup per texel), read back, and continues as a DataTexture.  The cube as dumped state: an object/dict with
    `isCubeTexture`, `faces` ((6, S, S, 4) float32 linear values, +X -X +Y -Y +Z -Z, row j = t as uploaded), and the two sampler fields
    the lookup depends on: `minFilter` (default LinearMipmapLinearFilter, three's Texture default) and `generateMipmaps` (default True)."""

    def generateEquirectEnvMap(self, renderer, cubeMap, width=None, height=None, maxWidth=4096):
        get = (lambda k, d=None: cubeMap.get(k, d)) if isinstance(cubeMap, dict) else (lambda k, d=None: getattr(cubeMap, k, d))
        faces = np.ascontiguousarray(get("faces"), np.float32)
        if width is None and height is None:  # :62-69
            w = faces.shape[1]
            width = int(2 ** math.ceil(math.log2(2 * w * 3 ** 0.5)))
            height = int(2 ** math.ceil(math.log2(w * 3 ** 0.5)))
        if width > maxWidth:  # :71-74
            width, height = maxWidth, maxWidth // 2
        min_filter = get("minFilter", LinearMipmapLinearFilter)
        if min_filter == LinearMipmapLinearFilter:
            if not get("generateMipmaps", True):
                raise ValueError("CubeToEquirectEnvPass: a LinearMipmapLinearFilter cube texture without generated mipmaps is incomplete (samples black)")
            mips = True
        elif min_filter == LinearFilter:
            mips = False
        else:
            raise NotImplementedError("CubeToEquirectEnvPass: cube minFilter %r — LinearFilter and LinearMipmapLinearFilter are built" % (min_filter,))
        data = renderer.cube_to_equirect(faces, width, height, generate_mipmaps=mips)  # render + readRenderTargetPixels :76-85
        # :87-97 DataTexture(pixelBuffer, width, height, RGBAFormat, FloatType), ClampToEdge, EquirectangularReflectionMapping
        return dict(data=data, type=FloatType, generateMipmaps=False, isCubeTexture=False)

    def dispose(self):
        pass


class SSGIEffect:
    """src/ssgi/SSGIEffect.js:27-439 — owns SSGIPass + Denoiser, reactive options."""

    DefaultOptions = defaultSSGIOptions

    def __init__(self, composer, scene, camera, options=None, seeds=None, half_store_rtz=True):
        options = dict(defaultSSGIOptions, **(options or {}))
        self._scene, self._camera, self.composer = scene, camera, composer
        self.isUsingRenderPass = True
        if options["mode"] == "ssr":  # :70-73
            options["reprojectSpecular"] = True
            options["neighborhoodClamp"] = True
            options["inputType"] = "specular"
        elif options["mode"] == "ssgi":  # :74-77
            options["reprojectSpecular"] = [False, True]
            options["neighborhoodClamp"] = [False, True]
        preset = options.get("preset")
        if isinstance(preset, str):  # :79-99 (the second `case "medium"` is unreachable)
            if preset == "low":
                options.update(steps=10, refineSteps=2, denoiseMode="full_temporal")
            elif preset == "medium":
                options.update(steps=20, refineSteps=4, denoiseMode="full")
        seeds = seeds or {}
        self._options = options
        self._half_store_rtz = half_store_rtz
        self.ssgiPass = SSGIPass(self, options, seeds.get("ssgi"))
        dopt = dict(gBufferPass=self.ssgiPass.gBufferPass, velocityDepthNormalPass=options.get("velocityDepthNormalPass"))
        dopt.update(options)
        self.denoiser = Denoiser(scene, camera, self.ssgiPass.texture, dopt, seeds.get("denoise"), half_store_rtz)
        self.lastSize = dict(width=options.get("width"), height=options.get("height"), resolutionScale=options["resolutionScale"])
        self.setSize(options.get("width"), options.get("height"))
        self.uniforms = abi.FinalParams(camera=abi.Camera.from_scene(camera), isDebug=0, fogMode=0)  # FinalSSGIMaterial :47-66
        self._reactive = False
        for key in list(options.keys()):  # makeOptionsReactive :157-268 — apply every option once
            self._apply(key, options[key])
        self._reactive = True
        self.outputTexture = self.denoiser.texture
        # the composer's RenderPass has run before update(): direct light is available (:124-138,143-151)
        self.updateUsingRenderPass()

    # reactive option surface: effect.<option> reads/writes go through _apply like the JS setters
    def __getattr__(self, key):
        o = self.__dict__.get("_options")
        if o is not None and key in o:
            return o[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        o = self.__dict__.get("_options")
        if o is not None and key in o and self.__dict__.get("_reactive"):
            if o[key] == value:
                return
            o[key] = value
            self._apply(key, value)
        else:
            object.__setattr__(self, key, value)

    def _apply(self, key, value):
        dp = self.denoiser.denoisePass
        if key == "denoiseIterations":
            if dp:
                dp.iterations = value
        elif key in ("radius", "phi", "lumaPhi", "depthPhi", "normalPhi", "roughnessPhi", "specularPhi"):
            if dp:
                setattr(dp.uniforms, key, float(value))
                self.reset()
        elif key == "resolutionScale":
            self.setSize(self.lastSize["width"], self.lastSize["height"])
            self.reset()
        elif key in ("steps", "refineSteps"):
            setattr(self.ssgiPass.uniforms, key, int(value))
            self.reset()
        elif key == "importanceSampling":
            # only effective with an env map (SSGIEffect.js:344-354): keepEnvMapUpdated re-reads the option when the environment is (re)set
            object.__setattr__(self, "_env_uuid", None)
            self.reset()
        elif key == "missedRays":
            self.ssgiPass.uniforms.missedRays = 1 if value else 0
            self.reset()
        elif key == "distance":
            self.ssgiPass.uniforms.rayDistance = float(value)
            self.reset()
        elif key in ("thickness", "envBlur"):  # default branch: a uniform of the same name
            setattr(self.ssgiPass.uniforms, key, float(value))
            self.reset()
        # denoiseKernel / denoiseDiffuse / denoiseSpecular: accepted, no consumer (Appendix D-3)

    def updateUsingRenderPass(self):
        self.ssgiPass.uniforms.useDirectLight = 1 if self.isUsingRenderPass else 0

    def reset(self):
        self.denoiser.reset()

    def setSize(self, width, height, force=False):
        if width is None and height is None:
            return
        self.ssgiPass.setSize(width, height)
        self.denoiser.setSize(width, height)
        # K2 samples the pass's (possibly smaller) texture NEAREST at full-resolution vUv: the device needs its size
        tw, th = self.ssgiPass.renderTargetSize
        tu = self.denoiser.temporalReprojectPass.uniforms
        tu.inputWidth, tu.inputHeight = (0, 0) if (tw is None or self._options["resolutionScale"] == 1) else (int(tw), int(th))
        self.lastSize = dict(width=width, height=height, resolutionScale=self._options["resolutionScale"])

    @property
    def depthTexture(self):
        return self.ssgiPass.gBufferPass.depthTexture

    def initialize(self, renderer=None, *args):
        pass

    def dispose(self):
        self.ssgiPass.dispose()
        self.denoiser.dispose()

    def keepEnvMapUpdated(self, renderer):
        """:309-362.  `scene.environment`: None, or an equirectangular HDR map as dumped state — an object/dict with `data`
        ((H, W, 4) float32, row 0 = bottom) and optionally `type` (HalfFloatType, what RGBELoader yields and the default, or
        FloatType).  The effect turns its mipmaps on (:323-328): here the device builds the chain (rfx_set_environment)."""
        env = getattr(self._scene, "environment", None)
        u = self.ssgiPass.uniforms
        if env is not None:
            if self.__dict__.get("_env_uuid") is not env:
                get = (lambda k, d=None: env.get(k, d)) if isinstance(env, dict) else (lambda k, d=None: getattr(env, k, d))
                if get("isCubeTexture"):  # :316-321 convert it to an equirectangular texture so the pass can sample it and use MIS
                    if self.__dict__.get("cubeToEquirectEnvPass") is None:
                        object.__setattr__(self, "cubeToEquirectEnvPass", CubeToEquirectEnvPass())
                    converted = self.cubeToEquirectEnvPass.generateEquirectEnvMap(renderer, env)
                    get = lambda k, d=None: converted.get(k, d)  # noqa: E731
                t = get("type", HalfFloatType)
                data = np.ascontiguousarray(get("data"), np.float32)
                renderer.set_environment(data, half_float_type=(t == HalfFloatType), half_store_rtz=self._half_store_rtz)
                u.importanceSampling = 0
                if self._options["importanceSampling"]:  # :348-351 EquirectHdrInfoUniform.updateFrom, then the define
                    from .envmap import build_importance
                    texels = data.astype(np.float16).astype(np.float32) if t == HalfFloatType else data  # the worker's fromHalfFloat view
                    # `data` is in GL row order (row 0 = bottom, as sampled).  With texture.flipY the reference's DataTexture array is the
                    # other way up and the worker "un-flips" it (its own, lossy way): hand it what it would have seen
                    flip = bool(get("flipY", False))
                    renderer.set_environment_importance(*build_importance(texels[::-1] if flip else texels, flip))
                    u.importanceSampling = 1
                object.__setattr__(self, "_env_uuid", env)
                u.useEnvMap = 1  # defines.USE_ENVMAP :344
                self.reset()     # :356
        elif u.useEnvMap:  # :361-366
            u.useEnvMap = u.importanceSampling = 0
            renderer.set_environment(None)
            object.__setattr__(self, "_env_uuid", None)

    def update(self, renderer, inputBuffer=None):
        """:372-436.  `inputBuffer` = the composer's input buffer (direct lighting) as an (H,W,4)
        float32 array covering the frame, or None to take `scene.frame.direct`."""
        self.keepEnvMapUpdated(renderer)
        direct = inputBuffer if inputBuffer is not None else self._scene.frame.direct
        _upload_plane(renderer, abi.TEX_DIRECT_LIGHT, direct, getattr(self._scene.frame, "static", False))
        self.ssgiPass.render(renderer)
        self.denoiser.render(renderer, inputBuffer)
        # :400-417 the effect's own uniforms: inputTexture = the denoiser's texture, sceneTexture = the input buffer, fog from the scene
        u = self.uniforms
        out = self.denoiser.texture  # :139,402 inputTexture = outputTexture[0] ?? outputTexture
        out = out[0] if isinstance(out, (tuple, list)) else out
        u.inputSource = {abi.TEX_COMPOSE: 0, abi.TEX_TEMPORAL0: 1, abi.TEX_DENOISE_B0: 2}[out]
        fog = getattr(self._scene, "fog", None)
        u.fogMode = 0 if fog is None else (2 if getattr(fog, "isFogExp2", False) else 1)
        if fog is not None:
            u.fogColor[:] = [float(x) for x in fog.color]
            u.fogNear, u.fogFar = float(getattr(fog, "near", 0.0) or 0.0), float(getattr(fog, "far", 0.0) or 0.0)
            u.fogDensity = float(getattr(fog, "density", 0.0) or 0.0)
            u.camera = abi.Camera.from_scene(self._camera)

    def mainImage(self, renderer):
        """The effect's own fragment (src/ssgi/shader/ssgi_compose.frag:20-45), which postprocessing's EffectPass runs after
        update(): scene colour on background texels, composed GI (+ fog) elsewhere, alpha 1 -> RFX_TEX_FINAL."""
        renderer.final_compose(self.uniforms)
        return abi.TEX_FINAL


class SSREffect(SSGIEffect):
    """src/ssgi/SSREffect.js:3-9."""

    def __init__(self, composer, scene, camera, options=None, **kw):
        options = dict(options or {})
        options["mode"] = "ssr"
        super().__init__(composer, scene, camera, options, **kw)


class TRAAEffect:
    """src/traa/TRAAEffect.js:10-78.  K2 alone, on the composer's input buffer: one texture, inputType "diffuse",
    history = the pass's own framebuffer copy.  `inputBuffer` is the composer buffer as dumped state: an object (or
    dict) with `texture` (anything carrying three's `type`: HalfFloatType for `frameBufferType: HalfFloatType`,
    example/main.js:173, or FloatType), `width`, `height` and `data` — the H x W x 4 float32 scene colour, which
    stands where the reference's previous pass rendered into the buffer."""

    DefaultOptions = defaultTemporalReprojectPassOptions

    def __init__(self, scene, camera, velocityDepthNormalPass, options=None, half_store_rtz=True):
        self._scene, self._camera = scene, camera
        self.velocityDepthNormalPass = velocityDepthNormalPass
        o = dict(options or defaultTemporalReprojectPassOptions)
        o.update(maxBlend=0.9, neighborhoodClamp=True, neighborhoodClampIntensity=1, neighborhoodClampRadius=1, logTransform=True,
                 confidencePower=4)  # :21-31
        self.options = dict(defaultTemporalReprojectPassOptions, **o)
        self.temporalReprojectPass = None
        self._half_store_rtz = half_store_rtz
        self.uniforms = {"accumulatedTexture": None}
        self.unjitteredProjectionMatrix = None

    def reset(self):
        self.temporalReprojectPass.reset()

    def setSize(self, width, height):
        if self.temporalReprojectPass:
            self.temporalReprojectPass.setSize(width, height)

    def dispose(self):
        if self.temporalReprojectPass:
            self.temporalReprojectPass.dispose()

    @staticmethod
    def _field(obj, name):
        return obj[name] if isinstance(obj, dict) else getattr(obj, name)

    def temporal_params(self, texture=None):
        """The K2 launch parameters this effect draws with (textureCount 1, inputType DIFFUSE)."""
        if self.temporalReprojectPass is None:
            self.temporalReprojectPass = TemporalReprojectPass(self._scene, self._camera, self.velocityDepthNormalPass, texture, 1, self.options,
                                                               half_store_rtz=self._half_store_rtz)
        return self.temporalReprojectPass.uniforms

    def update(self, renderer, inputBuffer):
        if self.temporalReprojectPass is None:  # :53-66
            self.temporal_params(self._field(inputBuffer, "texture"))
            self.temporalReprojectPass.setSize(self._field(inputBuffer, "width"), self._field(inputBuffer, "height"))
            self.uniforms["accumulatedTexture"] = self.temporalReprojectPass.texture
        # the raster shims stand where the composer's earlier passes ran: velocity/depth/normal plane and the input buffer
        self.velocityDepthNormalPass.render(renderer)
        data = self._field(inputBuffer, "data")
        if self.temporalReprojectPass.targetType == HalfFloatType:
            # a HalfFloatType buffer holds half-precision texels: state that on the way in (exact for a real dump of one)
            cache = self.__dict__.setdefault("_half_cache", [None, None])
            if cache[0] is not data:
                cache[0], cache[1] = data, np.asarray(data, np.float32).astype(np.float16).astype(np.float32)
            data = cache[1]
        _upload_plane(renderer, abi.TEX_SSGI, data, bool(inputBuffer.get("static", False) if isinstance(inputBuffer, dict) else getattr(inputBuffer, "static", False)))  # K2's `inputTexture` (:118)
        self.temporalReprojectPass.unjitter()  # :68-73
        self.unjitteredProjectionMatrix = np.array(self._camera.projectionMatrix, np.float32).copy()
        self.temporalReprojectPass.jitter()
        self.temporalReprojectPass.render(renderer)  # :75

    def output(self, renderer, row0=None, rows=None):
        """traa_compose.frag (src/traa/shader/traa_compose.frag:3-7): outputColor = vec4(accumulatedTexel.rgb, 1.)."""
        t = renderer.download(self.uniforms["accumulatedTexture"], row0, rows).copy()
        t[..., 3] = 1.0
        return t
