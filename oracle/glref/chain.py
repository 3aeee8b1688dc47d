"""Run the reference's OWN GLSL (read from /root/reference at run time) on CPU llvmpipe.

TEST INFRASTRUCTURE ONLY — the product never imports this (see oracle/README.md).

This is the strongest oracle available for the path: the three.js/postprocessing runtime is
not installable here (no network, Node 12), but the fragment shaders — where ALL of the
arithmetic lives — compile unmodified on Mesa llvmpipe.  This module

  1. assembles each pass's final fragment source exactly the way the reference's JS does
     (string surgery cited per function), with the three.js prefix of SURVEY.md Appendix E and
     the two restated three.js chunks `<packing>` / `<common>`;
  2. drives the passes with the uniform values the reference's JS drivers would set
     (SSGIPass.js:68-95, TemporalReprojectPass.js:162-214, PoissonDenoisePass.js:135-149,
     DenoiserComposePass.js:129-135), including ping-pong, history wiring, keepData and the
     "discard keeps previous contents" behaviour;
  3. returns every stage's render target as numpy arrays.

It is used (a) here in the build container to generate tests/golden/*.npz
(tests/golden/make_golden.py) and to validate the C restatement oracle/rfx_oracle.c, and
(b) optionally on the GPU box as the "reference" CPU baseline, from the assembled shader files
that `make -C oracle ref` drops into oracle/_ref/shaders/ (build products, git-ignored).
"""
from __future__ import annotations

import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_OUT = os.path.join(HERE, "..", "_ref")
REFERENCE_SRC = "/root/reference/src/"

FMT_RGBA32F, FMT_RGBA16F, FMT_R32F, FMT_RGBA8 = 0, 1, 2, 3

# ------------------------------------------------------------------ three.js chunks (restated)
# three@0.151.3 is an un-vendored peer dependency (package.json:52-55).  Only these pieces of
# its ShaderChunks are reached by the path:
CHUNK_COMMON = """
#define PI 3.141592653589793
#define PI2 6.283185307179586
#define PI_HALF 1.5707963267948966
#define RECIPROCAL_PI 0.3183098861837907
#define RECIPROCAL_PI2 0.15915494309189535
#define EPSILON 1e-6
#ifndef saturate
#define saturate( a ) clamp( a, 0.0, 1.0 )
#endif
"""
CHUNK_PACKING = """
float viewZToOrthographicDepth( const in float viewZ, const in float near, const in float far ) { return ( viewZ + near ) / ( near - far ); }
float orthographicDepthToViewZ( const in float depth, const in float near, const in float far ) { return depth * ( near - far ) - near; }
float viewZToPerspectiveDepth( const in float viewZ, const in float near, const in float far ) { return ( ( near + viewZ ) * far ) / ( ( far - near ) * viewZ ); }
float perspectiveDepthToViewZ( const in float depth, const in float near, const in float far ) { return ( near * far ) / ( ( far - near ) * depth - far ); }
"""


def _rd(rel):
    with open(REFERENCE_SRC + rel, encoding="utf-8-sig") as f:
        return f.read()


def unroll_loops(s: str) -> str:
    """src/ssgi/utils/Utils.js:71-89 (three's WebGLProgram loop unroller)."""
    pat = re.compile(
        r"#pragma unroll_loop_start\s+for\s*\(\s*int\s+i\s*=\s*(\d+)\s*;\s*i\s*<\s*(\d+)\s*;\s*i\s*\+\+\s*\)\s*{([\s\S]+?)}\s+#pragma unroll_loop_end")

    def rep(m):
        out = ""
        for i in range(int(m.group(1)), int(m.group(2))):
            out += re.sub(r"\[\s*i\s*\]", "[ %d ]" % i, m.group(3)).replace("UNROLLED_LOOP_INDEX", str(i))
        return out

    return pat.sub(rep, s)


def three_prefix(defines: dict, glsl3: bool) -> str:
    """WebGLProgram fragment prefix for a non-raw ShaderMaterial on WebGL2."""
    p = "#version 300 es\n#define varying in\n"
    if not glsl3:
        p += "layout(location = 0) out highp vec4 pc_fragColor;\n#define gl_FragColor pc_fragColor\n"
    p += ("#define texture2D texture\n#define textureCube texture\n"
          "precision highp float;\nprecision highp int;\nprecision highp sampler2D;\n")
    for k, v in defines.items():
        p += "#define %s %s\n" % (k, v)
    p += "uniform mat4 viewMatrix;\nuniform vec3 cameraPosition;\nuniform bool isOrthographic;\n"
    return p


def _with_blue_noise(src: str) -> str:
    """src/utils/BlueNoiseUtils.js:35 — inserted after the first `uniform vec2 resolution;`."""
    return src.replace("uniform vec2 resolution;", "uniform vec2 resolution;\n" + _rd("utils/shader/blue_noise.glsl"), 1)


def assemble_ssgi(steps=20, refine_steps=5, mode=0, use_direct_light=True, missed_rays=False, use_envmap=False, perspective=True,
                  importance_sampling=False) -> str:
    """SSGIMaterial.js:44-56 + SSGIPass.js:38-40 + SSGIEffect.js:143-151,203-221."""
    gb = _rd("gbuffer/shader/gbuffer_packing.glsl")
    s = (_rd("ssgi/shader/ssgi.frag").replace("#include <ssgi_utils>", _rd("ssgi/shader/ssgi_utils.frag"))
         .replace("#include <gbuffer_packing>", gb).replace("#include <packing>", CHUNK_PACKING))
    s = _with_blue_noise(s)
    d = {"steps": int(steps), "refineSteps": int(refine_steps), "CUBEUV_TEXEL_WIDTH": 0, "CUBEUV_TEXEL_HEIGHT": 0,
         "CUBEUV_MAX_MIP": 0, "vWorldPosition": "worldPos", "mode": int(mode)}
    if perspective:  # SSGIPass.js:38
        d["PERSPECTIVE_CAMERA"] = ""
    if use_direct_light:
        d["useDirectLight"] = ""
    if missed_rays:
        d["missedRays"] = ""
    if use_envmap:  # SSGIEffect.js:344
        d["USE_ENVMAP"] = ""
    if importance_sampling:  # :348-351, once EquirectHdrInfoUniform.updateFrom has resolved
        d["importanceSampling"] = ""
    return three_prefix(d, False) + s


def assemble_temporal(texture_count=2, input_type=0, confidence_power=0.75, reproject_specular=(False, True),
                      neighborhood_clamp=(False, True), log_transform=True, neighborhood_clamp_radius=2, perspective=True) -> str:
    """TemporalReprojectMaterial.js:11-41 + TemporalReprojectPass.js:76-117."""
    gb = _rd("gbuffer/shader/gbuffer_packing.glsl")
    s = (_rd("temporal-reproject/shader/temporal_reproject.frag")
         .replace("#include <reproject>", _rd("temporal-reproject/shader/reproject.frag"))
         .replace("#include <gbuffer_packing>", gb))
    d = "".join("uniform sampler2D accumulatedTexture%d;\nlayout(location = %d) out vec4 gOutput%d;\n" % (i, i, i)
                for i in range(texture_count))
    s = unroll_loops(d + s.replace("textureCount", str(texture_count)))
    s = re.sub(r"accumulatedTexture\[\s*(\d+)\s*\]", r"accumulatedTexture\1", s)
    s = re.sub(r"gOutput\[\s*(\d+)\s*\]", r"gOutput\1", s)
    s = s.replace("#include <packing>", CHUNK_PACKING)

    def boolarr(v):
        # TemporalReprojectPass.js:109-116: `typeof value !== "array"` is always true -> the
        # option (already an array) is wrapped in Array(textureCount).fill(value) and joined.
        v = list(v) if isinstance(v, (list, tuple)) else [v]
        flat = [("true" if x else "false") for _ in range(texture_count) for x in v]
        if len(v) > 1:  # JS join of nested arrays: "false,true, false,true"
            return "bool[](" + ", ".join(",".join("true" if x else "false" for x in v) for _ in range(texture_count)) + ")"
        return "bool[](" + ", ".join(flat) + ")"

    defines = {"textureCount": texture_count, "neighborhoodClampRadius": int(neighborhood_clamp_radius),
               "depthDistance": "2.0000", "worldDistance": "4.0000", "inputType": int(input_type)}
    if perspective:  # TemporalReprojectPass.js:82
        defines["PERSPECTIVE_CAMERA"] = ""
    if log_transform:
        defines["logTransform"] = ""
    # (:79 first sets neighborhoodClamp as a flag; :115 overwrites it with the array form below)
    defines["reprojectSpecular"] = boolarr(reproject_specular)
    defines["neighborhoodClamp"] = boolarr(neighborhood_clamp)
    defines["confidencePower"] = _to_precision5(confidence_power)
    return three_prefix(defines, True) + s


def assemble_traa() -> str:
    """TRAAEffect.js:21-33 over defaultTemporalReprojectPassOptions: one texture, inputType "diffuse", logTransform,
    confidencePower 4, reprojectSpecular false, neighborhoodClamp true, neighborhoodClampRadius 1 (a define the shader never reads)."""
    return assemble_temporal(texture_count=1, input_type=1, confidence_power=4, reproject_specular=False, neighborhood_clamp=True,
                             log_transform=True, neighborhood_clamp_radius=1)


def _to_precision5(x: float) -> str:
    """JS Number.prototype.toPrecision(5) for the values used (0.75 -> '0.75000', 4 -> '4.0000')."""
    s = "%.5g" % x
    if "e" in s:
        return s
    digits = len(s.replace("-", "").replace(".", "").lstrip("0"))
    if "." not in s:
        s += "."
    return s + "0" * max(0, 5 - digits)


def assemble_denoise(texture_count=2, is_texture_specular=(False, True)) -> str:
    """PoissonDenoisePass.js:14,43-71,108-117."""
    gb = _rd("gbuffer/shader/gbuffer_packing.glsl")
    s = _rd("denoise/shader/poisson_denoise.frag").replace("#include <gbuffer_packing>", gb).replace("textureCount", str(texture_count))
    s = unroll_loops(s).replace("#include <common>", CHUNK_COMMON)
    s = _with_blue_noise(s)
    d = {"isTextureSpecular": "bool[2](" + ",".join("true" if x else "false" for x in is_texture_specular) + ")",
         "GBUFFER_TEXTURE": ""}
    return three_prefix(d, True) + s


def assemble_compose(input_type=0, perspective=True) -> str:
    """DenoiserComposePass.js:35-110 (inline template literal)."""
    js = _rd("denoise/pass/DenoiserComposePass.js")
    m = re.search(r"fragmentShader:\s*/\* glsl \*/\s*`([\s\S]*?)`,\s*vertexShader", js)
    s = m.group(1)
    s = s.replace("${gbuffer_packing}", _rd("gbuffer/shader/gbuffer_packing.glsl"))
    s = s.replace("${ssgi_poisson_compose_functions}", _rd("denoise/shader/denoiser_compose_functions.glsl"))
    s = s.replace("#include <common>", CHUNK_COMMON).replace("#include <packing>", CHUNK_PACKING)
    d = {"inputType": int(input_type)}
    if perspective:  # DenoiserComposePass.js:110
        d["PERSPECTIVE_CAMERA"] = ""
    return three_prefix(d, False) + s


# three@0.151 ShaderChunk.fog_pars_fragment / fog_fragment (un-vendored dependency, restated: SURVEY.md Appendix H)
CHUNK_FOG_PARS = """
#ifdef USE_FOG
	uniform vec3 fogColor;
	varying float vFogDepth;
	#ifdef FOG_EXP2
		uniform float fogDensity;
	#else
		uniform float fogNear;
		uniform float fogFar;
	#endif
#endif
"""
CHUNK_FOG = """
#ifdef USE_FOG
	#ifdef FOG_EXP2
		float fogFactor = 1.0 - exp( - fogDensity * fogDensity * vFogDepth * vFogDepth );
	#else
		float fogFactor = smoothstep( fogNear, fogFar, vFogDepth );
	#endif
	gl_FragColor.rgb = mix( gl_FragColor.rgb, fogColor, fogFactor );
#endif
"""


def assemble_final(fog_mode=0, perspective=True) -> str:
    """SSGIEffect.js:34-66 (FinalSSGIMaterial): ssgi_compose.frag with the fog chunks spliced in as the ctor does
    (`.replace("varying", "")`, the gl_FragColor line deleted by regex), wrapped the way postprocessing's EffectMaterial
    calls an Effect's mainImage (harness boundary): mainImage(inputColor, vUv, outputColor)."""
    s = _rd("ssgi/shader/ssgi_compose.frag")
    s = s.replace("#include <fog_pars_fragment>", CHUNK_FOG_PARS.replace("varying", ""))
    s = s.replace("#include <fog_fragment>", re.sub(r".*gl_FragColor.*", "", CHUNK_FOG))
    d = {"PERSPECTIVE_CAMERA": 1 if perspective else 0}  # SSGIEffect.js:63
    if fog_mode:
        d["USE_FOG"] = ""
    if fog_mode == 2:
        d["FOG_EXP2"] = ""
    main = "\nvarying vec2 vUv;\nvoid main() { vec4 c; mainImage(vec4(0.), vUv, c); gl_FragColor = c; }\n"
    return three_prefix(d, False) + CHUNK_PACKING + s + main


def write_assembled(outdir: str, **kw):
    """Build products for the GPU box (no /root/reference there): oracle/_ref/shaders/*.frag."""
    os.makedirs(outdir, exist_ok=True)
    more = tuple(("ssgi_%d_%d" % sr, assemble_ssgi(*sr)) for sr in ((1, 0), (3, 1), (12, 3), (17, 6), (24, 0)))  # tools/fuzz_vs_reference_gl.py --device
    for name, src in more + (("ssgi_20_5", assemble_ssgi(20, 5)), ("ssgi_8_2", assemble_ssgi(8, 2)), ("ssgi_40_5", assemble_ssgi(40, 5)),
                      ("temporal", assemble_temporal()), ("denoise", assemble_denoise()), ("compose", assemble_compose()),
                      ("ssgi_ssr_20_5", assemble_ssgi(20, 5, 1)), ("temporal_ssr", assemble_temporal(texture_count=1, input_type=2, reproject_specular=True, neighborhood_clamp=True)),
                      ("denoise_ssr", assemble_denoise(texture_count=1, is_texture_specular=(True, True))), ("compose_ssr", assemble_compose(input_type=2)),
                      ("ssgi_env_20_5", assemble_ssgi(20, 5, use_envmap=True)), ("ssgi_envmis_20_5", assemble_ssgi(20, 5, use_envmap=True, importance_sampling=True)), ("temporal_traa", assemble_traa()), ("final_fog0", assemble_final(0)), ("final_fog1", assemble_final(1)),
                      ("final_fog2", assemble_final(2))):
        with open(os.path.join(outdir, name + ".frag"), "w") as f:
            f.write(src)


# ------------------------------------------------------------------ ctypes layer


class GL:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            os.environ.setdefault("glsl_zero_init", "true")  # WebGL zero-init (Appendix C-6)
            os.environ.setdefault("LP_NUM_THREADS", str(os.cpu_count() or 1))
            path = os.path.join(REF_OUT, "libglref.so")
            L = ctypes.CDLL(path)
            L.glref_info.restype = ctypes.c_char_p
            L.glref_last_ms.restype = ctypes.c_double
            rc = L.glref_init()
            if rc != 0:
                raise RuntimeError("glref_init failed %d: %s" % (rc, L.glref_info()))
            cls._lib = L
        return cls._lib

    @classmethod
    def info(cls):
        return cls.lib().glref_info().decode()


class Tex:
    def __init__(self, w, h, fmt, linear=False, repeat=False, data=None):
        self.w, self.h, self.fmt = w, h, fmt
        ptr = None
        if data is not None:
            data = np.ascontiguousarray(data)
            ptr = data.ctypes.data_as(ctypes.c_void_p)
        self.id = GL.lib().glref_texture(w, h, fmt, int(linear), int(repeat), ptr)
        assert self.id > 0

    def upload(self, data):
        data = np.ascontiguousarray(data)
        rc = GL.lib().glref_tex_upload(self.id, data.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, rc

    def read(self) -> np.ndarray:
        out = np.empty((self.h, self.w, 4), np.float32)
        rc = GL.lib().glref_read(self.id, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, hex(rc)
        return out

    def generate_mipmaps(self):
        """three: generateMipmaps + LinearMipMapLinearFilter / LinearFilter (SSGIEffect.js:323-328)"""
        rc = GL.lib().glref_gen_mipmaps(self.id)
        assert rc == 0, rc

    def read_level(self, level) -> np.ndarray:
        out = np.empty((max(self.h >> level, 1), max(self.w >> level, 1), 4), np.float32)
        rc = GL.lib().glref_read_level(self.id, level, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, hex(rc)
        return out

    def free(self):
        GL.lib().glref_tex_free(self.id)


class CubeTex:
    """six size x size faces (+X -X +Y -Y +Z -Z, row j = t as handed to glTexImage2D).  mipmaps: three's CubeTexture default
    (LinearMipmapLinearFilter, chain from glGenerateMipmap); otherwise LinearFilter, level 0 only (HDRCubeTextureLoader's set-up)."""

    def __init__(self, faces, mipmaps=False):
        faces = np.ascontiguousarray(faces, np.float32)
        self.w = self.h = faces.shape[1]
        self.id = GL.lib().glref_cube_texture(self.w, FMT_RGBA32F, 2 if mipmaps else 1, faces.ctypes.data_as(ctypes.c_void_p))
        assert self.id > 0, self.id

    def free(self):
        GL.lib().glref_tex_free(self.id)


class Program:
    def __init__(self, src: str):
        log = ctypes.create_string_buffer(16384)
        self.id = GL.lib().glref_program(src.encode(), log, 16384)
        if self.id <= 0:
            raise RuntimeError("GLSL compile/link failed (%d): %s" % (self.id, log.value.decode(errors="replace")))
        self.last_ms = 0.0

    def sampler(self, name, tex: Tex):
        GL.lib().glref_bind_sampler(self.id, name.encode(), tex.id)

    def set(self, name, value):
        L = GL.lib()
        if isinstance(value, (bool, np.bool_)) or isinstance(value, (int, np.integer)):
            L.glref_uniform_int(self.id, name.encode(), int(value))
            return
        v = np.atleast_1d(np.asarray(value, np.float32)).ravel()
        kind = {1: 1, 2: 2, 3: 3, 16: 4}[v.size]
        L.glref_uniform(self.id, name.encode(), kind, v.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))

    def draw(self, targets):
        arr = (ctypes.c_int * len(targets))(*[t.id for t in targets])
        rc = GL.lib().glref_draw(self.id, arr, len(targets))
        assert rc == 0, "glref_draw failed: %x" % (rc & 0xffffffff)
        self.last_ms = GL.lib().glref_last_ms()


# ------------------------------------------------------------------ the chain


DEFAULTS = dict(  # src/ssgi/SSGIOptions.js:26-48
    distance=10.0, thickness=10.0, denoiseIterations=1, radius=3.0, phi=0.5, lumaPhi=5.0, depthPhi=2.0, normalPhi=50.0,
    roughnessPhi=50.0, specularPhi=50.0, envBlur=0.5, steps=20, refineSteps=5, missedRays=False, mode="ssgi", denoiseMode="full", resolutionScale=1.0, orthographic=False)


class GLRefChain:
    """SSGIEffect's K1->K2->K3->K4 chain (denoiseMode "full", inputType "diffuseSpecular") on llvmpipe.

    Persistent state mirrors the reference objects: K2 targets, K3 ping-pong A/B, K4 target,
    prev-camera uniforms, keepData flag, per-material blue-noise index (passed in explicitly).
    """

    def __init__(self, width, height, blue_noise_table: np.ndarray, shader_dir: str | None = None, **options):
        self.W, self.H = width, height
        self.o = dict(DEFAULTS)
        self.o.update(options)
        if shader_dir is None and not os.path.isdir(REFERENCE_SRC):
            shader_dir = os.path.join(REF_OUT, "shaders")  # build products of `make -C oracle ref` (GPU box: no /root/reference)
        self.t_env = None
        env = self.o.pop("environment", None)  # scene.environment: (H, W, 4) float32 equirect; HalfFloatType like RGBELoader's
        # importanceSampling: (marginalWeights[H], conditionalWeights[H, W], totalSumValue) as EquirectHdrInfoUniform's worker computes them
        self.importance = self.o.pop("importance", None)
        ssr = self.o["mode"] == "ssr"
        self.tc = 1 if ssr else 2  # SSGIEffect.js:70-77: "ssr" -> inputType "specular", one texture
        sfx = "_ssr" if ssr else ""
        if shader_dir is not None:
            def rd(name):
                with open(os.path.join(shader_dir, name + ".frag")) as f:
                    return f.read()
            if self.o["missedRays"]:
                raise RuntimeError("prebuilt shaders cover missedRays=false only")
            self.p_ssgi = Program(rd("ssgi%s%s_%d_%d" % (sfx, ("_envmis" if self.importance is not None else "_env") if env is not None else "",
                                                            self.o["steps"], self.o["refineSteps"])))
            self.p_temporal, self.p_denoise, self.p_compose = Program(rd("temporal" + sfx)), Program(rd("denoise" + sfx)), Program(rd("compose" + sfx))
        elif ssr:
            self.p_ssgi = Program(assemble_ssgi(self.o["steps"], self.o["refineSteps"], 1, True, self.o["missedRays"]))
            self.p_temporal = Program(assemble_temporal(texture_count=1, input_type=2, reproject_specular=True, neighborhood_clamp=True))
            self.p_denoise = Program(assemble_denoise(texture_count=1, is_texture_specular=(True, True)))
            self.p_compose = Program(assemble_compose(input_type=2))
        else:
            persp = not self.o["orthographic"]  # camera.isPerspectiveCamera -> every pass's PERSPECTIVE_CAMERA define
            self.p_ssgi = Program(assemble_ssgi(self.o["steps"], self.o["refineSteps"], 0, True, self.o["missedRays"], use_envmap=env is not None, perspective=persp,
                                                importance_sampling=self.importance is not None))
            self.p_temporal = Program(assemble_temporal(perspective=persp))
            self.p_denoise = Program(assemble_denoise())
            self.p_compose = Program(assemble_compose(perspective=persp))
        W, H = width, height
        self.t_depth = Tex(W, H, FMT_R32F)
        self.t_gbuffer = Tex(W, H, FMT_RGBA32F)
        self.t_velocity = Tex(W, H, FMT_RGBA32F)
        self.t_direct = Tex(W, H, FMT_RGBA32F)
        self.t_blue = Tex(128, 128, FMT_RGBA8, repeat=True, data=blue_noise_table)
        self.t_empty = Tex(1, 1, FMT_RGBA8, data=np.zeros(4, np.uint8))  # three's empty texture (Appendix D-1)
        # SSGIPass.setSize :52-57: the pass's target is (W*s) x (H*s)
        rs = float(self.o["resolutionScale"])
        self.ssgi_size = (int(W * rs), int(H * rs))
        assert self.ssgi_size == (W * rs, H * rs), "W*resolutionScale and H*resolutionScale must be whole"
        self.t_ssgi = Tex(self.ssgi_size[0], self.ssgi_size[1], FMT_RGBA32F)
        self.t_temporal = [Tex(W, H, FMT_RGBA32F), Tex(W, H, FMT_RGBA32F)]
        self.t_A = [Tex(W, H, FMT_RGBA16F, linear=True), Tex(W, H, FMT_RGBA16F, linear=True)]
        self.t_B = [Tex(W, H, FMT_RGBA16F, linear=True), Tex(W, H, FMT_RGBA16F, linear=True)]
        self.t_compose = Tex(W, H, FMT_RGBA32F)
        if env is not None:  # SSGIEffect.keepEnvMapUpdated :309-362
            self.env_size = (env.shape[1], env.shape[0])
            self.t_env = Tex(env.shape[1], env.shape[0], FMT_RGBA16F, linear=True, data=np.ascontiguousarray(env, np.float32).astype(np.float16))
            self.t_env.generate_mipmaps()
            if self.importance is not None:  # EquirectHdrInfoUniform.updateFrom :383-389: height x 1 and width x height R32F, NearestFilter
                mw, cw, _ = self.importance
                self.t_marginal = Tex(env.shape[0], 1, FMT_R32F, data=np.ascontiguousarray(mw, np.float32))
                self.t_conditional = Tex(env.shape[1], env.shape[0], FMT_R32F, data=np.ascontiguousarray(cw, np.float32))
        # Denoiser.js:41-61: a denoise pass only in "full"/"denoised" (its target B then overrides K2's history), a compose pass only in "full*"
        self.dm = self.o["denoiseMode"]
        assert self.dm in ("full", "full_temporal", "denoised", "temporal")
        self.has_denoise, self.has_compose = self.dm in ("full", "denoised"), self.dm.startswith("full")
        self.t_fb = Tex(W, H, FMT_RGBA32F, linear=True) if not self.has_denoise else None  # TemporalReprojectPass.framebufferTexture (:137-142)
        self.keep_data = 0.0  # SSGIEffect's ctor setters call reset() (SSGIEffect.js:264)
        self.prev = None
        self.ms = {}

    def upload_frame(self, frame):
        self.t_depth.upload(frame.depth)
        self.t_gbuffer.upload(frame.gbuffer.view(np.float32))
        self.t_velocity.upload(frame.velocity.view(np.float32))
        self.t_direct.upload(frame.direct)

    # -- K1: SSGIPass.render (src/ssgi/pass/SSGIPass.js:68-95)
    def ssgi(self, cam, blue_noise_index: int):
        p, o = self.p_ssgi, self.o
        near, far = float(cam.near), float(cam.far)
        # SSGIPass.js:89 accumulatedTexture = denoiser.texture (Denoiser.js:67-78); "denoised": the getter returns an ARRAY, which
        # three binds as its empty texture
        p.sampler("accumulatedTexture", {"full": self.t_compose, "full_temporal": self.t_compose, "temporal": self.t_temporal[0],
                                         "denoised": self.t_empty}[self.dm])
        p.sampler("gBufferTexture", self.t_gbuffer)
        p.sampler("depthTexture", self.t_depth)
        p.sampler("velocityTexture", self.t_empty)  # SSGIPass.js:89 reads an undefined property
        p.sampler("directLightTexture", self.t_direct)
        p.sampler("blueNoiseTexture", self.t_blue)
        p.set("cameraMatrixWorld", cam.matrixWorld)
        p.set("viewMatrix", cam.matrixWorldInverse)
        p.set("projectionMatrix", cam.projectionMatrix)
        p.set("projectionMatrixInverse", cam.projectionMatrixInverse)
        p.set("cameraNear", near)
        p.set("cameraFar", far)
        p.set("nearMinusFar", near - far)
        p.set("farMinusNear", far - near)
        p.set("nearMulFar", near * far)
        p.set("rayDistance", float(o["distance"]))
        p.set("thickness", float(o["thickness"]))
        p.set("envBlur", float(o["envBlur"]))
        if self.t_env is not None:
            p.sampler("envMapInfo.map", self.t_env)
            if self.importance is not None:
                tot = float(self.importance[2])
                p.sampler("envMapInfo.marginalWeights", self.t_marginal)
                p.sampler("envMapInfo.conditionalWeights", self.t_conditional)
                p.set("envMapInfo.size", [float(self.env_size[0]), float(self.env_size[1])])
                p.set("envMapInfo.totalSumWhole", float(int(tot)))          # ~~totalSumValue (:391-394)
                p.set("envMapInfo.totalSumDecimal", float(tot - int(tot)))
            import math
            p.set("maxEnvMapMipLevel", float(math.floor(math.log2(max(self.env_size))) + 1))  # getMaxMipLevel, Utils.js:30-34
        else:
            p.set("maxEnvMapMipLevel", 0.0)
        p.set("backgroundColor", [0.0, 0.0, 0.0])
        p.set("resolution", [float(self.ssgi_size[0]), float(self.ssgi_size[1])])  # = renderTarget size (SSGIPass.js:56)
        p.set("blueNoiseSize", [128.0, 128.0])
        p.set("blueNoiseIndex", int(blue_noise_index))
        p.draw([self.t_ssgi])
        self.ms["ssgi"] = p.last_ms

    # -- K2: TemporalReprojectPass.render (src/temporal-reproject/TemporalReprojectPass.js:162-214)
    def temporal(self, cam, camera_moved: bool = True, full_accumulate_option=True, max_blend=1.0, nci=0.5):
        p = self.p_temporal
        prev = self.prev if self.prev is not None else cam  # ctor clones the current matrices (:95-104)
        p.sampler("inputTexture", self.t_ssgi)
        p.sampler("velocityTexture", self.t_velocity)
        if self.has_denoise:
            p.sampler("accumulatedTexture0", self.t_B[0])  # Denoiser.js:51 overrideAccumulatedTextures
            if self.tc == 2:
                p.sampler("accumulatedTexture1", self.t_B[1])
        else:  # TemporalReprojectPass.js:148-151: every index reads the pass's one framebuffer copy
            for i in range(self.tc):
                p.sampler("accumulatedTexture%d" % i, self.t_fb)
        p.set("projectionMatrix", cam.projectionMatrix)
        p.set("projectionMatrixInverse", cam.projectionMatrixInverse)
        p.set("cameraMatrixWorld", cam.matrixWorld)
        p.set("viewMatrix", cam.matrixWorldInverse)
        p.set("cameraPos", cam.position)
        p.set("prevViewMatrix", prev.matrixWorldInverse)
        p.set("prevCameraMatrixWorld", prev.matrixWorld)
        p.set("prevProjectionMatrix", prev.projectionMatrix)
        p.set("prevProjectionMatrixInverse", prev.projectionMatrixInverse)
        p.set("prevCameraPos", prev.position)
        p.set("fullAccumulate", bool(full_accumulate_option and not camera_moved))
        p.set("keepData", float(self.keep_data))
        p.set("invTexSize", [1.0 / self.W, 1.0 / self.H])
        p.set("cameraNear", float(cam.near))
        p.set("cameraFar", float(cam.far))
        p.set("maxBlend", float(max_blend))
        p.set("neighborhoodClampIntensity", float(nci))
        p.draw(self.t_temporal[:self.tc])
        self.ms["temporal"] = p.last_ms
        self.keep_data = 1.0
        if not self.has_denoise:  # :197-201 copyFramebufferToTexture reads the framebuffer's read buffer = colour attachment 0
            self.t_fb.upload(self.t_temporal[0].read())
        self.prev = cam

    # -- K3: PoissonDenoisePass.render (src/denoise/pass/PoissonDenoisePass.js:135-149)
    def denoise(self, cam, blue_noise_indices):
        p, o = self.p_denoise, self.o
        p.sampler("depthTexture", self.t_depth)
        p.sampler("gBufferTexture", self.t_gbuffer)
        p.sampler("blueNoiseTexture", self.t_blue)
        for k in ("radius", "phi", "lumaPhi", "depthPhi", "normalPhi", "roughnessPhi", "specularPhi"):
            p.set(k, float(o[k]))
        p.set("projectionMatrix", cam.projectionMatrix)
        p.set("projectionMatrixInverse", cam.projectionMatrixInverse)
        p.set("cameraMatrixWorld", cam.matrixWorld)
        p.set("viewMatrix", cam.matrixWorldInverse)
        p.set("resolution", [float(self.W), float(self.H)])
        p.set("blueNoiseSize", [128.0, 128.0])
        self.ms["denoise"] = []
        for i in range(2 * int(o["denoiseIterations"]) if self.has_denoise else 0):
            horizontal = i % 2 == 0
            src = self.t_temporal if i == 0 else (self.t_B if horizontal else self.t_A)
            dst = self.t_A if horizontal else self.t_B
            p.sampler("inputTexture", src[0])
            if self.tc == 2:
                p.sampler("inputTexture2", src[1])
            p.set("blueNoiseIndex", int(blue_noise_indices[i]))
            p.draw(dst[:self.tc])
            self.ms["denoise"].append(p.last_ms)

    # -- K4: DenoiserComposePass.render (src/denoise/pass/DenoiserComposePass.js:129-135)
    def compose(self, cam):
        p = self.p_compose
        p.sampler("depthTexture", self.t_depth)
        p.sampler("gBufferTexture", self.t_gbuffer)
        if not self.has_compose:
            return
        src = self.t_B if self.has_denoise else self.t_temporal  # Denoiser.js:53 composerInputTextures
        if self.tc == 2:
            p.sampler("diffuseGiTexture", src[0])
            p.sampler("specularGiTexture", src[1])
        else:  # DenoiserComposePass.js:26-33 inputType "specular": textures[0] is the specular GI; Denoiser.js:101-103 sceneTexture
            p.sampler("diffuseGiTexture", self.t_empty)
            p.sampler("specularGiTexture", src[0])
            p.sampler("sceneTexture", self.t_direct)
        p.set("viewMatrix", cam.matrixWorldInverse)
        p.set("cameraMatrixWorld", cam.matrixWorld)
        p.set("projectionMatrix", cam.projectionMatrix)
        p.set("projectionMatrixInverse", cam.projectionMatrixInverse)
        p.set("cameraNear", float(cam.near))
        p.set("cameraFar", float(cam.far))
        p.draw([self.t_compose])
        self.ms["compose"] = p.last_ms


class GLRefTRAA:
    """TRAAEffect's device work (src/traa/TRAAEffect.js:52-75) on llvmpipe: TemporalReprojectPass alone on the composer's
    input buffer, history = the pass's own framebuffer copy (TemporalReprojectPass.js:137-151,197-201).

    `half` selects the composer's frameBufferType: HalfFloatType (example/main.js:173) -> RGBA16F input buffer, render
    target and framebuffer copy; FloatType -> RGBA32F throughout.  copyFramebufferToTexture copies between two textures
    of the same format, i.e. exactly: done here as read-back + upload of the same texel values."""

    def __init__(self, width, height, half=True, shader_dir: str | None = None, full_accumulate=True):
        self.W, self.H, self.half = width, height, half
        if shader_dir is None and not os.path.isdir(REFERENCE_SRC):
            shader_dir = os.path.join(REF_OUT, "shaders")
        if shader_dir is not None:
            with open(os.path.join(shader_dir, "temporal_traa.frag")) as f:
                self.p = Program(f.read())
        else:
            self.p = Program(assemble_traa())
        fmt = FMT_RGBA16F if half else FMT_RGBA32F
        self.t_input = Tex(width, height, fmt)
        self.t_velocity = Tex(width, height, FMT_RGBA32F)
        self.t_out = Tex(width, height, fmt)                # renderTarget: Nearest (:63-68)
        self.t_fb = Tex(width, height, fmt, linear=True)    # framebufferTexture: LinearFilter (:139-142)
        self.full_accumulate = full_accumulate
        self.keep_data = 1.0  # the uniform's initial value (TemporalReprojectMaterial.js); TRAAEffect never resets on its own
        self.prev = None
        self.ms = 0.0

    def upload_frame(self, frame):
        d = frame.direct  # the composer's input buffer = the lit scene
        self.t_input.upload(d.astype(np.float16) if self.half else d)
        self.t_velocity.upload(frame.velocity.view(np.float32))

    def render(self, cam, camera_moved: bool = True):
        p = self.p
        prev = self.prev if self.prev is not None else cam
        p.sampler("inputTexture", self.t_input)
        p.sampler("velocityTexture", self.t_velocity)
        p.sampler("accumulatedTexture0", self.t_fb)
        for name, v in (("projectionMatrix", cam.projectionMatrix), ("projectionMatrixInverse", cam.projectionMatrixInverse),
                        ("cameraMatrixWorld", cam.matrixWorld), ("viewMatrix", cam.matrixWorldInverse), ("cameraPos", cam.position),
                        ("prevViewMatrix", prev.matrixWorldInverse), ("prevCameraMatrixWorld", prev.matrixWorld),
                        ("prevProjectionMatrix", prev.projectionMatrix), ("prevProjectionMatrixInverse", prev.projectionMatrixInverse),
                        ("prevCameraPos", prev.position)):
            p.set(name, v)
        p.set("fullAccumulate", bool(self.full_accumulate and not camera_moved))
        p.set("keepData", float(self.keep_data))
        p.set("invTexSize", [1.0 / self.W, 1.0 / self.H])
        p.set("cameraNear", float(cam.near))
        p.set("cameraFar", float(cam.far))
        p.set("maxBlend", 0.9)                     # TRAAEffect.js:24
        p.set("neighborhoodClampIntensity", 1.0)   # :26
        p.draw([self.t_out])
        self.ms = p.last_ms
        self.keep_data = 1.0
        out = self.t_out.read()
        self.t_fb.upload(out.astype(np.float16) if self.half else out)  # copyFramebufferToTexture (:197-201)
        self.prev = cam
        return out


def run_final(width, height, depth, gi, scene, cam, fog_mode=0, fog_color=(0.5, 0.6, 0.7), fog_near=1.0, fog_far=30.0, fog_density=0.05,
              is_debug=False, shader_dir: str | None = None):
    """One draw of SSGIEffect's own fragment (FinalSSGIMaterial) on llvmpipe; returns the RGBA32F output."""
    persp = bool(getattr(cam, "isPerspectiveCamera", True))
    if shader_dir is None and not os.path.isdir(REFERENCE_SRC):
        shader_dir = os.path.join(REF_OUT, "shaders")
    if shader_dir is not None and persp:
        with open(os.path.join(shader_dir, "final_fog%d.frag" % fog_mode)) as f:
            p = Program(f.read())
    else:
        p = Program(assemble_final(fog_mode, perspective=persp))
    t_depth, t_gi, t_scene = Tex(width, height, FMT_R32F, data=depth), Tex(width, height, FMT_RGBA32F, data=gi), Tex(width, height, FMT_RGBA32F, data=scene)
    t_out = Tex(width, height, FMT_RGBA32F)
    p.sampler("inputTexture", t_gi)
    p.sampler("sceneTexture", t_scene)
    p.sampler("depthTexture", t_depth)
    p.set("isDebug", bool(is_debug))
    if fog_mode:
        p.set("fogColor", list(fog_color))
        if fog_mode == 2:
            p.set("fogDensity", float(fog_density))
        else:
            p.set("fogNear", float(fog_near))
            p.set("fogFar", float(fog_far))
    p.set("cameraNear", float(cam.near))
    p.set("cameraFar", float(cam.far))
    p.draw([t_out])
    out = t_out.read()
    for t in (t_depth, t_gi, t_scene, t_out):
        t.free()
    return out


def chain_final(c: "GLRefChain", frame, fog_mode=0, **kw):
    """SSGIEffect's own fragment over a chain's current state: inputTexture = outputTexture[0] ?? outputTexture (SSGIEffect.js:139,402)."""
    src = {"full": c.t_compose, "full_temporal": c.t_compose, "temporal": c.t_temporal[0], "denoised": c.t_B[0]}[c.dm]
    return run_final(c.W, c.H, frame.depth, src.read(), frame.direct, frame.camera, fog_mode=fog_mode, **kw)


def assemble_pack() -> str:
    """The encode side of the codec as the raster passes' fragment epilogues call it (GBufferMaterial.js:84-89, VelocityDepthNormalMaterial.js
    :76-83,186-188), fed from attribute textures instead of the rasteriser's interpolants: target 0 = packGBuffer(...), target 1 =
    vec4(vel.xy, packNormal(worldNormal), fragCoordZ)."""
    gb = _rd("gbuffer/shader/gbuffer_packing.glsl")
    main = """
varying vec2 vUv;
uniform highp sampler2D tDiffuse; uniform highp sampler2D tNormal; uniform highp sampler2D tRoughMetal; uniform highp sampler2D tEmissive;
uniform highp sampler2D tVelocityDepth;
layout(location = 0) out highp vec4 oGBuffer;
layout(location = 1) out highp vec4 oVelocity;
void main() {
  vec4 diffuseColor = textureLod(tDiffuse, vUv, 0.);
  vec3 worldNormal = textureLod(tNormal, vUv, 0.).xyz;
  vec2 rm = textureLod(tRoughMetal, vUv, 0.).xy;
  vec3 totalEmissiveRadiance = textureLod(tEmissive, vUv, 0.).xyz;
  vec4 vd = textureLod(tVelocityDepth, vUv, 0.);
  oGBuffer = packGBuffer(diffuseColor, worldNormal, rm.x, rm.y, totalEmissiveRadiance);
  oVelocity = vec4(vd.x, vd.y, packNormal(worldNormal), vd.z);
}
"""
    return three_prefix({}, True) + gb + main


def assemble_cube_to_equirect() -> str:
    """CubeToEquirectEnvPass.js:21-42 (inline template literal) under three's fragment prefix; the sampler precision is the one every
    desktop GL gives a samplerCube (GLSL ES would default it to lowp, which Mesa lowers to fp16 fetches — not what any GPU runs)."""
    js = _rd("ssgi/pass/CubeToEquirectEnvPass.js")
    m = re.search(r"fragmentShader:\s*/\* glsl \*/\s*`([\s\S]*?)`,\s*vertexShader", js)
    return three_prefix({}, False) + "precision highp samplerCube;\n" + m.group(1)


def cube_equirect_size(face_size: int, max_width: int = 4096):
    """generateEquirectEnvMap's target size (CubeToEquirectEnvPass.js:62-74)"""
    import math
    w = int(2 ** math.ceil(math.log2(2 * face_size * 3 ** 0.5)))
    h = int(2 ** math.ceil(math.log2(face_size * 3 ** 0.5)))
    return (max_width, max_width // 2) if w > max_width else (w, h)


def run_cube_to_equirect(faces: np.ndarray, W: int, H: int, mipmaps: bool) -> np.ndarray:
    """The pass on llvmpipe: (H, W, 4) float32, row 0 = bottom (what readRenderTargetPixels returns)."""
    p = Program(assemble_cube_to_equirect())
    c = CubeTex(faces, mipmaps)
    p.sampler("cubeMap", c)
    out = Tex(W, H, FMT_RGBA32F)
    p.draw([out])
    r = out.read()
    out.free()
    c.free()
    return r


def run_pack(aov: dict, depth: np.ndarray):
    """Both packers on llvmpipe; returns (gbuffer, velocity) as (H, W, 4) uint32 bit patterns (texels with depth == 1 are whatever the
    packers make of the planes there — the clear colour is the rasteriser's business, not theirs)."""
    H, W = depth.shape
    p = Program(assemble_pack())
    rgba = lambda a: np.ascontiguousarray(np.concatenate([a, np.zeros(a.shape[:2] + (4 - a.shape[2],), np.float32)], -1), np.float32)
    t = {"tDiffuse": Tex(W, H, FMT_RGBA32F, data=rgba(aov["diffuse"])), "tNormal": Tex(W, H, FMT_RGBA32F, data=rgba(aov["normal"])),
         "tRoughMetal": Tex(W, H, FMT_RGBA32F, data=rgba(np.stack([aov["roughness"], aov["metalness"]], -1))),
         "tEmissive": Tex(W, H, FMT_RGBA32F, data=rgba(aov["emissive"])),
         "tVelocityDepth": Tex(W, H, FMT_RGBA32F, data=rgba(np.concatenate([aov["velocity"], depth[..., None]], -1)))}
    for k, v in t.items():
        p.sampler(k, v)
    og, ov = Tex(W, H, FMT_RGBA32F), Tex(W, H, FMT_RGBA32F)
    p.draw([og, ov])
    g, v = og.read().view(np.uint32), ov.read().view(np.uint32)
    for x in list(t.values()) + [og, ov]:
        x.free()
    return g, v
