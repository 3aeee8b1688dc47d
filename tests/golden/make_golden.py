#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own GLSL on CPU llvmpipe.

Build-container only: needs /root/reference (the shaders are read from there at run time and are
never copied into this repository) and Mesa's swrast_dri.so.  Usage:

    make -C oracle && python tests/golden/make_golden.py

Each .npz holds, for a short frame sequence of the synthetic dump (seed 1234), the INPUT planes
and the render target of every pass (K1 ssgi, K2 temporal x2, K3 A/B x2, K4 compose; for the traa_* files the
TemporalReprojectPass target of TRAAEffect) as produced by oracle/glref/chain.py — i.e. by src/ssgi/shader/ssgi.frag,
src/temporal-reproject/shader/temporal_reproject.frag, src/denoise/shader/poisson_denoise.frag and
the DenoiserComposePass shader, unmodified, under the uniform values of the reference's JS drivers.
The blue-noise indices follow src/utils/BlueNoiseUtils.js:24-32 from pinned start indices.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "realism-effects_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))

from rfx_amd.scene import synthetic_frame  # noqa: E402  (input generator only)
import chain  # noqa: E402

M = 0x7FFFFFFF


def reference_importance(env, flip_y=False):
    """EquirectHdrInfoUniform's CPU pass run by the REFERENCE'S OWN code: the worker function (plain ES2015) is cut out of
    src/ssgi/utils/EquirectHdrInfoUniform.js at run time and executed by node with a stub postMessage."""
    import json
    import re
    import subprocess
    import tempfile
    src = open("/root/reference/src/ssgi/utils/EquirectHdrInfoUniform.js", encoding="utf-8-sig").read()
    fn = re.search(r"const workerOnMessage = (\(\{ data: \{ width, height, isFloatType, flipY, data \} \}\) => \{[\s\S]*?\n\})\n\nconst blob", src).group(1)
    d = tempfile.mkdtemp()
    h, w = env.shape[:2]
    np.ascontiguousarray(env, np.float32).tofile(os.path.join(d, "env.bin"))
    js = ("const fs=require('fs');let r=null;global.postMessage=x=>{r=x};const f=%s;const b=fs.readFileSync('%s/env.bin');"
          "f({data:{width:%d,height:%d,isFloatType:true,flipY:%s,data:new Float32Array(b.buffer,b.byteOffset,b.length/4)}});"
          "fs.writeFileSync('%s/m.bin',Buffer.from(r.marginalDataArray.buffer));fs.writeFileSync('%s/c.bin',Buffer.from(r.conditionalDataArray.buffer));"
          "console.log(JSON.stringify(r.totalSumValue))") % (fn, d, w, h, "true" if flip_y else "false", d, d)
    open(os.path.join(d, "run.js"), "w").write(js)
    tot = json.loads(subprocess.check_output(["node", os.path.join(d, "run.js")], text=True))
    return np.fromfile(os.path.join(d, "m.bin"), np.float32), np.fromfile(os.path.join(d, "c.bin"), np.float32).reshape(h, w), float(tot)


def cam_arrays(cam, prefix):
    return {prefix + k: np.asarray(getattr(cam, k)) for k in
            ("projectionMatrix", "projectionMatrixInverse", "matrixWorld", "matrixWorldInverse", "position", "quaternion")}


def run(name, W, H, frames, steps, refine, iterations, ssgi_start=1000, denoise_start=2000, mode="ssgi", missed_rays=False, denoise_mode="full",
        environment=None, env_blur=0.5, resolution_scale=1.0, ortho_half_height=None, importance_sampling=False):
    bn = np.fromfile(os.path.join(ROOT, "realism-effects_amd", "data", "blue_noise_128_rgba8.bin"), np.uint8).reshape(128, 128, 4)
    importance = None
    if importance_sampling:  # the half-float map's texels, as the worker converts them (fromHalfFloat) before the CDF pass
        importance = reference_importance(np.ascontiguousarray(environment, np.float32).astype(np.float16).astype(np.float32))
    c = chain.GLRefChain(W, H, bn, steps=steps, refineSteps=refine, denoiseIterations=iterations, mode=mode, missedRays=missed_rays, importance=importance,
                         denoiseMode=denoise_mode, environment=environment, envBlur=env_blur, resolutionScale=resolution_scale,
                         orthographic=ortho_half_height is not None)
    tc = c.tc
    out = dict(width=W, height=H, frames=frames, steps=steps, refineSteps=refine, denoiseIterations=iterations, ssgi_start=ssgi_start,
               denoise_start=denoise_start, gl_info=chain.GL.info(), mode=mode, textureCount=tc, missedRays=int(missed_rays), denoiseMode=denoise_mode)
    out["resolutionScale"] = resolution_scale
    out["orthographic"] = int(ortho_half_height is not None)
    if environment is not None:  # scene.environment (HalfFloatType, mipmapped by the effect) + the envBlur option
        out["environment"], out["envBlur"] = environment, env_blur
    out["importanceSampling"] = int(importance_sampling)
    if importance is not None:  # what the reference's own JS produced: pins rfx_amd.envmap / js/envmap.js
        out["marginalWeights"], out["conditionalWeights"], out["totalSumValue"] = importance
    si = di = 0
    for fi in range(frames):
        f = synthetic_frame(W, H, fi, ortho_half_height=ortho_half_height)
        c.upload_frame(f)
        k = "f%d_" % fi
        out[k + "depth"], out[k + "gbuffer"], out[k + "velocity"], out[k + "direct"] = f.depth, f.gbuffer, f.velocity, f.direct
        out.update(cam_arrays(f.camera, k + "cam_"))
        out[k + "near"], out[k + "far"] = f.camera.near, f.camera.far
        si = (ssgi_start + si + 1) % M
        c.ssgi(f.camera, si)
        out[k + "ssgi"] = c.t_ssgi.read().view(np.uint32)
        out[k + "ssgi_index"] = si
        c.temporal(f.camera, camera_moved=True)
        for j in range(tc):
            out[k + "temporal%d" % j] = c.t_temporal[j].read()
        idx = []
        for _ in range(2 * iterations if c.has_denoise else 0):
            di = (denoise_start + di + 1) % M
            idx.append(di)
        c.denoise(f.camera, idx)
        out[k + "denoise_index"] = np.array(idx, np.int64)
        if c.has_denoise:
            for j in range(tc):
                # RGBA16F targets read back as float32 are exactly representable in half
                out[k + "A%d" % j] = c.t_A[j].read().astype(np.float16).view(np.uint16)
                out[k + "B%d" % j] = c.t_B[j].read().astype(np.float16).view(np.uint16)
        c.compose(f.camera)
        if c.has_compose:
            out[k + "compose"] = c.t_compose.read()
        if denoise_mode != "full":
            out[k + "final"] = chain.chain_final(c, f)
        if ortho_half_height is not None:  # the effect's own fragment with fog: the last orthographic getViewZ site
            out[k + "final_fog2"] = chain.chain_final(c, f, fog_mode=2)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def run_traa(name, W, H, frames, half):
    """TRAAEffect (src/traa/TRAAEffect.js): temporal_reproject.frag alone with TRAA's defines, its own framebuffer copy as
    history; composer buffers HalfFloatType (half=True, example/main.js:173) or FloatType."""
    c = chain.GLRefTRAA(W, H, half=half)
    out = dict(width=W, height=H, frames=frames, half=int(half), gl_info=chain.GL.info())
    for fi in range(frames):
        f = synthetic_frame(W, H, fi)
        c.upload_frame(f)
        k = "f%d_" % fi
        out[k + "velocity"], out[k + "direct"] = f.velocity, f.direct
        out.update(cam_arrays(f.camera, k + "cam_"))
        out[k + "near"], out[k + "far"] = f.camera.near, f.camera.far
        out[k + "out"] = c.render(f.camera, camera_moved=True)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def run_final(name, W, H):
    """SSGIEffect's own fragment (ssgi_compose.frag) over the K4 output of a 2-frame chain: no fog / Fog / FogExp2 / isDebug."""
    bn = np.fromfile(os.path.join(ROOT, "realism-effects_amd", "data", "blue_noise_128_rgba8.bin"), np.uint8).reshape(128, 128, 4)
    c = chain.GLRefChain(W, H, bn, steps=12, refineSteps=3, denoiseIterations=1)
    for fi in range(2):
        f = synthetic_frame(W, H, fi)
        c.upload_frame(f)
        c.ssgi(f.camera, 31 + fi)
        c.temporal(f.camera)
        c.denoise(f.camera, [41 + 2 * fi, 42 + 2 * fi])
        c.compose(f.camera)
    gi = c.t_compose.read()
    out = dict(width=W, height=H, depth=f.depth, gi=gi, scene=f.direct, near=f.camera.near, far=f.camera.far, gl_info=chain.GL.info(),
               fogColor=np.array([0.5, 0.6, 0.7], np.float32), fogNear=1.0, fogFar=30.0, fogDensity=0.05)
    for m in (0, 1, 2):
        out["final_fog%d" % m] = chain.run_final(W, H, f.depth, gi, f.direct, f.camera, fog_mode=m)
    out["final_debug"] = chain.run_final(W, H, f.depth, gi, f.direct, f.camera, fog_mode=0, is_debug=True)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def run_pack(name, W, H):
    """The encode side of the codec (packGBuffer, packNormal) on llvmpipe over unpacked attribute planes of the synthetic scene, with HDR
    emissive values (incl. exact powers of two) sprinkled in."""
    from rfx_amd.scene import AnalyticScene
    f = AnalyticScene(1234).render(W, H, 1, aov=True)
    rng = np.random.RandomState(3)
    a = {k: v.copy() for k, v in f.aov.items()}
    m = rng.rand(H, W) < 0.3
    a["emissive"][m] = (rng.rand(int(m.sum()), 3) * np.array([8, 2, 0.5])).astype(np.float32)
    a["emissive"][5, 5], a["emissive"][5, 6], a["emissive"][5, 7] = (1.0, 0.5, 0.25), (2.0, 0, 0), (0.5, 0.5, 0.5)
    g, v = chain.run_pack(a, f.depth)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, width=W, height=H, depth=f.depth, gbuffer=g, velocity=v, gl_info=chain.GL.info(), **{"aov_" + k: x for k, x in a.items()})
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


def synthetic_cube(S, seed=21):
    """An HDR cube environment: per-face sky gradient, a sun blob on +Y that spills over the edges onto its neighbours, a checker on -Z and
    texel noise everywhere (every face, edge and corner carries contrast)."""
    rng = np.random.RandomState(seed)
    g = (np.arange(S, dtype=np.float32) + 0.5) / S * 2 - 1
    a, b = np.meshgrid(g, g)  # a = sc / ma along i, b = tc / ma along j
    one = np.ones_like(a)
    dirs = [(one, -b, -a), (-one, -b, a), (a, one, b), (a, -one, -b), (a, -b, one), (-a, -b, -one)]
    faces = np.zeros((6, S, S, 4), np.float32)
    sun = np.array([0.35, 0.8, 0.45], np.float32)
    sun /= np.linalg.norm(sun)
    for f, (x, y, z) in enumerate(dirs):
        n = np.sqrt(x * x + y * y + z * z)
        x, y, z = x / n, y / n, z / n
        sky = 0.3 + 0.7 * np.clip(y, 0, 1)
        faces[f, ..., 0] = 0.4 * sky + 0.05 * (1 + x)
        faces[f, ..., 1] = 0.6 * sky + 0.05 * (1 + z)
        faces[f, ..., 2] = 1.0 * sky
        c = np.clip(x * sun[0] + y * sun[1] + z * sun[2], 0, 1)
        faces[f, ..., :3] += (40.0 * c ** 64)[..., None] * np.array([1.0, 0.9, 0.7], np.float32)
        faces[f, ..., 3] = 1.0
    faces[5, ..., :3] *= (0.6 + 0.4 * ((np.floor((a + 1) * 3) + np.floor((b + 1) * 3)) % 2))[..., None]
    faces[..., :3] *= (0.9 + 0.2 * rng.rand(6, S, S, 1)).astype(np.float32)
    return faces.astype(np.float32)


def run_cube(name, S):
    """CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js) on llvmpipe: the cube sampled LinearFilter (no mip chain) and as a
    three CubeTexture by default (LinearMipmapLinearFilter over glGenerateMipmap's chain), at generateEquirectEnvMap's own target size."""
    faces = synthetic_cube(S)
    W, H = chain.cube_equirect_size(S)
    out = dict(size=S, width=W, height=H, faces=faces, gl_info=chain.GL.info())
    out["equirect_linear"] = chain.run_cube_to_equirect(faces, W, H, False)
    out["equirect_mipmapped"] = chain.run_cube_to_equirect(faces, W, H, True)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    run("chain_160x90_s20r5_it1", 160, 90, frames=3, steps=20, refine=5, iterations=1)
    run("chain_97x55_s8r2_it2", 97, 55, frames=2, steps=8, refine=2, iterations=2)
    run("chain_ssr_128x72_s20r5_it1", 128, 72, frames=2, steps=20, refine=5, iterations=1, mode="ssr")
    run("chain_missed_96x54_s12r3_it1", 96, 54, frames=2, steps=12, refine=3, iterations=1, missed_rays=True)
    from rfx_amd.scene import synthetic_environment
    run("chain_env_128x72_s12r3_it1", 128, 72, frames=2, steps=12, refine=3, iterations=1, environment=synthetic_environment(128, 64))
    run("chain_envsharp_96x54_s12r3_it1", 96, 54, frames=2, steps=12, refine=3, iterations=1, environment=synthetic_environment(128, 64), env_blur=0.1)
    run("chain_envmis_128x72_s12r3_it1", 128, 72, frames=2, steps=12, refine=3, iterations=1, environment=synthetic_environment(128, 64),
        importance_sampling=True)  # the reference's DEFAULT with an environment: importanceSampling (MIS)
    # resolutionScale option.  Only scales whose sample positions fall on texel CENTRES of the full-resolution inputs are comparable
    # across rasterisers (0.5 at these sizes): 0.75 puts every third row exactly on a texel boundary, where the nearest texel depends on
    # the last ulp of the rasteriser's varying interpolation (llvmpipe's vUv differs from (i+0.5)/n by <= 1 ulp, measured) — there the
    # reference itself is implementation-defined.
    run("chain_rs050_128x72_s12r3_it1", 128, 72, frames=2, steps=12, refine=3, iterations=1, resolution_scale=0.5)
    run("chain_ortho_120x68_s12r3_it1", 120, 68, frames=3, steps=12, refine=3, iterations=1, ortho_half_height=3.2)  # OrthographicCamera
    for dm in ("full_temporal", "temporal", "denoised"):  # the other Denoiser modes (Denoiser.js:7,41-78)
        run("chain_%s_104x58_s10r2" % dm, 104, 58, frames=3, steps=10, refine=2, iterations=1, denoise_mode=dm)
    run_traa("traa_half_128x72", 128, 72, frames=3, half=True)
    run_traa("traa_float_96x54", 96, 54, frames=3, half=False)
    run_final("final_112x63", 112, 63)
    run_pack("pack_96x54", 96, 54)
    run_cube("cube_32", 32)
