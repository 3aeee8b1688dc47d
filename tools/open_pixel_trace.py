#!/usr/bin/env python3
"""Build container (needs /root/reference + llvmpipe): the march of the ONE pinned "open pixel" (configs[4]'s options at 1920x1080, frame 10, K1, pixel
(y 584, x 676)), value by value, in the reference GL and in the C restatement — where do the two first differ, and by how much?

The reference chain runs to the frame on llvmpipe; then K1's fragment program is run once more per probed quantity with a DIAGNOSTIC patch that writes
that quantity into the colour output (the patch is applied to the assembled shader text in memory: a handful of `dbgOut = ...` assignments; nothing else
changes, and nothing is written to disk).  The restatement's trace comes from rfxo_set_trace.

    python tools/open_pixel_trace.py [frame] [y x]
"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "realism-effects_amd", os.path.join("oracle", "glref")):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import chain
import rfx_oracle as O
import stagewise as S
from rfx_amd.context import load_blue_noise_table
from rfx_amd.scene import synthetic_frame_parallel

FRAME = int(sys.argv[1]) if len(sys.argv) > 1 else 10
PY, PX = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (584, 676)
W, H, steps, refine, it = 1920, 1080, 40, 5, 3
blue = load_blue_noise_table()

# ---- the reference chain up to the frame (exactly as tests/stagewise.py _run drives it: the blue-noise index recurrences, camera_moved, keepData)
ref = chain.GLRefChain(W, H, blue, steps=steps, refineSteps=refine, denoiseIterations=it)
si = di = 0
prev_cam, keep = None, 0.0
for fi in range(FRAME + 1):
    f = synthetic_frame_parallel(W, H, fi)
    ref.upload_frame(f)
    sp, tp, dp, cp = S.stage_params(f.camera, prev_cam or f.camera, keep, steps, refine)
    si = (1000 + si + 1) % S.M31
    if fi == FRAME:
        break
    ref.ssgi(f.camera, si)
    ref.temporal(f.camera, camera_moved=True)
    for pi in range(2 * it):
        di = (2000 + di + 1) % S.M31
        S._one_denoise_pass(ref, f.camera, pi, di)
    ref.compose(f.camera)
    keep, prev_cam = 1.0, f.camera
history = np.ascontiguousarray(ref.t_compose.read())
sp.blueNoiseIndex = si

# ---- K1 of the frame, patched
src = chain.assemble_ssgi(steps, refine, 0, True, False)
def rep(old, new, count=1):
    global src
    assert src.count(old) >= 1, old[:60]
    src = src.replace(old, new, count)
rep("vec3 SampleGGXVNDF(const vec3 V,", "uniform int dbgSel; vec4 dbgOut = vec4(0.); bool dbgSpec = false;\nvec3 SampleGGXVNDF(const vec3 V,")
rep("  // specular ray (traced every frame)\n  l = specularRay;", "  dbgSpec = true;\n  l = specularRay;")
rep("  vec3 Nh = t1 * T1 + t2 * T2 + sqrt(max(0.0, 1.0 - t1 * t1 - t2 * t2)) * Vh;", "  vec3 Nh = t1 * T1 + t2 * T2 + sqrt(max(0.0, 1.0 - t1 * t1 - t2 * t2)) * Vh;\n  if (dbgSel == 20) dbgOut = vec4(r1, r2, t1, t2); if (dbgSel == 21) dbgOut = vec4(1.0 - t1 * t1 - t2 * t2, sqrt(max(0.0, 1.0 - t1 * t1 - t2 * t2)), s, Vh.z); if (dbgSel == 22) dbgOut = vec4(Nh, 0.); if (dbgSel == 25) dbgOut = vec4(T1, 0.);")
rep("  l = normalize(reflect(-V, H));\n  l = ToWorld(T, B, N, l);", "  if (dbgSel == 23) dbgOut = vec4(H, 0.); if (dbgSel == 26) dbgOut = vec4(V, 0.);\n  l = normalize(reflect(-V, H));\n  if (dbgSel == 24) dbgOut = vec4(l, dot(H, -V));\n  l = ToWorld(T, B, N, l);")
rep("  specularHitPos = hitPos;", "  specularHitPos = hitPos;\n  if (dbgSel == 0) dbgOut = vec4(viewPos, viewZ); if (dbgSel == 1) dbgOut = vec4(specularRay, random.b); if (dbgSel == 2) dbgOut = vec4(specularHitPos, 0.); if (dbgSel == 3) dbgOut = vec4(viewNormal, roughnessSq);")
rep("  gl_FragColor = packTwoVec4(gDiffuse, gSpecular);", "  gl_FragColor = dbgSel < 0 ? packTwoVec4(gDiffuse, gSpecular) : dbgOut;")
rep("    rayHitDepthDifference = z - hitPos.z;\n\n    if (rayHitDepthDifference >= 0.0 && rayHitDepthDifference < thickness) {",
    "    rayHitDepthDifference = z - hitPos.z;\n    if (dbgSpec && dbgSel == 10 + i) dbgOut = vec4(hitPos, cs); if (dbgSpec && dbgSel == 110 + i) dbgOut = vec4(uv, rayHitDepthDifference, z);\n\n    if (rayHitDepthDifference >= 0.0 && rayHitDepthDifference < thickness) {")
rep("    rayHitDepthDifference = z - hitPos.z;\n\n    dir *= 0.5;", "    rayHitDepthDifference = z - hitPos.z;\n    if (dbgSpec && dbgSel == 200 + i) dbgOut = vec4(hitPos, 0.); if (dbgSpec && dbgSel == 300 + i) dbgOut = vec4(uv, rayHitDepthDifference, z);\n\n    dir *= 0.5;")
ref.p_ssgi = chain.Program(src)
scratch = chain.Tex(W, H, chain.FMT_RGBA32F)
keep = ref.t_ssgi
def gl_value(sel):
    ref.t_ssgi = scratch
    ref.p_ssgi.set("dbgSel", int(sel))
    ref.ssgi(f.camera, si)
    ref.t_ssgi = keep
    return scratch.read()[PY, PX].astype(np.float32)

# ---- the restatement's trace of the same pixel on the same inputs
def oracle_trace(gl_prims):
    O.lib().rfxo_set_gl_exp(1 if gl_prims else 0)
    buf = np.zeros(5 * 400, np.float32)
    O.lib().rfxo_set_trace(PX, PY, buf.ctypes.data_as(C.c_void_p), buf.size)
    mask = np.zeros((H, W), bool)
    mask[PY, PX] = True
    with O.pixel_mask(mask):
        out = O.ssgi(f.depth, f.gbuffer, f.direct, history, blue, sp)
    n = O.lib().rfxo_trace_count()
    O.lib().rfxo_set_trace(-1, -1, None, 0)
    O.lib().rfxo_set_gl_exp(0)
    rec = buf[:n].reshape(-1, 5)
    return {int(r[0]): r[1:].copy() for r in rec}, out[PY, PX]


NAMES = {0: "viewPos.xyz, viewZ", 1: "specular ray, random.b", 2: "final hit position", 3: "view normal, roughness^2", 20: "VNDF: random.r, random.g, t1, t2", 21: "VNDF: q = 1 - t1^2 - t2^2, sqrt(q), s, Vh.z", 22: "VNDF: Nh", 23: "H (local)", 24: "l = normalize(reflect(-V, H)) local, dot(H, -V)", 25: "VNDF: T1", 26: "V (local)"}
def name(tag):
    if tag in NAMES: return NAMES[tag]
    if 10 <= tag < 110: return "march step %d: position, cs" % (tag - 10)
    if 110 <= tag < 200: return "march step %d: tap u, v, z - h, z" % (tag - 110)
    if 200 <= tag < 300: return "refine step %d: position" % (tag - 200)
    return "refine step %d: tap u, v, z - h, z" % (tag - 300)


def ulps(a, b):
    a, b = np.float32(a), np.float32(b)
    if a == b: return 0
    ia, ib = int(a.view(np.int32)), int(b.view(np.int32))
    if (ia < 0) != (ib < 0): return 1 << 30
    return abs(ia - ib)


for gl_prims in (False, True):
    tr, texel = oracle_trace(gl_prims)
    print("==== restatement with %s against the reference GL, pixel (y %d, x %d) of frame %d" % ("the reference GL's own exp / sin / cos" if gl_prims else "true exp / sin / cos", PY, PX, FRAME))
    first = None
    for tag in sorted(tr):
        g = gl_value(tag)
        o = tr[tag]
        u = [ulps(o[k], g[k]) for k in range(4)]
        texels = ""
        if 110 <= tag < 200 or tag >= 300:  # a tap: which depth texel each side addresses
            tg = (int(np.float32(g[0]) * np.float32(W)), int(np.float32(g[1]) * np.float32(H)))
            to = (int(np.float32(o[0]) * np.float32(W)), int(np.float32(o[1]) * np.float32(H)))
            fx = float(np.float32(o[0]) * np.float32(W)); fy = float(np.float32(o[1]) * np.float32(H))
            texels = "  texel %s%s, %.4f / %.4f texel from a boundary" % (to, "" if to == tg else " (GL: %s)" % (tg,), min(fx - int(fx), 1 - fx + int(fx)), min(fy - int(fy), 1 - fy + int(fy)))
        flag = "" if max(u) == 0 else "   <-- differs (ulps %s)" % u
        if max(u) and first is None:
            first = tag
        print("%-44s restatement %s  GL %s%s%s" % (name(tag), np.array2string(o, precision=7), np.array2string(g, precision=7), texels, flag))
    print("first difference: %s" % (name(first) if first is not None else "none"))
    g = chain  # noqa
