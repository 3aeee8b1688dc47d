/*
 * rfx.h — C ABI of librfx_hip.so: the MI355X-native SSGI hot path
 * (SSGI ray-march -> TemporalReprojectPass -> PoissonDenoisePass -> DenoiserComposePass).
 *
 * Drop-in boundary.  In the reference every pass is a postprocessing `Pass` whose only
 * device-side operation is
 *      renderer.setRenderTarget(rt); renderer.render(this.scene, this.camera)
 * with the arithmetic defined by the pass's `fullscreenMaterial` (GLSL + uniforms + defines):
 *      src/ssgi/pass/SSGIPass.js:93-94                          -> rfx_ssgi_march
 *      src/temporal-reproject/TemporalReprojectPass.js:192-193  -> rfx_temporal_reproject
 *      src/denoise/pass/PoissonDenoisePass.js:146-147           -> rfx_poisson_denoise
 *      src/denoise/pass/DenoiserComposePass.js:133-134          -> rfx_compose
 * plus the one non-draw device operation on the path,
 *      renderer.copyFramebufferToTexture(...)  TemporalReprojectPass.js:198-201  -> rfx_copy_framebuffer
 *      (the pass's own history when no override textures are set: TRAAEffect, src/traa/TRAAEffect.js:52-75).
 * Each entry point below replaces exactly one of those calls.  The `*_params` structs
 * carry what the material's `uniforms` (run-time values) and `defines` (shader variants)
 * carried; textures are addressed by slot id (`rfx_tex`), the analogue of the
 * `WebGLRenderTarget.texture` objects the passes share by reference (Denoiser.js:45,51).
 *
 * Conventions
 *   - plain C, no exceptions: every call returns 0 (RFX_OK) or a negative RFX_E* code;
 *     rfx_last_error(ctx) gives the message (the reference signals nothing, GL errors only
 *     surface on the console).
 *   - images are row-major, row 0 = BOTTOM (GL / `vUv` convention), tightly packed.
 *   - matrices are column-major float[16], i.e. three.js `Matrix4.elements`.
 *   - one context per device; calls enqueue on the context's HIP stream and return;
 *     rfx_sync() blocks.  A context may own only a horizontal band ("tile") of the frame:
 *     rows [tile_y0, tile_y0 + tile_rows) are written, textures additionally hold
 *     `halo_rows` rows above and below (clipped to the frame) for the gathers
 *     (SURVEY.md §8e).  Row indices in every call are FRAME rows.
 *   - device buffers are owned by the library unless bound with rfx_bind_external().
 *   - threading: a context is driven by ONE host thread at a time (the reference drives its passes from the JS main thread); different contexts
 *     may be driven from different threads.  The only state the library shares between contexts are per-device caches of launch constants
 *     (the occupancy figure K1's persistent grid is sized with, the dynamic-LDS attribute of K3's kernels) and the RCCL binding: two threads that
 *     first use a kernel specialisation at the same moment both compute and store the SAME value — an unsynchronised but idempotent write.
 */
#ifndef RFX_H
#define RFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_ABI_VERSION 19

enum {
    RFX_OK = 0,
    RFX_EINVAL = -1,   /* bad argument / unsupported option combination */
    RFX_ENOMEM = -2,   /* device allocation failed */
    RFX_EDEVICE = -3,  /* HIP runtime error (no device, launch failure, ...) */
    RFX_ESTATE = -4,   /* texture not uploaded / wrong size */
    RFX_EUNSUPPORTED = -5
};

/* Texture slots.  Formats follow SURVEY.md Appendix B (the reference's render-target formats,
 * which parity dictates for every intermediate). */
typedef enum rfx_tex {
    RFX_TEX_DEPTH = 0,      /* R32F      GBufferPass depth texture (GBufferPass.js:42-44)           */
    RFX_TEX_GBUFFER,        /* RGBA32F   packed material (gbuffer_packing.glsl:166-178)             */
    RFX_TEX_VELOCITY,       /* RGBA32F   VelocityDepthNormalPass output (…Material.js:76-83,186-188)*/
    RFX_TEX_DIRECT_LIGHT,   /* RGBA32F   composer input buffer = direct lighting (SSGIEffect.js:396)*/
    RFX_TEX_BLUE_NOISE,     /* RGBA8 128x128, repeat (BlueNoiseUtils.js:9-15), already flipY'd      */
    RFX_TEX_SSGI,           /* RGBA32F   K1 out: 8 halfs {diffuse.rgb,roughness | specular.rgb,rayLength} */
    RFX_TEX_TEMPORAL0,      /* RGBA32F   K2 out 0 (diffuse), .a = age                               */
    RFX_TEX_TEMPORAL1,      /* RGBA32F   K2 out 1 (specular)                                        */
    RFX_TEX_DENOISE_A0,     /* RGBA16F   K3 ping-pong target A                                      */
    RFX_TEX_DENOISE_A1,
    RFX_TEX_DENOISE_B0,     /* RGBA16F   K3 ping-pong target B = K2 history = K4 input              */
    RFX_TEX_DENOISE_B1,
    RFX_TEX_COMPOSE,        /* RGBA32F   K4 out = next frame's K1 `accumulatedTexture`              */
    RFX_TEX_FBCOPY_F16,     /* RGBA16F linear   TemporalReprojectPass.framebufferTexture (:137-142) when the pass's    */
    RFX_TEX_FBCOPY_F32,     /* RGBA32F linear   input / render target is HalfFloatType resp. FloatType (:66,139-140)   */
    RFX_TEX_FINAL,          /* RGBA32F   SSGIEffect's own fragment (ssgi_compose.frag mainImage): the effect's output colour */
    RFX_TEX_COMPOSE_RGB,    /* RGB32F    .rgb of RFX_TEX_COMPOSE as 12-byte texels, held whole: the part of `accumulatedTexture` K1 reads,
                               kept beside it (rfx_compose_params.writeHistoryRGB) so that a row-tiled run all-gathers 12 B/px, not 16 */
    RFX_TEX_COUNT
} rfx_tex;

/* What the passes read from `this._camera` each frame. */
typedef struct rfx_camera {
    float projectionMatrix[16];
    float projectionMatrixInverse[16];
    float matrixWorld[16];        /* cameraMatrixWorld */
    float matrixWorldInverse[16]; /* viewMatrix        */
    float position[3];            /* cameraPos (TemporalReprojectPass.js:98) */
    float near_, far_;
    int32_t isPerspective;        /* camera.isPerspectiveCamera -> the PERSPECTIVE_CAMERA define of every pass (1), else the orthographic depth -> view-Z variants (0) */
} rfx_camera;

/* K1 — SSGIMaterial uniforms/defines (src/ssgi/material/SSGIMaterial.js:15-51,
 * SSGIPass.js:38-40,82-91, SSGIEffect.js:143-151,203-226,254-258). */
typedef struct rfx_ssgi_params {
    rfx_camera camera;
    int32_t steps;            /* #define steps        (default 20) */
    int32_t refineSteps;      /* #define refineSteps  (default 5)  */
    int32_t mode;             /* #define mode: 0 = MODE_SSGI (two packed vec4 of halfs), 1 = MODE_SSR (raw vec4: specular GI, packHalf2x16(rayLength, roughness)) */
    int32_t useDirectLight;   /* #define useDirectLight */
    int32_t missedRays;       /* #define missedRays   */
    int32_t importanceSampling; /* #define importanceSampling (env-map MIS, ssgi.frag:197-216): needs useEnvMap and the tables of
                                 rfx_set_environment_importance                                                              */
    int32_t useEnvMap;        /* #define USE_ENVMAP: the context holds scene.environment (rfx_set_environment); missed rays and
                                 the screen-border fade take its colour instead of black (getEnvColor, ssgi.frag:311-346)   */
    float rayDistance;        /* uniform rayDistance = options.distance */
    float thickness;
    float envBlur;            /* uniform envBlur: mip = envBlur * maxEnvMapMipLevel (ssgi.frag:322); unused without USE_ENVMAP */
    int32_t blueNoiseIndex;   /* uniform blueNoiseIndex (BlueNoiseUtils.js:24-32 recurrence, host side) */
    float resolutionScale;    /* SSGIPass.setSize (SSGIPass.js:52-57): the pass renders into a (W*s) x (H*s) target and `resolution` is that size;
                                 0 is read as 1.  W*s and H*s must be whole numbers; s != 1 needs a whole-frame context.  The target's
                                 texels are stored row-major at the start of RFX_TEX_SSGI (pitch W*s).                               */
    int32_t historySource;    /* uniform accumulatedTexture = ssgiEffect.denoiser.texture (SSGIPass.js:89, Denoiser.js:67-78):
                                 0  denoiseMode "full" / "full_temporal": K4's output, RFX_TEX_COMPOSE;
                                 1  "temporal": K2's texture[0], RFX_TEX_TEMPORAL0 (whole-frame contexts only);
                                 2  "denoised": the getter returns an ARRAY of textures, which three binds as its empty
                                    texture -> every history fetch reads (0,0,0,0);
                                 3  as 0, read from RFX_TEX_COMPOSE_RGB (same values: rfx_compose_params.writeHistoryRGB)    */
} rfx_ssgi_params;

/* K2 — TemporalReprojectMaterial uniforms/defines (TemporalReprojectPass.js:76-117,162-214). */
typedef struct rfx_temporal_params {
    rfx_camera camera;
    rfx_camera prevCamera;        /* prevViewMatrix, prevCameraMatrixWorld, prevProjectionMatrix(+Inverse), prevCameraPos */
    int32_t textureCount;         /* 2 (diffuseSpecular) or 1 */
    int32_t inputType;            /* 0 DIFFUSE_SPECULAR, 1 DIFFUSE, 2 SPECULAR */
    int32_t reprojectSpecular[2]; /* define bool[] */
    int32_t neighborhoodClamp[2]; /* define bool[]; accepted, the shader never reads it (Appendix D-6) */
    int32_t logTransform;
    int32_t fullAccumulate;       /* uniform: option && !didCameraMove */
    float confidencePower;        /* define, toPrecision(5) */
    float neighborhoodClampIntensity;
    float maxBlend;
    float keepData;               /* 0 for the first frame after reset(), else 1 */
    int32_t historySource;        /* which textures `accumulatedTexture[i]` are (TemporalReprojectPass.js:148-151):
                                     0  overrideAccumulatedTextures = K3's target B, RFX_TEX_DENOISE_B0/B1 (Denoiser.js:51);
                                     1  the pass's own framebuffer copy RFX_TEX_FBCOPY_F16 (render-target type HalfFloatType);
                                     2  the same, RFX_TEX_FBCOPY_F32 (FloatType).  With two textures (denoiseMode
                                        "full_temporal" / "temporal") BOTH read the one copy, which holds colour attachment 0
                                        = texture 0: copyFramebufferToTexture reads the framebuffer's read buffer.            */
    int32_t targetHalf;           /* render target type = type of the input texture (:63-68): 0 FloatType — texels stored as
                                     computed; 1 HalfFloatType — every output channel is rounded to half precision on store
                                     (RFX_TEX_TEMPORAL* then hold half-representable floats)                                 */
    int32_t halfStoreRTZ;         /* rounding of that store: 1 truncate (llvmpipe, parity with the oracle), 0 nearest-even    */
    int32_t inputWidth, inputHeight; /* size of `inputTexture` when it is smaller than the frame (K1 drawn with resolutionScale < 1:
                                     the pass samples it NEAREST at full-resolution vUv, TemporalReprojectPass.js:118); 0 = frame size */
} rfx_temporal_params;

/* K3 — PoissonDenoisePass uniforms/defines (PoissonDenoisePass.js:43-71, SSGIEffect.js:175-190). */
typedef struct rfx_denoise_params {
    float radius, phi, lumaPhi, depthPhi, normalPhi, roughnessPhi, specularPhi;
    int32_t textureCount;          /* 2 or 1 */
    int32_t isTextureSpecular[2];  /* define bool[2] */
    int32_t blueNoiseIndex;        /* advances once per draw */
    int32_t inputIsTemporal;       /* 1: inputs = RFX_TEX_TEMPORAL* (RGBA32F, nearest) — pass 0;
                                      0: inputs = the other ping-pong target (RGBA16F, linear)   */
    int32_t writeToB;              /* 0: render into A (even pass index), 1: into B (odd)       */
    int32_t halfStoreRTZ;          /* 1: RGBA16F stores truncate like llvmpipe's colour-buffer
                                         store (parity with the oracle); 0: round-to-nearest-even
                                         like GPU ROPs                                          */
} rfx_denoise_params;

/* K4 — DenoiserComposePass uniforms/defines (DenoiserComposePass.js:87-110). */
typedef struct rfx_compose_params {
    rfx_camera camera;
    int32_t inputType; /* 0 TYPE_DIFFUSE_SPECULAR; 2 TYPE_SPECULAR (diffuse component = sceneTexture = RFX_TEX_DIRECT_LIGHT, specular GI = B0) */
    int32_t giSource;  /* composerInputTextures (Denoiser.js:53): 0 the denoise pass's textures, RFX_TEX_DENOISE_B0/B1 (RGBA16F);
                          1 denoiseMode "full_temporal": K2's textures, RFX_TEX_TEMPORAL0/1 (RGBA32F, nearest) */
    int32_t writeHistoryRGB; /* 1: also keep RFX_TEX_COMPOSE_RGB = .rgb of every tile texel of RFX_TEX_COMPOSE (discarded background
                                fragments copy what the target holds), for rfx_ssgi_params.historySource 3 */
} rfx_compose_params;

/* SSGIEffect's own fragment — FinalSSGIMaterial uniforms/defines (SSGIEffect.js:34-66,404-417; src/ssgi/shader/ssgi_compose.frag). */
typedef struct rfx_final_params {
    rfx_camera camera;     /* cameraNear / cameraFar / PERSPECTIVE_CAMERA (only read when fog is on) */
    int32_t isDebug;       /* uniform isDebug: output = inputTexture texel, unmodified */
    int32_t inputSource;   /* uniform inputTexture = outputTexture[0] ?? outputTexture = denoiser.texture (SSGIEffect.js:139,402):
                              0 RFX_TEX_COMPOSE ("full", "full_temporal"); 1 RFX_TEX_TEMPORAL0 ("temporal"); 2 RFX_TEX_DENOISE_B0 ("denoised") */
    int32_t fogMode;       /* 0: scene.fog unset; 1: THREE.Fog (USE_FOG); 2: THREE.FogExp2 (USE_FOG + FOG_EXP2) */
    float fogColor[3];
    float fogNear, fogFar; /* fogMode 1 */
    float fogDensity;      /* fogMode 2 */
} rfx_final_params;

typedef struct rfx_ctx rfx_ctx;

/* ---- lifetime (Pass ctor / setSize / dispose) */
int rfx_abi_version(void);
rfx_ctx *rfx_create(int device, int width, int height, int tile_y0, int tile_rows, int halo_rows);
void rfx_destroy(rfx_ctx *);
const char *rfx_last_error(const rfx_ctx *);
/* Geometry the context was created with (any out pointer may be NULL). */
int rfx_get_geometry(const rfx_ctx *, int *width, int *height, int *tile_y0, int *tile_rows, int *halo_rows);
/* Run the context's kernels on a caller-provided hipStream_t (e.g. a stream the framework also uses for its
 * collectives, so that they are ordered against the kernels); NULL restores the context's own stream.  The legacy
 * default stream has handle 0 == NULL and therefore cannot be selected: create a stream.  Switching drains the
 * stream used so far (its uploads and zero-fills must not race kernels on the new one): set it once, not per frame. */
int rfx_set_stream(rfx_ctx *, void *hip_stream);
/* Restrict the rows the following draws PRODUCE to frame rows [y0, y1) (intersected with what each draw would produce anyway);
 * y1 <= y0 resets.  Every pixel's result is independent of how the rows are split over launches, so a row-tiled run can draw the
 * interior of its tile while the halo rows of the input are still being exchanged, then the two boundary strips (rfx_amd/tiling.py).
 * Draws whose window is empty return RFX_OK without launching.  Ignored by rfx_ssgi_* with resolutionScale != 1 (whole-frame only). */
int rfx_set_row_window(rfx_ctx *, int y0, int y1);
/* Which vUv the draws' fragments see (every full-screen pass of the reference reads its inputs at the interpolated varying vUv,
 * src/utils/shader/basic.vert; e.g. ssgi.frag:107, temporal_reproject.frag:118, poisson_denoise.frag:128, DenoiserComposePass.js:58).
 *   RFX_UV_REFERENCE_GL  (default since ABI 15) the value the rasteriser of the reference's GL (Mesa llvmpipe, the oracle of the parity
 *                        tests) interpolates, bit for bit: three's full-screen triangle is clipped into two triangles along the frame
 *                        diagonal, each with its own fp32 plane equations (up to 2^-24 from the ideal value, different on either side of
 *                        the diagonal).  With it the NEAREST taps of the denoiser and every LINEAR fetch at vUv land where the
 *                        reference's land: what is left between the two implementations is transcendental rounding alone.
 *   RFX_UV_IDEAL         (i + 0.5) / n, correctly rounded: the value the shader authors mean, and what a GL that does not clip the
 *                        triangle is closest to.  Costs two IEEE divisions per fragment instead of an fma.
 * Applies to every following draw; row-tiled contexts evaluate the whole frame's planes. */
enum { RFX_UV_IDEAL = 0, RFX_UV_REFERENCE_GL = 1 };
int rfx_set_uv_model(rfx_ctx *, int model);
/* (ABI 17-18 had rfx_set_compose_fold: the compose draw made inside the last denoise launch from the texel just stored — an approximation of
 * the reference's LINEAR fetch.  Removed in ABI 19: an exact fold needs a second launch for the tile-edge pixels and saves at most 0.008 ms of
 * a 1.25 ms 4K frame before its LDS exchange is paid for (profiles/r06_k4/exact_fold_bounds.txt), and the library keeps ONE parity contract:
 * one launch per draw, exactly the reference's fetches.) */

/* ---- textures.  `row0`/`rows` are FRAME rows of the band being transferred; the band must lie
 * inside the rows the context holds: [max(0,tile_y0-halo), min(H,tile_y0+tile_rows+halo)).
 * Read-only full-frame inputs of K1 (depth, compose history) are always held whole. */
size_t rfx_tex_texel_bytes(rfx_tex id);
int rfx_tex_held_rows(const rfx_ctx *, rfx_tex id, int *row0, int *rows);
int rfx_upload(rfx_ctx *, rfx_tex id, const void *host, int row0, int rows);
int rfx_download(rfx_ctx *, rfx_tex id, void *host, int row0, int rows);
int rfx_clear(rfx_ctx *, rfx_tex id); /* zero-fill (render targets start zeroed) */
/* Device pointer of the first HELD row of a slot (for halo exchange / zero-copy interop).  Work the caller enqueues on the buffer must be
 * ordered against the context's draw stream (rfx_set_stream).  Taking RFX_TEX_DEPTH's pointer also moves K1's depth pre-pass from its own
 * stream into the draw stream for the rest of the context's life, exactly as rfx_bind_external(RFX_TEX_DEPTH) does. */
void *rfx_tex_device_ptr(rfx_ctx *, rfx_tex id);
/* Use caller-owned device memory (held-rows x width x texel bytes) for a slot. */
int rfx_bind_external(rfx_ctx *, rfx_tex id, void *device_ptr);

/* ---- importer: engine-side attribute planes (AOVs) instead of pre-packed dumps.  The two calls stand where the raster passes' fragment
 * epilogues pack their render targets (GBufferMaterial.js:84-89 `packGBuffer(diffuseColor, worldNormal, roughnessFactor, metalnessFactor,
 * totalEmissiveRadiance)`; VelocityDepthNormalMaterial.js:76-83,186-188 `vec4(vel.xy, packNormal(worldNormal), fragCoordZ)`) and fill rows
 * [row0, row0+rows) of RFX_TEX_GBUFFER / RFX_TEX_VELOCITY on the device (encode side of gbuffer_packing.glsl).  Planes are host pointers to
 * `rows` x width tightly packed float32 texels, row 0 of the band first (bottom up).  Texels with depth == 1 keep the passes' clear colour
 * (0, 0, 0, 1) (GBufferPass.js:103-105). */
typedef struct rfx_aov_gbuffer {
    const float *diffuse;    /* RGBA  diffuseColor (material colour x map, alpha)                     */
    const float *normal;     /* xyz   WORLD-space normal (normalised by the producer; re-scaled by the oct encoder) */
    const float *roughness;  /* 1     roughnessFactor                                                 */
    const float *metalness;  /* 1     metalnessFactor                                                 */
    const float *emissive;   /* rgb   totalEmissiveRadiance                                           */
    const float *depth;      /* 1     gl_FragCoord.z, 1.0 = not covered; may be NULL (every texel covered) */
} rfx_aov_gbuffer;
typedef struct rfx_aov_velocity {
    const float *velocity;   /* xy    uv-space motion: pos1 - pos0 with pos = clip.xy / clip.w * .5 + .5 (VelocityDepthNormalMaterial.js:76-80) */
    const float *normal;     /* xyz   WORLD-space normal                                              */
    const float *depth;      /* 1     gl_FragCoord.z, 1.0 = not covered                               */
} rfx_aov_velocity;
int rfx_pack_gbuffer(rfx_ctx *, const rfx_aov_gbuffer *, int row0, int rows);
int rfx_pack_velocity(rfx_ctx *, const rfx_aov_velocity *, int row0, int rows);

/* ---- scene.environment (SSGIEffect.keepEnvMapUpdated, SSGIEffect.js:309-362): an equirectangular HDR map.  The effect
 * turns its mipmaps on (`generateMipmaps`, LinearMipMapLinearFilter / LinearFilter, :323-328) and K1 samples it with
 * textureLod(map, equirectDirectionToUv(dir), envBlur * maxEnvMapMipLevel).  rfx_set_environment takes the base level as
 * width x height RGBA float32 texels (row 0 = bottom, v = 0), rounds them to half precision when the texture's type is
 * HalfFloatType (`halfFloatType`, what RGBELoader produces), and builds the mip chain the way glGenerateMipmap does on
 * the oracle's GL: every level the 2x2 bilinear-centre average of the one above, lerp(.5, lerp(.5,a,b), lerp(.5,c,d)), stored
 * in the texture's type (half: `halfStoreRTZ` 1 truncates like llvmpipe, 0 rounds to nearest even).  width and height must
 * be powers of two (wrap: ClampToEdge, three's default for a DataTexture).  maxEnvMapMipLevel = floor(log2(max(w,h))) + 1
 * (src/ssgi/utils/Utils.js:30-34).  rfx_set_environment(ctx, NULL, 0, 0, 0, 0) removes it. */
int rfx_set_environment(rfx_ctx *, const float *rgba, int width, int height, int halfFloatType, int halfStoreRTZ);
/* The two inverse-CDF tables and the luminance sum that EquirectHdrInfoUniform.updateFrom computes on the CPU (a Web Worker in the
 * reference, src/ssgi/utils/EquirectHdrInfoUniform.js:148-245,365-400) for `sampleEquirectProbability` (ssgi_utils.frag:210-225):
 * `marginalWeights` (height floats: an height x 1 NEAREST texture), `conditionalWeights` (width x height floats, row-major), and
 * totalSumValue split as the reference splits it (`~~total` and the rest).  Sizes are those of the environment set before. */
int rfx_set_environment_importance(rfx_ctx *, const float *marginalWeights, size_t marginalCount, const float *conditionalWeights,
                                   size_t conditionalCount, float totalSumWhole, float totalSumDecimal);  /* counts in floats: RFX_EINVAL unless height / width*height */
/* Read mip level `level` of the environment back (max(w>>level,1) x max(h>>level,1) RGBA float32); *levels (may be NULL) receives the
 * number of levels.  For inspection and for checking the chain against the driver's. */
int rfx_download_environment(rfx_ctx *, int level, float *rgba, int *levels);
/* CubeToEquirectEnvPass.generateEquirectEnvMap's draw + read-back (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42,78-85; called from
 * SSGIEffect.js:316-321 when scene.environment is a CubeTexture): `faces_rgba` = the six faces +X -X +Y -Y +Z -Z, each size x size RGBA
 * float32 (linear values), row j = t as handed to glTexImage2D; `equirect_rgba` receives width x height RGBA float32 texels (the pass's
 * FloatType render target as readRenderTargetPixels returns it: row 0 = bottom) — the DataTexture the effect then treats like any
 * equirectangular environment: hand it to rfx_set_environment(…, halfFloatType = 0) and build the importance tables from it.
 * generateMipmaps = 0: the cube is sampled LINEAR, seamless, at level 0 (minFilter LinearFilter: HDRCubeTextureLoader's set-up);
 * 1: three's CubeTexture default, LinearMipmapLinearFilter over the chain glGenerateMipmap builds (any size since round 6: an odd level is
 * reduced by the GL's bilinear blit, an even one by the 2x2 average it degenerates to), with the implicit level of detail of `textureCube`
 * (the pass minifies near the face edges: levels 0-2 take part). */
int rfx_cube_to_equirect(rfx_ctx *, const float *faces_rgba, int size, int generateMipmaps, float *equirect_rgba, int width, int height);

/* ---- the four draws (+ the framebuffer copy and the effect's own fragment) */
int rfx_ssgi_march(rfx_ctx *, const rfx_ssgi_params *);
/* The same draw in two launches, for a row-tiled run: rfx_ssgi_trace runs the fragment up to the end of RayMarch/BinarySearch
 * (ssgi.frag:441-503) and keeps the rays' end state in context scratch (32 B per pixel); rfx_ssgi_shade finishes the fragment
 * (doSample's shading :385-439 and the output packing) from it.  Only the shading reads `accumulatedTexture` (RFX_TEX_COMPOSE or
 * RFX_TEX_TEMPORAL0, historySource) — gathered anywhere on screen — so the all-gather that refreshes it between frames may still be in
 * flight during the trace and has to have landed only before the shade (rfx_amd/tiling.py).  Same params for both calls; the
 * result in RFX_TEX_SSGI is bit-identical to rfx_ssgi_march's.  rfx_ssgi_shade without a pending trace: RFX_ESTATE. */
int rfx_ssgi_trace(rfx_ctx *, const rfx_ssgi_params *);
int rfx_ssgi_shade(rfx_ctx *, const rfx_ssgi_params *);
int rfx_temporal_reproject(rfx_ctx *, const rfx_temporal_params *);
/* renderer.copyFramebufferToTexture(tmpVec2, this.framebufferTexture), TemporalReprojectPass.js:198-201: the tile rows of
 * the pass's render target RFX_TEX_TEMPORAL0 become the history the NEXT rfx_temporal_reproject samples (linear filter).
 * `dst` is RFX_TEX_FBCOPY_F16 (source texels must be half-representable, i.e. drawn with targetHalf = 1: the copy is then
 * exact, as in the reference where both sides have the same type) or RFX_TEX_FBCOPY_F32. */
int rfx_copy_framebuffer(rfx_ctx *, rfx_tex dst);
/* One PoissonDenoisePass draw. */
int rfx_poisson_denoise(rfx_ctx *, const rfx_denoise_params *);
int rfx_compose(rfx_ctx *, const rfx_compose_params *);
/* The effect's mainImage (src/ssgi/shader/ssgi_compose.frag:20-45), which postprocessing's EffectPass runs after
 * SSGIEffect.update(): background texels take the scene colour (RFX_TEX_DIRECT_LIGHT = the composer's input buffer),
 * the rest the composed GI (RFX_TEX_COMPOSE), fogged when the scene has fog; alpha 1.  Writes RFX_TEX_FINAL. */
int rfx_final_compose(rfx_ctx *, const rfx_final_params *);

int rfx_sync(rfx_ctx *);

/* ---- streaming dumps: host buffers that cross PCIe every frame (an offline run over a dumped sequence).  rfx_upload is synchronous
 * (the caller may free the plane on return).  The streaming form double-buffers the four input planes of the dump (depth, gbuffer,
 * velocity, direct light): rfx_stage_upload enqueues the copy of the NEXT frame's plane into the slot's back buffer on the context's
 * upload stream and returns; rfx_stage_flip makes everything staged since the last flip current — draws enqueued after it wait for
 * those copies, and copies staged after it wait for the draws enqueued before it (they overwrite the buffer those draws read).
 *     stage(frame 0); flip();   loop: stage(frame n+1); draws of frame n; flip()
 * `host` must stay valid and unchanged until the flip AFTER the one that publishes it has returned: rfx_stage_flip returns only when
 * the copies published by the previous flip have executed (host-side back pressure — a host with two alternating sets of pinned planes
 * can refill a set as soon as the next flip is back, and never runs more than two frames ahead of the device).  Pinned memory
 * (rfx_host_alloc = hipHostMalloc) is what makes the copy asynchronous; a pageable plane is accepted and simply does not overlap. */
void *rfx_host_alloc(size_t bytes);
void rfx_host_free(void *);
int rfx_stage_upload(rfx_ctx *, rfx_tex id, const void *host, int row0, int rows);
int rfx_stage_flip(rfx_ctx *);

/* ---- row-tiled runs: the exchanges, one process per GPU, RCCL over xGMI.  RCCL is bound at run time (a
 * single-GPU host needs none; a process that already maps an RCCL — e.g. torch's — shares it).
 * Tiles: rank r of n owns rfx_split_rows(height, n, r) — boundaries on even rows, the last tile takes the remainder.  The exchanges run on a second stream of the context: each call orders itself AFTER all draws
 * enqueued so far and returns; rfx_comm_wait orders all LATER draws after the exchanges issued so far.  In between the host may
 * enqueue draws that do not touch the rows in flight (the tile interior through rfx_set_row_window; rfx_ssgi_trace while the
 * composed GI is gathered), which is how their time is hidden.
 * Replaces nothing in the reference (it has no multi-GPU path); it is the halo step north_star / SURVEY.md §8e prescribe. */
int rfx_split_rows(int height, int nranks, int rank, int *tile_y0, int *tile_rows);
int rfx_comm_unique_id(void *id128);                                            /* ncclGetUniqueId: 128 bytes, rank 0 hands them to the others */
int rfx_comm_init(rfx_ctx *, const void *id128, int rank, int nranks);          /* ncclCommInitRank on the context's device (collective) */
int rfx_comm_destroy(rfx_ctx *);
/* ncclGroupStart; ncclSend/ncclRecv of halo_rows rows of texture `id` with the tile above (`up_rank`, higher frame rows) and
 * below (`down_rank`); ncclGroupEnd — -1 = no such neighbour.  `ncclComm` (an ncclComm_t) may be NULL: the context's own.
 * halo_rows taller than the split's tiles (many ranks, a fast camera: the band around a tile then reaches past its neighbours): every
 * tile whose rows lie inside this tile's band sends them directly, in the same group — that form needs rfx_comm_init on the context
 * (rank and size) and up / down = rank + 1 / rank - 1 (or -1). */
int rfx_halo_exchange(rfx_ctx *, rfx_tex id, void *ncclComm, int up_rank, int down_rank);
/* every rank's tile rows of RFX_TEX_COMPOSE or RFX_TEX_COMPOSE_RGB (held whole) to every rank, in place: next frame's K1 gathers
 * last frame's composed GI anywhere on screen.  ncclAllGather (equal tiles) or one ncclBroadcast per owner in a group (ragged). */
int rfx_allgather_history(rfx_ctx *, rfx_tex id, void *ncclComm);
/* The bounded form of that gather (ABI 15), called BETWEEN rfx_ssgi_trace and rfx_ssgi_shade: only the shading of a ray reads last
 * frame's composed GI, NEAREST at the ray's final uv (ssgi.frag:396-427), and after the trace those uvs are known.  The call reduces, on
 * the device, the ROW MASK of the history texels this tile's rays will read (ABI 16: one word per frame row, see rfx_ssgi_hit_mask; ABI 15
 * reduced one (min, max) row interval, which also moved every row between two needed ones), all-gathers the N masks, and moves with grouped
 * ncclSend / ncclRecv only the rows some rank needs from their owners (whose rows are current: K4 wrote them), in runs of consecutive
 * rows — instead of every rank receiving the other N - 1 tiles.  Bit-identical to the all-gather form.  It waits on the host for the N
 * masks (height words each; i.e. for the trace to finish); the row
 * transfers are asynchronous like the other exchanges (rfx_comm_wait before rfx_ssgi_shade).  `bytes_received` (may be NULL): what this
 * rank receives this frame.  A host uses EITHER this or rfx_allgather_history after K4. */
int rfx_gather_history_rows(rfx_ctx *, rfx_tex id, void *ncclComm, size_t *bytes_received);
/* The DEVICE-DRIVEN form of the bounded gather (ABI 19; csrc/rfx_peer.hip): no RCCL, no host wait, no packing.  xGMI lets a kernel load a
 * peer GPU's memory, so every rank exports the plane once (rfx_peer_export: RFX_PEER_BLOB_BYTES bytes holding HIP IPC handles of the plane and
 * of a small flag block; the host moves the N blobs to every rank by any means it has — they are plain bytes) and opens its peers'
 * (rfx_peer_open: `blobs` = the N blobs in rank order; the context's tile must be rfx_split_rows(height, nranks, rank)).  Then, per frame,
 * BETWEEN rfx_ssgi_trace and rfx_ssgi_shade and on EVERY rank, rfx_peer_gather_history enqueues on the exchange stream: a flag barrier through
 * the peers' mapped flag blocks ("every rank's compose draw of the previous frame has executed"), ONE kernel that walks this rank's own row
 * mask (rfx_ssgi_hit_mask's, already on the device) and copies the column blocks its rays will read straight out of their owners' planes, and
 * a second barrier ("every rank has pulled") that the next rfx_compose — which overwrites those rows — is ordered after.  The shade is ordered
 * after the pull by rfx_comm_wait, like the other exchanges.  Bit-identical to the all-gather form.  Nothing waits on the host:
 * `bytes_pulled_previous_call` (may be NULL) reports what the PREVIOUS call's kernel moved, and a peer that never reached a barrier (~2 s)
 * surfaces as RFX_EDEVICE from the next call instead of hanging the device.  The plane must be the library's own (not rfx_bind_external).
 * Contexts of one process (one per device, peer access enabled by the host) are recognised by the blob's process id and use each other's
 * addresses directly.  (Several contexts of one process on ONE device — a test set-up — work too, as long as their exchange streams sit on
 * different hardware queues: the barrier kernels of the ranks must be resident together, and the HIP runtime multiplexes a process's streams
 * onto GPU_MAX_HW_QUEUES (4) queues per device; with more streams than queues two barrier kernels can queue behind each other, and the call
 * after the bounded poll reports RFX_EDEVICE.  Set GPU_MAX_HW_QUEUES >= 4 x the contexts before the first HIP call.)  A host uses ONE of rfx_allgather_history / rfx_gather_history_rows / rfx_peer_gather_history. */
#define RFX_PEER_BLOB_BYTES 192
int rfx_peer_export(rfx_ctx *, rfx_tex id, void *blob);
int rfx_peer_open(rfx_ctx *, rfx_tex id, const void *blobs, int rank, int nranks);
int rfx_peer_gather_history(rfx_ctx *, rfx_tex id, size_t *bytes_pulled_previous_call);
int rfx_peer_close(rfx_ctx *);
/* The reduction on its own (any context, no communicator): after rfx_ssgi_trace, the inclusive range of history rows the shade of the
 * traced rows will read; row_hi < row_lo when it reads none.  Blocks until the trace has finished. */
int rfx_ssgi_hit_rows(rfx_ctx *, int *row_lo, int *row_hi);
/* ... and the mask form (ABI 16): row_mask[y], y in [0, height), gets bit b set when the shade of the traced rows reads a history texel of
 * frame row y in column block b (32 equal blocks across the frame: texel x is in block x * 32 / width); a row that is not read at all
 * gets 0.  `rows` must be the frame height.  Blocks until the trace has finished. */
int rfx_ssgi_hit_mask(rfx_ctx *, unsigned int *row_mask, int rows);
int rfx_comm_wait(rfx_ctx *);

/* Number of texel fetches that fell outside the rows a tile context holds since creation
 * (they are clamped into the band, i.e. the halo was too small for the frame's motion/radius).
 * 0 on a correct configuration; always 0 for a whole-frame context. */
unsigned int rfx_halo_violations(rfx_ctx *);

/* Timing helper: run `fn`-selected kernel `iters` times between two hipEvents on the context's
 * stream and return the mean milliseconds (used by bench.py for the roofline line). */
int rfx_time_begin(rfx_ctx *);
int rfx_time_end(rfx_ctx *, float *elapsed_ms);
/* Per-draw device timing INSIDE a frame loop (ABI 18; bench.py's `kernel_ms` and `roofline`).  rfx_profile(ctx, 1) resets the sums and makes every
 * draw entry point bracket the launches it makes with two hipEvents on the stream they run on (K1's depth pre-pass on its own stream, where it
 * overlaps the previous frame's later draws); rfx_profile(ctx, 0) stops.  rfx_profile_read waits for the recorded events and returns, per kind,
 * the summed milliseconds and the number of launches since the reset (arrays of RFX_PROF_COUNT entries; either may be NULL).  The events cost
 * a few microseconds per draw: the frame's own time (`value`) is measured without them.  At most 8192 launches are recorded per reset. */
enum { RFX_PROF_K1_PREPASS = 0, RFX_PROF_K1_MARCH, RFX_PROF_K2, RFX_PROF_K3_PASS0, RFX_PROF_K3_PASSN, RFX_PROF_K4, RFX_PROF_K5,
       RFX_PROF_COUNT };
int rfx_profile(rfx_ctx *, int enable);
int rfx_profile_read(rfx_ctx *, float *ms_sum, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* RFX_H */
