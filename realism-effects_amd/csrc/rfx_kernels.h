// rfx_kernels.h — launch-argument blocks and launcher prototypes of the four kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "rfx_device.h"

// Rows [y0, y1) of the frame are produced by a launch (the context's tile, plus whatever extra
// rows the host asks for, e.g. K1's +-2 rows that K2's neighbourhood clamp reads).

#ifndef RFX_K1_POW2
#define RFX_K1_POW2 1  // build knob: the march's (min, max) table gets a power-of-two row pitch where that costs no cell size, so that a tap's LDS address
                       // is two shifts and one v_bitop3_b32 (k1_tap_at).  0 = always rows of cells_w cells: two shifts, a 24-bit multiply-add and a shift
                       // (A/B measurements; same texels)
#endif
struct K1Args {
    FrameDims dims;
    int y0, y1;
    TexView depth, gbuffer, direct, history;  // history = K4 output of the previous frame (RGBA32F, nearest)
    const void *blue;
    int shift_x, shift_y;
    TexViewW out;  // RGBA32F holding 8 halfs
    rfx_ssgi_params p;
    float nearMulFar, farMinusNear, nearMinusFar;
    float *viewz;    // full-frame view-space Z (context scratch, filled by k1_prepare)
    float2 *coarse;  // exact (min, max) view Z per 16x16-texel base cell (k1_prepare)
    int coarse_w, coarse_h;
    unsigned int *cells;  // the march's table: two halfs per 2^cell_shift-texel cell, padded to whole uint4s (k1_pack_cells)
    int cells_w, cells_h, cell_shift, cells_vec4;
    int cells_pitch, cells_pitch_log2;  // cells per table row: cells_w, or (cells_pow2) the next power of two, at least 2^cell_shift — k1_tap_at
    int cells_pow2;                     // the table's rows are padded to a power of two (rfx_api: when that fits at the same cell size)
    // scene.environment: all mip levels as float4 texels, level l (max(w>>l,1) x max(h>>l,1)) at env + env_off[l]
    const float4 *env;
    int env_w, env_h, env_levels;
    unsigned int env_off[16];
    float maxEnvMapMipLevel;
    const float *env_marginal, *env_conditional;  // EquirectHdrInfo.marginalWeights (env_h) / conditionalWeights (env_w x env_h), importanceSampling
    float totalSumWhole, totalSumDecimal;
    int out_w, out_h;  // the pass's render target = `resolution` (frame size unless resolutionScale != 1)
    UvPlanes out_uv;   // that target's vUv
    float4 *hits;      // trace -> shade hand-over (2 texels per output pixel, indexed like `out`); null for the fused launch
    unsigned int *tile_counter;  // the persistent march kernel's work counter (context scratch; zero when the launch starts)
    int n_cu;                    // compute units of the device (sizes the persistent grid)
};

// one level of the environment's mip chain from the one above (glGenerateMipmap on the oracle's GL: 2x2 bilinear centre)
hipError_t rfx_launch_env_mip(const float4 *src, float4 *dst, int sw, int sh, int dw, int dh, bool to_half, bool rtz, hipStream_t);

struct K2Args {
    FrameDims dims;
    int y0, y1;
    TexView ssgi, velocity, hist0, hist1;  // hist* = K3 target B of the previous frame (RGBA16F, linear), or the pass's framebuffer copy
    int hist_f32;                          // history texels are RGBA32F (FloatType framebuffer copy) instead of RGBA16F
    int in_w, in_h;                        // size of the input texture (smaller than the frame when K1 ran with resolutionScale < 1)
    TexViewW out0, out1;
    rfx_temporal_params p;
    float invW, invH;        // invTexSize (TemporalReprojectPass.js:135)
    float rcpInvW, rcpInvH;  // RN(1 / invTexSize): the constant of the exact quotient P / invTexSize (RFX_DIV_CONST's form, k2_bicubic)
    float prevPV[16];  // prevProjectionMatrix * prevViewMatrix, multiplied in fp32 like the shader does per fragment
};

struct K3Args {
    FrameDims dims;
    int y0, y1;
    TexView depth, gbuffer, in0, in1;
    const void *blue;
    int shift_x, shift_y;
    TexViewW out0, out1;
    rfx_denoise_params p;
    struct { int Rx, Ry, LW, LH, skip; } tile;  // filled by the launcher (skip: texels shaved off each end of the staged rectangle, k3_tiled_body)
    float tap_ox[8], tap_oy[8];           // POISSON[k] / resolution, filled by the launcher
};

struct K4Args {
    FrameDims dims;
    int y0, y1;
    TexView depth, gbuffer, gi0, gi1;  // gi*: K3 target B (RGBA16F, linear) or, giSource 1, K2's targets (RGBA32F, nearest)
    TexView scene;  // the composer's input buffer (sceneTexture): read only by inputType "specular"
    TexViewW out;
    float *rgb_out;  // RFX_TEX_COMPOSE_RGB (whole frame, 3 floats per texel) or null
    rfx_compose_params p;
};

// K0 importer (k0_import.hip): device staging planes of `rows` rows -> packed texels
hipError_t rfx_launch_pack_gbuffer(int W, int rows, const float *diffuse, const float *normal, const float *roughness, const float *metalness,
                                   const float *emissive, const float *depth, void *out, hipStream_t);
hipError_t rfx_launch_pack_velocity(int W, int rows, const float *velocity, const float *normal, const float *depth, void *out, hipStream_t);
// CubeToEquirectEnvPass (k0_import.hip): six S x S RGBA32F faces -> a W x H RGBA32F equirectangular image
hipError_t rfx_launch_cube_to_equirect(float4 *chain, int size, int levels, float4 *out, int W, int H, const UvPlanes &uv, hipStream_t);

struct K5Args {
    FrameDims dims;
    int y0, y1;
    TexView depth, gi, scene;  // gi = K4 output (inputTexture), scene = the composer's input buffer (sceneTexture)
    TexViewW out;
    rfx_final_params p;
};

hipError_t rfx_launch_k5(const K5Args &, hipStream_t);
int rfx_k1_base_cell();  // edge of k1_prepare's base cells in texels
hipError_t rfx_launch_k1_prepare(const K1Args &, hipStream_t);
hipError_t rfx_launch_k1(const K1Args &, int stage /* 0 fused, 1 trace, 2 shade */, hipStream_t);
// rows[0] = min, rows[1] = max history row the shade stage of the traced rays of rows [y0, y1) will read (device ints, preset INT_MAX / -1)
hipError_t rfx_launch_k1_hit_rows(const FrameDims &, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, bool allow_missed, int *rows, hipStream_t);
// mask[row] |= 1 << column block (32 blocks across the frame) for every history texel the shade stage of the traced rays of rows [y0, y1) will read (H words, zeroed)
hipError_t rfx_launch_k1_hit_mask(const FrameDims &, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, bool allow_missed, unsigned int *mask, hipStream_t);
hipError_t rfx_launch_k2(const K2Args &, hipStream_t);
hipError_t rfx_launch_k3(const K3Args &, hipStream_t);
hipError_t rfx_launch_k4(const K4Args &, hipStream_t);
// rows [y0, y1) of an RGBA32F plane -> the same rows of an RGBA16F (to_half) or RGBA32F plane
hipError_t rfx_launch_copy_fb(const FrameDims &, int y0, int y1, TexView src, TexViewW dst, bool to_half, hipStream_t);
