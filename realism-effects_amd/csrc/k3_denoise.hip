// K3 — PoissonDenoisePass: 8-tap rotated-Poisson edge-aware blur, one draw per call.
// Replaces `renderer.render` of src/denoise/pass/PoissonDenoisePass.js:146-147 with the
// fragment program src/denoise/shader/poisson_denoise.frag (GBUFFER_TEXTURE variant).
//
// Two implementations of the same arithmetic:
//   * k3_tiled  (radius <= 3, the default): a 64x8-pixel workgroup tile with a 4-texel apron is
//     staged through LDS ONCE — G-buffer texels are decoded to (normal, roughness) + depth, and in
//     pass 0 the RGBA32F inputs are pre-transformed to (log(rgb+1), luma^(1/8)) — so the 8 taps x 2
//     textures of every pixel read LDS instead of re-decoding / re-`log`-ing global texels
//     (13 G-buffer decodes and 48 logs per pixel become 2.25 and 13.5).  Later passes stage the raw
//     RGBA16F texels and do the sampler's bilinear blend from LDS.
//   * k3_generic: one pixel per lane with direct global gathers, for radius > 3.
// 64 consecutive pixels of a row per wavefront -> coalesced 16 B / 8 B per lane row reads and writes.
#include <type_traits>
#include "rfx_device.h"
#include "rfx_kernels.h"
#include "k3_rotation_table.h"

namespace {

constexpr int TW = 64, TH = 8;  // pixels per workgroup tile
constexpr int NT = TW * TH;     // 512 threads
#ifndef RFX_K3_XCD_G
#define RFX_K3_XCD_G 1  // tile rows per XCD group (rfx_xcd_tile); 0 = plain row-major.  Measured at 4K (pass 0 / pass 1 ms): 0: 0.248/0.380, 1: 0.244/0.376, 2: 0.245/0.390, 4: 0.255/0.414, 8: 0.266/0.423, 16: 0.288/0.434
#endif
// The tile is staged with an apron of (Rx, Ry) texels.  The reference rotates the Poisson offsets in UV space
// (`rm * (offset / resolution)`, poisson_denoise.frag:183-189), so on a W x H frame the tap footprint is
// radius * max(1, W/H) pixels wide and radius * max(1, H/W) pixels high — NOT a circle of `radius` pixels;
// +1 texel for the bilinear footprint of the RGBA16F passes.

struct CenterTexel {
    float3 rgb;     // log-space colour accumulator
    float a;        // age passes through
    float lumaPow;  // pow(lum, 1/8)
    float w;        // age weight
    float total;
};

RFX_DEV float k3_luma(float3 a) { return rfx_pow(rfx_lum(a), 0.125f); }  // poisson_denoise.frag:28
RFX_DEV float3 k3_log3(float x, float y, float z) { return make_float3(rfx_log(x + 1.0f), rfx_log(y + 1.0f), rfx_log(z + 1.0f)); }

// applyWeight poisson_denoise.frag:102-124 on an already log-transformed tap.  The bilateral weight arrives as its base-2
// LOGARITHM `l2w`: the reference forms w = exp(-a) (getBasicNeighborWeight :52-78) [* exp(-g) for a specular texture], then
// needs w * exp(-lumaDiff * lumaPhi) and pow(w, 0.1) — i.e. exp2(l2w + l2luma) and exp2(0.1 * l2w): two v_exp_f32 instead of
// exp, log, exp, exp.  l2w = -inf (background tap, :60) gives 0 for both, as w = 0 does in the reference.
constexpr float K3_LOG2E = 1.4426950408889634f;
RFX_DEV void k3_apply(CenterTexel &c, float l2w, float3 tl, float tapLuma, float lumaPhiL2) {
    const float disocclW = rfx_exp2(0.1f * l2w);
    const float lumaDiff = fminf(fabsf(c.lumaPow - tapLuma), 0.5f);
    const float wl = rfx_exp2(l2w - lumaDiff * lumaPhiL2);  // w * lumaFactor
    float w = rfx_mix(wl, disocclW, c.w) * c.w;
    w = (w < 0.0001f) ? 0.0f : w;  // w *= step(0.0001, w)
    c.rgb = c.rgb + tl * w;
    c.total += w;
}

// dynamic LDS carve-up (all 16-byte aligned: LW*LH is padded to a multiple of 4 texels):
//   float4 geom[n]          normal.xyz, roughness
//   pass 0 : float4 in[2][n] log(rgb+1), luma^(1/8)     pass >= 1 : uint2 in[2][n] raw RGBA16F
//   float  depth[n]

template <bool IN_TEMPORAL, int TC>
RFX_DEV void k3_tiled_body(const K3Args &A, const FrameDims &d) {
    extern __shared__ float4 lds[];
    const int Rx = A.tile.Rx, Ry = A.tile.Ry, LW = A.tile.LW, LH = A.tile.LH;
    const int ntex = (LW * LH + 3) & ~3;
    float4 *s_geom = lds;
    float4 *s_in0 = lds + ntex;                                   // pass 0 view
    uint2 *s_inN = reinterpret_cast<uint2 *>(lds + ntex);         // pass >= 1 view
    float *s_depth = reinterpret_cast<float *>(lds + ntex) + (IN_TEMPORAL ? 8 : 4) * (size_t)ntex;
    const rfx_denoise_params &p = A.p;
    const TileXY tile = rfx_xcd_tile<RFX_K3_XCD_G>((d.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH);
    if (!tile.valid) return;  // grid padding (uniform per workgroup, before any barrier)
    const int tx0 = tile.bx * TW, ty0 = A.y0 + tile.by * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const float *depthp = (const float *)A.depth.ptr;
    const uint4 *gbp = (const uint4 *)A.gbuffer.ptr;

    // ---- stage the tile + apron (texels outside the frame are never addressed: taps clamp to the edge first)
    const float invLW = 1.0f / (float)LW;
    for (int i = tid; i < LW * LH; i += NT) {
        // i / LW without the integer-division sequence: (i + 0.5) / LW is at least 0.5 / LW away from an integer, far above the
        // rounding error of the product for these sizes (i < 2^16, LW < 2^8)
        const int ly = (int)(((float)i + 0.5f) * invLW), lx = i - __mul24(ly, LW);
        const int gx = tx0 - Rx + lx, gy = ty0 - Ry + ly;
        // skip texels no tap of a PRODUCED pixel can address: outside the frame (taps clamp to the edge first) or
        // beyond the apron of the last produced row (the workgroup may overhang the launch's row range)
        if (gx < 0 || gx >= d.W || gy < 0 || gy >= d.H || gy > A.y1 - 1 + Ry) continue;
        const uint4 g = rfx_gather<uint4>(gbp, (unsigned int)(__mul24(rfx_local_row(d, A.gbuffer.row0, A.gbuffer.rows, gy), d.W) + gx));
        const float3 n = rfx_unpack_normal(g.y);
        s_geom[i] = make_float4(n.x, n.y, n.z, rfx_decode_roughness(g.z));
        s_depth[i] = rfx_gather<float>(depthp, (unsigned int)(__mul24(rfx_local_row(d, A.depth.row0, A.depth.rows, gy), d.W) + gx));
#pragma unroll
        for (int t = 0; t < TC; t++) {
            const TexView &src = t ? A.in1 : A.in0;
            const unsigned int idx = (unsigned int)(__mul24(rfx_local_row(d, src.row0, src.rows, gy), d.W) + gx);
            if constexpr (IN_TEMPORAL) {
                const float4 v = rfx_gather<float4>(src.ptr, idx);
                const float3 l = k3_log3(v.x, v.y, v.z);
                s_in0[t * ntex + i] = make_float4(l.x, l.y, l.z, k3_luma(l));
            } else {
                s_inN[t * ntex + i] = rfx_gather<uint2>(src.ptr, idx);
            }
        }
    }
    __syncthreads();

    const int x = tx0 + threadIdx.x, y = ty0 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const int cx = threadIdx.x + Rx, cy = threadIdx.y + Ry;  // this pixel inside the staged tile
    const int ci = __mul24(cy, LW) + cx;
    const float u = rfx_frag_u(d.uv, x, y), v = rfx_frag_v(d.uv, y);
    const float depth = s_depth[ci];

    // fine 2x2 quad derivatives; a partner beyond the frame edge fetches the edge texel
    const int qx0 = __mul24(cy, LW) + min(x & ~1, d.W - 1) - tx0 + Rx, qx1 = __mul24(cy, LW) + min(x | 1, d.W - 1) - tx0 + Rx;
    const int qy0 = __mul24(min(y & ~1, d.H - 1) - ty0 + Ry, LW) + cx, qy1 = __mul24(min(y | 1, d.H - 1) - ty0 + Ry, LW) + cx;
    {
        const float fw = fabsf(s_depth[qx1] - s_depth[qx0]) + fabsf(s_depth[qy1] - s_depth[qy0]);
        if (depth == 1.0f && fw == 0.0f) return;  // discard (:129-132): target keeps its contents
    }
    const float4 gc = s_geom[ci];
    const float3 normal = make_float3(gc.x, gc.y, gc.z);
    const float roughness = gc.w;
    const float glossiness = fmaxf(0.0f, 4.0f * (1.0f - roughness * 4.0f));  // roughness / 0.25, exactly
    const float l2spec = -glossiness * p.specularPhi * K3_LOG2E;  // log2(specularFactor) :169
    const float lumaPhiL2 = p.lumaPhi * K3_LOG2E;
    float flatness;
    {
        const float4 nxa = s_geom[qx0], nxb = s_geom[qx1], nya = s_geom[qy0], nyb = s_geom[qy1];
        const float3 fw = make_float3(fabsf(nxb.x - nxa.x) + fabsf(nyb.x - nya.x), fabsf(nxb.y - nxa.y) + fabsf(nyb.y - nya.y),
                                      fabsf(nxb.z - nxa.z) + fabsf(nyb.z - nya.z));
        flatness = 1.0f - fminf(rfx_length(fw), 1.0f);
        flatness = (flatness * flatness) * 0.75f + 0.25f;  // :172-173
    }

    // bilinear blend of four staged RGBA16F texels, exactly as rfx_fetch_h4_linear does from global
    auto lds_linear = [&](int t, float fu, float fv) -> float4 {
        int x0, x1, y0, y1;
        float wx, wy;
        rfx_linear_coord(fu, d.fW, d.W, x0, x1, wx);
        rfx_linear_coord(fv, d.fH, d.H, y0, y1, wy);
        x0 = min(max(x0 - tx0 + Rx, 0), LW - 1); x1 = min(max(x1 - tx0 + Rx, 0), LW - 1);
        y0 = __mul24(min(max(y0 - ty0 + Ry, 0), LH - 1), LW); y1 = __mul24(min(max(y1 - ty0 + Ry, 0), LH - 1), LW);
        const uint2 *sn = s_inN + t * ntex;
        const float4 t00 = rfx_load_half4(sn[y0 + x0]), t10 = rfx_load_half4(sn[y0 + x1]);
        const float4 t01 = rfx_load_half4(sn[y1 + x0]), t11 = rfx_load_half4(sn[y1 + x1]);
        float4 r;
        r.x = rfx_lerp(wy, rfx_lerp(wx, t00.x, t10.x), rfx_lerp(wx, t01.x, t11.x));
        r.y = rfx_lerp(wy, rfx_lerp(wx, t00.y, t10.y), rfx_lerp(wx, t01.y, t11.y));
        r.z = rfx_lerp(wy, rfx_lerp(wx, t00.z, t10.z), rfx_lerp(wx, t01.z, t11.z));
        r.w = rfx_lerp(wy, rfx_lerp(wx, t00.w, t10.w), rfx_lerp(wx, t01.w, t11.w));
        return r;
    };

    CenterTexel c[TC];
    bool isSpec[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {  // :137-165
        isSpec[i] = p.isTextureSpecular[i] != 0;
        const int ti = (TC == 2 && isSpec[i]) ? 1 : 0;
        float4 t;
        if constexpr (IN_TEMPORAL) {
            const TexView &src = ti ? A.in1 : A.in0;
            t = rfx_gather<float4>(src.ptr, (unsigned int)(__mul24(rfx_local_row(d, src.row0, src.rows, y), d.W) + x));
        } else {
            t = lds_linear(ti, u, v);
        }
        c[i].w = rfx_rcp(rfx_pow(t.w + 1.0f, 1.2f * p.phi));
        const float3 col = k3_log3(t.x * 1.0003f, t.y * 1.0003f, t.z * 1.0003f);
        c[i].rgb = col;
        c[i].a = t.w;
        c[i].lumaPow = k3_luma(col);
        c[i].total = 1.0f;
    }

    // angle = blueNoise.r * 2 pi takes 256 values: (sin, cos) from the correctly rounded table (k3_rotation_table.h).  At 120 / 240 degrees
    // (bytes 85, 170) a radius-3 tap of a flat surface sits exactly on a texel boundary and the last bit of cos decides its texel
    const float2 rot = K3_ROTATION[rfx_blue_noise_texel((const uchar4 *)A.blue, x, y, A.shift_x, A.shift_y).x];
    const float sn = rot.x, co = rot.y;
    const float rf = p.radius * flatness;
    const float m00 = rf * co, m01 = rf * -sn, m10 = rf * sn, m11 = rf * co;  // mat2 rm = r*flatness*mat2(c,-s,s,c) :183

    // no unrolling: the bilinear variant carries 8 half4 texels per tap; occupancy beats ILP here (116 -> 78 VGPRs, -6 %)
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        const float ox = A.tap_ox[k], oy = A.tap_oy[k];  // POISSON[k] / resolution (:91-92,:189), divided once on the host
        const float nu = u + (m00 * ox + m10 * oy), nv = v + (m01 * ox + m11 * oy);
        const int nx = min(max(rfx_nearest_idx(nu, d.fW, d.W) - tx0 + Rx, 0), LW - 1);
        const int ny = min(max(rfx_nearest_idx(nv, d.fH, d.H) - ty0 + Ry, 0), LH - 1);
        const int ni = __mul24(ny, LW) + nx;
        // getBasicNeighborWeight :52-78
        const float nd = s_depth[ni];
        const float4 ng = s_geom[ni];
        const float normalDiff = 1.0f - fmaxf(rfx_dot(normal, make_float3(ng.x, ng.y, ng.z)), 0.0f);
        const float depthDiff = 10000.0f * fabsf(depth - nd);
        const float roughDiff = fabsf(roughness - ng.w);
        float l2basic = (-normalDiff * p.normalPhi - depthDiff * p.depthPhi - roughDiff * p.roughnessPhi) * K3_LOG2E;
        l2basic = (nd != 1.0f) ? l2basic : -__builtin_inff();
#pragma unroll
        for (int i = 0; i < TC; i++) {
            const int ti = (TC == 2 && isSpec[i]) ? 1 : 0;
            const float l2w = isSpec[i] ? l2basic + l2spec : l2basic;
            if constexpr (IN_TEMPORAL) {
                const float4 tl = s_in0[ti * ntex + ni];
                k3_apply(c[i], l2w, make_float3(tl.x, tl.y, tl.z), tl.w, lumaPhiL2);
            } else {
                const float4 t = lds_linear(ti, nu, nv);
                const float3 tl = k3_log3(t.x, t.y, t.z);
                k3_apply(c[i], l2w, tl, k3_luma(tl), lumaPhiL2);
            }
        }
    }

    const size_t oi = (size_t)(unsigned int)(__mul24(rfx_local_row(d, A.out0.row0, A.out0.rows, y), d.W) + x);
#pragma unroll
    for (int i = 0; i < TC; i++) {  // outputTexel :94-100
        const float inv = rfx_rcp(c[i].total);
        float3 o = make_float3(c[i].rgb.x * inv, c[i].rgb.y * inv, c[i].rgb.z * inv);
        o = make_float3(rfx_exp(o.x) - 1.0f, rfx_exp(o.y) - 1.0f, rfx_exp(o.z) - 1.0f);
        ((uint2 *)(i ? A.out1.ptr : A.out0.ptr))[oi] = rfx_store_half4(o.x, o.y, o.z, c[i].a, p.halfStoreRTZ != 0);
    }
}

// ---------------------------------------------------------------- generic variant (any radius)
template <bool IN_TEMPORAL>
RFX_DEV float4 k3_input(const TexView &t, const FrameDims &d, float u, float v) {
    if (IN_TEMPORAL) return rfx_fetch_f4(t, d, u, v);  // pass 0: K2 output, RGBA32F nearest
    return rfx_fetch_h4_linear(t, d, u, v);            // pass >= 1: ping-pong target, RGBA16F linear
}

template <bool IN_TEMPORAL, int TC>
RFX_DEV void k3_generic_body(const K3Args &A, const FrameDims &d) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = A.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const rfx_denoise_params &p = A.p;
    const float u = rfx_frag_u(d.uv, x, y), v = rfx_frag_v(d.uv, y);
    const float *depthp = (const float *)A.depth.ptr;
    const uint4 *gbp = (const uint4 *)A.gbuffer.ptr;
    const float depth = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, y)];
    const int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
    {
        float dxa = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx0, y)], dxb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx1, y)];
        float dya = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy0)], dyb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy1)];
        if (depth == 1.0f && (fabsf(dxb - dxa) + fabsf(dyb - dya)) == 0.0f) return;
    }
    CenterTexel c[TC];
    bool isSpec[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {
        isSpec[i] = p.isTextureSpecular[i] != 0;
        float4 t = k3_input<IN_TEMPORAL>(isSpec[i] ? A.in1 : A.in0, d, u, v);
        c[i].w = 1.0f / rfx_pow(t.w + 1.0f, 1.2f * p.phi);
        const float3 col = k3_log3(t.x * 1.0003f, t.y * 1.0003f, t.z * 1.0003f);
        c[i].rgb = col;
        c[i].a = t.w;
        c[i].lumaPow = k3_luma(col);
        c[i].total = 1.0f;
    }
    const uint4 g = gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, y)];
    const float3 normal = rfx_unpack_normal(g.y);
    const float roughness = rfx_decode_roughness(g.z);
    const float glossiness = fmaxf(0.0f, 4.0f * (1.0f - roughness / 0.25f));
    const float l2spec = -glossiness * p.specularPhi * K3_LOG2E;
    const float lumaPhiL2 = p.lumaPhi * K3_LOG2E;
    float flatness;
    {
        float3 nxa = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, qx0, y)].y);
        float3 nxb = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, qx1, y)].y);
        float3 nya = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, qy0)].y);
        float3 nyb = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, qy1)].y);
        float3 fw = make_float3(fabsf(nxb.x - nxa.x) + fabsf(nyb.x - nya.x), fabsf(nxb.y - nxa.y) + fabsf(nyb.y - nya.y),
                                fabsf(nxb.z - nxa.z) + fabsf(nyb.z - nya.z));
        flatness = 1.0f - fminf(rfx_length(fw), 1.0f);
        flatness = (flatness * flatness) * 0.75f + 0.25f;
    }
    // angle = blueNoise.r * 2 pi takes 256 values: (sin, cos) from the correctly rounded table (k3_rotation_table.h).  At 120 / 240 degrees
    // (bytes 85, 170) a radius-3 tap of a flat surface sits exactly on a texel boundary and the last bit of cos decides its texel
    const float2 rot = K3_ROTATION[rfx_blue_noise_texel((const uchar4 *)A.blue, x, y, A.shift_x, A.shift_y).x];
    const float sn = rot.x, co = rot.y;
    const float rf = p.radius * flatness;
    const float m00 = rf * co, m01 = rf * -sn, m10 = rf * sn, m11 = rf * co;
    for (int k = 0; k < 8; k++) {
        const float ox = A.tap_ox[k], oy = A.tap_oy[k];
        const float nu = u + (m00 * ox + m10 * oy), nv = v + (m01 * ox + m11 * oy);
        float l2basic = -__builtin_inff();
        {
            const uint4 ng = gbp[rfx_texel_index(d, A.gbuffer.row0, A.gbuffer.rows, nu, nv)];
            const float nd = depthp[rfx_texel_index(d, A.depth.row0, A.depth.rows, nu, nv)];
            if (nd != 1.0f) {
                float3 nn = rfx_unpack_normal(ng.y);
                float normalDiff = 1.0f - fmaxf(rfx_dot(normal, nn), 0.0f);
                float depthDiff = 10000.0f * fabsf(depth - nd);
                float roughDiff = fabsf(roughness - rfx_decode_roughness(ng.z));
                l2basic = (-normalDiff * p.normalPhi - depthDiff * p.depthPhi - roughDiff * p.roughnessPhi) * K3_LOG2E;
            }
        }
#pragma unroll
        for (int i = 0; i < TC; i++) {
            const float l2w = isSpec[i] ? l2basic + l2spec : l2basic;
            const float4 t = k3_input<IN_TEMPORAL>(isSpec[i] ? A.in1 : A.in0, d, nu, nv);
            const float3 tl = k3_log3(t.x, t.y, t.z);
            k3_apply(c[i], l2w, tl, k3_luma(tl), lumaPhiL2);
        }
    }
    const size_t oi = (size_t)rfx_local_row(d, A.out0.row0, A.out0.rows, y) * d.W + x;
#pragma unroll
    for (int i = 0; i < TC; i++) {
        float3 o = make_float3(c[i].rgb.x / c[i].total, c[i].rgb.y / c[i].total, c[i].rgb.z / c[i].total);
        o = make_float3(rfx_exp(o.x) - 1.0f, rfx_exp(o.y) - 1.0f, rfx_exp(o.z) - 1.0f);
        ((uint2 *)(i ? A.out1.ptr : A.out0.ptr))[oi] = rfx_store_half4(o.x, o.y, o.z, c[i].a, p.halfStoreRTZ != 0);
    }
}

template <bool IN_TEMPORAL, int TC>
__global__ __launch_bounds__(NT) void k3_tiled(K3Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k3_tiled_body<IN_TEMPORAL, TC>(A, d);
    rfx_flush_violations(d);
}
template <bool IN_TEMPORAL, int TC>
__global__ __launch_bounds__(256) void k3_generic(K3Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k3_generic_body<IN_TEMPORAL, TC>(A, d);
    rfx_flush_violations(d);
}

}  // namespace

hipError_t rfx_launch_k3(const K3Args &A_in, hipStream_t stream) {
    K3Args A = A_in;
    const bool temporal = A.p.inputIsTemporal != 0;
    // apron of the tap footprint: anisotropic because the reference rotates in UV space
    const float aspect = A.dims.fW / A.dims.fH;
    const float rx = A.p.radius * fmaxf(1.0f, aspect), ry = A.p.radius * fmaxf(1.0f, 1.0f / aspect);
    {
        const float SQ = 0.25f * 1.41421356237f;
        const float pox[8] = {-1.f, 0.f, 1.f, 0.f, -SQ, SQ, SQ, -SQ};
        const float poy[8] = {0.f, -1.f, 0.f, 1.f, -SQ, -SQ, SQ, SQ};
        for (int k = 0; k < 8; k++) {  // IEEE fp32 divisions, identical to the per-fragment `offset / resolution`
            volatile float ox = pox[k] / A.dims.fW, oy = poy[k] / A.dims.fH;
            A.tap_ox[k] = ox;
            A.tap_oy[k] = oy;
        }
    }
    A.tile.Rx = (int)ceilf(rx) + 1;
    A.tile.Ry = (int)ceilf(ry) + 1;
    A.tile.LW = TW + 2 * A.tile.Rx;
    A.tile.LH = TH + 2 * A.tile.Ry;
    const int ntex = (A.tile.LW * A.tile.LH + 3) & ~3;
    const size_t lds = (size_t)ntex * (16 + 4 + 2 * (temporal ? 16 : 8));
    // two workgroups per CU (160 KiB LDS) keep the staging of one tile under the arithmetic of another
    const bool tiled = A.p.radius >= 0.0f && lds <= 80 * 1024;
    if (tiled) {
        dim3 block(TW, TH), grid(rfx_xcd_grid(RFX_K3_XCD_G, (A.dims.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH));
        // the attribute is per device (a process may hold contexts on several): remembered per device ordinal
#define K3_TILED(T, C)                                                                                                       \
    do {                                                                                                                     \
        static bool attr_set[64] = {false};                                                                                  \
        int dev = 0;                                                                                                         \
        hipGetDevice(&dev);                                                                                                  \
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                        \
            hipFuncSetAttribute((const void *)k3_tiled<T, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);        \
            if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                  \
        }                                                                                                                    \
        hipLaunchKernelGGL((k3_tiled<T, C>), grid, block, lds, stream, A);                                                   \
    } while (0)
        if (A.p.textureCount == 2) {
            if (temporal) K3_TILED(true, 2);
            else K3_TILED(false, 2);
        } else {
            if (temporal) K3_TILED(true, 1);
            else K3_TILED(false, 1);
        }
#undef K3_TILED
    } else {
        dim3 block(64, 4), grid((A.dims.W + 63) / 64, (A.y1 - A.y0 + 3) / 4);
        if (A.p.textureCount == 2) {
            if (temporal) hipLaunchKernelGGL((k3_generic<true, 2>), grid, block, 0, stream, A);
            else hipLaunchKernelGGL((k3_generic<false, 2>), grid, block, 0, stream, A);
        } else {
            if (temporal) hipLaunchKernelGGL((k3_generic<true, 1>), grid, block, 0, stream, A);
            else hipLaunchKernelGGL((k3_generic<false, 1>), grid, block, 0, stream, A);
        }
    }
    return hipGetLastError();
}
