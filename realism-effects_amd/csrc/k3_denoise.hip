// K3 — PoissonDenoisePass: 8-tap rotated-Poisson edge-aware blur, one draw per call.
// Replaces `renderer.render` of src/denoise/pass/PoissonDenoisePass.js:146-147 with the
// fragment program src/denoise/shader/poisson_denoise.frag (GBUFFER_TEXTURE variant).
//
// Launch shape: 64x4-pixel workgroups (4 wavefronts, one image row segment per wavefront) so
// every centre fetch is a fully coalesced 16 B/lane (RGBA32F) or 8 B/lane (RGBA16F) row read;
// even-aligned tiles keep the 2x2 derivative quads inside one workgroup.
#include "rfx_device.h"
#include "rfx_kernels.h"

namespace {

struct CenterTexel {
    float3 rgb;     // log-space colour accumulator
    float a;        // age passes through
    float lumaPow;  // pow(lum, 1/8)
    float w;        // age weight
    float total;
};

template <bool IN_TEMPORAL>
RFX_DEV float4 k3_input(const TexView &t, const FrameDims &d, float u, float v) {
    if (IN_TEMPORAL) return rfx_fetch_f4(t, d, u, v);  // pass 0: K2 output, RGBA32F nearest
    return rfx_fetch_h4_linear(t, d, u, v);            // pass >= 1: ping-pong target, RGBA16F linear
}

RFX_DEV float k3_luma(float3 a) { return rfx_pow(rfx_lum(a), 0.125f); }  // poisson_denoise.frag:28

template <bool IN_TEMPORAL, int TC>
__global__ __launch_bounds__(256) void k3_poisson_denoise(K3Args A) {
    const FrameDims d = A.dims;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = A.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const rfx_denoise_params &p = A.p;

    const float u = ((float)x + 0.5f) / d.fW, v = ((float)y + 0.5f) / d.fH;
    const float *depthp = (const float *)A.depth.ptr;
    const uint4 *gbp = (const uint4 *)A.gbuffer.ptr;
    const float depth = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, y)];

    // fine 2x2 quad derivatives: partners are (x&~1 | x|1, y) and (x, y&~1 | y|1)
    const int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
    {
        float dxa = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx0, y)], dxb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx1, y)];
        float dya = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy0)], dyb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy1)];
        float fw = fabsf(dxb - dxa) + fabsf(dyb - dya);
        if (depth == 1.0f && fw == 0.0f) return;  // discard (:129-132): target keeps its contents
    }

    CenterTexel c[TC];
    bool isSpec[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {  // :137-165
        isSpec[i] = p.isTextureSpecular[i] != 0;
        float4 t = k3_input<IN_TEMPORAL>(isSpec[i] ? A.in1 : A.in0, d, u, v);
        c[i].w = 1.0f / rfx_pow(t.w + 1.0f, 1.2f * p.phi);
        float3 col = make_float3(rfx_log(t.x * 1.0003f + 1.0f), rfx_log(t.y * 1.0003f + 1.0f), rfx_log(t.z * 1.0003f + 1.0f));
        c[i].rgb = col;
        c[i].a = t.w;
        c[i].lumaPow = k3_luma(col);
        c[i].total = 1.0f;
    }

    const uint4 g = gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, y)];
    const float3 normal = rfx_unpack_normal(g.y);
    const float roughness = rfx_decode_roughness(g.z);
    const float glossiness = fmaxf(0.0f, 4.0f * (1.0f - roughness / 0.25f));
    const float specularFactor = rfx_exp(-glossiness * p.specularPhi);

    float flatness;
    {
        float3 nxa = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, qx0, y)].y);
        float3 nxb = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, qx1, y)].y);
        float3 nya = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, qy0)].y);
        float3 nyb = rfx_unpack_normal(gbp[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, qy1)].y);
        float3 fw = make_float3(fabsf(nxb.x - nxa.x) + fabsf(nyb.x - nya.x), fabsf(nxb.y - nxa.y) + fabsf(nyb.y - nya.y),
                                fabsf(nxb.z - nxa.z) + fabsf(nyb.z - nya.z));
        flatness = 1.0f - fminf(rfx_length(fw), 1.0f);
        flatness = (flatness * flatness) * 0.75f + 0.25f;  // :172-173
    }

    const float4 rnd = rfx_blue_noise((const uchar4 *)A.blue, x, y, A.shift_x, A.shift_y);
    const float angle = rnd.x * 2.0f * 3.141592653589793f;
    float s, co;
    __sincosf(angle, &s, &co);
    const float rf = p.radius * flatness;
    // mat2 rm = r * flatness * mat2(c, -s, s, c)  (columns (c,-s), (s,c))  :183
    const float m00 = rf * co, m01 = rf * -s, m10 = rf * s, m11 = rf * co;

    const float SQ = 0.25f * 1.41421356237f;
    const float px[8] = {-1.f, 0.f, 1.f, 0.f, -SQ, SQ, SQ, -SQ};
    const float py[8] = {0.f, -1.f, 0.f, 1.f, -SQ, -SQ, SQ, SQ};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float ox = px[k] / d.fW, oy = py[k] / d.fH;
        const float nu = u + (m00 * ox + m10 * oy), nv = v + (m01 * ox + m11 * oy);
        // getBasicNeighborWeight :52-78
        float wBasic = 0.0f;
        {
            const size_t ni = rfx_texel_index(d, A.gbuffer.row0, A.gbuffer.rows, nu, nv);
            const size_t di = rfx_texel_index(d, A.depth.row0, A.depth.rows, nu, nv);
            const uint4 ng = gbp[ni];
            const float nd = depthp[di];
            if (nd != 1.0f) {
                float3 nn = rfx_unpack_normal(ng.y);
                float nr = rfx_decode_roughness(ng.z);
                float normalDiff = 1.0f - fmaxf(rfx_dot(normal, nn), 0.0f);
                float depthDiff = 10000.0f * fabsf(depth - nd);
                float roughDiff = fabsf(roughness - nr);
                wBasic = rfx_exp(-normalDiff * p.normalPhi - depthDiff * p.depthPhi - roughDiff * p.roughnessPhi);
            }
        }
#pragma unroll
        for (int i = 0; i < TC; i++) {  // applyWeight :102-124
            float w = wBasic;
            float4 t = k3_input<IN_TEMPORAL>(isSpec[i] ? A.in1 : A.in0, d, nu, nv);
            if (isSpec[i]) w *= specularFactor;
            float3 tl = make_float3(rfx_log(t.x + 1.0f), rfx_log(t.y + 1.0f), rfx_log(t.z + 1.0f));
            float disocclW = rfx_pow(w, 0.1f);
            float lumaDiff = fminf(fabsf(c[i].lumaPow - k3_luma(tl)), 0.5f);
            float lumaFactor = rfx_exp(-lumaDiff * p.lumaPhi);
            w = rfx_mix(w * lumaFactor, disocclW, c[i].w) * c[i].w;
            w = (w < 0.0001f) ? 0.0f : w;  // w *= step(0.0001, w)
            c[i].rgb = c[i].rgb + tl * w;
            c[i].total += w;
        }
    }

    const size_t oi = (size_t)rfx_local_row(d, A.out0.row0, A.out0.rows, y) * d.W + x;
#pragma unroll
    for (int i = 0; i < TC; i++) {  // outputTexel :94-100
        float3 o = make_float3(c[i].rgb.x / c[i].total, c[i].rgb.y / c[i].total, c[i].rgb.z / c[i].total);
        o = make_float3(rfx_exp(o.x) - 1.0f, rfx_exp(o.y) - 1.0f, rfx_exp(o.z) - 1.0f);
        uint2 h = rfx_store_half4(o.x, o.y, o.z, c[i].a, p.halfStoreRTZ != 0);
        ((uint2 *)(i ? A.out1.ptr : A.out0.ptr))[oi] = h;
    }
}

}  // namespace

hipError_t rfx_launch_k3(const K3Args &A, hipStream_t stream) {
    dim3 block(64, 4), grid((A.dims.W + 63) / 64, (A.y1 - A.y0 + 3) / 4);
    const bool temporal = A.p.inputIsTemporal != 0;
    if (A.p.textureCount == 2) {
        if (temporal) hipLaunchKernelGGL((k3_poisson_denoise<true, 2>), grid, block, 0, stream, A);
        else hipLaunchKernelGGL((k3_poisson_denoise<false, 2>), grid, block, 0, stream, A);
    } else {
        if (temporal) hipLaunchKernelGGL((k3_poisson_denoise<true, 1>), grid, block, 0, stream, A);
        else hipLaunchKernelGGL((k3_poisson_denoise<false, 1>), grid, block, 0, stream, A);
    }
    return hipGetLastError();
}
