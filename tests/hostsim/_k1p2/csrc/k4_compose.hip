// K4 — DenoiserComposePass: gi = diffuse*(1-metal)*(1-F)*diffuseGi + specularGi*F + emissive.
// Replaces `renderer.render` of src/denoise/pass/DenoiserComposePass.js:133-134; arithmetic from
// the inline shader :36-86 and src/denoise/shader/denoiser_compose_functions.glsl:53-108.
// Pure streaming kernel: 52 B/px (4 depth + 16 gbuffer + 2x8 GI in, 16 out).
#include "k4_compose_texel.h"

namespace {

template <bool WHOLE>  // WHOLE: the GI views are the whole frame (no row rebasing / halo accounting in the bilinear fetches)
RFX_DEV void k4_compose_body(const K4Args &A, const FrameDims &d) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = A.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const float u = rfx_frag_u(d.uv, x, y), v = rfx_frag_v(d.uv, y);
    const float *depthp = (const float *)A.depth.ptr;
    const float depth = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, y)];
    {
        const int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
        float dxa = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx0, y)], dxb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, qx1, y)];
        float dya = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy0)], dyb = depthp[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, qy1)];
        if (depth == 1.0f && (fabsf(dxb - dxa) + fabsf(dyb - dya)) == 0.0f) {  // discard :61-64
            if (A.rgb_out) {  // the target keeps its texel: mirror it, so COMPOSE_RGB stays == COMPOSE.rgb on every tile texel
                const float4 keep = ((const float4 *)A.out.ptr)[(size_t)rfx_local_row(d, A.out.row0, A.out.rows, y) * d.W + x];
                float *r = A.rgb_out + ((size_t)y * d.W + x) * 3;
                r[0] = keep.x; r[1] = keep.y; r[2] = keep.z;
            }
            return;
        }
    }
    const Material mat = rfx_get_material<true>(((const uint4 *)A.gbuffer.ptr)[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, y)]);
    // DenoiserComposePass.js:26-33: "diffuseSpecular" -> (textures[0], textures[1]); "specular" -> specularGi = textures[0],
    // diffuseGiTexture unbound (zeros) and the diffuse component comes from sceneTexture
    float4 dgi = make_float4(0.f, 0.f, 0.f, 0.f), sgi;
    if (A.p.giSource) {  // denoiseMode "full_temporal": K2's own targets (RGBA32F, NearestFilter) — the texel itself
        const size_t gi = rfx_xy_index(d, A.gi0.row0, A.gi0.rows, x, y);
        if (A.p.inputType == 0) {
            dgi = ((const float4 *)A.gi0.ptr)[gi];
            sgi = ((const float4 *)A.gi1.ptr)[gi];
        } else {
            sgi = ((const float4 *)A.gi0.ptr)[gi];
        }
    } else if (A.p.inputType == 0) {
        dgi = rfx_fetch_h4_linear_fused<WHOLE>(A.gi0, d, u, v);  // the sampler's fused lerps on the half texels (rfx_device.h), as in K2 / K3
        sgi = rfx_fetch_h4_linear_fused<WHOLE>(A.gi1, d, u, v);
    } else {
        sgi = rfx_fetch_h4_linear_fused<WHOLE>(A.gi0, d, u, v);
    }
    float3 scene = make_float3(0.f, 0.f, 0.f);
    if (A.p.inputType == 2) {  // denoiser_compose_functions.glsl:97-101: diffuseComponent = textureLod(sceneTexture, vUv, 0.).rgb
        const float4 sc = ((const float4 *)A.scene.ptr)[rfx_xy_index(d, A.scene.row0, A.scene.rows, x, y)];
        scene = make_float3(sc.x, sc.y, sc.z);
    }
    const float4 o = k4_compose_texel(A.p, u, v, depth, mat, make_float3(dgi.x, dgi.y, dgi.z), make_float3(sgi.x, sgi.y, sgi.z), scene);
    ((float4 *)A.out.ptr)[(size_t)rfx_local_row(d, A.out.row0, A.out.rows, y) * d.W + x] = o;
    if (A.rgb_out) {
        float *r = A.rgb_out + ((size_t)y * d.W + x) * 3;
        r[0] = o.x; r[1] = o.y; r[2] = o.z;
    }
}

template <bool WHOLE>
__global__ __launch_bounds__(256) void k4_compose(K4Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k4_compose_body<WHOLE>(A, d);
    rfx_flush_violations(d);
}

// SSGIEffect's own fragment, src/ssgi/shader/ssgi_compose.frag:20-45 (the `mainImage` postprocessing's EffectPass runs
// after SSGIEffect.update).  Streaming: 4 B depth + 16 B (GI or scene) in, 16 B out.  The fog arithmetic is three.js'
// fog_fragment chunk (un-vendored dependency): FogExp2 1 - exp(-density^2 * d^2), Fog smoothstep(near, far, d).
__global__ __launch_bounds__(256) void k5_final_compose(K5Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    const int x = blockIdx.x * 64 + threadIdx.x, y = A.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x < d.W && y < A.y1) {
        const rfx_final_params &p = A.p;
        float4 o;
        // inputTexture at a texel centre: K4's / K2's RGBA32F texel, or (denoiseMode "denoised") K3's RGBA16F target B texel
        const size_t gii = rfx_xy_index(d, A.gi.row0, A.gi.rows, x, y);
        const float4 gi = p.inputSource == 2 ? rfx_load_half4(((const uint2 *)A.gi.ptr)[gii]) : ((const float4 *)A.gi.ptr)[gii];
        if (p.isDebug) {
            o = gi;  // :21-24
        } else {
            const float depth = ((const float *)A.depth.ptr)[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, y)];
            float3 c;
            if (depth == 1.0f) {
                const float4 sc = ((const float4 *)A.scene.ptr)[rfx_xy_index(d, A.scene.row0, A.scene.rows, x, y)];
                c = make_float3(sc.x, sc.y, sc.z);
            } else {
                c = make_float3(gi.x, gi.y, gi.z);
                if (p.fogMode) {
                    const float n_ = p.camera.near_, f_ = p.camera.far_;
                    const float viewZ = rfx_depth_to_view_z(depth, n_, f_, p.camera.isPerspective != 0) * 0.4f;  // getViewZ(depth) * 0.4 :36
                    const float fd = -viewZ;
                    float ff;
                    if (p.fogMode == 2) {
                        ff = 1.0f - rfx_exp(((-p.fogDensity * p.fogDensity) * fd) * fd);
                    } else {
                        const float t = rfx_clamp((fd - p.fogNear) / (p.fogFar - p.fogNear), 0.0f, 1.0f);
                        ff = t * t * (3.0f - 2.0f * t);
                    }
                    c = rfx_mix(c, make_float3(p.fogColor[0], p.fogColor[1], p.fogColor[2]), ff);
                }
            }
            o = make_float4(c.x, c.y, c.z, 1.0f);
        }
        ((float4 *)A.out.ptr)[(size_t)rfx_local_row(d, A.out.row0, A.out.rows, y) * d.W + x] = o;
    }
    rfx_flush_violations(d);
}

}  // namespace

hipError_t rfx_launch_k5(const K5Args &A, hipStream_t stream) {
    dim3 block(64, 4), grid((A.dims.W + 63) / 64, (A.y1 - A.y0 + 3) / 4);
    hipLaunchKernelGGL(k5_final_compose, grid, block, 0, stream, A);
    return hipGetLastError();
}

hipError_t rfx_launch_k4(const K4Args &A, hipStream_t stream) {
    dim3 block(64, 4), grid((A.dims.W + 63) / 64, (A.y1 - A.y0 + 3) / 4);
    const auto whole_view = [&](const void *ptr, int row0, int rows) { return ptr == nullptr || (row0 == 0 && rows == A.dims.H); };
    if (whole_view(A.gi0.ptr, A.gi0.row0, A.gi0.rows) && whole_view(A.gi1.ptr, A.gi1.row0, A.gi1.rows)) hipLaunchKernelGGL(k4_compose<true>, grid, block, 0, stream, A);
    else hipLaunchKernelGGL(k4_compose<false>, grid, block, 0, stream, A);
    return hipGetLastError();
}
