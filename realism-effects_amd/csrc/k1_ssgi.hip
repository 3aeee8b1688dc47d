// K1 — SSGI ray-march.  Replaces `renderer.render` of src/ssgi/pass/SSGIPass.js:93-94 with
// the fragment program src/ssgi/shader/ssgi.frag (main :105-309, doSample :362-439,
// RayMarch :441-475, BinarySearch :477-503) in the MODE_SSGI / PERSPECTIVE_CAMERA /
// no-env-map variant (the configs carry no env map).
//
// One pixel per lane, 64x4-pixel workgroups: the G-buffer/direct-light/output planes are read
// and written as coalesced 16 B/lane rows; the depth taps of the march are data-dependent
// gathers into the (L2/MALL-resident) R32F depth plane.
#include "rfx_brdf.h"
#include "rfx_kernels.h"

namespace {

struct MarchCtx {
    const float *P;  // projectionMatrix
    const float *depth;
    int depth_row0, depth_rows;
    float nearMulFar, farMinusNear, cameraFar;
    float rayDistance, thickness;
    int steps, refineSteps;
};

RFX_DEV float k1_view_z(const MarchCtx &m, float depth) {  // getViewZ ssgi_utils.frag:7-13
    return m.nearMulFar / (m.farMinusNear * depth - m.cameraFar);
}
RFX_DEV float2 k1_project(const MarchCtx &m, float3 p) {  // viewSpaceToScreenSpace :26-33
    float4 pc = rfx_mat_mul(m.P, p.x, p.y, p.z, 1.0f);
    return make_float2((pc.x / pc.w) * 0.5f + 0.5f, (pc.y / pc.w) * 0.5f + 0.5f);
}
RFX_DEV float k1_depth_tap(const MarchCtx &m, const FrameDims &d, float2 uv) {
    return m.depth[rfx_texel_index(d, m.depth_row0, m.depth_rows, uv.x, uv.y)];
}

// RayMarch + BinarySearch.  Returns the hit uv; hitPos.x == 1e10 marks a miss.
RFX_DEV float2 k1_ray_march(const MarchCtx &m, const FrameDims &d, float3 dir, float3 &hitPos, float random_b) {
    dir = dir * (m.rayDistance / (float)m.steps);
    float2 uv = make_float2(0.f, 0.f);
    for (int i = 1; i < m.steps; i++) {
        const float t = (float)i + random_b - 0.5f;
        const float cs = 1.0f - rfx_exp(-0.25f * (t * t));
        hitPos = hitPos + dir * cs;
        uv = k1_project(m, hitPos);
        const float z = k1_view_z(m, k1_depth_tap(m, d, uv));
        const float diff = z - hitPos.z;
        if (diff >= 0.0f && diff < m.thickness) {
            if (m.refineSteps == 0) return uv;
            dir = dir * 0.5f;
            hitPos = hitPos - dir;
            for (int k = 0; k < m.refineSteps; k++) {
                uv = k1_project(m, hitPos);
                const float zz = k1_view_z(m, k1_depth_tap(m, d, uv));
                const float dd = zz - hitPos.z;
                dir = dir * 0.5f;
                hitPos = (dd >= 0.0f) ? hitPos - dir : hitPos + dir;
            }
            return k1_project(m, hitPos);
        }
    }
    hitPos = make_float3(10.0e9f, 10.0e9f, 10.0e9f);
    return uv;
}

RFX_DEV float k1_smoothstep(float e0, float e1, float x) {
    float t = rfx_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

struct Angles {
    float NoL, NoH, LoH, VoH;
};
RFX_DEV Angles k1_angles(float3 l, float3 v, float3 n) {  // calculateAngles :93-100
    const float E = 0.00001f, OME = 1.0f - 0.00001f;
    float3 h = rfx_normalize(v + l);
    Angles a;
    a.NoL = rfx_clamp(rfx_dot(n, l), E, OME);
    a.NoH = rfx_clamp(rfx_dot(n, h), E, OME);
    a.LoH = rfx_clamp(rfx_dot(l, h), E, OME);
    a.VoH = rfx_clamp(rfx_dot(v, h), E, OME);
    return a;
}

// doSample :362-439 without env map (getEnvColor == 0).  Returns gi * brdf / pdf.
RFX_DEV float3 k1_do_sample(const MarchCtx &m, const FrameDims &d, const K1Args &A, const Material &mat, float3 viewPos, float3 viewNormal,
                            float roughness, bool isDiffuseSample, float NoV, const Angles &an, float random_b, float3 l, float3 &hitPos) {
    const float cosTheta = fmaxf(0.0f, rfx_dot(viewNormal, l));
    float brdf, pdf;
    if (isDiffuseSample) {
        brdf = rfx_eval_disney_diffuse(an.NoL, NoV, an.LoH, roughness, mat.metalness);
        pdf = an.NoL / RFX_PI;
    } else {
        brdf = rfx_eval_disney_specular(roughness, an.NoH, NoV, an.NoL);
        pdf = rfx_ggx_vndf_pdf(an.NoH, NoV, roughness);
    }
    brdf *= cosTheta;
    pdf = fmaxf(0.00001f, pdf);
    hitPos = viewPos;
    const float2 coords = k1_ray_march(m, d, l, hitPos, random_b);
    const bool allowMissed = A.p.missedRays != 0;
    const bool isMissed = hitPos.x == 10.0e9f;
    float3 ssgi = make_float3(0.f, 0.f, 0.f);
    if (isMissed && !allowMissed) return ssgi;
    // velocityTexture is never wired in the reference (SSGIPass.js:89) -> velocity == 0
    const float ru = coords.x, rv = coords.y;
    if (ru >= 0.0f && ru <= 1.0f && rv >= 0.0f && rv <= 1.0f) {
        const float4 h = rfx_fetch_f4(A.history, d, ru, rv);
        float3 gi = make_float3(h.x, h.y, h.z);
        const float mx = fmaxf(fmaxf(mat.diffuse.x, mat.diffuse.y), mat.diffuse.z);
        const float mn = fminf(fminf(mat.diffuse.x, mat.diffuse.y), mat.diffuse.z);
        const float sat = (mx == mn) ? 0.0f : (mx - mn) / mx;  // getSaturation :348-360
        const float L = rfx_lum(gi);
        gi = rfx_mix(gi, make_float3(L, L, L), (1.0f - roughness) * sat * 0.4f);
        const float border = 0.15f;
        float bf = k1_smoothstep(0.0f, border, coords.x) * k1_smoothstep(1.0f, 1.0f - border, coords.x) * k1_smoothstep(0.0f, border, coords.y) *
                   k1_smoothstep(1.0f, 1.0f - border, coords.y);
        bf = rfx_sqrt(bf);
        ssgi = rfx_mix(make_float3(0.f, 0.f, 0.f), gi, bf);
        if (allowMissed && 0.0f > rfx_lum(ssgi)) ssgi = make_float3(0.f, 0.f, 0.f);  // :430-436 with envMapSample == 0
    }
    ssgi = ssgi * brdf;
    return make_float3(ssgi.x / pdf, ssgi.y / pdf, ssgi.z / pdf);
}

RFX_DEV void k1_ssgi_march_body(const K1Args &A, const FrameDims &d) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = A.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const rfx_ssgi_params &p = A.p;
    const float *C = p.camera.matrixWorld, *Vw = p.camera.matrixWorldInverse;
    const float *P = p.camera.projectionMatrix, *Pi = p.camera.projectionMatrixInverse;

    const float u = ((float)x + 0.5f) / d.fW, v = ((float)y + 0.5f) / d.fH;
    const float depth = ((const float *)A.depth.ptr)[rfx_xy_index(d, A.depth.row0, A.depth.rows, x, y)];
    uint4 *outp = (uint4 *)A.out.ptr + ((size_t)rfx_local_row(d, A.out.row0, A.out.rows, y) * d.W + x);
    const size_t gi_idx = rfx_xy_index(d, A.direct.row0, A.direct.rows, x, y);
    if (depth == 1.0f) {  // background :109-113
        const float4 dl = ((const float4 *)A.direct.ptr)[gi_idx];
        *outp = rfx_pack_two_vec4(dl, dl);
        return;
    }
    const Material mat = rfx_get_material<false>(((const uint4 *)A.gbuffer.ptr)[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, x, y)]);
    const float roughnessSq = rfx_clamp(mat.roughness * mat.roughness, 0.000001f, 1.0f);

    MarchCtx m;
    m.P = P;
    m.depth = (const float *)A.depth.ptr;
    m.depth_row0 = A.depth.row0;
    m.depth_rows = A.depth.rows;
    m.nearMulFar = A.nearMulFar;
    m.farMinusNear = A.farMinusNear;
    m.cameraFar = p.camera.far_;
    m.rayDistance = p.rayDistance;
    m.thickness = p.thickness;
    m.steps = p.steps;
    m.refineSteps = p.refineSteps;

    const float viewZ = k1_view_z(m, depth);
    // getViewPosition ssgi_utils.frag:17-24
    const float clipW = P[2 * 4 + 3] * viewZ + P[3 * 4 + 3];
    const float4 pp = rfx_mat_mul(Pi, ((u - 0.5f) * 2.0f) * clipW, ((v - 0.5f) * 2.0f) * clipW, ((viewZ - 0.5f) * 2.0f) * clipW, 1.0f * clipW);
    const float3 viewPos = make_float3(pp.x, pp.y, viewZ);
    const float3 viewDir = rfx_normalize(viewPos);
    const float3 N = mat.normal;
    const float3 viewNormal = rfx_normalize(rfx_vec_mul_mat(C, N, 0.0f));
    const float3 n = viewNormal, vv = -viewDir;
    const float NoV = fmaxf(0.00001f, rfx_dot(n, vv));
    float3 V = rfx_vec_mul_mat(Vw, vv, 0.0f);
    float3 T, B;
    rfx_onb(N, T, B);
    V = rfx_to_local(T, B, N, V);
    const float3 f0 = rfx_mix(make_float3(0.04f, 0.04f, 0.04f), mat.diffuse, mat.metalness);
    const float4 rnd = rfx_blue_noise((const uchar4 *)A.blue, x, y, A.shift_x, A.shift_y);

    float3 H = rfx_sample_ggx_vndf(V, roughnessSq, roughnessSq, rnd.x, rnd.y);
    if (H.z < 0.0f) H = -H;
    float3 l = rfx_normalize(rfx_reflect(-V, H));
    l = rfx_to_world(T, B, N, l);
    l = rfx_normalize(rfx_vec_mul_mat(C, l, 0.0f));
    Angles an = k1_angles(l, vv, n);

    // diffuse-vs-specular lobe selection :169-186
    bool isDiffuseSample;
    {
        const float3 F = rfx_f_schlick(f0, an.VoH);
        float diffW = (1.0f - mat.metalness) * rfx_lum(mat.diffuse);
        float specW = rfx_lum(F);
        diffW = fmaxf(diffW, 0.00001f);
        specW = fmaxf(specW, 0.00001f);
        const float invW = 1.0f / (diffW + specW);
        diffW *= invW;
        isDiffuseSample = rnd.z < diffW;
    }
    const float3 specularRay = l;
    float3 diffuseGI = make_float3(-1.0f, -1.0f, -1.0f);  // "not sampled this frame" marker :277-278
    float3 hitPos;
    float3 dl = make_float3(0.f, 0.f, 0.f);
    if (p.useDirectLight) {
        const float4 t = ((const float4 *)A.direct.ptr)[gi_idx];
        dl = make_float3(t.x, t.y, t.z);
    }
    if (isDiffuseSample) {  // :222-242
        const float3 diffuseRay = rfx_cosine_sample_hemisphere(viewNormal, rnd.x, rnd.y);
        const Angles ad = k1_angles(diffuseRay, vv, n);
        diffuseGI = k1_do_sample(m, d, A, mat, viewPos, viewNormal, roughnessSq, true, NoV, ad, rnd.z, diffuseRay, hitPos);
        diffuseGI = diffuseGI + dl;
    }
    // specular ray, traced every frame — evaluated with the SAME isDiffuseSample flag (:246-265)
    an = k1_angles(specularRay, vv, n);
    float3 specularGI = k1_do_sample(m, d, A, mat, viewPos, viewNormal, roughnessSq, isDiffuseSample, NoV, an, rnd.z, specularRay, hitPos);
    specularGI = specularGI + dl;

    float rayLength = 0.0f;  // :284-296
    if (!(hitPos.x > 10.0e8f)) {
        const float4 hw = rfx_mat_mul(C, hitPos.x, hitPos.y, hitPos.z, 1.0f);
        rayLength = rfx_length(make_float3(C[12], C[13], C[14]) - make_float3(hw.x, hw.y, hw.z));
    }
    *outp = rfx_pack_two_vec4(make_float4(diffuseGI.x, diffuseGI.y, diffuseGI.z, mat.roughness),
                              make_float4(specularGI.x, specularGI.y, specularGI.z, rayLength));
}

__global__ __launch_bounds__(256) void k1_ssgi_march(K1Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k1_ssgi_march_body(A, d);
    rfx_flush_violations(d);
}

}  // namespace

hipError_t rfx_launch_k1(const K1Args &A, hipStream_t stream) {
    dim3 block(64, 4), grid((A.dims.W + 63) / 64, (A.y1 - A.y0 + 3) / 4);
    hipLaunchKernelGGL(k1_ssgi_march, grid, block, 0, stream, A);
    return hipGetLastError();
}
