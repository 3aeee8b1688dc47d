// k4_compose_texel.h — the arithmetic of DenoiserComposePass's fragment for ONE texel whose inputs are at hand: the inline shader
// src/denoise/pass/DenoiserComposePass.js:66-85 and constructGlobalIllumination, src/denoise/shader/denoiser_compose_functions.glsl:53-108.
// Used by K4 (k4_compose.hip).  (Rounds 4-5 also compiled it into the last K3 launch, the compose fold that left the library in round 6: DESIGN.md §4 K4.)
#pragma once
#include "rfx_brdf.h"
#include "rfx_kernels.h"

// dgi / sgi: the diffuse / specular GI texels (rgb); scene: sceneTexture's texel, read only by inputType "specular" (2)
RFX_DEV float4 k4_compose_texel(const rfx_compose_params &p, float u, float v, float depth, const Material &mat, float3 dgi, float3 sgi, float3 scene) {
    const float *C = p.camera.matrixWorld, *Vw = p.camera.matrixWorldInverse;
    const float *P = p.camera.projectionMatrix, *Pi = p.camera.projectionMatrixInverse;
    const float3 viewNormal = rfx_vec_mul_mat(C, mat.normal, 0.0f);  // :71 (not normalised)
    const float n_ = p.camera.near_, f_ = p.camera.far_;
    const float viewZ = -rfx_depth_to_view_z(depth, n_, f_, p.camera.isPerspective != 0);  // -getViewZ(depth) :73
    const float clipW = P[2 * 4 + 3] * viewZ + P[3 * 4 + 3];
    const float4 pp = rfx_mat_mul(Pi, ((u - 0.5f) * 2.0f) * clipW, ((v - 0.5f) * 2.0f) * clipW, ((viewZ - 0.5f) * 2.0f) * clipW, 1.0f * clipW);
    const float3 viewDir = rfx_normalize(make_float3(pp.x, pp.y, -viewZ));

    // constructGlobalIllumination
    const float roughness = mat.roughness * mat.roughness;
    const float3 normal = rfx_vec_mul_mat(Vw, viewNormal, 0.0f);
    const float3 vv = -viewDir;
    float3 V = rfx_vec_mul_mat(Vw, vv, 0.0f);
    float3 T, B;
    rfx_onb(normal, T, B);
    V = rfx_to_local(T, B, normal, V);
    float3 H = rfx_sample_ggx_vndf(V, roughness, roughness, 0.25f, 0.25f);
    if (H.z < 0.0f) H = -H;
    float3 l = rfx_normalize(rfx_reflect(-V, H));
    l = rfx_to_world(T, B, normal, l);
    l = rfx_normalize(rfx_vec_mul_mat(C, l, 1.0f));  // vec4(l, 1.) quirk :81
    if (rfx_dot(viewNormal, l) < 0.0f) l = -l;
    const float3 h = rfx_normalize(vv + l);
    const float VoH = fmaxf(1e-6f, rfx_dot(vv, h));
    const float3 f0 = rfx_mix(make_float3(0.04f, 0.04f, 0.04f), mat.diffuse, mat.metalness);
    const float3 F = rfx_f_schlick(f0, VoH);
    const float om = 1.0f - mat.metalness;
    float3 dc = make_float3(mat.diffuse.x * om * (1.0f - F.x) * dgi.x, mat.diffuse.y * om * (1.0f - F.y) * dgi.y, mat.diffuse.z * om * (1.0f - F.z) * dgi.z);
    if (p.inputType == 2) dc = scene;  // denoiser_compose_functions.glsl:97-101: diffuseComponent = textureLod(sceneTexture, vUv, 0.).rgb
    float4 o;
    o.x = (dc.x + sgi.x * F.x) + mat.emissive.x;
    o.y = (dc.y + sgi.y * F.y) + mat.emissive.y;
    o.z = (dc.z + sgi.z * F.z) + mat.emissive.z;
    o.w = 1.0f;
    return o;
}
