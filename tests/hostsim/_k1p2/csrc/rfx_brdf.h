// rfx_brdf.h — closed-form BRDF / sampling math used by K1 (ssgi.frag) and K4 (compose).
// Follows src/ssgi/shader/ssgi_utils.frag:108-191 and the duplicated helpers in
// src/denoise/shader/denoiser_compose_functions.glsl:22-51.
#pragma once
#include "rfx_device.h"

#define RFX_PI 3.1415926535897932384626433832795f

RFX_DEV float rfx_pow5(float x) { return rfx_pow(x, 5.0f); }
// F_Schlick(vec3 f0, theta)  ssgi_utils.frag:108
RFX_DEV float3 rfx_f_schlick(float3 f0, float theta) {
    float p = rfx_pow5(1.0f - theta);
    return make_float3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
}
// F_Schlick(f0, f90, theta)  :110
RFX_DEV float rfx_f_schlick(float f0, float f90, float theta) { return f0 + (f90 - f0) * rfx_pow5(1.0f - theta); }
// D_GTR(roughness, NoH, 2.)  :112-115
RFX_DEV float rfx_d_gtr2(float roughness, float NoH) {
    float a2 = roughness * roughness;
    float t = (NoH * NoH) * (a2 * a2 - 1.0f) + 1.0f;
    return rfx_div_pos(a2, RFX_PI * (t * t));  // t >= 1 - NoH^2 > 0 (NoH is clamped below 1)
}
// SmithG  :117-121
RFX_DEV float rfx_smith_g(float NDotV, float alphaG) {
    float a = alphaG * alphaG, b = NDotV * NDotV;
    return rfx_div_pos(2.0f * NDotV, NDotV + rfx_sqrt(a + b - a * b));  // NDotV >= 1e-5
}
// GGXVNDFPdf  :123-127
RFX_DEV float rfx_ggx_vndf_pdf(float NoH, float NoV, float roughness) {
    float D = rfx_d_gtr2(roughness, NoH);
    float G1 = rfx_smith_g(NoV, roughness * roughness);
    return rfx_div_pos(D * G1, fmaxf(0.00001f, 4.0f * NoV));
}
// evalDisneyDiffuse  :136-142 (all three channels are equal)
RFX_DEV float rfx_eval_disney_diffuse(float NoL, float NoV, float LoH, float roughness, float metalness) {
    float FD90 = 0.5f + 2.0f * roughness * (LoH * LoH);
    float a = rfx_f_schlick(1.0f, FD90, NoL), b = rfx_f_schlick(1.0f, FD90, NoV);
    return RFX_DIV_CONST(a * b, RFX_PI) * (1.0f - metalness);
}
// evalDisneySpecular  :144-151 with GeometryTerm :129-134
RFX_DEV float rfx_eval_disney_specular(float roughness, float NoH, float NoV, float NoL) {
    float D = rfx_d_gtr2(roughness, NoH);
    float r2 = 0.5f + roughness * 0.5f;
    r2 = r2 * r2;
    float a2 = r2 * r2;
    float G = rfx_smith_g(NoV, a2) * rfx_smith_g(NoL, a2);
    return rfx_div_pos(D * G, 4.0f * NoL * NoV);  // NoL, NoV >= 1e-5
}
// SampleGGXVNDF  :153-170
RFX_DEV float3 rfx_sample_ggx_vndf(float3 V, float ax, float ay, float r1, float r2) {
    float3 Vh = rfx_normalize(make_float3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    float3 T1;
    if (lensq > 0.0f) {
        float is = rfx_rsqrt(lensq);
        T1 = make_float3(-Vh.y * is, Vh.x * is, 0.0f * is);
    } else {
        T1 = make_float3(1.0f, 0.0f, 0.0f);
    }
    float3 T2 = rfx_cross(Vh, T1);
    float r = rfx_sqrt(r1);
    float phi = 2.0f * RFX_PI * r2;
    float sp, cp;
    rfx_sincos(phi, sp, cp);
    float t1 = r * cp, t2 = r * sp;
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * rfx_sqrt(1.0f - t1 * t1) + s * t2;
    float k = rfx_sqrt(fmaxf(0.0f, 1.0f - t1 * t1 - t2 * t2));
    float3 Nh = (T1 * t1 + T2 * t2) + Vh * k;
    return rfx_normalize(make_float3(ax * Nh.x, ay * Nh.y, fmaxf(0.0f, Nh.z)));
}
// Onb  :172-176
RFX_DEV void rfx_onb(float3 N, float3 &T, float3 &B) {
    float3 up = fabsf(N.z) < 0.9999999f ? make_float3(0.f, 0.f, 1.f) : make_float3(1.f, 0.f, 0.f);
    T = rfx_normalize(rfx_cross(up, N));
    B = rfx_cross(N, T);
}
RFX_DEV float3 rfx_to_local(float3 X, float3 Y, float3 Z, float3 V) { return make_float3(rfx_dot(V, X), rfx_dot(V, Y), rfx_dot(V, Z)); }
RFX_DEV float3 rfx_to_world(float3 X, float3 Y, float3 Z, float3 V) { return (X * V.x + Y * V.y) + Z * V.z; }
RFX_DEV float3 rfx_reflect(float3 I, float3 N) { return I - N * (2.0f * rfx_dot(N, I)); }
// cosineSampleHemisphere  :183-191
RFX_DEV float3 rfx_cosine_sample_hemisphere(float3 n, float ux, float uy) {
    float r = rfx_sqrt(ux), theta = 2.0f * RFX_PI * uy;
    float st, ct;
    rfx_sincos(theta, st, ct);
    float3 b = rfx_normalize(rfx_cross(n, make_float3(0.0f, 1.0f, 1.0f)));
    float3 t = rfx_cross(b, n);
    return rfx_normalize((b * (r * st) + n * rfx_sqrt(1.0f - ux)) + t * (r * ct));
}
