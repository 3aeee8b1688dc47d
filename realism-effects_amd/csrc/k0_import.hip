// K0 — importer: engine-side attribute planes (AOVs) -> the reference's packed render targets, on the device.
// Replaces the fragment epilogues of the two raster passes for an importer of engine dumps:
//   GBufferMaterial.js:84-89            gl_FragColor = packGBuffer(diffuseColor, worldNormal, roughnessFactor, metalnessFactor, totalEmissiveRadiance)
//   VelocityDepthNormalMaterial.js:76-83,186-188   gl_FragColor = vec4(vel.x, vel.y, packNormal(worldNormal), fragCoordZ)
// with the encode side of src/gbuffer/shader/gbuffer_packing.glsl (vec4ToFloat :143-149, packNormal / encodeOctWrap :44-61,
// color2float :17-22, encodeRGBE8 :127-134, packGBuffer :166-178).  Texels the rasteriser would not cover (depth == 1) keep the
// passes' clear colour, `scene.background = Color(0)` with alpha 1 (GBufferPass.js:103-105, VelocityDepthNormalPass.js:177-179).
// Streaming: 13 floats in, 16 B out per pixel (G-buffer); 6 floats in, 16 B out (velocity).
#include "rfx_device.h"
#include "rfx_kernels.h"

namespace {

RFX_DEV uint32_t k0_vec4_to_float(float x, float y, float z, float w) {  // vec4ToFloat
    const float o = 0.0001f, one_safe = 0.999999f;
    // min() as the GLSL evaluates it: NaN (the 0/0 of encodeRGBE8 on a black emissive, Appendix D-9) does not propagate
    const uint32_t r = (uint32_t)(fminf(x + o, one_safe) * 255.0f), g = (uint32_t)(fminf(y + o, one_safe) * 255.0f);
    const uint32_t b = (uint32_t)(fminf(z + o, one_safe) * 255.0f), a = (uint32_t)(fminf(w + o, one_safe) * 255.0f);
    return (a << 24) | (b << 16) | (g << 8) | r;
}
RFX_DEV uint32_t k0_pack_normal(float3 n) {  // packNormal(encodeOctWrap(n))
    const float s = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    n.x /= s; n.y /= s; n.z /= s;
    float wx = 1.0f - fabsf(n.y), wy = 1.0f - fabsf(n.x);  // OctWrap
    if (n.x < 0.0f) wx = -wx;
    if (n.y < 0.0f) wy = -wy;
    const float ox = n.z > 0.0f ? n.x : wx, oy = n.z > 0.0f ? n.y : wy;
    return rfx_pack_half2(ox * 0.5f + 0.5f, oy * 0.5f + 0.5f);
}
RFX_DEV float k0_color2float(float r, float g, float b) {  // color2float
    const float o = 0.0001f, one_safe = 0.999999f, P = 256.0f, P1 = 257.0f;
    r = fminf(r + o, one_safe); g = fminf(g + o, one_safe); b = fminf(b + o, one_safe);
    return floorf(r * P + 0.5f) + floorf(b * P + 0.5f) * P1 + floorf(g * P + 0.5f) * P1 * P1;
}
RFX_DEV uint32_t k0_rgbe8(float3 c) {  // vec4ToFloat(encodeRGBE8(rgb))
    const float mx = fmaxf(fmaxf(c.x, c.y), c.z);
    const float fexp = ceilf(rfx_log2(mx));
    const float sc = rfx_exp2(fexp);
    return k0_vec4_to_float(c.x / sc, c.y / sc, c.z / sc, (fexp + 128.0f) / 255.0f);
}

struct K0GBuffer {
    int W, rows;
    const float *diffuse, *normal, *roughness, *metalness, *emissive, *depth;  // device staging planes of `rows` rows
    uint4 *out;                                                                // first output row
};
__global__ __launch_bounds__(256) void k0_pack_gbuffer(K0GBuffer A) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.W || y >= A.rows) return;
    const size_t i = (size_t)y * A.W + x;
    uint4 o;
    if (A.depth && A.depth[i] == 1.0f) {
        o = make_uint4(0u, 0u, 0u, 0x3f800000u);  // the clear colour (0, 0, 0, 1)
    } else {
        o.x = k0_vec4_to_float(A.diffuse[4 * i], A.diffuse[4 * i + 1], A.diffuse[4 * i + 2], A.diffuse[4 * i + 3]);
        o.y = k0_pack_normal(make_float3(A.normal[3 * i], A.normal[3 * i + 1], A.normal[3 * i + 2]));
        o.z = __float_as_uint(k0_color2float(A.roughness[i], A.metalness[i], 0.0f));
        o.w = k0_rgbe8(make_float3(A.emissive[3 * i], A.emissive[3 * i + 1], A.emissive[3 * i + 2]));
    }
    A.out[i] = o;
}

struct K0Velocity {
    int W, rows;
    const float *velocity, *normal, *depth;
    uint4 *out;
};
__global__ __launch_bounds__(256) void k0_pack_velocity(K0Velocity A) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.W || y >= A.rows) return;
    const size_t i = (size_t)y * A.W + x;
    const float d = A.depth[i];
    uint4 o;
    if (d == 1.0f) o = make_uint4(0u, 0u, 0u, 0x3f800000u);
    else
        o = make_uint4(__float_as_uint(A.velocity[2 * i]), __float_as_uint(A.velocity[2 * i + 1]),
                       k0_pack_normal(make_float3(A.normal[3 * i], A.normal[3 * i + 1], A.normal[3 * i + 2])), __float_as_uint(d));
    A.out[i] = o;
}

}  // namespace

hipError_t rfx_launch_pack_gbuffer(int W, int rows, const float *diffuse, const float *normal, const float *roughness, const float *metalness,
                                   const float *emissive, const float *depth, void *out, hipStream_t stream) {
    K0GBuffer A = {W, rows, diffuse, normal, roughness, metalness, emissive, depth, (uint4 *)out};
    dim3 block(64, 4), grid((W + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k0_pack_gbuffer, grid, block, 0, stream, A);
    return hipGetLastError();
}
hipError_t rfx_launch_pack_velocity(int W, int rows, const float *velocity, const float *normal, const float *depth, void *out, hipStream_t stream) {
    K0Velocity A = {W, rows, velocity, normal, depth, (uint4 *)out};
    dim3 block(64, 4), grid((W + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k0_pack_velocity, grid, block, 0, stream, A);
    return hipGetLastError();
}
