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

// ---------------------------------------------------------------- CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42)
// One `textureCube(cubeMap, dir)` per texel of the equirectangular render target, with the lookup rules of the oracle's GL (measured
// bit-exact on every interior lookup, oracle/glref/probes/probe_cube.py): major axis by >= in x, y, z order,
// (s, t) = (sc * (1 / ma)) * 0.5 + 0.5, seamless edges (a footprint texel beyond the face edge is the neighbouring face's texel the
// extended position projects to; beyond a corner: the average of the other three), blend = fused lerp in x, then in y.
struct K0Cube {
    const float4 *chain;  // level l at chain + off[l]: six faces +X -X +Y -Y +Z -Z, each (S >> l) x (S >> l), row j = t
    unsigned int off[14];
    int S, levels;        // levels = 1: the cube has no mip chain (minFilter LinearFilter)
    float4 *out;
    int W, H;
    UvPlanes uv;
};
struct CubeLevel { const float4 *faces; int S; };
struct CubeFace { int face; float s, t; };
RFX_DEV CubeFace k0_cube_face(float x, float y, float z) {
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    CubeFace r;
    float sc, tc, ma;
    if (ax >= ay && ax >= az) { r.face = x >= 0.0f ? 0 : 1; sc = x >= 0.0f ? -z : z; tc = -y; ma = ax; }
    else if (ay >= az) { r.face = y >= 0.0f ? 2 : 3; sc = x; tc = y >= 0.0f ? z : -z; ma = ay; }
    else { r.face = z >= 0.0f ? 4 : 5; sc = z >= 0.0f ? x : -x; tc = -y; ma = az; }
    const float ima = 1.0f / ma;  // IEEE division, as the GL's
    r.s = (sc * ima) * 0.5f + 0.5f;
    r.t = (tc * ima) * 0.5f + 0.5f;
    return r;
}
// texel (i, j) of the footprint on `face`; false for the texel beyond a corner
RFX_DEV bool k0_cube_texel(const CubeLevel &A, int face, int i, int j, float4 &out) {
    const int S = A.S;
    const bool oi = i < 0 || i >= S, oj = j < 0 || j >= S;
    if (oi && oj) return false;
    if (oi || oj) {  // the centre of the would-be texel, projected onto the neighbouring face
        const float a = (((float)i + 0.5f) / (float)S) * 2.0f - 1.0f, b = (((float)j + 0.5f) / (float)S) * 2.0f - 1.0f;
        float x, y, z;
        switch (face) {
        case 0: x = 1.0f; y = -b; z = -a; break;
        case 1: x = -1.0f; y = -b; z = a; break;
        case 2: x = a; y = 1.0f; z = b; break;
        case 3: x = a; y = -1.0f; z = -b; break;
        case 4: x = a; y = -b; z = 1.0f; break;
        default: x = -a; y = -b; z = -1.0f; break;
        }
        const CubeFace n = k0_cube_face(x, y, z);
        face = n.face;
        i = min(max((int)floorf(n.s * (float)S), 0), S - 1);
        j = min(max((int)floorf(n.t * (float)S), 0), S - 1);
    }
    out = A.faces[((size_t)face * S + j) * S + i];
    return true;
}
// seamless bilinear lookup of one level
RFX_DEV float4 k0_cube_linear(const CubeLevel &L, float dx, float dy, float dz) {
    const CubeFace f = k0_cube_face(dx, dy, dz);
    const float cu = f.s * (float)L.S - 0.5f, cv = f.t * (float)L.S - 0.5f;
    const float fi = floorf(cu), fj = floorf(cv), fu = cu - fi, fv = cv - fj;
    const int i0 = (int)fi, j0 = (int)fj;
    float4 q[4];
    bool have[4];
    have[0] = k0_cube_texel(L, f.face, i0, j0, q[0]);
    have[1] = k0_cube_texel(L, f.face, i0 + 1, j0, q[1]);
    have[2] = k0_cube_texel(L, f.face, i0, j0 + 1, q[2]);
    have[3] = k0_cube_texel(L, f.face, i0 + 1, j0 + 1, q[3]);
    int missing = -1;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (!have[k]) missing = k;
    if (missing >= 0) {  // beyond the corner: the average of the three texels that exist (summed in footprint order)
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k != missing) { sum.x += q[k].x; sum.y += q[k].y; sum.z += q[k].z; sum.w += q[k].w; }
        const float4 avg = make_float4(sum.x / 3.0f, sum.y / 3.0f, sum.z / 3.0f, sum.w / 3.0f);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k == missing) q[k] = avg;
    }
    float4 o;
    float top, bot;
    top = __fmaf_rn(fu, q[1].x - q[0].x, q[0].x); bot = __fmaf_rn(fu, q[3].x - q[2].x, q[2].x); o.x = __fmaf_rn(fv, bot - top, top);
    top = __fmaf_rn(fu, q[1].y - q[0].y, q[0].y); bot = __fmaf_rn(fu, q[3].y - q[2].y, q[2].y); o.y = __fmaf_rn(fv, bot - top, top);
    top = __fmaf_rn(fu, q[1].z - q[0].z, q[0].z); bot = __fmaf_rn(fu, q[3].z - q[2].z, q[2].z); o.z = __fmaf_rn(fv, bot - top, top);
    top = __fmaf_rn(fu, q[1].w - q[0].w, q[0].w); bot = __fmaf_rn(fu, q[3].w - q[2].w, q[2].w); o.w = __fmaf_rn(fv, bot - top, top);
    return o;
}
// the pass's direction at fragment (x, y).  A one-off set-up pass whose direction error is multiplied by the face size in texels and the
// cube's local contrast: the library sin / cos (<= 1-2 ulp), not the hardware approximations the per-frame kernels use
RFX_DEV float3 k0_cube_direction(const UvPlanes &uv, int x, int y) {
    const float PI = 3.1415926535897932384626433832795f;
    const float u = rfx_frag_u(uv, x, y), v = rfx_frag_v(uv, y);
    const float longitude = ((u * 2.0f) * PI - PI) + PI / 2.0f, latitude = v * PI;
    const float sLat = sinf(latitude);
    return make_float3(-sinf(longitude) * sLat, -cosf(latitude), -cosf(longitude) * sLat);  // dir.y = -dir.y
}
RFX_DEV void k0_cube_components(int face, float3 p, float &sc, float &tc, float &ma) {
    switch (face) {
    case 0: sc = -p.z; tc = -p.y; ma = p.x; break;
    case 1: sc = p.z; tc = -p.y; ma = -p.x; break;
    case 2: sc = p.x; tc = p.z; ma = p.y; break;
    case 3: sc = p.x; tc = -p.z; ma = -p.y; break;
    case 4: sc = p.x; tc = -p.y; ma = p.z; break;
    default: sc = -p.x; tc = -p.y; ma = -p.z; break;
    }
}
// textureCube's implicit level of detail as the oracle's GL derives it: per pixel, from the direction differences within the pixel's own
// row and own column of its 2x2 quad, through the quotient rule on the pixel's own face; linear-mantissa log2
RFX_DEV float k0_cube_lod(float3 p, float3 ddx, float3 ddy, int S) {
    const CubeFace f = k0_cube_face(p.x, p.y, p.z);
    float sc, tc, ma, xsc, xtc, xma, ysc, ytc, yma;
    k0_cube_components(f.face, p, sc, tc, ma);
    k0_cube_components(f.face, ddx, xsc, xtc, xma);
    k0_cube_components(f.face, ddy, ysc, ytc, yma);
    const float ima = 1.0f / ma, k = (ima * ima) * 0.5f;
    const float dsx = (xsc * ma - sc * xma) * k, dtx = (xtc * ma - tc * xma) * k;
    const float dsy = (ysc * ma - sc * yma) * k, dty = (ytc * ma - tc * yma) * k;
    const float rho2 = fmaxf(dsx * dsx + dtx * dtx, dsy * dsy + dty * dty) * ((float)S * (float)S);
    const uint32_t bits = __float_as_uint(rho2);
    const float mant = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u);
    return 0.5f * ((float)((int)((bits >> 23) & 0xffu) - 127) + (mant - 1.0f));
}
__global__ __launch_bounds__(256) void k0_cube_to_equirect(K0Cube A) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= A.W || y >= A.H) return;
    const float3 p = k0_cube_direction(A.uv, x, y);
    const CubeLevel base = {A.chain, A.S};
    float4 o;
    if (A.levels == 1) {
        o = k0_cube_linear(base, p.x, p.y, p.z);
    } else {
        const float3 pr = k0_cube_direction(A.uv, x ^ 1, y), pc = k0_cube_direction(A.uv, x, y ^ 1);
        const float3 ddx = (x & 1) ? make_float3(p.x - pr.x, p.y - pr.y, p.z - pr.z) : make_float3(pr.x - p.x, pr.y - p.y, pr.z - p.z);
        const float3 ddy = (y & 1) ? make_float3(p.x - pc.x, p.y - pc.y, p.z - pc.z) : make_float3(pc.x - p.x, pc.y - p.y, pc.z - p.z);
        float lod = k0_cube_lod(p, ddx, ddy, A.S);
        lod = fminf(fmaxf(lod, 0.0f), (float)(A.levels - 1));
        const float fl = floorf(lod), f = lod - fl;
        const int l0 = (int)fl, l1 = min(l0 + 1, A.levels - 1);
        const CubeLevel L0 = {A.chain + A.off[l0], max(A.S >> l0, 1)}, L1 = {A.chain + A.off[l1], max(A.S >> l1, 1)};
        const float4 c0 = k0_cube_linear(L0, p.x, p.y, p.z), c1 = k0_cube_linear(L1, p.x, p.y, p.z);
        o = make_float4(__fmaf_rn(f, c1.x - c0.x, c0.x), __fmaf_rn(f, c1.y - c0.y, c0.y), __fmaf_rn(f, c1.z - c0.z, c0.z), __fmaf_rn(f, c1.w - c0.w, c0.w));
    }
    A.out[(size_t)y * A.W + x] = o;
}
// one level of the cube's chain from the one above, per face (glGenerateMipmap on the oracle's GL: the 2x2 bilinear centre)
__global__ __launch_bounds__(256) void k0_cube_mip(const float4 *src, float4 *dst, int s0, int s1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 6 * s1 * s1) return;
    const int f = i / (s1 * s1), r = i - f * s1 * s1, y = r / s1, x = r - y * s1;
    const float4 *q = src + (size_t)f * s0 * s0;
    if (s0 & 1) {  // a level of odd size (a face that is not a power of two): the GL's LINEAR blit of the face at the target texel's centre, CLAMP_TO_EDGE
        int x0, x1, y0, y1;
        float wx, wy;
        rfx_linear_coord(((float)x + 0.5f) / (float)s1, (float)s0, s0, x0, x1, wx);
        rfx_linear_coord(((float)y + 0.5f) / (float)s1, (float)s0, s0, y0, y1, wy);
        const float4 t00 = q[(size_t)y0 * s0 + x0], t10 = q[(size_t)y0 * s0 + x1], t01 = q[(size_t)y1 * s0 + x0], t11 = q[(size_t)y1 * s0 + x1];
        dst[i] = make_float4(rfx_lerp(wy, rfx_lerp(wx, t00.x, t10.x), rfx_lerp(wx, t01.x, t11.x)), rfx_lerp(wy, rfx_lerp(wx, t00.y, t10.y), rfx_lerp(wx, t01.y, t11.y)),
                             rfx_lerp(wy, rfx_lerp(wx, t00.z, t10.z), rfx_lerp(wx, t01.z, t11.z)), rfx_lerp(wy, rfx_lerp(wx, t00.w, t10.w), rfx_lerp(wx, t01.w, t11.w)));
        return;
    }
    const float4 a = q[(size_t)(2 * y) * s0 + 2 * x], b = q[(size_t)(2 * y) * s0 + 2 * x + 1];
    const float4 c = q[(size_t)(2 * y + 1) * s0 + 2 * x], e = q[(size_t)(2 * y + 1) * s0 + 2 * x + 1];
    float4 o;
    o.x = rfx_lerp(0.5f, rfx_lerp(0.5f, a.x, b.x), rfx_lerp(0.5f, c.x, e.x));
    o.y = rfx_lerp(0.5f, rfx_lerp(0.5f, a.y, b.y), rfx_lerp(0.5f, c.y, e.y));
    o.z = rfx_lerp(0.5f, rfx_lerp(0.5f, a.z, b.z), rfx_lerp(0.5f, c.z, e.z));
    o.w = rfx_lerp(0.5f, rfx_lerp(0.5f, a.w, b.w), rfx_lerp(0.5f, c.w, e.w));
    dst[i] = o;
}

}  // namespace

// `chain` holds level 0 (6 * size * size texels) and room for the further levels (size >> l, at least 1), built here
hipError_t rfx_launch_cube_to_equirect(float4 *chain, int size, int levels, float4 *out, int W, int H, const UvPlanes &uv, hipStream_t stream) {
    K0Cube A;
    A.chain = chain; A.S = size; A.levels = levels; A.out = out; A.W = W; A.H = H; A.uv = uv;
    unsigned int off = 0;
    for (int l = 0; l < 14; l++) {
        A.off[l] = off;
        const int s = (size >> l) > 0 ? size >> l : 1;
        off += 6u * (unsigned int)s * (unsigned int)s;
    }
    for (int l = 1; l < levels; l++) {
        const int s0 = size >> (l - 1), s1 = size >> l;
        hipLaunchKernelGGL(k0_cube_mip, dim3((6 * s1 * s1 + 255) / 256), dim3(256), 0, stream, (const float4 *)chain + A.off[l - 1], chain + A.off[l], s0, s1);
    }
    dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
    hipLaunchKernelGGL(k0_cube_to_equirect, grid, block, 0, stream, A);
    return hipGetLastError();
}

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
