// rfx_device.h — device-side building blocks shared by the four kernels (gfx950 only).
//
// Everything here is the GPU statement of semantics the reference's GLSL relies on:
// texel codecs (src/gbuffer/shader/gbuffer_packing.glsl), the blue-noise RNG
// (src/utils/shader/blue_noise.glsl), texture addressing rules (nearest / bilinear,
// CLAMP_TO_EDGE) and the half-float rounding modes of packHalf2x16 and RGBA16F stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rfx.h"

#define RFX_DEV __device__ __forceinline__
// register-allocation bound of a kernel: at least n waves per SIMD (n = 8: at most 64 VGPRs)
#ifndef RFX_WAVES_PER_EU
#define RFX_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif

// A persistent kernel runs its per-pixel body in a loop; left alone, the compiler hoists every scalar load of the argument block (four camera
// matrices, the views, the options) out of that loop and keeps them live across the whole body — far more than the 102 SGPRs there are, so
// they spill into VGPR lanes and from there to scratch.  This returns the kernel's argument block (the FIRST kernel parameter, passed by value:
// offset 0 of the kernarg segment) through a pointer the optimiser cannot see through, once per loop iteration: the loads stay scalar loads
// next to their uses, as in a kernel without the loop.
#ifndef RFX_KERNARGS_IN_LOOP
template <class T>
__device__ __forceinline__ const T &rfx_kernargs_in_loop(const T &) {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();  // (a pointer into the constant address space)
    asm volatile("" : "+s"(p));
    return *(const T *)p;
}
#define RFX_KERNARGS_IN_LOOP(A) rfx_kernargs_in_loop(A)
#endif

// The point where the lanes of a wavefront meet again after a region some of them left early (a `return` out of an inlined per-pixel body that
// runs in a loop).  The hardware's exec mask does this by itself — this is a compiler-level marker only (no instruction); it exists so that the
// tests' host simulator, which runs lanes as independent fibers, can model the reconvergence when the region contains wave operations.
#ifndef RFX_WAVE_JOIN
#define RFX_WAVE_JOIN() __builtin_amdgcn_wave_barrier()
#endif

// ---------------------------------------------------------------- texture views
// A view addresses rows [row0, row0+rows) of a W x H frame held contiguously in HBM.
// Fetch coordinates are FRAME coordinates: CLAMP_TO_EDGE happens against the frame, then the
// row is rebased into the held band.  A row outside the band is a halo violation: the access
// is clamped into the band (memory-safe) and counted.
struct TexView {
    const void *ptr;
    int row0, rows;
};
struct TexViewW {
    void *ptr;
    int row0, rows;
};
// ---------------------------------------------------------------- the fragment's vUv
// RFX_UV_IDEAL: (i + 0.5) / n, correctly rounded.  RFX_UV_REFERENCE_GL: what the rasteriser of the reference's GL (Mesa llvmpipe, the
// oracle of SURVEY.md 8c) interpolates for three's full-screen triangle, bit for bit: the triangle (-1,-1) (3,-1) (-1,3) leaves the guard
// band and is clipped to the viewport, so the frame is drawn as two triangles split along the diagonal (0,0)-(W,H), each with its own
// fp32 plane equations a0 + du * x (+ dv * y) evaluated with fma on the integer pixel position (oracle/rfx_oracle.c frag_u / frag_v,
// oracle/glref/probes/probe_varying.py: exact on every fragment of every size tried).  The host fills the planes (rfx_uv_planes).
struct UvPlanes {
    int model;
    int W, H;
    float fW, fH;
    float du, dv;            // H * (1 / (W * H)), W * (1 / (W * H)), every product rounded
    float u0_upper, u0_lower;  // du / 2 above the diagonal (provoking vertex (0,H)); 1 - du * (W - 0.5) on and below it (vertex (W,H))
    float v0;                // 1 - dv * (H - 0.5) in both triangles
};
__device__ __forceinline__ float rfx_frag_u(const UvPlanes &q, int x, int y) {
    if (q.model == RFX_UV_IDEAL) return ((float)x + 0.5f) / q.fW;
    const bool upper = __mul24(2 * y + 1, q.W) > __mul24(2 * x + 1, q.H);  // < 2^31: rfx_create bounds W, H
    return __fmaf_rn(q.du, (float)x, upper ? q.u0_upper : q.u0_lower);
}
__device__ __forceinline__ float rfx_frag_v(const UvPlanes &q, int y) {
    if (q.model == RFX_UV_IDEAL) return ((float)y + 0.5f) / q.fH;
    return __fmaf_rn(q.dv, (float)y, q.v0);
}

struct FrameDims {
    int W, H;
    float fW, fH;
    UvPlanes uv;  // vUv of a frame-sized render target
    unsigned int *halo_violations;  // device counter (may be null)
    mutable unsigned int viol;      // per-lane sticky flag, flushed once by rfx_flush_violations()
};

// branch-free: clamp into the held band and remember that a clamp happened
RFX_DEV int rfx_local_row(const FrameDims &d, int row0, int rows, int y) {
    const int l = y - row0;
    const int c = min(max(l, 0), rows - 1);
    d.viol |= (unsigned int)(l != c);
    return c;
}
RFX_DEV void rfx_flush_violations(const FrameDims &d) {
    if (d.viol && d.halo_violations) atomicAdd(d.halo_violations, 1u);
}

// ---------------------------------------------------------------- XCD-aware tile order (speed only, never correctness)
// The LDS-tiled kernels (K2, K3) re-read an apron around every tile: 1.8-2.4 staged texels per produced pixel.  Hardware block
// b runs on XCD b % 8 (observed; MI355X_MICROARCH.md "Workgroup dispatch"), each XCD has its own 4 MiB L2, so with a plain 2-D
// grid the eight neighbours of a tile sit on other XCDs and every apron texel crosses the fabric once per tile that stages it.
// Here a 1-D grid is folded so that XCD k owns every 8th GROUP of G tile rows, and walks a group column by column (down the G
// rows first): the 64-96 workgroups resident on an XCD at any time then cover a compact G x ~20-tile patch whose shared apron
// texels are L2 hits, while groups of all image regions stay interleaved over the XCDs (a band-per-XCD split idles the XCDs that
// own sky).  The launch pads the grid to 8 * ceil(groups / 8) * G * nbx blocks; a block whose tile row is past the end returns.
struct TileXY { int bx, by; bool valid; };
template <int G>
RFX_DEV TileXY rfx_xcd_tile(int nbx, int nby) {
    TileXY t;
    if (G <= 0) {  // plain row-major order over a 1-D grid
        t.by = blockIdx.x / nbx; t.bx = blockIdx.x - t.by * nbx;
        t.valid = t.by < nby;
        return t;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_group = G * nbx;
    const int gi = slot / per_group, within = slot - gi * per_group;
    const int group = gi * 8 + xcd;
    t.bx = within / G;
    t.by = group * G + (within - t.bx * G);
    t.valid = t.by < nby;
    return t;
}
inline int rfx_xcd_grid(int G, int nbx, int nby) {
    if (G <= 0) return nbx * nby;
    const int groups = (nby + G - 1) / G;
    return 8 * ((groups + 7) / 8) * G * nbx;
}

// nearest CLAMP_TO_EDGE index as x86 cvttss2si + clamp computes it:
// NaN and |c| >= 2^31 give INT_MIN -> texel 0 (AMD's v_cvt_i32_f32 would saturate to size-1).
RFX_DEV int rfx_nearest_idx(float u, float fsize, int size) {
    float c = u * fsize;
    c = (c < 2147483648.0f) ? c : 0.0f;  // NaN compares false -> 0
    // clamp in float, then truncate: the same index as max((int)c, 0) then min(.., size - 1) for every finite c (size <= 2^24)
    return (int)__builtin_amdgcn_fmed3f(c, 0.0f, (float)(size - 1));
}

RFX_DEV size_t rfx_texel_index(const FrameDims &d, int row0, int rows, float u, float v) {
    int x = rfx_nearest_idx(u, d.fW, d.W);
    int y = rfx_nearest_idx(v, d.fH, d.H);
    // rows and widths are < 2^23 and a plane < 2^28 texels (rfx_create): the full-rate 24-bit multiply, not a 64-bit mad
    return (size_t)(unsigned int)(__mul24(rfx_local_row(d, row0, rows, y), d.W) + x);
}

RFX_DEV float rfx_fetch_r32f(const TexView &t, const FrameDims &d, float u, float v) {
    return ((const float *)t.ptr)[rfx_texel_index(d, t.row0, t.rows, u, v)];
}
RFX_DEV uint4 rfx_fetch_u4(const TexView &t, const FrameDims &d, float u, float v) {
    return ((const uint4 *)t.ptr)[rfx_texel_index(d, t.row0, t.rows, u, v)];
}
RFX_DEV float4 rfx_fetch_f4(const TexView &t, const FrameDims &d, float u, float v) {
    return ((const float4 *)t.ptr)[rfx_texel_index(d, t.row0, t.rows, u, v)];
}
// integer-addressed variants (pixel centres: the nearest fetch at vUv is the texel itself)
RFX_DEV size_t rfx_xy_index(const FrameDims &d, int row0, int rows, int x, int y) {
    x = min(max(x, 0), d.W - 1);
    y = min(max(y, 0), d.H - 1);
    return (size_t)(unsigned int)(__mul24(rfx_local_row(d, row0, rows, y), d.W) + x);
}

// ---------------------------------------------------------------- half floats
RFX_DEV float rfx_h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
RFX_DEV uint32_t rfx_f2h_rne(float f) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f); }  // v_cvt_f16_f32, RNE
// packHalf2x16: round-to-nearest-even
RFX_DEV uint32_t rfx_pack_half2(float a, float b) {
    return rfx_f2h_rne(a) | (rfx_f2h_rne(b) << 16);
}
RFX_DEV void rfx_unpack_half2(uint32_t u, float &a, float &b) {
    a = rfx_h2f((unsigned short)(u & 0xffffu));
    b = rfx_h2f((unsigned short)(u >> 16));
}
// RGBA16F render-target store.  RTZ = what the llvmpipe oracle does (vcvtps2ph imm 3,
// Appendix C-3): v_cvt_pkrtz_f16_f32 truncates and saturates finite overflow at 65504.
typedef __fp16 rfx_half2_t __attribute__((ext_vector_type(2)));
RFX_DEV uint2 rfx_store_half4(float x, float y, float z, float w, bool rtz) {
    uint2 r;
    if (rtz) {
        rfx_half2_t a = __builtin_amdgcn_cvt_pkrtz(x, y);
        rfx_half2_t b = __builtin_amdgcn_cvt_pkrtz(z, w);
        r.x = __builtin_bit_cast(uint32_t, a);
        r.y = __builtin_bit_cast(uint32_t, b);
    } else {
        r.x = rfx_pack_half2(x, y);
        r.y = rfx_pack_half2(z, w);
    }
    return r;
}
RFX_DEV float4 rfx_load_half4(uint2 t) {
    float4 r;
    rfx_unpack_half2(t.x, r.x, r.y);
    rfx_unpack_half2(t.y, r.z, r.w);
    return r;
}

// bilinear fetch of an RGBA16F target, CLAMP_TO_EDGE, as the llvmpipe sampler computes it:
//   c = min(u*size, size) - 0.5;  c = max(c, 0);  i0 = floor(c);  w = c - i0;  i1 = min(i0+1, size-1)
//   texel = lerp(wy, lerp(wx, t00, t10), lerp(wx, t01, t11)),  lerp(w,a,b) = a + w*(b-a)
RFX_DEV void rfx_linear_coord(float u, float fsize, int size, int &i0, int &i1, float &w) {
    float c = u * fsize;
    c = fminf(c, fsize);
    c = c - 0.5f;
    c = fmaxf(c, 0.0f);
    float fl = floorf(c);
    i0 = (int)fl;
    w = c - fl;
    i1 = min(i0 + 1, size - 1);
}
RFX_DEV float rfx_lerp(float w, float a, float b) { return a + w * (b - a); }
// ... and the form the LDS-tiled kernels use: one v_med3 + v_cvt + v_fract per axis instead of min, sub, max, floor, cvt, sub.
// c = u * size (the caller shares the product with its nearest taps).  Identical to rfx_linear_coord's (i0, w) for every finite c:
// x -> x - 0.5 is monotonic, so min(c, size) - 0.5 == min(c - 0.5, size - 0.5) (size - 0.5 is exact); c2 >= 0, so the truncating
// conversion is the floor; c2 - floor(c2) is exact in fp32, which is what v_fract_f32 returns.
struct LinearCoord {
    int i0;   // lower texel; the upper one is min(i0 + 1, size - 1)
    float w;  // weight of the upper texel
};
// lo: 0 for a frame; a tile kernel passes its staged window's first texel so that a NaN coordinate (v_med3 returns the lower bound) stays inside
// the window — for finite coordinates whose footprint the window holds, the same (i0, w)
RFX_DEV LinearCoord rfx_linear_coord_fast(float c, float lo, float size_minus_half) {
#pragma clang fp contract(off)  // c - 0.5 must not fuse with the product that formed c (K3 / K4 are compiled with contraction on)
    const float c2 = __builtin_amdgcn_fmed3f(c - 0.5f, lo, size_minus_half);
    LinearCoord r;
    r.i0 = (int)c2;
    r.w = __builtin_amdgcn_fractf(c2);
    return r;
}
RFX_DEV LinearCoord rfx_linear_coord_fast(float c, float size_minus_half) { return rfx_linear_coord_fast(c, 0.0f, size_minus_half); }
// The sampler's lerp on HALF texels without converting them first: v_fma_mix_f32 reads either half of a 32-bit register as an f16
// source of an fp32 fma.  d = fp32(b) - fp32(a) (one rounding, as v_sub_f32 on the converted values), then fma(w, d, fp32(a)) — the fused
// lerp of the oracle GL's sampler (oracle/rfx_oracle.c fetch_h4_linear).  Two instructions per channel instead of two v_cvt_f32_f16,
// a subtraction and an fma.  SEL: 0 = low half of the word, 1 = high half.
template <int SEL>
RFX_DEV float rfx_half_diff(uint32_t b, uint32_t a) {
    float r;
    if (SEL == 0) r = hostsim_half_diff(b, a, 0);
    else r = hostsim_half_diff(b, a, 1);
    return r;
}
template <int SEL>
RFX_DEV float rfx_half_fma(float w, float d, uint32_t a) {
    float r;
    if (SEL == 0) r = hostsim_half_fma(w, d, a, 0);
    else r = hostsim_half_fma(w, d, a, 1);
    return r;
}
template <int SEL>
RFX_DEV float rfx_half_lerp(float w, uint32_t a, uint32_t b) { return rfx_half_fma<SEL>(w, rfx_half_diff<SEL>(b, a), a); }
// bilinear blend of four RGBA16F texels (two 32-bit words each: r|g, b|a): lerp in x on both rows, then in y, every lerp fused
RFX_DEV float3 rfx_bilerp_half_rgb(uint2 t00, uint2 t10, uint2 t01, uint2 t11, float wx, float wy) {
    const float r0 = rfx_half_lerp<0>(wx, t00.x, t10.x), g0 = rfx_half_lerp<1>(wx, t00.x, t10.x), b0 = rfx_half_lerp<0>(wx, t00.y, t10.y);
    const float r1 = rfx_half_lerp<0>(wx, t01.x, t11.x), g1 = rfx_half_lerp<1>(wx, t01.x, t11.x), b1 = rfx_half_lerp<0>(wx, t01.y, t11.y);
    return make_float3(__builtin_fmaf(wy, r1 - r0, r0), __builtin_fmaf(wy, g1 - g0, g0), __builtin_fmaf(wy, b1 - b0, b0));
}
RFX_DEV float4 rfx_bilerp_half_rgba(uint2 t00, uint2 t10, uint2 t01, uint2 t11, float wx, float wy) {
    const float3 c = rfx_bilerp_half_rgb(t00, t10, t01, t11, wx, wy);
    const float a0 = rfx_half_lerp<1>(wx, t00.y, t10.y), a1 = rfx_half_lerp<1>(wx, t01.y, t11.y);
    return make_float4(c.x, c.y, c.z, __builtin_fmaf(wy, a1 - a0, a0));
}
// gather with a 32-bit BYTE offset from a wave-uniform base (one plane is < 4 GiB: 8K RGBA32F = 0.53 GB): the
// compiler keeps the base in SGPRs and the offset in one VGPR instead of a 64-bit add per lane per tap
template <typename T>
RFX_DEV T rfx_gather(const void *base, unsigned int texel) {
    return *(const T *)((const char *)base + (texel * (unsigned int)sizeof(T)));
}
RFX_DEV float4 rfx_fetch_h4_linear(const TexView &t, const FrameDims &d, float u, float v) {
    int x0, x1, y0, y1;
    float wx, wy;
    rfx_linear_coord(u, d.fW, d.W, x0, x1, wx);
    rfx_linear_coord(v, d.fH, d.H, y0, y1, wy);
    const unsigned int r0 = (unsigned int)__mul24(rfx_local_row(d, t.row0, t.rows, y0), d.W), r1 = (unsigned int)__mul24(rfx_local_row(d, t.row0, t.rows, y1), d.W);
    float4 t00 = rfx_load_half4(rfx_gather<uint2>(t.ptr, r0 + x0)), t10 = rfx_load_half4(rfx_gather<uint2>(t.ptr, r0 + x1));
    float4 t01 = rfx_load_half4(rfx_gather<uint2>(t.ptr, r1 + x0)), t11 = rfx_load_half4(rfx_gather<uint2>(t.ptr, r1 + x1));
    float4 r;
    r.x = rfx_lerp(wy, rfx_lerp(wx, t00.x, t10.x), rfx_lerp(wx, t01.x, t11.x));
    r.y = rfx_lerp(wy, rfx_lerp(wx, t00.y, t10.y), rfx_lerp(wx, t01.y, t11.y));
    r.z = rfx_lerp(wy, rfx_lerp(wx, t00.z, t10.z), rfx_lerp(wx, t01.z, t11.z));
    r.w = rfx_lerp(wy, rfx_lerp(wx, t00.w, t10.w), rfx_lerp(wx, t01.w, t11.w));
    return r;
}

// row of a view that may be the whole frame (WHOLE: row0 == 0 and rows == H — nothing to rebase and nothing to count) or a held band
template <bool WHOLE>
RFX_DEV int rfx_view_row(const FrameDims &d, const TexView &t, int y) {
    return WHOLE ? y : rfx_local_row(d, t.row0, t.rows, y);
}
// the bilinear RGBA16F fetch again, as the LDS-tiled kernels and K2's history taps issue it: three instructions per coordinate
// (rfx_linear_coord_fast) and the sampler's fused lerps on the half texels themselves (rfx_bilerp_half_rgba)
template <bool WHOLE>
RFX_DEV float4 rfx_fetch_h4_linear_fused(const TexView &t, const FrameDims &d, float u, float v) {
    float cx, cy;
    {
#pragma clang fp contract(off)
        cx = u * d.fW;
        cy = v * d.fH;
    }
    const LinearCoord lx = rfx_linear_coord_fast(cx, d.fW - 0.5f), ly = rfx_linear_coord_fast(cy, d.fH - 0.5f);
    const int x1 = min(lx.i0 + 1, d.W - 1), y1 = min(ly.i0 + 1, d.H - 1);
    const unsigned int r0 = (unsigned int)__mul24(rfx_view_row<WHOLE>(d, t, ly.i0), d.W), r1 = (unsigned int)__mul24(rfx_view_row<WHOLE>(d, t, y1), d.W);
    const uint2 t00 = rfx_gather<uint2>(t.ptr, r0 + lx.i0), t10 = rfx_gather<uint2>(t.ptr, r0 + x1);
    const uint2 t01 = rfx_gather<uint2>(t.ptr, r1 + lx.i0), t11 = rfx_gather<uint2>(t.ptr, r1 + x1);
    return rfx_bilerp_half_rgba(t00, t10, t01, t11, lx.w, ly.w);
}

// the same sampler over an RGBA32F texture (FloatType framebuffer copy, TemporalReprojectPass.js:137-142)
RFX_DEV float4 rfx_fetch_f4_linear(const TexView &t, const FrameDims &d, float u, float v) {
    int x0, x1, y0, y1;
    float wx, wy;
    rfx_linear_coord(u, d.fW, d.W, x0, x1, wx);
    rfx_linear_coord(v, d.fH, d.H, y0, y1, wy);
    const unsigned int r0 = (unsigned int)__mul24(rfx_local_row(d, t.row0, t.rows, y0), d.W), r1 = (unsigned int)__mul24(rfx_local_row(d, t.row0, t.rows, y1), d.W);
    const float4 t00 = rfx_gather<float4>(t.ptr, r0 + x0), t10 = rfx_gather<float4>(t.ptr, r0 + x1);
    const float4 t01 = rfx_gather<float4>(t.ptr, r1 + x0), t11 = rfx_gather<float4>(t.ptr, r1 + x1);
    float4 r;
    r.x = rfx_lerp(wy, rfx_lerp(wx, t00.x, t10.x), rfx_lerp(wx, t01.x, t11.x));
    r.y = rfx_lerp(wy, rfx_lerp(wx, t00.y, t10.y), rfx_lerp(wx, t01.y, t11.y));
    r.z = rfx_lerp(wy, rfx_lerp(wx, t00.z, t10.z), rfx_lerp(wx, t01.z, t11.z));
    r.w = rfx_lerp(wy, rfx_lerp(wx, t00.w, t10.w), rfx_lerp(wx, t01.w, t11.w));
    return r;
}
// value of an RGBA16F render-target texel after the store (rounded to half, read back as float)
RFX_DEV float4 rfx_round_half4(float4 v, bool rtz) { return rfx_load_half4(rfx_store_half4(v.x, v.y, v.z, v.w, rtz)); }

// three.js <packing>: perspectiveDepthToViewZ / orthographicDepthToViewZ (reproject.frag:13-19, denoiser_compose_functions.glsl:3-9,
// ssgi_compose.frag:12-18 choose by the PERSPECTIVE_CAMERA define)
RFX_DEV float rfx_depth_to_view_z(float depth, float n, float f, bool perspective) {
    return perspective ? (n * f) / ((f - n) * depth - f) : depth * (n - f) - n;
}

// ---------------------------------------------------------------- float3 helpers
RFX_DEV float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
RFX_DEV float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
RFX_DEV float3 operator*(float3 a, float s) { return make_float3(a.x * s, a.y * s, a.z * s); }
RFX_DEV float3 operator-(float3 a) { return make_float3(-a.x, -a.y, -a.z); }
RFX_DEV float rfx_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RFX_DEV float3 rfx_cross(float3 a, float3 b) { return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// Hardware transcendental set (v_exp_f32 / v_log_f32 / v_sqrt_f32 / v_rsq_f32 / v_rcp_f32: 1 ulp).  The libm-grade
// expf/logf/powf/sqrtf that hipcc emits by default cost 10-100x more VALU issue slots (range reduction, denormal
// scaling, Newton fix-ups) for accuracy the 1e-3 parity budget cannot see; the llvmpipe oracle's own exp/log/pow are
// polynomial approximations of similar accuracy.
RFX_DEV float rfx_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RFX_DEV float rfx_log2(float x) { return __builtin_amdgcn_logf(x); }
RFX_DEV float rfx_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RFX_DEV float rfx_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
RFX_DEV float rfx_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RFX_DEV float3 rfx_normalize(float3 a) { return a * rfx_rsqrt(rfx_dot(a, a)); }
RFX_DEV float rfx_length(float3 a) { return rfx_sqrt(rfx_dot(a, a)); }
RFX_DEV float rfx_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RFX_DEV float3 rfx_mix(float3 x, float3 y, float a) { return make_float3(rfx_mix(x.x, y.x, a), rfx_mix(x.y, y.y, a), rfx_mix(x.z, y.z, a)); }
RFX_DEV float rfx_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// v_min/v_max issued as written.  fminf/fmaxf make hipcc canonicalise (v_max_f32 x, x) every operand it cannot prove
// quiet — one extra VALU per value that comes from memory or LDS.  In the kernel's IEEE mode these return the non-NaN
// operand when the other is a quiet NaN, the same NaN-suppressing min()/max() the llvmpipe oracle has (Appendix C).
RFX_DEV float rfx_min_raw(float a, float b) { float r; r = hostsim_vmin(a, b); return r; }
RFX_DEV float rfx_max_raw(float a, float b) { float r; r = hostsim_vmax(a, b); return r; }
RFX_DEV float rfx_min3_raw(float a, float b, float c) { float r; r = hostsim_vmin(hostsim_vmin(a, b), c); return r; }
RFX_DEV float rfx_max3_raw(float a, float b, float c) { float r; r = hostsim_vmax(hostsim_vmax(a, b), c); return r; }
RFX_DEV float rfx_lum(float3 c) { return 0.2125f * c.x + 0.7154f * c.y + 0.0721f * c.z; }
RFX_DEV float rfx_exp(float x) { return rfx_exp2(x * 1.4426950408889634f); }
RFX_DEV float rfx_log(float x) { return rfx_log2(x) * 0.6931471805599453f; }
RFX_DEV float rfx_pow(float x, float y) { return rfx_exp2(y * rfx_log2(x)); }  // GLSL pow: undefined for x < 0
// sin/cos of an angle in [0, 2pi]: v_sin_f32 / v_cos_f32 take revolutions
RFX_DEV void rfx_sincos(float a, float &s, float &c) {
    const float r = a * 0.15915494309189535f;
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

// column-major mat4 (three.js Matrix4.elements).  M * vec4(x,y,z,w)
RFX_DEV float4 rfx_mat_mul(const float *M, float x, float y, float z, float w) {
    float4 r;
    r.x = ((M[0] * x + M[4] * y) + M[8] * z) + M[12] * w;
    r.y = ((M[1] * x + M[5] * y) + M[9] * z) + M[13] * w;
    r.z = ((M[2] * x + M[6] * y) + M[10] * z) + M[14] * w;
    r.w = ((M[3] * x + M[7] * y) + M[11] * z) + M[15] * w;
    return r;
}
// (vec4(v, w) * M).xyz
RFX_DEV float3 rfx_vec_mul_mat(const float *M, float3 v, float w) {
    float3 r;
    r.x = ((v.x * M[0] + v.y * M[1]) + v.z * M[2]) + w * M[3];
    r.y = ((v.x * M[4] + v.y * M[5]) + v.z * M[6]) + w * M[7];
    r.z = ((v.x * M[8] + v.y * M[9]) + v.z * M[10]) + w * M[11];
    return r;
}

// x / D for a constant D, CORRECTLY ROUNDED — the IEEE quotient the GLSL's `/` yields — in three instructions instead of the
// v_div_scale / v_rcp / v_fma x4 / v_div_fmas / v_div_fixup sequence hipcc emits for `/`: q = RN(x * R) with R = RN(1 / D) is a faithful
// quotient, the residual x - D q is exact in an fma, and RN(q + residual * R) is the correctly rounded quotient (Markstein 1990; holds
// for every x whose quotient neither overflows nor is subnormal, D's significand not all ones).  Checked against `/` on all 2^24 integers
// and 1.5e8 random floats for every divisor used below.
RFX_DEV float rfx_div_const_impl(float x, float d, float r) {
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
}
#define RFX_DIV_CONST(x, D) rfx_div_const_impl((x), (D), 1.0f / (D))
// ... and for a variable divisor KNOWN to be a positive normal number well inside the exponent range (a pdf clamped from below, a sum of
// squares, a luminance that passed a `>` test): v_rcp_f32 (1 ulp) refined by one Newton step is RN(1 / d) except for ~1e-6 of the divisors,
// and the corrected quotient is the IEEE one (0 mismatches in 1e8 random trials with a +-1 ulp starting reciprocal).  One reciprocal serves
// every numerator over the same divisor.  No scaling and no special-case fix-up: NOT for divisors that may be 0, infinite or subnormal.
RFX_DEV float rfx_rcp_rn(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
RFX_DEV float rfx_div_pos(float x, float d) { return rfx_div_const_impl(x, d, rfx_rcp_rn(d)); }

// ---------------------------------------------------------------- G-buffer codec (decode side)
struct Material {
    float3 diffuse;
    float3 normal;
    float roughness, metalness;
    float3 emissive;
};
// unpackNormal / decodeOctWrap, gbuffer_packing.glsl:52-63
RFX_DEV float3 rfx_unpack_normal(uint32_t bits) {
    float fx, fy;
    rfx_unpack_half2(bits, fx, fy);
    fx = fx * 2.0f - 1.0f;
    fy = fy * 2.0f - 1.0f;
    float3 n = make_float3(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    float t = fmaxf(-n.z, 0.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return rfx_normalize(n);
}
// floatToVec4, gbuffer_packing.glsl:151-164 (one byte)
RFX_DEV float rfx_byte_unorm(uint32_t b) { return fmaxf(RFX_DIV_CONST((float)b, 255.0f) - 0.0001f, 0.0f); }
// float2color .r (roughness), gbuffer_packing.glsl:24-34
RFX_DEV float rfx_decode_roughness(uint32_t bits) {
    float value = __uint_as_float(bits);
    float q = RFX_DIV_CONST(value, 257.0f);
    float cr = (value - 257.0f * floorf(q)) / 256.0f; // mod(value, 257) / 256
    return fmaxf(cr - 0.0001f, 0.0f);
}
RFX_DEV float rfx_decode_metalness(uint32_t bits) {
    float value = __uint_as_float(bits);
    float cg = floorf(RFX_DIV_CONST(value, 257.0f * 257.0f)) / 256.0f;
    return fmaxf(cg - 0.0001f, 0.0f);
}
template <bool WITH_EMISSIVE>
RFX_DEV Material rfx_get_material(uint4 g) {
    Material m;
    m.diffuse = make_float3(rfx_byte_unorm(g.x & 0xffu), rfx_byte_unorm((g.x >> 8) & 0xffu), rfx_byte_unorm((g.x >> 16) & 0xffu));
    m.normal = rfx_unpack_normal(g.y);
    m.roughness = rfx_decode_roughness(g.z);
    m.metalness = rfx_decode_metalness(g.z);
    if (WITH_EMISSIVE) {
        float ex = rfx_byte_unorm(g.w & 0xffu), ey = rfx_byte_unorm((g.w >> 8) & 0xffu), ez = rfx_byte_unorm((g.w >> 16) & 0xffu);
        float ea = rfx_byte_unorm(g.w >> 24);
        float sc = rfx_exp2(ea * 255.0f - 128.0f); // decodeRGBE8 :136-141
        m.emissive = make_float3(ex * sc, ey * sc, ez * sc);
    } else {
        m.emissive = make_float3(0.f, 0.f, 0.f);
    }
    return m;
}
// packTwoVec4 / unpackTwoVec4, gbuffer_packing.glsl:65-98
RFX_DEV uint4 rfx_pack_two_vec4(float4 a, float4 b) {
    const float o = 0.0001f;
    return make_uint4(rfx_pack_half2(a.x + o, a.y + o), rfx_pack_half2(a.z + o, a.w + o), rfx_pack_half2(b.x + o, b.y + o),
                      rfx_pack_half2(b.z + o, b.w + o));
}
RFX_DEV float4 rfx_unpack_vec4(uint32_t rg, uint32_t ba) {
    float4 v;
    rfx_unpack_half2(rg, v.x, v.y);
    rfx_unpack_half2(ba, v.z, v.w);
    const float o = 0.0001f;
    v.x -= o; v.y -= o; v.z -= o; v.w -= o;
    return v;
}

// ---------------------------------------------------------------- blue noise (blue_noise.glsl:9-48)
// One pcg4d round seeded by the draw's blueNoiseIndex gives the per-draw toroidal shift; it is
// identical for every pixel, so the host-independent part is hoisted: the kernels receive the
// shift (sx, sy) computed once per launch by rfx_blue_noise_shift().
RFX_DEV void rfx_pcg4d(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w) {
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
}
// RGBA8 -> float as the sampler does it: float(byte) * (1/255)
RFX_DEV uchar4 rfx_blue_noise_texel(const uchar4 *table, int px, int py, int shift_x, int shift_y) {
    int sx = (px + shift_x) & 127, sy = (py + shift_y) & 127; // (pixel + shift) % 128, operands >= 0
    return table[sy * 128 + sx];
}
RFX_DEV float4 rfx_blue_noise(const uchar4 *table, int px, int py, int shift_x, int shift_y) {
    uchar4 t = rfx_blue_noise_texel(table, px, py, shift_x, shift_y);
    const float k = (float)(1.0 / 255.0);
    return make_float4(t.x * k, t.y * k, t.z * k, t.w * k);
}
