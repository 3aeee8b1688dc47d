// K3 — PoissonDenoisePass: 8-tap rotated-Poisson edge-aware blur, one draw per call.
// Replaces `renderer.render` of src/denoise/pass/PoissonDenoisePass.js:146-147 with the
// fragment program src/denoise/shader/poisson_denoise.frag (GBUFFER_TEXTURE variant).
//
// Two implementations of the same arithmetic:
//   * k3_tiled  (radius <= 3, the default): a 64x8-pixel workgroup tile with the apron its taps can address is
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

constexpr int TW = 64, TH = 8;                 // pixels per workgroup tile (64 x 16, 128 x 8, 64 x 7 measured slower: profiles/r05_k3)
constexpr int NT = TW * TH;                    // threads per workgroup
static_assert(TW % 64 == 0 && NT <= 1024, "a tile row is whole wavefronts; at most 16 wavefronts per workgroup");
constexpr int K3_XCD_G = 1;                    // tile rows per XCD group (rfx_xcd_tile); 0 = plain row-major
constexpr int K3_LDS_MAX = 80 * 1024;          // dynamic LDS a tiled launch may ask for (80 KiB: at least two workgroups per CU; the CU has 160 KiB, handed out in
                                               // 1 280-byte granules: <= 53 760 B fit three times, <= 40 960 B four times — profiles/r05_microbench/lds_occupancy.txt)
// The tile is staged with an apron of (Rx, Ry) texels.  The reference rotates the Poisson offsets in UV space
// (`rm * (offset / resolution)`, poisson_denoise.frag:183-189), so on a W x H frame a tap lies within
// r = radius * max(1, W/H) pixels horizontally and radius * max(1, H/W) vertically of the pixel centre — NOT in a circle of
// `radius` pixels.  The apron is exactly what those taps can address (k3_apron): a NEAREST tap reads texel floor(x + 0.5 +- r),
// i.e. floor(r + 0.5) texels out at most; the LINEAR taps of the RGBA16F passes read floor(x +- r) and its +1 neighbour:
// floor(r) + 1.  Pass 0 fetches everything NEAREST and so stages the narrower apron (4K: 74 x 14 texels for 64 x 8 pixels
// instead of the 78 x 16 of rounds 2-4, which added a texel to ceil(r) for every pass).

struct CenterTexel {
    float3 rgb;     // log-space colour accumulator
    float a;        // age passes through
    float lumaPow;  // pow(lum, 1/8)
    float w;        // age weight
    float total;
};

// ---- the tap arithmetic of round 5 (the round-4 forms — natural logarithms, scalar accumulators, separate RGBA16F planes — and their A/B at 4K,
// pass 0 / later pass 0.205 / 0.274 ms -> 0.202 / 0.250: profiles/r05_k3/, code in profiles/r06_cleanup/k3_rejected_variants.patch).
// The log-space colour `log(c + 1)` (poisson_denoise.frag:150,193) is carried as log2(c + 1): the weighted mean of logarithms is linear in
// them, so the base only matters where the luminance of the log colour enters (k3_luma: lum is linear too, lum(ln) = ln2 * lum(log2), and
// pow(x, 1/8) = exp2(log2(x) / 8) takes the factor as an added constant) and at the end (exp2 instead of exp): no `* ln 2` per tap and channel.
RFX_DEV float k3_logc(float x) { return rfx_log2(x); }
RFX_DEV float k3_unlog(float o) { return rfx_exp2(o) - 1.0f; }
RFX_DEV float k3_luma(float3 a) {  // poisson_denoise.frag:28: pow(luminance(a), 1 / 8) of the LOG colour
    return rfx_exp2(__builtin_fmaf(0.125f, rfx_log2(rfx_lum(a)), 0.125f * -0.5287663729448977f));  // + log2(ln 2) / 8
}
RFX_DEV float3 k3_log3(float x, float y, float z) { return make_float3(k3_logc(x + 1.0f), k3_logc(y + 1.0f), k3_logc(z + 1.0f)); }

// applyWeight poisson_denoise.frag:102-124 on an already log-transformed tap.  The bilateral weight arrives as its base-2
// LOGARITHM `l2w`: the reference forms w = exp(-a) (getBasicNeighborWeight :52-78) [* exp(-g) for a specular texture], then
// needs w * exp(-lumaDiff * lumaPhi) and pow(w, 0.1) — i.e. exp2(l2w + l2luma) and exp2(0.1 * l2w): two v_exp_f32 instead of
// exp, log, exp, exp.  l2w = -inf (background tap, :60) gives 0 for both, as w = 0 does in the reference.
constexpr float K3_LOG2E = 1.4426950408889634f;
// `disocclW` = pow(w, 0.1) = exp2(0.1 * l2w).  The tiled kernels form it as exp2(0.1 * l2basic) * exp2(0.1 * l2spec): the first factor is
// shared by the pixel's textures, the second is a per-pixel constant — one v_exp_f32 per tap instead of one per tap and texture.
RFX_DEV void k3_apply_d(CenterTexel &c, float l2w, float disocclW, float3 tl, float tapLuma, float lumaPhiL2) {
    const float lumaDiff = fminf(fabsf(c.lumaPow - tapLuma), 0.5f);
    const float wl = rfx_exp2(l2w - lumaDiff * lumaPhiL2);  // w * lumaFactor
    float w = rfx_mix(wl, disocclW, c.w) * c.w;
    w = (w < 0.0001f) ? 0.0f : w;  // w *= step(0.0001, w)
    c.rgb = c.rgb + tl * w;
    c.total += w;
}
RFX_DEV void k3_apply(CenterTexel &c, float l2w, float3 tl, float tapLuma, float lumaPhiL2) {
    k3_apply_d(c, l2w, rfx_exp2(0.1f * l2w), tl, tapLuma, lumaPhiL2);
}
// ... and for BOTH textures of a pixel at once (TC == 2): lane .x of every pair is accumulator 0, .y accumulator 1.  The same operations
// in the same order as k3_apply_d; what can be a packed fp32 instruction (v_pk_add / mul / fma_f32: 4.5 issue cycles for two results against
// 2 x 2.7, profiles/r03_microbench) is written as one vector operation, the rest (|x|, min, exp2, the threshold select) per lane.
typedef float rfx_f2 __attribute__((ext_vector_type(2)));
struct CenterPair {
    rfx_f2 r, g, b;   // log-space colour accumulators
    rfx_f2 lumaPow, w, omw, total;  // omw = 1 - w (rfx_mix's first weight)
};
RFX_DEV rfx_f2 k3_f2(float a, float b) { rfx_f2 v; v.x = a; v.y = b; return v; }
RFX_DEV void k3_apply_pair(CenterPair &c, rfx_f2 l2w, rfx_f2 disocclW, rfx_f2 tr, rfx_f2 tg, rfx_f2 tb, rfx_f2 tapLuma, float lumaPhiL2) {
    const rfx_f2 d = c.lumaPow - tapLuma;
    const rfx_f2 lumaDiff = k3_f2(fminf(fabsf(d.x), 0.5f), fminf(fabsf(d.y), 0.5f));
    const rfx_f2 e = l2w - lumaDiff * lumaPhiL2;
    const rfx_f2 wl = k3_f2(rfx_exp2(e.x), rfx_exp2(e.y));  // w * lumaFactor
    rfx_f2 w = (wl * c.omw + disocclW * c.w) * c.w;         // rfx_mix(wl, disocclW, c.w) * c.w
    w = k3_f2((w.x < 0.0001f) ? 0.0f : w.x, (w.y < 0.0001f) ? 0.0f : w.y);  // w *= step(0.0001, w)
    c.r += tr * w;
    c.g += tg * w;
    c.b += tb * w;
    c.total += w;
}
// log(c + 1) and pow(lum(.), 1/8) of a pair of colours (k3_log3 / k3_luma lane by lane)
RFX_DEV rfx_f2 k3_logc2(rfx_f2 x) {
    const rfx_f2 x1 = x + 1.0f;
    return k3_f2(rfx_log2(x1.x), rfx_log2(x1.y));
}
RFX_DEV rfx_f2 k3_luma2(rfx_f2 r, rfx_f2 g, rfx_f2 b) {
    const rfx_f2 lum = 0.2125f * r + 0.7154f * g + 0.0721f * b;
    const rfx_f2 l = k3_f2(rfx_log2(lum.x), rfx_log2(lum.y));
    const rfx_f2 e = 0.125f * l + (0.125f * -0.5287663729448977f);
    return k3_f2(rfx_exp2(e.x), rfx_exp2(e.y));
}

// dynamic LDS carve-up (all 16-byte aligned: the row pitch is a multiple of 8 texels), n = PITCH * LH texels:
//   float4 geom[n]          normal.xyz, roughness
//   pass 0 : float4 in[2][n] log(rgb+1), luma^(1/8)     pass >= 1 : uint2 in[2][n] raw RGBA16F
//   float  depth[n]
// The tile is staged WITH the sampler's CLAMP_TO_EDGE built in: a staged position beyond the frame holds a copy of the edge texel.  A tap
// then needs no clamp of its own in LDS space, and the upper texels of a bilinear footprint are always the +1 / +PITCH neighbours of the
// lower one (at the frame edge the reference fetches the edge texel twice, min(i0 + 1, size - 1): lerp(w, a, a) == a exactly, which is
// what the copy gives) — one LDS address per footprint, its four texels at compile-time offsets (two ds_read2_b64).
// PITCH is a template parameter for that reason: LW = 64 + 2 Rx rounded up to 72 / 74 / 76 / 80 / 96 texels (Rx <= 4 / 5 / 6 / 8 / 16).

// WHOLE: every view is the whole frame (a context that owns no row tile): rows need no rebasing and no halo accounting
template <bool IN_TEMPORAL, int TC, int PITCH, bool WHOLE>
RFX_DEV void k3_tiled_body(const K3Args &A, const FrameDims &d) {
    constexpr bool PAIR = TC == 2;  // the pixel's two accumulators as float2 pairs (k3_apply_pair)
    extern __shared__ float4 lds[];
    const int Rx = A.tile.Rx, Ry = A.tile.Ry, LW = A.tile.LW, LH = A.tile.LH;
    const int ntex = PITCH * LH;
    float4 *s_geom = lds;
    float4 *s_in0 = lds + ntex;                                   // pass 0 view
    uint2 *s_inN = reinterpret_cast<uint2 *>(lds + ntex);         // pass >= 1 view
    uint4 *s_in2 = reinterpret_cast<uint4 *>(lds + ntex);         // ... and its interleaved form (two accumulators): both accumulators' texels of a position
    float *s_depth = reinterpret_cast<float *>(lds + ntex) + (IN_TEMPORAL ? 8 : 4) * (size_t)ntex;
    // Pass 0 with two accumulators: the first and last `skip` texels of the staged rectangle — the ends of its first and last row, corners no tap
    // reaches (the taps lie in an ellipse, k3 launcher) — are not held: at 4K that is 4 of 1036 texels and the difference between two and three
    // workgroups per CU (53 872 -> 53 744 B; the CU's 160 KiB are handed out in 1 280-byte granules, profiles/r05_microbench/lds_occupancy.txt).
    // Same arrays under the same indices, moved down by `skip` texels, in the order depth | geometry | inputs: an index inside a shaved corner (only
    // a tap coordinate that is not a finite number produces one: v_med3_f32 sends a NaN to the window's first texel, +inf to its last) falls
    // into the 16-byte pad in front of the depth array, into the tail of the array before its own, or into the pad after the last — never
    // outside the allocation.
    const int skip = (PAIR && IN_TEMPORAL) ? A.tile.skip : 0;
    if (PAIR && IN_TEMPORAL && skip > 0) {
        const int nal = ntex - 2 * skip;  // a multiple of 4 (launcher)
        s_depth = reinterpret_cast<float *>(lds) + 4 - skip;
        s_geom = lds + 1 + nal / 4 - skip;
        s_in0 = lds + 1 + nal / 4 + nal - 2 * skip;
    }
    const rfx_denoise_params &p = A.p;
    if (PAIR && IN_TEMPORAL && skip > 0) {
        // ... and the two pads hold zeros, not what the previous workgroup left in LDS: a frame with a NaN / inf depth or normal stays deterministic
        const int t = threadIdx.y * TW + threadIdx.x, nal = ntex - 2 * skip;
        float *f = reinterpret_cast<float *>(lds);
        if (t < 4) f[t] = 0.0f;
        if (t < 8 * skip) f[4 + nal * 13 + t] = 0.0f;  // behind depth (1) + geometry (4) + interleaved inputs (8 floats per texel)
    }
    const TileXY tile = rfx_xcd_tile<K3_XCD_G>((d.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH);
    if (!tile.valid) return;  // grid padding (uniform per workgroup, before any barrier)
    const int tx0 = tile.bx * TW, ty0 = A.y0 + tile.by * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;
    const float *depthp = (const float *)A.depth.ptr;
    const uint4 *gbp = (const uint4 *)A.gbuffer.ptr;

    // ---- stage the tile + apron, CLAMP_TO_EDGE applied to the staged position
    const float invLW = 1.0f / (float)LW;
    for (int i = tid; i < LW * LH; i += NT) {
        // i / LW without the integer-division sequence: (i + 0.5) / LW is at least 0.5 / LW away from an integer, far above the
        // rounding error of the product for these sizes (i < 2^16, LW < 2^8)
        const int ly = (int)(((float)i + 0.5f) * invLW), lx = i - __mul24(ly, LW);
        const int li = __mul24(ly, PITCH) + lx;
        if (li < skip || li >= ntex - skip) continue;  // a shaved corner (pass 0 only; skip == 0 otherwise)
        const int uy = ty0 - Ry + ly;
        // rows beyond the apron of the last produced row are never addressed (the workgroup may overhang the launch's row range; a
        // row-tiled context does not hold them)
        if (uy > A.y1 - 1 + Ry) continue;
        const int gx = min(max(tx0 - Rx + lx, 0), d.W - 1), gy = min(max(uy, 0), d.H - 1);
        const uint4 g = rfx_gather<uint4>(gbp, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, A.gbuffer, gy), d.W) + gx));
        const float3 n = rfx_unpack_normal(g.y);
        s_geom[li] = make_float4(n.x, n.y, n.z, rfx_decode_roughness(g.z));
        s_depth[li] = rfx_gather<float>(depthp, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, A.depth, gy), d.W) + gx));
        if constexpr (PAIR && IN_TEMPORAL) {
            // pass 0, two accumulators: the log-transformed texels the two ACCUMULATORS read (accumulator i reads texture ti(i), :137-165),
            // interleaved — (x0, x1, y0, y1) (z0, z1, luma0, luma1) — so that a tap's two ds_read_b128 deliver float2 pairs in place
            float3 l[2];
            float lu[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const TexView &src = p.isTextureSpecular[i] ? A.in1 : A.in0;
                const float4 v = rfx_gather<float4>(src.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, src, gy), d.W) + gx));
                l[i] = k3_log3(v.x, v.y, v.z);
                lu[i] = k3_luma(l[i]);
            }
            s_in0[2 * li] = make_float4(l[0].x, l[1].x, l[0].y, l[1].y);
            s_in0[2 * li + 1] = make_float4(l[0].z, l[1].z, lu[0], lu[1]);
        } else if constexpr (PAIR) {
            // later pass, two accumulators: the raw RGBA16F texels the two accumulators read side by side, 16 bytes per position — a bilinear
            // footprint is four aligned ds_read_b128 (7.1 LDS cycles each on this part) instead of four ds_read2_b64 per texture pair (9.0 each)
            uint2 t2[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const TexView &src = p.isTextureSpecular[i] ? A.in1 : A.in0;
                t2[i] = rfx_gather<uint2>(src.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, src, gy), d.W) + gx));
            }
            s_in2[li] = make_uint4(t2[0].x, t2[0].y, t2[1].x, t2[1].y);
        } else {
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const TexView &src = t ? A.in1 : A.in0;
                const unsigned int idx = (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, src, gy), d.W) + gx);
                if constexpr (IN_TEMPORAL) {
                    const float4 v = rfx_gather<float4>(src.ptr, idx);
                    const float3 l = k3_log3(v.x, v.y, v.z);
                    s_in0[t * ntex + li] = make_float4(l.x, l.y, l.z, k3_luma(l));
                } else {
                    s_inN[t * ntex + li] = rfx_gather<uint2>(src.ptr, idx);
                }
            }
        }
    }
    __syncthreads();

    const int x = tx0 + threadIdx.x, y = ty0 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const int cx = threadIdx.x + Rx, cy = threadIdx.y + Ry;  // this pixel inside the staged tile
    const int ci = __mul24(cy, PITCH) + cx;
    const float u = rfx_frag_u(d.uv, x, y), v = rfx_frag_v(d.uv, y);
    const float depth = s_depth[ci];

    // fine 2x2 quad derivatives; a partner beyond the frame edge fetches the edge texel (staged there)
    const int qx0 = ci - (x & 1), qx1 = qx0 + 1;
    const int qy0 = ci - __mul24(y & 1, PITCH), qy1 = qy0 + PITCH;
    {
        const float fw = fabsf(s_depth[qx1] - s_depth[qx0]) + fabsf(s_depth[qy1] - s_depth[qy0]);
        if (depth == 1.0f && fw == 0.0f) {  // discard (:129-132): target keeps its contents
            return;
        }
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

    // frame -> tile coordinates folded into the array bases: a tap's texel (ix, iy) of the frame sits at base[iy * PITCH + ix]
    const int koff = __mul24(Ry - ty0, PITCH) + (Rx - tx0);
    const float4 *g_geom = s_geom + koff;
    const float *g_depth = s_depth + koff;
    // CLAMP_TO_EDGE bounds of the taps, as floats: the FRAME's, intersected with the staged window.  For every finite tap coordinate the frame's
    // bounds alone give the same texel (the apron is the tap footprint, so the clamped texel is staged); the intersection only matters for a
    // coordinate that is not a number (a NaN depth or normal in the dump): v_med3_f32 then returns the lower bound, which is inside the tile —
    // with the frame's bounds it was frame texel (0, 0), an LDS address far outside most tiles.  Same instruction count.
    const int lastx = tx0 + TW - 1 + Rx, lasty = min(ty0 + TH - 1 + Ry, A.y1 - 1 + Ry);  // last staged column / row (frame coordinates; may lie beyond the frame)
    const float xlo = (float)max(tx0 - Rx, 0), ylo = (float)max(ty0 - Ry, 0);
    const float wm1 = (float)min(d.W - 1, lastx), hm1 = (float)min(d.H - 1, lasty);
    // ... and of the lower texel of a bilinear footprint (rfx_linear_coord_fast): its +1 neighbour must be staged too, i.e. the clamped coordinate
    // stays BELOW the last staged column / row (the largest float below it: the apron is exactly the taps' reach, so a coordinate in
    // (last - 1, last) is a real one and keeps its weight; rounds 2-4 clamped at last - 0.5 under an apron one texel wider than the reach)
    const float wmh = fminf(d.fW - 0.5f, __uint_as_float(__float_as_uint((float)lastx) - 1u)), hmh = fminf(d.fH - 0.5f, __uint_as_float(__float_as_uint((float)lasty) - 1u));

    CenterTexel c[TC];
    float l2spec_i[TC];      // log2 of the extra specular factor of accumulator i (0 for a diffuse texture)
    float dspec_i[TC];       // that factor ^ 0.1: exp2(0.1 * l2spec_i) (1 for a diffuse texture)
    const float4 *g_in0[TC];  // the staged input accumulator i reads, rebased like g_geom
    const uint2 *g_inN[TC];
#pragma unroll
    for (int i = 0; i < TC; i++) {  // :137-165
        const bool isSpec = p.isTextureSpecular[i] != 0;
        const int ti = (TC == 2 && isSpec) ? 1 : 0;
        l2spec_i[i] = isSpec ? l2spec : 0.0f;
        dspec_i[i] = isSpec ? rfx_exp2(0.1f * l2spec) : 1.0f;
        g_in0[i] = s_in0 + ti * ntex + koff;
        g_inN[i] = s_inN + ti * ntex + koff;
        float4 t;
        if constexpr (IN_TEMPORAL) {
            const TexView &src = ti ? A.in1 : A.in0;
            t = rfx_gather<float4>(src.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, src, y), d.W) + x));
        } else {  // the sampler's bilinear fetch at the pixel's own vUv
            float fx, fy;
            {
#pragma clang fp contract(off)
                fx = u * d.fW;
                fy = v * d.fH;
            }
            const LinearCoord lx = rfx_linear_coord_fast(fx, xlo, wmh), ly = rfx_linear_coord_fast(fy, ylo, hmh);
            if constexpr (PAIR) {
                const uint4 *q = s_in2 + koff + (__mul24(ly.i0, PITCH) + lx.i0);
                const uint4 p00 = q[0], p10 = q[1], p01 = q[PITCH], p11 = q[PITCH + 1];
                t = i ? rfx_bilerp_half_rgba(make_uint2(p00.z, p00.w), make_uint2(p10.z, p10.w), make_uint2(p01.z, p01.w), make_uint2(p11.z, p11.w), lx.w, ly.w)
                      : rfx_bilerp_half_rgba(make_uint2(p00.x, p00.y), make_uint2(p10.x, p10.y), make_uint2(p01.x, p01.y), make_uint2(p11.x, p11.y), lx.w, ly.w);
            } else {
                const uint2 *q = g_inN[i] + (__mul24(ly.i0, PITCH) + lx.i0);
                t = rfx_bilerp_half_rgba(q[0], q[1], q[PITCH], q[PITCH + 1], lx.w, ly.w);
            }
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

    CenterPair cp2;
    rfx_f2 l2spec2, dspec2;
    if constexpr (PAIR) {
        cp2.r = k3_f2(c[0].rgb.x, c[1].rgb.x); cp2.g = k3_f2(c[0].rgb.y, c[1].rgb.y); cp2.b = k3_f2(c[0].rgb.z, c[1].rgb.z);
        cp2.lumaPow = k3_f2(c[0].lumaPow, c[1].lumaPow);
        cp2.w = k3_f2(c[0].w, c[1].w);
        cp2.omw = 1.0f - cp2.w;
        cp2.total = k3_f2(c[0].total, c[1].total);
        l2spec2 = k3_f2(l2spec_i[0], l2spec_i[1]);
        dspec2 = k3_f2(dspec_i[0], dspec_i[1]);
    }
    const float4 *g_in01 = s_in0 + 2 * koff;  // (PAIR, pass 0: the interleaved pairs, two float4 per texel)

    // no unrolling: occupancy beats ILP here (2 / 4 / 8 taps per iteration: static issue cost -9 .. -19 %, measured +-1 %, profiles/r05_k3)
#pragma clang loop unroll_count(1)
    for (int k = 0; k < 8; k++) {
        const float ox = A.tap_ox[k], oy = A.tap_oy[k];  // POISSON[k] / resolution (:91-92,:189), divided once on the host
        // the tap's texture coordinate, every product and sum rounded on its own as in the GLSL (it addresses NEAREST fetches)
        float fx, fy;
        {
#pragma clang fp contract(off)
            const float nu = u + (m00 * ox + m10 * oy), nv = v + (m01 * ox + m11 * oy);
            fx = nu * d.fW;
            fy = nv * d.fH;
        }
        // nearest CLAMP_TO_EDGE texel of the tap (rfx_nearest_idx without its NaN / 2^31 guard: the coordinates are finite and O(size) here)
        const int ni = __mul24((int)__builtin_amdgcn_fmed3f(fy, ylo, hm1), PITCH) + (int)__builtin_amdgcn_fmed3f(fx, xlo, wm1);
        // getBasicNeighborWeight :52-78
        const float nd = g_depth[ni];
        const float4 ng = g_geom[ni];
        const float normalDiff = 1.0f - fmaxf(rfx_dot(normal, make_float3(ng.x, ng.y, ng.z)), 0.0f);
        const float depthDiff = 10000.0f * fabsf(depth - nd);
        const float roughDiff = fabsf(roughness - ng.w);
        float l2basic = (-normalDiff * p.normalPhi - depthDiff * p.depthPhi - roughDiff * p.roughnessPhi) * K3_LOG2E;
        l2basic = (nd != 1.0f) ? l2basic : -__builtin_inff();
        const float dbasic = rfx_exp2(0.1f * l2basic);  // pow(basic weight, 0.1), shared by the textures
        if constexpr (PAIR) {
            rfx_f2 tr, tg, tb, tluma;
            if constexpr (IN_TEMPORAL) {
                const float4 a = g_in01[2 * ni], b = g_in01[2 * ni + 1];
                tr = k3_f2(a.x, a.y); tg = k3_f2(a.z, a.w); tb = k3_f2(b.x, b.y); tluma = k3_f2(b.z, b.w);
            } else {
                const LinearCoord lx = rfx_linear_coord_fast(fx, xlo, wmh), ly = rfx_linear_coord_fast(fy, ylo, hmh);
                const int li = __mul24(ly.i0, PITCH) + lx.i0;
                // the sampler's bilinear blend (rfx_bilerp_half_rgb): x-lerps on the half texels per texture, the y-lerp on the pairs
                const uint4 *q = s_in2 + koff + li;
                const uint4 p00 = q[0], p10 = q[1], p01 = q[PITCH], p11 = q[PITCH + 1];
                const uint2 a00 = make_uint2(p00.x, p00.y), a10 = make_uint2(p10.x, p10.y), a01 = make_uint2(p01.x, p01.y), a11 = make_uint2(p11.x, p11.y);
                const uint2 b00 = make_uint2(p00.z, p00.w), b10 = make_uint2(p10.z, p10.w), b01 = make_uint2(p01.z, p01.w), b11 = make_uint2(p11.z, p11.w);
                const rfx_f2 r0 = k3_f2(rfx_half_lerp<0>(lx.w, a00.x, a10.x), rfx_half_lerp<0>(lx.w, b00.x, b10.x));
                const rfx_f2 g0 = k3_f2(rfx_half_lerp<1>(lx.w, a00.x, a10.x), rfx_half_lerp<1>(lx.w, b00.x, b10.x));
                const rfx_f2 b0 = k3_f2(rfx_half_lerp<0>(lx.w, a00.y, a10.y), rfx_half_lerp<0>(lx.w, b00.y, b10.y));
                const rfx_f2 r1 = k3_f2(rfx_half_lerp<0>(lx.w, a01.x, a11.x), rfx_half_lerp<0>(lx.w, b01.x, b11.x));
                const rfx_f2 g1 = k3_f2(rfx_half_lerp<1>(lx.w, a01.x, a11.x), rfx_half_lerp<1>(lx.w, b01.x, b11.x));
                const rfx_f2 b1 = k3_f2(rfx_half_lerp<0>(lx.w, a01.y, a11.y), rfx_half_lerp<0>(lx.w, b01.y, b11.y));
                const rfx_f2 wy = k3_f2(ly.w, ly.w);
                tr = k3_logc2(__builtin_elementwise_fma(wy, r1 - r0, r0));
                tg = k3_logc2(__builtin_elementwise_fma(wy, g1 - g0, g0));
                tb = k3_logc2(__builtin_elementwise_fma(wy, b1 - b0, b0));
                tluma = k3_luma2(tr, tg, tb);
            }
            k3_apply_pair(cp2, l2basic + l2spec2, dbasic * dspec2, tr, tg, tb, tluma, lumaPhiL2);
        } else if constexpr (IN_TEMPORAL) {
#pragma unroll
            for (int i = 0; i < TC; i++) {
                const float4 tl = g_in0[i][ni];
                k3_apply_d(c[i], l2basic + l2spec_i[i], dbasic * dspec_i[i], make_float3(tl.x, tl.y, tl.z), tl.w, lumaPhiL2);
            }
        } else {
            const LinearCoord lx = rfx_linear_coord_fast(fx, xlo, wmh), ly = rfx_linear_coord_fast(fy, ylo, hmh);
            const int li = __mul24(ly.i0, PITCH) + lx.i0;
#pragma unroll
            for (int i = 0; i < TC; i++) {
                const uint2 *q = g_inN[i] + li;
                const float3 t = rfx_bilerp_half_rgb(q[0], q[1], q[PITCH], q[PITCH + 1], lx.w, ly.w);
                const float3 tl = k3_log3(t.x, t.y, t.z);
                k3_apply_d(c[i], l2basic + l2spec_i[i], dbasic * dspec_i[i], tl, k3_luma(tl), lumaPhiL2);
            }
        }
    }
    if constexpr (PAIR) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            c[i].rgb = make_float3(cp2.r[i], cp2.g[i], cp2.b[i]);
            c[i].total = cp2.total[i];
        }
    }

    const size_t oi = (size_t)(unsigned int)(__mul24(WHOLE ? y : rfx_local_row(d, A.out0.row0, A.out0.rows, y), d.W) + x);
#pragma unroll
    for (int i = 0; i < TC; i++) {  // outputTexel :94-100
        const float inv = rfx_rcp(c[i].total);
        float3 o = make_float3(c[i].rgb.x * inv, c[i].rgb.y * inv, c[i].rgb.z * inv);
        o = make_float3(k3_unlog(o.x), k3_unlog(o.y), k3_unlog(o.z));
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
        o = make_float3(k3_unlog(o.x), k3_unlog(o.y), k3_unlog(o.z));
        ((uint2 *)(i ? A.out1.ptr : A.out0.ptr))[oi] = rfx_store_half4(o.x, o.y, o.z, c[i].a, p.halfStoreRTZ != 0);
    }
}

template <bool IN_TEMPORAL, int TC, int PITCH, bool WHOLE>
__global__ __launch_bounds__(NT) void k3_tiled(K3Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k3_tiled_body<IN_TEMPORAL, TC, PITCH, WHOLE>(A, d);
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
    // the apron the taps can address (file header); SLACK covers the rounding of the tap coordinate itself (one ulp of vUv * size: 1e-3 pixel
    // on a 16K frame) — a tap offset that close to a half-integer (nearest) or an integer (linear) boundary stages one texel more
    const float SLACK = 4e-3f;
    const auto k3_apron = [&](float r) {  // (never below 1: the 2x2-quad partners and the centre's own LINEAR fetch)
        const int a = temporal ? (int)floorf(r + 0.5f + SLACK) : (int)floorf(r + SLACK) + 1;
        return a < 1 ? 1 : a;
    };
    A.tile.Rx = k3_apron(rx);
    A.tile.Ry = k3_apron(ry);
    A.tile.LW = TW + 2 * A.tile.Rx;
    A.tile.LH = TH + 2 * A.tile.Ry;
    A.tile.skip = 0;
    // LDS row pitch: a compile-time constant of the tiled kernels (the footprint's second row is an immediate offset; padding it by 1 / 2 / 4 texels
    // moves neither the time nor the bank-conflict share: the conflicts are collisions of per-pixel-rotated taps, profiles/r04_k3)
    const int pitch = A.tile.LW <= TW + 8 ? TW + 8 : A.tile.LW <= TW + 10 ? TW + 10 : A.tile.LW <= TW + 12 ? TW + 12 : A.tile.LW <= TW + 16 ? TW + 16 : A.tile.LW <= TW + 32 ? TW + 32 : 0;
    size_t lds = (size_t)pitch * A.tile.LH * (16 + 4 + 2 * (temporal ? 16 : 8));
    if (temporal && A.p.textureCount == 2 && pitch == A.tile.LW && pitch != 0) {
        // The corners of the staged rectangle no tap reaches: a tap's offset from its pixel, in pixels, lies in the ellipse (dx / rx)^2 + (dy / ry)^2 <= 1
        // (the rotation acts in UV space, flatness <= 1, |POISSON[k]| <= 1), and the rectangle's first row is addressed only by the tile's first
        // row of pixels with dy in [-Ry - 0.5, -Ry + 0.5): there |dx| <= rx * sqrt(1 - ((Ry - 0.5) / ry)^2), i.e. a NEAREST tap reaches at most X texels
        // sideways and the first Rx - X texels of that row (and, mirrored, the last Rx - X of the last row) are never read.  Pass 0 only (the later
        // passes' LINEAR footprints reach further and their LDS size is nowhere near a granule boundary).
        const float t = ((float)A.tile.Ry - 0.5f - SLACK) / ry;
        const int X = (int)floorf(0.5f + rx * sqrtf(fmaxf(0.0f, 1.0f - t * t)) + SLACK);
        int skip = A.tile.Rx - X;
        if (skip > 4) skip = 4;  // (the pad in front of the depth array holds four floats)
        const int ntex = pitch * A.tile.LH;
        while (skip > 0 && ((ntex - 2 * skip) & 3) != 0) skip--;  // the float4 arrays behind the depth array stay 16-byte aligned
        if (skip > 0) {
            A.tile.skip = skip;
            lds = 16 + (size_t)(ntex - 2 * skip) * (4 + 16 + 32) + (size_t)skip * 32;  // pad | depth | geometry | interleaved inputs | pad
        }
    }
    // at least two workgroups per CU (160 KiB LDS) keep the staging of one tile under the arithmetic of another (4K: three of either pass kind)
    const bool tiled = A.p.radius >= 0.0f && pitch != 0 && lds <= (size_t)K3_LDS_MAX;
    // every view the whole frame (a context that owns no row tile): the kernels skip row rebasing and halo accounting
    const auto whole_view = [&](const void *ptr, int row0, int rows) { return ptr == nullptr || (row0 == 0 && rows == A.dims.H); };
    const bool whole = whole_view(A.depth.ptr, A.depth.row0, A.depth.rows) && whole_view(A.gbuffer.ptr, A.gbuffer.row0, A.gbuffer.rows) &&
                       whole_view(A.in0.ptr, A.in0.row0, A.in0.rows) && whole_view(A.in1.ptr, A.in1.row0, A.in1.rows) &&
                       whole_view(A.out0.ptr, A.out0.row0, A.out0.rows) && whole_view(A.out1.ptr, A.out1.row0, A.out1.rows);
    if (tiled) {
        dim3 block(TW, TH), grid(rfx_xcd_grid(K3_XCD_G, (A.dims.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH));
        // the attribute is per device (a process may hold contexts on several): remembered per device ordinal
#define K3_TILED(T, C, P, WH)                                                                                                \
    do {                                                                                                                     \
        static bool attr_set[64] = {false};                                                                                  \
        int dev = 0;                                                                                                         \
        hipGetDevice(&dev);                                                                                                  \
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                        \
            hipFuncSetAttribute((const void *)k3_tiled<T, C, P, WH>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_LDS_MAX); \
            if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                  \
        }                                                                                                                    \
        hipLaunchKernelGGL((k3_tiled<T, C, P, WH>), grid, block, lds, stream, A);                                            \
    } while (0)
#define K3_TILED_W(T, C, P) do { if (whole) K3_TILED(T, C, P, true); else K3_TILED(T, C, P, false); } while (0)
#define K3_TILED_P(T, C)                    \
    do {                                    \
        if (pitch == TW + 8) K3_TILED_W(T, C, TW + 8); \
        else if (pitch == TW + 10) K3_TILED_W(T, C, TW + 10); \
        else if (pitch == TW + 12) K3_TILED_W(T, C, TW + 12); \
        else if (pitch == TW + 16) K3_TILED_W(T, C, TW + 16); \
        else K3_TILED_W(T, C, TW + 32);          \
    } while (0)
        if (A.p.textureCount == 2) {
            if (temporal) K3_TILED_P(true, 2);
            else K3_TILED_P(false, 2);
        } else {
            if (temporal) K3_TILED_P(true, 1);
            else K3_TILED_P(false, 1);
        }
#undef K3_TILED_P
#undef K3_TILED_W
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
