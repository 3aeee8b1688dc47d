// K2 — TemporalReprojectPass: velocity / hit-point reprojection, disocclusion confidence,
// neighbourhood clamp and age-driven accumulation.  Replaces `renderer.render` of
// src/temporal-reproject/TemporalReprojectPass.js:192-193 with the fragment program
// src/temporal-reproject/shader/temporal_reproject.frag (+ reproject.frag), PERSPECTIVE_CAMERA.
//
// A 64x8-pixel workgroup tile (eight waves) with a 2-texel apron is staged through LDS once: the packed K1 output
// is unpacked (8 halfs -> two float4) and the velocity texel is decoded (normal + depth) ONE time per
// texel, so the (2r+1)^2 neighbourhood AABB of both textures (up to 50 taps per pixel) and the 2x2-quad
// derivatives read LDS instead of re-fetching and re-unpacking global texels.  The history taps
// (5 bilinear taps x 2 textures at the reprojected uv) and the validation fetch stay global gathers:
// their position is data dependent.  A tap's y-lerp and the weighted sum of the five taps are written on (r, g) / (b, a) float2 PAIRS (k2_f2:
// v_pk_add / mul / fma_f32, 4.5 issue cycles for two results against 2 x 2.7; the x-lerps read halfs in place and stay scalar): every lane
// operation is the IEEE one of the scalar form and this file is compiled without contraction — the same bits (sha1 of both targets at 4K),
// K2 0.324 -> 0.314 ms (profiles/r06_k2/; the bicubic weights as an (u, v) pair and the log-space colours as (r, g) + b on top of it: 80 VGPRs, no
// faster).  (The AABB as column reductions exchanged between lanes with DPP wave shifts — 30 LDS reads per pixel instead
// of 50, bit-identical — was built and measured in round 5: no faster, 13 % more VALU instructions; profiles/r05_k2/.)
#include "rfx_device.h"
#include "rfx_kernels.h"

namespace {

constexpr int K2_XCD_G = 0;  // tile rows per XCD group (rfx_xcd_tile); 0 = plain row-major (measurements: profiles/HISTORY.md)
// 8 tile rows (eight waves per workgroup) stage 1.6 texels per pixel instead of 2.1 and keep 6 waves per SIMD resident (46 KB of LDS per workgroup)
// against 5 with 4 rows (measurements: profiles/r04_k2)
constexpr int TW = 64, TH = 8, AP = 2;             // tile, apron (neighbourhood radius <= 2)
constexpr int LW = TW + 2 * AP, LH = TH + 2 * AP;  // 68 x 12 staged texels
constexpr int NT = TW * TH;

struct VND {
    float vx, vy, depth;
    float3 normal;
};
// getVelocityNormalDepth reproject.frag:97-105
RFX_DEV VND k2_vnd(uint4 t) {
    VND r;
    r.vx = __uint_as_float(t.x);
    r.vy = __uint_as_float(t.y);
    r.normal = rfx_unpack_normal(t.z);
    r.depth = __uint_as_float(t.w);
    return r;
}
// screenSpaceToWorldSpace reproject.frag:21-28
RFX_DEV float3 k2_ss_to_ws(float u, float v, float depth, const float *matWorld, const float *projInv) {
    const float4 clip = rfx_mat_mul(projInv, (u - 0.5f) * 2.0f, (v - 0.5f) * 2.0f, (depth - 0.5f) * 2.0f, 1.0f);
    const float iw = rfx_rcp(clip.w);
    const float4 w = rfx_mat_mul(matWorld, clip.x * iw, clip.y * iw, clip.z * iw, clip.w * iw);
    return make_float3(w.x, w.y, w.z);
}
// validateReprojectedUV reproject.frag:130-167 (the angleMix / lastViewAngle computation is dead code)
template <bool WHOLE>
RFX_DEV float k2_validate(const K2Args &A, const FrameDims &d, float ru, float rv, float3 worldPos, float3 worldNormal, float distFactor) {
    if (ru > 1.0f || ru < 0.0f || rv > 1.0f || rv < 0.0f) return 0.0f;
    // NB: the CURRENT velocity texture, NEAREST
    const int vx = rfx_nearest_idx(ru, d.fW, d.W), vy = rfx_nearest_idx(rv, d.fH, d.H);
    const VND last = k2_vnd(rfx_gather<uint4>(A.velocity.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, A.velocity, vy), d.W) + vx)));
    const float3 lastWorldPos = k2_ss_to_ws(ru, rv, last.depth, A.p.prevCamera.matrixWorld, A.p.prevCamera.projectionMatrixInverse);
    const float3 dp = worldPos - lastWorldPos;
    float disoccl = 0.0f;
    disoccl += rfx_length(dp) * 0.1f * distFactor;                                 // worldDistanceDisocclusionCheck (/ 10.)
    disoccl += fabsf(rfx_dot(dp, worldNormal)) * 0.05f * distFactor;                // planeDistanceDisocclusionCheck (/ 20.)
    disoccl += fminf(1.0f - rfx_dot(worldNormal, last.normal), 1.0f) * distFactor;  // normalDisocclusionCheck (/ 1.)
    const float conf = fmaxf(1.0f - fminf(disoccl, 1.0f), 0.0f);
    return rfx_pow(conf, A.p.confidencePower);
}

typedef float k2_f2 __attribute__((ext_vector_type(2)));
RFX_DEV k2_f2 k2_mk2(float a, float b) { k2_f2 r; r.x = a; r.y = b; return r; }
struct k2_rgba { k2_f2 rg, ba; };

// BiCubicCatmullRom5Tap reproject.frag:212-255 — five hardware-bilinear taps of the RGBA16F (or RGBA32F) history
// (RGBA16F: the sampler's lerps fused, on the half texels themselves; what the oracle GL does — rfx_fetch_h4_linear_fused's arithmetic with the
// y-lerp of the four channels as two packed fmas)
template <bool HIST_F32, bool WHOLE>
RFX_DEV k2_rgba k2_history_tap(const TexView &t, const FrameDims &d, float u, float v) {
    k2_rgba r;
    if constexpr (HIST_F32) {
        const float4 c = rfx_fetch_f4_linear(t, d, u, v);
        r.rg = k2_mk2(c.x, c.y);
        r.ba = k2_mk2(c.z, c.w);
    } else {
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
        const float wx = lx.w;
        const k2_f2 rg0 = k2_mk2(rfx_half_lerp<0>(wx, t00.x, t10.x), rfx_half_lerp<1>(wx, t00.x, t10.x)), ba0 = k2_mk2(rfx_half_lerp<0>(wx, t00.y, t10.y), rfx_half_lerp<1>(wx, t00.y, t10.y));
        const k2_f2 rg1 = k2_mk2(rfx_half_lerp<0>(wx, t01.x, t11.x), rfx_half_lerp<1>(wx, t01.x, t11.x)), ba1 = k2_mk2(rfx_half_lerp<0>(wx, t01.y, t11.y), rfx_half_lerp<1>(wx, t01.y, t11.y));
        const k2_f2 wy = k2_mk2(ly.w, ly.w);
        r.rg = __builtin_elementwise_fma(wy, rg1 - rg0, rg0);
        r.ba = __builtin_elementwise_fma(wy, ba1 - ba0, ba0);
    }
    return r;
}
template <bool HIST_F32, bool WHOLE>
RFX_DEV float4 k2_bicubic(const K2Args &A, const FrameDims &d, const TexView &tex, float pu, float pv) {
    float Wa[2], Wb[2], Wc[2], S0[2], S1[2], S2[2];
    // The three quotients of the GLSL are the IEEE ones here: UV = P / invTexSize decides the texel (tc) and the weights (f) — at 8K one
    // ulp of a v_rcp-based quotient is 5e-4 texel, which an age channel that differs by ~2 between neighbouring texels turns into 1e-3
    // (measured: round 2's 8K frame-2 K2 population; the rgb channels never showed it).  invTexSize is a uniform, so the exact quotient
    // costs three instructions with the host's RN(1 / invTexSize) (rfx_div_const_impl); w2 / (w1 + w2) and 1 / sum take the refined
    // reciprocal (both divisors are ~1).
    const float its[2] = {A.invW, A.invH}, rits[2] = {A.rcpInvW, A.rcpInvH}, P[2] = {pu, pv};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float UV = rfx_div_const_impl(P[k], its[k], rits[k]);
        const float tc = floorf(UV - 0.5f) + 0.5f;
        const float f = UV - tc, f2 = f * f, f3 = f2 * f;
        const float w0 = f2 - 0.5f * (f3 + f);
        const float w1 = 1.5f * f3 - 2.5f * f2 + 1.0f;
        const float w3 = 0.5f * (f3 - f2);
        const float w2 = 1.0f - w0 - w1 - w3;
        Wa[k] = w0;
        Wb[k] = w1 + w2;
        Wc[k] = w3;
        S0[k] = (tc - 1.0f) * its[k];
        S1[k] = (tc + rfx_div_pos(w2, Wb[k])) * its[k];  // Wb = w1 + w2 in [1, 1.125]
        S2[k] = (tc + 2.0f) * its[k];
    }
    const float sw0 = Wb[0] * Wa[1], sw1 = Wa[0] * Wb[1], sw2 = Wb[0] * Wb[1], sw3 = Wc[0] * Wb[1], sw4 = Wb[0] * Wc[1];
    // A compiler barrier between the bilinear taps: left alone, the scheduler puts all 20 texels of the five taps in flight at once and the
    // kernel needs ~20 more VGPRs (one wave per SIMD fewer); fenced after every second tap it fits its occupancy.  Same texels.
    // (measurements and the ablation of the kernel's parts: profiles/HISTORY.md)
    const k2_rgba Ct = k2_history_tap<HIST_F32, WHOLE>(tex, d, S1[0], S0[1]);
    const k2_rgba Cl = k2_history_tap<HIST_F32, WHOLE>(tex, d, S0[0], S1[1]);
    asm volatile("" ::: "memory");
    const k2_rgba Cc = k2_history_tap<HIST_F32, WHOLE>(tex, d, S1[0], S1[1]);
    const k2_rgba Cr = k2_history_tap<HIST_F32, WHOLE>(tex, d, S2[0], S1[1]);
    asm volatile("" ::: "memory");
    const k2_rgba Cb = k2_history_tap<HIST_F32, WHOLE>(tex, d, S1[0], S2[1]);
    const float wm = rfx_rcp_rn((((sw0 + sw1) + sw2) + sw3) + sw4);  // 1. / sum, sum ~ 1
    const k2_f2 rg = ((((Ct.rg * sw0 + Cl.rg * sw1) + Cc.rg * sw2) + Cr.rg * sw3) + Cb.rg * sw4) * wm;
    const k2_f2 ba = ((((Ct.ba * sw0 + Cl.ba * sw1) + Cc.ba * sw2) + Cr.ba * sw3) + Cb.ba * sw4) * wm;
    return make_float4(fmaxf(rg.x, 0.0f), fmaxf(rg.y, 0.0f), fmaxf(ba.x, 0.0f), fmaxf(ba.y, 0.0f));
}

template <bool LOGT>
RFX_DEV float3 k2_to_log(float3 c) {  // transformColor reproject.frag:42
    return LOGT ? make_float3(rfx_log(c.x + 1.0f), rfx_log(c.y + 1.0f), rfx_log(c.z + 1.0f)) : c;
}
template <bool LOGT>
RFX_DEV float3 k2_from_log(float3 c) {  // undoColorTransform :43
    return LOGT ? make_float3(rfx_exp(c.x) - 1.0f, rfx_exp(c.y) - 1.0f, rfx_exp(c.z) - 1.0f) : c;
}

// input texel `idx` of the packed K1 output (DIFFUSE_SPECULAR) or the raw texel
template <int INPUT_TYPE>
RFX_DEV float4 k2_unpack(uint4 t, int idx) {
    if (INPUT_TYPE == 0) return idx ? rfx_unpack_vec4(t.z, t.w) : rfx_unpack_vec4(t.x, t.y);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}

RFX_DEV float4 k2_mask_unsampled(float4 t) {
    const float qnan = __builtin_nanf("");
    return (t.x >= 0.0f) ? t : make_float4(qnan, qnan, qnan, t.w);
}

struct Tile {
    float4 tex[2][LH * LW];  // unpacked input texels: [0] = diffuse (or the raw single texture), [1] = specular
    float4 vn[LH * LW];      // velocity texel: world normal.xyz, depth
    float2 vel[LH * LW];     // velocity.xy
};

template <int INPUT_TYPE, int TC, bool LOGT, bool HIST_F32, bool WHOLE>
RFX_DEV void k2_body(const K2Args &A, const FrameDims &d) {
    __shared__ Tile s;
    const rfx_temporal_params &p = A.p;
    const TileXY tile = rfx_xcd_tile<K2_XCD_G>((d.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH);
    if (!tile.valid) return;  // grid padding (uniform per workgroup, before the barrier)
    const int tx0 = tile.bx * TW, ty0 = A.y0 + tile.by * TH;
    const int tid = threadIdx.y * TW + threadIdx.x;

    // ---- stage tile + apron (out-of-frame texels are never addressed: CLAMP_TO_EDGE is applied first)
    for (int i = tid; i < LW * LH; i += NT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = tx0 - AP + lx, gy = ty0 - AP + ly;
        if (gx < 0 || gx >= d.W || gy < 0 || gy >= d.H || gy > A.y1 - 1 + AP) continue;
        // the input texel this full-resolution position samples: itself, or — K1 drawn at resolutionScale < 1 — the NEAREST texel of the
        // smaller target at this pixel's vUv (whole-frame contexts only; the target is stored at the start of the slot, pitch in_w)
        uint4 t;
        if (A.in_w != d.W || A.in_h != d.H) {
            const int ix = rfx_nearest_idx(rfx_frag_u(d.uv, gx, gy), (float)A.in_w, A.in_w), iy = rfx_nearest_idx(rfx_frag_v(d.uv, gy), (float)A.in_h, A.in_h);
            t = ((const uint4 *)A.ssgi.ptr)[(size_t)iy * A.in_w + ix];
        } else {
            t = rfx_gather<uint4>(A.ssgi.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, A.ssgi, gy), d.W) + gx));
        }
        // a texel that was not sampled (`!(t.r >= 0.)`) takes no part in any neighbourhood AABB (reproject.frag:66) and its
        // colour is never read as a centre texel either: stage its rgb as quiet NaNs, which v_min/v_max skip, so the
        // 25-tap loops below need no per-tap test.  .a (roughness / ray length) is kept.
        s.tex[0][i] = k2_mask_unsampled(k2_unpack<INPUT_TYPE>(t, 0));
        if (INPUT_TYPE == 0) s.tex[1][i] = k2_mask_unsampled(k2_unpack<INPUT_TYPE>(t, 1));
        const VND vd = k2_vnd(rfx_gather<uint4>(A.velocity.ptr, (unsigned int)(__mul24(rfx_view_row<WHOLE>(d, A.velocity, gy), d.W) + gx)));
        s.vn[i] = make_float4(vd.normal.x, vd.normal.y, vd.normal.z, vd.depth);
        s.vel[i] = make_float2(vd.vx, vd.vy);
    }
    __syncthreads();

    const int x = tx0 + threadIdx.x, y = ty0 + threadIdx.y;
    if (x >= d.W || y >= A.y1) return;
    const int cx = threadIdx.x + AP, cy = threadIdx.y + AP, ci = cy * LW + cx;
    const float u = rfx_frag_u(d.uv, x, y), v = rfx_frag_v(d.uv, y);

    const float4 cvn = s.vn[ci];
    const float2 cvel = s.vel[ci];
    const float depth = cvn.w;
    // 2x2 quad partners for fwidth(depth) / fwidth(worldNormal)
    const int qx0 = cy * LW + min(x & ~1, d.W - 1) - tx0 + AP, qx1 = cy * LW + min(x | 1, d.W - 1) - tx0 + AP;
    const int qy0 = (min(y & ~1, d.H - 1) - ty0 + AP) * LW + cx, qy1 = (min(y | 1, d.H - 1) - ty0 + AP) * LW + cx;
    const float4 xa = s.vn[qx0], xb = s.vn[qx1], ya = s.vn[qy0], yb = s.vn[qy1];
    if (INPUT_TYPE != 1) {  // temporal_reproject.frag:188-193
        const float fw = fabsf(xb.w - xa.w) + fabsf(yb.w - ya.w);
        if (depth == 1.0f && fw == 0.0f) return;  // discard
    }
    const float3 fwn = make_float3(fabsf(xb.x - xa.x) + fabsf(yb.x - ya.x), fabsf(xb.y - xa.y) + fabsf(yb.y - ya.y), fabsf(xb.z - xa.z) + fabsf(yb.z - ya.z));
    const float curvature = rfx_length(fwn);  // getCurvature reproject.frag:265-269

    // getTexels + preprocessInput :124-145 happen per texture below (the centre texel is re-read from LDS there instead of being held in
    // registers across the disocclusion tests); only the two scalars getRoughnessRayLength needs are taken here
    const float3 worldNormal = make_float3(cvn.x, cvn.y, cvn.z);
    const float3 worldPos = k2_ss_to_ws(u, v, depth, p.camera.matrixWorld, p.camera.projectionMatrixInverse);
    float rayLength = 0.0f, roughness = 1.0f;  // getRoughnessRayLength :167-176
    if (INPUT_TYPE == 0) {
        rayLength = s.tex[TC - 1][ci].w;
        roughness = rfx_clamp(s.tex[0][ci].w, 0.0f, 1.0f);
    } else if (INPUT_TYPE == 2) {
        float rl, ro;
        rfx_unpack_half2(__float_as_uint(s.tex[0][ci].w), rl, ro);
        rayLength = rl;
        roughness = rfx_clamp(ro, 0.0f, 1.0f);
    }
    const float n_ = p.camera.near_, f_ = p.camera.far_;
    const float viewZ = p.camera.isPerspective ? fabsf((n_ * f_) * rfx_rcp((f_ - n_) * depth - f_)) : fabsf(depth * (n_ - f_) - n_);  // getViewZ reproject.frag:13-19
    const float distFactor = 1.0f + rfx_rcp(viewZ + 1.0f);

    // computeReprojectedUv :155-165
    float3 rd, rs;
    rd.x = u - cvel.x;
    rd.y = v - cvel.y;
    rd.z = k2_validate<WHOLE>(A, d, rd.x, rd.y, worldPos, worldNormal, distFactor);
    rs = rd;
    if (INPUT_TYPE != 1) {
        if (!(curvature > 0.05f || rayLength < 0.01f)) {  // reprojectHitPoint reproject.frag:169-193
            const float3 camPos = make_float3(p.camera.position[0], p.camera.position[1], p.camera.position[2]);
            const float3 cameraRay = rfx_normalize(worldPos - camPos);
            const float3 hp = camPos + cameraRay * rayLength;
            const float4 r = rfx_mat_mul(A.prevPV, hp.x, hp.y, hp.z, 1.0f);
            // IEEE divisions: this uv addresses a NEAREST fetch (the validation texel)
            const float hu = (r.x / r.w) * 0.5f + 0.5f, hv = (r.y / r.w) * 0.5f + 0.5f;
            const float conf = k2_validate<WHOLE>(A, d, hu, hv, worldPos, worldNormal, distFactor);
            if (hu != -1.0f) rs = make_float3(hu, hv, conf);  // :161-163 falls back to the diffuse triple
        }
    }
    const float moveFactor = fminf((cvel.x * cvel.x + cvel.y * cvel.y) * 10000.0f, 1.0f);
    const size_t oi = (size_t)(unsigned int)(__mul24(WHOLE ? y : rfx_local_row(d, A.out0.row0, A.out0.rows, y), d.W) + x);

    // neighbourhood columns with CLAMP_TO_EDGE, as LDS offsets
    int nxo[5];
#pragma unroll
    for (int o = -2; o <= 2; o++) nxo[o + 2] = min(max(x + o, 0), d.W - 1) - tx0 + AP;

#pragma unroll
    for (int i = 0; i < TC; i++) {
        const bool spec = p.reprojectSpecular[i] != 0;
        const float3 uvc = spec ? rs : rd;
        // reproject() :83-122.  The 5 bilinear history fetches of THIS texture (sampleReprojectedTexture, reproject.frag:257-263)
        // are issued here, ahead of the LDS neighbourhood reduction that hides their latency; fetching both textures' taps
        // up front held 80 VGPRs of texels and capped the kernel at one workgroup per CU.
        const float4 acc = k2_bicubic<HIST_F32, WHOLE>(A, d, i ? A.hist1 : A.hist0, uvc.x, uvc.y);
        float3 accrgb = k2_to_log<LOGT>(make_float3(acc.x, acc.y, acc.z));
        float acca = acc.w;
        const float4 inp = s.tex[i][ci];  // preprocessInput :124-128 (an unsampled texel was staged with NaN rgb: !(NaN >= 0))
        const bool sampled_i = inp.x >= 0.0f;
        float3 inrgb = k2_to_log<LOGT>(make_float3(fmaxf(inp.x, 0.0f), fmaxf(inp.y, 0.0f), fmaxf(inp.z, 0.0f)));
        if (!sampled_i) {
            inrgb = accrgb;
        } else {
            acca += 1.0f;
            const int cr = (spec && roughness < 0.25f) ? 1 : 2;
            // clampNeighborhood reproject.frag:83-95 / getNeighborhoodAABB :53-81 (raw neighbour texels, centre included)
            const float3 ic = k2_from_log<LOGT>(inrgb);
            // The 3x3 core is always inside the window; the outer ring only when the radius is 2.  min/max are order
            // independent, so the two sets are reduced separately (three-operand v_min3/v_max3) and joined by one select.
            const float qnan = __builtin_nanf("");
            float3 mni = ic, mxi = ic;
            float3 mno = make_float3(qnan, qnan, qnan), mxo = mno;
            const float4 *nt = (INPUT_TYPE == 0 && spec) ? s.tex[1] : s.tex[0];
#pragma unroll
            for (int oy = 0; oy < 5; oy++) {  // one row of five 12-byte LDS reads in flight at a time
                const int nrow = __mul24(min(max(y + oy - 2, 0), d.H - 1) - ty0 + AP, LW);  // CLAMP_TO_EDGE row, as an LDS offset
                const float4 t0 = nt[nrow + nxo[0]], t1 = nt[nrow + nxo[1]], t2 = nt[nrow + nxo[2]], t3 = nt[nrow + nxo[3]], t4 = nt[nrow + nxo[4]];
#define K2_RED3(acc_mn, acc_mx, a, b)                                                                                         \
    acc_mn = make_float3(rfx_min3_raw(acc_mn.x, a.x, b.x), rfx_min3_raw(acc_mn.y, a.y, b.y), rfx_min3_raw(acc_mn.z, a.z, b.z)); \
    acc_mx = make_float3(rfx_max3_raw(acc_mx.x, a.x, b.x), rfx_max3_raw(acc_mx.y, a.y, b.y), rfx_max3_raw(acc_mx.z, a.z, b.z))
#define K2_RED2(acc_mn, acc_mx, a)                                                                                 \
    acc_mn = make_float3(rfx_min_raw(acc_mn.x, a.x), rfx_min_raw(acc_mn.y, a.y), rfx_min_raw(acc_mn.z, a.z)); \
    acc_mx = make_float3(rfx_max_raw(acc_mx.x, a.x), rfx_max_raw(acc_mx.y, a.y), rfx_max_raw(acc_mx.z, a.z))
                if (oy >= 1 && oy <= 3) {
                    K2_RED3(mni, mxi, t1, t2);
                    K2_RED2(mni, mxi, t3);
                    K2_RED3(mno, mxo, t0, t4);
                } else {
                    K2_RED3(mno, mxo, t0, t1);
                    K2_RED3(mno, mxo, t2, t3);
                    K2_RED2(mno, mxo, t4);
                }
            }
            float3 mn, mx;
            {
                const bool wide = cr == 2;
                mn = make_float3(wide ? rfx_min_raw(mni.x, mno.x) : mni.x, wide ? rfx_min_raw(mni.y, mno.y) : mni.y, wide ? rfx_min_raw(mni.z, mno.z) : mni.z);
                mx = make_float3(wide ? rfx_max_raw(mxi.x, mxo.x) : mxi.x, wide ? rfx_max_raw(mxi.y, mxo.y) : mxi.y, wide ? rfx_max_raw(mxi.z, mxo.z) : mxi.z);
            }
#undef K2_RED3
#undef K2_RED2
            mn = k2_to_log<LOGT>(mn);
            mx = k2_to_log<LOGT>(mx);
            const float3 clamped = make_float3(rfx_clamp(accrgb.x, mn.x, mx.x), rfx_clamp(accrgb.y, mn.y, mx.y), rfx_clamp(accrgb.z, mn.z, mx.z));
            const float r = spec ? roughness : 1.0f;
            const float aggr = fminf(1.0f, uvc.z * r);
            const float ci2 = rfx_mix(0.0f, fminf(1.0f, moveFactor * 50.0f + p.neighborhoodClampIntensity), aggr);
            const float3 nc = rfx_mix(accrgb, clamped, ci2);
            const float cd = fminf(rfx_length(nc - accrgb), 1.0f);
            acca *= 1.0f - cd;
            accrgb = nc;
        }
        // accumulate() :42-79
        const float conf = rfx_pow(uvc.z, p.confidencePower);  // second power on purpose (Appendix D-6)
        float accumBlend = 1.0f - rfx_rcp(acca + 1.0f);
        accumBlend = rfx_mix(0.0f, accumBlend, conf);
        float maxValue = (p.fullAccumulate ? 1.0f : p.maxBlend) * p.keepData;
        if (INPUT_TYPE != 1) {
            const float rmax = 0.1f;
            if (spec && roughness >= 0.0f && roughness < rmax) {
                const float mrv = rfx_mix(0.0f, maxValue, roughness * 10.0f);
                maxValue = rfx_mix(maxValue, mrv, fminf(100.0f * moveFactor, 1.0f));
            }
        }
        const float mixv = fminf(accumBlend, maxValue);
        acca = fminf(65536.0f, rfx_rcp(1.0f - mixv) - 1.0f);
        const float3 o = k2_from_log<LOGT>(rfx_mix(inrgb, accrgb, mixv));
        float4 texel = make_float4(o.x, o.y, o.z, acca);
        if (p.targetHalf) texel = rfx_round_half4(texel, p.halfStoreRTZ != 0);  // HalfFloatType render target (TemporalReprojectPass.js:63-68)
        ((float4 *)(i ? A.out1.ptr : A.out0.ptr))[oi] = texel;
    }
}

template <int INPUT_TYPE, int TC, bool LOGT, bool HIST_F32, bool WHOLE>
__global__ __launch_bounds__(NT) void k2_temporal_reproject(K2Args A) {
    FrameDims d = A.dims;
    d.viol = 0;
    k2_body<INPUT_TYPE, TC, LOGT, HIST_F32, WHOLE>(A, d);
    rfx_flush_violations(d);
}

// renderer.copyFramebufferToTexture(tmpVec2, this.framebufferTexture) (TemporalReprojectPass.js:198-201): the pass's render
// target becomes its own history.  Both sides have the same type in the reference; here the target always lives in an
// RGBA32F slot, so the HalfFloatType case narrows texels that are half-representable already (drawn with targetHalf).
template <bool TO_HALF>
__global__ __launch_bounds__(256) void k2_copy_framebuffer(FrameDims d, int y0, int y1, TexView src, TexViewW dst) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = y0 + blockIdx.y * 4 + threadIdx.y;
    d.viol = 0;
    if (x < d.W && y < y1) {
        const float4 t = ((const float4 *)src.ptr)[(size_t)rfx_local_row(d, src.row0, src.rows, y) * d.W + x];
        const size_t o = (size_t)rfx_local_row(d, dst.row0, dst.rows, y) * d.W + x;
        if (TO_HALF) ((uint2 *)dst.ptr)[o] = rfx_store_half4(t.x, t.y, t.z, t.w, false);
        else ((float4 *)dst.ptr)[o] = t;
    }
    rfx_flush_violations(d);
}

}  // namespace

hipError_t rfx_launch_copy_fb(const FrameDims &d, int y0, int y1, TexView src, TexViewW dst, bool to_half, hipStream_t stream) {
    dim3 block(64, 4), grid((d.W + 63) / 64, (y1 - y0 + 3) / 4);
    if (to_half) hipLaunchKernelGGL(k2_copy_framebuffer<true>, grid, block, 0, stream, d, y0, y1, src, dst);
    else hipLaunchKernelGGL(k2_copy_framebuffer<false>, grid, block, 0, stream, d, y0, y1, src, dst);
    return hipGetLastError();
}

hipError_t rfx_launch_k2(const K2Args &A, hipStream_t stream) {
    dim3 block(TW, TH), grid(rfx_xcd_grid(K2_XCD_G, (A.dims.W + TW - 1) / TW, (A.y1 - A.y0 + TH - 1) / TH));
    const bool lt = A.p.logTransform != 0;
    // every view is the whole frame (a context that owns no row tile): no row rebasing, no halo accounting in the kernel
    const auto whole_view = [&](const void *ptr, int row0, int rows) { return ptr == nullptr || (row0 == 0 && rows == A.dims.H); };
    const bool whole = whole_view(A.ssgi.ptr, A.ssgi.row0, A.ssgi.rows) && whole_view(A.velocity.ptr, A.velocity.row0, A.velocity.rows) &&
                       whole_view(A.hist0.ptr, A.hist0.row0, A.hist0.rows) && whole_view(A.hist1.ptr, A.hist1.row0, A.hist1.rows) &&
                       whole_view(A.out0.ptr, A.out0.row0, A.out0.rows) && whole_view(A.out1.ptr, A.out1.row0, A.out1.rows);
#define K2_LAUNCH_W(IT, TC, LT, HF)                                                                                         \
    do {                                                                                                                    \
        if (whole) hipLaunchKernelGGL((k2_temporal_reproject<IT, TC, LT, HF, true>), grid, block, 0, stream, A);            \
        else hipLaunchKernelGGL((k2_temporal_reproject<IT, TC, LT, HF, false>), grid, block, 0, stream, A);                 \
    } while (0)
#define K2_LAUNCH(IT, TC)                               \
    do {                                                \
        if (A.hist_f32) {                               \
            if (lt) K2_LAUNCH_W(IT, TC, true, true);    \
            else K2_LAUNCH_W(IT, TC, false, true);      \
        } else {                                        \
            if (lt) K2_LAUNCH_W(IT, TC, true, false);   \
            else K2_LAUNCH_W(IT, TC, false, false);     \
        }                                               \
    } while (0)
    if (A.p.inputType == 0 && A.p.textureCount == 2) K2_LAUNCH(0, 2);
    else if (A.p.inputType == 1 && A.p.textureCount == 1) K2_LAUNCH(1, 1);
    else if (A.p.inputType == 2 && A.p.textureCount == 1) K2_LAUNCH(2, 1);
    else return hipErrorInvalidValue;
#undef K2_LAUNCH
#undef K2_LAUNCH_W
    return hipGetLastError();
}
