// K1 — SSGI ray-march.  Replaces `renderer.render` of src/ssgi/pass/SSGIPass.js:93-94 with
// the fragment program src/ssgi/shader/ssgi.frag (main :105-309, doSample :362-439,
// RayMarch :441-475, BinarySearch :477-503) in the MODE_SSGI / PERSPECTIVE_CAMERA /
// no-env-map variant (the configs carry no env map).
//
// One pixel per lane; a wavefront is 64 consecutive pixels of one row (a 64 x K1_TH-pixel tile, K1_TH = 1): the G-buffer / direct-light /
// output planes are read and written as coalesced 16 B/lane rows.  The march's depth taps are data-dependent gathers anywhere on screen; two
// things keep them off the HBM / fabric path:
//   * k1_prepare (one streaming pre-pass per draw) converts the depth plane to VIEW-SPACE Z once per texel (the same IEEE expression every tap
//     would evaluate, ssgi_utils.frag:9) and reduces it to exact 16x16-texel (min, max) cells, which k1_pack_cells folds into a table of
//     half-packed cells of at most 36 KiB, its rows padded to a power of two where that fits (4K: 32-texel cells, 128 x 68 = 34 KiB);
//   * every tap first consults its cell: when the cell's range proves the texel cannot satisfy `0 <= z - hitPos.z < thickness` (RayMarch :463)
//     — or fixes the sign BinarySearch tests (:493) — the exact texel is never fetched.  The decision is exact, not approximate: fp
//     subtraction is monotonic, so the cell bounds bound the per-texel difference.
// The march kernel is PERSISTENT: as many 8-wave workgroups as the chip holds at once are launched, each copies the (min, max) table into LDS
// once (the only barrier), and every WAVEFRONT then takes tiles from one of 64 device queues until none is left — no workgroup waits for its
// slowest wave, and a cell lookup (two per march step and ray, 64 unrelated addresses each) is an LDS read.  Which tiles a queue holds:
// k1_ssgi_march.  Measurements: profiles/r04_k1, profiles/r05_k1; DESIGN.md §4.
#include "rfx_brdf.h"
#include "rfx_kernels.h"

namespace {

// The (min, max) view-Z table the march consults before touching a texel: one 4-byte cell = two halfs, min rounded
// DOWN and max rounded UP, so a widened range can only reject fewer taps — the rejection tests stay exact.  The cell edge is
// 2^cell_shift texels, chosen per frame size so that the whole table stays <= 36 KiB (rfx_api, which also picks the row layout; 4K: 32-texel cells): every workgroup of the
// march keeps its own copy in LDS.
constexpr int BASE = 16;  // edge of the pre-pass's exact (float) cells, reduced to the final cells by k1_pack_cells
constexpr int K1_TH = 1;               // rows of 64 pixels per tile a wavefront takes from its queue (2 / 4 measured slower: profiles/r04_k1)
typedef uint32_t k1_cell_t;
constexpr int K1_WAVES = 8;            // wavefronts per workgroup of the persistent march kernel
constexpr int K1_COUNTERS = 64;        // tile queues of the persistent march kernel (one queue: +1/3 time; XCD-grouped or static dealing: slower — profiles/r04_k1, r05_k1)
constexpr int K1_TABLE_CELLS = 9216;   // 36 KiB: rfx_api keeps the table within it for every frame size (cell edge doubled until it fits)
RFX_DEV uint32_t k1_half_toward(float v, bool up) {  // nearest half not below (up) / not above (!up) v
    uint32_t h = rfx_f2h_rne(v) & 0xffffu;
    const float f = rfx_h2f((unsigned short)h);
    if (up ? (f < v) : (f > v)) {
        const bool neg = (h & 0x8000u) != 0;
        if (up) h = neg ? (h == 0x8000u ? 0x0001u : h - 1u) : h + 1u;
        else h = neg ? h + 1u : (h == 0x0000u ? 0x8001u : h - 1u);
    }
    return h;
}
RFX_DEV k1_cell_t k1_cell_pack(float mn, float mx) { return k1_half_toward(mn, false) | (k1_half_toward(mx, true) << 16); }
RFX_DEV float2 k1_cell_load(const k1_cell_t *t, unsigned int i) {  // t: the workgroup's LDS copy of the table; i: Tap::cell
    const uint32_t v = *(const k1_cell_t *)((const char *)t + i);
    return make_float2(rfx_h2f((unsigned short)(v & 0xffffu)), rfx_h2f((unsigned short)(v >> 16)));
}

struct MarchCtx {
    const float *P;            // projectionMatrix (column-major)
    const float *viewz;        // full-frame view-space Z plane (k1_prepare)
    const k1_cell_t *coarse;   // (min, max) view Z per 2^cell_shift-texel cell: the LDS copy
    int coarse_w, cell_shift;  // cells per table row, log2 of the cell edge
    unsigned int cell_xmask;   // PROJ_TABLE_POW2: the bits of a cell's byte offset that come from the column, ((pitch - 1) << 2)
    int cell_xshift, cell_yshift;  // ... and the two shifts of k1_tap_at
    float rayDistance, thickness;
    int steps, refineSteps;
};

// viewSpaceToScreenSpace ssgi_utils.frag:26-33.  PERSP: the projection matrix has the sparsity of a (possibly
// off-centre / jittered) perspective matrix — P[1,2,3,4,6,7,12,13,15] == 0, P[11] == -1 — so the general
// mat4*vec4 collapses to the same values (only exact zeros are dropped): x' = P0 x + P8 z, y' = P5 y + P9 z, w = -z.
// The two quotients share one v_rcp_f32 and get one fused Newton step each (the residual x - w*q is exact in an
// fma), i.e. they are correctly rounded except in rare double-rounding cases — 7 VALU ops instead of the ~24 of
// two full IEEE division sequences.  The quotient addresses a NEAREST fetch, so this matters for parity: measured
// K1 stays >99.9 % bit-identical to the oracle.
RFX_DEV float k1_div(float x, float w, float r) {
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-w, q, x), r, q);
}
// PROJ 2: additionally P[8] == P[9] == 0 (a centred frustum: every three.js PerspectiveCamera without a view offset) — the products
// P8 z, P9 z are zeros and adding them changes at most the sign of a zero numerator, which the * 0.5 + 0.5 below erases.
constexpr int PROJ_GENERAL = 0, PROJ_PERSPECTIVE = 1, PROJ_CENTRED = 2;
// ... and, as bit 2 of the same template argument (it reaches every function of the march already), the layout of the (min, max) table:
// PROJ_TABLE_POW2 = its rows are padded to a power of two (K1Args::cells_pow2; k1_tap_at).  A launch-time choice: no branch in the march.
constexpr int PROJ_TABLE_POW2 = 4;
constexpr int k1_proj_kind(int proj) { return proj & 3; }
constexpr bool k1_table_pow2(int proj) { return (proj & PROJ_TABLE_POW2) != 0; }
template <int PROJ_T>
RFX_DEV float2 k1_project(const MarchCtx &m, float3 p) {
    constexpr int PROJ = k1_proj_kind(PROJ_T);
    float px, py, pw;
    if (PROJ == PROJ_CENTRED) {
        px = m.P[0] * p.x;
        py = m.P[5] * p.y;
        pw = -p.z;
    } else if (PROJ == PROJ_PERSPECTIVE) {
        px = m.P[0] * p.x + m.P[8] * p.z;
        py = m.P[5] * p.y + m.P[9] * p.z;
        pw = -p.z;
    } else {
        const float4 pc = rfx_mat_mul(m.P, p.x, p.y, p.z, 1.0f);
        px = pc.x; py = pc.y; pw = pc.w;
    }
    const float r = rfx_rcp(pw);
    // q * 0.5 + 0.5 as ONE fma: the product by 0.5 is exact (or so small that the sum is 0.5 either way), so the fused form rounds once, to
    // the same value (this file is compiled without contraction: the compiler may not make that step itself)
    return make_float2(__builtin_fmaf(k1_div(px, pw, r), 0.5f, 0.5f), __builtin_fmaf(k1_div(py, pw, r), 0.5f, 0.5f));
}

struct Tap {
    unsigned int idx;   // texel index into the view-Z plane (32-bit byte offsets from the wave-uniform base: the plane is < 4 GiB)
    unsigned int cell;  // the cell's byte offset in the (min, max) table
};
template <bool P2>
RFX_DEV Tap k1_tap_at(const MarchCtx &m, const FrameDims &d, int xi, int yi) {
    Tap t;
    t.idx = (unsigned int)(__mul24(yi, d.W) + xi);  // rows and widths are < 2^23: the full-rate 24-bit multiply-add
    if (P2) {
        // the cell's BYTE offset ((yi >> s) << (k + 2)) | ((xi >> s) << 2) with a row pitch of 2^k cells, k >= s >= 2: the column bits are bits
        // 2 .. k+1 of xi >> (s - 2), everything above them bits k+2 .. of yi << (k + 2 - s) (whose low two bits are zero)
        const unsigned int a = (unsigned int)xi >> m.cell_xshift, b = (unsigned int)yi << m.cell_yshift;
        t.cell = (unsigned int)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)m.cell_xmask, 0xE4);  // (a & mask) | (b & ~mask) as ONE v_bitop3_b32 (truth table 0xE4)
    } else {
        t.cell = (unsigned int)(__mul24(yi >> m.cell_shift, m.coarse_w) + (xi >> m.cell_shift)) << 2;
    }
    return t;
}
template <bool P2>
RFX_DEV Tap k1_tap(const MarchCtx &m, const FrameDims &d, float2 uv) {
    return k1_tap_at<P2>(m, d, rfx_nearest_idx(uv.x, d.fW, d.W), rfx_nearest_idx(uv.y, d.fH, d.H));
}
// The taps of both rays of a march step.  rfx_nearest_idx guards every coordinate against |u * size| >= 2^31 (texel 0 in the reference: x86
// cvttss2si) with a compare and a select; a projected uv is that large only when a sample falls within ~1e-6 of the
// camera plane, so the test is made ONCE per step for the whole wavefront (three v_max on the four coordinates, one compare) and the guarded
// form runs in the wavefronts that need it.  v_med3_f32 sends a NaN to 0 as the guard does.  Same indices in every case.
template <bool P2>
RFX_DEV void k1_taps(const MarchCtx &m, const FrameDims &d, const float2 (&uv)[2], Tap (&tap)[2]) {
    const float cx0 = uv[0].x * d.fW, cy0 = uv[0].y * d.fH, cx1 = uv[1].x * d.fW, cy1 = uv[1].y * d.fH;
    const float big = fmaxf(fmaxf(fabsf(cx0), fabsf(cy0)), fmaxf(fabsf(cx1), fabsf(cy1)));
    if (__builtin_amdgcn_ballot_w64(big >= 2147483648.0f) != 0) {
        tap[0] = k1_tap<P2>(m, d, uv[0]);
        tap[1] = k1_tap<P2>(m, d, uv[1]);
    } else {
        const float wm1 = (float)(d.W - 1), hm1 = (float)(d.H - 1);
        tap[0] = k1_tap_at<P2>(m, d, (int)__builtin_amdgcn_fmed3f(cx0, 0.0f, wm1), (int)__builtin_amdgcn_fmed3f(cy0, 0.0f, hm1));
        tap[1] = k1_tap_at<P2>(m, d, (int)__builtin_amdgcn_fmed3f(cx1, 0.0f, wm1), (int)__builtin_amdgcn_fmed3f(cy1, 0.0f, hm1));
    }
}
struct Ray {
    float3 pos, dir;
    float2 uv;
    float live;  // 1.0f while the ray marches, 0.0f once it has hit (or never existed): the step's scale factor, see k1_march_rays
    bool hit;
};
#define K1_ANY_LIVE(rays) (((rays)[0].live != 0.0f) | ((rays)[1].live != 0.0f))
// One march step of both rays.  CS1: the step's cs is exactly 1 (see below) — the position update is then pos + dir * live with an exact
// product, i.e. ONE fma per coordinate with the same bits.
template <int PROJ, bool CS1>
RFX_DEV void k1_march_step(const MarchCtx &m, const FrameDims &d, Ray (&rays)[2], float cs) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
        // straight-line: a stopped ray advances by dir * 0 (pos + (+-0) == pos) and re-derives the same uv — no exec-mask region per ray
        // (same texels).  `live` is the select as a product: cs * 1 == cs, cs * 0 == +0 (cs is in [0, 1]).
        if (CS1) {
            rays[r].pos = make_float3(__builtin_fmaf(rays[r].dir.x, rays[r].live, rays[r].pos.x), __builtin_fmaf(rays[r].dir.y, rays[r].live, rays[r].pos.y),
                                      __builtin_fmaf(rays[r].dir.z, rays[r].live, rays[r].pos.z));
        } else {
            const float csr = cs * rays[r].live;
            rays[r].pos = rays[r].pos + rays[r].dir * csr;
        }
        rays[r].uv = k1_project<PROJ>(m, rays[r].pos);
    }
    // taps of both rays in flight together: the two coarse cells first, then the exact texels of the cells that cannot
    // rule a hit out (a hit needs 0 <= z - h < thickness; the cell range rules it out when max - h < 0 or min - h >= thickness)
    Tap tap[2];
    float2 mm[2];
    bool need[2];
    {
        const float2 uvs[2] = {rays[0].uv, rays[1].uv};
        k1_taps<k1_table_pow2(PROJ)>(m, d, uvs, tap);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) mm[r] = k1_cell_load(m.coarse, tap[r].cell);
#pragma unroll
    for (int r = 0; r < 2; r++) {  // (bitwise on purpose: no short-circuit branches in the loop)
        const float h = rays[r].pos.z;
        need[r] = (rays[r].live != 0.0f) & !((mm[r].y - h < 0.0f) | (mm[r].x - h >= m.thickness));
    }
    // ONE exec region and ONE wait for both rays' exact texels (a random 4-byte gather over a 33 MB plane each); a lane that needs only one of its
    // two texels fetches that one twice (the same address: no extra cache line).  The hit tests live inside the same region: a step in which no
    // lane of the wavefront needs an exact texel (most steps of most wavefronts) executes neither the fetches nor the two rays' subtract /
    // compare / compare / select.  (Per-ray regions, tests outside the region, unconditional fetches and the step on (ray 0, ray 1) float2
    // pairs were all built and measured, same texels, none faster: profiles/r04_k1, profiles/r05_k1, profiles/r06_cleanup.)
    if (need[0] | need[1]) {
        const unsigned int i0 = need[0] ? tap[0].idx : tap[1].idx, i1 = need[1] ? tap[1].idx : tap[0].idx;
        const float z0 = rfx_gather<float>(m.viewz, i0), z1 = rfx_gather<float>(m.viewz, i1);
        const float d0 = z0 - rays[0].pos.z, d1 = z1 - rays[1].pos.z;
        rays[0].live = (need[0] & (d0 >= 0.0f) & (d0 < m.thickness)) ? 0.0f : rays[0].live;  // a hit: the ray stops here
        rays[1].live = (need[1] & (d1 >= 0.0f) & (d1 < m.thickness)) ? 0.0f : rays[1].live;
    }
}
// BinarySearch (:477-503) for the pixel's two rays in their two slots (rays that did not hit idle along): the form used when the wavefront's
// hit rays do not fit one per lane
template <int PROJ>
RFX_DEV void k1_refine_pairs(const MarchCtx &m, const FrameDims &d, Ray (&rays)[2]) {
    // (a ray that did not hit takes steps of dir * 0: its position is replaced below anyway, its direction is not read again)
#pragma unroll
    for (int r = 0; r < 2; r++) {
        rays[r].dir = rays[r].dir * 0.5f;
        rays[r].pos = rays[r].pos + rays[r].dir * (rays[r].hit ? -1.0f : 0.0f);  // pos - dir, exactly
    }
    for (int k = 0; k < m.refineSteps; k++) {
#pragma unroll
        for (int r = 0; r < 2; r++)
            if (rays[r].hit) rays[r].uv = k1_project<PROJ>(m, rays[r].pos);
        // BinarySearch only tests the sign of z_tap - h (:493): decided by the cell when max - h < 0 or min - h >= 0
        Tap tap[2];
        float2 mm[2];
        bool need[2], behind[2];
#pragma unroll
        for (int r = 0; r < 2; r++) tap[r] = k1_tap<k1_table_pow2(PROJ)>(m, d, rays[r].uv);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            need[r] = rays[r].hit;
            behind[r] = false;
        }
#pragma unroll
        for (int r = 0; r < 2; r++) mm[r] = k1_cell_load(m.coarse, tap[r].cell);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float h = rays[r].pos.z;
            const bool below = mm[r].y - h < 0.0f;          // diff < 0 everywhere in the cell
            const bool above = !below & (mm[r].x - h >= 0.0f);  // diff >= 0 everywhere
            need[r] = need[r] & !(below | above);
            behind[r] = above;
        }
        float z[2];
#pragma unroll
        for (int r = 0; r < 2; r++) z[r] = need[r] ? rfx_gather<float>(m.viewz, tap[r].idx) : 0.0f;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (need[r]) behind[r] = z[r] - rays[r].pos.z >= 0.0f;
            rays[r].dir = rays[r].dir * 0.5f;
            // pos -+ dir as pos + dir * (-+1): the product is exact, so the sum rounds as the difference does
            rays[r].pos = rays[r].pos + rays[r].dir * (rays[r].hit ? (behind[r] ? -1.0f : 1.0f) : 0.0f);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
        if (rays[r].hit) rays[r].uv = k1_project<PROJ>(m, rays[r].pos);
}
// ... and ONE ray per lane: of the 128 ray slots of a wavefront typically 35-45 hold a ray that hit, and the five refinement steps cost a
// fifth of K1's instructions with most lanes idle in both slots.  When all 64 lanes are here and at most 63 rays hit, the hit rays are
// packed into lanes 0 .. n-1 (ds_permute: lane i sends slot r's position and direction to lane rank_r; a lane without that ray sends to lane
// 63, which no rank reaches), refined there with the same arithmetic, and fetched back by their owners (ds_bpermute from lane rank_r).
// Nothing is staged in LDS memory (the permutes use the LDS crossbar only) and nothing waits: it is the wavefront's own business.
RFX_DEV float k1_push(int dest_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_permute(dest_lane << 2, __float_as_int(v))); }
RFX_DEV float k1_pull(int src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
template <int PROJ>
RFX_DEV void k1_refine_packed(const MarchCtx &m, const FrameDims &d, Ray (&rays)[2], unsigned long long h0, unsigned long long h1, int n0, int n1) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int rank0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(h0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)h0, 0u));
    const int rank1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(h1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)h1, 0u));
    const int dest0 = rays[0].hit ? rank0 : 63, dest1 = rays[1].hit ? rank1 : 63;
    const bool from0 = lane < n0, mine = lane < n0 + n1;
    float3 pos, dir;
#define K1_PACK(dst, field) { const float a = k1_push(dest0, rays[0].field), b = k1_push(dest1, rays[1].field); dst = from0 ? a : b; }
    K1_PACK(pos.x, pos.x) K1_PACK(pos.y, pos.y) K1_PACK(pos.z, pos.z) K1_PACK(dir.x, dir.x) K1_PACK(dir.y, dir.y) K1_PACK(dir.z, dir.z)
#undef K1_PACK
    // the same steps as k1_refine_pairs, for this lane's one ray (a lane beyond the last rank holds zeros and idles)
    dir = dir * 0.5f;
    pos = pos + dir * (mine ? -1.0f : 0.0f);  // pos - dir, exactly
    float2 uv = make_float2(0.f, 0.f);
    for (int k = 0; k < m.refineSteps; k++) {
        if (mine) uv = k1_project<PROJ>(m, pos);
        const Tap tap = k1_tap<k1_table_pow2(PROJ)>(m, d, uv);
        const float2 mm = k1_cell_load(m.coarse, tap.cell);
        const float h = pos.z;
        const bool below = mm.y - h < 0.0f;             // diff < 0 everywhere in the cell
        const bool above = !below & (mm.x - h >= 0.0f);  // diff >= 0 everywhere
        const bool need = mine & !(below | above);
        bool behind = above;
        const float z = need ? rfx_gather<float>(m.viewz, tap.idx) : 0.0f;
        if (need) behind = z - pos.z >= 0.0f;
        dir = dir * 0.5f;
        pos = pos + dir * (mine ? (behind ? -1.0f : 1.0f) : 0.0f);
    }
    if (mine) uv = k1_project<PROJ>(m, pos);
    // every owner fetches its rays back (all lanes take part in the permutes; the value is kept only where the slot's ray hit)
#define K1_UNPACK(val, f0, f1) { const float a = k1_pull(rank0, val), b = k1_pull(rank1, val); if (rays[0].hit) f0 = a; if (rays[1].hit) f1 = b; }
    K1_UNPACK(pos.x, rays[0].pos.x, rays[1].pos.x) K1_UNPACK(pos.y, rays[0].pos.y, rays[1].pos.y) K1_UNPACK(pos.z, rays[0].pos.z, rays[1].pos.z)
    K1_UNPACK(uv.x, rays[0].uv.x, rays[1].uv.x) K1_UNPACK(uv.y, rays[0].uv.y, rays[1].uv.y)
#undef K1_UNPACK
}
// RayMarch (:441-475) + BinarySearch (:477-503) for the pixel's TWO rays at once (slot 0 = optional diffuse ray,
// slot 1 = specular ray), restructured for the SIMT machine without changing any per-ray arithmetic:
//   * both rays advance in the same loop iteration (they share cs(i), and their taps are in flight together);
//   * a ray that finds a hit only stops marching (live = 0; "hit" = started and no longer live, derived once after the loop); the binary
//     search runs ONCE after the march loop for every ray that hit.  In the GLSL the search is nested in the march loop, so a wavefront
//     whose lanes hit at k different steps executes the 5-step search k times with mostly idle lanes — here it is 19 + 5 iterations, always;
//   * cs(i) = 1 - exp(-t^2 / 4) with t = i + random.b - 0.5 >= i - 0.5 (:453-454): from i = 9 on, t >= 8.5 and exp(-t^2 / 4) <= 1.5e-8 <
//     2^-25, so the subtraction rounds to exactly 1.0f — in the reference's fp32 as here — and `dir * cs` is `dir`: the second loop below
//     evaluates no cs at all (no v_exp, no products), the same bits.
// Other forms of this loop that were built and measured, all with the same texels and all slower (profiles/r04_k1, profiles/r05_k1): packing a
// wavefront's live rays one per lane through the LDS crossbar once they fit, marching two steps per iteration with the second speculated
// under the first's fetches, and the step's arithmetic on (ray 0, ray 1) float2 pairs (profiles/r06_cleanup/k1_rejected_variants.patch).
template <int PROJ>
RFX_DEV void k1_march_rays(const MarchCtx &m, const FrameDims &d, Ray (&rays)[2], float random_b) {
    const float scale = m.rayDistance / (float)m.steps;
    const bool started[2] = {rays[0].live != 0.0f, true};  // (the specular ray always marches)
#pragma unroll
    for (int r = 0; r < 2; r++) {
        rays[r].dir = rays[r].dir * scale;
        rays[r].uv = make_float2(0.f, 0.f);
    }
    const int split = min(m.steps, 9);
    int i = 1;
    for (; i < split && K1_ANY_LIVE(rays); i++) {
        const float t = (float)i + random_b - 0.5f;
        // exp(-0.25 t^2) = exp2((-0.25 t^2) log2e): the scaling by -1/4 is exact, so it folds into the constant (one product instead of two,
        // the same bits)
        const float cs = 1.0f - rfx_exp2((t * t) * (-0.25f * 1.4426950408889634f));
        k1_march_step<PROJ, false>(m, d, rays, cs);
    }
    for (; i < m.steps && K1_ANY_LIVE(rays); i++) k1_march_step<PROJ, true>(m, d, rays, 1.0f);
#pragma unroll
    for (int r = 0; r < 2; r++) rays[r].hit = started[r] & (rays[r].live == 0.0f);
    // (a wavefront none of whose rays hit — sky above the horizon, a wall facing away — has nothing to refine)
    if (m.refineSteps > 0) {
        const unsigned long long h0 = __ballot(rays[0].hit), h1 = __ballot(rays[1].hit);
        const int n0 = __popcll(h0), n1 = __popcll(h1);
        if (n0 + n1 == 0) {
        } else if (n0 + n1 <= 63 && __ballot(1) == ~0ull) {
            k1_refine_packed<PROJ>(m, d, rays, h0, h1, n0, n1);
        } else {
            k1_refine_pairs<PROJ>(m, d, rays);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
        if (!rays[r].hit) rays[r].pos = make_float3(10.0e9f, 10.0e9f, 10.0e9f);  // :472
}

// smoothstep with constant edges: the divisor e1 - e0 is a constant, the quotient the correctly rounded one (RFX_DIV_CONST)
#define K1_SMOOTHSTEP(e0, e1, x) k1_smoothstep_q(RFX_DIV_CONST((x) - (e0), (e1) - (e0)))
RFX_DEV float k1_smoothstep_q(float q) {
    const float t = rfx_clamp(q, 0.0f, 1.0f);
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

// doSample :362-439 without env map (getEnvColor == 0), split around the march: the BRDF/pdf factor first ...
RFX_DEV float k1_brdf_over_pdf_parts(const Material &mat, float3 viewNormal, float roughness, bool isDiffuseSample, float NoV, const Angles &an, float3 l,
                                      float &pdf) {
    const float cosTheta = fmaxf(0.0f, rfx_dot(viewNormal, l));
    float brdf;
    if (isDiffuseSample) {
        brdf = rfx_eval_disney_diffuse(an.NoL, NoV, an.LoH, roughness, mat.metalness);
        pdf = RFX_DIV_CONST(an.NoL, RFX_PI);
    } else {
        brdf = rfx_eval_disney_specular(roughness, an.NoH, NoV, an.NoL);
        pdf = rfx_ggx_vndf_pdf(an.NoH, NoV, roughness);
    }
    brdf *= cosTheta;
    pdf = fmaxf(0.00001f, pdf);
    return brdf;
}
// ---- scene.environment (USE_ENVMAP).  One CLAMP_TO_EDGE bilinear tap of a mip level, as the oracle's sampler computes it
RFX_DEV float3 k1_env_level(const K1Args &A, int level, float u, float v) {
    const int w = max(A.env_w >> level, 1), h = max(A.env_h >> level, 1);
    const float4 *t = A.env + A.env_off[level];
    int x0, x1, y0, y1;
    float wx, wy;
    rfx_linear_coord(u, (float)w, w, x0, x1, wx);
    rfx_linear_coord(v, (float)h, h, y0, y1, wy);
    const float4 t00 = t[y0 * w + x0], t10 = t[y0 * w + x1], t01 = t[y1 * w + x0], t11 = t[y1 * w + x1];
    return make_float3(rfx_lerp(wy, rfx_lerp(wx, t00.x, t10.x), rfx_lerp(wx, t01.x, t11.x)), rfx_lerp(wy, rfx_lerp(wx, t00.y, t10.y), rfx_lerp(wx, t01.y, t11.y)),
                       rfx_lerp(wy, rfx_lerp(wx, t00.z, t10.z), rfx_lerp(wx, t01.z, t11.z)));
}
// acos as the oracle's GLSL compiler evaluates it (Mesa: pi/2 - asin polynomial, |err| <= 1.6e-4 rad) — cheaper than libm's and
// the same function on both sides of the parity test
RFX_DEV float k1_acos(float x) {
    const float ax = fabsf(x);
    const float r = 1.5707963267948966f - rfx_sqrt(1.0f - ax) * (1.5707963267948966f + ax * (-0.21460183660255172f + ax * (0.08132463f + ax * -0.02363318f)));
    return 1.5707963267948966f - (x < 0.0f ? -r : r);
}
// getEnvColor ssgi.frag:311-346 (no BOX_PROJECTED_ENV_MAP, isEnvSample false without MIS): textureLod(map, equirectDirectionToUv(dir), mip)
// with LinearMipMapLinearFilter = two bilinear taps blended by fract(lod), lod clamped to the chain
RFX_DEV float3 k1_env_trilinear(const K1Args &A, float u, float v, float lod_unclamped) {
    const float lod = fminf(fmaxf(lod_unclamped, 0.0f), (float)(A.env_levels - 1));
    const float fl = floorf(lod);
    const int l0 = (int)fl, l1 = min(l0 + 1, A.env_levels - 1);
    const float3 c0 = k1_env_level(A, l0, u, v), c1 = k1_env_level(A, l1, u, v);
    const float f = lod - fl;
    return make_float3(rfx_lerp(f, c0.x, c1.x), rfx_lerp(f, c0.y, c1.y), rfx_lerp(f, c0.z, c1.z));
}
RFX_DEV float3 k1_env_color(const K1Args &A, float3 l, float roughness, bool isDiffuseSample, bool isEnvSample) {
    const float3 dir = rfx_normalize(rfx_vec_mul_mat(A.p.camera.matrixWorldInverse, l, 0.0f));  // (vec4(l, 0.) * viewMatrix).xyz :315
    float mip = A.p.envBlur * A.maxEnvMapMipLevel;
    if (!isDiffuseSample && roughness < 0.15f) mip *= RFX_DIV_CONST(roughness, 0.15f);
    // equirectDirectionToUv ssgi_utils.frag:64-74
    float u = RFX_DIV_CONST(atan2f(dir.z, dir.x), 2.0f * 3.1415926535897932384626433832795f), v = RFX_DIV_CONST(k1_acos(dir.y), 3.1415926535897932384626433832795f);
    u += 0.5f;
    v = 1.0f - v;
    float3 c = k1_env_trilinear(A, u, v, mip);
    const float maxEnvLum = isEnvSample ? 100.0f : 25.0f, envLum = rfx_lum(c);  // :328-340
    // (an HDR texel can be huge or infinite: rfx_div_pos needs a divisor well inside the exponent range, IEEE `/` gives 0 for inf)
    if (envLum > maxEnvLum) c = c * (envLum < 1.0e30f ? rfx_div_pos(maxEnvLum, envLum) : maxEnvLum / envLum);
    return c;
}

// ---- importanceSampling (ssgi.frag:197-216, sampleEquirectProbability ssgi_utils.frag:210-225)
// uv of the environment texel the pixel's blue-noise pair selects through the two inverse-CDF tables: marginalWeights is an env_h x 1 NEAREST
// texture read at (blueNoise.x, 0), conditionalWeights an env_w x env_h one read at (blueNoise.y, v)
// A pixel that is background (main() returned before the blue-noise fetch, ssgi.frag:109-113) takes part in its quad's derivatives with
// `random` = 0, its zero initialisation: GLSL leaves derivatives after a non-uniform return undefined, this is what the oracle's GL does
// (measured).  A quad partner OUTSIDE the target (the last column / row of an odd-sized one) is a helper invocation that runs the fragment on its
// extrapolated vUv: the depth fetch clamps to the edge texel, its blue-noise pixel is (px, py) beyond the target (round 6: the reference GLSL at
// random odd sizes, tools/fuzz_variants_vs_reference_gl.py).
RFX_DEV float2 k1_cdf_uv(const K1Args &A, const FrameDims &d, int px, int py) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    const int sx = rfx_nearest_idx(rfx_frag_u(A.out_uv, px, py), d.fW, d.W), sy = rfx_nearest_idx(rfx_frag_v(A.out_uv, py), d.fH, d.H);
    if (((const float *)A.depth.ptr)[rfx_xy_index(d, A.depth.row0, A.depth.rows, sx, sy)] != 1.0f)
        r = rfx_blue_noise((const uchar4 *)A.blue, px, py, A.shift_x, A.shift_y);
    const float v = A.env_marginal[rfx_nearest_idx(r.x, (float)A.env_h, A.env_h)];
    const float u = A.env_conditional[rfx_nearest_idx(v, (float)A.env_h, A.env_h) * A.env_w + rfx_nearest_idx(r.y, (float)A.env_w, A.env_w)];
    return make_float2(u, v);
}
// `texture(info.map, uv)` — implicit level of detail.  uv is an unrelated value in every pixel, so the level is whatever the rasteriser derives
// from the 2x2 quad; the oracle's (llvmpipe, measured to 4 decimals on a chain whose level l holds the constant l): ONE lod per quad, from
// the differences tr - tl and bl - tl of the quad's top-left pixel, rho^2 = max(|dP/dx|^2, |dP/dy|^2) with P in texels, and
// lod = 0.5 * (exponent(rho^2) + mantissa(rho^2) - 1)  (a linear-mantissa log2), clamped to the chain, blended by fract(lod).
RFX_DEV float k1_implicit_lod(const K1Args &A, float2 tl, float2 tr, float2 bl) {
    const float fw = (float)A.env_w, fh = (float)A.env_h;
    const float ax = (tr.x - tl.x) * fw, ay = (tr.y - tl.y) * fh, bx = (bl.x - tl.x) * fw, by = (bl.y - tl.y) * fh;
    const float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
    const uint32_t bits = __float_as_uint(rho2);
    const float e = (float)((int)((bits >> 23) & 0xffu) - 127), m = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u);
    return 0.5f * (e + (m - 1.0f));
}
struct EnvMis {  // EnvMisSample ssgi.frag:79-83
    float pdf;
    bool isEnvSample;
};
RFX_DEV float k1_mis_heuristic(float a, float b) {  // misHeuristic ssgi_utils.frag:227-231 (the pdfs are unbounded: the refined reciprocal only inside its range)
    const float n = a * a, d = a * a + b * b;
    return (d > 1.0e-30f && d < 1.0e30f) ? rfx_div_pos(n, d) : n / d;
}

// ... and the shading of the marched ray: gi * brdf / pdf
template <bool ENV, bool MIS>
RFX_DEV float3 k1_shade(const FrameDims &d, const K1Args &A, const Material &mat, float roughness, const Ray &ray, float3 l, bool isDiffuseSample, float brdf,
                        float pdf, EnvMis ems) {
    const bool allowMissed = A.p.missedRays != 0;
    const bool isMissed = ray.pos.x == 10.0e9f;
    // without an env map getEnvColor is black (:342-345)
    float3 env = make_float3(0.f, 0.f, 0.f);
    if (ENV) env = k1_env_color(A, l, roughness, isDiffuseSample, MIS && ems.isEnvSample);
    float3 ssgi = env;
    if (!(isMissed && !allowMissed)) {  // :393-395 a missed ray takes the environment
        // velocityTexture is never wired in the reference (SSGIPass.js:89) -> velocity == 0
        const float2 coords = ray.uv;
        const float ru = coords.x, rv = coords.y;
        if (ru >= 0.0f && ru <= 1.0f && rv >= 0.0f && rv <= 1.0f) {
            // accumulatedTexture: K4's output, K2's texture[0], or (denoiseMode "denoised") three's empty texture — rfx.h historySource
            float3 gi = make_float3(0.f, 0.f, 0.f);
            if (A.p.historySource == 3) {  // RFX_TEX_COMPOSE_RGB: the same .rgb as 12-byte texels
                const float *h = (const float *)A.history.ptr + rfx_texel_index(d, A.history.row0, A.history.rows, ru, rv) * 3;
                gi = make_float3(h[0], h[1], h[2]);
            } else if (A.p.historySource != 2) {
                const float4 h = rfx_fetch_f4(A.history, d, ru, rv);
                gi = make_float3(h.x, h.y, h.z);
            }
            const float mx = fmaxf(fmaxf(mat.diffuse.x, mat.diffuse.y), mat.diffuse.z);
            const float mn = fminf(fminf(mat.diffuse.x, mat.diffuse.y), mat.diffuse.z);
            const float sat = (mx == mn) ? 0.0f : rfx_div_pos(mx - mn, mx);  // getSaturation :348-360 (mx > mn >= 0: a byte / 255 - 1e-4, at least 0.0038)
            const float L = rfx_lum(gi);
            gi = rfx_mix(gi, make_float3(L, L, L), (1.0f - roughness) * sat * 0.4f);
            const float border = 0.15f;
            float bf = K1_SMOOTHSTEP(0.0f, border, coords.x) * K1_SMOOTHSTEP(1.0f, 1.0f - border, coords.x) * K1_SMOOTHSTEP(0.0f, border, coords.y) *
                       K1_SMOOTHSTEP(1.0f, 1.0f - border, coords.y);
            bf = rfx_sqrt(bf);
            ssgi = rfx_mix(env, gi, bf);  // :424
            if (allowMissed && 0.0f > rfx_lum(ssgi)) ssgi = make_float3(0.f, 0.f, 0.f);  // :430-436: `envMapSample` is never assigned -> 0
        }
        // else :425-427 the reprojected coordinates left the screen: the environment
    }
    ssgi = ssgi * brdf;  // :236-244 / :256-264
    if (MIS && ems.isEnvSample) ssgi = ssgi * k1_mis_heuristic(ems.pdf, pdf);
    else {  // pdf >= 1e-5 (k1_brdf_over_pdf_parts): one refined reciprocal for the three quotients
        const float r = rfx_rcp_rn(pdf);
        ssgi = make_float3(rfx_div_const_impl(ssgi.x, pdf, r), rfx_div_const_impl(ssgi.y, pdf, r), rfx_div_const_impl(ssgi.z, pdf, r));
    }
    if (MIS) ssgi = make_float3(ssgi.x / ems.pdf, ssgi.y / ems.pdf, ssgi.z / ems.pdf);  // without MIS ems.pdf == 1
    return ssgi;
}

// STAGE 0: the whole fragment in one launch.  STAGE 1 ("trace") stops after the march and leaves the two rays' end state in
// A.hits (2 x float4 per pixel: uv0 uv1 | pos0.x pos1.xyz); STAGE 2 ("shade") redoes the cheap per-pixel setup, takes the rays
// from A.hits instead of marching and finishes the fragment.  Only the shading reads last frame's composed GI anywhere on
// screen, so a row-tiled run can let that texture's all-gather overlap the march (rfx.h rfx_ssgi_trace / rfx_ssgi_shade).
// Same arithmetic in the same order either way (no contraction in this file): split == fused bit for bit (tests).
template <int PROJ, bool ENV, bool MIS, int STAGE>
RFX_DEV void k1_ssgi_march_body(const K1Args &A, const FrameDims &d, const k1_cell_t *s_cells, int x, int y) {
    if (x >= A.out_w || y >= A.y1) return;
    const rfx_ssgi_params &p = A.p;
    const float *C = p.camera.matrixWorld, *Vw = p.camera.matrixWorldInverse;
    const float *P = p.camera.projectionMatrix, *Pi = p.camera.projectionMatrixInverse;

    // vUv of the (possibly smaller, resolutionScale) render target; the full-resolution inputs are fetched NEAREST at vUv
    const bool scaled = A.out_w != d.W || A.out_h != d.H;
    const float u = rfx_frag_u(A.out_uv, x, y), v = rfx_frag_v(A.out_uv, y);
    const int sx = scaled ? rfx_nearest_idx(u, d.fW, d.W) : x, sy = scaled ? rfx_nearest_idx(v, d.fH, d.H) : y;
    const float depth = ((const float *)A.depth.ptr)[rfx_xy_index(d, A.depth.row0, A.depth.rows, sx, sy)];
    const size_t out_idx = scaled ? (size_t)y * A.out_w + x : (size_t)rfx_local_row(d, A.out.row0, A.out.rows, y) * d.W + x;
    uint4 *outp = (uint4 *)A.out.ptr + out_idx;
    const size_t gi_idx = rfx_xy_index(d, A.direct.row0, A.direct.rows, sx, sy);
    if (depth == 1.0f) {  // background :109-113
        if (STAGE != 1) {
            const float4 dl = ((const float4 *)A.direct.ptr)[gi_idx];
            *outp = rfx_pack_two_vec4(dl, dl);
        }
        return;
    }
    const Material mat = rfx_get_material<false>(((const uint4 *)A.gbuffer.ptr)[rfx_xy_index(d, A.gbuffer.row0, A.gbuffer.rows, sx, sy)]);
    const float roughnessSq = rfx_clamp(mat.roughness * mat.roughness, 0.000001f, 1.0f);

    MarchCtx m;
    m.P = P;
    m.viewz = A.viewz;
    m.coarse = s_cells;
    m.coarse_w = A.cells_pitch;
    m.cell_shift = A.cell_shift;
    m.cell_xmask = (unsigned int)(A.cells_pitch - 1) << 2;
    m.cell_xshift = A.cell_shift - 2;
    m.cell_yshift = A.cells_pow2 ? A.cells_pitch_log2 + 2 - A.cell_shift : 0;
    m.rayDistance = p.rayDistance;
    m.thickness = p.thickness;
    m.steps = p.steps;
    m.refineSteps = p.refineSteps;

    const float viewZ = A.viewz[(size_t)sy * d.W + sx];  // getViewZ(depth) ssgi_utils.frag:7-13, from the pre-pass
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
    bool isDiffuseSample = false;  // MODE_SSR: never (:187-189)
    if (p.mode == 0) {
        const float3 F = rfx_f_schlick(f0, an.VoH);
        float diffW = (1.0f - mat.metalness) * rfx_lum(mat.diffuse);
        float specW = rfx_lum(F);
        diffW = fmaxf(diffW, 0.00001f);
        specW = fmaxf(specW, 0.00001f);
        const float invW = 1.0f / (diffW + specW);
        diffW *= invW;
        isDiffuseSample = rnd.z < diffW;
    }
    EnvMis ems;
    ems.pdf = 1.0f;
    ems.isEnvSample = false;
    float3 envMisDir = make_float3(0.f, 0.f, 0.f);
    if (MIS) {  // :197-216
        const float2 uv = k1_cdf_uv(A, d, x, y);
        // equirectUvToDirection ssgi_utils.frag:77-86
        float sph, cph, sth, cth;
        {
            const float theta = ((uv.x - 0.5f) * 2.0f) * 3.141592653589793f, phi = (1.0f - uv.y) * 3.141592653589793f;
            sph = sinf(phi); cph = cosf(phi); sth = sinf(theta); cth = cosf(theta);
        }
        const float3 derived = make_float3(sph * cth, cph, sph * sth);
        const int qx = x & ~1, qy = y & ~1;
        const float lod = k1_implicit_lod(A, k1_cdf_uv(A, d, qx, qy), k1_cdf_uv(A, d, qx + 1, qy), k1_cdf_uv(A, d, qx, qy + 1));
        const float3 col = k1_env_trilinear(A, uv.x, uv.y, lod);
        const float totalSum = A.totalSumWhole + A.totalSumDecimal;
        ems.pdf = ((float)A.env_w * (float)A.env_h) * (rfx_lum(col) / totalSum);
        envMisDir = rfx_normalize(rfx_vec_mul_mat(C, derived, 0.0f));  // (vec4(dir, 0.) * cameraMatrixWorld).xyz :199
        float prob = rfx_dot(envMisDir, viewNormal);
        prob *= mat.roughness;
        prob = fminf(1.0f - 0.00001f, prob);
        ems.isEnvSample = rnd.w < prob;
        if (ems.isEnvSample) {
            ems.pdf /= 1.0f - prob;
            l = envMisDir;
        } else {
            ems.pdf = 1.0f - prob;
        }
    }
    const float3 specularRay = l;  // :218-219 (l already is envMisDir for an env sample)
    float3 dl = make_float3(0.f, 0.f, 0.f);
    if (p.useDirectLight) {
        const float4 t = ((const float4 *)A.direct.ptr)[gi_idx];
        dl = make_float3(t.x, t.y, t.z);
    }
    Ray rays[2];
    float3 diffuseRayDir = make_float3(0.f, 0.f, 0.f);
    float brdfD = 0.f, pdfD = 1.f, brdfS, pdfS;
    rays[0].live = isDiffuseSample ? 1.0f : 0.0f;
    rays[0].pos = viewPos;
    rays[0].dir = make_float3(0.f, 0.f, 0.f);
    if (isDiffuseSample) {  // :222-242
        const float3 diffuseRay = (MIS && ems.isEnvSample) ? envMisDir : rfx_cosine_sample_hemisphere(viewNormal, rnd.x, rnd.y);
        diffuseRayDir = diffuseRay;
        const Angles ad = k1_angles(diffuseRay, vv, n);
        brdfD = k1_brdf_over_pdf_parts(mat, viewNormal, roughnessSq, true, NoV, ad, diffuseRay, pdfD);
        rays[0].dir = diffuseRay;
    }
    // specular ray, traced every frame — evaluated with the SAME isDiffuseSample flag (:246-265)
    an = k1_angles(specularRay, vv, n);
    brdfS = k1_brdf_over_pdf_parts(mat, viewNormal, roughnessSq, isDiffuseSample, NoV, an, specularRay, pdfS);
    rays[1].live = 1.0f;
    rays[1].pos = viewPos;
    rays[1].dir = specularRay;
    if (STAGE == 2) {
        const float4 h0 = A.hits[2 * out_idx], h1 = A.hits[2 * out_idx + 1];
        rays[0].uv = make_float2(h0.x, h0.y);
        rays[1].uv = make_float2(h0.z, h0.w);
        rays[0].pos = make_float3(h1.x, h1.x, h1.x);  // only "missed" (pos.x == 10.0e9) is read of the diffuse ray
        rays[1].pos = make_float3(h1.y, h1.z, h1.w);
    } else {
        k1_march_rays<PROJ>(m, d, rays, rnd.z);
    }
    if (STAGE == 1) {
        A.hits[2 * out_idx] = make_float4(rays[0].uv.x, rays[0].uv.y, rays[1].uv.x, rays[1].uv.y);
        A.hits[2 * out_idx + 1] = make_float4(rays[0].pos.x, rays[1].pos.x, rays[1].pos.y, rays[1].pos.z);
        return;
    }

    float3 diffuseGI = make_float3(-1.0f, -1.0f, -1.0f);  // "not sampled this frame" marker :277-278
    if (isDiffuseSample) diffuseGI = k1_shade<ENV, MIS>(d, A, mat, roughnessSq, rays[0], diffuseRayDir, true, brdfD, pdfD, ems) + dl;
    const float3 specularGI = k1_shade<ENV, MIS>(d, A, mat, roughnessSq, rays[1], specularRay, isDiffuseSample, brdfS, pdfS, ems) + dl;
    const float3 hitPos = rays[1].pos;

    float rayLength = 0.0f;  // :284-296
    if (!(hitPos.x > 10.0e8f)) {
        const float4 hw = rfx_mat_mul(C, hitPos.x, hitPos.y, hitPos.z, 1.0f);
        rayLength = rfx_length(make_float3(C[12], C[13], C[14]) - make_float3(hw.x, hw.y, hw.z));
    }
    if (p.mode == 0) {  // :302-304
        *outp = rfx_pack_two_vec4(make_float4(diffuseGI.x, diffuseGI.y, diffuseGI.z, mat.roughness),
                                  make_float4(specularGI.x, specularGI.y, specularGI.z, rayLength));
    } else {  // MODE_SSR :298-300,306-307: raw vec4(specularGI, uintBitsToFloat(packHalf2x16(vec2(rayLength, roughness))))
        *outp = make_uint4(__float_as_uint(specularGI.x), __float_as_uint(specularGI.y), __float_as_uint(specularGI.z), rfx_pack_half2(rayLength, mat.roughness));
    }
}

// (without an environment map the fragment fits 64 VGPRs = the hardware's 8 waves per SIMD; the bound keeps the register allocator there)
// K1Args must stay the FIRST by-value parameter: RFX_KERNARGS_IN_LOOP re-reads it from offset 0 of the kernarg segment (rfx_device.h)
template <int PROJ, bool ENV, bool MIS, int STAGE>
__global__ __launch_bounds__(64 * K1_WAVES) RFX_WAVES_PER_EU(ENV ? 1 : 8) void k1_ssgi_march(K1Args A) {
    __shared__ k1_cell_t s_cells[K1_TABLE_CELLS];
    if (STAGE != 2) {  // (the shade stage marches nothing)
        for (int i = threadIdx.x; i < A.cells_vec4; i += 64 * K1_WAVES) ((uint4 *)s_cells)[i] = ((const uint4 *)A.cells)[i];
        __syncthreads();  // the only barrier: from here on every wavefront runs on its own
    }
    FrameDims d = A.dims;
    d.viol = 0;
    const int lane = threadIdx.x & 63;
    const unsigned int nbx = (unsigned int)(A.out_w + 63) / 64u, nby = (unsigned int)(A.y1 - A.y0 + K1_TH - 1) / (unsigned int)K1_TH, ntiles = nbx * nby;
    // Every wavefront takes tiles from one of K1_COUNTERS device queues; workgroup b serves queue b % K1_COUNTERS (the same-address atomics of the
    // whole chip are spread over K1_COUNTERS cache lines: one counter for all 8192 waves costs the launch a third more time, profiles/r04_k1).  The
    // next tile's number is requested before this tile's work: the wavefront never waits for the atomic.  A queue holds the tiles n * nq + first in
    // launch order — tiles of every image region, so the queues drain together.  (Dealing compact row bands to the queues an XCD serves, or a band per
    // XCD, or static tiles without counters, all measured slower: the fabric reads follow the wavefronts in flight, not the XCD a tile lands on —
    // profiles/r05_k1/summary.txt; the code: profiles/r06_cleanup/k1_rejected_variants.patch.)
    const unsigned int nq = min((unsigned int)K1_COUNTERS, gridDim.x);  // (a small launch has fewer workgroups than queues: every queue needs a server)
    const unsigned int first = blockIdx.x % nq;
    unsigned int *counter = A.tile_counter + first * 32u;  // 128 bytes apart
    // n-th tile of this workgroup's queue -> (bx, by); false: the queue is exhausted
    const auto k1_tile_of = [&](unsigned int n, unsigned int &bx, unsigned int &by) -> bool {
        const unsigned int tile = n * nq + first;
        by = tile / nbx;
        bx = tile - by * nbx;
        return tile < ntiles;
    };
    unsigned int n = 0;
    if (lane == 0) n = atomicAdd(counter, 1u);
    n = (unsigned int)__builtin_amdgcn_readfirstlane((int)n);
    unsigned int bx, by;
    while (k1_tile_of(n, bx, by)) {
        unsigned int next = 0;
        if (lane == 0) next = atomicAdd(counter, 1u);
        const int x = (int)bx * 64 + lane, y0 = A.y0 + (int)by * K1_TH;
#pragma unroll 1
        for (int r = 0; r < K1_TH; r++) {
            k1_ssgi_march_body<PROJ, ENV, MIS, STAGE>(RFX_KERNARGS_IN_LOOP(A), d, s_cells, x, y0 + r);
            RFX_WAVE_JOIN();  // background / out-of-frame lanes left the body early: the wavefront is whole again here
        }
        n = (unsigned int)__builtin_amdgcn_readfirstlane((int)next);
    }
    rfx_flush_violations(d);
}

// Row-tiled runs, between trace and shade: the rows of the history texture (last frame's composed GI) that the shading of THESE rays will
// fetch — k1_shade reads it NEAREST at a ray's final uv when that uv is on screen and the ray hit (or missed rays are allowed), ssgi.frag:396-427.
// rows[0] / rows[1] take the min / max row over the launch's pixels (preset INT_MAX / -1 by the caller); a superset is harmless, a miss is
// not: the tests hold the bounded gather bit-identical to the whole-frame all-gather.  One wave = 64 pixels of a row: shuffle reduction,
// one atomic pair per wave that can still move the bounds.
__global__ __launch_bounds__(256) void k1_hit_rows(FrameDims d, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, int allow_missed, int *rows) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = y0 + blockIdx.y * 4 + threadIdx.y;
    d.viol = 0;
    float lo = 16777216.0f, hi = -1.0f;  // rows are < 2^23: exact in fp32 (the shuffle moves floats)
    if (x < d.W && y < y1) {
        const float dp = ((const float *)depth.ptr)[rfx_xy_index(d, depth.row0, depth.rows, x, y)];
        if (dp != 1.0f) {  // background fragments return before tracing (their hand-over texels are stale)
            const size_t i = (size_t)rfx_local_row(d, out.row0, out.rows, y) * d.W + x;
            const float4 h0 = hits[2 * i], h1 = hits[2 * i + 1];
            const float u[2] = {h0.x, h0.z}, v[2] = {h0.y, h0.w}, px[2] = {h1.x, h1.y};
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const bool missed = px[r] == 10.0e9f;
                if ((allow_missed || !missed) && u[r] >= 0.0f && u[r] <= 1.0f && v[r] >= 0.0f && v[r] <= 1.0f) {
                    const float row = (float)rfx_nearest_idx(v[r], d.fH, d.H);
                    lo = fminf(lo, row);
                    hi = fmaxf(hi, row);
                }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    if (threadIdx.x == 0 && hi >= 0.0f) {
        if ((int)lo < rows[0]) atomicMin(&rows[0], (int)lo);  // (a stale read only costs a redundant atomic)
        if ((int)hi > rows[1]) atomicMax(&rows[1], (int)hi);
    }
}

// ... and the finer form the bounded gather uses since round 4: ONE 32-bit word per frame row, bit b = some ray of the launch reads column block
// b of that row (32 equal blocks across the frame).  A row whose word is 0 is not read at all — the (min, max) interval above also covers every
// row between two rows that are.  Guarded atomics: a set bit is never set again (a stale read only costs a redundant atomic).
__global__ __launch_bounds__(256) void k1_hit_mask(FrameDims d, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, int allow_missed, unsigned int *mask) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = y0 + blockIdx.y * 4 + threadIdx.y;
    d.viol = 0;
    if (x >= d.W || y >= y1) return;
    const float dp = ((const float *)depth.ptr)[rfx_xy_index(d, depth.row0, depth.rows, x, y)];
    if (dp == 1.0f) return;  // background fragments return before tracing (their hand-over texels are stale)
    const size_t i = (size_t)rfx_local_row(d, out.row0, out.rows, y) * d.W + x;
    const float4 h0 = hits[2 * i], h1 = hits[2 * i + 1];
    const float u[2] = {h0.x, h0.z}, v[2] = {h0.y, h0.w}, px[2] = {h1.x, h1.y};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const bool missed = px[r] == 10.0e9f;
        if ((allow_missed || !missed) && u[r] >= 0.0f && u[r] <= 1.0f && v[r] >= 0.0f && v[r] <= 1.0f) {
            const int row = rfx_nearest_idx(v[r], d.fH, d.H), col = rfx_nearest_idx(u[r], d.fW, d.W);
            const unsigned int bit = 1u << ((unsigned int)(col * 32) / (unsigned int)d.W);
            if (!(mask[row] & bit)) atomicOr(&mask[row], bit);
        }
    }
}

// Pre-pass: view-space Z per texel (getViewZ, ssgi_utils.frag:9: nearMulFar / (farMinusNear * depth - cameraFar), IEEE) and its exact
// (min, max) per BASE x BASE-texel cell.  A streaming kernel: 4 B/px read, 4 B/px written.  Every WAVEFRONT owns a strip of 64 * VEC columns x BASE
// rows — whole cells — and each lane walks its VEC consecutive texels down the BASE rows with all BASE loads in flight (VEC = 4: 16 bytes per
// lane and row, when the row pitch keeps them aligned), so a cell's (min, max) is a register reduction plus log2(BASE / VEC) lane exchanges: no
// LDS, no barrier.  (Rounds 1-4 drew one texel per thread in 64 x 16-thread workgroups with an LDS reduction: 0.109 ms at 4K = 0.6 TB/s, hidden
// under the previous frame's later draws on its own stream but taking its CU time from them.)
template <int VEC>
__global__ __launch_bounds__(256) void k1_prepare(const float *depth, float *viewz, float2 *base, int W, int H, int base_w, float nearMulFar,
                                                   float farMinusNear, float cameraFar, float nearMinusFar, float cameraNear, int perspective) {
    static_assert(VEC == 1 || VEC == 4, "one texel or one aligned float4 per lane and row");
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * VEC, cy = blockIdx.y * 4 + threadIdx.y, y0 = cy * BASE;
    if (y0 >= H) return;  // (wave-uniform: threadIdx.y is the wavefront)
    float mn = INFINITY, mx = -INFINITY;
    float d[BASE][VEC];
    const bool in_x = x0 < W;  // W % VEC == 0: a lane's texels are all inside or all outside
#pragma unroll
    for (int r = 0; r < BASE; r++) {
        const int y = y0 + r;
#pragma unroll
        for (int k = 0; k < VEC; k++) d[r][k] = 1.0f;
        if (in_x && y < H) {
            if (VEC == 4) {
                const float4 t = *(const float4 *)(depth + (size_t)y * W + x0);
                d[r][0] = t.x; d[r][1] = t.y; d[r][2] = t.z; d[r][3] = t.w;
            } else {
                d[r][0] = depth[(size_t)y * W + x0];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < BASE; r++) {
        const int y = y0 + r;
        if (in_x && y < H) {
            float z[VEC];
#pragma unroll
            for (int k = 0; k < VEC; k++) {  // getViewZ ssgi_utils.frag:7-13, both camera variants
                z[k] = perspective ? nearMulFar / (farMinusNear * d[r][k] - cameraFar) : d[r][k] * nearMinusFar - cameraNear;
                mn = fminf(mn, z[k]);
                mx = fmaxf(mx, z[k]);
            }
            if (VEC == 4) *(float4 *)(viewz + (size_t)y * W + x0) = make_float4(z[0], z[1], z[2], z[3]);
            else viewz[(size_t)y * W + x0] = z[0];
        }
    }
    // the BASE / VEC lanes of a cell
#pragma unroll
    for (int o = 1; o < BASE / VEC; o <<= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    const int cx = x0 / BASE;
    if ((threadIdx.x & (BASE / VEC - 1)) == 0 && cx < base_w) base[(size_t)cy * base_w + cx] = make_float2(mn, mx);
}

// ... and the march's table: cell (cx, cy) of edge BASE << up = the (min, max) of its (1 << up)^2 base cells, packed to two halfs
__global__ __launch_bounds__(256) void k1_pack_cells(const float2 *base, int base_w, int base_h, k1_cell_t *cells, int cells_w, int cells_h, int cells_pitch, int up,
                                                     int cells_padded, unsigned int *tile_counter) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 64) tile_counter[i * 32] = 0u;  // the march launch that follows this pre-pass hands its tiles out from 0 (64 counters, 128 bytes apart)
    if (i >= cells_padded) return;
    const int cy = i / cells_pitch, cx = i - cy * cells_pitch;
    if (cy >= cells_h || cx >= cells_w) {  // padding: the columns beyond the last cell of a row, and up to a whole 16-byte vector (the LDS copy moves uint4s)
        cells[i] = 0u;
        return;
    }
    float mn = INFINITY, mx = -INFINITY;
    for (int by = cy << up; by < min((cy + 1) << up, base_h); by++)
        for (int bx = cx << up; bx < min((cx + 1) << up, base_w); bx++) {
            const float2 b = base[(size_t)by * base_w + bx];
            mn = fminf(mn, b.x);
            mx = fmaxf(mx, b.y);
        }
    cells[i] = k1_cell_pack(mn, mx);
}

// scene.environment mip chain: dst texel = bilinear centre of the 2x2 (2x1, 1x2) source block, lerp(.5, lerp(.5,a,b), lerp(.5,c,d)), stored
// in the texture's type.  dw == sw && dh == sh is the level-0 "upload" (type conversion only).
__global__ __launch_bounds__(256) void k1_env_mip(const float4 *src, float4 *dst, int sw, int sh, int dw, int dh, int to_half, int rtz) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    float4 o;
    if (dw == sw && dh == sh) {
        o = src[(size_t)y * sw + x];
    } else {
        const int fx = sw > dw ? 2 : 1, fy = sh > dh ? 2 : 1;
        const int x0 = x * fx, x1 = min(x0 + fx - 1, sw - 1), y0 = y * fy, y1 = min(y0 + fy - 1, sh - 1);
        const float4 a = src[(size_t)y0 * sw + x0], b = src[(size_t)y0 * sw + x1], c = src[(size_t)y1 * sw + x0], e = src[(size_t)y1 * sw + x1];
        o.x = rfx_lerp(0.5f, rfx_lerp(0.5f, a.x, b.x), rfx_lerp(0.5f, c.x, e.x));
        o.y = rfx_lerp(0.5f, rfx_lerp(0.5f, a.y, b.y), rfx_lerp(0.5f, c.y, e.y));
        o.z = rfx_lerp(0.5f, rfx_lerp(0.5f, a.z, b.z), rfx_lerp(0.5f, c.z, e.z));
        o.w = rfx_lerp(0.5f, rfx_lerp(0.5f, a.w, b.w), rfx_lerp(0.5f, c.w, e.w));
    }
    if (to_half) o = rfx_round_half4(o, rtz != 0);
    dst[(size_t)y * dw + x] = o;
}

}  // namespace

hipError_t rfx_launch_env_mip(const float4 *src, float4 *dst, int sw, int sh, int dw, int dh, bool to_half, bool rtz, hipStream_t stream) {
    dim3 block(64, 4), grid((dw + 63) / 64, (dh + 3) / 4);
    hipLaunchKernelGGL(k1_env_mip, grid, block, 0, stream, src, dst, sw, sh, dw, dh, to_half ? 1 : 0, rtz ? 1 : 0);
    return hipGetLastError();
}

int rfx_k1_base_cell() { return BASE; }

hipError_t rfx_launch_k1_hit_rows(const FrameDims &d, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, bool allow_missed, int *rows, hipStream_t stream) {
    dim3 block(64, 4), grid((d.W + 63) / 64, (y1 - y0 + 3) / 4);
    hipLaunchKernelGGL(k1_hit_rows, grid, block, 0, stream, d, y0, y1, depth, out, hits, allow_missed ? 1 : 0, rows);
    return hipGetLastError();
}

hipError_t rfx_launch_k1_hit_mask(const FrameDims &d, int y0, int y1, TexView depth, TexViewW out, const float4 *hits, bool allow_missed, unsigned int *mask, hipStream_t stream) {
    dim3 block(64, 4), grid((d.W + 63) / 64, (y1 - y0 + 3) / 4);
    hipLaunchKernelGGL(k1_hit_mask, grid, block, 0, stream, d, y0, y1, depth, out, hits, allow_missed ? 1 : 0, mask);
    return hipGetLastError();
}

hipError_t rfx_launch_k1_prepare(const K1Args &A, hipStream_t stream) {
    // 16-byte loads when every row starts 16-byte aligned (the planes come from hipMalloc: the pitch decides)
    const int W = A.dims.W, H = A.dims.H;
    const dim3 block(64, 4);
    if (W % 4 == 0 && ((uintptr_t)A.depth.ptr & 15u) == 0 && ((uintptr_t)A.viewz & 15u) == 0)  // (a caller's depth buffer may sit at any address)
        hipLaunchKernelGGL(k1_prepare<4>, dim3((W + 255) / 256, ((H + BASE - 1) / BASE + 3) / 4), block, 0, stream, (const float *)A.depth.ptr, A.viewz, A.coarse, W, H, A.coarse_w,
                           A.nearMulFar, A.farMinusNear, A.p.camera.far_, A.nearMinusFar, A.p.camera.near_, A.p.camera.isPerspective);
    else
        hipLaunchKernelGGL(k1_prepare<1>, dim3((W + 63) / 64, ((H + BASE - 1) / BASE + 3) / 4), block, 0, stream, (const float *)A.depth.ptr, A.viewz, A.coarse, W, H, A.coarse_w,
                           A.nearMulFar, A.farMinusNear, A.p.camera.far_, A.nearMinusFar, A.p.camera.near_, A.p.camera.isPerspective);
    int up = 0;
    while ((BASE << up) < (1 << A.cell_shift)) up++;
    const int padded = A.cells_vec4 * 4;
    hipLaunchKernelGGL(k1_pack_cells, dim3((padded + 255) / 256), dim3(256), 0, stream, (const float2 *)A.coarse, A.coarse_w, A.coarse_h,
                       (k1_cell_t *)A.cells, A.cells_w, A.cells_h, A.cells_pitch, up, padded, A.tile_counter);
    return hipGetLastError();
}

hipError_t rfx_launch_k1(const K1Args &A, int stage, hipStream_t stream) {
    // persistent: what the chip holds at once (4 workgroups of 8 waves per CU at <= 64 VGPRs; fewer fit with an environment map — the
    // surplus workgroups start late and find the counter exhausted), never more workgroups than there are tiles for their waves
    const int nbx = (A.out_w + 63) / 64, nby = (A.y1 - A.y0 + K1_TH - 1) / K1_TH;
    const int want = (nbx * nby + K1_WAVES - 1) / K1_WAVES, n_cu = A.n_cu > 0 ? A.n_cu : 256;
    dim3 block(64 * K1_WAVES), grid(1);
    const float *P = A.p.camera.projectionMatrix;
    const bool persp = P[1] == 0.f && P[2] == 0.f && P[3] == 0.f && P[4] == 0.f && P[6] == 0.f && P[7] == 0.f && P[12] == 0.f && P[13] == 0.f &&
                       P[15] == 0.f && P[11] == -1.f;
    const bool env = A.p.useEnvMap != 0, mis = env && A.p.importanceSampling != 0;
    // the persistent grid: what the chip holds of THIS specialisation at once — the runtime's occupancy figure for its registers and static LDS
    // (4 workgroups per CU at <= 64 VGPRs; fewer with an environment map), asked once per specialisation and device
#define K1_GO_S(P, E, M, S)                                                                                                   \
    do {                                                                                                                      \
        static int per_cu[64] = {0};                                                                                          \
        int dev = 0;                                                                                                          \
        hipGetDevice(&dev);                                                                                                   \
        int nb = (dev >= 0 && dev < 64) ? per_cu[dev] : 0;                                                                    \
        if (nb <= 0) {                                                                                                        \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k1_ssgi_march<P, E, M, S>, 64 * K1_WAVES, 0) != hipSuccess || nb <= 0) \
                nb = 32 / K1_WAVES;                                                                                           \
            if (dev >= 0 && dev < 64) per_cu[dev] = nb;                                                                       \
        }                                                                                                                     \
        const int fit = n_cu * (nb > 0 ? nb : 1);                                                                             \
        grid = dim3(want < fit ? want : fit);                                                                                 \
        hipLaunchKernelGGL((k1_ssgi_march<P, E, M, S>), grid, block, 0, stream, A);                                           \
    } while (0)
#define K1_GO(P, E, M)                            \
    do {                                          \
        if (stage == 0) K1_GO_S(P, E, M, 0);      \
        else if (stage == 1) K1_GO_S(P, E, M, 1); \
        else K1_GO_S(P, E, M, 2);                 \
    } while (0)
    const bool centred = persp && P[8] == 0.f && P[9] == 0.f;
#define K1_GO_P(PJ) do { if (mis) K1_GO(PJ, true, true); else if (env) K1_GO(PJ, true, false); else K1_GO(PJ, false, false); } while (0)
    // (the table's layout is a template argument too: PROJ_TABLE_POW2)
#define K1_GO_T(PJ) do { if (A.cells_pow2) K1_GO_P((PJ) | PROJ_TABLE_POW2); else K1_GO_P(PJ); } while (0)
    if (centred) K1_GO_T(PROJ_CENTRED);
    else if (persp) K1_GO_T(PROJ_PERSPECTIVE);
    else K1_GO_T(PROJ_GENERAL);
#undef K1_GO_T
#undef K1_GO_P
#undef K1_GO
#undef K1_GO_S
    return hipGetLastError();
}
