/*
 * rfx_oracle.c — CPU restatement of the reference's SSGI hot path (K1..K4).
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this.  The product (librfx_hip.so and the host packages) never links, imports or
 * falls back to it.
 *
 * What it is: a scalar, per-fragment restatement in plain C of the arithmetic in the
 * reference's fragment shaders, written from the GLSL (each function cites the file:line it
 * follows), with the execution semantics of the GL implementation the golden vectors were
 * produced on (Mesa llvmpipe):
 *   - fine 2x2-quad derivatives aligned to even pixels (dFdx/dFdy/fwidth),
 *   - nearest CLAMP_TO_EDGE fetch = clamp(cvttss2si(u*W), 0, W-1) (NaN/out-of-int-range -> 0),
 *   - bilinear fetch of the RGBA16F targets as llvmpipe computes it,
 *   - packHalf2x16 = round-to-nearest-even; RGBA16F colour-buffer store = round-toward-zero
 *     saturating at 65504,
 *   - min/max are NaN-suppressing, uninitialised locals are zero (WebGL),
 *   - `discard` leaves the render target's previous contents.
 * Parity pin: the reference has no tests / golden vectors of its own; this
 * restatement is pinned against the reference GLSL executed on llvmpipe (oracle/glref),
 * see tests/golden/ and tests/test_oracle_vs_glref.py.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, OpenMP over rows)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "../include/rfx.h"

/* ------------------------------------------------------------------ perturbed primitives (parity tests only)
 * Second, independent proof of a flip (the first: the discontinuity margins below).  The only freedom two correct
 * implementations have is the last bits of exp/log/pow/sqrt/sin/cos/atan: the reference GL's are accurate to <= 6e-6
 * relative (measured, oracle/glref/probes/probe_transcendentals.py), the GPU's hardware forms to ~1 ulp.  With a
 * perturbation armed (rfxo_set_perturbation) every such result is multiplied by (1 +- rel) — and sin/cos moved by
 * +- abs — with a sign drawn per call from a per-fragment seeded generator.  A fragment whose output moves by more
 * than the tolerance under such perturbations is UNSTABLE: it sits on a branch / nearest-texel boundary or on an
 * ill-conditioned expression (e.g. SampleGGXVNDF's sqrt(1 - t1^2 - t2^2) at blueNoise.x = 1), and two correct
 * implementations may disagree on it.  A fragment that is stable may not.  (tests/parity.py, tests/stagewise.py) */
static float g_pert_rel = 0.0f, g_pert_abs = 0.0f; /* rel: exp/log/pow class; sqrt and the angle functions take 2 ulps relative (+ abs) */
static uint32_t g_pert_seed = 0; /* 0 = off */
static _Thread_local uint32_t g_pert_state = 0;
void rfxo_set_perturbation(uint32_t seed, float rel, float abs_) { g_pert_seed = seed; g_pert_rel = rel; g_pert_abs = abs_; }
static inline void pert_begin(int x, int y) {
    if (!g_pert_seed) { g_pert_state = 0; return; }
    uint32_t h = g_pert_seed * 0x9E3779B1u ^ ((uint32_t)x * 0x85EBCA77u) ^ ((uint32_t)y * 0xC2B2AE3Du);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    g_pert_state = h | 1u;
}
static inline float pert_sign(void) {
    g_pert_state = g_pert_state * 1664525u + 1013904223u;
    return (g_pert_state & 0x00800000u) ? 1.0f : -1.0f;
}
static inline float pert_rel(float v) { return (g_pert_seed && g_pert_state) ? v * (1.0f + pert_sign() * g_pert_rel) : v; }
static inline float pert_ang(float v) { return (g_pert_seed && g_pert_state) ? v * (1.0f + pert_sign() * 2.4e-7f) + pert_sign() * g_pert_abs : v; }
/* the fragment's vUv: a rasteriser interpolates the varying from plane equations in fp32.  Measured on the reference GL at 1080p / 4K / 8K
 * (oracle/glref/probes/probe_varying.py): vUv is exact on only 23-74 % of the positions and off by up to 2^-24 ABSOLUTE (one ulp of the
 * [0.5, 1) binade, at any u) = 1.1e-4 / 2.3e-4 / 4.6e-4 texel.  Every fetch at a vUv-derived coordinate inherits it: a NEAREST tap that
 * close to a texel boundary flips (K3's rotated taps: 8 taps x 2 axes -> ~0.2 % of the pixels, the measured pass-0 flip rate), a LINEAR
 * fetch turns it into a weight error (K4, K3's later passes, K2's history). */
/* ... plus what the sampler adds on its own: one rounding of u * size (half an ulp of the texel coordinate, 2.4e-4 texel at 8K).  Together:
 * 2^-23.  At 8K that is ~1e-3 texel: a LINEAR fetch of a texture whose neighbouring texels differ by ~1 (K2's history AGE channel from the
 * third frame on, when ages exceed 1) moves by ~1e-3 — the absolute tolerance sits at the resolution of an fp32 texture coordinate there. */
#define UV_ABS_ERR 1.1920929e-7f
/* ... unless the fragments see the reference GL's own vUv (rfxo_set_uv_model(1), frag_u / frag_v below): then nothing is left to bound */
static int g_uv_model = 1; /* the default on both sides of every parity test since round 3 (rfx_ctx.h uv_model) */
static inline float uv_err(void) { return g_uv_model ? 0.0f : UV_ABS_ERR; }
static inline float pert_uv(float v) { return (g_pert_seed && g_pert_state) ? v + pert_sign() * uv_err() : v; }
/* Which vUv a fragment sees.  Model 0: (i + 0.5) / n, correctly rounded — what the HIP kernels compute.  Model 1: the reference GL's own
 * value, bit for bit (oracle/glref/probes/probe_varying.py: exact on every fragment of every size tried, 55x97 ... 7680x4320).  three's
 * full-screen triangle (-1,-1) (3,-1) (-1,3) leaves the guard band (|x| <= 2w), so Mesa's draw module clips it to the viewport: two
 * triangles split along the diagonal (0,0)-(W,H), each with its own plane equations (lp_setup_coef: a0 = a(v0) - (dadx * x0c + dady * y0c),
 * dadx = H * (1 / (W * H)), every product rounded) and provoking vertices (0,H) above the diagonal and (W,H) below it; the fragment then
 * evaluates fma(dady, y, fma(dadx, x, a0)) on the integer pixel position (lp_bld_interp, pixel offset folded into a0).  The first has
 * a0 = dadx / 2 for u, the second 1 - dadx * (W - 0.5); v's a0 = 1 - dady * (H - 0.5) in both.  A fragment centre on the diagonal belongs
 * to the lower triangle. */
void rfxo_set_uv_model(int m) { g_uv_model = m; }
int rfxo_get_uv_model(void) { return g_uv_model; }
/* K3's tap rotation: 0 (default) = the correctly rounded (sin, cos) of the 256 possible angles — what the kernel's table holds; 1 = libm
 * sinf / cosf of the fp32 angle, the restatement's form before round 3.  Kept so that a test can bound what the choice moves. */
static int g_k3_rotation_libm = 0;
void rfxo_set_k3_rotation_libm(int on) { g_k3_rotation_libm = on; }
static inline float frag_u(int x, int y, int W, int H) {
    if (!g_uv_model) return ((float)x + 0.5f) / (float)W;
    float ooa = 1.0f / ((float)W * (float)H), dudx = (float)H * ooa;
    int upper = (int64_t)(2 * y + 1) * W > (int64_t)(2 * x + 1) * H;
    float far_side = dudx * ((float)W - 0.5f);
    float a0 = upper ? 0.5f * dudx : 1.0f - far_side;
    return fmaf(dudx, (float)x, a0);
}
static inline float frag_v(int y, int W, int H) {
    if (!g_uv_model) return ((float)y + 0.5f) / (float)H;
    float ooa = 1.0f / ((float)W * (float)H), dvdy = (float)W * ooa;
    float far_side = dvdy * ((float)H - 0.5f);
    return fmaf(dvdy, (float)y, 1.0f - far_side);
}
/* a coordinate FORMED in uv space by a chain of fp32 operations (a projected point: normalize, matrix products, a division, * 0.5 + 0.5)
 * carries a few 2^-24 ABSOLUTE between two correct implementations (operation order, fused or not): the perturbed runs move it by `a` */
static inline float pert_coord(float v, float a) { return (g_pert_seed && g_pert_state) ? v + pert_sign() * a : v; }
static inline float pert_sqrt(float v) { return (g_pert_seed && g_pert_state) ? v * (1.0f + pert_sign() * 2.4e-7f) : v; }
/* test probe: the (u, v) planes of a W x H target under a model, interleaved */
void rfxo_frag_uv(int model, int W, int H, float *out) {
    int keep = g_uv_model;
    g_uv_model = model;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) { out[((size_t)y * W + x) * 2] = frag_u(x, y, W, H); out[((size_t)y * W + x) * 2 + 1] = frag_v(y, W, H); }
    g_uv_model = keep;
}
/* The reference GL's OWN exp, sin and cos, bit for bit (diagnostic mode, rfxo_set_gl_exp(1); default off).  llvmpipe (Mesa 23.2 gallivm, lp_bld_arit.c — a
 * dependency of the reference's execution here that is not under /root/reference; its published algorithm, restated): exp2(x) clamps x to
 * [-126.99999, 128], splits it into floor and fraction, builds 2^floor from exponent bits and 2^fraction from a degree-5 polynomial evaluated as
 * even(f^2) + f * odd(f^2) with fused multiply-adds; exp(x) = exp2(x * log2 e).  oracle/glref/probes/probe_exp_restatement.py holds the restatement
 * equal to the GL on 65 536 inputs, and shows how the march's `cs = 1. - exp(-0.25 * pow(t, 2.))` (ssgi.frag:453-454) reaches it after the GLSL
 * compiler's algebraic pass: exp2(RN(t * RN(t * c))), c = -0.25 * log2 e — equal on 6 x 4096 inputs.  The GL's cs is up to 2.6e-6 (relative) from the
 * true value at the first step; the product and the default oracle use the true exp (the GLSL's meaning), this mode is for root-causing a pixel. */
static int g_gl_exp = 0;
void rfxo_set_gl_exp(int on) { g_gl_exp = on; }
static inline float gl_exp2(float x) {
    x = fminf(128.0f, x);
    x = fmaxf(-126.99999f, x);
    const float ip = floorf(x), fp = x - ip, x2 = fp * fp;
    const float even = fmaf(x2, fmaf(x2, 0.00898934009049466391101f, 0.240153617044375388211f), 1.0f);
    const float odd = fmaf(x2, fmaf(x2, 0.00187757667519147912699f, 0.0558263180532956664775f), 0.693153073200168932794f);
    const float poly = fmaf(odd, fp, even);
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)((int32_t)ip + 127) << 23;
    return sc.f * poly;
}
/* 1. - exp(-0.25 * pow(t, 2.)) as the reference GL evaluates it */
static inline float gl_march_cs(float t) {
    const float c = -0.25f * 1.4426950408889634074f;
    return 1.0f - gl_exp2(t * (t * c));
}
/* ... and its sin / cos (lp_build_sin_or_cos: the Cephes / sse_mathfun single-precision sincos — |x| * 4/pi truncated to the next even octant, the
 * three-constant "extended precision" argument reduction with fused multiply-adds, one of two degree-3 polynomials in z = x^2 selected by the octant, the
 * sign from the octant and the argument): the restatement equals the GL on 8 x 4096 inputs over [-6.3, 6.3] (tools/open_pixel.py's header has the probe). */
static inline float gl_sincos(float a, int want_cos) {
    union { float f; uint32_t u; int32_t i; } av, r;
    av.f = a;
    const float x = fabsf(a);
    const float y = x * 1.27323954473516f;
    const int32_t j = (int32_t)y, jadd = j + 1, jand = jadd & ~1;
    const float y2 = (float)jand;
    const int32_t e2 = want_cos ? jand - 2 : jand;
    const uint32_t sign = want_cos ? ((uint32_t)(4 & ~e2) << 29) : ((av.u ^ ((uint32_t)jadd << 29)) & 0x80000000u);
    const float x1 = fmaf(y2, -0.78515625f, x), x2 = fmaf(y2, -2.4187564849853515625e-4f, x1), x3 = fmaf(y2, -3.77489497744594108e-8f, x2);
    const float z = x3 * x3;
    float res;
    if ((e2 & 2) == 0) {
        const float q = fmaf(fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f), z, -1.6666654611E-1f) * z;
        res = fmaf(q, x3, x3);
    } else {
        const float c = (fmaf(fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f), z, 4.166664568298827E-002f) * z) * z;
        res = (c - z * 0.5f) + 1.0f;
    }
    r.f = res;
    r.u ^= sign;
    return r.f;
}
/* function-like macros are not re-expanded inside their own expansion: (expf)(x) is libm's */
#define expf(x) pert_rel((expf)(x))
#define logf(x) pert_rel((logf)(x))
#define powf(x, y) pert_rel((powf)(x, y))
#define sqrtf(x) pert_sqrt((sqrtf)(x))
#define exp2f(x) pert_rel((exp2f)(x))
#define log2f(x) pert_rel((log2f)(x))
#define sinf(x) (g_gl_exp ? gl_sincos((x), 0) : pert_ang((sinf)(x)))
#define cosf(x) (g_gl_exp ? gl_sincos((x), 1) : pert_ang((cosf)(x)))
#define atan2f(y, x) pert_ang((atan2f)(y, x))

/* ------------------------------------------------------------------ small vector helpers */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* GLSL normalize: v * inversesqrt(dot(v,v)); llvmpipe's rsq is 1/sqrt (full precision) */
static inline v3 normalize3(v3 a) { float s = 1.0f / sqrtf(dot3(a, a)); return mul3(a, s); }
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline v3 mix3(v3 x, v3 y, float a) { return V3(mixf(x.x, y.x, a), mixf(x.y, y.y, a), mixf(x.z, y.z, a)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float lum(v3 c) { return 0.2125f * c.x + 0.7154f * c.y + 0.0721f * c.z; } /* dot(vec3(0.2125,0.7154,0.0721), c) */

/* ------------------------------------------------------------------ discontinuity margins (parity tests only)
 * The path branches on comparisons and addresses NEAREST texels: two correct implementations whose exp/log/sin differ in the
 * last ulp take different sides of such a decision in a few pixels and differ there by O(1).  To PROVE that an out-of-tolerance
 * pixel is such a flip — instead of assuming it — every decision a fragment takes records how close it was, normalised by the
 * distance an ulp-level perturbation of its operands can move it (`scale`): margin = gap / scale.  The minimum over the
 * fragment's decisions goes to an optional per-pixel plane (rfxo_set_margin_plane).  margin < 1: the fragment sits on a
 * discontinuity and may legitimately flip; margin >= 1: it may not, a mismatch there is a bug.  (tests/parity.py) */
static _Thread_local float g_margin = 3.0e38f;
static float *g_margin_plane = NULL; /* W*H floats of the stage being run, or NULL */
void rfxo_set_margin_plane(float *plane) { g_margin_plane = plane; }
/* optional W*H byte mask: fragments whose byte is 0 are SKIPPED by every stage driver (their output texels keep what the caller put
 * there).  Lets the parity tests re-evaluate — with margins, perturbed — only the out-of-tolerance pixels and a random sample instead of
 * whole 4K / 8K frames. */
static const uint8_t *g_pixel_mask = NULL;
void rfxo_set_pixel_mask(const uint8_t *mask) { g_pixel_mask = mask; }
/* ... and the same minimum WITHOUT the texel-boundary reach of the march / refine taps (margin_tap below): a second optional plane, so that the tests can
 * count the pixels whose only proof is that reach (the widened predicate is tracked, not just used) */
static _Thread_local float g_margin_notap = 3.0e38f;
static float *g_margin_notap_plane = NULL;
void rfxo_set_margin_notap_plane(float *plane) { g_margin_notap_plane = plane; }
static inline void margin_note(float m) { if (m < g_margin) g_margin = m; if (m < g_margin_notap) g_margin_notap = m; }
static inline void margin_note_tap(float m) { if (m < g_margin) g_margin = m; }
/* decision `a ? b` between two computed quantities; rel = relative perturbation either side can carry */
static inline void margin_cmp(float a, float b, float rel) {
    float s = rel * fmaxf(fmaxf(fabsf(a), fabsf(b)), 1e-30f);
    margin_note(fabsf(a - b) / s);
}
/* scales (relative error an operand of the decision can carry between two implementations with ulp-accurate primitives) */
#define MARGIN_REL_MARCH 1e-6f   /* ray position after <= 40 accumulated steps of dir * (1 - exp(..)), projected */
#define MARGIN_REL_SHORT 4e-6f   /* a handful of fp32 operations incl. one transcendental */
#define MARGIN_REL_WEIGHT 1e-4f  /* products of exp(-phi * diff): the exponents reach ~10 and carry their own rounding */
#define MARGIN_REL_CURVATURE 2e-5f /* length(fwidth(normal)) against 0.05: a sum of |differences of unit-vector components| — each component carries
                                   * ~1e-7 ABSOLUTE (oct decode + normalize), six of them against a threshold of 0.05: ~1e-5 relative, not a few ulps */

static _Thread_local float g_unc_dir = 4e-7f; /* absolute uncertainty of the direction the next calc_angles() receives (set by the sampler) */

/* column-major mat4: M[c*4+r].  M * vec4(p, w): ((M0*x + M1*y) + M2*z) + M3*w */
static inline v4 mat_mul_v4(const float *M, float x, float y, float z, float w) {
    v4 r;
    r.x = ((M[0] * x + M[4] * y) + M[8] * z) + M[12] * w;
    r.y = ((M[1] * x + M[5] * y) + M[9] * z) + M[13] * w;
    r.z = ((M[2] * x + M[6] * y) + M[10] * z) + M[14] * w;
    r.w = ((M[3] * x + M[7] * y) + M[11] * z) + M[15] * w;
    return r;
}
/* vec4(v, w) * M  ->  (dot(v4, M[0]), dot(v4, M[1]), dot(v4, M[2])) */
static inline v3 v4_mul_mat_xyz(const float *M, v3 v, float w) {
    v3 r;
    r.x = ((v.x * M[0] + v.y * M[1]) + v.z * M[2]) + w * M[3];
    r.y = ((v.x * M[4] + v.y * M[5]) + v.z * M[6]) + w * M[7];
    r.z = ((v.x * M[8] + v.y * M[9]) + v.z * M[10]) + w * M[11];
    return r;
}

/* ------------------------------------------------------------------ half floats */
static inline float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { /* subnormal */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            u = s | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
/* packHalf2x16 conversion: round-to-nearest-even, overflow -> inf (Appendix C-2) */
static inline uint16_t float_to_half_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0));
    if (a >= 0x477ff000u) return (uint16_t)(s | 0x7c00u); /* rounds to >= 65520 -> inf */
    if (a < 0x38800000u) { /* half subnormal or zero */
        if (a < 0x33000000u) return (uint16_t)s; /* < 2^-25 -> 0 */
        int e = (int)(a >> 23);
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        int shift = 126 - e; /* 14..24 */
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(s | r);
    }
    uint32_t r = (a - 0x38000000u) >> 13, rem = a & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(s | r);
}
/* RGBA16F render-target store on llvmpipe: vcvtps2ph with imm=3 (truncate); finite overflow
 * saturates at 65504 (Appendix C-3) */
static inline uint16_t float_to_half_rtz(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(s | 0x7c00u | ((a > 0x7f800000u) ? 0x200u : 0));
    if (a >= 0x47800000u) return (uint16_t)(s | 0x7bffu);
    if (a < 0x38800000u) {
        if (a < 0x33800000u) return (uint16_t)s; /* < 2^-24 -> 0 */
        int e = (int)(a >> 23);
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        return (uint16_t)(s | (m >> (126 - e)));
    }
    return (uint16_t)(s | ((a - 0x38000000u) >> 13));
}

/* ------------------------------------------------------------------ texture fetch */
/* nearest CLAMP_TO_EDGE index: clamp(cvttss2si(u*size), 0, size-1); cvttss2si returns INT_MIN
 * for NaN and anything outside int range (Appendix C-4) */
/* texel-boundary margin of a nearest fetch at coordinate c (texels): the coordinate carries a few ulps of its own magnitude
 * plus whatever its inputs carry (rel_in, relative to the OFFSET that was added to a pixel centre, passed in texels) */
static _Thread_local float g_fetch_rel = 0.0f, g_fetch_abs = 0.0f; /* what the coordinate's INPUTS carry (set by the caller that knows) */
/* ... and what a coordinate that was FORMED in uv space carries whatever its size in texels: u = ndc * 0.5 + 0.5 (a projected point) or
 * u = vUv - velocity is rounded at the magnitude of 1, i.e. a few 2^-24 ABSOLUTE in uv = g_fetch_uv * size texels — near the left / bottom
 * edge (small c) far more than any relative model of c allows.  (While the fragments' vUv itself is uncertain, uv_err() below carries the
 * same kind of term; under the reference vUv model it is zero and this one is what remains.) */
static _Thread_local float g_fetch_uv = 0.0f;
static inline void margin_texel(float c, int size) {
    if (!(c > 0.0f && c < (float)size)) return; /* clamped region: flat */
    float fl = floorf(c), fr = c - fl;
    float dlo = fl >= 1.0f ? fr : 3.0e38f;                      /* the boundary at fl exists unless it is the clamp at 0 */
    float dhi = fl <= (float)(size - 2) ? 1.0f - fr : 3.0e38f;  /* the boundary at fl + 1 exists unless it is the clamp at size */
    float scale = (4.0f * 1.1920929e-7f + g_fetch_rel) * fabsf(c) + g_fetch_abs + (uv_err() + g_fetch_uv) * (float)size; /* the rasteriser's vUv error in texels scales with the texture size */
    margin_note(fminf(dlo, dhi) / fmaxf(scale, 1e-30f));
}
static inline int nearest_idx(float u, int size) {
    float c = u * (float)size;
    margin_texel(c, size);
    int i;
    if (!(c > -2147483904.0f && c < 2147483648.0f)) i = INT_MIN;
    else i = (int)c;
    if (i < 0) i = 0;
    if (i > size - 1) i = size - 1;
    return i;
}
typedef struct { int W, H; } dims;

static inline float fetch_r32f(const float *t, dims d, float u, float v) {
    return t[(size_t)nearest_idx(v, d.H) * d.W + nearest_idx(u, d.W)];
}
static inline const uint32_t *fetch_u4(const uint32_t *t, dims d, float u, float v) {
    return t + 4 * ((size_t)nearest_idx(v, d.H) * d.W + nearest_idx(u, d.W));
}
static inline v4 fetch_f4(const float *t, dims d, float u, float v) {
    const float *p = t + 4 * ((size_t)nearest_idx(v, d.H) * d.W + nearest_idx(u, d.W));
    v4 r = {p[0], p[1], p[2], p[3]}; return r;
}
/* llvmpipe bilinear, CLAMP_TO_EDGE, normalised coords (lp_bld_sample_soa.c):
 * c = min(u*size, size) - 0.5; c = max(c, 0); i0 = floor(c); w = fract(c); i1 = min(i0+1, size-1)
 * texel = lerp(wy, lerp(wx, t00, t10), lerp(wx, t01, t11)), lerp(w,a,b) = a + w*(b-a) */
static inline void linear_coord(float u, int size, int *i0, int *i1, float *w) {
    float c = u * (float)size;
    c = fminf(c, (float)size); /* NaN -> size */
    c = c - 0.5f;
    c = fmaxf(c, 0.0f);
    float fl = floorf(c);
    *i0 = (int)fl;
    *w = c - fl;
    *i1 = *i0 + 1 > size - 1 ? size - 1 : *i0 + 1;
}
static inline float lerpf(float w, float a, float b) { return a + w * (b - a); }
static inline v4 fetch_h4_linear(const uint16_t *t, dims d, float u, float v) {
    int x0, x1, y0, y1; float wx, wy;
    linear_coord(u, d.W, &x0, &x1, &wx);
    linear_coord(v, d.H, &y0, &y1, &wy);
    const uint16_t *p00 = t + 4 * ((size_t)y0 * d.W + x0), *p10 = t + 4 * ((size_t)y0 * d.W + x1);
    const uint16_t *p01 = t + 4 * ((size_t)y1 * d.W + x0), *p11 = t + 4 * ((size_t)y1 * d.W + x1);
    float o[4];
    for (int c = 0; c < 4; c++) {
        /* the GL's lerp is one fused multiply-add (measured on llvmpipe: with it the later K3 passes' targets are bit-identical to the
         * reference on 99.9 % of the texels instead of 98.6 %; the kernels that fetch RGBA16F bilinearly in bulk, K3 / K4, contract the
         * same way).  fetch_f4_linear below — K1's environment taps, the FloatType framebuffer copy — stays unfused like the kernel. */
        float a = fmaf(wx, half_to_float(p10[c]) - half_to_float(p00[c]), half_to_float(p00[c]));
        float b = fmaf(wx, half_to_float(p11[c]) - half_to_float(p01[c]), half_to_float(p01[c]));
        o[c] = fmaf(wy, b - a, a);
    }
    v4 r = {o[0], o[1], o[2], o[3]}; return r;
}

/* the same sampler over an RGBA32F texture (the FloatType framebuffer copy, TemporalReprojectPass.js:137-142) */
static inline v4 fetch_f4_linear(const float *t, dims d, float u, float v) {
    int x0, x1, y0, y1; float wx, wy;
    linear_coord(u, d.W, &x0, &x1, &wx);
    linear_coord(v, d.H, &y0, &y1, &wy);
    const float *p00 = t + 4 * ((size_t)y0 * d.W + x0), *p10 = t + 4 * ((size_t)y0 * d.W + x1);
    const float *p01 = t + 4 * ((size_t)y1 * d.W + x0), *p11 = t + 4 * ((size_t)y1 * d.W + x1);
    float o[4];
    for (int c = 0; c < 4; c++) o[c] = lerpf(wy, lerpf(wx, p00[c], p10[c]), lerpf(wx, p01[c], p11[c]));
    v4 r = {o[0], o[1], o[2], o[3]}; return r;
}

/* ------------------------------------------------------------------ codec (gbuffer_packing.glsl) */
typedef struct { v3 diffuse; float alpha; v3 normal; float roughness, metalness; v3 emissive; } material;

static inline void unpack_half2(uint32_t u, float *a, float *b) { *a = half_to_float((uint16_t)(u & 0xffffu)); *b = half_to_float((uint16_t)(u >> 16)); }
static inline uint32_t pack_half2(float a, float b) { return (uint32_t)float_to_half_rne(a) | ((uint32_t)float_to_half_rne(b) << 16); }

/* decodeOctWrap + unpackNormal, gbuffer_packing.glsl:52-63 */
static inline v3 unpack_normal(uint32_t bits) {
    float fx, fy; unpack_half2(bits, &fx, &fy);
    fx = fx * 2.0f - 1.0f; fy = fy * 2.0f - 1.0f;
    v3 n = V3(fx, fy, 1.0f - fabsf(fx) - fabsf(fy));
    float t = fmaxf(-n.z, 0.0f);
    n.x += n.x >= 0.0f ? -t : t;
    n.y += n.y >= 0.0f ? -t : t;
    return normalize3(n);
}
/* floatToVec4, gbuffer_packing.glsl:151-164 */
static inline v4 float_to_vec4(uint32_t value) {
    v4 v;
    v.x = fmaxf((float)(value & 0xffu) / 255.0f - 0.0001f, 0.0f);
    v.y = fmaxf((float)((value >> 8) & 0xffu) / 255.0f - 0.0001f, 0.0f);
    v.z = fmaxf((float)((value >> 16) & 0xffu) / 255.0f - 0.0001f, 0.0f);
    v.w = fmaxf((float)((value >> 24) & 0xffu) / 255.0f - 0.0001f, 0.0f);
    return v;
}
static inline float glsl_mod(float x, float y) { return x - y * floorf(x / y); }
/* getMaterial, gbuffer_packing.glsl:181-196 */
static inline material get_material(const uint32_t *g) {
    material m;
    v4 d = float_to_vec4(g[0]);
    m.diffuse = V3(d.x, d.y, d.z); m.alpha = d.w;
    m.normal = unpack_normal(g[1]);
    float value; memcpy(&value, &g[2], 4);
    /* float2color :24-34 : r = mod(v,257)/256, g = floor(v/(257*257))/256 ; -1e-4 ; max 0 */
    const float p1 = 257.0f;
    float cr = glsl_mod(value, p1) / 256.0f;
    float cg = floorf(value / (p1 * p1)) / 256.0f;
    m.roughness = fmaxf(cr - 0.0001f, 0.0f);
    m.metalness = fmaxf(cg - 0.0001f, 0.0f);
    v4 e = float_to_vec4(g[3]);
    float fexp = e.w * 255.0f - 128.0f; /* decodeRGBE8 :136-141 */
    float sc = exp2f(fexp);
    m.emissive = V3(e.x * sc, e.y * sc, e.z * sc);
    return m;
}
/* packTwoVec4 :65-83 */
static inline void pack_two_vec4(v4 a, v4 b, uint32_t *out) {
    const float o = 0.0001f;
    out[0] = pack_half2(a.x + o, a.y + o);
    out[1] = pack_half2(a.z + o, a.w + o);
    out[2] = pack_half2(b.x + o, b.y + o);
    out[3] = pack_half2(b.z + o, b.w + o);
}
/* unpackTwoVec4 :85-98 */
static inline void unpack_two_vec4(const uint32_t *e, v4 *a, v4 *b) {
    const float o = 0.0001f;
    unpack_half2(e[0], &a->x, &a->y); unpack_half2(e[1], &a->z, &a->w);
    unpack_half2(e[2], &b->x, &b->y); unpack_half2(e[3], &b->z, &b->w);
    a->x -= o; a->y -= o; a->z -= o; a->w -= o;
    b->x -= o; b->y -= o; b->z -= o; b->w -= o;
}

/* ------------------------------------------------------------------ blue noise (blue_noise.glsl) */
static inline void pcg4d(uint32_t *v) { /* :17-28 */
    for (int i = 0; i < 4; i++) v[i] = v[i] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int i = 0; i < 4; i++) v[i] ^= v[i] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
}
/* blueNoise() :37-47 with index != 0; px,py = ivec2(vUv * resolution).
 * index == 0 takes the textureLod(uv*resolution/128) path (repeat wrap, nearest). */
static inline v4 blue_noise(const uint8_t *table, int px, int py, int index, float u, float v, dims d) {
    int sx, sy;
    if (index == 0) {
        float cu = u * (float)d.W / 128.0f, cv = v * (float)d.H / 128.0f;
        /* nearest REPEAT: ifloor(coord*size) & (size-1) */
        sx = ((int)floorf(cu * 128.0f)) & 127; sy = ((int)floorf(cv * 128.0f)) & 127;
    } else {
        uint32_t i = (uint32_t)index;
        uint32_t s[4] = {i, i * 15843u, i * 31u + 4566u, i * 2345u + 58585u};
        pcg4d(s);
        sx = (px + (int)(s[0] % 0x0fffffffu)) % 128;
        sy = (py + (int)(s[1] % 0x0fffffffu)) % 128;
    }
    const uint8_t *t = table + 4 * (sy * 128 + sx);
    /* llvmpipe unorm8 -> float: float(byte) * (1.0/255.0) */
    const float k = (float)(1.0 / 255.0);
    v4 r = {t[0] * k, t[1] * k, t[2] * k, t[3] * k};
    return r;
}

/* ------------------------------------------------------------------ quad derivatives */
/* fine derivatives on 2x2 quads aligned to even pixels (Appendix C-1):
 * dFdx(f)(x,y) = f(x|1, y) - f(x&~1, y);  dFdy(f)(x,y) = f(x, y|1) - f(x, y&~1).
 * The partner fragment may lie outside the target (odd sizes): it still executes, its nearest
 * CLAMP_TO_EDGE fetch lands on the edge texel. */
static inline int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

/* ==================================================================== K1: ssgi.frag */
#define M_PIf 3.1415926535897932384626433832795f

static inline float pow5(float x) { return powf(x, 5.0f); }
/* F_Schlick(vec3 f0, theta) ssgi_utils.frag:108 */
static inline v3 f_schlick3(v3 f0, float theta) {
    float p = pow5(1.0f - theta);
    return V3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
}
static inline float f_schlick1(float f0, float f90, float theta) { return f0 + (f90 - f0) * pow5(1.0f - theta); } /* :110 */
static inline float d_gtr2(float roughness, float NoH) { /* D_GTR(roughness, NoH, 2.) :112-115 */
    float a2 = roughness * roughness; /* pow(x, 2.) -> x*x */
    float t = (NoH * NoH) * (a2 * a2 - 1.0f) + 1.0f;
    return a2 / (M_PIf * (t * t));
}
static inline float smith_g(float NDotV, float alphaG) { /* :117-121 */
    float a = alphaG * alphaG, b = NDotV * NDotV;
    return (2.0f * NDotV) / (NDotV + sqrtf(a + b - a * b));
}
static inline float ggx_vndf_pdf(float NoH, float NoV, float roughness) { /* :123-127 */
    float D = d_gtr2(roughness, NoH);
    float G1 = smith_g(NoV, roughness * roughness);
    return (D * G1) / fmaxf(0.00001f, 4.0f * NoV);
}
static inline float eval_disney_diffuse(float NoL, float NoV, float LoH, float roughness, float metalness) { /* :136-142 */
    float FD90 = 0.5f + 2.0f * roughness * (LoH * LoH);
    float a = f_schlick1(1.0f, FD90, NoL), b = f_schlick1(1.0f, FD90, NoV);
    return (a * b / M_PIf) * (1.0f - metalness);
}
static inline float eval_disney_specular(float roughness, float NoH, float NoV, float NoL) { /* :144-151 */
    float D = d_gtr2(roughness, NoH);
    float r2 = 0.5f + roughness * 0.5f; r2 = r2 * r2;
    float a2 = r2 * r2; /* GeometryTerm: a2 = roughness*roughness, SmithG(.., a2) */
    float G = smith_g(NoV, a2) * smith_g(NoL, a2);
    return D * G / (4.0f * NoL * NoV);
}
/* SampleGGXVNDF :153-170 */
/* diagnostic (tools/open_pixel_trace.py): the march of ONE pixel's specular ray, value by value.  rfxo_set_trace(x, y, buf, n): while pixel (x, y) is
 * evaluated, every TRACE(tag, a, b, c, d) appends (tag, a, b, c, d) to buf (n floats); tags: 0 viewPos+viewZ, 1 specular ray + random.b, 2 final hit position,
 * 3 view normal + roughness^2, 20-25 inside SampleGGXVNDF (random.rg + t1 t2; q k s Vh.z; Nh; H; l local; T1), 10+i march step i (position, cs), 110+i its tap (u, v, z - h), 200+k refine step k (position), 300+k its tap. */
static int g_trace_x = -1, g_trace_y = -1, g_trace_n = 0, g_trace_cap = 0, g_trace_on = 0, g_trace_spec = 0;
static float *g_trace_buf = NULL;
void rfxo_set_trace(int x, int y, float *buf, int cap) { g_trace_x = x; g_trace_y = y; g_trace_buf = buf; g_trace_cap = cap; g_trace_n = 0; }
int rfxo_trace_count(void) { return g_trace_n; }
static inline void TRACE(int tag, float a, float b, float c, float d) {
    if (!g_trace_on || !g_trace_buf || g_trace_n + 5 > g_trace_cap) return;
    float *q = g_trace_buf + g_trace_n;
    q[0] = (float)tag; q[1] = a; q[2] = b; q[3] = c; q[4] = d;
    g_trace_n += 5;
}
static inline v3 sample_ggx_vndf(v3 V, float ax, float ay, float r1, float r2) {
    v3 Vh = normalize3(V3(ax * V.x, ay * V.y, V.z));
    float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
    v3 T1;
    if (lensq > 0.0f) { float is = 1.0f / sqrtf(lensq); T1 = V3(-Vh.y * is, Vh.x * is, 0.0f * is); }
    else T1 = V3(1.0f, 0.0f, 0.0f);
    v3 T2 = cross3(Vh, T1);
    float r = sqrtf(r1);
    float phi = 2.0f * M_PIf * r2;
    float t1 = r * cosf(phi), t2 = r * sinf(phi);
    float s = 0.5f * (1.0f + Vh.z);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + s * t2;
    float q = 1.0f - t1 * t1 - t2 * t2;
    float k = sqrtf(fmaxf(0.0f, q));
    /* ill-conditioning, not a branch: at blueNoise.r -> 1 the sample sits on the rim of the projected disk, q cancels to a few ulps and
     * sqrt amplifies them: dk = dq / 2k (or sqrt(dq) at q ~ 0).  H carries that uncertainty into l = reflect(-V, H) (twice) and into
     * every angle derived from it (calc_angles, where h = normalize(v + l) and v + l -> 0 for a rim sample: dot(V, H) -> 0). */
    {
        const float dq = 6e-7f; /* ~5 ulps of the O(1) terms */
        float dk = q > dq ? dq / (2.0f * (sqrtf)(q)) : (sqrtf)(dq);
        g_unc_dir = 2.0f * dk + 4e-7f;
        /* ... and a second one (round 6, the root cause of the 16-frame sequence's "open pixel", tools/open_pixel_trace.py): T1 = (-Vh.y, Vh.x, 0) *
         * inversesqrt(lensq) is the DIRECTION of Vh's tangential part.  When the view vector is almost the normal (V local ~ (1e-3, 4e-4, 1)) that part is
         * the small difference of O(1) products in ToLocal's dot(V, T), dot(V, B): one ulp of those (6e-8 — a fused against an unfused multiply-add, the
         * order of a three-term sum) is a relative 1e-4 of it, T1 turns by that, and the sampling disk — hence H, l and the whole ray — turns with
         * it by r * t1 error.  Measured at the open pixel: V local y -3.5566e-4 (restatement) against -3.5569e-4 (reference GL), T1 off by 2.2e-5,
         * the ray direction by 2.1e-5, the refine tap by 0.0074 texel across a silhouette. */
        if (lensq > 0.0f) g_unc_dir += r * 1.2e-7f / (sqrtf)(lensq);
    }
    v3 Nh = add3(add3(mul3(T1, t1), mul3(T2, t2)), mul3(Vh, k));
    TRACE(20, r1, r2, t1, t2); TRACE(21, q, k, s, Vh.z); TRACE(22, Nh.x, Nh.y, Nh.z, 0.0f); TRACE(25, T1.x, T1.y, T1.z, 0.0f);
    return normalize3(V3(ax * Nh.x, ay * Nh.y, fmaxf(0.0f, Nh.z)));
}
static inline void onb(v3 N, v3 *T, v3 *B) { /* :172-176 */
    v3 up = fabsf(N.z) < 0.9999999f ? V3(0, 0, 1) : V3(1, 0, 0);
    *T = normalize3(cross3(up, N));
    *B = cross3(N, *T);
}
static inline v3 cosine_sample_hemisphere(v3 n, float ux, float uy) { /* :183-191 */
    float r = sqrtf(ux), theta = 2.0f * M_PIf * uy;
    v3 b = normalize3(cross3(n, V3(0.0f, 1.0f, 1.0f)));
    v3 t = cross3(b, n);
    v3 s = add3(add3(mul3(b, r * sinf(theta)), mul3(n, sqrtf(1.0f - ux))), mul3(t, r * cosf(theta)));
    return normalize3(s);
}

typedef struct {
    int W, H;
    const float *depth; const uint32_t *gbuffer; const float *direct; const float *history; const uint8_t *blue;
    const rfx_ssgi_params *p;
    float nearMulFar, farMinusNear, cameraFar;
    const float *env; int env_w, env_h, env_levels; /* scene.environment: the whole mip chain, RGBA float32, level l after level l-1 */
    int outW, outH; /* the pass's render target = `resolution` (SSGIPass.js:52-57): W*resolutionScale x H*resolutionScale */
    const float *marginal, *conditional; float totalSumWhole, totalSumDecimal; /* EquirectHdrInfo tables (importanceSampling) */
} k1_ctx;

static inline float k1_view_z(const k1_ctx *c, float depth) { /* getViewZ ssgi_utils.frag:7-13, both camera variants */
    if (!c->p->camera.isPerspective) return depth * (float)((double)c->p->camera.near_ - (double)c->p->camera.far_) - c->p->camera.near_;
    return c->nearMulFar / (c->farMinusNear * depth - c->cameraFar);
}
static inline void k1_project(const k1_ctx *c, v3 pos, float *u, float *v) { /* viewSpaceToScreenSpace :26-33 */
    v4 pc = mat_mul_v4(c->p->camera.projectionMatrix, pos.x, pos.y, pos.z, 1.0f);
    *u = (pc.x / pc.w) * 0.5f + 0.5f;
    *v = (pc.y / pc.w) * 0.5f + 0.5f;
}
/* The texel a march / refine tap lands in.  The tap's coordinate is the projection of a position that went through `n_updates` rounded
 * updates (half an ulp each, relative to the position and so to the projected coordinate).  The reference GL's roundings are the
 * restatement's IEEE ones almost everywhere (K1's packed texels are bit-identical on > 99.9 % of the pixels), but not everywhere —
 * llvmpipe's code generator may contract and reorder — and what has accumulated by step n is of that size.  A tap closer than that to a
 * texel boundary is only a DECISION when the texel across the boundary decides differently (a depth discontinuity: a silhouette), so that is
 * what is tested: kind 0 = RayMarch's hit test (:463), kind 1 = BinarySearch's sign (:493).  Found by the 16-frame sequence of round 4: one
 * pixel of 2 M at frame 10 (steps 40), a refine tap 2e-3 texel from the boundary between a surface at z = -8.6 and one at z = -14.3;
 * restatement and kernel bit-identical, the GL on the other side; perturbed transcendentals cannot move a refined position that far. */
static inline int k1_tap_decides(const k1_ctx *c, float z, float h, int kind) {
    float diff = z - h;
    return kind ? (diff >= 0.0f) : (diff >= 0.0f && diff < c->p->thickness);
}
static void margin_tap(const k1_ctx *c, float u, float v, float h, int n_updates, int kind) {
    const float cc[2] = {u * (float)c->W, v * (float)c->H};
    const int size[2] = {c->W, c->H};
    int idx[2];
    for (int a = 0; a < 2; a++) {
        if (!(cc[a] > 0.0f && cc[a] < (float)size[a])) return; /* clamped region (or not a number): flat */
        idx[a] = (int)cc[a];
        if (idx[a] > size[a] - 1) idx[a] = size[a] - 1;
    }
    const int here = k1_tap_decides(c, k1_view_z(c, c->depth[(size_t)idx[1] * c->W + idx[0]]), h, kind);
    for (int a = 0; a < 2; a++) {
        const float fr = cc[a] - floorf(cc[a]);
        const float slack = (4.0f + 0.5f * (float)n_updates) * 1.1920929e-7f * fabsf(cc[a]);
        for (int side = -1; side <= 1; side += 2) {
            const float dist = side < 0 ? fr : 1.0f - fr;
            int n[2] = {idx[0], idx[1]};
            n[a] += side;
            if (dist >= slack || n[a] < 0 || n[a] > size[a] - 1) continue;
            if (k1_tap_decides(c, k1_view_z(c, c->depth[(size_t)n[1] * c->W + n[0]]), h, kind) != here) margin_note_tap(dist / fmaxf(slack, 1e-30f));
        }
    }
}
/* BinarySearch ssgi.frag:477-503 */
static void k1_binary_search(const k1_ctx *c, v3 *dir, v3 *hitPos, float *u, float *v, int n_updates) {
    dims d = {c->W, c->H};
    *dir = mul3(*dir, 0.5f);
    *hitPos = sub3(*hitPos, *dir);
    for (int i = 0; i < c->p->refineSteps; i++) {
        k1_project(c, *hitPos, u, v);
        float z = k1_view_z(c, fetch_r32f(c->depth, d, *u, *v));
        float diff = z - hitPos->z;
        if (g_trace_spec) { TRACE(200 + i, hitPos->x, hitPos->y, hitPos->z, 0.0f); TRACE(300 + i, *u, *v, diff, z); }
        margin_cmp(z, hitPos->z, MARGIN_REL_MARCH); /* :493 sign of diff */
        margin_tap(c, *u, *v, hitPos->z, n_updates + i + 1, 1);
        *dir = mul3(*dir, 0.5f);
        if (diff >= 0.0f) *hitPos = sub3(*hitPos, *dir); else *hitPos = add3(*hitPos, *dir);
    }
    k1_project(c, *hitPos, u, v);
}
/* RayMarch ssgi.frag:441-475 */
static void k1_ray_march(const k1_ctx *c, v3 *dir, v3 *hitPos, float random_b, float *u, float *v) {
    dims d = {c->W, c->H};
    /* the taps' texel boundaries get no blanket slack: a tap that changes texel rarely changes the hit decision.  The perturbed runs move
     * the ray by its actual error and re-take every decision; the margins below cover the z compares themselves */
    *dir = mul3(*dir, c->p->rayDistance / (float)c->p->steps);
    *u = 0.0f; *v = 0.0f;
    for (int i = 1; i < c->p->steps; i++) {
        float m = (float)i + random_b - 0.5f;
        float cs = g_gl_exp ? gl_march_cs(m) : 1.0f - expf(-0.25f * (m * m));
        *hitPos = add3(*hitPos, mul3(*dir, cs));
        k1_project(c, *hitPos, u, v);
        float z = k1_view_z(c, fetch_r32f(c->depth, d, *u, *v));
        float diff = z - hitPos->z;
        if (g_trace_spec) { TRACE(10 + i, hitPos->x, hitPos->y, hitPos->z, cs); TRACE(110 + i, *u, *v, diff, z); }
        margin_cmp(z, hitPos->z, MARGIN_REL_MARCH);                                   /* :463 diff >= 0 */
        margin_cmp(z - c->p->thickness, hitPos->z, MARGIN_REL_MARCH);                 /* :463 diff < thickness */
        margin_tap(c, *u, *v, hitPos->z, i, 0);
        if (diff >= 0.0f && diff < c->p->thickness) {
            if (c->p->refineSteps == 0) return;
            k1_binary_search(c, dir, hitPos, u, v, i);
            return;
        }
    }
    *hitPos = V3(10.0e9f, 10.0e9f, 10.0e9f);
}
static inline float smoothstepf(float e0, float e1, float x) {
    float t = clampf((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
/* ---- scene.environment (USE_ENVMAP) */
static inline const float *env_level(const k1_ctx *c, int level, int *w, int *h) {
    const float *t = c->env;
    int lw = c->env_w, lh = c->env_h;
    for (int l = 0; l < level; l++) { t += 4 * (size_t)lw * lh; lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1; }
    *w = lw; *h = lh;
    return t;
}
/* GLSL acos as the oracle GL's compiler (Mesa) evaluates it: pi/2 - asin polynomial (measured against llvmpipe: 2.4e-7) */
static inline float mesa_acos(float x) {
    float ax = fabsf(x);
    float r = 1.5707963267948966f - sqrtf(1.0f - ax) * (1.5707963267948966f + ax * (-0.21460183660255172f + ax * (0.08132463f + ax * -0.02363318f)));
    return 1.5707963267948966f - (x < 0.0f ? -r : r);
}
/* getEnvColor ssgi.frag:311-346 (no BOX_PROJECTED_ENV_MAP; isEnvSample is false without MIS): textureLod on a LinearMipMapLinear
 * texture = two CLAMP_TO_EDGE bilinear taps blended by fract(lod), lod clamped to the chain (measured on llvmpipe) */
static v3 k1_env_trilinear(const k1_ctx *c, float u, float v, float lod_unclamped) {
    float lod = fminf(fmaxf(lod_unclamped, 0.0f), (float)(c->env_levels - 1));
    float fl = floorf(lod);
    int l0 = (int)fl, l1 = l0 + 1 > c->env_levels - 1 ? c->env_levels - 1 : l0 + 1;
    int w0, h0, w1, h1;
    const float *t0 = env_level(c, l0, &w0, &h0), *t1 = env_level(c, l1, &w1, &h1);
    dims d0 = {w0, h0}, d1 = {w1, h1};
    v4 c0 = fetch_f4_linear(t0, d0, u, v), c1 = fetch_f4_linear(t1, d1, u, v);
    float f = lod - fl;
    return V3(lerpf(f, c0.x, c1.x), lerpf(f, c0.y, c1.y), lerpf(f, c0.z, c1.z));
}
static v3 k1_env_color(const k1_ctx *c, v3 l, float roughness, int isDiffuseSample, int isEnvSample) {
    if (!c->p->useEnvMap) return V3(0, 0, 0);
    v3 dir = normalize3(v4_mul_mat_xyz(c->p->camera.matrixWorldInverse, l, 0.0f)); /* (vec4(l, 0.) * viewMatrix).xyz :315 */
    float maxMip = 0.0f;
    { int m = c->env_w > c->env_h ? c->env_w : c->env_h, lg = 0; while ((m >> (lg + 1)) > 0) lg++; maxMip = (float)(lg + 1); } /* Utils.js:30-34 */
    float mip = c->p->envBlur * maxMip;
    if (!isDiffuseSample && roughness < 0.15f) mip *= roughness / 0.15f;
    /* equirectDirectionToUv ssgi_utils.frag:64-74 */
    float u = atan2f(dir.z, dir.x) / (2.0f * M_PIf), v = mesa_acos(dir.y) / M_PIf;
    u += 0.5f; v = 1.0f - v;
    /* The equirect u is undefined at the poles: a direction whose components are known to +-unc has u known to +-unc / (2 pi rho) of the
     * map's width, rho = |(dir.x, dir.z)|.  A cosine-hemisphere sample with blue-noise byte 0 IS the surface normal — straight up over a floor —
     * and its u is then the atan2 of two rounding residues of the view -> world transform: two implementations with ulp-accurate
     * inversesqrt agree on nothing there, and the deeper levels of the chain (envBlur) do differ along their top row.  Unstable when that
     * uncertainty exceeds 1e-3 texel of level 0.  (Round 6: the kernels against the reference chain over the variants, 1 pixel in 4000 at
     * envBlur 0.5; tools/diag/env_device_vs_oracle.py.) */
    margin_note((sqrtf)(dir.x * dir.x + dir.z * dir.z) * (2.0f * M_PIf) * 1e-3f / (g_unc_dir * (float)c->env_w));
    v3 col = k1_env_trilinear(c, u, v, mip);
    const float maxEnvLum = isEnvSample ? 100.0f : 25.0f; /* :328-340 */
    float envLum = lum(col);
    if (envLum > maxEnvLum) col = mul3(col, maxEnvLum / envLum);
    return col;
}
/* doSample ssgi.frag:362-439 */
static v3 k1_do_sample(const k1_ctx *c, const material *mat, v3 viewPos, v3 viewNormal, float metalness, float roughness,
                       int isDiffuseSample, int isEnvSample, float NoV, float NoL, float NoH, float LoH, float VoH, v4 random,
                       v3 *l, v3 *hitPos, float *brdf, float *pdf) {
    (void)VoH;
    dims d = {c->W, c->H};
    float cosTheta = fmaxf(0.0f, dot3(viewNormal, *l));
    if (isDiffuseSample) {
        *brdf = eval_disney_diffuse(NoL, NoV, LoH, roughness, metalness);
        *pdf = NoL / M_PIf;
    } else {
        *brdf = eval_disney_specular(roughness, NoH, NoV, NoL);
        *pdf = ggx_vndf_pdf(NoH, NoV, roughness);
    }
    *brdf *= cosTheta;
    *pdf = fmaxf(0.00001f, *pdf);
    *hitPos = viewPos;
    float cu, cv;
    k1_ray_march(c, l, hitPos, random.z, &cu, &cv);
    /* (g_fetch_rel stays set for the history fetch below; k1_pixel resets it) */
    int allowMissed = c->p->missedRays != 0;
    int isMissed = hitPos->x == 10.0e9f;
    v3 env = k1_env_color(c, *l, roughness, isDiffuseSample, isEnvSample); /* black without an env map (:342-345) */
    if (isMissed && !allowMissed) return env;
    /* velocityTexture is never wired (SSGIPass.js:89) -> three's empty texture -> velocity = 0 */
    float ru = cu - 0.0f, rv = cv - 0.0f;
    v3 ssgi;
    margin_cmp(ru, 0.0f, MARGIN_REL_MARCH); margin_cmp(ru, 1.0f, MARGIN_REL_MARCH);
    margin_cmp(rv, 0.0f, MARGIN_REL_MARCH); margin_cmp(rv, 1.0f, MARGIN_REL_MARCH);
    if (ru >= 0.0f && ru <= 1.0f && rv >= 0.0f && rv <= 1.0f) {
        /* accumulatedTexture = denoiser.texture (SSGIPass.js:89): K4's / K2's RGBA32F target, or three's empty texture ("denoised") */
        v4 h = {0.f, 0.f, 0.f, 0.f};
        if (c->p->historySource != 2) h = fetch_f4(c->history, d, ru, rv);
        v3 gi = V3(h.x, h.y, h.z);
        /* getSaturation :348-360 */
        float mx = fmaxf(fmaxf(mat->diffuse.x, mat->diffuse.y), mat->diffuse.z);
        float mn = fminf(fminf(mat->diffuse.x, mat->diffuse.y), mat->diffuse.z);
        float sat = (mx == mn) ? 0.0f : (mx - mn) / mx;
        float L = lum(gi);
        gi = mix3(gi, V3(L, L, L), (1.0f - roughness) * sat * 0.4f);
        const float border = 0.15f;
        float bf = smoothstepf(0.0f, border, cu) * smoothstepf(1.0f, 1.0f - border, cu) * smoothstepf(0.0f, border, cv) *
                   smoothstepf(1.0f, 1.0f - border, cv);
        bf = sqrtf(bf);
        ssgi = mix3(env, gi, bf);
    } else {
        return env;
    }
    if (allowMissed) { /* :430-436, envMapSample is vec3(0) */
        if (0.0f > lum(ssgi)) ssgi = V3(0, 0, 0);
    }
    return ssgi;
}
static inline void calc_angles(v3 l, v3 v, v3 n, float *NoL, float *NoH, float *LoH, float *VoH) { /* :93-100 */
    const float E = 0.00001f, OME = 1.0f - 0.00001f;
    /* h = normalize(v + l): a direction error u in l turns h by u / |v + l|; the BRDF, pdf and Fresnel terms move by about as much.
     * margin < 1: they may move by more than the 1e-3 tolerance */
    margin_note(length3(add3(v, l)) * 1e-3f / g_unc_dir);
    v3 h = normalize3(add3(v, l));
    *NoL = clampf(dot3(n, l), E, OME);
    *NoH = clampf(dot3(n, h), E, OME);
    *LoH = clampf(dot3(l, h), E, OME);
    *VoH = clampf(dot3(v, h), E, OME);
}

/* sampleEquirectProbability's table walk (ssgi_utils.frag:212-214): marginalWeights is an env_h x 1 NEAREST texture read at (blueNoise.x, 0),
 * conditionalWeights env_w x env_h read at (blueNoise.y, v); the pixel is whatever ivec2(vUv * resolution) makes of it = (px, py) */
static inline void k1_cdf_uv(const k1_ctx *c, int px, int py, float *u, float *v) {
    dims dres = {c->outW, c->outH};
    {   /* A quad partner that is a background texel (main() returned at :109-113) never ran the blue-noise fetch: on the oracle's GL its
         * `random` still holds the zero initialisation, so it contributes the table entry of (0, 0) to the quad's derivatives (GLSL leaves
         * derivatives after a non-uniform return undefined; measured on llvmpipe, reproduced).  A partner OUTSIDE the target (the last
         * column / row of an odd-sized one) is a helper invocation like any other: it runs the fragment on its extrapolated vUv — the depth
         * fetch clamps to the edge texel, the blue-noise pixel is ivec2(vUv * resolution) = (px, py) beyond the target.  (Until round 6 it
         * was modelled like a background partner; tools/fuzz_variants_vs_reference_gl.py --only-envmis on odd sizes: 651 unexplained
         * pixels of the last column before, none after.) */
        dims d = {c->W, c->H};
        float pu = frag_u(px, py, c->outW, c->outH), pv = frag_v(py, c->outW, c->outH);
        if (fetch_r32f(c->depth, d, pu, pv) == 1.0f) {
            *v = c->marginal[0];
            *u = c->conditional[(size_t)nearest_idx(*v, c->env_h) * c->env_w + 0];
            return;
        }
    }
    v4 r = blue_noise(c->blue, px, py, c->p->blueNoiseIndex, frag_u(px, py, c->outW, c->outH), frag_v(py, c->outW, c->outH), dres);
    *v = c->marginal[nearest_idx(r.x, c->env_h)];
    *u = c->conditional[(size_t)nearest_idx(*v, c->env_h) * c->env_w + nearest_idx(r.y, c->env_w)];
}

static void k1_pixel(const k1_ctx *c, int x, int y, uint32_t *out) {
    const rfx_ssgi_params *p = c->p;
    const float *C = p->camera.matrixWorld, *Vw = p->camera.matrixWorldInverse;
    const float *P = p->camera.projectionMatrix, *Pi = p->camera.projectionMatrixInverse;
    dims d = {c->W, c->H};
    float u = pert_uv(frag_u(x, y, c->outW, c->outH)), v = pert_uv(frag_v(y, c->outW, c->outH)); /* vUv of the (possibly smaller) target */
    float depth = fetch_r32f(c->depth, d, u, v);
    if (depth == 1.0f) { /* :109-113 */
        v4 dl = fetch_f4(c->direct, d, u, v);
        pack_two_vec4(dl, dl, out);
        return;
    }
    material mat = get_material(fetch_u4(c->gbuffer, d, u, v));
    float roughnessSq = clampf(mat.roughness * mat.roughness, 0.000001f, 1.0f);
    float viewZ = k1_view_z(c, depth);
    /* getViewPosition ssgi_utils.frag:17-24 */
    float clipW = P[2 * 4 + 3] * viewZ + P[3 * 4 + 3];
    float cx = ((u - 0.5f) * 2.0f) * clipW, cy = ((v - 0.5f) * 2.0f) * clipW, cz = ((viewZ - 0.5f) * 2.0f) * clipW, cw = 1.0f * clipW;
    v4 pp = mat_mul_v4(Pi, cx, cy, cz, cw);
    v3 viewPos = V3(pp.x, pp.y, viewZ);
    v3 viewDir = normalize3(viewPos);
    v3 worldNormal = mat.normal;
    v3 viewNormal = normalize3(v4_mul_mat_xyz(C, worldNormal, 0.0f));
    v3 n = viewNormal, vv = neg3(viewDir);
    float NoV = fmaxf(0.00001f, dot3(n, vv));
    v3 V = v4_mul_mat_xyz(Vw, vv, 0.0f);
    v3 N = worldNormal, T, B;
    onb(N, &T, &B);
    V = V3(dot3(V, T), dot3(V, B), dot3(V, N)); /* ToLocal */
    v3 f0 = mix3(V3(0.04f, 0.04f, 0.04f), mat.diffuse, mat.metalness);
    v4 random = blue_noise(c->blue, x, y, p->blueNoiseIndex, u, v, (dims){c->outW, c->outH}); /* `resolution` = the render target's size */
    g_trace_on = (x == g_trace_x && y == g_trace_y);
    v3 Hh = sample_ggx_vndf(V, roughnessSq, roughnessSq, random.x, random.y);
    TRACE(23, Hh.x, Hh.y, Hh.z, 0.0f); TRACE(26, V.x, V.y, V.z, 0.0f);
    margin_note(fabsf(Hh.z) / MARGIN_REL_SHORT); /* |H| = 1 */
    if (Hh.z < 0.0f) Hh = neg3(Hh);
    /* reflect(-V, H) = I - 2*dot(N,I)*N with I=-V */
    v3 I = neg3(V);
    float dNI = dot3(Hh, I);
    v3 l = normalize3(sub3(I, mul3(Hh, 2.0f * dNI)));
    TRACE(24, l.x, l.y, l.z, dNI);
    l = add3(add3(mul3(T, l.x), mul3(B, l.y)), mul3(N, l.z)); /* ToWorld */
    l = normalize3(v4_mul_mat_xyz(C, l, 0.0f));
    float NoL, NoH, LoH, VoH;
    calc_angles(l, vv, n, &NoL, &NoH, &LoH, &VoH);
    int isDiffuseSample = 0;
    if (p->mode == 0) { /* :169-186 */
        v3 F = f_schlick3(f0, VoH);
        float diffW = (1.0f - mat.metalness) * lum(mat.diffuse);
        float specW = lum(F);
        diffW = fmaxf(diffW, 0.00001f);
        specW = fmaxf(specW, 0.00001f);
        float invW = 1.0f / (diffW + specW);
        diffW *= invW;
        isDiffuseSample = random.z < diffW;
        margin_cmp(random.z, diffW, MARGIN_REL_SHORT); /* :186 */
    }
    /* importanceSampling :197-216 */
    float emsPdf = 1.0f;
    int isEnvSample = 0;
    v3 envMisDir = V3(0, 0, 0);
    if (p->importanceSampling) {
        float cu, cv;
        k1_cdf_uv(c, x, y, &cu, &cv);
        /* equirectUvToDirection ssgi_utils.frag:77-86 */
        float theta = ((cu - 0.5f) * 2.0f) * M_PIf, phi = (1.0f - cv) * M_PIf;
        float sinPhi = sinf(phi);
        v3 derived = V3(sinPhi * cosf(theta), cosf(phi), sinPhi * sinf(theta));
        /* texture(info.map, uv): implicit lod, one per 2x2 quad from its top-left pixel's differences (measured on llvmpipe) */
        int qx = x & ~1, qy = y & ~1;
        float tlu, tlv, tru, trv, blu, blv;
        k1_cdf_uv(c, qx, qy, &tlu, &tlv); k1_cdf_uv(c, qx + 1, qy, &tru, &trv); k1_cdf_uv(c, qx, qy + 1, &blu, &blv);
        float fw = (float)c->env_w, fh = (float)c->env_h;
        float ax = (tru - tlu) * fw, ay = (trv - tlv) * fh, bx = (blu - tlu) * fw, by = (blv - tlv) * fh;
        float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
        uint32_t bits; memcpy(&bits, &rho2, 4);
        uint32_t mb = (bits & 0x7fffffu) | 0x3f800000u; float mant; memcpy(&mant, &mb, 4);
        float lod = 0.5f * ((float)((int)((bits >> 23) & 0xffu) - 127) + (mant - 1.0f));
        v3 col = k1_env_trilinear(c, cu, cv, lod);
        float totalSum = c->totalSumWhole + c->totalSumDecimal;
        emsPdf = ((float)c->env_w * (float)c->env_h) * (lum(col) / totalSum);
        envMisDir = normalize3(v4_mul_mat_xyz(C, derived, 0.0f)); /* (vec4(dir, 0.) * cameraMatrixWorld).xyz */
        float prob = dot3(envMisDir, viewNormal);
        prob *= mat.roughness;
        prob = fminf(1.0f - 0.00001f, prob);
        isEnvSample = random.w < prob;
        margin_cmp(random.w, prob, MARGIN_REL_SHORT);
        if (isEnvSample) {
            emsPdf /= 1.0f - prob;
            l = envMisDir;
            calc_angles(l, vv, n, &NoL, &NoH, &LoH, &VoH);
        } else {
            emsPdf = 1.0f - prob;
        }
    }
    float unc_spec = g_unc_dir;
    g_unc_dir = 4e-7f; /* the cosine-hemisphere and environment directions are well conditioned */
    v3 diffuseRay = isEnvSample ? envMisDir : cosine_sample_hemisphere(viewNormal, random.x, random.y);
    v3 specularRay = isEnvSample ? envMisDir : l;
    v3 diffuseGI = V3(0, 0, 0), specularGI = V3(0, 0, 0), hitPos = V3(0, 0, 0);
    float diffuseSamples = 0.0f;
    float brdf, pdf;
    if (p->mode == 0 && isDiffuseSample) { /* :222-242 */
        l = diffuseRay;
        calc_angles(l, vv, n, &NoL, &NoH, &LoH, &VoH);
        v3 gi = k1_do_sample(c, &mat, viewPos, viewNormal, mat.metalness, roughnessSq, isDiffuseSample, isEnvSample, NoV, NoL, NoH, LoH, VoH, random,
                             &l, &hitPos, &brdf, &pdf);
        gi = mul3(gi, brdf);
        if (isEnvSample) gi = mul3(gi, (emsPdf * emsPdf) / (emsPdf * emsPdf + pdf * pdf)); /* misHeuristic */
        else gi = V3(gi.x / pdf, gi.y / pdf, gi.z / pdf);
        gi = V3(gi.x / emsPdf, gi.y / emsPdf, gi.z / emsPdf);
        diffuseSamples += 1.0f;
        diffuseGI = gi; /* mix(0, gi, 1/1) */
    }
    l = specularRay; /* :246-265 */
    /* proofs only: a direction whose conditioning (sample_ggx_vndf) leaves it uncertain by more than a few ulps is moved by that uncertainty, so that the
     * perturbed re-evaluations re-take every march and refine decision of the ray under it (well-conditioned rays — 4e-7 — are left alone: moving every
     * ray by an ulp would put every tap near a texel boundary at risk) */
    if (g_pert_seed && g_pert_state && !isEnvSample && unc_spec > 2e-6f) {
        const float a = unc_spec - 4e-7f;
        l = normalize3(V3(l.x + pert_sign() * a, l.y + pert_sign() * a, l.z + pert_sign() * a));
    }
    g_trace_on = (x == g_trace_x && y == g_trace_y);
    g_trace_spec = g_trace_on;
    TRACE(0, viewPos.x, viewPos.y, viewPos.z, viewZ);
    TRACE(1, specularRay.x, specularRay.y, specularRay.z, random.z);
    TRACE(3, viewNormal.x, viewNormal.y, viewNormal.z, roughnessSq);
    g_unc_dir = isEnvSample ? 4e-7f : unc_spec;
    calc_angles(l, vv, n, &NoL, &NoH, &LoH, &VoH);
    {
        v3 gi = k1_do_sample(c, &mat, viewPos, viewNormal, mat.metalness, roughnessSq, isDiffuseSample, isEnvSample, NoV, NoL, NoH, LoH, VoH, random,
                             &l, &hitPos, &brdf, &pdf);
        gi = mul3(gi, brdf);
        if (isEnvSample) gi = mul3(gi, (emsPdf * emsPdf) / (emsPdf * emsPdf + pdf * pdf));
        else gi = V3(gi.x / pdf, gi.y / pdf, gi.z / pdf);
        gi = V3(gi.x / emsPdf, gi.y / emsPdf, gi.z / emsPdf);
        specularGI = gi;
        TRACE(2, hitPos.x, hitPos.y, hitPos.z, 0.0f);
        g_trace_spec = 0;
        g_trace_on = 0;
    }
    v3 specularHitPos = hitPos;
    if (p->useDirectLight) { /* :267-272 */
        v4 dl = fetch_f4(c->direct, d, u, v);
        diffuseGI = add3(diffuseGI, V3(dl.x, dl.y, dl.z));
        specularGI = add3(specularGI, V3(dl.x, dl.y, dl.z));
    }
    if (p->mode == 0 && diffuseSamples == 0.0f) diffuseGI = V3(-1.0f, -1.0f, -1.0f); /* :277-278 */
    float rayLength = 0.0f; /* :284-296 */
    int missed = hitPos.x > 10.0e8f;
    if (!missed) {
        v4 hw = mat_mul_v4(C, specularHitPos.x, specularHitPos.y, specularHitPos.z, 1.0f);
        v3 camPos = V3(C[12], C[13], C[14]);
        rayLength = length3(sub3(camPos, V3(hw.x, hw.y, hw.z)));
    }
    if (p->mode == 0) { /* :302-304 */
        v4 gD = {diffuseGI.x, diffuseGI.y, diffuseGI.z, mat.roughness};
        v4 gS = {specularGI.x, specularGI.y, specularGI.z, rayLength};
        pack_two_vec4(gD, gS, out);
    } else { /* MODE_SSR :298-300,306-307: raw vec4(specularGI, uintBitsToFloat(packHalf2x16(rayLength, roughness))) */
        memcpy(&out[0], &specularGI.x, 4); memcpy(&out[1], &specularGI.y, 4); memcpy(&out[2], &specularGI.z, 4);
        out[3] = pack_half2(rayLength, mat.roughness);
    }
}

/* ==================================================================== encode side of the codec (the raster passes' fragment epilogues)
 * packGBuffer gbuffer_packing.glsl:166-178 over attribute planes; texels with depth == 1 keep the clear colour (0,0,0,1). */
static inline uint32_t enc_vec4_to_float(float x, float y, float z, float w) { /* vec4ToFloat :143-149 */
    const float o = 0.0001f, one = 0.999999f;
    uint32_t r = (uint32_t)(fminf(x + o, one) * 255.0f), g = (uint32_t)(fminf(y + o, one) * 255.0f);
    uint32_t b = (uint32_t)(fminf(z + o, one) * 255.0f), a = (uint32_t)(fminf(w + o, one) * 255.0f);
    return (a << 24) | (b << 16) | (g << 8) | r;
}
static inline uint32_t enc_pack_normal(float nx, float ny, float nz) { /* packNormal(encodeOctWrap) :36-61 */
    float s = fabsf(nx) + fabsf(ny) + fabsf(nz);
    nx /= s; ny /= s; nz /= s;
    float wx = 1.0f - fabsf(ny), wy = 1.0f - fabsf(nx);
    if (nx < 0.0f) wx = -wx;
    if (ny < 0.0f) wy = -wy;
    float ox = nz > 0.0f ? nx : wx, oy = nz > 0.0f ? ny : wy;
    return pack_half2(ox * 0.5f + 0.5f, oy * 0.5f + 0.5f);
}
static inline float enc_color2float(float r, float g, float b) { /* color2float :17-22 */
    const float o = 0.0001f, one = 0.999999f;
    r = fminf(r + o, one); g = fminf(g + o, one); b = fminf(b + o, one);
    return floorf(r * 256.0f + 0.5f) + floorf(b * 256.0f + 0.5f) * 257.0f + floorf(g * 256.0f + 0.5f) * 257.0f * 257.0f;
}
static inline uint32_t enc_rgbe8(float r, float g, float b) { /* vec4ToFloat(encodeRGBE8) :127-134 */
    float mx = fmaxf(fmaxf(r, g), b);
    float fexp = ceilf(log2f(mx)), sc = exp2f(fexp);
    return enc_vec4_to_float(r / sc, g / sc, b / sc, (fexp + 128.0f) / 255.0f);
}
int rfxo_pack_gbuffer(int n, const float *diffuse, const float *normal, const float *roughness, const float *metalness, const float *emissive,
                      const float *depth, uint32_t *out) {
    for (int i = 0; i < n; i++) {
        uint32_t *o = out + 4 * (size_t)i;
        if (depth && depth[i] == 1.0f) { o[0] = o[1] = o[2] = 0; o[3] = 0x3f800000u; continue; }
        o[0] = enc_vec4_to_float(diffuse[4 * i], diffuse[4 * i + 1], diffuse[4 * i + 2], diffuse[4 * i + 3]);
        o[1] = enc_pack_normal(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        float f = enc_color2float(roughness[i], metalness[i], 0.0f);
        memcpy(&o[2], &f, 4);
        o[3] = enc_rgbe8(emissive[3 * i], emissive[3 * i + 1], emissive[3 * i + 2]);
    }
    return 0;
}
/* VelocityDepthNormalMaterial.js:76-83,186-188: vec4(vel.xy, packNormal(worldNormal), fragCoordZ) */
int rfxo_pack_velocity(int n, const float *velocity, const float *normal, const float *depth, uint32_t *out) {
    for (int i = 0; i < n; i++) {
        uint32_t *o = out + 4 * (size_t)i;
        if (depth[i] == 1.0f) { o[0] = o[1] = o[2] = 0; o[3] = 0x3f800000u; continue; }
        memcpy(&o[0], &velocity[2 * i], 4); memcpy(&o[1], &velocity[2 * i + 1], 4);
        o[2] = enc_pack_normal(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
        memcpy(&o[3], &depth[i], 4);
    }
    return 0;
}

/* ------------------------------------------------------------------ CubeToEquirectEnvPass (src/ssgi/pass/CubeToEquirectEnvPass.js:21-42)
 * One `textureCube(cubeMap, dir)` per texel of the equirectangular target.  The cube lookup as the oracle's GL does it (measured on
 * llvmpipe WITH three's `precision highp samplerCube;` — without it Mesa lowers the fetch to fp16 and the filter looks 8-bit;
 * oracle/glref/probes/probe_cube.py): major axis by >= in x, y, z order; (s, t) = (sc * (1 / ma)) * 0.5 + 0.5 (bit-exact on every
 * interior lookup); seamless: a footprint texel beyond the face edge is the texel of the neighbouring face the extended position
 * projects to, a footprint texel beyond a CORNER (no such texel exists) is the average of the other three; blend = lerp in x, then
 * in y, each lerp a fused a + w * (b - a).  Faces: +X -X +Y -Y +Z -Z, each S x S RGBA32F, row j = t index (as handed to glTexImage2D). */
static inline void cube_face(float x, float y, float z, int *face, float *s, float *t) {
    float ax = fabsf(x), ay = fabsf(y), az = fabsf(z), sc, tc, ma;
    if (ax >= ay && ax >= az) { *face = x >= 0.0f ? 0 : 1; sc = x >= 0.0f ? -z : z; tc = -y; ma = ax; }
    else if (ay >= az) { *face = y >= 0.0f ? 2 : 3; sc = x; tc = y >= 0.0f ? z : -z; ma = ay; }
    else { *face = z >= 0.0f ? 4 : 5; sc = z >= 0.0f ? x : -x; tc = -y; ma = az; }
    float ima = 1.0f / ma;
    *s = (sc * ima) * 0.5f + 0.5f;
    *t = (tc * ima) * 0.5f + 0.5f;
}
/* the point (a, b) of face `face` in its (sc / ma, tc / ma) coordinates, as a direction */
static inline void cube_face_dir(int face, float a, float b, float *x, float *y, float *z) {
    switch (face) {
    case 0: *x = 1.0f; *y = -b; *z = -a; break;
    case 1: *x = -1.0f; *y = -b; *z = a; break;
    case 2: *x = a; *y = 1.0f; *z = b; break;
    case 3: *x = a; *y = -1.0f; *z = -b; break;
    case 4: *x = a; *y = -b; *z = 1.0f; break;
    default: *x = -a; *y = -b; *z = -1.0f; break;
    }
}
/* texel (i, j) of the footprint on `face`; NULL for the texel beyond a corner */
static inline const float *cube_texel(const float *faces, int S, int face, int i, int j) {
    const int oi = i < 0 || i >= S, oj = j < 0 || j >= S;
    if (oi && oj) return NULL;
    if (oi || oj) { /* the centre of the would-be texel, projected onto the neighbouring face */
        float a = (((float)i + 0.5f) / (float)S) * 2.0f - 1.0f, b = (((float)j + 0.5f) / (float)S) * 2.0f - 1.0f, x, y, z, s, t;
        cube_face_dir(face, a, b, &x, &y, &z);
        cube_face(x, y, z, &face, &s, &t);
        i = (int)floorf(s * (float)S); j = (int)floorf(t * (float)S);
        i = i < 0 ? 0 : (i > S - 1 ? S - 1 : i); j = j < 0 ? 0 : (j > S - 1 ? S - 1 : j);
    }
    return faces + (((size_t)face * S + j) * S + i) * 4;
}
static void cube_linear(const float *faces, int S, float x, float y, float z, float *out) {
    int face; float s, t;
    cube_face(x, y, z, &face, &s, &t);
    float u = s * (float)S - 0.5f, v = t * (float)S - 0.5f;
    float fi = floorf(u), fj = floorf(v), fu = u - fi, fv = v - fj;
    int i0 = (int)fi, j0 = (int)fj;
    const float *q[4] = {cube_texel(faces, S, face, i0, j0), cube_texel(faces, S, face, i0 + 1, j0),
                         cube_texel(faces, S, face, i0, j0 + 1), cube_texel(faces, S, face, i0 + 1, j0 + 1)};
    for (int c = 0; c < 4; c++) {
        float tx[4]; int missing = -1;
        for (int k = 0; k < 4; k++) { if (q[k]) tx[k] = q[k][c]; else missing = k; }
        if (missing >= 0) { /* beyond the corner: the average of the three texels that exist */
            float sum = 0.0f;
            for (int k = 0; k < 4; k++) if (k != missing) sum += tx[k];
            tx[missing] = sum / 3.0f;
        }
        float top = fmaf(fu, tx[1] - tx[0], tx[0]), bot = fmaf(fu, tx[3] - tx[2], tx[2]);
        out[c] = fmaf(fv, bot - top, top);
    }
}
/* A CubeTexture with mipmaps (three's default: LinearMipmapLinearFilter + generateMipmaps): the chain glGenerateMipmap builds on the
 * oracle's GL per face (the 2x2 bilinear-centre average, as for 2-D textures), the implicit level of detail of `textureCube` as llvmpipe
 * derives it (measured to < 1e-6 in lod on a chain whose level l holds the constant l, probe_cube.py): PER PIXEL, from the differences
 * of the direction within the pixel's own row and own column of its 2x2 quad, through the quotient rule on the pixel's own face
 *   ds = (d(sc) * ma - sc * d(ma)) * (1 / ma)^2 * 0.5,   rho^2 = max(dsdx^2 + dtdx^2, dsdy^2 + dtdy^2) * S^2,
 * lod = 0.5 * (exponent(rho^2) + mantissa(rho^2) - 1) (the linear-mantissa log2), clamped to the chain; two seamless bilinear lookups
 * blended by fract(lod). */
static inline void cube_components(int face, float x, float y, float z, float *sc, float *tc, float *ma) {
    switch (face) {
    case 0: *sc = -z; *tc = -y; *ma = x; break;
    case 1: *sc = z; *tc = -y; *ma = -x; break;
    case 2: *sc = x; *tc = z; *ma = y; break;
    case 3: *sc = x; *tc = -z; *ma = -y; break;
    case 4: *sc = x; *tc = -y; *ma = z; break;
    default: *sc = -x; *tc = -y; *ma = -z; break;
    }
}
static inline float cube_lod(const float p[3], const float ddx[3], const float ddy[3], int S) {
    int face; float s, t;
    cube_face(p[0], p[1], p[2], &face, &s, &t);
    float sc, tc, ma, xsc, xtc, xma, ysc, ytc, yma;
    cube_components(face, p[0], p[1], p[2], &sc, &tc, &ma);
    cube_components(face, ddx[0], ddx[1], ddx[2], &xsc, &xtc, &xma);
    cube_components(face, ddy[0], ddy[1], ddy[2], &ysc, &ytc, &yma);
    float ima = 1.0f / ma, k = (ima * ima) * 0.5f;
    float dsx = (xsc * ma - sc * xma) * k, dtx = (xtc * ma - tc * xma) * k;
    float dsy = (ysc * ma - sc * yma) * k, dty = (ytc * ma - tc * yma) * k;
    float rho2 = fmaxf(dsx * dsx + dtx * dtx, dsy * dsy + dty * dty) * ((float)S * (float)S);
    uint32_t bits; memcpy(&bits, &rho2, 4);
    uint32_t mb = (bits & 0x7fffffu) | 0x3f800000u; float mant; memcpy(&mant, &mb, 4);
    return 0.5f * ((float)((int)((bits >> 23) & 0xffu) - 127) + (mant - 1.0f));
}
static inline void cube_pass_direction(int x, int y, int W, int H, float *d) {
    const float PI = 3.1415926535897932384626433832795f;
    float u = pert_uv(frag_u(x, y, W, H)), v = pert_uv(frag_v(y, W, H));
    float longitude = ((u * 2.0f) * PI - PI) + PI / 2.0f;
    float latitude = v * PI;
    float sl = sinf(latitude);
    d[0] = -sinf(longitude) * sl; d[1] = -cosf(latitude); d[2] = -cosf(longitude) * sl; /* dir.y = -dir.y */
}
/* the pass: W x H RGBA32F (FloatType render target), row y = vUv.y (row 0 = bottom, as readRenderTargetPixels returns it).
 * mipmaps = 0: the cube has level 0 only (minFilter LinearFilter); 1: LinearMipmapLinearFilter over the generated chain (S a power of two) */
int rfxo_cube_to_equirect(const float *faces, int S, int mipmaps, int W, int H, float *out) {
    if (!faces || !out || S < 1 || W < 1 || H < 1) return -1;
    const float *lv[16] = {faces};
    float *owned[16] = {0};
    int sizes[16] = {S}, levels = 1;
    while (mipmaps && sizes[levels - 1] > 1) {
        int s0 = sizes[levels - 1], s1 = s0 >> 1;
        float *dst = (float *)malloc((size_t)6 * s1 * s1 * 4 * sizeof(float));
        const float *src = lv[levels - 1];
        /* glGenerateMipmap on the oracle's GL is a LINEAR blit of every face on its own: a target texel fetches the source level at its centre,
         * CLAMP_TO_EDGE, no neighbouring face.  From an even size that is the 2x2 average below (weights exactly one half); from an ODD size
         * (faces that are not a power of two: 31 -> 15 -> 7 -> 3 -> 1, 12 -> 6 -> 3 -> 1) the general bilinear tap.  Round 6, measured:
         * tools/fuzz_aux_vs_reference_gl.py, every equirect texel of faces 1..48 inside the fp32 rule. */
        for (int f = 0; f < 6 && (s0 & 1); f++) {
            dims d0 = {s0, s0};
            for (int y = 0; y < s1; y++)
                for (int x = 0; x < s1; x++) {
                    v4 c = fetch_f4_linear(src + ((size_t)f * s0 * s0) * 4, d0, ((float)x + 0.5f) / (float)s1, ((float)y + 0.5f) / (float)s1);
                    float *o = dst + 4 * (((size_t)f * s1 + y) * s1 + x);
                    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
                }
        }
        for (int f = 0; f < 6 && !(s0 & 1); f++)
            for (int y = 0; y < s1; y++)
                for (int x = 0; x < s1; x++)
                    for (int k = 0; k < 4; k++) {
                        const float *q = src + ((size_t)f * s0 * s0) * 4;
                        float a = q[4 * ((size_t)(2 * y) * s0 + 2 * x) + k], b = q[4 * ((size_t)(2 * y) * s0 + 2 * x + 1) + k];
                        float c = q[4 * ((size_t)(2 * y + 1) * s0 + 2 * x) + k], e = q[4 * ((size_t)(2 * y + 1) * s0 + 2 * x + 1) + k];
                        dst[4 * (((size_t)f * s1 + y) * s1 + x) + k] = lerpf(0.5f, lerpf(0.5f, a, b), lerpf(0.5f, c, e));
                    }
        lv[levels] = owned[levels] = dst; sizes[levels] = s1; levels++;
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            pert_begin(x, y);
            float p[3];
            cube_pass_direction(x, y, W, H, p);
            float *o = out + ((size_t)y * W + x) * 4;
            if (levels == 1) { cube_linear(faces, S, p[0], p[1], p[2], o); continue; }
            /* the quad partners: the other pixel of the own row, the other pixel of the own column (a target is even-sized here; a
             * partner outside an odd-sized target is extrapolated by the rasteriser: its planes evaluate there as well) */
            float pr[3], pc[3], ddx[3], ddy[3];
            cube_pass_direction(x ^ 1, y, W, H, pr);
            cube_pass_direction(x, y ^ 1, W, H, pc);
            for (int k = 0; k < 3; k++) {
                ddx[k] = (x & 1) ? p[k] - pr[k] : pr[k] - p[k];
                ddy[k] = (y & 1) ? p[k] - pc[k] : pc[k] - p[k];
            }
            float lod = cube_lod(p, ddx, ddy, S);
            lod = fminf(fmaxf(lod, 0.0f), (float)(levels - 1));
            float fl = floorf(lod), f = lod - fl;
            int l0 = (int)fl, l1 = l0 + 1 > levels - 1 ? levels - 1 : l0 + 1;
            float c0[4], c1[4];
            cube_linear(lv[l0], sizes[l0], p[0], p[1], p[2], c0);
            cube_linear(lv[l1], sizes[l1], p[0], p[1], p[2], c1);
            for (int k = 0; k < 4; k++) o[k] = fmaf(f, c1[k] - c0[k], c0[k]);
        }
    for (int l = 1; l < levels; l++) free(owned[l]);
    return 0;
}

/* The mip chain of scene.environment as glGenerateMipmap builds it on the oracle's GL (measured on llvmpipe): level 0 = the texels in the
 * texture's type, every further level the 2x2 bilinear-centre average lerp(.5, lerp(.5,a,b), lerp(.5,c,d)) stored in that type
 * (half: RTZ when `rtz`, as llvmpipe).  `out` receives all levels back to back; returns the number of levels. */
int rfxo_env_build(const float *base, int w, int h, int half, int rtz, float *out) {
    if (w < 1 || h < 1 || (w & (w - 1)) || (h & (h - 1))) return RFX_EINVAL;
    for (size_t i = 0; i < (size_t)w * h * 4; i++) out[i] = half ? half_to_float(float_to_half_rne(base[i])) : base[i];
    int levels = 1;
    const float *src = out;
    while (w > 1 || h > 1) {
        int dw = w > 1 ? w >> 1 : 1, dh = h > 1 ? h >> 1 : 1, fx = w > dw ? 2 : 1, fy = h > dh ? 2 : 1;
        float *dst = (float *)src + 4 * (size_t)w * h;
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                int x0 = x * fx, x1 = x0 + fx - 1, y0 = y * fy, y1 = y0 + fy - 1;
                for (int k = 0; k < 4; k++) {
                    float a = src[4 * ((size_t)y0 * w + x0) + k], b = src[4 * ((size_t)y0 * w + x1) + k];
                    float cc = src[4 * ((size_t)y1 * w + x0) + k], e = src[4 * ((size_t)y1 * w + x1) + k];
                    float o = lerpf(0.5f, lerpf(0.5f, a, b), lerpf(0.5f, cc, e));
                    if (half) o = half_to_float(rtz ? float_to_half_rtz(o) : float_to_half_rne(o));
                    dst[4 * ((size_t)y * dw + x) + k] = o;
                }
            }
        src = dst; w = dw; h = dh; levels++;
    }
    return levels;
}

/* env: the chain rfxo_env_build made (NULL without USE_ENVMAP) */
int rfxo_ssgi(int W, int H, int y0, int y1, const float *depth, const uint32_t *gbuffer, const float *direct, const float *history,
              const uint8_t *blue, const rfx_ssgi_params *p, uint32_t *out, const float *env, int env_w, int env_h, int env_levels,
              const float *marginal, const float *conditional, float totalSumWhole, float totalSumDecimal) {
    if (p->mode != 0 && p->mode != 1) return RFX_EUNSUPPORTED;
    if (p->useEnvMap && !env) return RFX_ESTATE;
    if (p->importanceSampling && (!p->useEnvMap || !marginal || !conditional)) return RFX_ESTATE;
    /* resolutionScale (SSGIPass.js:52-57): `out` is then the (W*s) x (H*s) target, pitch W*s, and y0/y1 are rows of THAT target */
    const float rs = p->resolutionScale == 0.0f ? 1.0f : p->resolutionScale;
    const int oW = (int)((float)W * rs), oH = (int)((float)H * rs);
    if ((float)oW != (float)W * rs || (float)oH != (float)H * rs || oW < 1 || oH < 1) return RFX_EINVAL;
    k1_ctx c = {W, H, depth, gbuffer, direct, history, blue, p, 0, 0, 0, env, env_w, env_h, env_levels, oW, oH, marginal, conditional, totalSumWhole, totalSumDecimal};
    /* SSGIPass.js:84-87: JS doubles rounded to float uniforms */
    c.nearMulFar = (float)((double)p->camera.near_ * (double)p->camera.far_);
    c.farMinusNear = (float)((double)p->camera.far_ - (double)p->camera.near_);
    c.cameraFar = p->camera.far_;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < oW; x++) {
            if (g_pixel_mask && !g_pixel_mask[(size_t)y * oW + x]) continue;
            g_margin = 3.0e38f; g_margin_notap = 3.0e38f; g_fetch_rel = 0.0f; g_fetch_abs = 0.0f; g_fetch_uv = 0.0f; pert_begin(x, y);
            k1_pixel(&c, x, y, out + 4 * ((size_t)y * oW + x));
            if (g_margin_plane) g_margin_plane[(size_t)y * oW + x] = g_margin;
            if (g_margin_notap_plane) g_margin_notap_plane[(size_t)y * oW + x] = g_margin_notap;
        }
    return 0;
}

/* ==================================================================== K2: temporal_reproject.frag */
typedef struct {
    int W, H;
    const uint32_t *ssgi, *velocity; const void *hist[2];
    const rfx_temporal_params *p;
    float invW, invH;
} k2_ctx;

static inline float perspective_depth_to_view_z(float depth, float n, float f) { return (n * f) / ((f - n) * depth - f); }
static inline float depth_to_view_z(float depth, float n, float f, int perspective) { /* three <packing>, chosen by PERSPECTIVE_CAMERA */
    return perspective ? perspective_depth_to_view_z(depth, n, f) : depth * (n - f) - n;
}

/* getVelocityNormalDepth reproject.frag:97-105 */
static inline void k2_vnd(const k2_ctx *c, float u, float v, float *vx, float *vy, v3 *normal, float *depth) {
    dims d = {c->W, c->H};
    const uint32_t *t = fetch_u4(c->velocity, d, u, v);
    memcpy(vx, &t[0], 4); memcpy(vy, &t[1], 4);
    *normal = unpack_normal(t[2]);
    memcpy(depth, &t[3], 4);
}
/* screenSpaceToWorldSpace reproject.frag:21-28 */
static inline v3 ss_to_ws(float u, float v, float depth, const float *matWorld, const float *projInv) {
    v4 clip = mat_mul_v4(projInv, (u - 0.5f) * 2.0f, (v - 0.5f) * 2.0f, (depth - 0.5f) * 2.0f, 1.0f);
    v4 view = mat_mul_v4(matWorld, clip.x / clip.w, clip.y / clip.w, clip.z / clip.w, clip.w / clip.w);
    return V3(view.x, view.y, view.z);
}
/* validateReprojectedUV reproject.frag:130-167 (angleMix/lastViewAngle are dead) */
static float k2_validate(const k2_ctx *c, float ru, float rv, v3 worldPos, v3 worldNormal, float depth) {
    margin_cmp(ru, 0.0f, MARGIN_REL_SHORT); margin_cmp(ru, 1.0f, MARGIN_REL_SHORT);
    margin_cmp(rv, 0.0f, MARGIN_REL_SHORT); margin_cmp(rv, 1.0f, MARGIN_REL_SHORT);
    if (ru > 1.0f || ru < 0.0f || rv > 1.0f || rv < 0.0f) return 0.0f;
    float lvx, lvy, lastDepth; v3 lastN;
    k2_vnd(c, ru, rv, &lvx, &lvy, &lastN, &lastDepth); /* samples the CURRENT velocity texture */
    v3 lastWorldPos = ss_to_ws(ru, rv, lastDepth, c->p->prevCamera.matrixWorld, c->p->prevCamera.projectionMatrixInverse);
    float viewZ = fabsf(depth_to_view_z(depth, c->p->camera.near_, c->p->camera.far_, c->p->camera.isPerspective));
    float distFactor = 1.0f + 1.0f / (viewZ + 1.0f);
    float disoccl = 0.0f;
    v3 dp = sub3(worldPos, lastWorldPos);
    disoccl += length3(dp) / 10.0f * distFactor;                     /* worldDistanceDisocclusionCheck */
    disoccl += fabsf(dot3(dp, worldNormal)) / 20.0f * distFactor;    /* planeDistanceDisocclusionCheck */
    disoccl += fminf(1.0f - dot3(worldNormal, lastN), 1.0f) / 1.0f * distFactor; /* normalDisocclusionCheck */
    float conf = 1.0f - fminf(disoccl, 1.0f);
    conf = fmaxf(conf, 0.0f);
    return powf(conf, c->p->confidencePower);
}
/* one LINEAR tap of the history: RGBA16F (K3's target B, or a HalfFloatType framebuffer copy) or RGBA32F (FloatType copy) */
static inline v4 k2_hist_tap(const k2_ctx *c, const void *tex, dims d, float u, float v) {
    return c->p->historySource == 2 ? fetch_f4_linear((const float *)tex, d, u, v) : fetch_h4_linear((const uint16_t *)tex, d, u, v);
}
/* BiCubicCatmullRom5Tap reproject.frag:212-255 on the linear-filtered history texture */
static v4 k2_bicubic(const k2_ctx *c, const void *tex, float pu, float pv) {
    dims d = {c->W, c->H};
    float its[2] = {c->invW, c->invH}, P[2] = {pu, pv};
    float w0[2], w1[2], w2[2], w3[2], W0[2], W1[2], W2[2], S0[2], S1[2], S2[2];
    for (int k = 0; k < 2; k++) {
        float UV = P[k] / its[k];
        float tc = floorf(UV - 0.5f) + 0.5f;
        float f = UV - tc, f2 = f * f, f3 = f2 * f;
        w0[k] = f2 - 0.5f * (f3 + f);
        w1[k] = 1.5f * f3 - 2.5f * f2 + 1.0f;
        w3[k] = 0.5f * (f3 - f2);
        w2[k] = 1.0f - w0[k] - w1[k] - w3[k];
        W0[k] = w0[k]; W1[k] = w1[k] + w2[k]; W2[k] = w3[k];
        S0[k] = (tc - 1.0f) * its[k];
        S1[k] = (tc + w2[k] / W1[k]) * its[k];
        S2[k] = (tc + 2.0f) * its[k];
    }
    float sw[5] = {W1[0] * W0[1], W0[0] * W1[1], W1[0] * W1[1], W2[0] * W1[1], W1[0] * W2[1]};
    v4 Ct = k2_hist_tap(c, tex, d, S1[0], S0[1]);
    v4 Cl = k2_hist_tap(c, tex, d, S0[0], S1[1]);
    v4 Cc = k2_hist_tap(c, tex, d, S1[0], S1[1]);
    v4 Cr = k2_hist_tap(c, tex, d, S2[0], S1[1]);
    v4 Cb = k2_hist_tap(c, tex, d, S1[0], S2[1]);
    float wm = 1.0f / (sw[0] + sw[1] + sw[2] + sw[3] + sw[4]);
    v4 r;
    r.x = fmaxf(((((Ct.x * sw[0] + Cl.x * sw[1]) + Cc.x * sw[2]) + Cr.x * sw[3]) + Cb.x * sw[4]) * wm, 0.0f);
    r.y = fmaxf(((((Ct.y * sw[0] + Cl.y * sw[1]) + Cc.y * sw[2]) + Cr.y * sw[3]) + Cb.y * sw[4]) * wm, 0.0f);
    r.z = fmaxf(((((Ct.z * sw[0] + Cl.z * sw[1]) + Cc.z * sw[2]) + Cr.z * sw[3]) + Cb.z * sw[4]) * wm, 0.0f);
    r.w = fmaxf(((((Ct.w * sw[0] + Cl.w * sw[1]) + Cc.w * sw[2]) + Cr.w * sw[3]) + Cb.w * sw[4]) * wm, 0.0f);
    return r;
}
static inline v3 log1p3(v3 c, int on) { return on ? V3(logf(c.x + 1.0f), logf(c.y + 1.0f), logf(c.z + 1.0f)) : c; } /* transformColor */
static inline v3 expm13(v3 c, int on) { return on ? V3(expf(c.x) - 1.0f, expf(c.y) - 1.0f, expf(c.z) - 1.0f) : c; } /* undoColorTransform */

/* input texel i at (u,v) after unpack (DIFFUSE_SPECULAR) or raw */
static inline v4 k2_input_texel(const k2_ctx *c, float u, float v, int idx) {
    /* inputTexture may be smaller than the frame (K1 drawn with resolutionScale < 1): NEAREST at the full-resolution uv */
    dims d = {c->p->inputWidth > 0 ? c->p->inputWidth : c->W, c->p->inputHeight > 0 ? c->p->inputHeight : c->H};
    const uint32_t *t = fetch_u4(c->ssgi, d, u, v);
    if (c->p->inputType == 0) {
        v4 a, b; unpack_two_vec4(t, &a, &b);
        return idx ? b : a;
    }
    v4 r; memcpy(&r, t, 16); return r;
}

static void k2_pixel(const k2_ctx *c, int x, int y, float *out0, float *out1) {
    const rfx_temporal_params *p = c->p;
    const int tc = p->textureCount, lt = p->logTransform;
    float u = pert_uv(frag_u(x, y, c->W, c->H)), v = pert_uv(frag_v(y, c->W, c->H));
    float velx, vely, depth; v3 worldNormal;
    k2_vnd(c, u, v, &velx, &vely, &worldNormal, &depth);
    /* getTexels + preprocessInput temporal_reproject.frag:124-145 */
    v4 inp[2]; int sampled[2];
    for (int i = 0; i < tc; i++) {
        inp[i] = k2_input_texel(c, u, v, i);
        sampled[i] = inp[i].x >= 0.0f;
        v3 rgb = V3(fmaxf(inp[i].x, 0.0f), fmaxf(inp[i].y, 0.0f), fmaxf(inp[i].z, 0.0f));
        rgb = log1p3(rgb, lt);
        inp[i].x = rgb.x; inp[i].y = rgb.y; inp[i].z = rgb.z;
    }
    /* quad partners for fwidth */
    int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
    float fx0 = frag_u(qx0, y, c->W, c->H), fx1 = frag_u(qx1, y, c->W, c->H);
    float fy0 = frag_v(qy0, c->W, c->H), fy1 = frag_v(qy1, c->W, c->H);
    float tvx, tvy, dxa, dxb, dya, dyb; v3 nxa, nxb, nya, nyb;
    k2_vnd(c, fx0, v, &tvx, &tvy, &nxa, &dxa); k2_vnd(c, fx1, v, &tvx, &tvy, &nxb, &dxb);
    k2_vnd(c, u, fy0, &tvx, &tvy, &nya, &dya); k2_vnd(c, u, fy1, &tvx, &tvy, &nyb, &dyb);
    if (p->inputType != 1) { /* :188-193 */
        float fw = fabsf(dxb - dxa) + fabsf(dyb - dya);
        if (depth == 1.0f && fw == 0.0f) return; /* discard */
    }
    v3 fwn = V3(fabsf(nxb.x - nxa.x) + fabsf(nyb.x - nya.x), fabsf(nxb.y - nxa.y) + fabsf(nyb.y - nya.y), fabsf(nxb.z - nxa.z) + fabsf(nyb.z - nya.z));
    float curvature = length3(fwn); /* getCurvature reproject.frag:265-269 */
    v3 worldPos = ss_to_ws(u, v, depth, p->camera.matrixWorld, p->camera.projectionMatrixInverse);
    float rayLength = 0.0f, roughness = 1.0f; /* globals: roughness = 1. (reproject.frag:6) */
    if (p->inputType == 0) { rayLength = inp[1].w; roughness = clampf(inp[0].w, 0.0f, 1.0f); }
    else if (p->inputType == 2) { uint32_t b; memcpy(&b, &inp[0].w, 4); float rl, ro; unpack_half2(b, &rl, &ro); rayLength = rl; roughness = clampf(ro, 0.0f, 1.0f); }
    /* computeReprojectedUv temporal_reproject.frag:155-165 */
    float rd[3], rs[3] = {-1.0f, -1.0f, -1.0f};
    rd[0] = u - velx; rd[1] = v - vely;
    g_fetch_uv = 1.1920929e-7f; /* one subtraction at the magnitude of 1 */
    rd[2] = k2_validate(c, rd[0], rd[1], worldPos, worldNormal, depth);
    g_fetch_uv = 0.0f;
    if (p->inputType == 0 || p->inputType == 2) {
        /* reprojectHitPoint reproject.frag:169-193 */
        float hu, hv;
        margin_cmp(curvature, 0.05f, MARGIN_REL_CURVATURE);
        if (curvature > 0.05f || rayLength < 0.01f) { hu = -1.0f; hv = -1.0f; }
        else {
            v3 camPos = V3(p->camera.position[0], p->camera.position[1], p->camera.position[2]);
            v3 cameraRay = normalize3(sub3(worldPos, camPos));
            v3 hp = add3(camPos, mul3(cameraRay, rayLength));
            /* prevProjectionMatrix * prevViewMatrix * vec4(hp, 1): GLSL evaluates (A*B)*v */
            float PV[16];
            const float *A = p->prevCamera.projectionMatrix, *Bm = p->prevCamera.matrixWorldInverse;
            for (int col = 0; col < 4; col++)
                for (int row = 0; row < 4; row++)
                    PV[col * 4 + row] = ((A[0 * 4 + row] * Bm[col * 4 + 0] + A[1 * 4 + row] * Bm[col * 4 + 1]) + A[2 * 4 + row] * Bm[col * 4 + 2]) +
                                        A[3 * 4 + row] * Bm[col * 4 + 3];
            v4 r = mat_mul_v4(PV, hp.x, hp.y, hp.z, 1.0f);
            hu = pert_coord((r.x / r.w) * 0.5f + 0.5f, 4.0f * 5.9604645e-8f); hv = pert_coord((r.y / r.w) * 0.5f + 0.5f, 4.0f * 5.9604645e-8f);
            g_fetch_rel = MARGIN_REL_SHORT; /* the validation fetch sits at a projected point (normalize, two matrix products, a division) ... */
            g_fetch_uv = 4.0f * 1.1920929e-7f; /* ... whose ndc -> uv step rounds at the magnitude of 1 */
        }
        rs[0] = hu; rs[1] = hv;
        rs[2] = k2_validate(c, hu, hv, worldPos, worldNormal, depth);
        g_fetch_rel = 0.0f; g_fetch_uv = 0.0f;
        if (rs[0] == -1.0f) { rs[0] = rd[0]; rs[1] = rd[1]; rs[2] = rd[2]; }
    }
    float moveFactor = fminf((velx * velx + vely * vely) * 10000.0f, 1.0f);
    for (int i = 0; i < tc; i++) {
        int spec = p->reprojectSpecular[i] != 0;
        const float *uvc = spec ? rs : rd;
        /* reproject() temporal_reproject.frag:83-122 */
        v4 acc = k2_bicubic(c, c->hist[i], uvc[0], uvc[1]);
        v3 accrgb = log1p3(V3(acc.x, acc.y, acc.z), lt);
        float acca = acc.w;
        v3 inrgb = V3(inp[i].x, inp[i].y, inp[i].z);
        if (!sampled[i]) {
            inrgb = accrgb;
        } else {
            acca += 1.0f;
            v3 clamped = accrgb;
            int cr = (spec && roughness < 0.25f) ? 1 : 2;
            /* clampNeighborhood reproject.frag:83-95 + getNeighborhoodAABB :53-81 */
            v3 ic = expm13(inrgb, lt);
            v3 mn = ic, mx = ic;
            for (int ox = -cr; ox <= cr; ox++)
                for (int oy = -cr; oy <= cr; oy++) {
                    float nu = u + (float)ox * c->invW, nv = v + (float)oy * c->invH;
                    v4 t = k2_input_texel(c, nu, nv, (p->inputType == 0) ? spec : 0);
                    if (t.x >= 0.0f) {
                        mn = V3(fminf(t.x, mn.x), fminf(t.y, mn.y), fminf(t.z, mn.z));
                        mx = V3(fmaxf(t.x, mx.x), fmaxf(t.y, mx.y), fmaxf(t.z, mx.z));
                    }
                }
            mn = log1p3(mn, lt); mx = log1p3(mx, lt);
            clamped = V3(clampf(clamped.x, mn.x, mx.x), clampf(clamped.y, mn.y, mx.y), clampf(clamped.z, mn.z, mx.z));
            float r = spec ? roughness : 1.0f;
            float aggr = fminf(1.0f, uvc[2] * r);
            float ci = mixf(0.0f, fminf(1.0f, moveFactor * 50.0f + p->neighborhoodClampIntensity), aggr);
            v3 nc = mix3(accrgb, clamped, ci);
            float cd = fminf(length3(sub3(nc, accrgb)), 1.0f);
            acca *= 1.0f - cd;
            accrgb = nc;
        }
        /* accumulate() temporal_reproject.frag:42-79 */
        float conf = powf(uvc[2], p->confidencePower);
        float accumBlend = 1.0f - 1.0f / (acca + 1.0f);
        accumBlend = mixf(0.0f, accumBlend, conf);
        float maxValue = (p->fullAccumulate ? 1.0f : p->maxBlend) * p->keepData;
        if (p->inputType != 1) {
            const float rmax = 0.1f;
            if (spec && roughness >= 0.0f && roughness < rmax) {
                float mrv = mixf(0.0f, maxValue, roughness / rmax);
                maxValue = mixf(maxValue, mrv, fminf(100.0f * moveFactor, 1.0f));
            }
        }
        float m = fminf(accumBlend, maxValue);
        acca = 1.0f / (1.0f - m) - 1.0f;
        acca = fminf(65536.0f, acca);
        v3 o = expm13(mix3(inrgb, accrgb, m), lt);
        float *dst = i ? out1 : out0;
        dst[0] = o.x; dst[1] = o.y; dst[2] = o.z; dst[3] = acca;
        if (p->targetHalf) /* HalfFloatType render target (TemporalReprojectPass.js:63-68): the store rounds to half */
            for (int k = 0; k < 4; k++) dst[k] = half_to_float(p->halfStoreRTZ ? float_to_half_rtz(dst[k]) : float_to_half_rne(dst[k]));
    }
}

int rfxo_temporal(int W, int H, int y0, int y1, const uint32_t *ssgi, const uint32_t *velocity, const void *hist0, const void *hist1,
                  const rfx_temporal_params *p, float *out0, float *out1) {
    if (p->historySource < 0 || p->historySource > 2) return RFX_EINVAL;
    k2_ctx c = {W, H, ssgi, velocity, {hist0, hist1}, p, 0, 0};
    /* TemporalReprojectPass.js:135: invTexSize.set(1 / width, 1 / height) — JS doubles -> float */
    c.invW = (float)(1.0 / (double)W); c.invH = (float)(1.0 / (double)H);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t o = 4 * ((size_t)y * W + x);
            if (g_pixel_mask && !g_pixel_mask[(size_t)y * W + x]) continue;
            g_margin = 3.0e38f; g_margin_notap = 3.0e38f; g_fetch_rel = 0.0f; g_fetch_abs = 0.0f; g_fetch_uv = 0.0f; pert_begin(x, y);
            k2_pixel(&c, x, y, out0 + o, out1 ? out1 + o : NULL);
            if (g_margin_plane) g_margin_plane[(size_t)y * W + x] = g_margin;
            if (g_margin_notap_plane) g_margin_notap_plane[(size_t)y * W + x] = g_margin_notap;
        }
    return 0;
}

/* ==================================================================== K3: poisson_denoise.frag */
typedef struct {
    int W, H;
    const float *depth; const uint32_t *gbuffer;
    const void *in[2]; int in_half; /* in_half: RGBA16F linear, else RGBA32F nearest */
    const uint8_t *blue; const rfx_denoise_params *p;
} k3_ctx;

static inline v4 k3_input(const k3_ctx *c, int idx, float u, float v) {
    dims d = {c->W, c->H};
    if (c->in_half) return fetch_h4_linear((const uint16_t *)c->in[idx], d, u, v);
    return fetch_f4((const float *)c->in[idx], d, u, v);
}
static inline float k3_lum(v3 a) { return powf(lum(a), 0.125f); } /* #define luminance(a) pow(dot(..), 0.125) :28 */

static void k3_pixel(const k3_ctx *c, int x, int y, uint16_t *out0, uint16_t *out1) {
    const rfx_denoise_params *p = c->p;
    dims d = {c->W, c->H};
    const int tc = p->textureCount;
    float u = pert_uv(frag_u(x, y, c->W, c->H)), v = pert_uv(frag_v(y, c->W, c->H));
    float depth = fetch_r32f(c->depth, d, u, v);
    int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
    float fx0 = frag_u(qx0, y, c->W, c->H), fx1 = frag_u(qx1, y, c->W, c->H);
    float fy0 = frag_v(qy0, c->W, c->H), fy1 = frag_v(qy1, c->W, c->H);
    {
        float fw = fabsf(fetch_r32f(c->depth, d, fx1, v) - fetch_r32f(c->depth, d, fx0, v)) +
                   fabsf(fetch_r32f(c->depth, d, u, fy1) - fetch_r32f(c->depth, d, u, fy0));
        if (depth == 1.0f && fw == 0.0f) return; /* discard :129-132 */
    }
    v3 rgb[2]; float a[2], L[2], w_age[2], tw[2]; int isSpec[2];
    for (int i = 0; i < tc; i++) { /* :137-165 */
        isSpec[i] = p->isTextureSpecular[i] != 0;
        v4 t = k3_input(c, isSpec[i] ? 1 : 0, u, v);
        float age = 1.0f / powf(t.w + 1.0f, 1.2f * p->phi);
        v3 col = V3(t.x * 1.0003f, t.y * 1.0003f, t.z * 1.0003f);
        col = V3(logf(col.x + 1.0f), logf(col.y + 1.0f), logf(col.z + 1.0f));
        rgb[i] = col; a[i] = t.w; L[i] = k3_lum(col); w_age[i] = age; tw[i] = 1.0f;
    }
    material mat = get_material(fetch_u4(c->gbuffer, d, u, v));
    v3 normal = mat.normal;
    float glossiness = fmaxf(0.0f, 4.0f * (1.0f - mat.roughness / 0.25f));
    float specularFactor = expf(-glossiness * p->specularPhi);
    /* fwidth(normal) over the quad */
    v3 nxa = get_material(fetch_u4(c->gbuffer, d, fx0, v)).normal, nxb = get_material(fetch_u4(c->gbuffer, d, fx1, v)).normal;
    v3 nya = get_material(fetch_u4(c->gbuffer, d, u, fy0)).normal, nyb = get_material(fetch_u4(c->gbuffer, d, u, fy1)).normal;
    v3 fwn = V3(fabsf(nxb.x - nxa.x) + fabsf(nyb.x - nya.x), fabsf(nxb.y - nxa.y) + fabsf(nyb.y - nya.y), fabsf(nxb.z - nxa.z) + fabsf(nyb.z - nya.z));
    float flatness = 1.0f - fminf(length3(fwn), 1.0f);
    flatness = (flatness * flatness) * 0.75f + 0.25f;
    v4 random = blue_noise(c->blue, x, y, p->blueNoiseIndex, u, v, d);
    float r = p->radius;
    float angle = random.x * 2.0f * 3.141592653589793f;
    /* 256 possible angles (an 8-bit blue-noise channel): the correctly rounded values, which is what the kernel's table holds
     * (csrc/k3_rotation_table.h) and what the reference GL returns at the two angles that decide taps (bytes 85, 170) */
    float s, co;
    if (g_k3_rotation_libm) { s = sinf(angle); co = cosf(angle); }  /* (the macros above perturb these too) */
    else { s = pert_ang((float)sin((double)angle)); co = pert_ang((float)cos((double)angle)); }
    /* mat2 rm = r * flatness * mat2(c, -s, s, c): columns (c,-s), (s,c) */
    float rf = r * flatness;
    float m00 = rf * co, m01 = rf * -s, m10 = rf * s, m11 = rf * co; /* m<col><row> */
    static const float SQ = 0.25f * 1.41421356237f;
    const float POI[8][2] = {{-1, 0}, {0, -1}, {1, 0}, {0, 1}, {-SQ, -SQ}, {SQ, -SQ}, {SQ, SQ}, {-SQ, SQ}};
    for (int k = 0; k < 8; k++) {
        float ox = POI[k][0] / (float)c->W, oy = POI[k][1] / (float)c->H;
        float nu = u + (m00 * ox + m10 * oy), nv = v + (m01 * ox + m11 * oy);
        /* the tap offset is r * flatness * (cos, sin)(angle) * POISSON[k]: sin/cos/sqrt carry ~1e-6 of an offset of up to `reach` texels */
        g_fetch_abs = 4e-6f * (r * fmaxf((float)c->W / (float)c->H, (float)c->H / (float)c->W) + 1.0f);
        /* getBasicNeighborWeight :52-78 */
        float wBasic;
        {
            material nm = get_material(fetch_u4(c->gbuffer, d, nu, nv));
            float nd = fetch_r32f(c->depth, d, nu, nv);
            if (nd == 1.0f) wBasic = 0.0f;
            else {
                float normalDiff = 1.0f - fmaxf(dot3(normal, nm.normal), 0.0f);
                float depthDiff = 10000.0f * fabsf(depth - nd);
                float roughDiff = fabsf(mat.roughness - nm.roughness);
                wBasic = expf(-normalDiff * p->normalPhi - depthDiff * p->depthPhi - roughDiff * p->roughnessPhi);
            }
        }
        for (int i = 0; i < tc; i++) { /* applyWeight :102-124 */
            float w = wBasic;
            v4 t = k3_input(c, isSpec[i] ? 1 : 0, nu, nv);
            if (isSpec[i]) w *= specularFactor;
            v3 tl = V3(logf(t.x + 1.0f), logf(t.y + 1.0f), logf(t.z + 1.0f));
            float disocclW = powf(w, 0.1f);
            float lumaDiff = fminf(fabsf(L[i] - k3_lum(tl)), 0.5f);
            float lumaFactor = expf(-lumaDiff * p->lumaPhi);
            w = mixf(w * lumaFactor, disocclW, w_age[i]) * w_age[i];
            margin_cmp(w, 0.0001f, MARGIN_REL_WEIGHT);
            w *= (w < 0.0001f) ? 0.0f : 1.0f; /* step(0.0001, w) */
            rgb[i] = add3(rgb[i], mul3(tl, w));
            tw[i] += w;
        }
    }
    for (int i = 0; i < tc; i++) { /* outputTexel :94-100 */
        v3 o = V3(rgb[i].x / tw[i], rgb[i].y / tw[i], rgb[i].z / tw[i]);
        o = V3(expf(o.x) - 1.0f, expf(o.y) - 1.0f, expf(o.z) - 1.0f);
        uint16_t *dst = i ? out1 : out0;
        if (p->halfStoreRTZ) { dst[0] = float_to_half_rtz(o.x); dst[1] = float_to_half_rtz(o.y); dst[2] = float_to_half_rtz(o.z); dst[3] = float_to_half_rtz(a[i]); }
        else { dst[0] = float_to_half_rne(o.x); dst[1] = float_to_half_rne(o.y); dst[2] = float_to_half_rne(o.z); dst[3] = float_to_half_rne(a[i]); }
    }
}

int rfxo_denoise(int W, int H, int y0, int y1, const float *depth, const uint32_t *gbuffer, const void *in0, const void *in1, int in_is_half,
                 const uint8_t *blue, const rfx_denoise_params *p, uint16_t *out0, uint16_t *out1) {
    k3_ctx c = {W, H, depth, gbuffer, {in0, in1}, in_is_half, blue, p};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t o = 4 * ((size_t)y * W + x);
            if (g_pixel_mask && !g_pixel_mask[(size_t)y * W + x]) continue;
            g_margin = 3.0e38f; g_margin_notap = 3.0e38f; g_fetch_rel = 0.0f; g_fetch_abs = 0.0f; g_fetch_uv = 0.0f; pert_begin(x, y);
            k3_pixel(&c, x, y, out0 + o, out1 ? out1 + o : NULL);
            if (g_margin_plane) g_margin_plane[(size_t)y * W + x] = g_margin;
            if (g_margin_notap_plane) g_margin_notap_plane[(size_t)y * W + x] = g_margin_notap;
        }
    return 0;
}

/* ==================================================================== K4: DenoiserComposePass */
/* gi0/gi1: K3 target B ([0] = diffuse, [1] = specular; inputType "specular": gi0 is the specular GI, gi1 unused);
 * scene: the composer's input buffer (sceneTexture), only read when inputType == TYPE_SPECULAR */
/* diagnostic (tools/k4_error_tail.py): when set, every composed pixel writes (|v + l| before the half vector's normalisation :90, VoH, |reflect(-V, H)|
 * before ITS normalisation :77, dot(viewNormal, l) before the sign test :87) into this W x H x 4 plane */
static float *g_compose_probe = NULL;
void rfxo_set_compose_probe(float *plane) { g_compose_probe = plane; }
int rfxo_compose(int W, int H, int y0, int y1, const float *depth, const uint32_t *gbuffer, const void *gi0v, const void *gi1v,
                 const float *scene, const rfx_compose_params *p, float *out) {
    const uint16_t *gi0 = (const uint16_t *)gi0v, *gi1 = (const uint16_t *)gi1v; /* giSource 0: RGBA16F linear; 1: RGBA32F nearest (K2's targets) */
    if (p->inputType != 0 && p->inputType != 2) return RFX_EUNSUPPORTED;
    if (p->inputType == 2 && !scene) return RFX_EINVAL;
    const float *C = p->camera.matrixWorld, *Vw = p->camera.matrixWorldInverse, *P = p->camera.projectionMatrix, *Pi = p->camera.projectionMatrixInverse;
    dims d = {W, H};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            if (g_pixel_mask && !g_pixel_mask[(size_t)y * W + x]) continue;
            g_margin = 3.0e38f; g_margin_notap = 3.0e38f; g_fetch_rel = 0.0f; g_fetch_abs = 0.0f; g_fetch_uv = 0.0f; pert_begin(x, y);
            if (g_margin_plane) g_margin_plane[(size_t)y * W + x] = g_margin;
            if (g_margin_notap_plane) g_margin_notap_plane[(size_t)y * W + x] = g_margin_notap;
            float u = pert_uv(frag_u(x, y, W, H)), v = pert_uv(frag_v(y, W, H));
            float dep = fetch_r32f(depth, d, u, v);
            int qx0 = x & ~1, qx1 = x | 1, qy0 = y & ~1, qy1 = y | 1;
            float fx0 = frag_u(qx0, y, W, H), fx1 = frag_u(qx1, y, W, H);
            float fy0 = frag_v(qy0, W, H), fy1 = frag_v(qy1, W, H);
            float fw = fabsf(fetch_r32f(depth, d, fx1, v) - fetch_r32f(depth, d, fx0, v)) + fabsf(fetch_r32f(depth, d, u, fy1) - fetch_r32f(depth, d, u, fy0));
            if (dep == 1.0f && fw == 0.0f) continue; /* discard DenoiserComposePass.js:61-64 */
            material mat = get_material(fetch_u4(gbuffer, d, u, v));
            v3 viewNormal = v4_mul_mat_xyz(C, mat.normal, 0.0f); /* :71, not normalised */
            float viewZ = -depth_to_view_z(dep, p->camera.near_, p->camera.far_, p->camera.isPerspective); /* :73 */
            /* getViewPosition denoiser_compose_functions.glsl:13-20 */
            float clipW = P[2 * 4 + 3] * viewZ + P[3 * 4 + 3];
            v4 pp = mat_mul_v4(Pi, ((u - 0.5f) * 2.0f) * clipW, ((v - 0.5f) * 2.0f) * clipW, ((viewZ - 0.5f) * 2.0f) * clipW, 1.0f * clipW);
            v3 viewPos = V3(pp.x, pp.y, -viewZ);
            v3 viewDir = normalize3(viewPos);
            /* DenoiserComposePass.js:26-33,78-79: diffuseSpecular -> (textures[0], textures[1]); specular -> specularGi = textures[0],
               diffuseGiTexture unbound (zeros) */
            v4 dgi = {0, 0, 0, 0}, sgi;
            if (p->giSource) {
                if (p->inputType == 0) { dgi = fetch_f4((const float *)gi0v, d, u, v); sgi = fetch_f4((const float *)gi1v, d, u, v); }
                else sgi = fetch_f4((const float *)gi0v, d, u, v);
            } else if (p->inputType == 0) { dgi = fetch_h4_linear(gi0, d, u, v); sgi = fetch_h4_linear(gi1, d, u, v); }
            else sgi = fetch_h4_linear(gi0, d, u, v);
            /* constructGlobalIllumination :53-108 */
            float roughness = mat.roughness * mat.roughness;
            v3 normal = v4_mul_mat_xyz(Vw, viewNormal, 0.0f);
            v3 vv = neg3(viewDir);
            v3 V = v4_mul_mat_xyz(Vw, vv, 0.0f);
            v3 N = normal, T, B;
            onb(N, &T, &B);
            V = V3(dot3(V, T), dot3(V, B), dot3(V, N));
            v3 Hh = sample_ggx_vndf(V, roughness, roughness, 0.25f, 0.25f);
            margin_note(fabsf(Hh.z) / MARGIN_REL_SHORT);
            if (Hh.z < 0.0f) Hh = neg3(Hh);
            v3 I = neg3(V);
            const v3 refl = sub3(I, mul3(Hh, 2.0f * dot3(Hh, I)));
            v3 l = normalize3(refl);
            l = add3(add3(mul3(T, l.x), mul3(B, l.y)), mul3(N, l.z));
            l = normalize3(v4_mul_mat_xyz(C, l, 1.0f)); /* vec4(l, 1.) quirk :81 */
            margin_note(fabsf(dot3(viewNormal, l)) / (MARGIN_REL_SHORT * fmaxf(length3(viewNormal), 1e-30f)));
            const float nl = dot3(viewNormal, l);
            if (dot3(viewNormal, l) < 0.0f) l = neg3(l);
            v3 h = normalize3(add3(vv, l));
            float VoH = fmaxf(1e-6f, dot3(vv, h)); /* EPSILON from <common> */
            if (g_compose_probe) {
                float *q = g_compose_probe + 4 * ((size_t)y * W + x);
                q[0] = length3(add3(vv, l)); q[1] = VoH; q[2] = length3(refl); q[3] = nl;
            }
            v3 f0 = mix3(V3(0.04f, 0.04f, 0.04f), mat.diffuse, mat.metalness);
            v3 F = f_schlick3(f0, VoH);
            float om = 1.0f - mat.metalness;
            v3 diffuseC = V3(mat.diffuse.x * om * (1.0f - F.x) * dgi.x, mat.diffuse.y * om * (1.0f - F.y) * dgi.y, mat.diffuse.z * om * (1.0f - F.z) * dgi.z);
            if (p->inputType == 2) { /* denoiser_compose_functions.glsl:97-101: diffuseComponent = textureLod(sceneTexture, vUv, 0.).rgb */
                v4 sc = fetch_f4(scene, d, u, v);
                diffuseC = V3(sc.x, sc.y, sc.z);
            }
            v3 specC = V3(sgi.x * F.x, sgi.y * F.y, sgi.z * F.z);
            float *o = out + 4 * ((size_t)y * W + x);
            o[0] = diffuseC.x + specC.x + mat.emissive.x;
            o[1] = diffuseC.y + specC.y + mat.emissive.y;
            o[2] = diffuseC.z + specC.z + mat.emissive.z;
            o[3] = 1.0f;
            if (g_margin_plane) g_margin_plane[(size_t)y * W + x] = g_margin;
            if (g_margin_notap_plane) g_margin_notap_plane[(size_t)y * W + x] = g_margin_notap;
        }
    return 0;
}

/* exported helpers for unit tests of the codec / conversions */
/* ==================================================================== SSGIEffect's own fragment: ssgi_compose.frag:20-45
 * gi = the denoiser's texture (K4 output, `inputTexture`), scene = the composer's input buffer (`sceneTexture`).
 * Fog: three.js fog_fragment (un-vendored, three@0.151: SURVEY.md Appendix H) with its gl_FragColor line removed (SSGIEffect.js:40-44):
 *   FOG_EXP2: fogFactor = 1.0 - exp( - fogDensity * fogDensity * vFogDepth * vFogDepth );  else smoothstep( fogNear, fogFar, vFogDepth ) */
int rfxo_final(int W, int H, int y0, int y1, const float *depth, const void *giv, const float *scene, const rfx_final_params *p, float *out) {
    if (p->fogMode < 0 || p->fogMode > 2 || p->inputSource < 0 || p->inputSource > 2) return RFX_EINVAL;
    const float *gi = (const float *)giv;
    float *tmp = NULL;
    if (p->inputSource == 2) { /* "denoised": K3's RGBA16F target B sampled at texel centres = the texel */
        tmp = (float *)malloc((size_t)W * H * 4 * sizeof(float));
        for (size_t k = 0; k < (size_t)W * H * 4; k++) tmp[k] = half_to_float(((const uint16_t *)giv)[k]);
        gi = tmp;
    }
#pragma omp parallel for schedule(static)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            float *o = out + 4 * i;
            if (p->isDebug) { memcpy(o, gi + 4 * i, 16); continue; }
            float dep = depth[i];
            float c[3];
            if (dep == 1.0f) { c[0] = scene[4 * i]; c[1] = scene[4 * i + 1]; c[2] = scene[4 * i + 2]; }
            else {
                c[0] = gi[4 * i]; c[1] = gi[4 * i + 1]; c[2] = gi[4 * i + 2];
                if (p->fogMode) {
                    float viewZ = depth_to_view_z(dep, p->camera.near_, p->camera.far_, p->camera.isPerspective) * 0.4f;
                    float fd = -viewZ, ff;
                    if (p->fogMode == 2) ff = 1.0f - expf(-p->fogDensity * p->fogDensity * fd * fd);
                    else { float t = fminf(fmaxf((fd - p->fogNear) / (p->fogFar - p->fogNear), 0.0f), 1.0f); ff = t * t * (3.0f - 2.0f * t); }
                    for (int k = 0; k < 3; k++) c[k] = mixf(c[k], p->fogColor[k], ff);
                }
            }
            o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = 1.0f;
        }
    free(tmp);
    return 0;
}

uint16_t rfxo_f2h_rne(float f) { return float_to_half_rne(f); }
uint16_t rfxo_f2h_rtz(float f) { return float_to_half_rtz(f); }
float rfxo_h2f(uint16_t h) { return half_to_float(h); }
int rfxo_nearest_idx(float u, int size) { return nearest_idx(u, size); }
void rfxo_get_material(const uint32_t *g, float *out12) {
    material m = get_material(g);
    out12[0] = m.diffuse.x; out12[1] = m.diffuse.y; out12[2] = m.diffuse.z; out12[3] = m.alpha;
    out12[4] = m.normal.x; out12[5] = m.normal.y; out12[6] = m.normal.z;
    out12[7] = m.roughness; out12[8] = m.metalness;
    out12[9] = m.emissive.x; out12[10] = m.emissive.y; out12[11] = m.emissive.z;
}
void rfxo_blue_noise(const uint8_t *table, int px, int py, int index, float *out4) {
    dims d = {128, 128};
    v4 r = blue_noise(table, px, py, index, 0, 0, d);
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
}
void rfxo_pack_two_vec4(const float *a, const float *b, uint32_t *out) {
    v4 A = {a[0], a[1], a[2], a[3]}, Bv = {b[0], b[1], b[2], b[3]};
    pack_two_vec4(A, Bv, out);
}
