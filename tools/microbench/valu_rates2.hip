// Development microbenchmark #2 (GPU box): issue cost of the instruction forms the round-3 instruction diet chooses between
// (v_fma_mix_f32 vs v_cvt_f32_f16 + fma, VOP2 v_fmac vs VOP3 v_fma, integer address ops, selects, LDS read forms).
//   hipcc --offload-arch=gfx950 -O3 valu_rates2.hip -o bin/valu_rates2 && bin/valu_rates2
// Each test runs 8 independent dependency chains per lane, 4096 iterations, 8 waves per SIMD on every CU.  Printed: cycles per wave64
// instruction per SIMD at the clock the run sustained (s_memtime-free: wall time x 2.4 GHz, so a lower sustained clock shows up as
// proportionally more "cycles" for every row alike — compare rows, not absolutes).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096

#define DEF_KERNEL(NAME, ASM_STR, ...)                                                     \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                 \
        float a[8];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-6f + i; \
        float b = seed * 0.5f, c = seed * 0.25f;                                           \
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cmp_gt_f32_e64 s[10:11], %0, %1" ::"v"(b), "v"(c) : "vcc", "s10", "s11"); \
        for (int it = 0; it < ITERS; it++) {                                               \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM_STR : "+v"(a[i]) : "v"(b), "v"(c)__VA_ARGS__); } \
        }                                                                                  \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i];                           \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
    }

// %0 = chain register (in/out), %1 = b, %2 = c
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1")
DEF_KERNEL(k_sub, "v_sub_f32 %0, %0, %1")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %0, %1")
DEF_KERNEL(k_add_e64, "v_add_f32_e64 %0, %0, -%1")
DEF_KERNEL(k_mul_e64abs, "v_mul_f32_e64 %0, |%0|, %1")
DEF_KERNEL(k_add_lit, "v_add_f32 %0, 0x3f8ccccd, %0")
DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_fmamk, "v_fmamk_f32 %0, %0, 0x3f372474, %1")
DEF_KERNEL(k_fma_mix_f32in, "v_fma_mix_f32 %0, %0, %1, %2")
DEF_KERNEL(k_fma_mix_h, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]")
DEF_KERNEL(k_fma_mix_hh, "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,1,1] op_sel_hi:[0,1,1]")
DEF_KERNEL(k_cvt_f16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL(k_cvt_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
DEF_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL(k_cmp, "v_cmp_gt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %2", : "vcc")
DEF_KERNEL(k_cmp_e64, "v_cmp_gt_f32_e64 s[10:11], %0, %1\n\tv_add_f32 %0, %0, %2", : "s10", "s11")
DEF_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEF_KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
DEF_KERNEL(k_min_i32, "v_min_i32 %0, %0, %1")
DEF_KERNEL(k_med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEF_KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
DEF_KERNEL(k_floor, "v_floor_f32 %0, %0")
DEF_KERNEL(k_mov, "v_mov_b32 %0, %1")
DEF_KERNEL(k_and, "v_and_b32 %0, %0, %1")
DEF_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 8")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
DEF_KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1")
DEF_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL(k_cvt_i32, "v_cvt_i32_f32 %0, %0")
DEF_KERNEL(k_exp, "v_exp_f32 %0, %0")
DEF_KERNEL(k_dpp_add, "v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
// select idioms (round 3: v_cndmask_b32 measured ~22 cycles with a loop-invariant VCC — which form of a select is cheap?)
DEF_KERNEL(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", : "s10", "s11")
DEF_KERNEL(k_cndmask_src2, "v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL(k_cndmask_lit, "v_cndmask_b32 %0, 0, %0, vcc")
DEF_KERNEL(k_cmp_cndmask, "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", : "vcc")
DEF_KERNEL(k_cmp_add_cndmask, "v_cmp_gt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", : "vcc")
DEF_KERNEL(k_cmp_e64_cndmask_e64, "v_cmp_gt_f32_e64 s[10:11], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[10:11]", : "s10", "s11")
DEF_KERNEL(k_sel_bits, "v_sub_u32 %0, %1, %0\n\tv_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %2")
DEF_KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
DEF_KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_fract, "v_fract_f32 %0, %0")
DEF_KERNEL(k_ashr, "v_ashrrev_i32 %0, 3, %0")
DEF_KERNEL(k_fma_same, "v_fma_f32 %0, %0, %1, %1")
DEF_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEF_KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %2", : "vcc")
DEF_KERNEL(k_mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
// mixes: does a transcendental overlap with plain VALU of the SAME wave / other waves?  (trans + 3 add) vs the sum of parts
DEF_KERNEL(k_mix_exp_3add, "v_exp_f32 %0, %0\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1")
DEF_KERNEL(k_mix_exp_3fma, "v_exp_f32 %0, %0\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mix_add_fma, "v_add_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mix_add_cvt, "v_add_f32 %0, %0, %1\n\tv_cvt_i32_f32 %0, %0")

typedef float f2_t __attribute__((ext_vector_type(2)));
#define DEF_PK(NAME, ASM_STR)                                                              \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                 \
        f2_t a[8];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = (f2_t){seed + threadIdx.x * 1e-6f + i, seed + i * 0.5f}; \
        const f2_t m = (f2_t){1.0000001f, 0.9999999f}, c = (f2_t){1e-7f, 2e-7f};           \
        for (int it = 0; it < ITERS; it++) {                                               \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM_STR : "+v"(a[i]) : "v"(m), "v"(c)); } \
        }                                                                                  \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;                \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
    }
DEF_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_PK(k_pk_fma_bcast, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]")
DEF_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEF_PK(k_pk_add, "v_pk_add_f32 %0, %0, %1")
DEF_PK(k_pk_add_neg, "v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]")
DEF_PK(k_pk_mov, "v_pk_mov_b32 %0, %1, %2")

// LDS read forms: 512 threads / workgroup, each lane reads `n` values per iteration at a per-lane pseudo-random texel of a
// 1248-texel tile (the K3 footprint) or at consecutive texels; value is folded into the address chain so nothing is hoisted
template <int FORM, bool RANDOM>
__global__ __launch_bounds__(512) void k_lds(float *out, int seed) {
    __shared__ uint4 tile[2560];  // 40 KiB
    for (int i = threadIdx.x; i < 2560; i += 512) tile[i] = make_uint4(i * 7 + seed, i * 13, i * 29, i * 31);
    __syncthreads();
    unsigned int idx = RANDOM ? (threadIdx.x * 2654435761u + seed) % 1248u : threadIdx.x % 1248u;
    unsigned int acc = 0;
    const char *base = (const char *)tile;
    for (int it = 0; it < 1024; it++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned int o = ((idx + j * 79u) % 1248u);
            if (FORM == 0) {  // one 8-byte texel
                const uint2 v = *(const uint2 *)(base + o * 8u);
                acc += v.x ^ v.y;
            } else if (FORM == 1) {  // two adjacent 8-byte texels as ds_read2_b64 (or whatever the compiler picks for a 16-byte unaligned pair)
                const uint2 v = *(const uint2 *)(base + o * 8u), w = *(const uint2 *)(base + o * 8u + 8u);
                acc += v.x ^ v.y ^ w.x ^ w.y;
            } else if (FORM == 2) {  // one aligned 16-byte read
                const uint4 v = *(const uint4 *)(base + o * 16u);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            } else {  // 4-byte
                acc += *(const unsigned int *)(base + o * 4u);
            }
        }
        idx = RANDOM ? (idx * 1664525u + 1013904223u + (acc & 1u)) % 1248u : (idx + 64u + (acc & 1u)) % 1248u;
    }
    out[blockIdx.x * 512 + threadIdx.x] = (float)acc;
}

template <typename K>
static void run_kernel(const char *name, K kern, int waves_per_simd, int instr_per_body) {
    int blocks = 256 * waves_per_simd;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)ITERS * 8 * waves_per_simd;  // bodies per SIMD
    printf("%-34s %d w/SIMD %.3f ms  %6.2f cycles per body of %d instr  (%.2f per instr)\n", name, waves_per_simd, ms, (ms * 1e6) * 2.4 / per_simd, instr_per_body,
           (ms * 1e6) * 2.4 / per_simd / instr_per_body);
    hipFree(out);
}
template <typename K>
static void run_lds(const char *name, K kern, int bytes_per_read) {
    int blocks = 256 * 3;  // three 512-thread workgroups per CU
    float *out;
    hipMalloc(&out, (size_t)blocks * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    kern<<<blocks, 512>>>(out, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 512>>>(out, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_reads_per_cu = 3.0 * 8 * 1024 * 4;  // wave-level read groups per CU
    printf("%-34s %.3f ms  %6.1f cycles per wave-level read group per CU  (%d B/lane)\n", name, ms, ms * 1e6 * 2.4 / wave_reads_per_cu, bytes_per_read);
    hipFree(out);
}
#define RUN(k, n) run_kernel(#k, k, w, n)
int main() {
    for (int w : {8, 4}) {
        RUN(k_add, 1); RUN(k_sub, 1); RUN(k_mul, 1); RUN(k_add_e64, 1); RUN(k_mul_e64abs, 1); RUN(k_add_lit, 1);
        RUN(k_fma, 1); RUN(k_fmac, 1); RUN(k_fmamk, 1); RUN(k_fma_mix_f32in, 1); RUN(k_fma_mix_h, 1); RUN(k_fma_mix_hh, 1);
        RUN(k_cvt_f16, 1); RUN(k_cvt_f16_sdwa, 1); RUN(k_cndmask, 1); RUN(k_cmp, 2); RUN(k_cmp_e64, 2);
        RUN(k_add_u32, 1); RUN(k_mad_i24, 1); RUN(k_lshl_add, 1); RUN(k_add_lshl, 1); RUN(k_min_i32, 1); RUN(k_med3_i32, 1); RUN(k_min_f32, 1);
        RUN(k_floor, 1); RUN(k_mov, 1); RUN(k_and, 1); RUN(k_bfe, 1); RUN(k_perm, 1); RUN(k_pkrtz, 1); RUN(k_ldexp, 1); RUN(k_mul_lo, 1);
        RUN(k_cvt_i32, 1); RUN(k_exp, 1); RUN(k_dpp_add, 1);
        RUN(k_cndmask_e64, 1); RUN(k_cndmask_src2, 1); RUN(k_cndmask_lit, 1); RUN(k_cmp_cndmask, 2); RUN(k_cmp_add_cndmask, 4); RUN(k_cmp_e64_cndmask_e64, 2);
        RUN(k_sel_bits, 3); RUN(k_max_f32, 1); RUN(k_med3_f32, 1); RUN(k_fract, 1); RUN(k_ashr, 1); RUN(k_fma_same, 1); RUN(k_bfi, 1); RUN(k_cmp_class, 2);
        RUN(k_mul_legacy, 1);
        RUN(k_mix_exp_3add, 4); RUN(k_mix_exp_3fma, 4); RUN(k_mix_add_fma, 2); RUN(k_mix_add_cvt, 2);
        RUN(k_pk_fma, 1); RUN(k_pk_fma_bcast, 1); RUN(k_pk_mul, 1); RUN(k_pk_add, 1); RUN(k_pk_add_neg, 1); RUN(k_pk_mov, 1);
    }
    run_lds("lds b64 random", k_lds<0, true>, 8);
    run_lds("lds b64 linear", k_lds<0, false>, 8);
    run_lds("lds 2 x b64 adjacent random", k_lds<1, true>, 16);
    run_lds("lds 2 x b64 adjacent linear", k_lds<1, false>, 16);
    run_lds("lds b128 random", k_lds<2, true>, 16);
    run_lds("lds b128 linear", k_lds<2, false>, 16);
    run_lds("lds b32 random", k_lds<3, true>, 4);
    run_lds("lds b32 linear", k_lds<3, false>, 4);
    return 0;
}
