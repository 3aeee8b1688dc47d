// Development microbenchmark #3 (GPU box, round 6): issue cost of the forms K2's neighbourhood AABB and the pair experiments choose between —
// packed-half min / max against the fp32 three-operand forms, the conversions around them, SGPR-pair operands of packed fp32.
//   hipcc --offload-arch=gfx950 -O3 valu_rates3.hip -o bin/valu_rates3 && bin/valu_rates3
// Same frame as valu_rates2.hip, but the 8-chain body is repeated four times per loop iteration (valu_rates2's one repeat carries ~8 cycles of loop overhead per
// 8 instructions: its one-instruction rows read ~0.9 cycles high): 8 independent chains per lane, 4096 bodies, 8 (then 6) waves per SIMD on every CU; cycles per wave64
// instruction per SIMD at a nominal 2.4 GHz — compare rows, not absolutes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096

#define DEF_KERNEL(NAME, ASM_STR, ...)                                                     \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                 \
        float a[8];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-6f + i; \
        float b = seed * 0.5f, c = seed * 0.25f;                                           \
        const float sg = __builtin_amdgcn_readfirstlane(seed * 1.0000001f);               \
        for (int it = 0; it < ITERS / 4; it++) {                                           \
            _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                          \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM_STR : "+v"(a[i]) : "v"(b), "v"(c), "s"(sg)__VA_ARGS__); } \
            }                                                                              \
        }                                                                                  \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i];                           \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
    }
DEF_KERNEL(k_add, "v_add_f32 %0, %0, %1")
DEF_KERNEL(k_min_f32, "v_min_f32 %0, %0, %1")
DEF_KERNEL(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_pk_min_f16, "v_pk_min_f16 %0, %0, %1")
DEF_KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
DEF_KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
DEF_KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF_KERNEL(k_min_f16, "v_min_f16 %0, %0, %1")
DEF_KERNEL(k_min3_f16, "v_min3_f16 %0, %0, %1, %2")
DEF_KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %0")
DEF_KERNEL(k_fma_mix_h, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]")
DEF_KERNEL(k_log, "v_log_f32 %0, %0")
DEF_KERNEL(k_rcp, "v_rcp_f32 %0, %0")
DEF_KERNEL(k_mul_sgpr, "v_mul_f32 %0, %3, %0")
DEF_KERNEL(k_fma_sgpr, "v_fma_f32 %0, %3, %0, %1")
DEF_KERNEL(k_mix_min3_add, "v_min3_f32 %0, %0, %1, %2\n\tv_add_f32 %0, %0, %1")
DEF_KERNEL(k_mix_pkmin_add, "v_pk_min_f16 %0, %0, %1\n\tv_add_f32 %0, %0, %1")
DEF_KERNEL(k_mix_mix_add, "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]\n\tv_add_f32 %0, %0, %1")
DEF_KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
DEF_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
DEF_KERNEL(k_mul, "v_mul_f32 %0, %1, %0")
DEF_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
DEF_KERNEL(k_fmac_sgpr, "v_fmac_f32 %0, %3, %2")
DEF_KERNEL(k_add_sgpr, "v_add_f32 %0, %3, %0")
DEF_KERNEL(k_mul_inl, "v_mul_f32 %0, 0.5, %0")
DEF_KERNEL(k_fma_inl, "v_fma_f32 %0, %0, 0.5, 0.5")
DEF_KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
DEF_KERNEL(k_lshr_sgpr, "v_lshrrev_b32 %0, %3, %0")
DEF_KERNEL(k_lshr_inl, "v_lshrrev_b32 %0, 3, %0")
DEF_KERNEL(k_lshr_vgpr, "v_lshrrev_b32 %0, %1, %0")
DEF_KERNEL(k_and_sgpr, "v_and_b32 %0, %3, %0")
DEF_KERNEL(k_addu_sgpr, "v_add_u32 %0, %3, %0")
DEF_KERNEL(k_addu, "v_add_u32 %0, %1, %0")
DEF_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xe4")
DEF_KERNEL(k_bitop3_sgpr, "v_bitop3_b32 %0, %0, %1, %3 bitop3:0xe4")
DEF_KERNEL(k_max_abs, "v_max_f32_e64 %0, |%0|, |%1|")
DEF_KERNEL(k_cmp_sgpr_add, "v_cmp_gt_f32 vcc, %3, %0\n\tv_add_f32 %0, %0, %2", : "vcc")
DEF_KERNEL(k_cmp_add, "v_cmp_gt_f32 vcc, %1, %0\n\tv_add_f32 %0, %0, %2", : "vcc")
DEF_KERNEL(k_mix_add_cvt, "v_add_f32 %0, %0, %1\n\tv_cvt_i32_f32 %0, %0")
DEF_KERNEL(k_mix_add_mul, "v_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2")
DEF_KERNEL(k_mix_2add_min3, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_min3_f32 %0, %0, %1, %2")
DEF_KERNEL(k_mix_2add_mix, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,1]")
DEF_KERNEL(k_mix_3add_log, "v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %1\n\tv_log_f32 %0, %0")
DEF_KERNEL(k_cmp_lt_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc", : "vcc")

typedef float f2_t __attribute__((ext_vector_type(2)));
#define DEF_PK(NAME, ASM_STR, ...)                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, float seed) {                 \
        f2_t a[8];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 8; i++) a[i] = (f2_t){seed + threadIdx.x * 1e-6f + i, seed + i * 0.5f}; \
        const f2_t m = (f2_t){1.0000001f, 0.9999999f}, c = (f2_t){1e-7f, 2e-7f};           \
        for (int it = 0; it < ITERS / 4; it++) {                                           \
            _Pragma("unroll") for (int rep = 0; rep < 4; rep++) {                          \
                _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM_STR : "+v"(a[i]) : "v"(m), "v"(c)__VA_ARGS__); } \
            }                                                                              \
        }                                                                                  \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;                \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                           \
    }
DEF_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
DEF_PK(k_pk_mul_sgpr, "v_pk_mul_f32 %0, %0, s[12:13]", : "s12", "s13")
DEF_PK(k_pk_fma_sgpr, "v_pk_fma_f32 %0, %0, s[12:13], %2", : "s12", "s13")
DEF_PK(k_pk_fma_bcast_lo, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]")
DEF_PK(k_pk_add, "v_pk_add_f32 %0, %0, %1")

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
    printf("%-26s %d w/SIMD %.3f ms  %6.2f cycles per body of %d instr  (%.2f per instr)\n", name, waves_per_simd, ms, (ms * 1e6) * 2.4 / per_simd, instr_per_body,
           (ms * 1e6) * 2.4 / per_simd / instr_per_body);
    hipFree(out);
}
#define RUN(k, n) run_kernel(#k, k, w, n)
int main() {
    for (int w : {8, 6}) {
        RUN(k_add, 1); RUN(k_min_f32, 1); RUN(k_min3_f32, 1); RUN(k_max3_f32, 1); RUN(k_pk_min_f16, 1); RUN(k_pk_max_f16, 1); RUN(k_pk_add_f16, 1); RUN(k_pk_fma_f16, 1);
        RUN(k_min_f16, 1); RUN(k_min3_f16, 1); RUN(k_cvt_f32_f16, 1); RUN(k_cvt_f16_f32, 1); RUN(k_fma_mix_h, 1); RUN(k_log, 1); RUN(k_rcp, 1);
        RUN(k_mul, 1); RUN(k_mul_sgpr, 1); RUN(k_mul_inl, 1); RUN(k_mul_lit, 1); RUN(k_fma, 1); RUN(k_fma_sgpr, 1); RUN(k_fma_inl, 1); RUN(k_fmac, 1); RUN(k_fmac_sgpr, 1); RUN(k_add_sgpr, 1);
        RUN(k_lshr_sgpr, 1); RUN(k_lshr_inl, 1); RUN(k_lshr_vgpr, 1); RUN(k_and_sgpr, 1); RUN(k_addu, 1); RUN(k_addu_sgpr, 1); RUN(k_bitop3, 1); RUN(k_bitop3_sgpr, 1); RUN(k_max_abs, 1);
        RUN(k_cmp_add, 2); RUN(k_cmp_sgpr_add, 2); RUN(k_mix_add_cvt, 2); RUN(k_mix_add_mul, 2); RUN(k_mix_2add_min3, 3); RUN(k_mix_2add_mix, 3); RUN(k_mix_3add_log, 4); RUN(k_mix_min3_add, 2); RUN(k_mix_pkmin_add, 2); RUN(k_mix_mix_add, 2); RUN(k_dot2_f32_f16, 1); RUN(k_perm, 1); RUN(k_cmp_lt_cndmask, 2);
        RUN(k_pk_fma, 1); RUN(k_pk_mul, 1); RUN(k_pk_mul_sgpr, 1); RUN(k_pk_fma_sgpr, 1); RUN(k_pk_fma_bcast_lo, 1); RUN(k_pk_add, 1);
    }
    return 0;
}
