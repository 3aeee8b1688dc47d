// Development microbenchmark (GPU box): issue rates of plain and transcendental fp32 VALU on gfx950, the numbers DESIGN.md's
// "VALU floor" is priced with.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
typedef float f2_t __attribute__((ext_vector_type(2)));
// packed fp32: one instruction = two fma per lane.  Is it issued at the rate of a plain v_fma_f32 (2x the flops) or at half of it?
template <int PK>
__global__ __launch_bounds__(256) void kpk(float *out, float seed) {
    f2_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (f2_t){seed + threadIdx.x * 1e-6f + i, seed + i * 0.5f};
    const f2_t m = (f2_t){1.0000001f, 0.9999999f}, c = (f2_t){1e-7f, 2e-7f};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (PK == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(m), "v"(c));
            if (PK == 2) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(m));
            if (PK == 3) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// other single-issue ops the kernels lean on
template <int OP>
__global__ __launch_bounds__(256) void kop(float *out, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 1) asm volatile("v_med3_f32 %0, %1, %2, 0" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 2) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 4) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(seed), "v"(seed));
            if (OP == 5) asm volatile("v_mul_i32_i24 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 6) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 7) asm volatile("v_sin_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            // the UNPACKED forms, as inline asm: left to the compiler, eight independent fma chains become four v_pk_fma_f32 (that is what
            // run<0> "v_fma_f32" above measures: 2 fma per ~4.8 cycles, not one per 2.4)
            if (OP == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(seed), "v"(seed));
            if (OP == 9) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 10) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 11) asm volatile("v_exp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 12) asm volatile("v_log_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 13) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 14) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
            if (OP == 15) asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(seed) : "vcc");
            if (OP == 16) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(seed));
            if (OP == 17) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(a[i]) : "v"(a[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K>
void run_kernel(const char *name, K kern, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double per_simd = (double)ITERS * 8 * waves_per_simd;
    printf("%-28s %d waves/SIMD: %.3f ms  -> %.3f instr/ns/SIMD  (%.2f cycles per instr at 2.4 GHz)\n", name, waves_per_simd, ms, per_simd / (ms * 1e6),
           (ms * 1e6) * 2.4 / per_simd);
    hipFree(out);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-6f + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], 1.0000001f, 1e-7f);
            if (MODE == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 1e-3f);
            if (MODE == 2) a[i] = __builtin_amdgcn_logf(a[i] + 2.0f);
            if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.0f);
            if (MODE == 4) a[i] = __builtin_amdgcn_sqrtf(a[i] + 2.0f);
            if (MODE == 5) {  // 1 transcendental : 12 plain, the K3 mix
                float t = __builtin_amdgcn_exp2f(a[i] * 1e-3f);
#pragma unroll
                for (int j = 0; j < 12; j++) t = __builtin_fmaf(t, 1.0000001f, 1e-7f);
                a[i] = t;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int per_iter_valu, int per_iter_trans, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs x waves / (4 waves per block)
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)ITERS * 8 * (per_iter_valu + per_iter_trans);  // per wave
    double per_simd = wave_instr * waves_per_simd;                              // waves per SIMD
    printf("%-28s %d waves/SIMD: %.3f ms  -> %.3f VALU instr/ns/SIMD  (%.2f cycles per instr at 2.4 GHz)\n", name, waves_per_simd, ms, per_simd / (ms * 1e6),
           (ms * 1e6) * 2.4 / per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 8}) {
        run<0>("v_fma_f32", 1, 0, w);
        run<1>("v_mul + v_exp_f32", 1, 1, w);
        run<2>("v_add + v_log_f32", 1, 1, w);
        run<3>("v_add + v_rcp_f32", 1, 1, w);
        run<4>("v_add + v_sqrt_f32", 1, 1, w);
        run<5>("v_mul + v_exp + 12 fma", 13, 1, w);
    }
    for (int w : {2, 8}) {
        run_kernel("v_pk_fma_f32", kpk<1>, w);
        run_kernel("v_pk_mul_f32", kpk<2>, w);
        run_kernel("v_pk_add_f32", kpk<3>, w);
        run_kernel("v_med3_f32", kop<1>, w);
        run_kernel("v_cvt_f32_f16", kop<2>, w);
                run_kernel("v_min3_f32", kop<4>, w);
        run_kernel("v_mul_i32_i24", kop<5>, w);
        run_kernel("v_cvt_i32_f32", kop<6>, w);
        run_kernel("v_sin_f32", kop<7>, w);
        run_kernel("v_fma_f32 (unpacked, asm)", kop<8>, w);
        run_kernel("v_add_f32 (asm)", kop<9>, w);
        run_kernel("v_mul_f32 (asm)", kop<10>, w);
        run_kernel("v_exp_f32 (asm)", kop<11>, w);
        run_kernel("v_log_f32 (asm)", kop<12>, w);
        run_kernel("v_rcp_f32 (asm)", kop<13>, w);
        run_kernel("v_sqrt_f32 (asm)", kop<14>, w);
        run_kernel("v_cmp + v_cndmask (2 instr)", kop<15>, w);
        run_kernel("v_max_f32 (asm)", kop<16>, w);
        run_kernel("v_lshlrev_b32 (asm)", kop<17>, w);
    }
    return 0;
}
