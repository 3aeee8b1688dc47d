// Development microbenchmark (GPU box): issue rates of plain and transcendental fp32 VALU on gfx950, the numbers DESIGN.md's
// "VALU floor" is priced with.  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 4096
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
    return 0;
}
