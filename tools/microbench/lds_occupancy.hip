// Development microbenchmark #4 (GPU box): how many 512-thread workgroups of a given register count and dynamic LDS size one CU holds — the
// runtime's own occupancy figure, i.e. the LDS allocation granularity and the per-CU LDS budget as this driver applies them.
//   hipcc --offload-arch=gfx950 -O3 lds_occupancy.hip -o bin/lds_occupancy && bin/lds_occupancy
// Why: K3's tiled launches ask for pitch * rows * bytes-per-texel of dynamic LDS (4K: pass 0 74 x 14 x 52 = 53 872 B, later passes 76 x 16 x 36 =
// 43 776 B; K2 45 696 B static).  Three workgroups of pass 0 need 161 616 B of the CU's 160 KiB = 163 840 B: whether they fit is a question of the
// allocation granule, which no document in this container states.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int REGS>
__global__ __launch_bounds__(512) void k_dummy(float *out, int n) {
    extern __shared__ float lds[];
    float acc[REGS];
#pragma unroll
    for (int i = 0; i < REGS; i++) acc[i] = out[(threadIdx.x + i * 977) % n];
    lds[threadIdx.x] = acc[0];
    __syncthreads();
    float s = lds[(threadIdx.x * 7) & 511];
#pragma unroll
    for (int i = 0; i < REGS; i++) s = s * acc[i] + acc[(i + 1) % REGS];
    out[threadIdx.x % n] = s;
}

template <int REGS>
static void probe(const char *name) {
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void *)k_dummy<REGS>);
    hipFuncSetAttribute((const void *)k_dummy<REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%s: %d VGPRs, static LDS %zu B\n", name, fa.numRegs, (size_t)fa.sharedSizeBytes);
    const size_t sizes[] = {32768, 36864, 38912, 40960, 41040, 43776, 44032, 45696, 46176, 50024, 51200, 53248, 53760, 53872, 54016, 54272, 54528, 54613, 55040, 55296, 57344, 65536, 66560, 81920};
    for (size_t s : sizes) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_dummy<REGS>, 512, s);
        printf("  dynamic LDS %6zu B -> %d workgroups of 512 threads per CU (%s)\n", s, nb, hipGetErrorString(e));
    }
}

// ... and what the hardware does with it: `grid` workgroups of 512 threads and `lds` bytes each count themselves in and wait (bounded: 2 ms of the
// wall clock) until all have arrived — they all arrive only if the whole grid is resident at once.
__global__ __launch_bounds__(512) void k_resident(unsigned int *arrived, unsigned int *seen_all, unsigned int grid, long long budget_ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(arrived, 1u);
        const long long t0 = wall_clock64();
        bool all = false;
        while (wall_clock64() - t0 < budget_ticks) {
            if (atomicAdd(arrived, 0u) >= grid) { all = true; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        if (all) atomicAdd(seen_all, 1u);
    }
    __syncthreads();
    if (lds[(threadIdx.x * 5) & 511] == 2.0f) arrived[1] = 1u;
}
static void resident(int ncu, int per_cu, size_t lds) {
    unsigned int *d;
    hipMalloc(&d, 64);
    hipMemset(d, 0, 64);
    hipFuncSetAttribute((const void *)k_resident, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned int grid = (unsigned int)(ncu * per_cu);
    hipLaunchKernelGGL(k_resident, dim3(grid), dim3(512), lds, 0, d, d + 8, grid, (long long)200000);  // wall_clock64: 100 MHz -> 2 ms
    hipDeviceSynchronize();
    unsigned int h[16];
    hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    printf("  %d workgroups per CU x %6zu B: %u of %u workgroups saw the whole grid resident -> %s\n", per_cu, lds, h[8], grid, h[8] == grid ? "RESIDENT" : "not resident");
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    probe<24>("k_dummy<24> (about K3 pass 0's register count)");
    probe<40>("k_dummy<40> (about the later passes')");
    printf("resident at once (512-thread workgroups, measured):\n");
    const size_t sizes[] = {43776, 53248, 53760, 53872, 54272, 54528};
    for (size_t s : sizes) resident(p.multiProcessorCount, 3, s);
    const size_t sizes4[] = {38304, 38912, 40960, 41040};
    for (size_t s : sizes4) resident(p.multiProcessorCount, 4, s);
    return 0;
}
