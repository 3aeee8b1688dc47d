// Development microbenchmark #3 (GPU box): what rocprofv3's FETCH_SIZE counts for the access patterns of this path, on KNOWN request counts.
//   hipcc --offload-arch=gfx950 -O3 gather_fetch.hip -o bin/gather_fetch
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o p --output-format csv -- bin/gather_fetch
// The guide (MI355X_MICROARCH.md, HBM) calibrates FETCH_SIZE only for wide coalesced reads (it reports half their bytes); K1's extra traffic is
// 4-byte gathers.  Kernels, each over a plane far larger than the eight 4 MiB L2s (so that a request is a miss):
//   k_stream16   every lane reads 16 consecutive bytes, a wave 1 KiB: N bytes read exactly once                      (the guide's case)
//   k_stream4    every lane reads 4 consecutive bytes, a wave 256 B: N bytes read exactly once
//   k_gather4    every lane reads ONE 4-byte word at a hashed address: G independent words, no two of a wave in one 128-byte line (w.h.p.)
//   k_gather4_x8 the same words, eight hashed passes over the SAME 2 MiB window: hits in L2 after the first touch (what locality looks like)
// Printed per kernel: the bytes the program asked for and the count of distinct 32 / 64 / 128-byte blocks it touched (what a memory system with
// that granularity must move at least); FETCH_SIZE of the same dispatch comes from the profiler's CSV.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void k_stream16(const uint4 *p, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream4(const uint32_t *p, size_t n4, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// word index of gather g: a hash of g folded into [0, words)
__global__ __launch_bounds__(256) void k_gather4(const uint32_t *p, uint32_t words_mask, uint32_t gathers, uint32_t salt, uint32_t *out) {
    uint32_t acc = 0;
    for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < gathers; g += gridDim.x * 256) acc ^= p[hash32(g ^ salt) & words_mask];
    if (acc == 0x12345678u) out[0] = acc;
}
// window: every workgroup's gathers fall into ONE 2 MiB window (2^19 words) chosen by the block index, touched `passes` times with new hashes
__global__ __launch_bounds__(256) void k_gather4_window(const uint32_t *p, uint32_t windows_mask, uint32_t per_pass, uint32_t passes, uint32_t *out) {
    uint32_t acc = 0;
    const uint32_t base = (hash32(blockIdx.x) & windows_mask) << 19;
    for (uint32_t q = 0; q < passes; q++)
        for (uint32_t g = threadIdx.x; g < per_pass; g += 256) acc ^= p[base + (hash32(g * 977u + q * 0x9e3779b9u + blockIdx.x) & 0x7ffffu)];
    if (acc == 0x12345678u) out[0] = acc;
}

static void distinct_blocks(const std::vector<uint32_t> &word_idx, const char *name, size_t asked_bytes) {
    size_t cnt[3] = {0, 0, 0};
    const int sh[3] = {3, 4, 5};  // words -> 32 / 64 / 128-byte blocks
    for (int k = 0; k < 3; k++) {
        std::vector<uint32_t> b(word_idx.size());
        for (size_t i = 0; i < b.size(); i++) b[i] = word_idx[i] >> sh[k];
        std::vector<uint8_t> seen(((size_t)1 << 32 >> sh[k]) / 8 + 1, 0);
        for (uint32_t v : b) { if (!(seen[v >> 3] & (1u << (v & 7)))) { seen[v >> 3] |= 1u << (v & 7); cnt[k]++; } }
    }
    printf("%-18s asked %12zu B   distinct blocks x size: 32 B %12zu B, 64 B %12zu B, 128 B %12zu B\n", name, asked_bytes, cnt[0] * 32, cnt[1] * 64, cnt[2] * 128);
}
static uint32_t h32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB plane: 2^28 words
    uint32_t *p, *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8;
    const uint32_t words_mask = (uint32_t)(bytes / 4 - 1), gathers = 1u << 24;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(256), 0, 0, (const uint4 *)p, bytes / 16, out);
        hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(256), 0, 0, p, bytes / 4, out);
        hipLaunchKernelGGL(k_gather4, dim3(grid), dim3(256), 0, 0, p, words_mask, gathers, 17u * rep, out);
        hipLaunchKernelGGL(k_gather4_window, dim3(grid), dim3(256), 0, 0, p, (uint32_t)(bytes >> 21) - 1, 1u << 13, 8u, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    printf("%-18s asked %12zu B   (every byte exactly once)\n", "k_stream16", bytes);
    printf("%-18s asked %12zu B   (every byte exactly once)\n", "k_stream4", bytes);
    {
        std::vector<uint32_t> idx(gathers);
        for (uint32_t g = 0; g < gathers; g++) idx[g] = h32(g ^ 0u) & words_mask;
        distinct_blocks(idx, "k_gather4", (size_t)gathers * 4);
    }
    {
        std::vector<uint32_t> idx;
        idx.reserve((size_t)grid * 8 * (1u << 13));
        for (uint32_t b = 0; b < (uint32_t)grid; b++) {
            const uint32_t base = (h32(b) & ((uint32_t)(bytes >> 21) - 1)) << 19;
            for (uint32_t q = 0; q < 8; q++)
                for (uint32_t g = 0; g < (1u << 13); g++) idx.push_back(base + (h32(g * 977u + q * 0x9e3779b9u + b) & 0x7ffffu));
        }
        distinct_blocks(idx, "k_gather4_window", idx.size() * 4);
    }
    return 0;
}
