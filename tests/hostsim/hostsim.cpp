// tests/hostsim/hostsim.cpp — TEST INFRASTRUCTURE (see hip/hip_runtime.h): the simulator's per-thread execution state, and the one kernel
// that is substituted rather than executed.
#include <hip/hip_runtime.h>

thread_local hostsim_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local int hostsim_phase = 0;
thread_local std::jmp_buf hostsim_barrier;
thread_local unsigned char *hostsim_lds = nullptr;

// k1_prepare (k1_ssgi.hip) reduces with wave shuffles, which a thread-at-a-time simulator cannot run.  Its results are exact by
// construction (view Z per texel with IEEE arithmetic, min / max per base cell), so this loop IS its specification.
void hostsim_k1_prepare(int base_cell, const float *depth, float *viewz, float2 *base, int W, int H, int base_w, float nearMulFar, float farMinusNear,
                        float cameraFar, float nearMinusFar, float cameraNear, int perspective) {
    const int base_h = (H + base_cell - 1) / base_cell;
#pragma omp parallel for schedule(static)
    for (int cy = 0; cy < base_h; cy++)
        for (int cx = 0; cx < base_w; cx++) {
            float mn = INFINITY, mx = -INFINITY;
            for (int y = cy * base_cell; y < std::min((cy + 1) * base_cell, H); y++)
                for (int x = cx * base_cell; x < std::min((cx + 1) * base_cell, W); x++) {
                    const float dpt = depth[(size_t)y * W + x];
                    const float z = perspective ? nearMulFar / (farMinusNear * dpt - cameraFar) : dpt * nearMinusFar - cameraNear;
                    viewz[(size_t)y * W + x] = z;
                    mn = std::fmin(mn, z);
                    mx = std::fmax(mx, z);
                }
            base[(size_t)cy * base_w + cx] = make_float2(mn, mx);
        }
}
