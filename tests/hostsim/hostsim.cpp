// tests/hostsim/hostsim.cpp — TEST INFRASTRUCTURE (see hip/hip_runtime.h): the simulator's per-thread execution state.
#include <hip/hip_runtime.h>

// bench.py asks the loaded library whether it is this simulator (no device to select, drain or profile then)
extern "C" int rfx_hostsim_build(void) { return 1; }

thread_local hostsim_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local int hostsim_phase = 0, hostsim_sync_count = 0;
thread_local std::jmp_buf hostsim_barrier;
thread_local unsigned char *hostsim_lds = nullptr;
thread_local float *hostsim_shfl = nullptr;
thread_local unsigned int hostsim_nthreads = 0;
