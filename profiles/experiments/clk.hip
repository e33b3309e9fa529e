// shader-clock probe: one wave spins for `iters` sleeps and reports (s_memtime ticks, 100 MHz wall ticks)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void clk_kernel(uint64_t* out, int64_t iters) {
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t w0 = wall_clock64();
    for (int64_t i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(8);
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
extern "C" int clk_probe(uint64_t* out, int64_t iters, void* stream) {
    hipLaunchKernelGGL(clk_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, iters);
    return (int)hipGetLastError();
}
