#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC]; f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (SHAPE == 32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int SHAPE>
void run(int blocks_per_cu, float* out) {
    int iters = 4096 / NACC;
    dim3 g(256 * blocks_per_cu), b(256);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipLaunchKernelGGL((k<NACC, SHAPE>), g, b, 0, 0, out, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NACC, SHAPE>), g, b, 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 5;
    double flop = (double)g.x * 4 /*waves*/ * iters * NACC * (SHAPE == 32 ? 4096.0 : 2048.0);
    printf("shape %dx%d nacc=%d blocks/CU=%d: %.3f ms  %.1f TFLOP/s\n", SHAPE, SHAPE, NACC, blocks_per_cu, ms, flop / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1, 32>(1, out); run<2, 32>(1, out); run<4, 32>(1, out);
    run<1, 32>(2, out); run<2, 32>(2, out); run<2, 32>(3, out); run<4, 32>(2, out);
    run<1, 16>(1, out); run<2, 16>(1, out); run<4, 16>(1, out); run<4, 16>(2, out); run<8, 16>(1, out);
    return 0;
}
