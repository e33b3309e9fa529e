// Measurement aid for bench.py (NOT part of libbbb_hip.so, not on any product path): what a loop of nothing but
// v_mfma_f32_32x32x2_f32 reaches on the box the bench runs on.  MI355X_MICROARCH.md's dense fp32 matrix peak (157.3 TFLOP/s =
// 256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz) assumes an MFMA issued every 64 cycles at 2.4 GHz; this loop -- 4 independent
// accumulators per wave, 2 workgroups of 4 waves per CU, operands in registers, no LDS, no global memory -- is the practical
// ceiling any kernel built on that instruction can approach.  bench.py prints it next to roofline.peak.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop_kernel(float* out, int iters, float a0, float b0, unsigned long long* clk) {
    // shader clock over the loop: s_memtime counts shader-clock cycles, wall_clock64 a constant 100 MHz
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk != nullptr) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = wall_clock64() - w0;
    }
}
// The same loop on v_mfma_f32_32x32x16_f16 (the instruction a split-fp16 contraction would run on: DESIGN.md section 9, item 2).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_f16_loop_kernel(float* out, int iters, float a0, float b0, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    f16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(a0 + threadIdx.x * 1e-3f + j * 0.01f);
        b[j] = (_Float16)(b0 + threadIdx.x * 2e-3f - j * 0.01f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk != nullptr) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = wall_clock64() - w0;
    }
}
}  // namespace

// tflops_out[0] = best of `reps` timed launches (HIP events on `stream`), in TFLOP/s; tflops_out[1] = the shader clock (GHz) the
// kernel itself measured during that launch (0 if unavailable).  Returns a hipError_t.
static int probe_impl(double* tflops_out, int reps, void* stream, bool f16);
extern "C" int probe_mfma_f32_ceiling(double* tflops_out, int reps, void* stream) { return probe_impl(tflops_out, reps, stream, false); }
// the same for v_mfma_f32_32x32x16_f16 (32768 FLOP per instruction and wave)
extern "C" int probe_mfma_f16_ceiling(double* tflops_out, int reps, void* stream) { return probe_impl(tflops_out, reps, stream, true); }

static int probe_impl(double* tflops_out, int reps, void* stream, bool f16) {
    if (tflops_out == nullptr || reps <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    hipError_t er = hipGetDevice(&dev);
    if (er != hipSuccess) return (int)er;
    er = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (er != hipSuccess) return (int)er;
    const int blocks = cus * 2, iters = 4096;                    // 16384 MFMAs per wave: ~0.45 ms per launch
    float* out = nullptr;
    er = hipMalloc(&out, (size_t)blocks * 256 * sizeof(float) + 64);
    if (er != hipSuccess) return (int)er;
    unsigned long long* clk = reinterpret_cast<unsigned long long*>(out + (size_t)blocks * 256);
    double best_clk = 0.0;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(out); return (int)hipErrorUnknown; }
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {                           // first launch is a warm-up
        (void)hipEventRecord(e0, st);
        if (f16) hipLaunchKernelGGL(mfma_f16_loop_kernel, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, clk);
        else     hipLaunchKernelGGL(mfma_loop_kernel, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, clk);
        (void)hipEventRecord(e1, st);
        er = hipEventSynchronize(e1);
        if (er != hipSuccess) break;
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)blocks * 4.0 * iters * 4.0 * (f16 ? 32768.0 : 4096.0) / ((double)ms * 1e9);
        if (r > 0 && tf > best) {
            best = tf;
            unsigned long long h[2] = {0, 0};
            if (hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[1] > 0) best_clk = (double)h[0] / (double)h[1] * 0.1;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    tflops_out[0] = best;
    tflops_out[1] = best_clk;
    return (int)er;
}

// Write-only streaming probe (review r05 item 6): the bytes of the fused reparam+KL pass's output written with the same store shape
// (16 bytes per lane, consecutive lanes consecutive vectors, 256-thread blocks, one 4 KB piece per block and "draw", draws strided by
// the tensor size) and NO arithmetic or reads -- the HBM write roof of this box for that pattern.  gbps_out[0] = plain stores,
// gbps_out[1] = non-temporal stores; best of `reps` launches each, HIP events on `stream`.
namespace {
typedef float wf32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void write_stream_kernel(float* out, long long n_vec, int draws, long long draw_stride_vec, float v) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_vec) return;
    wf32x4* p = reinterpret_cast<wf32x4*>(out) + i;
    const wf32x4 val = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    for (int e = 0; e < draws; ++e) {
        if (NT) __builtin_nontemporal_store(val, p);
        else    *p = val;
        p += draw_stride_vec;
    }
}
}  // namespace

extern "C" int probe_write_roof(double* gbps_out, long long elems_per_draw, int draws, int reps, void* stream) {
    if (gbps_out == nullptr || elems_per_draw <= 0 || draws <= 0 || reps <= 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const long long n_vec = (elems_per_draw + 3) / 4;
    float* buf = nullptr;
    hipError_t er = hipMalloc(&buf, (size_t)n_vec * 16 * draws);
    if (er != hipSuccess) return (int)er;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(buf); return (int)hipErrorUnknown; }
    const dim3 grid((unsigned)((n_vec + 255) / 256)), block(256);
    for (int nt = 0; nt < 2; ++nt) {
        double best = 0.0;
        for (int r = 0; r < reps + 2; ++r) {                       // two warm-up rounds
            (void)hipEventRecord(e0, st);
            for (int k = 0; k < 10; ++k) {                         // ten launches back to back per bracket
                if (nt) hipLaunchKernelGGL(write_stream_kernel<true>, grid, block, 0, st, buf, n_vec, draws, n_vec, (float)k);
                else    hipLaunchKernelGGL(write_stream_kernel<false>, grid, block, 0, st, buf, n_vec, draws, n_vec, (float)k);
            }
            (void)hipEventRecord(e1, st);
            er = hipEventSynchronize(e1);
            if (er != hipSuccess) break;
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double gbps = 10.0 * (double)n_vec * 16.0 * draws / ((double)ms * 1e6);
            if (r >= 2 && gbps > best) best = gbps;
        }
        gbps_out[nt] = best;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return (int)er;
}
