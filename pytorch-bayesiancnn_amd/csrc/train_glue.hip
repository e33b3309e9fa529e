// Small reductions / element-wise steps of the training backward (training extension) that used to be ATen
// kernels in bbb_hip/fast_train.py: bias gradients (main_bayesian.py:57: loss.backward() through F.conv2d's bias), the sum over
// draws of a first LRT layer's moment gradients, x^2 for the LRT variance contraction's weight gradient, and the LRT input
// gradient g_x = dgrad(g_mu, mu) + 2 x dgrad(g_var, sigma^2) (layers/BBB_LRT/BBBConv.py:71-81 differentiated).
// All bandwidth-bound, deterministic (fixed reduction trees, no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[r] = sum over o < outer, j < cols of x[o * outer_stride + r * row_pitch + j].  One 256-thread block per row: every thread
// walks 16-byte vectors at a stride of 256 (fp32 accumulation in index order), then a fixed LDS tree.
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t outer, int64_t cols,
                                                        int64_t row_pitch, int64_t outer_stride) {
    __shared__ float sm[256];
    const int64_t r = blockIdx.x;
    const int tid = threadIdx.x;
    const bool vec = ((cols | row_pitch | outer_stride) & 3) == 0 && (((uintptr_t)x) & 15u) == 0;
    float acc = 0.0f;
    for (int64_t o = 0; o < outer; ++o) {
        const float* p = x + o * outer_stride + r * row_pitch;
        if (vec) {
            const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
            for (int64_t j = tid; j < cols / 4; j += 256) {
                const f32x4 v = p4[j];
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else {
            for (int64_t j = tid; j < cols; j += 256) acc += p[j];
        }
    }
    sm[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[r] = sm[0];
}

// out[i] = sum over o < outer of x[o * n + i], added in index order (o = 0, 1, ...)
__global__ __launch_bounds__(256) void sum_leading_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t outer, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(x) + i;
    f32x4 acc = p[0];
    for (int64_t o = 1; o < outer; ++o) {
        const f32x4 v = p[o * n4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += v[c];
    }
    reinterpret_cast<f32x4*>(out)[i] = acc;
}

__global__ __launch_bounds__(256) void sum_leading_scalar_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t outer, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = x[i];
    for (int64_t o = 1; o < outer; ++o) acc += x[o * n + i];
    out[i] = acc;
}

// mode 0: out = x * x;  mode 1: out[i] = a[i] + 2 * x[i % xn] * b[i]   (x may be one slab shared by all draws)
__global__ __launch_bounds__(256) void lrt_glue_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ b,
                                                       float* __restrict__ out, int64_t n4, int64_t xn4, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[mode == 0 ? i : i % xn4];
    f32x4 o;
    if (mode == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = xv[c] * xv[c];
    } else {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = fmaf(2.0f * xv[c], bv[c], av[c]);
    }
    reinterpret_cast<f32x4*>(out)[i] = o;
}

// the same arithmetic one element per thread: any element count, 4-byte alignment (views with a storage offset, 3-channel inputs)
__global__ __launch_bounds__(256) void lrt_glue_scalar_kernel(const float* __restrict__ a, const float* __restrict__ x,
                                                              const float* __restrict__ b, float* __restrict__ out, int64_t n,
                                                              int64_t xn, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[mode == 0 ? i : i % xn];
    out[i] = mode == 0 ? xv * xv : fmaf(2.0f * xv, b[i], a[i]);
}

}  // namespace

extern "C" int bbb_plane_sum(const float* x, float* out, int64_t outer, int64_t rows, int64_t cols, int64_t row_pitch, int64_t outer_stride,
                             void* stream) {
    if (x == nullptr || out == nullptr || outer <= 0 || rows <= 0 || cols <= 0 || row_pitch < cols || outer_stride < 0) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    if (rows > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, out, outer, cols, row_pitch, outer_stride);
    return (int)hipGetLastError();
}

extern "C" int bbb_sum_leading(const float* x, float* out, int64_t outer, int64_t n, void* stream) {
    if (x == nullptr || out == nullptr || outer <= 0 || n <= 0) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const bool vec = n % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15u) == 0;      // same additions, same order, either way
    const int64_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    if (vec) hipLaunchKernelGGL(sum_leading_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, outer, n / 4);
    else     hipLaunchKernelGGL(sum_leading_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, outer, n);
    return (int)hipGetLastError();
}

extern "C" int bbb_lrt_glue(const float* a, const float* x, const float* b, float* out, int64_t n, int64_t x_n, int mode, void* stream) {
    if (x == nullptr || out == nullptr || n <= 0 || (mode != 0 && mode != 1)) return BBB_EINVAL;
    if (mode == 1 && (a == nullptr || b == nullptr || x_n <= 0 || n % x_n != 0)) return BBB_EINVAL;
    if ((((uintptr_t)a | (uintptr_t)x | (uintptr_t)b | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    // 16-byte vectors when every operand allows it; otherwise (ragged element counts, views with a storage offset) the scalar
    // kernel: the same products and the same fmaf per element, so the result does not depend on which one ran
    const bool vec = n % 4 == 0 && (mode == 0 || x_n % 4 == 0) && (((uintptr_t)a | (uintptr_t)x | (uintptr_t)b | (uintptr_t)out) & 15u) == 0;
    const int64_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    if (vec) hipLaunchKernelGGL(lrt_glue_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, x, b, out, n / 4, mode == 1 ? x_n / 4 : 1, mode);
    else     hipLaunchKernelGGL(lrt_glue_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, x, b, out, n, mode == 1 ? x_n : 1, mode);
    return (int)hipGetLastError();
}
