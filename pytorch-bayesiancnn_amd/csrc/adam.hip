// Multi-tensor Adam step for the (mu, rho) parameter pairs of the Bayesian layers (training extension):
// one launch updates up to 16 tensors.  Same update, in the same operation order, as torch.optim.Adam with
// amsgrad=False, weight_decay=0, maximize=False -- what main_bayesian.py:103 constructs:
//     m <- m + (g - m) * (1 - beta1)                     (lerp)
//     v <- v * beta2 + (1 - beta2) * g * g               (mul, addcmul)
//     p <- p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc1 = 1 - beta1^t, bc2 = 1 - beta2^t
// HBM traffic per element: 16 B read + 12 B written; purely bandwidth-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = kThreads * 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    bbb_adam_segment_t seg[BBB_MAX_SEGMENTS];
    int32_t chunk_begin[BBB_MAX_SEGMENTS + 1];
    int32_t nseg;
    float step_size, beta1, beta2, omb1, omb2, eps, sqrt_bc2;
    double lr_d, beta1_d, beta2_d;
    const float* step_dev;
    const float* lr_dev;       // optional DEVICE learning rate (captured graphs: schedulers change lr between replays)
};

__global__ __launch_bounds__(kThreads) void adam_step_kernel(const AdamArgs a) {
    const int chunk = blockIdx.x;
    float step_size = a.step_size, sqrt_bc2 = a.sqrt_bc2;
    if (a.step_dev != nullptr) {       // captured graph: the step count lives on the device; one thread derives the scalars
        __shared__ float sc[2];
        if (threadIdx.x == 0) {
            const double t = (double)*a.step_dev;
            const double lr = a.lr_dev != nullptr ? (double)*a.lr_dev : a.lr_d;
            sc[0] = (float)(lr / (1.0 - pow(a.beta1_d, t)));
            sc[1] = (float)sqrt(1.0 - pow(a.beta2_d, t));
        }
        __syncthreads();
        step_size = sc[0];
        sqrt_bc2 = sc[1];
    }
    int s = 0;
    while (s + 1 < a.nseg && chunk >= a.chunk_begin[s + 1]) ++s;
    const bbb_adam_segment_t sg = a.seg[s];
    const int64_t i0 = (int64_t)(chunk - a.chunk_begin[s]) * kChunk + (int64_t)threadIdx.x * 4;
    if (i0 >= sg.n) return;
    const int cnt = (sg.n - i0) >= 4 ? 4 : (int)(sg.n - i0);
    const bool vec = cnt == 4 && ((((uintptr_t)sg.param | (uintptr_t)sg.grad | (uintptr_t)sg.exp_avg | (uintptr_t)sg.exp_avg_sq) & 15u) == 0);
    float p[4], g[4], m[4], v[4];
    if (vec) {
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(sg.param + i0), g4 = *reinterpret_cast<const f32x4*>(sg.grad + i0);
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(sg.exp_avg + i0), v4 = *reinterpret_cast<const f32x4*>(sg.exp_avg_sq + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { p[j] = p4[j]; g[j] = g4[j]; m[j] = m4[j]; v[j] = v4[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = j < cnt;
            p[j] = ok ? sg.param[i0 + j] : 0.0f; g[j] = ok ? sg.grad[i0 + j] : 0.0f;
            m[j] = ok ? sg.exp_avg[i0 + j] : 0.0f; v[j] = ok ? sg.exp_avg_sq[i0 + j] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = m[j] + (g[j] - m[j]) * a.omb1;
        v[j] = v[j] * a.beta2 + a.omb2 * g[j] * g[j];
        const float denom = __fdiv_rn(__fsqrt_rn(v[j]), sqrt_bc2) + a.eps;
        p[j] = p[j] - step_size * (m[j] / denom);
    }
    if (vec) {
        f32x4 p4, m4, v4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { p4[j] = p[j]; m4[j] = m[j]; v4[j] = v[j]; }
        *reinterpret_cast<f32x4*>(sg.param + i0) = p4;
        *reinterpret_cast<f32x4*>(sg.exp_avg + i0) = m4;
        *reinterpret_cast<f32x4*>(sg.exp_avg_sq + i0) = v4;
    } else {
        for (int j = 0; j < cnt; ++j) { sg.param[i0 + j] = p[j]; sg.exp_avg[i0 + j] = m[j]; sg.exp_avg_sq[i0 + j] = v[j]; }
    }
}

}  // namespace

extern "C" int bbb_adam_step(const bbb_adam_segment_t* segs, int nseg, double lr, double beta1, double beta2, double eps,
                             int64_t step, const float* step_dev, const float* lr_dev, void* stream) {
    if (segs == nullptr || nseg <= 0 || nseg > BBB_MAX_SEGMENTS || (step <= 0 && step_dev == nullptr)) return BBB_EINVAL;
    if (lr_dev != nullptr && step_dev == nullptr) return BBB_EINVAL;      // device-side lr goes with the device-side step count
    if ((((uintptr_t)step_dev | (uintptr_t)lr_dev) & 3u) != 0) return BBB_EALIGN;
    if (step <= 0) step = 1;
    if (!(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return BBB_EINVAL;
    AdamArgs a = {};
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        const bbb_adam_segment_t& g = segs[s];
        if (g.param == nullptr || g.grad == nullptr || g.exp_avg == nullptr || g.exp_avg_sq == nullptr || g.n <= 0) return BBB_EINVAL;
        if ((((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 3u) != 0) return BBB_EALIGN;
        a.seg[s] = g;
        a.chunk_begin[s] = chunks;
        const int64_t c = (g.n + kChunk - 1) / kChunk;
        if (c + chunks > 0x7fffffffLL) return BBB_ESHAPE;
        chunks += (int)c;
    }
    for (int s = nseg; s <= BBB_MAX_SEGMENTS; ++s) a.chunk_begin[s] = chunks;
    a.nseg = nseg;
    // scalars in double on the host and rounded to fp32 once, as torch does with its Python-side hyper-parameters
    // (1 - 0.999 must become float(0.001), not 1.0f - 0.999f)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    a.step_size = (float)(lr / bc1);
    a.sqrt_bc2 = (float)sqrt(bc2);
    a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.lr_d = lr; a.beta1_d = beta1; a.beta2_d = beta2; a.step_dev = step_dev; a.lr_dev = lr_dev;
    hipLaunchKernelGGL(adam_step_kernel, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
