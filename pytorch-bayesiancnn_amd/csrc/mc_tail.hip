// Monte-Carlo ensemble tail on gfx950: per-draw log_softmax over classes fused with the log-sum-exp over
// draws.  Replaces F.log_softmax + the strided store into outputs[:, :, j] (main_bayesian.py:49,78) and
// utils.logmeanexp over the ensemble dim (utils.py:14-22, main_bayesian.py:53,80).
//
// One wave per image: lanes stride over classes, two wave reductions per draw (max, sum-exp), an online
// (max, sum) pair per class carried across draws in registers.  logits [E][B][C] -> lse [B][C].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"

namespace {

constexpr int kMaxPerLane = 16;   // classes <= 64 * 16

__global__ __launch_bounds__(64) void mc_tail_kernel(const float* __restrict__ logits, int E, int B, int C, float sub,
                                                     float* __restrict__ out) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    float run_m[kMaxPerLane], run_s[kMaxPerLane];
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) { run_m[i] = -INFINITY; run_s[i] = 0.0f; }
    for (int e = 0; e < E; ++e) {
        const float* row = logits + ((int64_t)e * B + b) * C;
        float v[kMaxPerLane];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int c = lane + i * 64;
            v[i] = c < C ? row[c] : -INFINITY;
            mx = fmaxf(mx, v[i]);
        }
        mx = bbb::wave_max(mx);
        float se = 0.0f;
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int c = lane + i * 64;
            if (c < C) se += expf(v[i] - mx);
        }
        se = bbb::wave_sum_all(se);
        const float lz = mx + logf(se);            // log partition of this draw
#pragma unroll
        for (int i = 0; i < kMaxPerLane; ++i) {
            const int c = lane + i * 64;
            if (c < C) {
                const float ls = v[i] - lz;         // log_softmax
                const float nm = fmaxf(run_m[i], ls);
                run_s[i] = run_s[i] * expf(run_m[i] - nm) + expf(ls - nm);
                run_m[i] = nm;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < C) out[(int64_t)b * C + c] = run_m[i] + logf(run_s[i]) - sub;
    }
}

}  // namespace

extern "C" int bbb_mc_tail(const float* logits, int draws, int batch, int classes, int mean_over, float* lse_out,
                           void* stream) {
    if (logits == nullptr || lse_out == nullptr || draws <= 0 || batch <= 0 || classes <= 0 || classes > 64 * kMaxPerLane ||
        mean_over < 0)
        return BBB_EINVAL;
    if ((((uintptr_t)logits | (uintptr_t)lse_out) & 3u) != 0) return BBB_EALIGN;
    const float sub = mean_over > 0 ? logf((float)mean_over) : 0.0f;
    hipLaunchKernelGGL(mc_tail_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, logits, draws, batch, classes, sub, lse_out);
    return (int)hipGetLastError();
}

extern "C" int bbb_abi_version(void) { return BBB_ABI_VERSION; }

extern "C" const char* bbb_build_info(void) {
    return "libbbb_hip gfx950 (CDNA4) fp32-MFMA; built " __DATE__ " " __TIME__;
}
