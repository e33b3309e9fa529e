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

// Same reduction for logits stored batch-innermost, [E][C][B] (what the batched ensemble path produces): one thread
// per image, lanes = consecutive images so every read is coalesced; the per-draw log-partition values sit in LDS.
constexpr int kCbThreads = 64;
constexpr int kCbRows = 16;
// blockDim = (64 images, ny <= 16): phase 1 spreads the draws over y (log-partition of each draw), phase 2 spreads the
// classes over y (log-sum-exp over draws) -- the serial chain per thread is E*C/ny long instead of E*C.
// Work units (ensemble sharding, include/bbb_hip.h): the E slabs are units u = off + e of the draw-major (draw, batch slice)
// grid with S slices of Bs images; image b = s*Bs + bl reduces over the local units of ITS slice, e = e0, e0 + S, ...
// (e0 = (s - off) mod S), and gets -inf when the rank holds none.  S = 1, off = 0 is the plain [E][C][B] case.
// Groups (grp > 0; several Monte-Carlo steps per launch): the E = S * grp slabs are S consecutive steps of grp draws each on
// S different batches; image b of step s = b / Bs reduces over slabs s*grp .. s*grp + grp - 1 (e = e0 + k, stride 1).
// Step epilogue riding on the tail launch (a captured Monte-Carlo step otherwise spends two more element-wise launches on
// them): kl_out = kl_in * kl_scale (the KL of ONE forward times the number of forwards this rank ran, main_bayesian.py:76-77)
// and counter += counter_add (the device-side noise call counter of include/bbb_hip.h `call_dev`: every kernel that reads it
// runs before the tail of the same step).
__global__ __launch_bounds__(kCbThreads * kCbRows) void mc_tail_cb_kernel(const float* __restrict__ logits, int E, int Bs, int S,
                                                                          int off, int C, float sub, float* __restrict__ out,
                                                                          const float* __restrict__ kl_in, float kl_scale,
                                                                          float* __restrict__ kl_out, uint32_t* counter,
                                                                          uint32_t counter_add, int grp, int goff) {
    extern __shared__ float lz[];                       // [ceil(E/S)][64]
    const int tx = threadIdx.x, ty = threadIdx.y, ny = blockDim.y;
    if (blockIdx.x == 0 && tx == 0 && ty == 0) {
        if (kl_out != nullptr) *kl_out = *kl_in * kl_scale;
        if (counter != nullptr) *counter += counter_add;
    }
    const int B = Bs * S;
    const int b = blockIdx.x * kCbThreads + tx;
    const bool ok = b < B;
    const int bb = ok ? b : B - 1;
    const int sl = bb / Bs, bl = bb - sl * Bs;
    const int es = grp > 0 ? 1 : S;                     // slab stride between the draws an image reduces over
    // groups: step sl owns slabs [sl*grp - goff, (sl+1)*grp - goff), clipped to the launch (a rank's share of a group of steps may
    // start and end in the middle of a step; goff = draws of the first step that belong to an earlier rank)
    const int g_lo = sl * grp - goff > 0 ? sl * grp - goff : 0;
    const int g_hi = (sl + 1) * grp - goff < E ? (sl + 1) * grp - goff : E;
    const int e0 = grp > 0 ? g_lo : (sl - off % S + S) % S;
    const int ne = grp > 0 ? (g_hi > g_lo ? g_hi - g_lo : 0) : (e0 < E ? (E - e0 + S - 1) / S : 0);
    for (int k = ty; k < ne; k += ny) {
        const float* p = logits + (int64_t)(e0 + k * es) * C * Bs + bl;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, p[(int64_t)c * Bs]);
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(p[(int64_t)c * Bs] - mx);
        lz[k * kCbThreads + tx] = mx + logf(se);
    }
    __syncthreads();
    for (int c = ty; c < C; c += ny) {
        float m = -INFINITY, s = 0.0f;
        for (int k = 0; k < ne; ++k) {
            const float ls = logits[((int64_t)(e0 + k * es) * C + c) * Bs + bl] - lz[k * kCbThreads + tx];
            const float nm = fmaxf(m, ls);
            s = s * expf(m - nm) + expf(ls - nm);
            m = nm;
        }
        if (ok) out[(int64_t)b * C + c] = ne > 0 ? m + logf(s) - sub : -INFINITY;
    }
}

// Aleatoric / epistemic decomposition over T stochastic forwards (uncertainty_estimation.py:37-58, 61-102 upstream):
// one wave per image, lanes over classes.  Pass 1: per draw p_hat = softmax(logits) (or softplus-normalised),
// accumulate sum logits, sum p, sum p^2.  Pass 2 recomputes p_hat to sum (p - p_bar)^2 without cancellation.
__global__ __launch_bounds__(64) void uncertainty_kernel(const float* __restrict__ logits, int T, int B, int C, int normalized,
                                                         float* __restrict__ pred, float* __restrict__ epi,
                                                         float* __restrict__ ale) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    float s_l[kMaxPerLane], s_p[kMaxPerLane], s_p2[kMaxPerLane], s_d2[kMaxPerLane];
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) { s_l[i] = 0.f; s_p[i] = 0.f; s_p2[i] = 0.f; s_d2[i] = 0.f; }
    for (int pass = 0; pass < 2; ++pass) {
        for (int t = 0; t < T; ++t) {
            const float* row = logits + ((int64_t)t * B + b) * C;
            float v[kMaxPerLane], q[kMaxPerLane];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) {
                const int c = lane + i * 64;
                v[i] = c < C ? row[c] : -INFINITY;
                mx = fmaxf(mx, v[i]);
            }
            mx = bbb::wave_max(mx);
            float z = 0.0f;
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) {
                const int c = lane + i * 64;
                q[i] = 0.0f;
                if (c < C) q[i] = normalized ? (v[i] > 20.0f ? v[i] : log1pf(expf(v[i]))) : expf(v[i] - mx);
                z += q[i];
            }
            z = bbb::wave_sum_all(z);
            const float iz = 1.0f / z;
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) {
                const int c = lane + i * 64;
                if (c < C) {
                    const float ph = q[i] * iz;
                    if (pass == 0) { s_l[i] += v[i]; s_p[i] += ph; s_p2[i] += ph * ph; }
                    else { const float d = ph - s_p[i]; s_d2[i] += d * d; }       // s_p holds p_bar in pass 1
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int i = 0; i < kMaxPerLane; ++i) s_p[i] /= (float)T;
        }
    }
#pragma unroll
    for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + i * 64;
        if (c < C) {
            const int64_t o = (int64_t)b * C + c;
            pred[o] = s_l[i] / (float)T;
            epi[o] = s_d2[i] / (float)T;
            ale[o] = s_p[i] - s_p2[i] / (float)T;
        }
    }
}

// [R][Ccols] -> [Ccols][R] through a padded 32x32 LDS tile (NCHW batch -> batch-innermost: R = B, Ccols = C*H*W).
// The tile grid is flattened into blockIdx.x (gx column tiles per row of tiles): neither extent meets the 65535 limit of
// gridDim.y, so an output map of any size transposes (64 channels x 224 x 224 per image: 100 352 row tiles).
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, unsigned gx) {
    __shared__ float tile[32][33];
    const unsigned ty_tile = blockIdx.x / gx;
    const int c0 = (int)(blockIdx.x - ty_tile * gx) * 32, r0 = (int)ty_tile * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + i * 8][tx] = in[(int64_t)r * Cc + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < R && c < Cc) out[(int64_t)c * R + r] = tile[tx][ty + i * 8];
    }
}

// im2col of an NCHW batch into the operand of the first layer's weight-gradient GEMM (training extension):
//   out[(p*B + b)*Jp + j] = x[b][ci][oh*s - pad + r*d][ow*s - pad + q*d]   (0 outside the image and for the pad columns j >= J)
// with p = oh*Wo + ow and j = (ci*kh + r)*kw + q.  One thread per output element, j innermost (coalesced writes; the 6 MB
// input stays in L2).
__global__ __launch_bounds__(256) void im2col_pbj_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t total, int B,
                                                         int Cin, int H, int W, int Wo, int kh, int kw, int sh, int sw, int ph,
                                                         int pw, int dh, int dw, int J, int Jp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % Jp);
    int64_t t = i / Jp;
    const int b = (int)(t % B);
    const int p = (int)(t / B);
    float v = 0.0f;
    if (j < J) {
        const int q = j % kw, r = (j / kw) % kh, ci = j / (kw * kh);
        const int oh = p / Wo, ow = p - oh * Wo;
        const int ih = oh * sh - ph + r * dh, iw = ow * sw - pw + q * dw;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[(((int64_t)b * Cin + ci) * H + ih) * W + iw];
    }
    out[i] = v;
}

// Weights of the input-gradient convolution (training extension): out[e][ci][n][r][q] = w[e][n][ci][kh-1-r][kw-1-q] -- the
// spatial flip and the channel transpose of conv2d_chwn_input_grad in one pass (coalesced writes; reads hit L2).
// (w1 != nullptr: draws e >= e1 read w1's draw e - e1 -- the mean and variance weights of an LRT layer as one two-draw operand)
__global__ __launch_bounds__(256) void flip_transpose_w_kernel(const float* __restrict__ w, const float* __restrict__ w1, int64_t e1,
                                                               float* __restrict__ out, int64_t total, int Cout, int Cin, int khkw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % khkw);
    int64_t r = i / khkw;
    const int n = (int)(r % Cout);
    r /= Cout;
    const int ci = (int)(r % Cin);
    int64_t e = r / Cin;
    const float* src = w;
    if (w1 != nullptr && e >= e1) src = w1, e -= e1;
    out[i] = src[((e * Cout + n) * Cin + ci) * khkw + (khkw - 1 - t)];
}

// Several weight sets in ONE launch (every layer's input-gradient weights of a training step, up front): block -> segment by a
// prefix table, then the one-thread-per-output gather above.
constexpr int kMaxFlipSegs = 16;
struct FlipMultiArgs {
    const float* w0[kMaxFlipSegs];
    const float* w1[kMaxFlipSegs];
    float* out[kMaxFlipSegs];
    int64_t total[kMaxFlipSegs];
    int64_t e1[kMaxFlipSegs];
    int32_t cout[kMaxFlipSegs], cin[kMaxFlipSegs], khkw[kMaxFlipSegs];
    uint32_t block_begin[kMaxFlipSegs + 1];
    int32_t n;
};
__global__ __launch_bounds__(256) void flip_transpose_w_multi_kernel(const FlipMultiArgs a) {
    int sgi = 0;
    while (sgi + 1 < a.n && blockIdx.x >= a.block_begin[sgi + 1]) ++sgi;
    const int64_t i = (int64_t)(blockIdx.x - a.block_begin[sgi]) * blockDim.x + threadIdx.x;
    if (i >= a.total[sgi]) return;
    const int Cout = a.cout[sgi], Cin = a.cin[sgi], khkw = a.khkw[sgi];
    const int t = (int)(i % khkw);
    int64_t r = i / khkw;
    const int n = (int)(r % Cout);
    r /= Cout;
    const int ci = (int)(r % Cin);
    int64_t e = r / Cin;
    const float* src = a.w0[sgi];
    if (a.w1[sgi] != nullptr && e >= a.e1[sgi]) src = a.w1[sgi], e -= a.e1[sgi];
    a.out[sgi][i] = src[((e * Cout + n) * Cin + ci) * khkw + (khkw - 1 - t)];
}

static int launch_flip_transpose(const float* w0, const float* w1, int64_t e1, float* out, int64_t draws, int cout, int cin, int khkw,
                                 hipStream_t stream) {
    const int64_t total = draws * cout * cin * khkw;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(flip_transpose_w_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, w0, w1, e1, out, total, cout, cin, khkw);
    return (int)hipGetLastError();
}

// Batched strided transpose (training extension: the operand permutations of the role-swapped weight-gradient launch):
//   out[i1*ob1 + i2*ob2 + c*oc + r] = in[i1*ib1 + i2*ib2 + r*ir + c],  r < R, c < C, (i1, i2) < (nb1, nb2)
// i.e. every batch entry is an [R][C] matrix with contiguous columns that lands as a [C][R] matrix with contiguous rows.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc,
                                                                int nb2, int64_t ib1, int64_t ib2, int64_t ir, int64_t ob1,
                                                                int64_t ob2, int64_t oc) {
    __shared__ float tile[32][33];
    const int i1 = blockIdx.z / nb2, i2 = blockIdx.z - i1 * nb2;
    const float* src = in + i1 * ib1 + i2 * ib2;
    float* dst = out + i1 * ob1 + i2 * ob2;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        if (r < R && c < Cc) tile[ty + i * 8][tx] = src[(int64_t)r * ir + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < R && c < Cc) dst[(int64_t)c * oc + r] = tile[tx][ty + i * 8];
    }
}

// The same tile transpose with three batch dimensions and a summed one:
//   out[i1*ob1 + i2*ob2 + i3*ob3 + c*oc + r] = sum_{s < ns, ascending} in[i1*ib1 + i2*ib2 + i3*ib3 + s*is + r*ir + c]
// (the S batch chunks of a role-swapped weight gradient summed in a fixed order while the taps move innermost; ns = 1 and a
// split batch index as i2 writes an output gradient straight into the chunked weight-operand layout).
struct TransposeExArgs {
    int32_t R, C, nb2, nb3, ns;
    int64_t ib1, ib2, ib3, is, ir, ob1, ob2, ob3, oc;
    int64_t sq;          // != 0: the squares of the outputs are written too, `sq` elements behind them (x and x^2 of an LRT weight gradient)
};
__global__ __launch_bounds__(256) void transpose_sum_batched_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                    const TransposeExArgs a) {
    __shared__ float tile[32][33];
    const int i3 = blockIdx.z % a.nb3, i12 = blockIdx.z / a.nb3;
    const int i2 = i12 % a.nb2, i1 = i12 / a.nb2;
    const float* src = in + i1 * a.ib1 + i2 * a.ib2 + i3 * a.ib3;
    float* dst = out + i1 * a.ob1 + i2 * a.ob2 + i3 * a.ob3;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        if (r < a.R && c < a.C) {
            const float* q = src + (int64_t)r * a.ir + c;
            float acc = q[0];
            for (int s = 1; s < a.ns; ++s) acc += q[(int64_t)s * a.is];
            tile[ty + i * 8][tx] = acc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < a.R && c < a.C) {
            const float v = tile[tx][ty + i * 8];
            dst[(int64_t)c * a.oc + r] = v;
            if (a.sq != 0) dst[a.sq + (int64_t)c * a.oc + r] = v * v;
        }
    }
}

// Few rows (R <= 32: the pixels of a small output map, the taps of a kernel) landing contiguously (oc == R): a 32 x 32 tile would be
// mostly empty and its 4R-byte output runs uncoalesced (measured 0.5 TB/s on the weight-gradient operands of the metric shape).
// One block moves blockDim.x columns of one batch entry: coalesced row reads -> LDS [column][R | 1] -> ONE linear, fully coalesced
// write of the blockDim.x * R contiguous outputs.
__global__ __launch_bounds__(256) void transpose_few_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                 const TransposeExArgs a) {
    __shared__ float tile[256 * 33];
    const int i3 = blockIdx.z % a.nb3, i12 = blockIdx.z / a.nb3;
    const int i2 = i12 % a.nb2, i1 = i12 / a.nb2;
    const float* src = in + i1 * a.ib1 + i2 * a.ib2 + i3 * a.ib3;
    float* dst = out + i1 * a.ob1 + i2 * a.ob2 + i3 * a.ob3;
    const int R = a.R, Rp = R | 1, nt = blockDim.x, tid = threadIdx.x;
    const int c0 = blockIdx.x * nt, c = c0 + tid;
    if (c < a.C) {
        for (int r = 0; r < R; ++r) {
            const float* q = src + (int64_t)r * a.ir + c;
            float acc = q[0];
            for (int s = 1; s < a.ns; ++s) acc += q[(int64_t)s * a.is];
            tile[tid * Rp + r] = acc;
        }
    }
    __syncthreads();
    const int ncol = min(nt, a.C - c0), total = ncol * R;
    float* d = dst + (int64_t)c0 * R;
    for (int i = tid; i < total; i += nt) {
        const int cc = i / R, r = i - cc * R;
        d[i] = tile[cc * Rp + r];
    }
}

static int launch_transpose_ex(const float* in, float* out, const TransposeExArgs& a, int64_t n, hipStream_t stream) {
    // (large operands only: on the small ones of a one-draw step the tiles' parallelism wins, above all with a summed dimension)
    if (a.R <= 32 && a.oc == a.R && a.ns == 1 && a.sq == 0 && n * a.R * a.C >= (1 << 20)) {
        const int nt = a.C <= 64 ? 64 : (a.C <= 128 ? 128 : 256);
        hipLaunchKernelGGL(transpose_few_rows_kernel, dim3((a.C + nt - 1) / nt, 1, (unsigned)n), dim3(nt), 0, stream, in, out, a);
    } else {
        hipLaunchKernelGGL(transpose_sum_batched_kernel, dim3((a.C + 31) / 32, (a.R + 31) / 32, (unsigned)n), dim3(256), 0, stream, in, out, a);
    }
    return (int)hipGetLastError();
}

// y[e] = act(act_mu + sqrt(act_var) * eps[e]) for E draws of ONE pair of LRT moments (batch-innermost layout).  Used when the
// input of an LRT layer is shared by all draws (the first layer): the two contractions run once, only the noise differs.
// eps element index = the canonical NCHW index ((b*C + n)*HW + pix) of draw e's output, as in the GEMM epilogue; one
// thread takes 4 consecutive pixels (one Philox call) when HW % 4 == 0, one pixel otherwise.
template <int PPT>
__global__ __launch_bounds__(256) void lrt_sample_chwn_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              float* __restrict__ y, int E, int C, int HW, int B, int act,
                                                              uint32_t k0, uint32_t k1, uint32_t call0, uint32_t stream_id,
                                                              const uint32_t* __restrict__ call_dev, int b_off) {
    const int64_t HWq = HW / PPT;
    const int64_t total = (int64_t)E * C * HWq * B;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i % B);
    int64_t t = i / B;
    const int q = (int)(t % HWq);
    t /= HWq;
    const int n = (int)(t % C);
    const int e = (int)(t / C);
    const uint32_t call = call0 + (call_dev ? *call_dev : 0u) + (uint32_t)e;
    const uint64_t idx0 = (uint64_t)(((int64_t)(b + b_off) * C + n) * HW + (int64_t)q * PPT);   // keyed by the GLOBAL image index
    float z4[4];
    bbb::normal4(idx0 >> 2, stream_id, call, k0, k1, z4);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int64_t o = ((int64_t)n * HW + (int64_t)q * PPT + j) * B + b;          // offset inside one draw's slab
        const int c = PPT == 4 ? j : (int)(idx0 & 3);
        const float z = c == 0 ? z4[0] : c == 1 ? z4[1] : c == 2 ? z4[2] : z4[3];
        const float v = mu[o] + __builtin_amdgcn_sqrtf(var[o]) * z;
        y[(int64_t)e * C * HW * B + o] = bbb::apply_act(v, act);
    }
}

// y = act_mu + sqrt(act_var) * eps over `draws` contiguous NCHW slabs of n elements each (element i of draw e uses noise
// element i of call call0 + e): the LRT sampling step on its own, for callers that computed the two moments separately.
__global__ __launch_bounds__(256) void lrt_sample_nchw_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                              float* __restrict__ y, int64_t n, int draws, uint32_t k0, uint32_t k1,
                                                              uint32_t call0, uint32_t stream_id, const uint32_t* __restrict__ call_dev) {
    const int64_t groups = (n + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * draws) return;
    const int e = (int)(i / groups);
    const int64_t g = i - (int64_t)e * groups;
    float z[4];
    bbb::normal4((uint64_t)g, stream_id, call0 + (call_dev ? *call_dev : 0u) + (uint32_t)e, k0, k1, z);
    const int64_t base = (int64_t)e * n + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (g * 4 + j < n) y[base + j] = mu[base + j] + __builtin_amdgcn_sqrtf(var[base + j]) * z[j];
}

}  // namespace

namespace {
int mc_tail_launch(const float* logits, int units, int slices, int unit_off, int batch_slice, int classes, int mean_over, float* lse_out,
                   const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter, uint32_t counter_add, int grp, int goff, void* stream) {
    if ((kl_out != nullptr && kl_in == nullptr) || (((uintptr_t)kl_in | (uintptr_t)kl_out | (uintptr_t)counter) & 3u) != 0) return BBB_EINVAL;
    if (logits == nullptr || lse_out == nullptr || units <= 0 || slices <= 0 || unit_off < 0 || batch_slice <= 0 || classes <= 0 ||
        mean_over < 0 || units > 4096 || grp < 0 || goff < 0 || (grp > 0 && (goff >= grp || (int64_t)grp * slices < (int64_t)units + goff)))
        return BBB_EINVAL;
    if ((((uintptr_t)logits | (uintptr_t)lse_out) & 3u) != 0) return BBB_EALIGN;
    const float sub = mean_over > 0 ? logf((float)mean_over) : 0.0f;
    const int64_t batch = (int64_t)batch_slice * slices;
    if (batch > 0x7fffffffLL) return BBB_ESHAPE;
    const int blocks = (int)((batch + kCbThreads - 1) / kCbThreads);
    const int per_slice = grp > 0 ? grp : (units + slices - 1) / slices;   // local units an image reduces over, at most
    const int mx = per_slice > classes ? per_slice : classes;
    const int ny = mx < kCbRows ? mx : kCbRows;
    hipLaunchKernelGGL(mc_tail_cb_kernel, dim3(blocks), dim3(kCbThreads, ny), (size_t)per_slice * kCbThreads * sizeof(float),
                       (hipStream_t)stream, logits, units, batch_slice, slices, unit_off, classes, sub, lse_out, kl_in, kl_scale, kl_out,
                       counter, counter_add, grp, goff);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bbb_mc_tail_units_step(const float* logits, int units, int slices, int unit_off, int batch_slice, int classes,
                                      int mean_over, float* lse_out, const float* kl_in, float kl_scale, float* kl_out,
                                      uint32_t* counter, uint32_t counter_add, void* stream) {
    return mc_tail_launch(logits, units, slices, unit_off, batch_slice, classes, mean_over, lse_out, kl_in, kl_scale, kl_out, counter,
                          counter_add, 0, 0, stream);
}

extern "C" int bbb_mc_tail_groups_step(const float* logits, int groups, int draws, int batch, int classes, int mean_over,
                                       float* lse_out, const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter,
                                       uint32_t counter_add, void* stream) {
    if (groups <= 0 || draws <= 0 || (int64_t)groups * draws > 4096) return BBB_EINVAL;
    return mc_tail_launch(logits, groups * draws, groups, 0, batch, classes, mean_over, lse_out, kl_in, kl_scale, kl_out, counter,
                          counter_add, draws, 0, stream);
}

extern "C" int bbb_mc_tail_share_step(const float* logits, int slabs, int steps, int draws, int first_off, int batch, int classes,
                                      float* lse_out, const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter,
                                      uint32_t counter_add, void* stream) {
    if (slabs <= 0 || steps <= 0 || draws <= 0 || first_off < 0 || slabs > 4096) return BBB_EINVAL;
    return mc_tail_launch(logits, slabs, steps, 0, batch, classes, 0, lse_out, kl_in, kl_scale, kl_out, counter, counter_add, draws,
                          first_off, stream);
}

extern "C" int bbb_mc_tail_units(const float* logits, int units, int slices, int unit_off, int batch_slice, int classes, int mean_over,
                                 float* lse_out, void* stream) {
    return bbb_mc_tail_units_step(logits, units, slices, unit_off, batch_slice, classes, mean_over, lse_out, nullptr, 0.0f, nullptr,
                                  nullptr, 0u, stream);
}

namespace {
// Backward of the tail (training extension; main_bayesian.py:49-53 under autograd): lse[b][c] = logsumexp_e ls[e][b][c] - log(mean_over),
// ls[e][b][:] = log_softmax(logits[e][:][b]).  With g = d loss / d lse:
//   H[e][b][c] = g[b][c] * exp(ls[e][b][c] - (lse[b][c] + log(mean_over)))          (softmax over the draws)
//   d loss / d logits[e][k][b] = H[e][b][k] - exp(ls[e][b][k]) * sum_c H[e][b][c]
// One thread per (draw, image); three passes over the classes (logits are [E][C][B]: consecutive lanes = consecutive images).
// ELBO = true (bbb_elbo_cb_bwd): g[b][c] is not read but IS the gradient of nll_loss(mean) * train_size w.r.t. lse:
// -(train_size / B) * g_loss at c == target[b], 0 elsewhere (gscale = train_size / B); thread 0 also writes d loss / d kl = beta * g_loss.
template <bool ELBO>
__global__ __launch_bounds__(256) void mc_tail_cb_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                             const float* __restrict__ g, float* __restrict__ g_logits, int E, int C, int B,
                                                             float add, const int64_t* __restrict__ target, const float* __restrict__ g_loss,
                                                             float gscale, float beta, const float* __restrict__ beta_dev,
                                                             float* __restrict__ g_kl) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float gl = 0.0f;
    if (ELBO) {
        gl = *g_loss;
        if (i == 0 && g_kl != nullptr) *g_kl = (beta_dev ? *beta_dev : beta) * gl;
    }
    if (i >= (int64_t)E * B) return;
    const int e = (int)(i / B), b = (int)(i - (int64_t)e * B);
    const float* p = logits + (int64_t)e * C * B + b;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, p[(int64_t)c * B]);
    float se = 0.0f;
    for (int c = 0; c < C; ++c) se += expf(p[(int64_t)c * B] - mx);
    const float lz = mx + logf(se);
    float* o = g_logits + (int64_t)e * C * B + b;
    if (ELBO) {
        const int64_t t = target[b];
        float h = 0.0f, sum_h = 0.0f;
        if (t >= 0 && t < C) {
            const float ls = p[t * B] - lz;
            h = -gscale * gl * expf(ls - (lse[(int64_t)b * C + t] + add));
            sum_h = h;
        }
        for (int c = 0; c < C; ++c) {
            const float ls = p[(int64_t)c * B] - lz;
            o[(int64_t)c * B] = (c == t ? h : 0.0f) - expf(ls) * sum_h;
        }
        return;
    }
    float sum_h = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float ls = p[(int64_t)c * B] - lz;
        sum_h += g[(int64_t)b * C + c] * expf(ls - (lse[(int64_t)b * C + c] + add));
    }
    for (int c = 0; c < C; ++c) {
        const float ls = p[(int64_t)c * B] - lz;
        const float h = g[(int64_t)b * C + c] * expf(ls - (lse[(int64_t)b * C + c] + add));
        o[(int64_t)c * B] = h - expf(ls) * sum_h;
    }
}

// loss = nll_loss(lse, target, mean) * train_size + beta * kl (metrics.py:7-14 upstream) in one block: per-thread partial sums over the
// images in a fixed stride, then a fixed tree -- the same value on every run.  Targets outside [0, C) contribute nothing.
__global__ __launch_bounds__(256) void elbo_cb_kernel(const float* __restrict__ lse, const int64_t* __restrict__ target,
                                                      const float* __restrict__ kl, float beta, const float* __restrict__ beta_dev,
                                                      float train_size, int B, int C, float* __restrict__ loss) {
    __shared__ float part[256];
    float acc = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int64_t t = target[b];
        if (t >= 0 && t < C) acc += lse[(int64_t)b * C + t];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (-part[0] / (float)B) * train_size + (beta_dev ? *beta_dev : beta) * (*kl);
}
}  // namespace

extern "C" int bbb_mc_tail_cb_bwd(const float* logits, const float* lse, const float* g_lse, float* g_logits, int draws, int batch,
                                  int classes, int mean_over, void* stream) {
    if (logits == nullptr || lse == nullptr || g_lse == nullptr || g_logits == nullptr || draws <= 0 || batch <= 0 || classes <= 0 ||
        mean_over < 0)
        return BBB_EINVAL;
    if ((((uintptr_t)logits | (uintptr_t)lse | (uintptr_t)g_lse | (uintptr_t)g_logits) & 3u) != 0) return BBB_EALIGN;
    const int64_t n = (int64_t)draws * batch;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(mc_tail_cb_bwd_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, lse, g_lse, g_logits,
                       draws, classes, batch, mean_over > 0 ? logf((float)mean_over) : 0.0f, (const int64_t*)nullptr, (const float*)nullptr,
                       0.0f, 0.0f, (const float*)nullptr, (float*)nullptr);
    return (int)hipGetLastError();
}

extern "C" int bbb_elbo_cb_fwd(const float* lse, const int64_t* target, const float* kl, float beta, const float* beta_dev,
                               float train_size, int batch, int classes, float* loss_out, void* stream) {
    if (lse == nullptr || target == nullptr || kl == nullptr || loss_out == nullptr || batch <= 0 || classes <= 0) return BBB_EINVAL;
    if ((((uintptr_t)lse | (uintptr_t)kl | (uintptr_t)beta_dev | (uintptr_t)loss_out) & 3u) != 0 || ((uintptr_t)target & 7u) != 0)
        return BBB_EALIGN;
    hipLaunchKernelGGL(elbo_cb_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lse, target, kl, beta, beta_dev, train_size, batch,
                       classes, loss_out);
    return (int)hipGetLastError();
}

extern "C" int bbb_elbo_cb_bwd(const float* logits, const float* lse, const int64_t* target, const float* g_loss, float beta,
                               const float* beta_dev, float train_size, float* g_logits, float* g_kl, int draws, int batch, int classes,
                               int mean_over, void* stream) {
    if (logits == nullptr || lse == nullptr || target == nullptr || g_loss == nullptr || g_logits == nullptr || draws <= 0 ||
        batch <= 0 || classes <= 0 || mean_over < 0)
        return BBB_EINVAL;
    if ((((uintptr_t)logits | (uintptr_t)lse | (uintptr_t)g_loss | (uintptr_t)beta_dev | (uintptr_t)g_logits | (uintptr_t)g_kl) & 3u) != 0 ||
        ((uintptr_t)target & 7u) != 0)
        return BBB_EALIGN;
    const int64_t n = (int64_t)draws * batch;
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(mc_tail_cb_bwd_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, lse,
                       (const float*)nullptr, g_logits, draws, classes, batch, mean_over > 0 ? logf((float)mean_over) : 0.0f, target, g_loss,
                       train_size / (float)batch, beta, beta_dev, g_kl);
    return (int)hipGetLastError();
}

extern "C" int bbb_mc_tail_cb(const float* logits, int draws, int batch, int classes, int mean_over, float* lse_out,
                              void* stream) {
    if (draws > 512) return BBB_EINVAL;
    return bbb_mc_tail_units(logits, draws, 1, 0, batch, classes, mean_over, lse_out, stream);
}

extern "C" int bbb_uncertainty(const float* logits, int draws, int batch, int classes, int normalized, float* pred,
                               float* epistemic, float* aleatoric, void* stream) {
    if (logits == nullptr || pred == nullptr || epistemic == nullptr || aleatoric == nullptr || draws <= 0 || batch <= 0 ||
        classes <= 0 || classes > 64 * kMaxPerLane)
        return BBB_EINVAL;
    if ((((uintptr_t)logits | (uintptr_t)pred | (uintptr_t)epistemic | (uintptr_t)aleatoric) & 3u) != 0) return BBB_EALIGN;
    hipLaunchKernelGGL(uncertainty_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, logits, draws, batch, classes,
                       normalized ? 1 : 0, pred, epistemic, aleatoric);
    return (int)hipGetLastError();
}

extern "C" int bbb_transpose2d(const float* in, float* out, int64_t rows, int64_t cols, void* stream) {
    if (in == nullptr || out == nullptr || rows <= 0 || cols <= 0 || rows > 0x7fffffffLL || cols > 0x7fffffffLL) return BBB_EINVAL;
    if ((((uintptr_t)in | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int64_t gx = (cols + 31) / 32, gy = (rows + 31) / 32;
    if (gx * gy > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)(gx * gy)), dim3(256), 0, (hipStream_t)stream, in, out, (int)rows,
                       (int)cols, (unsigned)gx);
    return (int)hipGetLastError();
}

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

extern "C" int bbb_lrt_sample_chwn(const float* act_mu, const float* act_var, float* y, int draws, int channels, int pixels,
                                   int batch, int b_offset, int act, uint64_t seed, uint32_t call0, uint32_t stream_id,
                                   const uint32_t* call_dev, void* stream) {
    if (act_mu == nullptr || act_var == nullptr || y == nullptr || draws <= 0 || channels <= 0 || pixels <= 0 || batch <= 0 ||
        b_offset < 0 || act < 0 || act > 2)
        return BBB_EINVAL;
    if ((((uintptr_t)act_mu | (uintptr_t)act_var | (uintptr_t)y) & 3u) != 0) return BBB_EALIGN;
    const bool quad = pixels % 4 == 0;
    const int64_t total = (int64_t)draws * channels * (quad ? pixels / 4 : pixels) * batch;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (quad)
        hipLaunchKernelGGL(lrt_sample_chwn_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, act_mu, act_var, y,
                           draws, channels, pixels, batch, act, k0, k1, call0, stream_id, call_dev, b_offset);
    else
        hipLaunchKernelGGL(lrt_sample_chwn_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, act_mu, act_var, y,
                           draws, channels, pixels, batch, act, k0, k1, call0, stream_id, call_dev, b_offset);
    return (int)hipGetLastError();
}

extern "C" int bbb_lrt_sample_nchw(const float* act_mu, const float* act_var, float* y, int64_t n, int draws, uint64_t seed,
                                   uint32_t call0, uint32_t stream_id, const uint32_t* call_dev, void* stream) {
    if (act_mu == nullptr || act_var == nullptr || y == nullptr || n <= 0 || draws <= 0) return BBB_EINVAL;
    if ((((uintptr_t)act_mu | (uintptr_t)act_var | (uintptr_t)y) & 3u) != 0) return BBB_EALIGN;
    const int64_t total = ((n + 3) >> 2) * draws;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(lrt_sample_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, act_mu, act_var, y, n, draws,
                       (uint32_t)seed, (uint32_t)(seed >> 32), call0, stream_id, call_dev);
    return (int)hipGetLastError();
}

extern "C" int bbb_transpose_batched(const float* in, float* out, int rows, int cols, int nb1, int nb2, int64_t in_b1, int64_t in_b2,
                                     int64_t in_row, int64_t out_b1, int64_t out_b2, int64_t out_col, void* stream) {
    if (in == nullptr || out == nullptr || rows <= 0 || cols <= 0 || nb1 <= 0 || nb2 <= 0) return BBB_EINVAL;
    if ((((uintptr_t)in | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int64_t nb = (int64_t)nb1 * nb2, gy = (rows + 31) / 32;
    if (nb > 65535 || gy > 65535) return BBB_ESHAPE;
    if (rows <= 32 && out_col == rows && nb * rows * cols >= (1 << 20)) {
        TransposeExArgs a;
        a.R = rows, a.C = cols, a.nb2 = nb2, a.nb3 = 1, a.ns = 1;
        a.ib1 = in_b1, a.ib2 = in_b2, a.ib3 = 0, a.is = 0, a.ir = in_row;
        a.ob1 = out_b1, a.ob2 = out_b2, a.ob3 = 0, a.oc = out_col, a.sq = 0;
        return launch_transpose_ex(in, out, a, nb, (hipStream_t)stream);
    }
    hipLaunchKernelGGL(transpose_batched_kernel, dim3((cols + 31) / 32, (unsigned)gy, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, out,
                       rows, cols, nb2, in_b1, in_b2, in_row, out_b1, out_b2, out_col);
    return (int)hipGetLastError();
}

extern "C" int bbb_transpose_sum_batched(const float* in, float* out, int rows, int cols, const int32_t* nb, const int64_t* in_b,
                                         const int64_t* out_b, int64_t in_row, int64_t out_col, int nsum, int64_t in_sum,
                                         int64_t square_off, void* stream) {
    if (in == nullptr || out == nullptr || nb == nullptr || in_b == nullptr || out_b == nullptr || rows <= 0 || cols <= 0 ||
        nb[0] <= 0 || nb[1] <= 0 || nb[2] <= 0 || nsum <= 0 || square_off < 0)
        return BBB_EINVAL;
    if ((((uintptr_t)in | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int64_t n = (int64_t)nb[0] * nb[1] * nb[2], gy = (rows + 31) / 32;
    if (n > 65535 || gy > 65535) return BBB_ESHAPE;
    TransposeExArgs a;
    a.R = rows, a.C = cols, a.nb2 = nb[1], a.nb3 = nb[2], a.ns = nsum;
    a.ib1 = in_b[0], a.ib2 = in_b[1], a.ib3 = in_b[2], a.is = in_sum, a.ir = in_row;
    a.ob1 = out_b[0], a.ob2 = out_b[1], a.ob3 = out_b[2], a.oc = out_col, a.sq = square_off;
    return launch_transpose_ex(in, out, a, n, (hipStream_t)stream);
}

extern "C" int bbb_im2col_pbj(const float* x, float* out, const bbb_conv_desc_t* d, void* stream) {
    if (x == nullptr || out == nullptr || d == nullptr || d->batch <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->kh <= 0 ||
        d->kw <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0)
        return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    const int J = d->cin * d->kh * d->kw, Jp = (J + 3) / 4 * 4;
    const int64_t total = (int64_t)ho * wo * d->batch * Jp;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(im2col_pbj_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, total, d->batch, d->cin,
                       d->h, d->w, wo, d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w, d->dil_h, d->dil_w, J, Jp);
    return (int)hipGetLastError();
}

extern "C" int bbb_flip_transpose_w(const float* w, float* out, int64_t draws, int cout, int cin, int khkw, void* stream) {
    if (w == nullptr || out == nullptr || draws <= 0 || cout <= 0 || cin <= 0 || khkw <= 0) return BBB_EINVAL;
    if ((((uintptr_t)w | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    return launch_flip_transpose(w, nullptr, 0, out, draws, cout, cin, khkw, (hipStream_t)stream);
}

extern "C" int bbb_flip_transpose_w_multi(const bbb_flip_seg_t* segs, int n_segs, void* stream) {
    if (segs == nullptr || n_segs <= 0 || n_segs > kMaxFlipSegs) return BBB_EINVAL;
    FlipMultiArgs a;
    a.n = n_segs;
    uint64_t blocks = 0;
    for (int i = 0; i < n_segs; ++i) {
        const bbb_flip_seg_t& g = segs[i];
        if (g.w0 == nullptr || g.out == nullptr || g.draws <= 0 || g.cout <= 0 || g.cin <= 0 || g.khkw <= 0 ||
            (g.w1 != nullptr && (g.draws & 1)))
            return BBB_EINVAL;
        if ((((uintptr_t)g.w0 | (uintptr_t)g.w1 | (uintptr_t)g.out) & 3u) != 0) return BBB_EALIGN;
        a.w0[i] = g.w0; a.w1[i] = g.w1; a.out[i] = g.out;
        a.e1[i] = g.w1 != nullptr ? g.draws / 2 : g.draws;
        a.cout[i] = g.cout; a.cin[i] = g.cin; a.khkw[i] = g.khkw;
        a.total[i] = g.draws * g.cout * g.cin * g.khkw;
        a.block_begin[i] = (uint32_t)blocks;
        blocks += (uint64_t)((a.total[i] + 255) / 256);
        if (blocks > 0x7fffffffULL) return BBB_ESHAPE;
    }
    a.block_begin[n_segs] = (uint32_t)blocks;
    hipLaunchKernelGGL(flip_transpose_w_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int bbb_flip_transpose_w_pair(const float* w0, const float* w1, float* out, int64_t draws_each, int cout, int cin, int khkw,
                                         void* stream) {
    if (w0 == nullptr || w1 == nullptr || out == nullptr || draws_each <= 0 || cout <= 0 || cin <= 0 || khkw <= 0) return BBB_EINVAL;
    if ((((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    return launch_flip_transpose(w0, w1, draws_each, out, 2 * draws_each, cout, cin, khkw, (hipStream_t)stream);
}
