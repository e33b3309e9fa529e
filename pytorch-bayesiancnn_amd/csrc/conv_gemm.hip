// Implicit-im2col GEMM on the gfx950 fp32 matrix cores, batched over Monte-Carlo draws.
//
//   y[e][b][n][oh][ow] = bias[e][n] + sum_k  w[e][n][k] * x[e|shared][b][ci][oh*s - p + r*d][ow*s - p + q*d],
//   k = (ci, r, q) in the weight's own [Cin][kh][kw] order.
//
// The contraction is computed as D[n][m] = W[n][k] * X[k][m] (weights as the MFMA "A" operand, the im2col
// matrix as "B"), so that in the v_mfma_f32_32x32x2_f32 result layout consecutive lanes hold consecutive
// output pixels of one channel: NCHW stores coalesce without a transpose.  v_mfma_f32_32x32x2_f32 is exact
// fp32 (an fmaf chain), 64 cycles per issue, so the kernel is matrix-pipe bound and the staging work
// (im2col gather, LDS transposes) hides in the MFMA shadow.
//
// Block = 256 threads (4 waves), tile = 64 output channels x BM output pixels x 32 k.  Register-staged
// double buffering: tile t+1's global loads are issued before tile t's MFMAs and written to the other LDS
// buffer after them.  The (ci, r, q) decode of each k-slice is computed once per tile by 32 lanes into a
// small LDS table (the k index of every gather load is wave-uniform).
//
// LRT variant (layers/BBB_LRT/BBBConv.py:71-79): a second accumulator set multiplies sigma^2 by x*x (squared
// in registers from the same LDS tile); the epilogue emits act_mu + sqrt(1e-16 + act_var) * eps with eps
// from Philox, indexed by output element.
//
// Replaces F.conv2d / F.linear of layers/BBB/BBBConv.py:77, layers/BBB/BBBLinear.py:70,
// layers/BBB_LRT/BBBConv.py:71-81, layers/BBB_LRT/BBBLinear.py:65-73.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kThreads = 256;
constexpr int BN = 64;    // output channels per block
constexpr int LDW = BN + 1;

struct ConvArgs {
    const float* x;
    const float* w;      // weights (LRT: w_mu)
    const float* w2;     // LRT: sigma^2
    const float* bias;   // (LRT: b_mu)
    const float* bias2;  // LRT: bias variance
    float* y;
    float* y_mu;         // LRT optional
    float* y_var;        // LRT optional
    const float* eps_ext;
    int64_t x_ds, w_ds, b_ds, y_ds;
    int32_t B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
    int32_t M, K, HoWo, khkw, act, sample;
    uint32_t k0, k1, call0, stream_id;
    const uint32_t* call_dev;
};


template <int BM, bool LRT, bool LINEAR>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(const ConvArgs p) {
    constexpr int BK = LRT ? 16 : 32;                    // k per LDS tile (LRT stages two weight tiles)
    constexpr int LDX = BM + 1;
    constexpr int NT = (BM == 128) ? 2 : 1;              // 32x32 MFMA tiles per wave (along n)
    constexpr int XPT = BK * BM / kThreads;              // gathered x elements per thread per tile
    constexpr int XKS = kThreads / BM;                   // k stride between a thread's gather loads
    constexpr int WSETS = LRT ? 2 : 1;
    constexpr int TPR = BK / 4;                          // threads per k-contiguous row (4 floats each)
    constexpr int RPP = kThreads / TPR;                  // rows per pass of the k-contiguous loaders
    constexpr int WPASS = BN / RPP;                      // passes over the 64 weight rows
    constexpr int XPASS = BM / RPP;                      // passes over the BM x rows (LINEAR)

    __shared__ float Ws[2][WSETS][BK * LDW];
    __shared__ float Xs[2][BK * LDX];
    __shared__ int32_t kt_off[2][BK];
    __shared__ int32_t kt_rs[2][BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int m0 = blockIdx.x * BM;
    const int M = p.M, K = p.K;

    const float* __restrict__ xg = p.x + (int64_t)e * p.x_ds;
    const float* __restrict__ wg = p.w + (int64_t)e * p.w_ds;
    const float* __restrict__ w2g = LRT ? p.w2 + (int64_t)e * p.w_ds : nullptr;

    // ---- wave -> sub-tile ----
    const int wn = (BM == 128) ? 0 : (wave >> 1) * 32;    // n offset of this wave inside the block tile
    const int wm = (BM == 128) ? wave * 32 : (wave & 1) * 32;

    // ---- per-thread gather coordinates (im2col variant) ----
    const int xm_l = LINEAR ? 0 : (tid % BM);
    const int xk_sub = LINEAR ? 0 : __builtin_amdgcn_readfirstlane(tid / BM);
    int64_t xbase = 0;
    int ih0 = 0, iw0 = 0;
    bool m_ok = false;
    if (!LINEAR) {
        const int mg = m0 + xm_l;
        m_ok = mg < M;
        const int mm = m_ok ? mg : 0;
        const int b = mm / p.HoWo;
        const int pix = mm - b * p.HoWo;
        const int oh = pix / p.Wo;
        const int ow = pix - oh * p.Wo;
        ih0 = oh * p.sh - p.ph;
        iw0 = ow * p.sw - p.pw;
        xbase = (int64_t)b * p.Cin * p.H * p.W + (int64_t)ih0 * p.W + iw0;
    }
    // ---- per-thread coordinates for k-contiguous loads (weights; x when LINEAR) ----
    const int wr = tid / TPR;            // row within a pass
    const int wkq = (tid % TPR) * 4;     // k offset of this thread's 4 consecutive elements

    float xreg[XPT];
    float wreg[WSETS][WPASS * 4];

    auto fill_ktab = [&](int tile, int buf) {
        if (!LINEAR && tid < BK) {
            const int k = tile * BK + tid;
            int off = 0, rs = 0x7fff;             // r = 0x7fff: fails every bounds test (k >= K)
            if (k < K) {
                const int ci = k / p.khkw;
                const int rq = k - ci * p.khkw;
                const int r = rq / p.kw;
                const int q = rq - r * p.kw;
                off = ci * p.H * p.W + r * p.dh * p.W + q * p.dw;
                rs = (r * p.dh) | ((q * p.dw) << 16);
            }
            kt_off[buf][tid] = off;
            kt_rs[buf][tid] = rs;
        }
    };

    auto load_tile = [&](int tile, int buf) {
        const int kbase = tile * BK;
        // weights: 64 rows x BK k in WPASS passes; 4 consecutive k per thread
#pragma unroll
        for (int s = 0; s < WSETS; ++s) {
            const float* __restrict__ src = (s == 0) ? wg : w2g;
#pragma unroll
            for (int pss = 0; pss < WPASS; ++pss) {
                const int n = n0 + pss * RPP + wr;
                const int k = kbase + wkq;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (n < p.Cout) {
                    const float* q = src + (int64_t)n * K + k;
                    if (k + 3 < K) {
                        const f32x4_u t = *reinterpret_cast<const f32x4_u*>(q);
                        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = q[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) wreg[s][pss * 4 + j] = v[j];
            }
        }
        if (LINEAR) {
            // x is [M][K] row-major: k-contiguous loads
#pragma unroll
            for (int pss = 0; pss < XPASS; ++pss) {
                const int m = m0 + pss * RPP + wr;
                const int k = kbase + wkq;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (m < M) {
                    const float* q = xg + (int64_t)m * K + k;
                    if (k + 3 < K) {
                        const f32x4_u t = *reinterpret_cast<const f32x4_u*>(q);
                        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = q[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) xreg[pss * 4 + j] = v[j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int kl = xk_sub + i * XKS;
                const int off = kt_off[buf][kl];
                const int rs = kt_rs[buf][kl];
                const int ih = ih0 + (rs & 0xffff);
                const int iw = iw0 + (rs >> 16);
                const bool ok = m_ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                xreg[i] = ok ? xg[xbase + off] : 0.0f;
            }
        }
    };

    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < WSETS; ++s)
#pragma unroll
            for (int pss = 0; pss < WPASS; ++pss)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    Ws[buf][s][(wkq + j) * LDW + pss * RPP + wr] = wreg[s][pss * 4 + j];
        if (LINEAR) {
#pragma unroll
            for (int pss = 0; pss < XPASS; ++pss)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    Xs[buf][(wkq + j) * LDX + pss * RPP + wr] = xreg[pss * 4 + j];
        } else {
#pragma unroll
            for (int i = 0; i < XPT; ++i) Xs[buf][(xk_sub + i * XKS) * LDX + xm_l] = xreg[i];
        }
    };

    f32x16 acc[NT];
    f32x16 accv[LRT ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; if (LRT) accv[t][r] = 0.0f; }
    }

    const int ntiles = (K + BK - 1) / BK;
    fill_ktab(0, 0);
    fill_ktab(1, 1);
    __syncthreads();
    load_tile(0, 0);
    store_tile(0);
    __syncthreads();

    const int lrow = lane & 31;
    const int lk = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1) < ntiles;
        if (more) load_tile(t + 1, cur ^ 1);     // global loads in flight during the MFMAs below
        // kt[cur] held tile t's decode, last read before the barrier that ended iteration t-1: refill it for
        // tile t+2 now; it is first read in iteration t+1, after the barrier below.
        if (t + 2 < ntiles) fill_ktab(t + 2, cur);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = kk * 2 + lk;
            const float b = Xs[cur][krow * LDX + wm + lrow];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float a = Ws[cur][0][krow * LDW + wn + nt * 32 + lrow];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
                if (LRT) {
                    const float a2 = Ws[cur][LRT ? 1 : 0][krow * LDW + wn + nt * 32 + lrow];
                    accv[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b * b, accv[nt], 0, 0, 0);
                }
            }
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[i][j], i = (r&3) + 8*(r>>2) + 4*(lane>>5) -> channel, j = lane&31 -> pixel ----
    const int mg = m0 + wm + lrow;
    if (mg < M) {
        const int b = mg / p.HoWo;
        const int pix = mg - b * p.HoWo;
        const int64_t ybase = (int64_t)e * p.y_ds + (int64_t)b * p.Cout * p.HoWo + pix;
        const float* __restrict__ bg = p.bias ? p.bias + (int64_t)e * p.b_ds : nullptr;
        const float* __restrict__ b2g = (LRT && p.bias2) ? p.bias2 + (int64_t)e * p.b_ds : nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (n < p.Cout) {
                    const int64_t o = ybase + (int64_t)n * p.HoWo;
                    float v = acc[nt][r] + (bg ? bg[n] : 0.0f);
                    if (LRT) {
                        const float var = 1e-16f + (accv[nt][r] + (b2g ? b2g[n] : 0.0f));
                        if (p.y_mu) p.y_mu[o] = v;
                        if (p.y_var) p.y_var[o] = var;
                        if (p.sample) {
                            float z;
                            if (p.eps_ext) {
                                z = p.eps_ext[o];
                            } else {
                                // element index inside this draw's [B][Cout][Ho][Wo] slab
                                const uint64_t idx = (uint64_t)((int64_t)b * p.Cout * p.HoWo + (int64_t)n * p.HoWo + pix);
                                float z4[4];
                                bbb::normal4(idx >> 2, p.stream_id, p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)e, p.k0, p.k1, z4);
                                const int c = (int)(idx & 3);
                                z = c == 0 ? z4[0] : c == 1 ? z4[1] : c == 2 ? z4[2] : z4[3];
                            }
                            v = v + __builtin_amdgcn_sqrtf(var) * z;
                        }
                    }
                    p.y[o] = bbb::apply_act(v, p.act);
                }
            }
        }
    }
}

int check_desc(const bbb_conv_desc_t* d, ConvArgs& a) {
    if (d == nullptr) return BBB_EINVAL;
    if (d->batch <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0 ||
        d->draws <= 0 || d->act < 0 || d->act > 2)
        return BBB_EINVAL;
    if (d->unit_div > 1 || d->x_unit_mod != 0 || d->b_offset != 0 || d->w_row_pitch != 0 || d->x_unit_div > 1 || d->x_unit_off != 0 || d->pool != 0 || d->w_tap_major != 0) return BBB_EINVAL;   // batch-innermost entries only
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    const int64_t M = (int64_t)d->batch * ho * wo;
    const int64_t K = (int64_t)d->cin * d->kh * d->kw;
    if (M > 0x7fffffffLL || K > 0x7fffffffLL) return BBB_ESHAPE;
    if ((int64_t)d->cin * d->h * d->w > 0x7fffffffLL) return BBB_ESHAPE;     // per-image offsets are int32
    if (d->h + d->pad_h + d->dil_h * d->kh >= 0x7000 || d->w + d->pad_w + d->dil_w * d->kw >= 0x7000) return BBB_ESHAPE;
    a.B = d->batch; a.Cin = d->cin; a.H = d->h; a.W = d->w; a.Cout = d->cout; a.kh = d->kh; a.kw = d->kw;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = ho; a.Wo = wo; a.M = (int32_t)M; a.K = (int32_t)K; a.HoWo = ho * wo; a.khkw = d->kh * d->kw;
    a.act = d->act;
    a.x_ds = d->x_draw_stride; a.w_ds = d->w_draw_stride; a.b_ds = d->b_draw_stride;
    a.y_ds = (int64_t)d->batch * d->cout * ho * wo;
    return 0;
}

bool is_linear(const ConvArgs& a) {
    return a.H == 1 && a.W == 1 && a.kh == 1 && a.kw == 1 && a.ph == 0 && a.pw == 0;
}

// Smaller pixel tile when the 128-wide grid would not give every CU a few blocks.
int pick_bm(const ConvArgs& a, int draws) {
    const int64_t nb = (int64_t)((a.Cout + BN - 1) / BN) * draws;
    const int64_t blocks128 = ((a.M + 127) / 128) * nb;
    return blocks128 >= 3 * 256 ? 128 : 64;
}

template <bool LRT>
int launch(const ConvArgs& a, int draws, hipStream_t st) {
    const int bm = pick_bm(a, draws);
    const dim3 grid((a.M + bm - 1) / bm, (a.Cout + BN - 1) / BN, draws);
    const bool lin = is_linear(a);
    if (bm == 128) {
        if (lin) hipLaunchKernelGGL((conv_gemm_kernel<128, LRT, true>), grid, dim3(kThreads), 0, st, a);
        else     hipLaunchKernelGGL((conv_gemm_kernel<128, LRT, false>), grid, dim3(kThreads), 0, st, a);
    } else {
        if (lin) hipLaunchKernelGGL((conv_gemm_kernel<64, LRT, true>), grid, dim3(kThreads), 0, st, a);
        else     hipLaunchKernelGGL((conv_gemm_kernel<64, LRT, false>), grid, dim3(kThreads), 0, st, a);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int bbb_conv2d_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias, float* y,
                              void* stream) {
    ConvArgs a = {};
    const int rc = check_desc(d, a);
    if (rc != 0) return rc;
    if (x == nullptr || w == nullptr || y == nullptr) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 3u) != 0) return BBB_EALIGN;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    return launch<false>(a, d->draws, (hipStream_t)stream);
}

extern "C" int bbb_lrt_conv2d_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                                  const float* b_mu, const float* b_var, float* y, float* act_mu_out, float* act_var_out,
                                  const float* eps_ext, uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                                  const uint32_t* call_dev, void* stream) {
    ConvArgs a = {};
    const int rc = check_desc(d, a);
    if (rc != 0) return rc;
    if (x == nullptr || w_mu == nullptr || w_var == nullptr || y == nullptr) return BBB_EINVAL;
    if ((b_mu == nullptr) != (b_var == nullptr)) return BBB_EINVAL;
    if (d->w_draw_stride != 0 || d->b_draw_stride != 0) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)w_mu | (uintptr_t)w_var | (uintptr_t)b_mu | (uintptr_t)b_var | (uintptr_t)y |
          (uintptr_t)act_mu_out | (uintptr_t)act_var_out | (uintptr_t)eps_ext) & 3u) != 0)
        return BBB_EALIGN;
    a.x = x; a.w = w_mu; a.w2 = w_var; a.bias = b_mu; a.bias2 = b_var; a.y = y;
    a.y_mu = act_mu_out; a.y_var = act_var_out; a.eps_ext = eps_ext;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.call0 = call0; a.stream_id = stream_id;
    a.sample = sample ? 1 : 0;
    a.call_dev = call_dev;
    return launch<true>(a, d->draws, (hipStream_t)stream);
}
