// Pixel-major implicit GEMM for the Monte-Carlo ensemble path on gfx950: batch-innermost activations,
// padding taps never computed.
//
// Layout: activations are [draw][C][H][W][B] (B innermost, B % 4 == 0); weights stay in the reference's
// [draw][Cout][Cin][kh][kw].  One workgroup owns ONE output pixel (oh, ow), 64 output channels and BM images:
//     D[n][b] = sum over ci and over the kernel taps (r, q) that fall INSIDE the image for this pixel
//               W[n][ci][r][q] * X[ci][oh*s - p + r*d][ow*s - p + q*d][b]
// so the contraction length is K_eff = Cin * nr * nq (nr, nq = number of in-bounds taps per axis) instead of
// Cin*kh*kw.  On AlexNet/CIFAR the 2x2 and 4x4 feature maps make more than half of the im2col matrix padding
// zeros; skipping them halves the matrix-core work (7.07 instead of 14.11 GFLOP per draw at bs=512).
// Every X row of a tile is one contiguous run of BM floats (16-byte vector loads straight into LDS rows),
// every output row is contiguous in b: fully coalesced on both sides, no per-element im2col arithmetic.
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32), weights as the "A" operand, X as "B" -> lanes = images.
// Workgroup = 256 threads = 4 waves, tile 64 (n) x BM (b) x 32 (k); register-staged double buffering;
// per-tile (k_eff -> weight offset, x row) tables computed by 32 lanes into LDS.
// Workgroups that share a weight tile (same draw, same 64 channels) are mapped to the same XCD (block id
// mod 8) so the tile stays in that XCD's private L2.
//
// LRT variant: second accumulator set for sigma^2 * x^2, epilogue act_mu + sqrt(1e-16 + act_var) * eps with eps
// indexed by the canonical NCHW element index (identical stream to the NCHW kernel and the oracle).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"
#include "pconv_args.h"
#include "pconv_body.cuh"
#include "pconv_bf16x3.cuh"

namespace {

using namespace pconv;

template <int BM, bool LRT, bool ILV, bool SEQ = false>
__global__ __launch_bounds__(kThreads) void pconv_gemm_kernel(const PConvArgs p) {
    // ---- block -> work item; weight-tile sharers on one XCD ----
    // Work items (group g = (draw, channel tile), m-tile j) in g-major order are cut into 8 equal contiguous chunks,
    // one per XCD (workgroup id mod 8 is the XCD the dispatcher places it on): perfectly balanced, and the
    // workgroups that share a weight tile run on the same XCD at the same time.  (A wrong placement guess costs
    // L2 hits, never correctness.)
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (item >= item_end || item >= (int64_t)p.G * p.Mtiles) return;
    pconv_item<BM, LRT, ILV, SEQ ? kSeq : kPlain>(p, item);
}

// The layer with its MaxPool2d(2, 2) (pconv_body.cuh, POOL): items are pooled pixels.
template <bool ILV>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void pconv_gemm_pool_kernel(const PConvArgs p) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (item >= item_end || item >= (int64_t)p.G * p.Mtiles) return;
    pconv_item<128, false, ILV, kPool>(p, item);
}

// The pooled LRT layer: 64-image tiles, dual accumulators, the sampling epilogue per window pixel.
__global__ __launch_bounds__(kThreads) void pconv_gemm_pool_lrt_kernel(const PConvArgs p) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (item >= item_end || item >= (int64_t)p.G * p.Mtiles) return;
    pconv_item<64, true, false, kPool>(p, item);
}

// Split contraction (pconv_body.cuh, SPLIT): block -> (item, k range).  The ksplit blocks of an item are consecutive, so they
// land on the same XCD chunk as their item's weight-tile sharers.
template <int BM, bool LRT, bool ILV>
__global__ __launch_bounds__(kThreads) void pconv_gemm_splitk_kernel(const PConvArgs p) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t blk = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t blk_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (blk >= blk_end || blk >= (int64_t)p.G * p.Mtiles * p.ksplit) return;
    const int64_t item = blk / p.ksplit;
    pconv_item<BM, LRT, ILV, kSplit>(p, item, (int)(blk - item * p.ksplit));
}

// maxpool over [planes][H][W][B] (planes = draws * channels), B innermost; 4 images per thread.
__global__ __launch_bounds__(256) void maxpool_chwn_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t total4,
                                                           int H, int W, int Ho, int Wo, int B4, int k, int s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int b4 = (int)(i % B4);
    int64_t t = i / B4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int64_t pl = t / Ho;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x) + ((pl * H + (int64_t)oh * s) * W + (int64_t)ow * s) * B4 + b4;
    f32x4 m = xp[0];
    for (int a = 0; a < k; ++a)
        for (int c = 0; c < k; ++c) {
            const f32x4 v = xp[((int64_t)a * W + c) * B4];
#pragma unroll
            for (int u = 0; u < 4; ++u) m[u] = fmaxf(m[u], v[u]);
        }
    reinterpret_cast<f32x4*>(y)[i] = m;
}

// Backward of [activation -> MaxPool2d(k, s)] (or of the activation alone, k = 0) in the batch-innermost layout -- training
// extension.  Gather form, one thread per (plane, h, w, 4 images) of the layer's activated output y:
//   g_act(h, w) = sum over the pooling windows that contain (h, w) of g_out(window) if (h, w) is that window's FIRST maximum
//                 in scan order (torch's max_pool2d backward routes the gradient to the first argmax), else 0;
//   g_pre = g_act * act'(pre-activation), written from y alone: Softplus' = sigmoid(v) = 1 - exp(-y), ReLU' = [y > 0].
// Overlapping windows (k > s: Bayesian3Conv3FC pools 3x3 / 2) make up to ceil(k/s)^2 windows per element; nothing is
// scattered, so no atomics and a deterministic result.
// LRT layers (am != NULL): the layer's output was act(act_mu + sqrt(act_var) * eps), so besides g_pre (= the gradient w.r.t.
// act_mu) the same pass writes the gradient w.r.t. act_var, g_pre * (v - act_mu) / (2 act_var), with the pre-activation v
// recovered from the stored activated output (softplus: y + log(1 - exp(-y)); nothing flows where ReLU clipped).  act_mu /
// act_var hold mom_planes planes: all of them, or one draw's worth when every draw shares one pair of moments (first layer).
__global__ __launch_bounds__(256) void pool_act_bwd_chwn_kernel(const float* __restrict__ g_out, const float* __restrict__ y,
                                                                float* __restrict__ g_pre, int64_t total4, int H, int W, int Hp,
                                                                int Wp, int B4, int k, int s, int act, int64_t out_pitch4,
                                                                const float* __restrict__ am, const float* __restrict__ av,
                                                                float* __restrict__ g_var, int64_t mom_planes,
                                                                const float* __restrict__ g2, const float* __restrict__ xc,
                                                                int64_t xc_total4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    // g2 != NULL (an LRT layer below another one): the incoming gradient is g_out + 2 * xc * g2 -- the two input gradients of the
    // layer above combined on the fly (bbb_lrt_glue mode 1's fmaf, bit for bit) instead of by a launch of its own
    auto incoming = [&](int64_t idx) {
        f32x4 gv = reinterpret_cast<const f32x4*>(g_out)[idx];
        if (g2 != nullptr) {
            const f32x4 b4v = reinterpret_cast<const f32x4*>(g2)[idx], x4v = reinterpret_cast<const f32x4*>(xc)[idx % xc_total4];
#pragma unroll
            for (int u = 0; u < 4; ++u) gv[u] = fmaf(2.0f * x4v[u], b4v[u], gv[u]);
        }
        return gv;
    };
    const int b4 = (int)(i % B4);
    int64_t t = i / B4;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int64_t pl = t / H;
    const f32x4* yp = reinterpret_cast<const f32x4*>(y) + pl * H * W * B4 + b4;
    const f32x4 me = yp[((int64_t)h * W + w) * B4];
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k == 0) {
        g = incoming(i);
    } else {
        const int64_t gbase = pl * Hp * Wp * B4 + b4;
        // windows (ph, pw) with ph*s <= h < ph*s + k
        const int ph_lo = h - k + 1 > 0 ? (h - k + 1 + s - 1) / s : 0, ph_hi = h / s < Hp - 1 ? h / s : Hp - 1;
        const int pw_lo = w - k + 1 > 0 ? (w - k + 1 + s - 1) / s : 0, pw_hi = w / s < Wp - 1 ? w / s : Wp - 1;
        for (int ph = ph_lo; ph <= ph_hi; ++ph)
            for (int pw = pw_lo; pw <= pw_hi; ++pw) {
                // am I the first maximum of this window?  (an earlier element >= me, or a later one > me, disqualifies)
                bool first[4] = {true, true, true, true};
                const int my = (h - ph * s) * k + (w - pw * s);
                for (int a = 0; a < k; ++a)
                    for (int c = 0; c < k; ++c) {
                        const int idx = a * k + c;
                        if (idx == my) continue;
                        const f32x4 v = yp[((int64_t)(ph * s + a) * W + (pw * s + c)) * B4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) first[u] = first[u] && (idx < my ? v[u] < me[u] : v[u] <= me[u]);
                    }
                const f32x4 go = incoming(gbase + ((int64_t)ph * Wp + pw) * B4);
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] += first[u] ? go[u] : 0.0f;
            }
    }
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float d = 1.0f;
        if (act == 1) d = me[u] > 0.0f ? 1.0f : 0.0f;
        else if (act == 2) d = me[u] > 20.0f ? 1.0f : -expm1f(-me[u]);      // y = softplus(v) -> sigmoid(v) = 1 - exp(-y)
        o[u] = g[u] * d;
    }
    // out_pitch4 != 0: planes are written at that pitch (in 16-byte units) instead of densely -- see bbb_pool_act_bwd_chwn
    const int64_t oi = out_pitch4 ? pl * out_pitch4 + (i - pl * (int64_t)H * W * B4) : i;
    reinterpret_cast<f32x4*>(g_pre)[oi] = o;
    if (am != nullptr) {
        const int64_t in_plane = i - pl * (int64_t)H * W * B4;
        const int64_t mi = (pl % mom_planes) * ((int64_t)H * W * B4) + in_plane;
        const f32x4 a4 = reinterpret_cast<const f32x4*>(am)[mi];
        const f32x4 v4 = reinterpret_cast<const f32x4*>(av)[mi];
        f32x4 gv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = me[u];
            if (act == 2 && !(v > 20.0f)) v = v + logf(-expm1f(-v));
            const float t = (act != 0 && !(me[u] > 0.0f)) ? 0.0f : v - a4[u];        // sqrt(act_var) * eps
            gv[u] = (o[u] * t) / (2.0f * v4[u]);
        }
        reinterpret_cast<f32x4*>(g_var)[oi] = gv;
    }
}

int fill(const bbb_conv_desc_t* d, PConvArgs& a) {
    if (d == nullptr) return BBB_EINVAL;
    if (d->batch <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0 ||
        d->draws <= 0 || d->act < 0 || d->act > 2)
        return BBB_EINVAL;
    if (d->batch % 4 != 0) return BBB_ESHAPE;        // batch-innermost rows are moved as 16-byte vectors
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    if ((int64_t)d->cin * d->h * d->w > 0x7fffffffLL || (int64_t)d->cin * d->kh * d->kw > 0x7fffffffLL) return BBB_ESHAPE;
    // per-draw slabs are addressed through 32-bit buffer offsets
    if ((int64_t)d->cin * d->h * d->w * d->batch * 4 > 0xFFFE0000LL || (int64_t)d->cout * ho * wo * d->batch * 4 > 0xFFFE0000LL ||
        ((int64_t)d->cout + 64) * d->cin * d->kh * d->kw * 4 > 0x3FFFFFFFLL || (int64_t)d->batch * 4 > 0x0FFFFFFFLL)
        return BBB_ESHAPE;
    a.x_inv = (0xFFFFFFF0u - ((uint32_t)d->batch + 512u) * 4u) & ~15u;   /* a ragged last tile reaches < 512 columns past the row */
    if ((int64_t)d->cin * d->h * d->w * d->batch * 4 > (int64_t)a.x_inv) return BBB_ESHAPE;
    a.B = d->batch; a.Cin = d->cin; a.H = d->h; a.W = d->w; a.Cout = d->cout; a.kh = d->kh; a.kw = d->kw;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = ho; a.Wo = wo; a.K = d->cin * d->kh * d->kw; a.khkw = d->kh * d->kw; a.act = d->act;
    if (d->w_row_pitch < 0 || (d->w_row_pitch > 0 && d->w_row_pitch < a.K)) return BBB_EINVAL;
    a.Kp = d->w_row_pitch > 0 ? d->w_row_pitch : a.K;
    if (((int64_t)d->cout + 64) * a.Kp * 4 > 0x3FFFFFFFLL) return BBB_ESHAPE;
    a.x_ds = d->x_draw_stride; a.w_ds = d->w_draw_stride; a.b_ds = d->b_draw_stride;
    a.y_ds = (int64_t)d->cout * ho * wo * d->batch;
    if (d->unit_div < 0 || d->unit_off < 0 || d->x_unit_mod < 0 || d->b_offset < 0) return BBB_EINVAL;
    if (d->unit_div > 1 && d->unit_off >= d->unit_div) return BBB_EINVAL;          // passed reduced modulo S
    if (d->x_unit_mod > 0 && d->x_unit_mod != d->unit_div) return BBB_EINVAL;
    if (d->x_unit_div < 0 || d->x_unit_off < 0 || (d->x_unit_div > 1 && (d->unit_div > 1 || d->x_unit_off >= d->x_unit_div)) ||
        (d->x_unit_div <= 1 && d->x_unit_off != 0))
        return BBB_EINVAL;
    a.unit_div = d->unit_div; a.unit_off = d->unit_div > 1 ? d->unit_off : 0; a.x_mod = d->x_unit_mod; a.b_off = d->b_offset;
    a.x_div = d->x_unit_div; a.x_off = d->x_unit_off;
    if (d->pool != 0 && d->pool != 1) return BBB_EINVAL;
    a.pool = d->pool;
    if (d->w_tap_major != 0 && d->w_tap_major != 1) return BBB_EINVAL;
    a.wtap = d->w_tap_major;
    if (a.pool) {
        if ((ho & 1) || (wo & 1)) return BBB_EINVAL;
        a.y_ds = (int64_t)d->cout * (ho / 2) * (wo / 2) * d->batch;
    }
    return 0;
}

// launches of more 64-image items than this run the in-workgroup form of a layer's split (measured, profiles/r03_notes.md
// section 2 and r04_notes.md: above ~400 items the cross-workgroup form only adds partial-tile traffic; LRT items carry two
// accumulator sets and their in-workgroup form runs at 3 waves per SIMD, so the crossover sits higher)
#ifndef PCONV_ILV_MAX
#define PCONV_ILV_MAX 12000          // launches of at most this many items interleave their staging loads with the MFMAs (see launch())
#endif
#ifndef PCONV_SPLIT_MAX
#define PCONV_SPLIT_MAX 384
#endif
constexpr int64_t split_max_items(bool lrt) { return lrt ? 512 : PCONV_SPLIT_MAX; }

template <bool LRT>
int launch(PConvArgs& a, int draws, hipStream_t st) {
    a.Ntiles = (a.Cout + BN - 1) / BN;
    a.G = a.Ntiles * draws;
    const int64_t pixels = (int64_t)a.Ho * a.Wo;
    // tile choice: 128 images per workgroup (two accumulator chains per wave) unless that leaves fewer than 3
    // workgroups per CU, then 64.  A 256-image tile (64x64 per wave) exists but measured 5-10 % slower on every
    // AlexNet layer (3 instead of 4 workgroups per CU); the launcher never selects it.
    // LRT stages two weight tiles and keeps two accumulator sets: 64-wide only.
    if (a.pool) {
        // one item per POOLED pixel, 128-image tiles (LRT: 64) (the callers fuse large launches only)
        if (a.ksplit > 1 || a.part != nullptr) return BBB_EINVAL;
        if (LRT && (a.y_mu != nullptr || a.y_var != nullptr)) return BBB_EINVAL;    // the moments of unpooled pixels are not kept
        a.nbt = (a.B + (LRT ? 63 : 127)) / (LRT ? 64 : 128);
        const int64_t mtp = (pixels / 4) * a.nbt;
        const int64_t itp = (int64_t)a.G * mtp;
        if (mtp > 0x7fffffffLL || itp > 0x7fffffffLL - 8) return BBB_ESHAPE;
        a.Mtiles = (int)mtp;
        const int64_t perp = (itp + 7) / 8;
        a.per_xcd = (int32_t)perp;
        // staging loads up front (ILV = false): with the running maximum in 32 more accumulation registers this form fits four
        // workgroups per CU (60 + 64 registers) and the interleaved one does not (82 + 64: three; measured 492-494 us against 480-482
        // for conv1 of the metric step, profiles/r04_notes.md section 7)
        if constexpr (LRT) hipLaunchKernelGGL(pconv_gemm_pool_lrt_kernel, dim3((unsigned)(8 * perp)), dim3(kThreads), 0, st, a);
        else               hipLaunchKernelGGL((pconv_gemm_pool_kernel<false>), dim3((unsigned)(8 * perp)), dim3(kThreads), 0, st, a);
        return (int)hipGetLastError();
    }
    const int64_t nb128 = pixels * ((a.B + 127) / 128) * a.G;
    int bm = (LRT || nb128 < 768) ? 64 : 128;        // (round 3 re-measured 600 / 300: conv4 +10 %, conv5 +25 % slower with 128)
    // an image axis that 128-wide tiles would pad by >= 25 % and 64-wide ones by less (192 = the input channels of conv3's
    // role-swapped weight gradient: 256 against 192): 64.  Same sums per element either way.
    if (bm == 128 && (a.B + 127) / 128 * 128 * 4 >= a.B * 5 && (a.B + 63) / 64 * 64 < (a.B + 127) / 128 * 128) bm = 64;
    const int64_t items64 = pixels * ((a.B + 63) / 64) * a.G;
    const bool cross = a.ksplit > 1 && a.part != nullptr && items64 <= split_max_items(LRT);
    if (cross) bm = 64;
    a.nbt = (a.B + bm - 1) / bm;
    const int64_t mt = pixels * a.nbt;
    if (mt > 0x7fffffffLL) return BBB_ESHAPE;
    a.Mtiles = (int)mt;
    const int64_t items = (int64_t)a.G * mt;
    if (cross) {
        // the layer's split contraction ACROSS workgroups: a launch this small cannot fill the chip otherwise
        if (items * a.ksplit > 0x7fffffffLL) return BBB_EINVAL;
        const int64_t perb = (items * a.ksplit + 7) / 8;
        a.per_xcd = (int32_t)perb;
        hipLaunchKernelGGL((pconv_gemm_splitk_kernel<64, LRT, true>), dim3((unsigned)(8 * perb)), dim3(kThreads), 0, st, a);
        return (int)hipGetLastError();
    }
    const int64_t per = (items + 7) / 8;
    const int64_t blocks = 8 * per;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    a.per_xcd = (int32_t)per;
    // staging loads interleaved with the MFMAs (ILV): round 1 measured it a loss beyond ~1.5 rounds of workgroups; re-measured in
    // round 3 on the current kernel (profiles/r03_notes.md section 3) it is a 1-2 % gain up to ~12k items, one or three steps in flight
    const bool ilv = items <= PCONV_ILV_MAX;
    const dim3 grid((unsigned)blocks), block(kThreads);
    if (a.ksplit > 1) {
        // the same summation order inside ONE workgroup per item (pconv_body.cuh, SEQ): no scratch, no extra traffic
        if constexpr (!LRT) {
            if (bm == 128) {
                // (88-100 VGPRs + 64 accumulation registers = three workgroups per CU; held to four by amdgpu_waves_per_eu the
                // prefetched tile spills around the range folds and the launch is 4-9 % slower: profiles/r04_notes.md section 4)
                if (ilv) hipLaunchKernelGGL((pconv_gemm_kernel<128, false, true, true>), grid, block, 0, st, a);
                else     hipLaunchKernelGGL((pconv_gemm_kernel<128, false, false, true>), grid, block, 0, st, a);
                return (int)hipGetLastError();
            }
        }
        const bool ilv_seq = ilv && !LRT;        // LRT: the interleaved form needs 172 registers (2 waves per SIMD), the plain one 161 (3)
        if (ilv_seq) hipLaunchKernelGGL((pconv_gemm_kernel<64, LRT, true, true>), grid, block, 0, st, a);
        else         hipLaunchKernelGGL((pconv_gemm_kernel<64, LRT, false, true>), grid, block, 0, st, a);
        return (int)hipGetLastError();
    }
    if constexpr (!LRT) {
        if (bm == 128) {
            if (ilv) hipLaunchKernelGGL((pconv_gemm_kernel<128, false, true>), grid, block, 0, st, a);
            else     hipLaunchKernelGGL((pconv_gemm_kernel<128, false, false>), grid, block, 0, st, a);
            return (int)hipGetLastError();
        }
    }
    if (ilv) hipLaunchKernelGGL((pconv_gemm_kernel<64, LRT, true>), grid, block, 0, st, a);
    else     hipLaunchKernelGGL((pconv_gemm_kernel<64, LRT, false>), grid, block, 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace

namespace {
constexpr int64_t kTicketBytes = 16384;         // arrival counters of up to 4096 items

// The split of a LAYER's contraction: a function of the layer's geometry ONLY (not of the batch, the number of draws or how a
// step is partitioned), so that every launch that computes an output element of this layer -- one draw alone, a 10-draw
// launch, a work unit of a sharded step, one of G steps per launch, a batch-parallel shard -- adds the same partial sums in the
// same order (cross-workgroup SPLIT form for small launches, in-workgroup SEQ form otherwise: same bits).
// Which layers: few (pixel, 64-channel tile) groups and a long contraction -- a draw of such a layer is a handful of workgroups
// each walking a long serial k loop (AlexNet conv4: 4 pixels x 4 tiles, 48 k tiles; conv5: 4 x 2, 32 tiles; measured one draw,
// bs 512: 42 -> 28 us and 28 -> 17 us with four ranges, profiles/r03_notes.md section 2).  Layers with more groups (conv2: 48, conv3:
// 24) fill the chip from a few hundred images on and keep the plain chain.
int canonical_ksplit(const PConvArgs& a) {
    const int ntl = (a.Cout + BN - 1) / BN;
    const int64_t groups = (int64_t)a.Ho * a.Wo * ntl;
    if (groups > 16) return 1;
    // longest contraction of any pixel, in 32-k tiles (taps that can fall inside the image)
    const int nr = a.kh < (a.H - 1) / a.dh + 1 ? a.kh : (a.H - 1) / a.dh + 1;
    const int nq = a.kw < (a.W - 1) / a.dw + 1 ? a.kw : (a.W - 1) / a.dw + 1;
    const int tiles = (a.Cin * nr * nq + BK - 1) / BK;
    if (tiles < 16) return 1;
    const int s = tiles / 8;                                             // >= 8 tiles per range
    return s > 4 ? 4 : s;
}

// Scratch bytes the cross-workgroup form of this launch needs (0: the launch is too large for it and runs the SEQ form).
int64_t split_scratch_bytes(const PConvArgs& a, int draws, bool lrt, int s) {
    if (s <= 1) return 0;
    const int ntl = (a.Cout + BN - 1) / BN;
    const int64_t items = (int64_t)a.Ho * a.Wo * ((a.B + 63) / 64) * ntl * draws;
    if (items > split_max_items(lrt)) return 0;
    // tickets live in a FIXED region at the start of the scratch (split launches have < 512 items), so that launches of
    // different sizes sharing one scratch buffer never put partial tiles where another launch expects zeroed tickets
    return kTicketBytes + items * s * (lrt ? 2 : 1) * 64 * 64 * 4;
}

int split_setup(PConvArgs& a, int draws, bool lrt, int k_split, void* scratch, int64_t scratch_bytes) {
    if (k_split <= 1) return 0;
    if (k_split != canonical_ksplit(a)) return BBB_EINVAL;               // the split is the layer's, not the caller's choice
    a.ksplit = k_split;
    const int64_t need = split_scratch_bytes(a, draws, lrt, k_split);
    if (need > 0 && scratch != nullptr) {
        if (scratch_bytes < need || (((uintptr_t)scratch) & 255u) != 0) return BBB_EINVAL;
        a.tickets = static_cast<int32_t*>(scratch);
        a.part = reinterpret_cast<float*>(static_cast<char*>(scratch) + kTicketBytes);
    }
    return 0;
}
}  // namespace

extern "C" int64_t bbb_conv2d_chwn_splitk_scratch(const bbb_conv_desc_t* d, int lrt, int32_t* k_split) {
    PConvArgs a = {};
    if (k_split) *k_split = 1;
    if (fill(d, a) != 0) return 0;
    const int s = canonical_ksplit(a);
    if (k_split) *k_split = s;
    return split_scratch_bytes(a, d->draws, lrt != 0, s);
}

extern "C" int bbb_conv2d_chwn_splitk_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias, float* y,
                                          int k_split, void* scratch, int64_t scratch_bytes, void* stream) {
    PConvArgs a = {};
    int rc = fill(d, a);
    if (rc != 0) return rc;
    if (x == nullptr || w == nullptr || y == nullptr) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0 || (((uintptr_t)w | (uintptr_t)bias) & 3u) != 0) return BBB_EALIGN;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    if ((rc = split_setup(a, d->draws, false, k_split, scratch, scratch_bytes)) != 0) return rc;
    return launch<false>(a, d->draws, (hipStream_t)stream);
}

extern "C" int bbb_conv2d_chwn_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias, float* y,
                                   void* stream) {
    PConvArgs a = {};
    const int rc = fill(d, a);
    if (rc != 0) return rc;
    if (x == nullptr || w == nullptr || y == nullptr) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0 || (((uintptr_t)w | (uintptr_t)bias) & 3u) != 0) return BBB_EALIGN;
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    return launch<false>(a, d->draws, (hipStream_t)stream);
}

// bbb_conv2d_chwn_fwd with the contraction on the 16-bit matrix pipe at fp32 accuracy, range-free (pconv_bf16x3.cuh)
extern "C" int bbb_conv2d_chwn_bf16x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w, const float* bias, void* y,
                                          uint32_t flags, void* stream) {
    PConvArgs a = {};
    const int rc = fill(d, a);
    if (rc != 0) return rc;
    if (x == nullptr || w == nullptr || y == nullptr || (flags & ~3u) != 0 || a.pool || a.wtap) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0 || (((uintptr_t)w | (uintptr_t)bias) & 3u) != 0) return BBB_EALIGN;
    const bool xs3 = (flags & BBB_S3_IN) != 0, ys3 = (flags & BBB_S3_OUT) != 0;
    if ((xs3 || ys3) && d->batch % 8 != 0) return BBB_ESHAPE;           // S3 rows move as 16-byte vectors of 8 bf16
    a.x = static_cast<const float*>(x); a.w = w; a.bias = bias; a.y = static_cast<float*>(y);
    a.x_ps = (int64_t)a.Cin * a.H * a.W * a.B;
    a.y_ps = (int64_t)a.Cout * a.Ho * a.Wo * a.B;
    if (ys3) a.y_ds = 3 * a.y_ps;
    if (xs3 && (d->x_draw_stride != 0 && d->x_draw_stride < 3 * a.x_ps)) return BBB_EINVAL;
    a.Ntiles = (a.Cout + BN - 1) / BN;
    a.G = a.Ntiles * d->draws;
    const int64_t pixels = (int64_t)a.Ho * a.Wo;
    // 128 images per workgroup (measured: the 256-image form -- 2 workgroups per CU, half the workgroups -- is slower on four of
    // the six launches of the metric step, profiles/r04_notes.md)
    constexpr int kMT = 1;
    a.nbt = (a.B + 128 * kMT - 1) / (128 * kMT);
    const int64_t mtiles = pixels * a.nbt;
    if (mtiles > 0x7fffffffLL) return BBB_ESHAPE;
    a.Mtiles = (int)mtiles;
    const int64_t per = ((int64_t)a.G * mtiles + 7) / 8;
    if (8 * per > 0x7fffffffLL) return BBB_ESHAPE;
    a.per_xcd = (int32_t)per;
    const dim3 grid((unsigned)(8 * per)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    if (xs3 && ys3)      hipLaunchKernelGGL((pconv_bf16x3_kernel<kMT, true, true>), grid, block, 0, st, a);
    else if (xs3)        hipLaunchKernelGGL((pconv_bf16x3_kernel<kMT, true, false>), grid, block, 0, st, a);
    else if (ys3)        hipLaunchKernelGGL((pconv_bf16x3_kernel<kMT, false, true>), grid, block, 0, st, a);
    else                 hipLaunchKernelGGL((pconv_bf16x3_kernel<kMT, false, false>), grid, block, 0, st, a);
    return (int)hipGetLastError();
}

namespace {
// S3 helpers: 8 images (16 bytes per plane) per thread
typedef uint32_t s3_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s3_join(const s3_u32x4 h, const s3_u32x4 m, const s3_u32x4 l, float (&v)[8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {        // hi + mid is exact (16 significant bits), + lo gives back the fp32 value exactly
        v[2 * c] = (__builtin_bit_cast(float, h[c] << 16) + __builtin_bit_cast(float, m[c] << 16)) + __builtin_bit_cast(float, l[c] << 16);
        v[2 * c + 1] = (__builtin_bit_cast(float, h[c] & 0xFFFF0000u) + __builtin_bit_cast(float, m[c] & 0xFFFF0000u)) +
                       __builtin_bit_cast(float, l[c] & 0xFFFF0000u);
    }
}

__device__ __forceinline__ void s3_cut(const float (&v)[8], s3_u32x4& h, s3_u32x4& m, s3_u32x4& l) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t a0, a1, a2;
        split3_pair(f32x2{v[2 * c], v[2 * c + 1]}, a0, a1, a2);
        h[c] = a0; m[c] = a1; l[c] = a2;
    }
}

// MaxPool2d(k, s) on S3 planes: x [slabs][3][C][H][W][B] -> y [slabs][3][C][Ho][Wo][B]; values are joined (exact), compared as
// fp32 and cut again (exact), so the result is the S3 form of what maxpool_chwn_kernel computes on the fp32 tensor.
// SQ (the LRT chain of pconv_c8x3): slabs hold SIX planes -- the values' pieces, then the pieces of their squares; the maximum is
// taken over the values and the squares are those of the pooled values (what the next LRT layer's second contraction reads).
template <bool SQ>
__global__ __launch_bounds__(256) void maxpool_s3_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                         int64_t total8, int C, int H, int W, int Ho, int Wo, int B8, int k, int s) {
    constexpr int NPL = SQ ? 6 : 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int b8 = (int)(i % B8);
    int64_t t = i / B8;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C);
    const int64_t slab = t / C;
    const int64_t xps = (int64_t)C * H * W * B8, yps = (int64_t)C * Ho * Wo * B8;     // plane strides in 16-byte units
    const s3_u32x4* xp = reinterpret_cast<const s3_u32x4*>(x) + slab * NPL * xps + (((int64_t)c * H + (int64_t)oh * s) * W + (int64_t)ow * s) * B8 + b8;
    float best[8];
    for (int a = 0; a < k; ++a)
        for (int q = 0; q < k; ++q) {
            const int64_t o = ((int64_t)a * W + q) * B8;
            float v[8];
            s3_join(xp[o], xp[o + xps], xp[o + 2 * xps], v);
#pragma unroll
            for (int u = 0; u < 8; ++u) best[u] = (a == 0 && q == 0) ? v[u] : fmaxf(best[u], v[u]);
        }
    s3_u32x4 h, m, l;
    s3_cut(best, h, m, l);
    s3_u32x4* yp = reinterpret_cast<s3_u32x4*>(y) + slab * NPL * yps + (((int64_t)c * Ho + oh) * Wo + ow) * B8 + b8;
    yp[0] = h; yp[yps] = m; yp[2 * yps] = l;
    if constexpr (SQ) {
#pragma unroll
        for (int u = 0; u < 8; ++u) best[u] = bbb::mul_rn(best[u], best[u]);      // (the rounded fp32 square: no fma contraction into the cut)
        s3_cut(best, h, m, l);
        yp[3 * yps] = h; yp[4 * yps] = m; yp[5 * yps] = l;
    }
}

// fp32 [slabs][n] <-> S3 [slabs][3][n] (n % 8 == 0): test / boundary helpers
__global__ __launch_bounds__(256) void s3_from_f32_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n8, int64_t slabs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8 * slabs) return;
    const int64_t slab = i / n8, j = i - slab * n8;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x) + (slab * n8 + j) * 2;
    const f32x4 a = xp[0], b = xp[1];
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    s3_u32x4 h, m, l;
    s3_cut(v, h, m, l);
    s3_u32x4* yp = reinterpret_cast<s3_u32x4*>(y) + slab * 3 * n8 + j;
    yp[0] = h; yp[n8] = m; yp[2 * n8] = l;
}

__global__ __launch_bounds__(256) void s3_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int64_t n8, int64_t slabs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8 * slabs) return;
    const int64_t slab = i / n8, j = i - slab * n8;
    const s3_u32x4* xp = reinterpret_cast<const s3_u32x4*>(x) + slab * 3 * n8 + j;
    float v[8];
    s3_join(xp[0], xp[n8], xp[2 * n8], v);
    f32x4* yp = reinterpret_cast<f32x4*>(y) + (slab * n8 + j) * 2;
    yp[0] = f32x4{v[0], v[1], v[2], v[3]};
    yp[1] = f32x4{v[4], v[5], v[6], v[7]};
}
}  // namespace

namespace {
int maxpool_s3_launch(const void* x, void* y, int64_t slabs, int channels, int h, int w, int batch, int k, int s, bool sq, void* stream) {
    if (x == nullptr || y == nullptr || slabs <= 0 || channels <= 0 || h <= 0 || w <= 0 || batch <= 0 || k <= 0 || s <= 0) return BBB_EINVAL;
    if (batch % 8 != 0 || h < k || w < k) return BBB_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) return BBB_EALIGN;
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const int64_t total8 = slabs * channels * ho * wo * (batch / 8);
    const int64_t blocks = (total8 + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    if (sq) hipLaunchKernelGGL(maxpool_s3_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const unsigned short*>(x),
                               static_cast<unsigned short*>(y), total8, channels, h, w, ho, wo, batch / 8, k, s);
    else    hipLaunchKernelGGL(maxpool_s3_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const unsigned short*>(x),
                               static_cast<unsigned short*>(y), total8, channels, h, w, ho, wo, batch / 8, k, s);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bbb_maxpool_chwn_s3(const void* x, void* y, int64_t slabs, int channels, int h, int w, int batch, int k, int s, void* stream) {
    return maxpool_s3_launch(x, y, slabs, channels, h, w, batch, k, s, false, stream);
}

extern "C" int bbb_maxpool_chwn_s3sq(const void* x, void* y, int64_t slabs, int channels, int h, int w, int batch, int k, int s, void* stream) {
    return maxpool_s3_launch(x, y, slabs, channels, h, w, batch, k, s, true, stream);
}

extern "C" int bbb_s3_convert(const void* src, void* dst, int64_t slabs, int64_t n, int to_s3, void* stream) {
    if (src == nullptr || dst == nullptr || slabs <= 0 || n <= 0) return BBB_EINVAL;
    if (n % 8 != 0) return BBB_ESHAPE;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) != 0) return BBB_EALIGN;
    const int64_t blocks = (slabs * (n / 8) + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    if (to_s3) hipLaunchKernelGGL(s3_from_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(src),
                                  static_cast<unsigned short*>(dst), n / 8, slabs);
    else       hipLaunchKernelGGL(s3_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                  static_cast<const unsigned short*>(src), static_cast<float*>(dst), n / 8, slabs);
    return (int)hipGetLastError();
}

extern "C" int bbb_lrt_conv2d_chwn_splitk_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                                              const float* b_mu, const float* b_var, float* y, float* act_mu_out,
                                              float* act_var_out, const float* eps_ext, uint64_t seed, uint32_t call0,
                                              uint32_t stream_id, int sample, const uint32_t* call_dev, int k_split,
                                              void* scratch, int64_t scratch_bytes, void* stream) {
    PConvArgs a = {};
    int rc = fill(d, a);
    if (rc != 0) return rc;
    if (k_split > 1 && (rc = split_setup(a, d->draws, true, k_split, scratch, scratch_bytes)) != 0) return rc;
    if (x == nullptr || w_mu == nullptr || w_var == nullptr || y == nullptr) return BBB_EINVAL;
    if ((b_mu == nullptr) != (b_var == nullptr)) return BBB_EINVAL;
    if (d->w_draw_stride != 0 || d->b_draw_stride != 0) return BBB_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) return BBB_EALIGN;
    if ((((uintptr_t)w_mu | (uintptr_t)w_var | (uintptr_t)b_mu | (uintptr_t)b_var | (uintptr_t)act_mu_out |
          (uintptr_t)act_var_out | (uintptr_t)eps_ext) & 3u) != 0)
        return BBB_EALIGN;
    a.x = x; a.w = w_mu; a.w2 = w_var; a.bias = b_mu; a.bias2 = b_var; a.y = y;
    a.y_mu = act_mu_out; a.y_var = act_var_out; a.eps_ext = eps_ext;
    a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.call0 = call0; a.stream_id = stream_id;
    a.sample = sample ? 1 : 0;
    a.call_dev = call_dev;
    return launch<true>(a, d->draws, (hipStream_t)stream);
}

extern "C" int bbb_lrt_conv2d_chwn_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                                       const float* b_mu, const float* b_var, float* y, float* act_mu_out,
                                       float* act_var_out, const float* eps_ext, uint64_t seed, uint32_t call0,
                                       uint32_t stream_id, int sample, const uint32_t* call_dev, void* stream) {
    return bbb_lrt_conv2d_chwn_splitk_fwd(d, x, w_mu, w_var, b_mu, b_var, y, act_mu_out, act_var_out, eps_ext, seed, call0, stream_id,
                                          sample, call_dev, 1, nullptr, 0, stream);
}

extern "C" int bbb_maxpool_chwn(const float* x, float* y, int64_t planes, int h, int w, int batch, int k, int s, void* stream) {
    if (x == nullptr || y == nullptr || planes <= 0 || h <= 0 || w <= 0 || batch <= 0 || k <= 0 || s <= 0) return BBB_EINVAL;
    if (batch % 4 != 0 || h < k || w < k) return BBB_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) return BBB_EALIGN;
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const int64_t total4 = planes * ho * wo * (batch / 4);
    const int64_t blocks = (total4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(maxpool_chwn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, total4, h, w, ho, wo,
                       batch / 4, k, s);
    return (int)hipGetLastError();
}

namespace {
int pool_act_bwd_launch(const float* g_out, const float* y, float* g_pre, int64_t planes, int h, int w, int batch, int k, int s, int act,
                        int64_t out_plane_pitch, const float* am, const float* av, float* g_var, int64_t mom_planes, void* stream,
                        const float* g2 = nullptr, const float* xc = nullptr, int64_t xc_planes = 0) {
    if (g_out == nullptr || y == nullptr || g_pre == nullptr || planes <= 0 || h <= 0 || w <= 0 || batch <= 0 || k < 0 ||
        (k > 0 && s <= 0) || act < 0 || act > 2)
        return BBB_EINVAL;
    if (batch % 4 != 0 || (k > 0 && (h < k || w < k))) return BBB_ESHAPE;
    if ((((uintptr_t)g_out | (uintptr_t)y | (uintptr_t)g_pre) & 15u) != 0) return BBB_EALIGN;
    if (out_plane_pitch != 0 && (out_plane_pitch < (int64_t)h * w * batch || out_plane_pitch % 4 != 0)) return BBB_EINVAL;
    const int hp = k > 0 ? (h - k) / s + 1 : h, wp = k > 0 ? (w - k) / s + 1 : w;
    const int64_t total4 = planes * h * w * (batch / 4);
    const int64_t blocks = (total4 + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    if ((g2 == nullptr) != (xc == nullptr)) return BBB_EINVAL;
    if (g2 != nullptr) {
        if (xc_planes <= 0 || planes % xc_planes != 0) return BBB_EINVAL;
        if ((((uintptr_t)g2 | (uintptr_t)xc) & 15u) != 0) return BBB_EALIGN;
    }
    hipLaunchKernelGGL(pool_act_bwd_chwn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g_out, y, g_pre, total4, h, w,
                       hp, wp, batch / 4, k, s, act, out_plane_pitch / 4, am, av, g_var, mom_planes, g2, xc,
                       g2 != nullptr ? xc_planes * hp * wp * (batch / 4) : (int64_t)1);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bbb_pool_act_bwd_chwn(const float* g_out, const float* y, float* g_pre, int64_t planes, int h, int w, int batch,
                                     int k, int s, int act, int64_t out_plane_pitch, void* stream) {
    return pool_act_bwd_launch(g_out, y, g_pre, planes, h, w, batch, k, s, act, out_plane_pitch, nullptr, nullptr, nullptr, 1, stream);
}

extern "C" int bbb_lrt_pool_act_bwd_chwn(const float* g_out, const float* y, const float* act_mu, const float* act_var, float* g_mu,
                                         float* g_var, int64_t planes, int64_t moment_planes, int h, int w, int batch, int k, int s,
                                         int act, int64_t out_plane_pitch, const float* g_out2, const float* x_out, int64_t x_planes,
                                         void* stream) {
    if (act_mu == nullptr || act_var == nullptr || g_var == nullptr) return BBB_EINVAL;
    if (moment_planes <= 0 || planes % moment_planes != 0) return BBB_EINVAL;
    if ((((uintptr_t)act_mu | (uintptr_t)act_var | (uintptr_t)g_var) & 15u) != 0) return BBB_EALIGN;
    return pool_act_bwd_launch(g_out, y, g_mu, planes, h, w, batch, k, s, act, out_plane_pitch, act_mu, act_var, g_var, moment_planes,
                               stream, g_out2, x_out, x_planes);
}
