// Pixel-major implicit GEMM, LDS-DMA pipelined variant (gfx950 `global_load_lds`).
//
// Same contraction, layouts and tiling as pconv_gemm.hip (one workgroup = one output pixel x 64 channels x BM images,
// only in-bounds kernel taps, batch-innermost activations), but the global->LDS staging no longer passes through
// VGPRs: every lane issues `global_load_lds_dword{,x4}` straight into a THREE-stage LDS ring and the loads of tile
// t+2 are in flight while tile t is multiplied.  s_memtime stamps of the register-staged kernel showed 2+ us between
// issuing a tile's loads and their arrival (first-touch lines come from Infinity Cache / HBM and all workgroups of a
// group miss on them together), i.e. one tile (~2k MFMA cycles) of prefetch distance left the matrix pipe idle
// ~35 % of the time; hipcc turns a two-deep *register* prefetch into loop-carried copies behind `s_waitcnt vmcnt(0)`.
// With LDS-DMA nothing is loop-carried in registers and the waits are counted by hand:
//
//     prologue: issue(0); issue(1)
//     tile t:   s_waitcnt vmcnt(OPS)      // tile t landed (tile t+1 may still be in flight)
//               s_barrier                 // ... for every wave; and every wave is done with tile t-1
//               issue(t+2)                // into the stage tile t-1 just vacated
//               32 x MFMA on tile t
//
// ONE barrier per tile.  LDS layout (single __shared__ array, no padding - the DMA destination is lane-linear):
//     X stage: [32 k][BM] floats;  W stage: [64 n][32 k] floats with the k column XOR-swizzled by (n & 31), applied on
//     the per-lane SOURCE address (the DMA writes linearly) and on the fragment read -> conflict-free ds_read_b32.
// Invalid k (last tile) reads a zero line for W and a valid row for X; channels >= Cout / images >= B are clamped to
// valid addresses (their D rows / columns are never stored).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"
#include "pconv_args.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

constexpr int kThreads = 256;
constexpr int BN = 64;
constexpr int BK = 32;
constexpr int KCH = 256;
constexpr int TPC = KCH / BK;
constexpr int NSTAGE = 3;

__device__ const float g_zero_line[64] = {0.0f};

typedef PConvArgs DmaArgs;

template <int BM, bool LRT>
constexpr int stage_floats() { return BK * BM + (LRT ? 2 : 1) * BN * BK; }
template <int BM, bool LRT>
constexpr int smem_bytes() { return (NSTAGE * stage_floats<BM, LRT>() + 4 * KCH) * 4 + 1024; }

template <int BM, bool LRT>
__global__ __launch_bounds__(kThreads) void pconv_dma_kernel(const DmaArgs p) {
    constexpr int NT = (BM == 128) ? 2 : 1;
    constexpr int WSETS = LRT ? 2 : 1;
    constexpr int XL = BM / 4;                 // lanes per X row (16 bytes each)
    constexpr int XRPP = kThreads / XL;        // X rows per pass (8 or 16)
    constexpr int XPASS = BK / XRPP;           // 4 or 2
    constexpr int SF = stage_floats<BM, LRT>();
    constexpr int OPS = 8 * WSETS + XPASS;     // DMA instructions per thread per tile

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int32_t* const kt_w = reinterpret_cast<int32_t*>(smem + NSTAGE * SF);      // [2][KCH] weight offset (elements) or -1
    int32_t* const kt_x = kt_w + 2 * KCH;                                       // [2][KCH] x row index or -1

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    if (item >= (int64_t)(xcd + 1) * p.per_xcd || item >= (int64_t)p.G * p.Mtiles) return;
    const int g = (int)(item / p.Mtiles);
    const int j = (int)(item - (int64_t)g * p.Mtiles);
    const int e = g / p.Ntiles;
    const int n0 = (g - e * p.Ntiles) * BN;
    const int pix = j / p.nbt;
    const int b0 = (j - pix * p.nbt) * BM;
    const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
    const int ihb = oh * p.sh - p.ph, iwb = ow * p.sw - p.pw;
    int r_lo = ihb < 0 ? (-ihb + p.dh - 1) / p.dh : 0;
    int q_lo = iwb < 0 ? (-iwb + p.dw - 1) / p.dw : 0;
    int r_hi = (p.H - 1 - ihb) >= 0 ? (p.H - 1 - ihb) / p.dh + 1 : 0;
    int q_hi = (p.W - 1 - iwb) >= 0 ? (p.W - 1 - iwb) / p.dw + 1 : 0;
    r_hi = r_hi < p.kh ? r_hi : p.kh;
    q_hi = q_hi < p.kw ? q_hi : p.kw;
    const int nr = r_hi > r_lo ? r_hi - r_lo : 0;
    const int nq = q_hi > q_lo ? q_hi - q_lo : 0;
    const int nrq = nr * nq;
    const int Keff = p.Cin * nrq;
    const int ntiles = (Keff + BK - 1) / BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (BM == 128) ? 0 : (wave >> 1) * 32;
    const int wm = (BM == 128) ? wave * 32 : (wave & 1) * 32;
    const int lrow = lane & 31, lk = lane >> 5;

    const float* __restrict__ xg = p.x + (int64_t)e * p.x_ds;
    const float* __restrict__ wg = p.w + (int64_t)e * p.w_ds;
    const float* __restrict__ w2g = LRT ? p.w2 + (int64_t)e * p.w_ds : nullptr;

    // ---- DMA lane roles ----
    // weights: per op a wave fills rows (2*wave + 8*ps) and (+1) of the [64][32] stage; lane -> physical column lane&31
    const int wcol = lane & 31;
    int wrow_n[8];            // clamped global channel of this lane's row in op ps
    int wrow_sw[8];           // its swizzle (local row & 31)
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int rl = 2 * wave + 8 * ps + (lane >> 5);          // local row 0..63
        const int n = n0 + rl;
        wrow_n[ps] = n < p.Cout ? n : p.Cout - 1;
        wrow_sw[ps] = rl & 31;
    }
    // x: XL lanes per row; clamp the image column so that every lane reads inside its row
    const int xr_l = lane / XL;                                  // row within the wave's group of rows
    int xb = b0 + (lane % XL) * 4;
    xb = xb <= p.B - 4 ? xb : p.B - 4;

    const float inv_nrq = nrq > 0 ? 1.0f / (float)nrq : 0.0f;
    const float inv_nq = nq > 0 ? 1.0f / (float)nq : 0.0f;
    auto fill_chunk = [&](int chunk) {
        const int k = chunk * KCH + tid;
        int wo = -1, xo = -1;
        if (k < Keff) {
            int ci = (int)((float)k * inv_nrq);
            int rq = k - ci * nrq;
            if (rq < 0) { --ci; rq += nrq; } else if (rq >= nrq) { ++ci; rq -= nrq; }
            int rr = (int)((float)rq * inv_nq);
            int qq = rq - rr * nq;
            if (qq < 0) { --rr; qq += nq; } else if (qq >= nq) { ++rr; qq -= nq; }
            const int r = r_lo + rr, q = q_lo + qq;
            wo = ci * p.khkw + r * p.kw + q;
            xo = (ci * p.H + ihb + r * p.dh) * p.W + iwb + q * p.dw;
        }
        kt_w[(chunk & 1) * KCH + tid] = wo;
        kt_x[(chunk & 1) * KCH + tid] = xo;
    };

    auto issue = [&](int tile) {
        float* const st = smem + (tile % NSTAGE) * SF;
        const int tb = ((tile / TPC) & 1) * KCH + (tile % TPC) * BK;
        // x rows first (16-byte DMA), then weights (4-byte DMA)
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) {
            const int rbase = wave * (64 / XL) + ps * XRPP;        // first row of this wave's op
            const int xo = kt_x[tb + rbase + xr_l];
            const float* src = xg + (int64_t)(xo >= 0 ? xo : 0) * p.B + xb;
            __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(st + rbase * BM), 16, 0, 0);
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int k = wcol ^ wrow_sw[ps];                     // logical k stored in this lane's physical column
            const int wo = kt_w[tb + k];
            const float* src = wo >= 0 ? wg + (int64_t)wrow_n[ps] * p.K + wo : g_zero_line + lane;
            __builtin_amdgcn_global_load_lds((gvoid_t*)src, (lvoid_t*)(st + BK * BM + (2 * wave + 8 * ps) * BK), 4, 0, 0);
            if (LRT) {
                const float* src2 = wo >= 0 ? w2g + (int64_t)wrow_n[ps] * p.K + wo : g_zero_line + lane;
                __builtin_amdgcn_global_load_lds((gvoid_t*)src2, (lvoid_t*)(st + BK * BM + BN * BK + (2 * wave + 8 * ps) * BK), 4, 0, 0);
            }
        }
    };

    f32x16 acc[NT];
    f32x16 accv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accv[t][r] = 0.0f; }

#ifdef BBB_TIMESTAMPS
    long long* tsb = reinterpret_cast<long long*>(kt_x + 2 * KCH);
    const bool tson = p.ts && (bid == 8 * 40 || bid == 8 * 100) && tid == 0;
    int tsi = 0;
#define TS() do { if (tson && tsi < 112) tsb[tsi++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TS() do { } while (0)
#endif
    TS();
    auto mma_tile = [&](int tile) {
        const float* const xs = smem + (tile % NSTAGE) * SF;
        const float* const ws = xs + BK * BM;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = kk * 2 + lk;
            const float b = xs[krow * BM + wm + lrow];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float a = ws[(wn + nt * 32 + lrow) * BK + (krow ^ lrow)];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
                if (LRT) {
                    const float a2 = ws[BN * BK + (wn + nt * 32 + lrow) * BK + (krow ^ lrow)];
                    accv[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b * b, accv[nt], 0, 0, 0);
                }
            }
        }
    };

    if (ntiles > 0) {
        fill_chunk(0);
        if (KCH < Keff) fill_chunk(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(0);
        if (ntiles > 1) issue(1);
        TS();
        for (int t = 0; t < ntiles; ++t) {
            // tile t has landed once at most one tile's worth of DMA (tile t+1) is still outstanding
            if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
            else                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // table writes / fragment reads of this wave retired
            __builtin_amdgcn_s_barrier();
            TS();
            if (t + 2 < ntiles) issue(t + 2);
            TS();
            // decode chunk c+1 early in chunk c (chunk 1 was decoded in the prologue); visible after the next barriers
            if ((t % TPC) == 1 && t / TPC >= 1 && (t / TPC + 1) * KCH < Keff) fill_chunk(t / TPC + 1);
            mma_tile(t);
            TS();
        }
    }
    TS();
#ifdef BBB_TIMESTAMPS
    if (tson) { long long* o = p.ts + (bid == 8 * 40 ? 0 : 128); for (int i = 0; i < tsi; ++i) o[i] = tsb[i]; }
#endif

    // ---- epilogue (same as the register-staged kernel) ----
    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const int HoWo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)e * p.b_ds : p.w), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        p.y + (int64_t)e * p.y_ds, 0, (int)((int64_t)p.Cout * HoWo * p.B * 4), 0x00020000);
    float bv[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            bv[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
        }
    const int b = b0 + wm + lrow;
    const bool b_ok = b < p.B;
    if (!LRT) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const uint32_t off = (b_ok & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 4u : kOOB;
                const float v = bbb::apply_act(acc[nt][r] + bv[nt][r], p.act);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrs, off, 0, 0);
            }
    } else if (b_ok) {
        const int64_t ybase = (int64_t)e * p.y_ds + (int64_t)pix * p.B + b;
        const float* __restrict__ b2g = p.bias2 ? p.bias2 + (int64_t)e * p.b_ds : nullptr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (n < p.Cout) {
                    const int64_t o = ybase + (int64_t)n * HoWo * p.B;
                    float v = acc[nt][r] + bv[nt][r];
                    const float var = 1e-16f + (accv[nt][r] + (b2g ? b2g[n] : 0.0f));
                    if (p.y_mu) p.y_mu[o] = v;
                    if (p.y_var) p.y_var[o] = var;
                    if (p.sample) {
                        float z;
                        if (p.eps_ext) {
                            z = p.eps_ext[o];
                        } else {
                            const uint64_t idx = (uint64_t)(((int64_t)b * p.Cout + n) * HoWo + pix);
                            float z4[4];
                            bbb::normal4(idx >> 2, p.stream_id, p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)e, p.k0, p.k1, z4);
                            const int c = (int)(idx & 3);
                            z = c == 0 ? z4[0] : c == 1 ? z4[1] : c == 2 ? z4[2] : z4[3];
                        }
                        v = v + __builtin_amdgcn_sqrtf(var) * z;
                    }
                    p.y[o] = bbb::apply_act(v, p.act);
                }
            }
        }
    }
}

template <int BM, bool LRT>
int launch_one(const DmaArgs& a, int64_t blocks, hipStream_t st) {
    static bool configured = false;
    constexpr int bytes = smem_bytes<BM, LRT>();
    if (!configured) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&pconv_dma_kernel<BM, LRT>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (err != hipSuccess) return (int)err;
        configured = true;
    }
    hipLaunchKernelGGL((pconv_dma_kernel<BM, LRT>), dim3((unsigned)blocks), dim3(kThreads), bytes, st, a);
    return (int)hipGetLastError();
}

}  // namespace

// Internal entry used by pconv_gemm.hip's launcher (not part of the C ABI): returns -1000 if this variant does not
// apply, otherwise the launch status.
int bbb_pconv_dma_launch(const void* args_v, int lrt, int bm, int64_t blocks, void* stream) {
    const DmaArgs& a = *static_cast<const DmaArgs*>(args_v);
    hipStream_t st = (hipStream_t)stream;
    if (a.B < 4) return -1000;
    if (lrt) return launch_one<64, true>(a, blocks, st);
    if (bm == 128) return launch_one<128, false>(a, blocks, st);
    if (bm == 64) return launch_one<64, false>(a, blocks, st);
    return -1000;
}
