// bf16 pixel-major implicit GEMM (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the batch-innermost ensemble path.
//
// Same decomposition as pconv_gemm.hip -- one workgroup = (output pixel, 64 channels, 128 images), weights as the MFMA
// "A" operand, images as the lanes -- with bf16 storage on both sides:
//   activations  [draw][C][H][W][B]  bf16, B % 8 == 0 (rows move as 16-byte vectors)
//   weights      [draw][Cout][Kp]    bf16, Kp = K rounded up to 8 (rows start 16-byte aligned; the pad is zero),
//                                    K = Cin*kh*kw in the reference's (ci, r, q) order, written by the reparam kernel
//   output       bf16 (hidden layers) or fp32 (the logits layer), bias fp32, fp32 accumulation and epilogue.
// At 16x the fp32 matrix rate the contraction is no longer the bound; feeding it is.  So, unlike the fp32 kernel:
//   * weights must move as 16-byte vectors of 8 consecutive k.  With rows in the reference's (ci, r, q) order that forces
//     k to run over the FULL range (taps that fall into the padding are zeroed on the image side only: their row offset
//     is out of range for the buffer unit) -- twice the work on AlexNet's 2x2 / 4x4 maps.  With TAP-MAJOR rows
//     ((r, q, ci) order, written that way by the reparam kernel when cin % 8 == 0) the in-bounds taps of a pixel are
//     runs of cin consecutive k, so padding taps are skipped exactly as in the fp32 kernel;
//   * image rows are loaded 8 images per lane and stored to LDS as they are ([k][b]); the k-contiguous operand layout
//     the bf16 MFMA wants comes from the LDS transpose read (ds_read_b64_tr_b16), not from a register shuffle.
// The launches of a 10-draw CIFAR step are small (320-2560 workgroups of work) and a k-tile's operands come from the
// Infinity Cache (the weights were written by the kernel before), ~1 us away: one workgroup walking its k range tile
// by tile is latency-bound (measured 1.0-1.3 us per 64-k tile, 256 cycles of which are MFMA).  So the k range is also
// split INSIDE the workgroup: KG groups of 4 waves each own every KG-th 64-k tile, with their own LDS stage and their
// own loads in flight, and the partial accumulators are summed through LDS in a fixed order (deterministic).
// LDS pitches: image rows 320 B (= 64 B mod 256: the four rows of a transpose read and the two 16-column halves of a
// 32-lane group land on disjoint banks), weight rows 144 B (conflict-free for the 16-byte reads).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"
#include "pconv_args.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int LDWB = BK + 8;             // weight row pitch (elements): 144 B
constexpr int KCH = 256;                 // k entries per decode chunk and k-group
constexpr int TPC = KCH / BK;

__device__ __forceinline__ uint16_t f2bf(float v) {              // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(uint16_t, (__bf16)v);
}

// Workgroup = KG k-groups x (WN x WM) waves; every wave owns a 64-channel x 64-image block of D (2 x 2 MFMA tiles, so
// each LDS operand read feeds two MFMAs: the LDS pipe, not the matrix pipe, is what this kernel saturates first).
//   (WN, WM) = (2, 2): 128 channels x 128 images    (1, 4): 64 x 256    (1, 2): 64 x 128, two waves
// WS = wave-specialised: the GT MFMA waves only read LDS and issue MFMAs; 4 extra staging waves (one per SIMD) only do
// table lookups, buffer loads and ds_writes, one tile ahead into the other of TWO LDS stages with the tile after that in
// flight in their registers; one barrier per tile.  (s_memtime of the unspecialised kernel: 800 cycles issuing loads +
// 520 in ds_writes + 290 in barriers per iteration against 512 of MFMA, all serial inside the workgroup.)
template <bool OUT_F32, int WN, int WM, int KG, bool WS>
__global__ __launch_bounds__(64 * WN * WM * KG + (WS ? 256 : 0)) void pconv_bf16_kernel(const PConvArgs p) {
    static_assert(!WS || KG == 1, "wave specialisation replaces the k-groups");
    constexpr int GT = 64 * WN * WM;                              // MFMA threads per k-group
    constexpr int LT = WS ? 256 : GT;                             // threads that stage one tile
    constexpr int NSTG = WS ? 2 : 1;                              // LDS stages per k-group
    constexpr int BN = 64 * WN, BM = 64 * WM;
    constexpr int LDXB = BM + 32;                                 // image row pitch: = 64 B mod 256 for BM = 128, 256
    constexpr int WPASS = BN * 8 / LT, WROWS = LT / 8;            // weight tile: 8 lanes x 16 B per 128-byte row
    constexpr int XL = BM / 8, XROWS = LT / XL, XPASS = BK / XROWS;   // image tile: XL lanes x 16 B per row
    constexpr int KCHG = KCH * KG;
    constexpr int kStage = BK * LDXB + BN * LDWB;                 // elements per stage
    // dynamic LDS only (a static array in front would shift the 16-byte alignment of the carve)
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const bool producer = WS && __builtin_amdgcn_readfirstlane((int)threadIdx.x >= GT ? 1 : 0) != 0;
    const int kg = WS ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x / GT);
    uint16_t* Xs = smem + kg * kStage;                            // stage 0 of this group (stage s: + s * kStage)
    uint16_t* Ws = Xs + BK * LDXB;
    int32_t* kt_all = reinterpret_cast<int32_t*>(smem + KG * NSTG * kStage);   // [2][KCHG] image-row offsets per k
    int32_t* kw_all = kt_all + 2 * KCHG;                                   // [2][KCHG] weight-column offsets per k

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (item >= item_end || item >= (int64_t)p.G * p.Mtiles) return;
    const int g = (int)(item / p.Mtiles);
    const int j = (int)(item - (int64_t)g * p.Mtiles);
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;                                   // work units, as in pconv_gemm.hip
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * BN;
    const int pix = j / p.nbt;
    const int b0 = (j - pix * p.nbt) * BM;
    const int oh = pix / p.Wo, ow = pix - oh * p.Wo;
    const int ihb = oh * p.sh - p.ph, iwb = ow * p.sw - p.pw;
    const int Kp = p.Kp;
    // in-bounds tap rectangle of this pixel (tap-major rows only; otherwise every tap is enumerated)
    int r_lo = 0, q_lo = 0, nr = p.kh, nq = p.kw;
    if (p.wtap) {
        r_lo = ihb < 0 ? (-ihb + p.dh - 1) / p.dh : 0;
        q_lo = iwb < 0 ? (-iwb + p.dw - 1) / p.dw : 0;
        int r_hi = (p.H - 1 - ihb) >= 0 ? (p.H - 1 - ihb) / p.dh + 1 : 0;
        int q_hi = (p.W - 1 - iwb) >= 0 ? (p.W - 1 - iwb) / p.dw + 1 : 0;
        r_hi = r_hi < p.kh ? r_hi : p.kh;
        q_hi = q_hi < p.kw ? q_hi : p.kw;
        nr = r_hi > r_lo ? r_hi - r_lo : 0;
        nq = q_hi > q_lo ? q_hi - q_lo : 0;
    }
    const int K = p.wtap ? p.Cin * nr * nq : p.K;                 // contraction length of this pixel
    const int niter = (K + BK * KG - 1) / (BK * KG);              // every group runs the same number of iterations

    const int tid = (int)threadIdx.x - kg * GT;                   // thread within its k-group (MFMA role)
    const int ltid = WS ? (int)threadIdx.x - GT : tid;            // thread within the staging team
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (wave / WM) * 64, wm = (wave % WM) * 64;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    constexpr uint32_t kWInv = 0x7FFFFFF0u;                       // + a row offset (< 2^31): out of range, no wrap
    const uint32_t kXInv = p.x_inv;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const int64_t x_bytes = (int64_t)p.Cin * p.H * p.W * p.B * 2;
    const int64_t w_bytes = (int64_t)p.Cout * Kp * 2;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)w_bytes, 0x00020000);

    const int wr = ltid >> 3, wseg = (ltid & 7) * 8;
    const uint32_t wbase = (uint32_t)(n0 + wr) * (uint32_t)Kp * 2u;
    const uint32_t wstep = (uint32_t)WROWS * (uint32_t)Kp * 2u;
    const int xkr = ltid / XL, xb8 = (ltid % XL) * 8;
    const uint32_t xcol = (uint32_t)(b0 + xb8) * 2u;

    // k -> (weight column offset, image row offset), float-reciprocal division + fix-up (exact below 2^24)
    const float inv_khkw = 1.0f / (float)p.khkw, inv_kw = 1.0f / (float)p.kw;
    const float inv_cin = 1.0f / (float)p.Cin, inv_nq = nq > 0 ? 1.0f / (float)nq : 0.0f;
    auto fill_chunk = [&](int chunk) {
        for (int i = WS ? ltid : (int)threadIdx.x; i < KCHG; i += WS ? LT : GT * KG) {
            const int k = chunk * KCHG + i;
            uint32_t xo = kXInv, wo = kWInv;
            if (p.wtap) {
                if (k < K) {
                    int t = (int)((float)k * inv_cin);
                    int ci = k - t * p.Cin;
                    if (ci < 0) { --t; ci += p.Cin; } else if (ci >= p.Cin) { ++t; ci -= p.Cin; }
                    int rr = (int)((float)t * inv_nq);
                    int qq = t - rr * nq;
                    if (qq < 0) { --rr; qq += nq; } else if (qq >= nq) { ++rr; qq -= nq; }
                    const int r = r_lo + rr, q = q_lo + qq;
                    wo = (uint32_t)((r * p.kw + q) * p.Cin + ci) * 2u;
                    xo = (uint32_t)((ci * p.H + ihb + r * p.dh) * p.W + iwb + q * p.dw) * (uint32_t)p.B * 2u;
                }
            } else {
                if (k < Kp) wo = (uint32_t)k * 2u;                 // the zero pad columns K..Kp-1 are part of the row
                if (k < K) {
                    int ci = (int)((float)k * inv_khkw);
                    int rq = k - ci * p.khkw;
                    if (rq < 0) { --ci; rq += p.khkw; } else if (rq >= p.khkw) { ++ci; rq -= p.khkw; }
                    int r = (int)((float)rq * inv_kw);
                    int q = rq - r * p.kw;
                    if (q < 0) { --r; q += p.kw; } else if (q >= p.kw) { ++r; q -= p.kw; }
                    const int ih = ihb + r * p.dh, iw = iwb + q * p.dw;
                    if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                        xo = (uint32_t)((ci * p.H + ih) * p.W + iw) * (uint32_t)p.B * 2u;
                }
            }
            kt_all[(chunk & 1) * KCHG + i] = (int32_t)xo;
            kw_all[(chunk & 1) * KCHG + i] = (int32_t)wo;
        }
    };

    u32x4 wreg[WPASS], xreg[XPASS];
    auto load_tile = [&](int it) {                       // iteration `it`: this group's 64-k tile is it*KG + kg
        const int buf = (it / TPC) & 1;
        const int kb = ((it % TPC) * KG + kg) * BK;
        uint32_t wo = wbase + (uint32_t)kw_all[buf * KCHG + kb + wseg];   // 8 consecutive k of one tap / one row
        uint32_t xo[XPASS];
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) xo[ps] = (uint32_t)kt_all[buf * KCHG + kb + xkr + ps * XROWS] + xcol;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) xreg[ps] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps], 0, 0));
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) wreg[ps] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wo + (uint32_t)ps * wstep, 0, 0));
    };
    auto store_tile = [&](int stage) {
        uint16_t* Xd = Xs + stage * kStage;
        uint16_t* Wd = Ws + stage * kStage;
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) *reinterpret_cast<u32x4*>(&Wd[(wr + ps * WROWS) * LDWB + wseg]) = wreg[ps];
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) *reinterpret_cast<u32x4*>(&Xd[(xkr + ps * XROWS) * LDXB + xb8]) = xreg[ps];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    const int lrow = lane & 31, lk = lane >> 5;
    // transpose-read address of this lane inside a 16-k step: 16-lane group -> (k half, 16-column block); the lane
    // supplies row (t >> 2) of 4, columns 4*(t & 3)..+3, and receives column t, rows 0..3
    const int tg = lane >> 4, tt = lane & 15;
    const int tr_off = ((8 * (tg >> 1) + (tt >> 2)) * LDXB + wm + 16 * (tg & 1) + 4 * (tt & 3));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

    auto mma_tile = [&](int stage) {
        const uint16_t* Xs = smem + (kg * NSTG + stage) * kStage;
        const uint16_t* Ws = Xs + BK * LDXB;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 b[2], a[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + kk * 16 * LDXB]));
                const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + (kk * 16 + 4) * LDXB]));
                b[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                a[nt] = *reinterpret_cast<const bf16x8*>(&Ws[(wn + nt * 32 + lrow) * LDWB + kk * 16 + lk * 8]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt], b[mt], acc[nt][mt], 0, 0, 0);
        }
    };

    if constexpr (WS) {
        if (producer) {
            // ---- staging waves: tile t+1 -> stage (t+1)&1 while the MFMA waves are on stage t&1; tile t+2 in flight ----
            if (niter > 0) fill_chunk(0);
            __syncthreads();                                       // P0: table chunk 0 visible to all staging waves
            if (niter > 0) {
                load_tile(0);
                if (KCHG < K) fill_chunk(1);
                store_tile(0);
            }
            __syncthreads();                                       // P1: stage 0 = tile 0, table chunk 1 visible
            if (niter > 1) load_tile(1);
            for (int t = 0; t < niter; ++t) {
                if (t + 1 < niter) {
                    store_tile((t + 1) & 1);                       // registers hold tile t+1, issued one iteration ago
                    if (t + 2 < niter) load_tile(t + 2);
                }
                // decode chunk c+1 during the first tile of chunk c (c >= 1): its buffer was last read two barriers
                // ago (load_tile(c*TPC - 1) at iteration c*TPC - 3) and is first read TPC - 2 barriers from now
                if ((t % TPC) == 0 && t / TPC >= 1 && (t / TPC + 1) * KCHG < K) fill_chunk(t / TPC + 1);
                __syncthreads();
            }
            return;
        }
        __syncthreads();                                           // P0
        __syncthreads();                                           // P1
        for (int t = 0; t < niter; ++t) {
            mma_tile(t & 1);
            __syncthreads();
        }
    } else {
        fill_chunk(0);
        __syncthreads();
        load_tile(0);
        if (KCHG < K) fill_chunk(1);
        store_tile(0);
        __syncthreads();
        for (int t = 0; t < niter; ++t) {
            const bool more = (t + 1) < niter;
            if (more) load_tile(t + 1);
            if ((t % TPC) == 1 && t / TPC >= 1 && (t / TPC + 1) * KCHG < K) fill_chunk(t / TPC + 1);
            mma_tile(0);
            __syncthreads();
            if (more) store_tile(0);
            __syncthreads();
        }
    }

    // ---- cross-group reduction (fixed order: group 0 + 1 + ...), through the now idle stage memory ----
    if (KG > 1) {
        float* red = reinterpret_cast<float*>(smem);               // [KG-1][64][GT] floats
        if (kg > 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((kg - 1) * 64 + (nt * 2 + mt) * 16 + r) * GT + tid] = acc[nt][mt][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g2 = 1; g2 < KG; ++g2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][mt][r] += red[((g2 - 1) * 64 + (nt * 2 + mt) * 16 + r) * GT + tid];
    }

    // ---- epilogue ----
    const int HoWo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : reinterpret_cast<const float*>(p.w)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    constexpr int OSZ = OUT_F32 ? 4 : 2;
    char* yb = reinterpret_cast<char*>(p.y) + (int64_t)e * p.y_ds * OSZ;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((int64_t)p.Cout * HoWo * p.B * OSZ), 0x00020000);
    if constexpr (OUT_F32) {
        // logits layer (a handful of channels): rows = channels, lanes = images, straight from the accumulators
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int b = b0 + wm + mt * 32 + lrow;
                    const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 4u : kOOB;
                    const float v = bbb::apply_act(acc[nt][mt][r] + bv, p.act);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrs, off, 0, 0);
                }
            }
    } else {
        // 64 two-byte stores per lane cost ~340 cycles each (s_memtime: the epilogue was as long as 8 k-iterations).
        // Instead every wave transposes its 64x64 block through LDS -- bias + activation + rounding in registers,
        // ds_write_b16 into a [channel][image] image -- and stores 16-byte vectors of 8 images: 8 stores per lane, each
        // wave-store covering 8 channels x 128 contiguous bytes.
        constexpr int TP = 64 + 8;                                  // staging row pitch (elements): 144 B
        __syncthreads();                                            // stage / reduction memory is free from here on
        uint16_t* T = smem + wave * (64 * TP);
        auto stage_block = [&](auto act) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int nl = nt * 32 + 8 * r4 + 4 * lk;           // 4 consecutive channels: accumulator rows r4*4 .. r4*4+3
                // (bit_cast of the WHOLE result: indexing the builtin's return value directly reads element 0 four times
                // with this hipcc -- caught by the parity tests)
                const f32x4 bq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n0 + wn + nl) * 4u, 0, 0));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        T[(nl + i) * TP + mt * 32 + lrow] = f2bf(act(acc[nt][mt][r4 * 4 + i] + bq[i]));
            }
        };
        // one branch on the activation kind instead of one per element
        if (p.act == 2)      stage_block([](float v) { return bbb::apply_act(v, 2); });
        else if (p.act == 1) stage_block([](float v) { return fmaxf(v, 0.0f); });
        else                 stage_block([](float v) { return v; });
        // same-wave LDS accesses complete in order, so no s_barrier is needed between the writes above and the reads
        // below -- but the compiler must not move the (differently typed) vector reads above the 2-byte writes
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int v = ps * 64 + lane;
            const int row = v >> 3, grp = v & 7;
            const u32x4 q = *reinterpret_cast<const u32x4*>(&T[row * TP + grp * 8]);
            const int n = n0 + wn + row, b = b0 + wm + grp * 8;
            const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 2u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(q, yrs, off, 0, 0);
        }
    }
}

// ---- first layers: rows of at most 128 k (3Conv3FC conv1: K = 75, LeNet conv1: K = 25) ----
// With so short a contraction the general kernel is all fixed cost: one workgroup per output pixel = offset table, weight
// tile, two barriers and an epilogue for 2 tiles of MFMAs, ~10 us of dependent latencies per workgroup (measured: conv1 of
// 3Conv3FC bs 256, four draws: 62 us for 5 GFLOP).  Here a workgroup keeps the weight tile of its 32 / 64 channels in
// REGISTERS (MFMA A operands, read once from global memory) and walks `px_run` consecutive output pixels for its 256 images:
// per pixel one image-row tile [K][256] through LDS (the next pixel's rows in flight in registers while this one multiplies
// and runs its epilogue), KS MFMA steps, and the bf16 epilogue through a wave-private LDS transpose: 62 -> 31 us.  What is
// left is the epilogue's VALU work: softplus is two quarter-rate transcendentals per output element, ~18 us for those 33.5 M
// elements on 1024 SIMDs.  (A variant that staged the input strip of the whole run once and addressed it per pixel -- 2.5x
// fewer image-row fetches, but one workgroup per CU -- measured 45 us: profiles/r03_notes.md section 10.)
// Same MFMA sequence per output element as the general kernel with one k-group (extra all-zero steps add +0): bit-identical.
template <int NT, int KS>
__global__ __launch_bounds__(256) void pconv_bf16_smallk_kernel(const PConvArgs p) {
    constexpr int BM = 256, BN = 32 * NT, KR = KS * 16, LDXB = BM + 32, TP = 64 + 8;
    constexpr int XPASS = KR / 8;                                 // 32 lanes x 16 B per row, 8 rows per pass
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* Xs = smem;                                          // [KR][LDXB]
    uint16_t* Tall = Xs + KR * LDXB;                              // [4 waves][BN][TP] epilogue staging
    int32_t* kt = reinterpret_cast<int32_t*>(Tall + 4 * BN * TP); // [px_run][KR] image-row offsets

    const int bid = blockIdx.x, xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    const int HoWo = p.Ho * p.Wo;
    const int runs = (HoWo + p.px_run - 1) / p.px_run;
    const int64_t per_g = (int64_t)runs * p.nbt;
    if (item >= item_end || item >= (int64_t)p.G * per_g) return;
    const int g = (int)(item / per_g);
    const int rem = (int)(item - (int64_t)g * per_g);
    const int run = rem / p.nbt;
    const int b0 = (rem - run * p.nbt) * BM;
    const int pix0 = run * p.px_run;
    const int npx = p.px_run < HoWo - pix0 ? p.px_run : HoWo - pix0;
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * BN;
    const int Kp = p.Kp;

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave * 64;
    const int lrow = lane & 31, lk = lane >> 5;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const uint32_t kXInv = p.x_inv;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((int64_t)p.Cin * p.H * p.W * p.B * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)((int64_t)p.Cout * Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : reinterpret_cast<const float*>(p.w)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    char* yb = reinterpret_cast<char*>(p.y) + (int64_t)e * p.y_ds * 2;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((int64_t)p.Cout * HoWo * p.B * 2), 0x00020000);

    // image-row offsets of every (pixel of the run, k): reference (ci, r, q) order, padding taps and k >= K invalid
    {
        const float inv_khkw = 1.0f / (float)p.khkw, inv_kw = 1.0f / (float)p.kw, inv_kr = 1.0f / (float)KR, inv_wo = 1.0f / (float)p.Wo;
        for (int i = tid; i < npx * KR; i += 256) {
            int pi = (int)((float)i * inv_kr);
            int k = i - pi * KR;
            if (k < 0) { --pi; k += KR; } else if (k >= KR) { ++pi; k -= KR; }
            const int pix = pix0 + pi;
            int oh = (int)((float)pix * inv_wo);
            int ow = pix - oh * p.Wo;
            if (ow < 0) { --oh; ow += p.Wo; } else if (ow >= p.Wo) { ++oh; ow -= p.Wo; }
            uint32_t xo = kXInv;
            if (k < p.K) {
                int ci = (int)((float)k * inv_khkw);
                int rq = k - ci * p.khkw;
                if (rq < 0) { --ci; rq += p.khkw; } else if (rq >= p.khkw) { ++ci; rq -= p.khkw; }
                int r = (int)((float)rq * inv_kw);
                int q = rq - r * p.kw;
                if (q < 0) { --r; q += p.kw; } else if (q >= p.kw) { ++r; q -= p.kw; }
                const int ih = oh * p.sh - p.ph + r * p.dh, iw = ow * p.sw - p.pw + q * p.dw;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) xo = (uint32_t)((ci * p.H + ih) * p.W + iw) * (uint32_t)p.B * 2u;
            }
            kt[i] = (int32_t)xo;
        }
    }
    // weights: MFMA A operands straight from global memory (row n, 8 consecutive k per lane), kept for the whole run
    bf16x8 a[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int n = n0 + nt * 32 + lrow, k = kk * 16 + lk * 8;
            const uint32_t off = (n < p.Cout && k < Kp) ? (uint32_t)(n * Kp + k) * 2u : kOOB;
            a[nt][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
        }
    f32x4 bq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            bq[nt][r4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n0 + nt * 32 + 8 * r4 + 4 * lk) * 4u, 0, 0));

    const int xkr = tid >> 5, xb8 = (tid & 31) * 8;
    const uint32_t xcol = (uint32_t)(b0 + xb8) * 2u;
    u32x4 xreg[XPASS];
    auto load_x = [&](int pi) {
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps)
            xreg[ps] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (uint32_t)kt[pi * KR + xkr + ps * 8] + xcol, 0, 0));
    };
    auto store_x = [&]() {
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) *reinterpret_cast<u32x4*>(&Xs[(xkr + ps * 8) * LDXB + xb8]) = xreg[ps];
    };
    const int tg = lane >> 4, tt = lane & 15;
    const int tr_off = ((8 * (tg >> 1) + (tt >> 2)) * LDXB + wm + 16 * (tg & 1) + 4 * (tt & 3));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    uint16_t* T = Tall + wave * (BN * TP);

    __syncthreads();                                               // offset table visible
    load_x(0);
    store_x();
    __syncthreads();
    for (int pi = 0; pi < npx; ++pi) {
        const bool more = pi + 1 < npx;
        if (more) load_x(pi + 1);
        f32x16 acc[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            bf16x8 b[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + kk * 16 * LDXB]));
                const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + (kk * 16 + 4) * LDXB]));
                b[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt][kk], b[mt], acc[nt][mt], 0, 0, 0);
        }
        // epilogue of this pixel: bias + activation + rounding in registers, wave-private LDS transpose, 16-byte stores -- BEFORE
        // the next pixel's rows go to LDS, so that their loads (issued above) have the whole epilogue to arrive
        const int pix = pix0 + pi;
        auto stage_block = [&](auto act) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nl = nt * 32 + 8 * r4 + 4 * lk;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            T[(nl + i) * TP + mt * 32 + lrow] = f2bf(act(acc[nt][mt][r4 * 4 + i] + bq[nt][r4][i]));
                }
        };
        if (p.act == 2)      stage_block([](float v) { return bbb::apply_act(v, 2); });
        else if (p.act == 1) stage_block([](float v) { return fmaxf(v, 0.0f); });
        else                 stage_block([](float v) { return v; });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < 4 * NT; ++ps) {
            const int v = ps * 64 + lane;
            const int row = v >> 3, grp = v & 7;
            const u32x4 q = *reinterpret_cast<const u32x4*>(&T[row * TP + grp * 8]);
            const int n = n0 + row, b = b0 + wm + grp * 8;
            const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 2u : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(q, yrs, off, 0, 0);
        }
        __syncthreads();                                           // every wave has read this pixel's rows
        if (more) store_x();
        __syncthreads();                                           // the next pixel's rows are in LDS
    }
}

// ---- first layers followed by [activation ->] MaxPool2d(pk, ps): the pooling inside the launch ----
// 3Conv3FC conv1 (bf16, bs 256, 16 steps per launch) is 104 us of which most is the per-pixel epilogue and the 268 MB it stores;
// the 3 x 3 / 2 pooling launch behind it reads those 268 MB again (63 us at the HBM rate).  Here a workgroup owns `csplit`-th of
// a POOLED row for its 256 images and 32 / 64 channels: it walks the conv pixels of that strip column by column (pk rows each),
// per pixel the unchanged contraction of pconv_bf16_smallk_kernel, then max into the (at most two, pk <= 2 ps) pooling windows
// that contain the column -- in fp32, BEFORE bias, activation and rounding: x -> round_bf16(act(x + bias)) is non-decreasing, so
// max-then-epilogue equals epilogue-then-max element for element -- and runs the epilogue ONCE per pooled pixel: 4.5x fewer
// epilogues and stored bytes for 3 x 3 / 2 windows, against pk / ps (1.5x) the contraction work for the rows two pooled rows
// share.  Image-row offsets are decoded per column (pk x KR entries, one per thread) into a two-deep table.
// (two waves per SIMD for the 32-channel form in either output layout: the channel-interleaved epilogue would otherwise take 231
// registers and lose the second wave; the 64-channel form needs more than 256 and runs one)
template <int NT, int KS, bool C8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT == 1 ? 2 : 1))) void pconv_bf16_smallk_pool_kernel(const PConvArgs p) {
    constexpr int BM = 256, BN = 32 * NT, KR = KS * 16, LDXB = BM + 32, TP = 64 + 8, PKMAX = 3;
    constexpr int XPASS = KR / 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* Xs = smem;                                          // [KR][LDXB]
    uint16_t* Tall = Xs + KR * LDXB;                              // [4 waves][BN][TP] epilogue staging
    int32_t* kt = reinterpret_cast<int32_t*>(Tall + 4 * BN * TP); // [2][PKMAX][KR] image-row offsets of the current / next column

    const int pk = p.pool >> 8, ps = p.pool & 255;
    const int Hp = (p.Ho - pk) / ps + 1, Wp = (p.Wo - pk) / ps + 1;
    const int csplit = p.px_run;                                  // workgroups per pooled row
    const int wpc = (Wp + csplit - 1) / csplit;                   // pooled pixels per workgroup
    const int bid = blockIdx.x, xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    const int64_t per_g = (int64_t)Hp * csplit * p.nbt;
    if (item >= item_end || item >= (int64_t)p.G * per_g) return;
    const int g = (int)(item / per_g);
    int rem = (int)(item - (int64_t)g * per_g);
    const int ph = rem / (csplit * p.nbt);
    rem -= ph * csplit * p.nbt;
    const int cs = rem / p.nbt;
    const int b0 = (rem - cs * p.nbt) * BM;
    const int pw0 = cs * wpc;
    const int pw1 = (pw0 + wpc) < Wp ? (pw0 + wpc) : Wp;
    if (pw0 >= pw1) return;
    const int c0 = pw0 * ps, c1 = (pw1 - 1) * ps + pk - 1;        // conv columns of the strip (inclusive)
    const int r0 = ph * ps;                                       // first conv row
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * BN;
    const int Kp = p.Kp;
    const int HpWp = Hp * Wp;

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave * 64;
    const int lrow = lane & 31, lk = lane >> 5;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const uint32_t kXInv = p.x_inv;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((int64_t)p.Cin * p.H * p.W * p.B * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)((int64_t)p.Cout * Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : reinterpret_cast<const float*>(p.w)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    char* yb = reinterpret_cast<char*>(p.y) + (int64_t)e * p.y_ds * 2;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((int64_t)p.Cout * HpWp * p.B * 2), 0x00020000);

    // image-row offsets of column c's pk pixels: reference (ci, r, q) order, padding taps and k >= K invalid; one entry per thread
    const float inv_khkw = 1.0f / (float)p.khkw, inv_kw = 1.0f / (float)p.kw, inv_kr = 1.0f / (float)KR;
    auto fill_col = [&](int c) {
        int32_t* dst = kt + (c & 1) * (PKMAX * KR);
        for (int i = tid; i < pk * KR; i += 256) {
            int rr = (int)((float)i * inv_kr);
            int k = i - rr * KR;
            if (k < 0) { --rr; k += KR; } else if (k >= KR) { ++rr; k -= KR; }
            uint32_t xo = kXInv;
            if (k < p.K) {
                int ci = (int)((float)k * inv_khkw);
                int rq = k - ci * p.khkw;
                if (rq < 0) { --ci; rq += p.khkw; } else if (rq >= p.khkw) { ++ci; rq -= p.khkw; }
                int r = (int)((float)rq * inv_kw);
                int q = rq - r * p.kw;
                if (q < 0) { --r; q += p.kw; } else if (q >= p.kw) { ++r; q -= p.kw; }
                const int ih = (r0 + rr) * p.sh - p.ph + r * p.dh, iw = c * p.sw - p.pw + q * p.dw;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) xo = (uint32_t)((ci * p.H + ih) * p.W + iw) * (uint32_t)p.B * 2u;
            }
            dst[rr * KR + k] = (int32_t)xo;
        }
    };
    fill_col(c0);
    // weights: MFMA A operands straight from global memory, kept for the whole strip
    bf16x8 a[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int n = n0 + nt * 32 + lrow, k = kk * 16 + lk * 8;
            const uint32_t off = (n < p.Cout && k < Kp) ? (uint32_t)(n * Kp + k) * 2u : kOOB;
            a[nt][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
        }
    f32x4 bq[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            bq[nt][r4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n0 + nt * 32 + 8 * r4 + 4 * lk) * 4u, 0, 0));

    const int xkr = tid >> 5, xb8 = (tid & 31) * 8;
    const uint32_t xcol = (uint32_t)(b0 + xb8) * 2u;
    u32x4 xreg[XPASS];
    auto load_x = [&](int c, int rr) {
        const int32_t* src = kt + (c & 1) * (PKMAX * KR) + rr * KR;
#pragma unroll
        for (int ps2 = 0; ps2 < XPASS; ++ps2)
            xreg[ps2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (uint32_t)src[xkr + ps2 * 8] + xcol, 0, 0));
    };
    auto store_x = [&]() {
#pragma unroll
        for (int ps2 = 0; ps2 < XPASS; ++ps2) *reinterpret_cast<u32x4*>(&Xs[(xkr + ps2 * 8) * LDXB + xb8]) = xreg[ps2];
    };
    const int tg = lane >> 4, tt = lane & 15;
    const int tr_off = ((8 * (tg >> 1) + (tt >> 2)) * LDXB + wm + 16 * (tg & 1) + 4 * (tt & 3));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    uint16_t* T = Tall + wave * (BN * TP);

    f32x16 pnew[NT][2], pold[NT][2];          // running maxima of the window that starts latest / of the one before it
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { pnew[nt][mt][r] = -__builtin_inff(); pold[nt][mt][r] = -__builtin_inff(); }

    __syncthreads();                                               // column c0's offsets visible
    load_x(c0, 0);
    store_x();
    __syncthreads();
#pragma clang loop unroll(disable)
    for (int c = c0; c <= c1; ++c) {
        // the windows this column belongs to (wave-uniform): w_hi starts latest (c / ps), w_hi - 1 may still be open
        const int w_hi = c / ps;
        const bool hi_ok = w_hi < pw1;                             // (>= pw0 by construction; the strip's last columns lie past its last window's start)
        const bool starts = c == w_hi * ps;                        // a window position starts at this column (inside the strip or not)
        const bool hi_first = hi_ok && starts;
        const bool lo_ok = (w_hi - 1) >= pw0 && c <= (w_hi - 1) * ps + pk - 1;
        if (starts) {
            // the window that was the newest becomes the older one (still open only where windows overlap, pk > ps); the new one
            // starts from -inf
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    pold[nt][mt] = pnew[nt][mt];
#pragma unroll
                    for (int r = 0; r < 16; ++r) pnew[nt][mt][r] = -__builtin_inff();
                }
        }
#pragma clang loop unroll(disable)
        for (int rr = 0; rr < pk; ++rr) {
            const bool last_row = rr + 1 == pk;
            const bool more = !(last_row && c == c1);
            if (rr == 0 && c < c1) fill_col(c + 1);                // (its buffer was last read a column ago: barriers in between)
            if (more) { if (!last_row) load_x(c, rr + 1); else load_x(c + 1, 0); }
            f32x16 acc[NT][2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                bf16x8 b[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + kk * 16 * LDXB]));
                    const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(&Xs[tr_off + mt * 32 + (kk * 16 + 4) * LDXB]));
                    b[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt][kk], b[mt], acc[nt][mt], 0, 0, 0);
            }
            // this conv pixel into its windows (fp32 contraction results: the epilogue comes after the maximum).  Both running maxima
            // are updated UNCONDITIONALLY: `pold` is dead between the column that closes its window and the next window start (where
            // it is overwritten), `pnew` is dead past the strip's last window start -- so values outside a window only ever land in
            // a register nobody reads.  One instruction per maximum: med3(a, v, +inf) = max(a, v) (fmaxf canonicalises its operands first;
            // both are arithmetic results) -- 2 VALU instructions per accumulator element instead of ~8 (strip form 124.6 -> 111 us at 16
            // steps per launch).  An intrinsic, not inline asm: the compiler must see the read of the MFMA's result registers to
            // keep the wait states that hazard needs (an asm v_max_f32 here computed wrong maxima once the schedule changed).
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[nt][mt][r];
                        pnew[nt][mt][r] = __builtin_amdgcn_fmed3f(pnew[nt][mt][r], v, __builtin_inff());
                        pold[nt][mt][r] = __builtin_amdgcn_fmed3f(pold[nt][mt][r], v, __builtin_inff());
                    }
            // a window is complete after the last row of its last column: older window first (overlapping windows), else the
            // newest (pk <= ps); the host admits only geometries where at most one window closes per column
            const bool emit_lo = last_row && lo_ok && c == (w_hi - 1) * ps + pk - 1;
            const bool emit_hi = last_row && !emit_lo && hi_ok && c == w_hi * ps + pk - 1;
            if (emit_lo || emit_hi) {
                const int ppix = ph * Wp + (emit_lo ? w_hi - 1 : w_hi);
                auto stage_block = [&](auto act) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const int nl = nt * 32 + 8 * r4 + 4 * lk;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int mt = 0; mt < 2; ++mt) {
                                    const float m = emit_lo ? pold[nt][mt][r4 * 4 + i] : pnew[nt][mt][r4 * 4 + i];
                                    T[(nl + i) * TP + mt * 32 + lrow] = f2bf(act(m + bq[nt][r4][i]));
                                }
                        }
                };
                // channel-interleaved output (BBB_BF16_OUT_C8): a lane's four consecutive channels of an image are 8 adjacent bytes of
                // [cout / 8][hp][wp][B][8] -- straight from the registers, a wave-store covers 32 images x 16 B
                auto store_c8 = [&](auto act) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) {
                                uint32_t h[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    h[i] = f2bf(act((emit_lo ? pold[nt][mt][r4 * 4 + i] : pnew[nt][mt][r4 * 4 + i]) + bq[nt][r4][i]));
                                const int n = n0 + nt * 32 + 8 * r4, b = b0 + wm + mt * 32 + lrow;
                                const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)((((int64_t)(n >> 3) * HpWp + ppix) * p.B + b) * 16 + lk * 8) : kOOB;
                                __builtin_amdgcn_raw_buffer_store_b64(u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)}, yrs, off, 0, 0);
                            }
                };
                if constexpr (C8) {
                    if (p.act == 2)      store_c8([](float v) { return bbb::apply_act(v, 2); });
                    else if (p.act == 1) store_c8([](float v) { return fmaxf(v, 0.0f); });
                    else                 store_c8([](float v) { return v; });
                } else {
                    if (p.act == 2)      stage_block([](float v) { return bbb::apply_act(v, 2); });
                    else if (p.act == 1) stage_block([](float v) { return fmaxf(v, 0.0f); });
                    else                 stage_block([](float v) { return v; });
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int ps2 = 0; ps2 < 4 * NT; ++ps2) {
                        const int v = ps2 * 64 + lane;
                        const int row = v >> 3, grp = v & 7;
                        const u32x4 q = *reinterpret_cast<const u32x4*>(&T[row * TP + grp * 8]);
                        const int n = n0 + row, b = b0 + wm + grp * 8;
                        const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HpWp + ppix) * p.B + b) * 2u : kOOB;
                        __builtin_amdgcn_raw_buffer_store_b128(q, yrs, off, 0, 0);
                    }
                }
            }
            __syncthreads();                                       // every wave has read this pixel's rows
            if (more) store_x();
            __syncthreads();                                       // the next pixel's rows (and the next column's offsets) are in LDS
        }
    }
}

// ---- the pooled first layer with its input WINDOW resident in LDS (round 5, second form) ----
// pconv_bf16_smallk_pool_kernel fetches a [K][256-image] tile of image rows per conv pixel (40 KB for 3Conv3FC conv1) although
// neighbouring pixels share 4/5 of their taps: measured, half of its time is that traffic (the kernel without its loads: 58 of
// 104 us, profiles/r05_notes.md).  Here a workgroup (128 images, <= 32 channels) loads the input window of its strip ONCE --
// Cin x WR x WC image rows: 189 rows of 256 B for a strip of two 3 x 3 / 2 pooled pixels of a 5 x 5 stride-1 layer -- and every conv
// pixel reads its B operands straight out of it: ds_read_tr16_b64 takes a per-lane row address, so the k -> (ci, r, q) -> window-row
// map is ten per-lane offsets plus one per-pixel offset, no staging, no barrier inside the pixel loop (the waves run free), operand
// reads of the next pixel in flight under the current pixel's MFMAs.  Out-of-image taps are rows the loader filled with zeros (the
// buffer unit returns 0 for their out-of-range offsets); k >= K reads a zero row.  Same MFMA sequence per output element as the
// other forms: bit-identical.
constexpr int kWinPasses = 13;             // 16-row passes of the window loader: windows of up to 13 * 16 - 1 = 207 image rows (+ the zero row)

template <int KS, bool C8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pconv_bf16_smallk_poolwin_kernel(const PConvArgs p) {
    constexpr int BM = 128, LDXB = BM + 32, TP = 32 + 8;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];

    const int pk = p.pool >> 8, ps = p.pool & 255;
    const int Hp = (p.Ho - pk) / ps + 1, Wp = (p.Wo - pk) / ps + 1;
    const int nstr = p.px_run;                                    // strips per pooled row
    const int wpc = (Wp + nstr - 1) / nstr;                       // pooled pixels per strip
    const int bid = blockIdx.x, xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    const int64_t per_g = (int64_t)Hp * nstr * p.nbt;
    if (item >= item_end || item >= (int64_t)p.G * per_g) return;
    const int g = (int)(item / per_g);
    int rem = (int)(item - (int64_t)g * per_g);
    const int ph = rem / (nstr * p.nbt);
    rem -= ph * nstr * p.nbt;
    const int cs = rem / p.nbt;
    const int b0 = (rem - cs * p.nbt) * BM;
    const int pw0 = cs * wpc;
    const int pw1 = (pw0 + wpc) < Wp ? (pw0 + wpc) : Wp;
    if (pw0 >= pw1) return;
    const int c0 = pw0 * ps, c1 = (pw1 - 1) * ps + pk - 1;        // conv columns of the strip (inclusive)
    const int ncols = c1 - c0 + 1;
    const int r0 = ph * ps;
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * 32;
    const int Kp = p.Kp;
    const int HpWp = Hp * Wp;
    // the window: input rows / columns the strip's conv pixels touch
    const int WR = (pk - 1) * p.sh + (p.kh - 1) * p.dh + 1;
    const int WC = (ncols - 1) * p.sw + (p.kw - 1) * p.dw + 1;
    const int NR = p.Cin * WR * WC;                               // + one all-zero row at index NR
    uint16_t* Xw = smem;                                          // [NR + 1][LDXB]
    uint16_t* Tall = smem + (size_t)p.Mtiles * LDXB;              // p.Mtiles = rows the host sized the window for (widest strip + 1)

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave * 32;
    const int lrow = lane & 31, lk = lane >> 5;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const uint32_t kXInv = p.x_inv;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((int64_t)p.Cin * p.H * p.W * p.B * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)((int64_t)p.Cout * Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : reinterpret_cast<const float*>(p.w)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    char* yb = reinterpret_cast<char*>(p.y) + (int64_t)e * p.y_ds * 2;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((int64_t)p.Cout * HpWp * p.B * 2), 0x00020000);

    // weights: MFMA A operands straight from global memory, kept for the whole strip (issued first: in flight under the window load)
    bf16x8 a[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int n = n0 + lrow, k = kk * 16 + lk * 8;
        const uint32_t off = (n < p.Cout && k < Kp) ? (uint32_t)(n * Kp + k) * 2u : kOOB;
        a[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
    }
    f32x4 bq[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
        bq[r4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n0 + 8 * r4 + 4 * lk) * 4u, 0, 0));

    // ---- the window, once: 16 lanes x 16 bytes per image row, 16 rows per pass; EVERY pass's load is issued before the first
    //      store (one memory round trip for the whole window: the host keeps it <= kWinPasses * 16 rows) ----
    const int l16 = tid & 15, rsub = tid >> 4;
    u32x4 wv[kWinPasses];
    {
        const uint32_t xcol = (uint32_t)(b0 + l16 * 8) * 2u;
        const int ih0 = r0 * p.sh - p.ph, iw0 = c0 * p.sw - p.pw;
        const int wrc = WR * WC;
        const float inv_wrc = 1.0f / (float)wrc, inv_wc = 1.0f / (float)WC;
#pragma unroll
        for (int u = 0; u < kWinPasses; ++u) {
            const int row = u * 16 + rsub;
            uint32_t xo = kXInv;
            if (row < NR) {
                int ci = (int)((float)row * inv_wrc);
                int r2 = row - ci * wrc;
                if (r2 < 0) { --ci; r2 += wrc; } else if (r2 >= wrc) { ++ci; r2 -= wrc; }
                int wr = (int)((float)r2 * inv_wc);
                int wc = r2 - wr * WC;
                if (wc < 0) { --wr; wc += WC; } else if (wc >= WC) { ++wr; wc -= WC; }
                const int ih = ih0 + wr, iw = iw0 + wc;
                if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) xo = (uint32_t)((ci * p.H + ih) * p.W + iw) * (uint32_t)p.B * 2u;
            }
            wv[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo + xcol, 0, 0));
        }
    }
    // -- while the window is in flight: per-lane window offsets (BYTES) of the k rows this lane SUPPLIES to a transpose read (row t >> 2 of its 16-lane group's four,
    // k half by the group): k -> (ci, r, q) -> ((ci * WR + r * dh) * WC + q * dw) * LDXB; k >= K: the zero row, pixel offset masked out
    const int tg = lane >> 4, tt = lane & 15;
    const int colo = wm + 16 * (tg & 1) + 4 * (tt & 3);
    uint32_t kbase[KS][2], kmask[KS][2];
    const float inv_khkw = 1.0f / (float)p.khkw, inv_kw = 1.0f / (float)p.kw;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kk * 16 + 8 * (tg >> 1) + (tt >> 2) + 4 * h;
            uint32_t v = (uint32_t)(NR * LDXB + colo) * 2u, m = 0u;
            if (k < p.K) {
                int ci = (int)((float)k * inv_khkw);             // float-reciprocal division + fix-up (exact for these sizes)
                int rq = k - ci * p.khkw;
                if (rq < 0) { --ci; rq += p.khkw; } else if (rq >= p.khkw) { ++ci; rq -= p.khkw; }
                int r = (int)((float)rq * inv_kw);
                int q = rq - r * p.kw;
                if (q < 0) { --r; q += p.kw; } else if (q >= p.kw) { ++r; q -= p.kw; }
                v = (uint32_t)(((ci * WR + r * p.dh) * WC + q * p.dw) * LDXB + colo) * 2u;
                m = 0xFFFFFFFFu;
            }
            kbase[kk][h] = v;
            kmask[kk][h] = m;
        }
    // (the per-lane offset tables above were computed while the window's loads were in flight)
#pragma unroll
    for (int u = 0; u < kWinPasses; ++u) {
        const int row = u * 16 + rsub;
        if (row <= NR) *reinterpret_cast<u32x4*>(&Xw[row * LDXB + l16 * 8]) = wv[u];             // (row NR: its offset is invalid -> zeros)
    }
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const lds_char_ptr xw_lds = (lds_char_ptr)Xw;
    auto load_b = [&](uint32_t pixoff, bf16x8 (&b)[KS]) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xw_lds + (kbase[kk][0] + (pixoff & kmask[kk][0]))));
            const s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(xw_lds + (kbase[kk][1] + (pixoff & kmask[kk][1]))));
            b[kk] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    uint16_t* T = Tall + wave * (32 * TP);
    // max without the canonicalisation fmaxf asks for (both operands are arithmetic results): ONE instruction per element
    // (med3(x, y, +inf) = max(x, y); an intrinsic rather than an asm v_max_f32, so that the compiler sees the read of MFMA results)
    auto vmax = [](float x, float y) { return __builtin_amdgcn_fmed3f(x, y, __builtin_inff()); };

    f32x16 pnew, pold;
#pragma unroll
    for (int r = 0; r < 16; ++r) { pnew[r] = -__builtin_inff(); pold[r] = -__builtin_inff(); }

    __syncthreads();                                               // the window is in LDS; from here on the waves run free
    const int npx = ncols * pk;
    const uint32_t rstep = (uint32_t)(p.sh * WC * LDXB) * 2u, cstep = (uint32_t)(p.sw * LDXB) * 2u;
    // Conv pixels are walked column-major (the rows of a column, then the next column) and processed in PAIRS: the two pixels'
    // MFMA chains (five dependent instructions each) are issued interleaved, so neither waits on its own latency; the window
    // maxima are then updated in pixel order.  The next pair's operand reads are issued before the current pair's MFMAs.
    int cc = 0, rr = 0;                                            // position of the next pixel whose operands are to be fetched
    auto fetch_next = [&](bf16x8 (&b)[KS]) {                       // operand reads of the pixel at (cc, rr); advances the position
        load_b((uint32_t)rr * rstep + (uint32_t)cc * cstep, b);
        if (++rr == pk) { rr = 0; ++cc; }
    };
    int ucc = 0, urr = 0;                                          // position of the next pixel whose result is to be consumed
    auto update = [&](const f32x16& acc) {
        // windows of this column (see pconv_bf16_smallk_pool_kernel); wave-uniform BRANCHES: as selects the two window updates were
        // ~140 VALU instructions per pixel against five MFMAs, and the kernel was bound by them
        const int c = c0 + ucc;
        const int w_hi = c / ps;
        const bool hi_ok = w_hi < pw1;
        const bool starts = c == w_hi * ps;
        const bool lo_ok = (w_hi - 1) >= pw0 && c <= (w_hi - 1) * ps + pk - 1;
        // (wave-uniform branches here: this form's pixel is ~330 instructions, and skipping the update of a closed window pays --
        // measured against the unconditional updates of the strip form: 34.8 vs 45.9 us at four steps per launch)
        if (starts && urr == 0) { pold = pnew; asm volatile("" ::: "memory"); }
        if (hi_ok) {
            if (starts && urr == 0) {
                pnew = acc;
                asm volatile("" ::: "memory");
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) pnew[r] = vmax(pnew[r], acc[r]);
                asm volatile("" ::: "memory");
            }
        }
        if (lo_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pold[r] = vmax(pold[r], acc[r]);
            asm volatile("" ::: "memory");
        }
        const bool last_row = urr + 1 == pk;
        const bool emit_lo = last_row && lo_ok && c == (w_hi - 1) * ps + pk - 1;
        const bool emit_hi = last_row && !emit_lo && hi_ok && c == w_hi * ps + pk - 1;
        if (emit_lo || emit_hi) {
            const int ppix = ph * Wp + (emit_lo ? w_hi - 1 : w_hi);
            if constexpr (C8) {
                // channel-interleaved output (BBB_BF16_OUT_C8): 8 bytes per lane and channel quad, straight from the registers
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    uint32_t h[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = f2bf(bbb::apply_act((emit_lo ? pold[r4 * 4 + i] : pnew[r4 * 4 + i]) + bq[r4][i], p.act));
                    const int n = n0 + 8 * r4, b = b0 + wm + lrow;
                    const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)((((int64_t)(n >> 3) * HpWp + ppix) * p.B + b) * 16 + lk * 8) : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)}, yrs, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nl = 8 * r4 + 4 * lk;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float m = emit_lo ? pold[r4 * 4 + i] : pnew[r4 * 4 + i];
                        T[(nl + i) * TP + lrow] = f2bf(bbb::apply_act(m + bq[r4][i], p.act));
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int ps2 = 0; ps2 < 2; ++ps2) {
                    const int v = ps2 * 64 + lane;
                    const int row = v >> 2, grp = v & 3;
                    const u32x4 q = *reinterpret_cast<const u32x4*>(&T[row * TP + grp * 8]);
                    const int n = n0 + row, b = b0 + wm + grp * 8;
                    const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HpWp + ppix) * p.B + b) * 2u : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(q, yrs, off, 0, 0);
                }
            }
        }
        if (++urr == pk) { urr = 0; ++ucc; }
    };
    // one PAIR of pixels t, t + 1: both MFMA chains interleaved, then -- the matrix instructions have read their operands -- the next
    // pair's operand reads into the SAME two register sets (in flight under the window updates below), then the updates in pixel order
    bf16x8 bA[KS], bB[KS];
    fetch_next(bA);
    if (npx > 1) fetch_next(bB);
#pragma clang loop unroll(disable)
    for (int t = 0; t < npx; t += 2) {
        const bool two = t + 1 < npx;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        if (two) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], bA[kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], bB[kk], acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], bA[kk], acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < npx) fetch_next(bA);
        if (t + 3 < npx) fetch_next(bB);
        update(acc0);
        if (two) update(acc1);
    }
}

// ---- the strip form over CHANNEL-INTERLEAVED activations ("c8": [C / 8][H][W][B][8], BBB_BF16_X_C8) ----
// In the batch-innermost layout a lane's MFMA B operand -- 8 consecutive channels of one image -- is 8 elements a whole plane apart,
// which is why every other kernel here stages image rows through LDS and transposes them on the way out (ds_read_tr16_b64): measured
// on pconv_bf16_strip_kernel, the LDS writes, the transposing reads and the barriers between them are 51 of its 107 us, and they
// do not overlap with the MFMAs.  With 8 channels of an image adjacent in memory the operand IS a 16-byte load: lane (image, k half)
// reads channels 16 cib + 8 half .. + 7 of its image at the position, 32 images = 512 contiguous bytes.  No LDS, no transposes and
// no barriers on the image side at all; the waves of a workgroup (4 x 32 images, all 64 channels each) only share the tap row's
// weights, staged through LDS once per tap row.  Same MFMA sequence per output element: bit for bit the other forms' result.
template <int P, int KW, int DB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pconv_bf16_strip8_kernel(const PConvArgs p) {
    constexpr int CIN = 32, CIB = CIN / 16, BM = 128, TP = 32 + 8;
    constexpr int NC = P + KW - 1;                                // window columns of a strip
    constexpr int NCP = (NC + DB - 1) / DB * DB;                  // padded: the operand set of a position is c' % DB, statically
    constexpr int LDA = KW * CIN + 8;                             // weight row pitch (elements): conflict-free 16-byte reads
    constexpr int ASEG = KW * CIN / 8, APASS = (64 * ASEG + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    uint16_t* As = smem;                                          // [2][64][LDA]; the epilogue staging [4][64][TP] aliases it afterwards

    const int nstr = p.px_run;                                    // strips per output row
    const int bid = blockIdx.x, xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    const int64_t per_g = (int64_t)p.Ho * nstr * p.nbt;
    if (item >= item_end || item >= (int64_t)p.G * per_g) return;
    const int g = (int)(item / per_g);
    int rem = (int)(item - (int64_t)g * per_g);
    const int oh = rem / (nstr * p.nbt);
    rem -= oh * nstr * p.nbt;
    const int cs = rem / p.nbt;
    const int b0 = (rem - cs * p.nbt) * BM;
    const int ow0 = cs * P;
    const int npix = (p.Wo - ow0) < P ? (p.Wo - ow0) : P;
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * 64;
    const int Kp = p.Kp;
    const int HoWo = p.Ho * p.Wo;
    const int ihb = oh - p.ph, c0 = ow0 - p.pw;                   // stride 1, dilation 1
    const int r_lo = ihb < 0 ? -ihb : 0;
    const int r_hi = (p.H - ihb) < p.kh ? (p.H - ihb) : p.kh;
    uint32_t cmask = 0;                                           // window columns inside the image
#pragma unroll
    for (int c = 0; c < NC; ++c) cmask |= (c0 + c >= 0 && c0 + c < p.W) ? (1u << c) : 0u;
    auto colok = [&](int c) { return c < NC && ((cmask >> (c < NC ? c : 0)) & 1u) != 0; };

    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave * 32;
    const int lrow = lane & 31, lk = lane >> 5;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((int64_t)p.Cin * p.H * p.W * p.B * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)((int64_t)p.Cout * Kp * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : reinterpret_cast<const float*>(p.w)), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    char* yb = reinterpret_cast<char*>(p.y) + (int64_t)e * p.y_ds * 2;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((int64_t)p.Cout * HoWo * p.B * 2), 0x00020000);

    // MFMA B operands of a position, straight from memory: lane (image lrow, k half lk), channel block 2 cib + lk
    const uint32_t col_b = (uint32_t)p.B * 16u, row_b = (uint32_t)p.W * col_b, blk_b = (uint32_t)p.H * row_b;
    const uint32_t xbase = (uint32_t)lk * blk_b + (uint32_t)(b0 + wm + lrow) * 16u + (uint32_t)ihb * row_b + (uint32_t)c0 * col_b;   // (mod 2^32)
    bf16x8 bfr[DB][CIB];
    // (unconditional: a position outside the image or past the last tap row fetches out of range, i.e. zeros nobody multiplies --
    // with branches around the loads the compiler drains every outstanding load at every position)
    auto load_pos = [&](int set, int r, int c, bool ok) {
        const uint32_t o = xbase + (uint32_t)r * row_b + (uint32_t)c * col_b;
#pragma unroll
        for (int cib = 0; cib < CIB; ++cib)
            bfr[set][cib] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? o + (uint32_t)(2 * cib) * blk_b : kOOB, 0, 0));
    };
    // weights of a tap row: 64 rows of KW * CIN consecutive k (tap-major), 16-byte segments dealt out to the 256 threads
    u32x4 areg[APASS];
    auto load_aseg = [&](int i, int r, bool ok) {
        const int idx = tid + 256 * i;
        const int row = idx / ASEG, seg = idx - row * ASEG;
        const uint32_t off = (ok && idx < 64 * ASEG && n0 + row < p.Cout) ? (uint32_t)((n0 + row) * Kp + r * (KW * CIN) + seg * 8) * 2u : kOOB;
        areg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
    };
    auto store_aseg = [&](int i, int slot) {
        const int idx = tid + 256 * i;
        const int row = idx / ASEG, seg = idx - row * ASEG;
        if (idx < 64 * ASEG) *reinterpret_cast<u32x4*>(&As[slot * (64 * LDA) + row * LDA + seg * 8]) = areg[i];
    };

    f32x16 acc[P][2];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.0f;

    static_assert(NCP >= DB && NCP % DB == 0, "static operand sets per window column");
    static_assert(NCP >= APASS + 2, "a tap row's segments are requested and stored within one pass over the window");
#pragma unroll
    for (int k = 0; k < APASS; ++k) load_aseg(k, r_lo, true);
#pragma unroll
    for (int c = 0; c < DB; ++c) load_pos(c, r_lo, c, colok(c));
#pragma unroll
    for (int k = 0; k < APASS; ++k) store_aseg(k, 0);
    for (int r = r_lo; r < r_hi; ++r) {
        const int aslot = (r - r_lo) & 1;
        const bool more_r = r + 1 < r_hi;
        __syncthreads();                                          // this tap row's weights are in LDS (written during the row before)
        bf16x8 a[KW][CIB][2];
#pragma unroll
        for (int q = 0; q < KW; ++q)
#pragma unroll
            for (int cib = 0; cib < CIB; ++cib)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    a[q][cib][nt] = *reinterpret_cast<const bf16x8*>(&As[aslot * (64 * LDA) + (nt * 32 + lrow) * LDA + q * CIN + cib * 16 + lk * 8]);
#pragma unroll
        for (int c = 0; c < NCP; ++c) {
            // the next tap row's weights, a 16-byte segment per thread and position: requested at position c, in LDS two positions later
            if (c < APASS) load_aseg(c, r + 1, more_r);
            if (c >= 2 && c - 2 < APASS) store_aseg(c - 2, aslot ^ 1);
            if (colok(c)) {
#pragma unroll
                for (int cib = 0; cib < CIB; ++cib)
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        const int q = c - j;
                        if (q >= 0 && q < KW) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[j][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q][cib][nt], bfr[c % DB][cib], acc[j][nt], 0, 0, 0);
                        }
                    }
            }
            {   // the position DB ahead goes into the operand set just consumed
                const int cn = (c + DB) % NCP;
                const bool wrap = (c + DB) >= NCP;
                load_pos(c % DB, wrap ? r + 1 : r, cn, colok(cn) && (!wrap || more_r));
            }
        }
    }

    // ---- epilogue: per pixel, bias + activation + rounding in registers, wave-private LDS transpose, 16-byte stores ----
    __syncthreads();                                              // the weight rows are dead: their memory is the staging area now
    uint16_t* T = As + wave * (64 * TP);
    f32x4 bq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            bq[nt][r4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n0 + nt * 32 + 8 * r4 + 4 * lk) * 4u, 0, 0));
    if (p.y_c8) {
        // channel-interleaved output: a lane's four consecutive channels of an image are 8 adjacent bytes -- no transpose, and a
        // wave-store covers 32 images x 16 B = 512 contiguous bytes
        auto emit = [&](auto act) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                if (j < npix) {
                    const int pix = oh * p.Wo + ow0 + j;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            uint32_t lo = (uint32_t)f2bf(act(acc[j][nt][r4 * 4 + 0] + bq[nt][r4][0])) | ((uint32_t)f2bf(act(acc[j][nt][r4 * 4 + 1] + bq[nt][r4][1])) << 16);
                            uint32_t hi = (uint32_t)f2bf(act(acc[j][nt][r4 * 4 + 2] + bq[nt][r4][2])) | ((uint32_t)f2bf(act(acc[j][nt][r4 * 4 + 3] + bq[nt][r4][3])) << 16);
                            const int n = n0 + nt * 32 + 8 * r4, b = b0 + wm + lrow;
                            const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)((((int64_t)(n >> 3) * HoWo + pix) * p.B + b) * 16 + lk * 8) : kOOB;
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo, hi}, yrs, off, 0, 0);
                        }
                }
            }
        };
        if (p.act == 2)      emit([](float v) { return bbb::apply_act(v, 2); });
        else if (p.act == 1) emit([](float v) { return fmaxf(v, 0.0f); });
        else                 emit([](float v) { return v; });
        return;
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if (j < npix) {
            const int pix = oh * p.Wo + ow0 + j;
            auto stage_block = [&](auto act) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            T[(nt * 32 + 8 * r4 + 4 * lk + i) * TP + lrow] = f2bf(act(acc[j][nt][r4 * 4 + i] + bq[nt][r4][i]));
            };
            if (p.act == 2)      stage_block([](float v) { return bbb::apply_act(v, 2); });
            else if (p.act == 1) stage_block([](float v) { return fmaxf(v, 0.0f); });
            else                 stage_block([](float v) { return v; });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int v = ps * 64 + lane;
                const int row = v >> 2, grp = v & 3;
                const u32x4 q = *reinterpret_cast<const u32x4*>(&T[row * TP + grp * 8]);
                const int n = n0 + row, b = b0 + wm + grp * 8;
                const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 2u : kOOB;
                __builtin_amdgcn_raw_buffer_store_b128(q, yrs, off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the staging rows are read before the next pixel overwrites them
        }
    }
}

template <int P, int KW, int DB>
int launch_strip8(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    constexpr int kSmem = 2 * 64 * (KW * 32 + 8) * 2;
    static_assert(4 * 64 * 40 <= 2 * 64 * (KW * 32 + 8), "epilogue staging aliases the weight rows");
    hipLaunchKernelGGL((pconv_bf16_strip8_kernel<P, KW, DB>), dim3((unsigned)blocks), dim3(256), kSmem, st, a);
    return (int)hipGetLastError();
}

// ---- classifiers with a handful of outputs and a long row (3Conv3FC fc3: 1000 -> 10) ----
// On the MFMA kernel this layer is ONE 64-channel tile per 128 images with 10 of its 64 rows alive, walked as a serial chain of 64-k
// tiles by 2 .. 32 workgroups: 16.5 us per launch whatever the launch holds (0.2 % matrix-pipe busy).  It is 80 kflop per image --
// plain FMAs finish it in the time the chain needs for two of its tiles.  A workgroup takes 32 images of a draw: thread (k slice, 8
// images) walks its slice of the row -- one 16-byte image vector and NP weight values per k, fp32 fmaf in ascending k -- and the
// slices' partial sums are added in slice order through LDS: a fixed summation order, independent of the launch.
template <int NP>
__global__ __launch_bounds__(256) void pconv_bf16_fewout_kernel(const PConvArgs p) {
    constexpr int NSL = 64, TI = 32;                              // k slices x 4 groups of 8 images
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    float* red = reinterpret_cast<float*>(smem);                  // [NSL][NP][TI] partial sums
    const int nbt = p.nbt;
    const int64_t item = blockIdx.x;
    const int e = (int)(item / nbt);
    const int b0 = (int)(item - (int64_t)e * nbt) * TI;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int Kp = p.Kp;
    const int tid = (int)threadIdx.x, grp = tid & 3, sl = tid >> 2;
    const int kps = p.px_run;                                     // k per slice (a multiple of 8)
    const int k0 = sl * kps;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    const uint16_t* xb = reinterpret_cast<const uint16_t*>(p.x) + (int64_t)ex * p.x_ds;
    const uint16_t* wb = reinterpret_cast<const uint16_t*>(p.w) + (int64_t)ew * p.w_ds;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(xb), 0, (int)((int64_t)p.K * p.B * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wb), 0, (int)((int64_t)p.Cout * Kp * 2), 0x00020000);

    float acc[NP][8];
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[n][i] = 0.0f;
    const int b = b0 + grp * 8;
    auto bf = [](uint32_t v, int hi) { return __builtin_bit_cast(float, hi ? (v & 0xFFFF0000u) : (v << 16)); };
    for (int kc = 0; kc < kps; kc += 8) {
        const int k = k0 + kc;                                    // 8 consecutive k: one 16-byte weight vector per output row
        u32x4 wv[NP], xv[8];
#pragma unroll
        for (int n = 0; n < NP; ++n)
            wv[n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (n < p.Cout && k < Kp) ? (uint32_t)(n * Kp + k) * 2u : kOOB, 0, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            xv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (k + j < p.K && b < p.B) ? (uint32_t)((k + j) * p.B + b) * 2u : kOOB, 0, 0));
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const float w = bf(wv[n][j >> 1], j & 1);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[n][i] = __builtin_fmaf(w, bf(xv[j][i >> 1], i & 1), acc[n][i]);
            }
    }
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(sl * NP + n) * TI + grp * 8 + i] = acc[n][i];
    __syncthreads();
    const int nsl = p.G;                                          // slices that hold any k
    const int HoWo = 1;
    for (int o = tid; o < p.Cout * TI; o += 256) {
        const int n = o / TI, i = o - n * TI;
        float s = red[n * TI + i];
        for (int q = 1; q < nsl; ++q) s += red[(q * NP + n) * TI + i];
        const float bv = p.bias ? p.bias[(int64_t)ew * p.b_ds + n] : 0.0f;
        const float v = bbb::apply_act(s + bv, p.act);
        const int bb = b0 + i;
        if (bb < p.B) {
            if (p.y_f32) reinterpret_cast<float*>(p.y)[(int64_t)e * p.y_ds + (int64_t)n * p.B + bb] = v;                       // fp32 output
            else reinterpret_cast<uint16_t*>(p.y)[(int64_t)e * p.y_ds + (int64_t)n * p.B + bb] = f2bf(v);
        }
    }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `state` (one array per kernel instantiation)
// remembers the bytes granted on each device, so that a process driving several GPUs -- or several threads (relaxed atomics: a
// duplicated call is harmless) -- sets it wherever a launch needs it.
constexpr int kMaxDevices = 64;
struct SmemAttrState { std::atomic<int> bytes[kMaxDevices]; };
inline int ensure_dynamic_smem(const void* fn, int bytes, SmemAttrState& state) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);       // (unknown device: every time)
    }
    if (state.bytes[dev].load(std::memory_order_relaxed) >= bytes) return 0;
    const hipError_t er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (er != hipSuccess) return (int)er;
    state.bytes[dev].store(bytes, std::memory_order_relaxed);
    return 0;
}

template <int NP>
int launch_fewout(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    constexpr int kSmem = 64 * NP * 32 * 4;
    static SmemAttrState attr_state;
    if (const int rc = ensure_dynamic_smem(reinterpret_cast<const void*>(&pconv_bf16_fewout_kernel<NP>), kSmem, attr_state)) return rc;
    hipLaunchKernelGGL((pconv_bf16_fewout_kernel<NP>), dim3((unsigned)blocks), dim3(256), kSmem, st, a);
    return (int)hipGetLastError();
}

template <int KS, bool C8>
int launch_smallk_poolwin_c(const PConvArgs& a, int64_t blocks, int smem_bytes, hipStream_t st) {
    static SmemAttrState attr_state;
    if (const int rc = ensure_dynamic_smem(reinterpret_cast<const void*>(&pconv_bf16_smallk_poolwin_kernel<KS, C8>), smem_bytes, attr_state)) return rc;
    hipLaunchKernelGGL((pconv_bf16_smallk_poolwin_kernel<KS, C8>), dim3((unsigned)blocks), dim3(256), smem_bytes, st, a);
    return (int)hipGetLastError();
}

// (the output layout is a template parameter: with both epilogues in one kernel the strip form needs 241 registers instead of 213 and
// loses its second wave per SIMD -- 101 -> 153 us on 3Conv3FC conv1 + pool1 at 16 steps per launch)
template <int KS>
int launch_smallk_poolwin(const PConvArgs& a, int64_t blocks, int smem_bytes, hipStream_t st) {
    return a.y_c8 ? launch_smallk_poolwin_c<KS, true>(a, blocks, smem_bytes, st) : launch_smallk_poolwin_c<KS, false>(a, blocks, smem_bytes, st);
}

template <int NT, int KS, bool C8>
int launch_smallk_pool_c(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    constexpr int kSmem = (KS * 16 * (256 + 32) + 4 * 32 * NT * 72) * 2 + 2 * 3 * KS * 16 * 4;
    static_assert(kSmem <= 160 * 1024, "LDS");
    static SmemAttrState attr_state;
    if (const int rc = ensure_dynamic_smem(reinterpret_cast<const void*>(&pconv_bf16_smallk_pool_kernel<NT, KS, C8>), kSmem, attr_state)) return rc;
    hipLaunchKernelGGL((pconv_bf16_smallk_pool_kernel<NT, KS, C8>), dim3((unsigned)blocks), dim3(256), kSmem, st, a);
    return (int)hipGetLastError();
}

template <int NT, int KS>
int launch_smallk_pool(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    return a.y_c8 ? launch_smallk_pool_c<NT, KS, true>(a, blocks, st) : launch_smallk_pool_c<NT, KS, false>(a, blocks, st);
}

template <int NT, int KS>
int launch_smallk(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    constexpr int kSmem = (KS * 16 * (256 + 32) + 4 * 32 * NT * 72) * 2 + 16 * KS * 16 * 4;      // px_run <= 16
    static_assert(kSmem <= 160 * 1024, "LDS");
    static SmemAttrState attr_state;
    if (const int rc = ensure_dynamic_smem(reinterpret_cast<const void*>(&pconv_bf16_smallk_kernel<NT, KS>), kSmem, attr_state)) return rc;
    hipLaunchKernelGGL((pconv_bf16_smallk_kernel<NT, KS>), dim3((unsigned)blocks), dim3(256), kSmem, st, a);
    return (int)hipGetLastError();
}

template <bool OUT_F32, int WN, int WM, int KG, bool WS>
int launch_cfg(const PConvArgs& a, int64_t blocks, hipStream_t st) {
    constexpr int kStageB = (BK * (64 * WM + 32) + 64 * WN * LDWB) * 2;
    constexpr int kSmem = KG * (WS ? 2 : 1) * kStageB + 4 * KCH * KG * 4;
    constexpr int kRed = (KG - 1) * 64 * (64 * WN * WM) * 4;
    static_assert(kRed <= KG * kStageB, "reduction buffer must fit in the stage memory");
    static_assert(kSmem <= 160 * 1024, "LDS");
    static_assert(WN * WM * 64 * 72 * 2 <= kStageB, "epilogue staging must fit in one stage");
    static SmemAttrState attr_state;
    if (const int rc = ensure_dynamic_smem(reinterpret_cast<const void*>(&pconv_bf16_kernel<OUT_F32, WN, WM, KG, WS>), kSmem, attr_state)) return rc;
    hipLaunchKernelGGL((pconv_bf16_kernel<OUT_F32, WN, WM, KG, WS>), dim3((unsigned)blocks),
                       dim3(64 * WN * WM * KG + (WS ? 256 : 0)), kSmem, st, a);
    return (int)hipGetLastError();
}

template <bool OUT_F32>
int launch_shape(const PConvArgs& a, int shape, int kgs, bool ws, int64_t blocks, hipStream_t st) {
    if (shape == 22) {
        if (ws) return launch_cfg<OUT_F32, 2, 2, 1, true>(a, blocks, st);
        return kgs == 2 ? launch_cfg<OUT_F32, 2, 2, 2, false>(a, blocks, st) : launch_cfg<OUT_F32, 2, 2, 1, false>(a, blocks, st);
    }
    if (shape == 14) {
        if (ws) return launch_cfg<OUT_F32, 1, 4, 1, true>(a, blocks, st);
        return kgs == 2 ? launch_cfg<OUT_F32, 1, 4, 2, false>(a, blocks, st) : launch_cfg<OUT_F32, 1, 4, 1, false>(a, blocks, st);
    }
    if (ws) return launch_cfg<OUT_F32, 1, 2, 1, true>(a, blocks, st);
    if (kgs == 4) return launch_cfg<OUT_F32, 1, 2, 4, false>(a, blocks, st);
    return kgs == 2 ? launch_cfg<OUT_F32, 1, 2, 2, false>(a, blocks, st) : launch_cfg<OUT_F32, 1, 2, 1, false>(a, blocks, st);
}

// maxpool over [planes][H][W][B] bf16, 8 images per thread (bf16 order = fp32 order of the widened values: exact)
__global__ __launch_bounds__(256) void maxpool_chwn_bf16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, int64_t total8,
                                                                int H, int W, int Ho, int Wo, int B8, int k, int s) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int b8 = (int)(i % B8);
    int64_t t = i / B8;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int64_t pl = t / Ho;
    const u32x4* xp = x + ((pl * H + (int64_t)oh * s) * W + (int64_t)ow * s) * B8 + b8;
    float m[8];
    {
        const u32x4 v = xp[0];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m[2 * u] = __builtin_bit_cast(float, v[u] << 16);
            m[2 * u + 1] = __builtin_bit_cast(float, v[u] & 0xFFFF0000u);
        }
    }
    for (int a = 0; a < k; ++a)
        for (int c = 0; c < k; ++c) {
            const u32x4 v = xp[((int64_t)a * W + c) * B8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                m[2 * u] = fmaxf(m[2 * u], __builtin_bit_cast(float, v[u] << 16));
                m[2 * u + 1] = fmaxf(m[2 * u + 1], __builtin_bit_cast(float, v[u] & 0xFFFF0000u));
            }
        }
    u32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        o[u] = (__builtin_bit_cast(uint32_t, m[2 * u]) >> 16) | (__builtin_bit_cast(uint32_t, m[2 * u + 1]) & 0xFFFF0000u);
    y[i] = o;
}

// fp32 NCHW [B][C][H][W] -> bf16 [C][H][W][B]: 32x32 tile transpose through LDS (planes = C*H*W)
__global__ __launch_bounds__(256) void nchw_to_chwn_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int B, int P) {
    __shared__ float tile[32][33];
    x += (int64_t)blockIdx.z * B * P;                 // batch slice blockIdx.z: its own [B][P] -> [P][B] block
    y += (int64_t)blockIdx.z * B * P;
    const int p0 = blockIdx.x * 32, bb0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int bi = bb0 + r, pi = p0 + tx;
        tile[r][tx] = (bi < B && pi < P) ? x[(int64_t)bi * P + pi] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int pi = p0 + r, bi = bb0 + tx;
        if (pi < P && bi < B) y[(int64_t)pi * B + bi] = f2bf(tile[tx][r]);
    }
}

}  // namespace

extern "C" int bbb_conv2d_chwn_bf16_fwd(const bbb_conv_desc_t* d, const void* x, const void* w, const float* bias, void* y,
                                        uint32_t flags, void* stream) {
    const int out_f32 = (flags & BBB_BF16_OUT_F32) ? 1 : 0;
    const bool tap_major = (flags & BBB_BF16_W_TAP_MAJOR) != 0;
    if (d == nullptr || x == nullptr || w == nullptr || y == nullptr) return BBB_EINVAL;
    if (d->batch <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 ||
        d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0 ||
        d->draws <= 0 || d->act < 0 || d->act > 2)
        return BBB_EINVAL;
    if (d->batch % 8 != 0) return BBB_ESHAPE;        // rows of 16-byte vectors of 8 bf16 images
    if (tap_major && d->cin % 8 != 0) return BBB_ESHAPE;   // a 16-byte weight vector must not straddle two taps
    if ((flags & ~(BBB_BF16_OUT_F32 | BBB_BF16_W_TAP_MAJOR | BBB_BF16_X_C8 | BBB_BF16_OUT_C8)) != 0) return BBB_EINVAL;
    const bool x_c8 = (flags & BBB_BF16_X_C8) != 0, out_c8 = (flags & BBB_BF16_OUT_C8) != 0;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    const int64_t K = (int64_t)d->cin * d->kh * d->kw;
    const int64_t Kp = (K + 7) & ~(int64_t)7;
    if (K >= (1 << 24)) return BBB_ESHAPE;           // float-reciprocal k decode is exact below 2^24
    if ((int64_t)d->cin * d->h * d->w * d->batch * 2 > 0xFFFE0000LL || (int64_t)d->cout * ho * wo * d->batch * 4 > 0xFFFE0000LL ||
        ((int64_t)d->cout + 64) * Kp * 2 > 0x7FFFFFFFLL || (int64_t)d->batch * 2 > 0x0FFFFFFFLL)
        return BBB_ESHAPE;
    const uint32_t x_inv = (0xFFFFFFF0u - ((uint32_t)d->batch + 512u) * 2u) & ~15u;   // a ragged last tile reaches < 512 columns past the row
    if ((int64_t)d->cin * d->h * d->w * d->batch * 2 > (int64_t)x_inv) return BBB_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)w) & 15u) != 0 || ((uintptr_t)y & (out_f32 ? 3u : 1u)) != 0 || ((uintptr_t)bias & 3u) != 0)
        return BBB_EALIGN;
    if ((d->x_draw_stride & 7) != 0 || (d->w_draw_stride & 7) != 0) return BBB_EALIGN;
    PConvArgs a = {};
    a.B = d->batch; a.Cin = d->cin; a.H = d->h; a.W = d->w; a.Cout = d->cout; a.kh = d->kh; a.kw = d->kw;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = ho; a.Wo = wo; a.K = (int32_t)K; a.Kp = (int32_t)Kp; a.khkw = d->kh * d->kw; a.act = d->act;
    a.x_ds = d->x_draw_stride; a.w_ds = d->w_draw_stride; a.b_ds = d->b_draw_stride;
    a.y_ds = (int64_t)d->cout * ho * wo * d->batch;
    a.x = reinterpret_cast<const float*>(x); a.w = reinterpret_cast<const float*>(w); a.bias = bias;
    a.y = reinterpret_cast<float*>(y);
    a.x_inv = x_inv;
    a.wtap = tap_major ? 1 : 0;
    if (d->unit_div < 0 || d->unit_off < 0 || d->x_unit_mod < 0 || (d->unit_div > 1 && d->unit_off >= d->unit_div) ||
        (d->x_unit_mod > 0 && d->x_unit_mod != d->unit_div) || d->w_row_pitch != 0 || d->w_tap_major != 0)
        return BBB_EINVAL;
    // pooling in the launch (bbb_conv_desc_t::pool): first layers with a short contraction only (pconv_bf16_smallk_pool_kernel);
    // 1 = MaxPool2d(2, 2), (k << 8) | s otherwise; admitted: 2 / 2 and 3 / 2 (at most one window closes per conv column)
    int pool_k = 0, pool_s = 0;
    if (d->pool != 0) {
        pool_k = d->pool == 1 ? 2 : (d->pool >> 8);
        pool_s = d->pool == 1 ? 2 : (d->pool & 255);
        if (!((pool_k == 2 && pool_s == 2) || (pool_k == 3 && pool_s == 2)) || ho < pool_k || wo < pool_k) return BBB_EINVAL;
        if (tap_major || out_f32 || Kp > 128) return BBB_EINVAL;
        if (x_c8 || (out_c8 && d->cout % 8 != 0)) return BBB_EINVAL;
        a.y_c8 = out_c8 ? 1 : 0;
    }
    if (d->x_unit_div < 0 || d->x_unit_off < 0 || (d->x_unit_div > 1 && (d->unit_div > 1 || d->x_unit_off >= d->x_unit_div)) ||
        (d->x_unit_div <= 1 && d->x_unit_off != 0))
        return BBB_EINVAL;
    a.x_div = d->x_unit_div; a.x_off = d->x_unit_off;
    a.unit_div = d->unit_div; a.unit_off = d->unit_div > 1 ? d->unit_off : 0; a.x_mod = d->x_unit_mod;
    if (pool_k != 0) {
        const int nt = a.Cout <= 32 ? 1 : 2;
        const int ks = Kp <= 32 ? 2 : (Kp <= 80 ? 5 : 8);
        a.Ntiles = (a.Cout + 32 * nt - 1) / (32 * nt);
        a.G = a.Ntiles * d->draws;
        a.nbt = (a.B + 255) / 256;
        a.pool = (pool_k << 8) | pool_s;
        const int hp = (ho - pool_k) / pool_s + 1, wp = (wo - pool_k) / pool_s + 1;
        a.y_ds = (int64_t)d->cout * hp * wp * d->batch;
        // the window-resident form (pconv_bf16_smallk_poolwin_kernel): <= 32 channels, 128-image tiles, the widest strip whose input
        // window (+ one zero row) and epilogue staging fit 72 KB of LDS (two workgroups per CU).  Its workgroups are short (one
        // memory round trip for the window, 15 pixels, two pooled outputs: ~8 us each), so it wins where the strip form is a
        // latency chain -- small launches -- and loses once the strip form fills the chip (3Conv3FC conv1 + pool1, bs 256, us per
        // launch, window / strip / conv + pool launches: 1 step 15.8 / 38.8 / 18.4, 4 steps 34.8 / 52.8 / 46.5, 5-6 steps 45.8 / 44.2 /
        // 82, 16 steps 128 / 101 / 167; profiles/r05_notes.md section 3): below 70 pooled rows x image tiles
        if (nt == 1 && (int64_t)a.G * a.nbt * hp < 70) {
            const int WRw = (pool_k - 1) * a.sh + (a.kh - 1) * a.dh + 1;
            int best_n = 0;
            for (int n = 1; n <= wp; ++n) {
                const int cols = (n - 1) * pool_s + pool_k;
                const int WCw = (cols - 1) * a.sw + (a.kw - 1) * a.dw + 1;
                const int64_t bytes = ((int64_t)a.Cin * WRw * WCw + 1) * (128 + 32) * 2 + 4 * 32 * 40 * 2;
                if (bytes <= 72 * 1024 && a.Cin * WRw * WCw + 1 <= kWinPasses * 16) best_n = n;
            }
            if (best_n >= 1) {
                const int nstr = (wp + best_n - 1) / best_n;
                const int wpc = (wp + nstr - 1) / nstr;               // what the kernel derives from nstr
                const int cols = (wpc - 1) * pool_s + pool_k;
                const int WCw = (cols - 1) * a.sw + (a.kw - 1) * a.dw + 1;
                const int rows = a.Cin * WRw * WCw + 1;
                a.Mtiles = rows;
                a.px_run = nstr;
                a.nbt = (a.B + 127) / 128;
                const int64_t itemsw = (int64_t)a.G * a.nbt * hp * nstr;
                const int64_t perw = (itemsw + 7) / 8;
                if (8 * perw > 0x7fffffffLL) return BBB_ESHAPE;
                a.per_xcd = (int32_t)perw;
                const int smem_bytes = rows * (128 + 32) * 2 + 4 * 32 * 40 * 2;
                hipStream_t stw = (hipStream_t)stream;
                return ks == 2 ? launch_smallk_poolwin<2>(a, 8 * perw, smem_bytes, stw)
                     : ks == 5 ? launch_smallk_poolwin<5>(a, 8 * perw, smem_bytes, stw) : launch_smallk_poolwin<8>(a, 8 * perw, smem_bytes, stw);
            }
        }
        // workgroups per pooled row.  A strip of n pooled pixels walks n * ps + pk - ps conv columns, so narrow strips compute
        // shared columns twice, wide strips leave the chip short of workgroups: take the split whose launch costs least in
        // (rounds of resident workgroups: 2 per CU at this kernel's register footprint) x (columns of its widest strip).
        // 3Conv3FC conv1, 16 steps per launch (240 pooled rows): 2 strips of 8 / 7 pooled pixels = 480 workgroups, one round, 17
        // columns (3 strips = 720 workgroups = two rounds of 11: measured 128 us against 1xx, profiles/r05_notes.md section 3)
        const int64_t rows = (int64_t)a.G * a.nbt * hp;
        const int64_t slots = 512;
        int csplit = 1;
        int64_t best = -1;
        const int max_split = wp >= 2 ? wp / 2 : 1;
        for (int c = 1; c <= max_split; ++c) {
            const int wpc = (wp + c - 1) / c;
            const int used = (wp + wpc - 1) / wpc;                // strips that actually hold pixels
            const int64_t rounds = (rows * used + slots - 1) / slots;
            const int64_t cost = rounds * (wpc * pool_s + pool_k - pool_s);
            if (best < 0 || cost < best) { best = cost; csplit = c; }
        }
        a.px_run = csplit;
        const int64_t items = rows * csplit;
        const int64_t per = (items + 7) / 8;
        if (8 * per > 0x7fffffffLL) return BBB_ESHAPE;
        a.per_xcd = (int32_t)per;
        hipStream_t st = (hipStream_t)stream;
        if (nt == 1) return ks == 2 ? launch_smallk_pool<1, 2>(a, 8 * per, st) : ks == 5 ? launch_smallk_pool<1, 5>(a, 8 * per, st) : launch_smallk_pool<1, 8>(a, 8 * per, st);
        return ks == 2 ? launch_smallk_pool<2, 2>(a, 8 * per, st) : ks == 5 ? launch_smallk_pool<2, 5>(a, 8 * per, st) : launch_smallk_pool<2, 8>(a, 8 * per, st);
    }
    if (x_c8) {
        // channel-interleaved input: the strip form that reads its MFMA operands straight from memory (pconv_bf16_strip8_kernel).
        // Tap-major rows of 32 input channels, 5 x 5 taps, stride 1, no dilation (3Conv3FC conv2); bf16 output in either layout.
        if (!(tap_major && !out_f32 && (!out_c8 || a.Cout % 8 == 0) && a.Cin == 32 && a.kh == 5 && a.kw == 5 && a.sh == 1 && a.sw == 1 &&
              a.dh == 1 && a.dw == 1 && a.pw < a.kw && a.ph < a.kh))
            return BBB_EINVAL;
        constexpr int P8 = 3;
        a.y_c8 = out_c8 ? 1 : 0;
        a.Ntiles = (a.Cout + 63) / 64;
        a.G = a.Ntiles * d->draws;
        a.nbt = (a.B + 127) / 128;
        a.px_run = (wo + P8 - 1) / P8;
        const int64_t sitems = (int64_t)a.G * ho * a.px_run * a.nbt;
        const int64_t sper = (sitems + 7) / 8;
        if (8 * sper > 0x7fffffffLL) return BBB_ESHAPE;
        a.per_xcd = (int32_t)sper;
        return launch_strip8<P8, 5, 4>(a, 8 * sper, (hipStream_t)stream);
    }
    if (out_c8) return BBB_EINVAL;                    // only the pooled first-layer forms and the strip form write that layout
    if (!tap_major && !out_f32 && Kp <= 128 && (int64_t)ho * wo >= 16) {
        // a first layer with a short contraction: weights in registers, a run of pixels per workgroup (pconv_bf16_smallk_kernel)
        const int nt = a.Cout <= 32 ? 1 : 2;
        const int ks = Kp <= 32 ? 2 : (Kp <= 80 ? 5 : 8);
        a.Ntiles = (a.Cout + 32 * nt - 1) / (32 * nt);
        a.G = a.Ntiles * d->draws;
        a.nbt = (a.B + 255) / 256;
        const int64_t npix = (int64_t)a.G * a.nbt * ho * wo;      // (pixel, channel tile, image tile) units of the launch
        int run = (int)(npix / 512);                              // >= 512 workgroups (2 per CU) before runs get longer
        run = run < 1 ? 1 : (run > 16 ? 16 : run);
        a.px_run = run;
        const int64_t items = (int64_t)a.G * a.nbt * (((int64_t)ho * wo + run - 1) / run);
        const int64_t per = (items + 7) / 8;
        if (8 * per > 0x7fffffffLL) return BBB_ESHAPE;
        a.per_xcd = (int32_t)per;
        hipStream_t st = (hipStream_t)stream;
        if (nt == 1) return ks == 2 ? launch_smallk<1, 2>(a, 8 * per, st) : ks == 5 ? launch_smallk<1, 5>(a, 8 * per, st) : launch_smallk<1, 8>(a, 8 * per, st);
        return ks == 2 ? launch_smallk<2, 2>(a, 8 * per, st) : ks == 5 ? launch_smallk<2, 5>(a, 8 * per, st) : launch_smallk<2, 8>(a, 8 * per, st);
    }
    if (a.Cout <= 16 && K >= 512 && a.kh == 1 && a.kw == 1 && a.H == 1 && a.W == 1 && a.ph == 0 && a.pw == 0) {
        // a classifier with a handful of outputs and a long row: plain FMAs, k slices summed in a fixed order (pconv_bf16_fewout_kernel)
        const int kps = (int)(((Kp + 63) / 64 + 7) / 8) * 8;
        a.px_run = kps;
        a.G = (int)((Kp + kps - 1) / kps);                // slices that hold any k (<= 64)
        a.nbt = (a.B + 31) / 32;
        a.y_f32 = out_f32 ? 1 : 0;
        const int64_t fitems = (int64_t)d->draws * a.nbt;
        if (fitems > 0x7fffffffLL) return BBB_ESHAPE;
        hipStream_t fst = (hipStream_t)stream;
        return a.Cout <= 10 ? launch_fewout<10>(a, fitems, fst) : launch_fewout<16>(a, fitems, fst);
    }
    // tile shape: LDS-pipe cycles per unit of useful work (see the kernel comment), including the waste of ragged
    // channel / image tiles: 128x128 -> 256, 64x256 -> 288, 64x128 (two waves) -> 320
    auto waste = [](int n, int t) { return (double)(((n + t - 1) / t) * t) / (double)n; };
    const double c22 = 256.0 * waste(a.Cout, 128) * waste(a.B, 128);
    const double c14 = 288.0 * waste(a.Cout, 64) * waste(a.B, 256);
    const double c12 = 320.0 * waste(a.Cout, 64) * waste(a.B, 128);
    int shape = (c22 <= c14 && c22 <= c12) ? 22 : (c14 <= c12 ? 14 : 12);
    // Few workgroups with long rows (a 1000 -> 10 classifier: ONE 64 x 256 tile per draw, 16 tiles of k; AlexNet's conv3-5 at one
    // draw) are nothing but their serial k loops: take the two-wave 64 x 128 shape, whose small stage leaves room for FOUR
    // k-groups per workgroup, each with its own stage and loads in flight (measured, profiles/r03_notes.md section 10: fc3 of
    // 3Conv3FC 20.7 -> 16.5 us, fc2 15.7 -> 14.6, AlexNet one draw conv2 19.0 -> 15.6, conv4 20.3 -> 18.4, conv5 15.4 -> 14.2).
    const int t64_all = (int)((K + BK - 1) / BK);
    const int64_t items12 = (int64_t)d->draws * ho * wo * ((a.Cout + 63) / 64) * ((a.B + 127) / 128);
    const bool tiny = t64_all >= 16 && items12 < 256;
    if (tiny) shape = 12;
    const int bn = shape == 22 ? 128 : 64, bm = shape == 14 ? 256 : 128;
    a.Ntiles = (a.Cout + bn - 1) / bn;
    a.G = a.Ntiles * d->draws;
    a.nbt = (a.B + bm - 1) / bm;
    const int64_t mt = (int64_t)ho * wo * a.nbt;
    if (mt > 0x7fffffffLL) return BBB_ESHAPE;
    a.Mtiles = (int)mt;
    const int64_t items = (int64_t)a.G * mt;
    const int64_t per = (items + 7) / 8;
    const int64_t blocks = 8 * per;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    a.per_xcd = (int32_t)per;
    // a second k-group (its own stage, its own loads in flight, deterministic LDS reduction) when the launch cannot
    // fill the chip with workgroups and the k loop is long: then per-tile latency, not LDS throughput, sets the pace
    const int t64 = (int)((K + BK - 1) / BK);
    int kgs = (items < 512 && t64 >= 8) ? 2 : 1;
    if (tiny) kgs = 4;
    // wave specialisation pays when few workgroups are resident per CU (nothing else hides the staging phases); measured
    // on AlexNet bs=512 E=10: conv3 31.7 -> 23.2 us, conv4 46.5 -> 33.3, conv5 18.2 -> 16.8, but conv1 / conv2 (1280+
    // workgroups, or the 64x256 shape whose two stages leave one workgroup per CU) 20-30 % slower
    bool ws = shape == 22 && items <= 1024;
    if (ws) kgs = 1;
    hipStream_t st = (hipStream_t)stream;
    return out_f32 ? launch_shape<true>(a, shape, kgs, ws, blocks, st) : launch_shape<false>(a, shape, kgs, ws, blocks, st);
}

extern "C" int bbb_maxpool_chwn_bf16(const void* x, void* y, int64_t planes, int h, int w, int batch, int k, int s, void* stream) {
    if (x == nullptr || y == nullptr || planes <= 0 || h <= 0 || w <= 0 || batch <= 0 || k <= 0 || s <= 0) return BBB_EINVAL;
    if (batch % 8 != 0 || h < k || w < k) return BBB_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0) return BBB_EALIGN;
    const int ho = (h - k) / s + 1, wo = (w - k) / s + 1;
    const int64_t total8 = planes * ho * wo * (batch / 8);
    const int64_t blocks = (total8 + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(maxpool_chwn_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u32x4*>(x), reinterpret_cast<u32x4*>(y), total8, h, w, ho, wo, batch / 8, k, s);
    return (int)hipGetLastError();
}

extern "C" int bbb_nchw_to_chwn_bf16(const float* x, void* y, int batch, int64_t plane, void* stream) {
    return bbb_nchw_to_chwn_bf16_slices(x, y, batch, plane, 1, stream);
}

extern "C" int bbb_nchw_to_chwn_bf16_slices(const float* x, void* y, int batch, int64_t plane, int slices, void* stream) {
    if (x == nullptr || y == nullptr || batch <= 0 || plane <= 0 || plane > 0x7fffffffLL || slices <= 0 || slices > 65535) return BBB_EINVAL;
    if (((uintptr_t)x & 3u) != 0 || ((uintptr_t)y & 1u) != 0) return BBB_EALIGN;
    const dim3 grid((unsigned)((plane + 31) / 32), (unsigned)((batch + 31) / 32), (unsigned)slices);
    if (grid.y > 65535u) return BBB_ESHAPE;
    hipLaunchKernelGGL(nchw_to_chwn_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<uint16_t*>(y), batch,
                       (int)plane);
    return (int)hipGetLastError();
}
