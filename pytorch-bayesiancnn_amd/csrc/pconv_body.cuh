// The work-item body of the pixel-major fp32 implicit GEMM (see pconv_gemm.hip for the algorithm): one device function shared
// by the plain and the split-contraction kernels, so that every form runs the SAME instruction sequence per output tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bbb_common.cuh"
#include "pconv_args.h"

namespace pconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int BN = 64;
constexpr int BK = 32;
constexpr int LDW = BN + 1;
constexpr int KCH = 256;                 // k_eff entries per decode chunk (one entry per thread)
constexpr int TPC = KCH / BK;            // tiles per chunk

// One work item = one (draw, 64-channel tile, output pixel, BM-image tile) of a pixel-major GEMM: the whole body of
// pconv_gemm_kernel.  `item` indexes the launch's items in (draw, channel tile)-major order, as the XCD-aware block mapping of
// the launcher enumerates them.
// SPLIT: the item's contraction is cut into p.ksplit consecutive ranges of k tiles, one per workgroup (`ks` = this
// workgroup's range) -- launches of a few dozen to a few hundred items cannot fill 256 CUs, and a lone workgroup is bounded by
// its serial k loop (conv4 at one draw: 48 tiles on 128 of the 256 CUs).  Every workgroup writes its raw accumulator tile(s) to
// scratch (write-through, 16 bytes per lane), takes a ticket, and the LAST arriver adds the ksplit partial tiles in range
// order 0, 1, ... (whoever it is: the sum does not depend on timing), applies the epilogue and stores the output.  The
// partial sums round differently from the unsplit fmaf chain: results agree with an unsplit launch to ~1e-7 relative.
// SEQ (MODE 2): the SAME summation order computed by ONE workgroup -- it walks the p.ksplit ranges one after the other with a
// fresh accumulator per range (the k loop itself does not change: at a range boundary the accumulator is added to a running
// total and zeroed), total = ((p0 + p1) + p2) + ... in range order: bit for bit what the SPLIT form's last arriver computes,
// without scratch, tickets or partial-tile traffic.  This is what makes the split a property of the LAYER (the plan depends on
// its geometry only): launches too large to profit from a cross-workgroup split run SEQ, small ones SPLIT, same bits -- a draw
// computed alone, inside a 10-draw launch, as a work unit of a sharded step or as one of G steps per launch is the same number.
// POOL (MODE 3): the layer is followed by [activation ->] MaxPool2d(2, 2): the item is a POOLED pixel and walks the four
// conv pixels of its window one after the other -- per pixel the unchanged k loop from a zeroed accumulator, then
// running_max = max(running_max, act(acc + bias)) in accumulation registers -- and stores one pooled row instead of four: the
// pooling launch and three quarters of the output traffic disappear, the matrix work is exactly the unfused launch's (no tap is
// multiplied that the unfused launch does not multiply), and the result is bit for bit pool(act(conv + bias)) of the separate
// launches (max is exact).  Items are four times fewer and four times longer: for launches that still fill the chip evenly.
constexpr int kPlain = 0, kSplit = 1, kSeq = 2, kPool = 3;

// SEQ keeps its running totals in ACCUMULATION registers through inline asm: they are touched three times per item, and left to
// the register allocator they became 28-55 extra VGPRs (one resident workgroup less per CU: +5-8 % per launch, measured);
// gfx950's register file is unified, so 16 AGPRs are just 16 registers the k loop never looks at.
__device__ __forceinline__ float agpr_read(float a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
__device__ __forceinline__ void agpr_write(float& a, float v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }
template <int BM, bool LRT, bool ILV, int MODE = kPlain>
__device__ __forceinline__ void pconv_item(const PConvArgs& p, const int64_t item, const int ks = 0) {
    constexpr bool SPLIT = MODE == kSplit;
    constexpr bool SEQ = MODE == kSeq;
    constexpr bool POOL = MODE == kPool;
    static_assert(!POOL || !LRT || BM == 64, "the pooled LRT form: 64-image tiles (one accumulator pair per wave)");
    constexpr int LDX = BM + 4;
    constexpr int NT = (BM >= 128) ? 2 : 1;              // 32-channel MFMA tiles per wave
    constexpr int MT = (BM == 256) ? 2 : 1;              // 32-image MFMA tiles per wave
    constexpr int WSETS = LRT ? 2 : 1;
    constexpr int XL = BM / 4;                 // lanes per X row (float4 each)
    constexpr int XRPP = kThreads / XL;        // X rows per pass
    constexpr int XPASS = BK / XRPP;

    // ONE LDS stage per operand; the second stage of the pipeline is the register file (loads for tile t+1 are
    // issued before tile t's MFMAs and written to LDS after them, inside one loop iteration: no loop-carried
    // in-flight registers, which hipcc would otherwise "fix" with copies behind a vmcnt(0)).  Small LDS + modest
    // VGPR use => 4 workgroups per CU; latency is hidden by occupancy, not by prefetch depth.
    __shared__ __attribute__((aligned(16))) float Xs[1][BK * LDX];
    __shared__ float Ws[1][WSETS][BK * LDW];
    __shared__ int32_t kt_w[2][KCH];   // (k_eff -> weight offset, x row) for a chunk of 256 k_eff = 8 tiles,
    __shared__ int32_t kt_x[2][KCH];   // filled by all 256 threads at once, double buffered
    __shared__ float sbias[POOL ? BN : 1];   // POOL: the item's 64 bias values (read once per window pixel, in the MFMA layout)
    __shared__ float sbias2[(POOL && LRT) ? BN : 1];   // ... and the 64 bias variances of an LRT layer
    // (static LDS arrays, one set per instantiation: a kernel should call ONE instantiation of this function)

    // ---- item -> (draw, channel tile, pixel, batch tile) ----
    // Default order: (draw, channel tile) major, (pixel, batch tile) minor -- neighbours on an XCD share a WEIGHT tile.  A layer
    // whose input is shared by the draws of a step (the first layer: x_ds == 0, or x_div draws per input slab) is enumerated
    // (step, pixel, batch tile) major and (draw, channel tile) minor instead: co-resident workgroups then read the SAME image rows
    // -- AlexNet's 6.3 MB input does not fit a 4 MB XCD L2, so in draw-major order every draw re-fetched it from the fabric (78 MB
    // per step for 7 MB of operands, profiles/r03_pmc_FETCH_SIZE.txt), while all draws' weights of such a layer (0.9 MB) do fit.
    int g, j;
    if (p.x_ds == 0 || (p.x_div > 1 && p.x_off == 0 && p.G % (p.x_div * p.Ntiles) == 0)) {
        const int gs = p.x_div > 1 ? p.x_div * p.Ntiles : p.G;       // (draw, channel tile) groups per step
        const int64_t per_step = (int64_t)p.Mtiles * gs;
        const int step = (int)(item / per_step);
        const int64_t r = item - (int64_t)step * per_step;
        j = (int)(r / gs);
        g = step * gs + (int)(r - (int64_t)j * gs);
    } else {
        g = (int)(item / p.Mtiles);
        j = (int)(item - (int64_t)g * p.Mtiles);
    }
    const int e = g / p.Ntiles;
    // work units (ensemble sharding): slab e is unit u = unit_off + e -> weight set u / S, input slab e (or u % S)
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
    const int n0 = (g - e * p.Ntiles) * BN;
    const int pix = j / p.nbt;
    const int b0 = (j - pix * p.nbt) * BM;
    // POOL: pix is the POOLED pixel; its window's conv pixels are (2 poh + wy, 2 pow + wx)
    const int Wq = POOL ? p.Wo / 2 : p.Wo;
    const int oh0 = pix / Wq, ow0 = pix - oh0 * Wq;
    // in-bounds tap ranges of the current conv pixel (POOL: re-set per window pixel; all wave-uniform)
    int ihb, iwb, r_lo, q_lo, nq, nrq, Keff, ntiles;
    float inv_nrq, inv_nq;
    const float inv_cin = 1.0f / (float)p.Cin;
    auto set_pixel = [&](int oh, int ow) {
        if constexpr (POOL) {            // re-set inside a loop: keep the pixel's state in scalar registers
            oh = __builtin_amdgcn_readfirstlane(oh);
            ow = __builtin_amdgcn_readfirstlane(ow);
        }
        ihb = oh * p.sh - p.ph;
        iwb = ow * p.sw - p.pw;
        r_lo = ihb < 0 ? (-ihb + p.dh - 1) / p.dh : 0;
        q_lo = iwb < 0 ? (-iwb + p.dw - 1) / p.dw : 0;
        int r_hi = (p.H - 1 - ihb) >= 0 ? (p.H - 1 - ihb) / p.dh + 1 : 0;
        int q_hi = (p.W - 1 - iwb) >= 0 ? (p.W - 1 - iwb) / p.dw + 1 : 0;
        r_hi = r_hi < p.kh ? r_hi : p.kh;
        q_hi = q_hi < p.kw ? q_hi : p.kw;
        const int nr = r_hi > r_lo ? r_hi - r_lo : 0;
        nq = q_hi > q_lo ? q_hi - q_lo : 0;
        nrq = nr * nq;
        Keff = p.Cin * nrq;
        ntiles = (Keff + BK - 1) / BK;
        // k_eff -> (ci, r, q) with float-reciprocal division + fix-up (exact for k_eff < 2^24)
        inv_nrq = nrq > 0 ? 1.0f / (float)nrq : 0.0f;
        inv_nq = nq > 0 ? 1.0f / (float)nq : 0.0f;
    };
    set_pixel(POOL ? 2 * oh0 : oh0, POOL ? 2 * ow0 : ow0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (BM >= 128) ? 0 : (wave >> 1) * 32;
    const int wm = (BM >= 128) ? wave * 32 * MT : (wave & 1) * 32;

    // Buffer descriptors (wave-uniform): out-of-range offsets read as 0, which is how invalid k / channel / image
    // lanes are masked without a branch around every load (a branch would make hipcc wait vmcnt(0) per element).
    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    constexpr uint32_t kWInv = 0x7FFFFFF0u;    // invalid weight k: slab bytes < 2^30, so row + kWInv is out of range
    const uint32_t kXInv = p.x_inv;            // invalid x row: + column bytes (< one row) neither wraps nor lands in range
    const int64_t w_elems = (int64_t)p.Cout * p.Kp;                  // Kp = weight row pitch (= K unless the caller passed one)
    const int64_t x_elems = (int64_t)p.Cin * p.H * p.W * p.B;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)ex * p.x_ds), 0, (int)(x_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w + (int64_t)ew * p.w_ds), 0, (int)(w_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(LRT ? p.w2 + (int64_t)ew * p.w_ds : p.w), 0, (int)(w_elems * 4), 0x00020000);

    // loaders
    const int wkl = tid & 31, wnl = tid >> 5;            // weights: lane -> k, 8 channel rows per pass
    const int xb4 = (tid % XL) * 4, xkr = tid / XL;      // x: lane -> 4 images, XRPP k rows per pass
    // No per-lane masking of channels >= Cout or images >= B: such rows / columns of D are never stored, GEMM
    // columns do not mix, and reads past a slab come back as 0 from the buffer unit.  Only invalid k must be zero
    // (on both operands), which the table encodes as out-of-range offsets.
    const uint32_t xcol = (uint32_t)(b0 + xb4) * 4u;                 // byte offset of this lane's 4 images in a row
    uint32_t wrow[8];                                                // byte offset of this lane's 8 channel rows
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) wrow[ps] = (uint32_t)(n0 + wnl + ps * 8) * (uint32_t)p.Kp * 4u;

    float wregA[WSETS][8];
    f32x4 xregA[XPASS];

    auto fill_chunk = [&](int chunk) {
        const int k = chunk * KCH + tid;
        uint32_t wo = kWInv, xo = kXInv;
        if (k < Keff) {
            int ci, rq;
            if (p.wtap) {                     // tap-major contraction (bbb_conv_desc_t::w_tap_major): k = (in-bounds tap, ci)
                rq = (int)((float)k * inv_cin);
                ci = k - rq * p.Cin;
                if (ci < 0) { --rq; ci += p.Cin; } else if (ci >= p.Cin) { ++rq; ci -= p.Cin; }
            } else {
                ci = (int)((float)k * inv_nrq);
                rq = k - ci * nrq;
                if (rq < 0) { --ci; rq += nrq; } else if (rq >= nrq) { ++ci; rq -= nrq; }
            }
            int rr = (int)((float)rq * inv_nq);
            int qq = rq - rr * nq;
            if (qq < 0) { --rr; qq += nq; } else if (qq >= nq) { ++rr; qq -= nq; }
            const int r = r_lo + rr, q = q_lo + qq;
            wo = (p.wtap ? (uint32_t)((r * p.kw + q) * p.Cin + ci) : (uint32_t)(ci * p.khkw + r * p.kw + q)) * 4u;   // byte offset in a row
            xo = (uint32_t)((ci * p.H + ihb + r * p.dh) * p.W + iwb + q * p.dw) * (uint32_t)p.B * 4u;   // row byte offset
        }
        kt_w[chunk & 1][tid] = (int32_t)wo;
        kt_x[chunk & 1][tid] = (int32_t)xo;
    };

    // Staging loads are split into an address phase (table lookups + adds, before the tile's MFMAs) and the individual
    // buffer loads.  ILV = true: mma_tile() interleaves the loads BETWEEN the MFMAs of the current tile - a VMEM
    // instruction costs the issuing wave 100-200 cycles, and among MFMAs that time hides in the 64-cycle shadows of the
    // matrix pipe (s_memtime: the load phase of a lone workgroup was as long as its MFMA phase).  Measured: +8-10 % on
    // launches of <= ~1.5 workgroup rounds (latency-bound), -5-15 % on multi-round launches where four co-resident
    // workgroups already keep the pipe busy and the interleaved loads only delay MFMA issue.  The launcher picks.
    constexpr int NLOADS = 8 + XPASS;
    uint32_t loff[NLOADS];
    auto load_addr = [&](int tile) {
        const int buf = (tile / TPC) & 1;
        const int kb = (tile % TPC) * BK;
        const uint32_t wob = (uint32_t)kt_w[buf][kb + wkl];
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) loff[ps] = wrow[ps] + wob;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) loff[8 + ps] = (uint32_t)kt_x[buf][kb + xkr + ps * XRPP] + xcol;
    };
    auto load_one = [&](int i, float (&wreg)[WSETS][8], f32x4 (&xreg)[XPASS]) {
        if (i < XPASS) {          // x rows first: they are the wider transfers
            xreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, loff[8 + i], 0, 0));
        } else {
            const int ps = i - XPASS;
            wreg[0][ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, loff[ps], 0, 0));
            if (LRT) wreg[WSETS - 1][ps] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(w2rs, loff[ps], 0, 0));
        }
    };
    auto load_tile = [&](int tile, float (&wreg)[WSETS][8], f32x4 (&xreg)[XPASS]) {
        load_addr(tile);
#pragma unroll
        for (int i = 0; i < NLOADS; ++i) load_one(i, wreg, xreg);
    };

    auto store_tile = [&](int, float (&wreg)[WSETS][8], f32x4 (&xreg)[XPASS]) {
        constexpr int buf = 0;
#pragma unroll
        for (int s = 0; s < WSETS; ++s)
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) Ws[buf][s][wkl * LDW + wnl + ps * 8] = wreg[s][ps];
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps)
            *reinterpret_cast<f32x4*>(&Xs[buf][(xkr + ps * XRPP) * LDX + xb4]) = xreg[ps];
    };

    f32x16 acc[NT][MT];
    f32x16 accv[NT][MT];
    float tot[(SEQ || POOL) ? NT : 1][(SEQ || POOL) ? MT : 1][16];   // SEQ: running total of the finished ranges' partial sums; POOL: running maximum (AGPRs)
    float totv[(SEQ && LRT) ? NT : 1][(SEQ && LRT) ? MT : 1][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[t][u][r] = 0.0f; accv[t][u][r] = 0.0f;
                if constexpr (SEQ) { agpr_write(tot[t][u][r], 0.0f); if constexpr (LRT) agpr_write(totv[t][u][r], 0.0f); }
                if constexpr (POOL) agpr_write(tot[t][u][r], -__builtin_inff());
            }

    const int lrow = lane & 31, lk = lane >> 5;

    // Operand reads from LDS run PD k-steps ahead of the MFMAs that use them (a small register ring): a wave never waits a
    // full LDS round trip between k-steps, so even a workgroup that is alone on its CU (launch tails, small launches)
    // keeps its matrix pipe fed inside a tile.
    constexpr int PD = (BM >= 128) ? 2 : 4;
    auto mma_tile = [&](bool more) {
        float ra[PD + 1][NT], ra2[PD + 1][NT], rb[PD + 1][MT];
        auto fetch = [&](int kk, int slot) {
            const int krow = kk * 2 + lk;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) rb[slot][mt] = Xs[0][krow * LDX + wm + mt * 32 + lrow];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ra[slot][nt] = Ws[0][0][krow * LDW + wn + nt * 32 + lrow];
                if (LRT) ra2[slot][nt] = Ws[0][WSETS - 1][krow * LDW + wn + nt * 32 + lrow];
            }
        };
#pragma unroll
        for (int i = 0; i < PD; ++i) fetch(i, i);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (ILV && kk < NLOADS) {
                if (more) load_one(kk, wregA, xregA);
                __builtin_amdgcn_sched_barrier(0);     // keep this load ahead of k-step kk's MFMAs, behind kk-1's
            }
            if (kk + PD < BK / 2) fetch(kk + PD, (kk + PD) % (PD + 1));
            const int s = kk % (PD + 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s][nt], rb[s][mt], acc[nt][mt], 0, 0, 0);
                if (LRT) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        accv[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra2[s][nt], rb[s][mt] * rb[s][mt], accv[nt][mt], 0, 0, 0);
                }
            }
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
        }
    };

    if constexpr (POOL) {
        // the item's bias values, once (channels >= Cout and a launch without bias read as 0 from the buffer unit)
        const __amdgpu_buffer_rsrc_t brs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : p.w), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
        if (tid < BN) sbias[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs0, (uint32_t)(n0 + tid) * 4u, 0, 0));
        if constexpr (LRT) {
            const __amdgpu_buffer_rsrc_t brs2 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.bias2 ? p.bias2 + (int64_t)ew * p.b_ds : p.w), 0, p.bias2 ? p.Cout * 4 : 0, 0x00020000);
            if (tid < BN) sbias2[tid] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs2, (uint32_t)(n0 + tid) * 4u, 0, 0));
        }
        __syncthreads();
    }
#pragma clang loop unroll(disable)
    for (int wp = 0; wp < (POOL ? 4 : 1); ++wp) {
    if (POOL && wp > 0) set_pixel(2 * oh0 + (wp >> 1), 2 * ow0 + (wp & 1));
    int t0 = 0, t1 = ntiles;
    if constexpr (SPLIT) {
        t0 = (int)((int64_t)ks * ntiles / p.ksplit);
        t1 = (int)((int64_t)(ks + 1) * ntiles / p.ksplit);
    }
    if (t1 > t0) {
        const int c0 = t0 / TPC;                                      // first decode chunk of this range (0 when not split)
        fill_chunk(c0);
        __syncthreads();
        load_tile(t0, wregA, xregA);
        if ((c0 + 1) * KCH < Keff) fill_chunk(c0 + 1);
        store_tile(0, wregA, xregA);
        __syncthreads();
        // SEQ: the ranges [r * ntiles / S, (r + 1) * ntiles / S) one after the other -- the tile loop (and its software pipeline:
        // tile t + 1 is prefetched across a range boundary) is the plain one; BETWEEN two ranges the accumulator is added to the
        // running total and zeroed (an empty range adds a zero partial, exactly as its SPLIT workgroup would)
        const int nranges = SEQ ? p.ksplit : 1;
        int t = t0;
#pragma clang loop unroll(disable)
        for (int sr = 0; sr < nranges; ++sr) {
            const int tr1 = SEQ ? (int)((int64_t)(sr + 1) * ntiles / p.ksplit) : t1;
#pragma clang loop unroll(disable)
            for (; t < tr1; ++t) {
                const bool more = (t + 1) < t1;
                if (more) {
                    if (ILV) load_addr(t + 1);                        // loads themselves are issued inside mma_tile()
                    else     load_tile(t + 1, wregA, xregA);          // all loads up front (large launches)
                }
                // decode chunk c+1 early in chunk c (c >= c0 + 1; chunk c0 + 1 is decoded in the prologue): its buffer was last
                // read by load_tile(TPC*c - 1), several barriers ago
                if ((t % TPC) == 1 && t / TPC >= c0 + 1 && (t / TPC + 1) * KCH < Keff) fill_chunk(t / TPC + 1);
                mma_tile(more);
                __syncthreads();                                      // every wave is done reading the LDS stage
                if (more) store_tile(0, wregA, xregA);
                __syncthreads();
            }
            if constexpr (SEQ) {
                if (sr + 1 < nranges) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                agpr_write(tot[nt][mt][r], agpr_read(tot[nt][mt][r]) + acc[nt][mt][r]);   // (0 + p0 = p0)
                                acc[nt][mt][r] = 0.0f;
                                if constexpr (LRT) {
                                    agpr_write(totv[nt][mt][r], agpr_read(totv[nt][mt][r]) + accv[nt][mt][r]);
                                    accv[nt][mt][r] = 0.0f;
                                }
                            }
                }
            }
        }
        if constexpr (SEQ) {
            // the last range's partial sum is added last
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc[nt][mt][r] = agpr_read(tot[nt][mt][r]) + acc[nt][mt][r];
                        if constexpr (LRT) accv[nt][mt][r] = agpr_read(totv[nt][mt][r]) + accv[nt][mt][r];
                    }
        }
    }
    if constexpr (POOL && !LRT) {
        // this window pixel is done: running maximum of act(conv + bias), accumulator back to zero for the next pixel
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bn = sbias[wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float v = bbb::apply_act(acc[nt][mt][r] + bn, p.act);
                    agpr_write(tot[nt][mt][r], fmaxf(agpr_read(tot[nt][mt][r]), v));
                    acc[nt][mt][r] = 0.0f;
                }
            }
    }
    if constexpr (POOL && LRT) {
        // the LRT layer's window pixel: out = act(act_mu + sqrt(act_var) * eps) exactly as the unpooled epilogue computes it (same
        // noise element: the canonical NCHW index of the CONV output), then the running maximum
        const int pixc = (2 * oh0 + (wp >> 1)) * p.Wo + 2 * ow0 + (wp & 1);
        const int HoWoC = p.Ho * p.Wo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = b0 + wm + mt * 32 + lrow;
            const int bglob = b + p.b_off + (p.unit_div > 1 ? (ue % p.unit_div) * p.B : 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int nl = wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    const int n = n0 + nl;
                    float v = acc[nt][mt][r] + sbias[nl];
                    const float var = 1e-16f + (accv[nt][mt][r] + sbias2[nl]);
                    if (p.sample) {
                        float z;
                        if (p.eps_ext) {
                            // (lanes past the batch / channel edge: their rows are never stored; keep the read in bounds)
                            const bool in = (b < p.B) & (n < p.Cout);
                            z = in ? p.eps_ext[(int64_t)e * ((int64_t)p.Cout * HoWoC * p.B) + ((int64_t)n * HoWoC + pixc) * p.B + b] : 0.0f;
                        } else {
                            const uint64_t idx = (uint64_t)(((int64_t)bglob * p.Cout + n) * HoWoC + pixc);
                            float z4[4];
                            bbb::normal4(idx >> 2, p.stream_id, p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)ew, p.k0, p.k1, z4);
                            const int c = (int)(idx & 3);
                            z = c == 0 ? z4[0] : c == 1 ? z4[1] : c == 2 ? z4[2] : z4[3];
                        }
                        v = v + __builtin_amdgcn_sqrtf(var) * z;
                    }
                    v = bbb::apply_act(v, p.act);
                    agpr_write(tot[nt][mt][r], fmaxf(agpr_read(tot[nt][mt][r]), v));
                    acc[nt][mt][r] = 0.0f;
                    accv[nt][mt][r] = 0.0f;
                    __builtin_amdgcn_sched_barrier(0);     // one element's Philox at a time: interleaved, sixteen of them hold ~90 VGPRs
                }
        }
    }
    }   // window pixels
    if constexpr (POOL) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][mt][r] = agpr_read(tot[nt][mt][r]);
    }

    if constexpr (SPLIT) {
        // ---- split contraction: partial tile(s) -> scratch, ticket, last arriver combines ----
        constexpr int TILE = 64 * BM;                                 // elements of one accumulator set's tile
        constexpr int TW = (BK * LDX) / 1024 >= 4 ? 4 : 2;            // wave tiles the X stage can hold (see the epilogue)
        __shared__ int s_last;
        const int S = p.ksplit;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
            p.part + (int64_t)item * S * (WSETS * TILE), 0, S * WSETS * TILE * 4, 0x00020000);
        float* const T = &Xs[0][(wave % TW) * 1024];
#pragma unroll
        for (int set = 0; set < WSETS; ++set)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int pass = 0; pass < 4 / TW; ++pass) {
                        if ((wave / TW) == pass) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                T[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + lrow] = set == 0 ? acc[nt][mt][r] : accv[nt][mt][r];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int nl = (lane >> 3) + 8 * i, b4 = (lane & 7) * 4;
                                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&T[nl * 32 + b4]);
                                const uint32_t off = (uint32_t)(((ks * WSETS + set) * 64 + wn + nt * 32 + nl) * BM + wm + mt * 32 + b4) * 4u;
                                __builtin_amdgcn_raw_buffer_store_b128(
                                    __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, v4), prs, off, 0, 16);
                            }
                        }
                        if (TW < 4) __syncthreads();                  // the other pair of waves reuses the tiles
                    }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave's write-through stores have left
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.tickets + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == S - 1;
            if (old == S - 1) __hip_atomic_store(p.tickets + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        // the last arriver: every partial tile of this item is in memory.  Thread -> (channel row, 4 consecutive images).
        const int HoWo = p.Ho * p.Wo;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            p.y + (int64_t)e * p.y_ds, 0, (int)((int64_t)p.Cout * HoWo * p.B * 4), 0x00020000);
        const float* __restrict__ b1g = p.bias ? p.bias + (int64_t)ew * p.b_ds : nullptr;
        const float* __restrict__ b2g = (LRT && p.bias2) ? p.bias2 + (int64_t)ew * p.b_ds : nullptr;
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
            const int f = q * kThreads + tid;
            const int nl = f / (BM / 4), b4 = (f % (BM / 4)) * 4;
            f32x4 sum[WSETS];
#pragma unroll
            for (int set = 0; set < WSETS; ++set) {
                sum[set] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (uint32_t)((set * 64 + nl) * BM + b4) * 4u, 0, 16));
                for (int k2 = 1; k2 < S; ++k2) {
                    const f32x4 v = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, (uint32_t)(((k2 * WSETS + set) * 64 + nl) * BM + b4) * 4u, 0, 16));
#pragma unroll
                    for (int c = 0; c < 4; ++c) sum[set][c] += v[c];
                }
            }
            const int n = n0 + nl, bb = b0 + b4;
            if (n < p.Cout && bb < p.B) {
                const float bn = b1g ? b1g[n] : 0.0f;
                f32x4 o;
                if constexpr (!LRT) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = bbb::apply_act(sum[0][c] + bn, p.act);
                } else {
                    const float b2n = b2g ? b2g[n] : 0.0f;
                    const int64_t obase = (int64_t)e * p.y_ds + ((int64_t)n * HoWo + pix) * p.B + bb;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = sum[0][c] + bn;
                        const float var = 1e-16f + (sum[WSETS - 1][c] + b2n);
                        if (p.y_mu) p.y_mu[obase + c] = v;
                        if (p.y_var) p.y_var[obase + c] = var;
                        if (p.sample) {
                            float z;
                            if (p.eps_ext) {
                                z = p.eps_ext[obase + c];
                            } else {
                                const int bglob = bb + c + p.b_off + (p.unit_div > 1 ? (ue % p.unit_div) * p.B : 0);
                                const uint64_t idx = (uint64_t)(((int64_t)bglob * p.Cout + n) * HoWo + pix);
                                float z4[4];
                                bbb::normal4(idx >> 2, p.stream_id, p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)ew, p.k0, p.k1, z4);
                                const int cc = (int)(idx & 3);
                                z = cc == 0 ? z4[0] : cc == 1 ? z4[1] : cc == 2 ? z4[2] : z4[3];
                            }
                            v = v + __builtin_amdgcn_sqrtf(var) * z;
                        }
                        o[c] = bbb::apply_act(v, p.act);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, o),
                                                       yrs, (uint32_t)(((int64_t)n * HoWo + pix) * p.B + bb) * 4u, 0, 0);
            }
        }
        return;
    }

    // ---- epilogue: rows = channels, lanes = images; bias via buffer loads, stores via buffer stores
    //      (out-of-range channel / image lanes get an out-of-range offset: no branches, no per-element waits) ----
    const int HoWo = POOL ? (p.Ho / 2) * (p.Wo / 2) : p.Ho * p.Wo;    // POOL: rows of the pooled map; pix is the pooled pixel
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : p.w), 0, (p.bias && !POOL) ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        p.y + (int64_t)e * p.y_ds, 0, (int)((int64_t)p.Cout * HoWo * p.B * 4), 0x00020000);
    if constexpr (!LRT || POOL) {
        // (pooled LRT items end like pooled BBB items: `acc` holds finished output values)
        // Every wave transposes its 32 x 32 accumulator tile(s) through LDS (the X stage is free after the k loop) so that a
        // lane ends up with FOUR consecutive images of one channel: 8 16-byte stores and 8 bias loads per lane instead of 32 +
        // 32 four-byte ones (measured neutral at 4 workgroups per CU -- profiles/r03_notes.md section 3; the split form's
        // write-through partial stores need the 16-byte shape).  Same values, same bits as the MFMA-layout epilogue of rounds 1-2.
        // The stage holds TW wave tiles at a time (BM = 64: two passes).
        constexpr int TW = (BK * LDX) / 1024 >= 4 ? 4 : 2;               // wave tiles the X stage can hold
        float* const T = &Xs[0][(wave % TW) * 1024];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int pass = 0; pass < 4 / TW; ++pass) {
                    if ((wave / TW) == pass) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + lrow] = acc[nt][mt][r];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int nl = (lane >> 3) + 8 * i, b4 = (lane & 7) * 4;
                            const int n = n0 + wn + nt * 32 + nl, bb = b0 + wm + mt * 32 + b4;
                            const float bn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
                            f32x4 v4 = *reinterpret_cast<const f32x4*>(&T[nl * 32 + b4]);
#pragma unroll
                            for (int c = 0; c < 4; ++c) v4[c] = POOL ? v4[c] : bbb::apply_act(v4[c] + bn, p.act);   // (POOL: applied per window pixel)
                            const uint32_t off = ((bb < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + bb) * 4u : kOOB;
                            __builtin_amdgcn_raw_buffer_store_b128(
                                __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, v4), yrs, off, 0, 0);
                        }
                    }
                    if (TW < 4) __syncthreads();                         // the other pair of waves reuses the tiles
                }
            }
        return;
    }
    // ---- LRT epilogue (MFMA layout: rows = channels, lanes = images) ----
    float bv[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            bv[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int b = b0 + wm + mt * 32 + lrow;
        const bool b_ok = b < p.B;
        if (b_ok) {
            const int64_t ybase = (int64_t)e * p.y_ds + (int64_t)pix * p.B + b;
            const float* __restrict__ b2g = p.bias2 ? p.bias2 + (int64_t)ew * p.b_ds : nullptr;
            const int bglob = b + p.b_off + (p.unit_div > 1 ? (ue % p.unit_div) * p.B : 0);      // image index that keys the noise
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + wn + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (n < p.Cout) {
                        const int64_t o = ybase + (int64_t)n * HoWo * p.B;
                        float v = acc[nt][mt][r] + bv[nt][r];
                        const float var = 1e-16f + (accv[nt][mt][r] + (b2g ? b2g[n] : 0.0f));
                        if (p.y_mu) p.y_mu[o] = v;
                        if (p.y_var) p.y_var[o] = var;
                        if (p.sample) {
                            float z;
                            if (p.eps_ext) {
                                z = p.eps_ext[o];
                            } else {   // canonical NCHW element index of this draw's [B][Cout][Ho][Wo] slab
                                const uint64_t idx = (uint64_t)(((int64_t)bglob * p.Cout + n) * HoWo + pix);
                                float z4[4];
                                bbb::normal4(idx >> 2, p.stream_id, p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)ew, p.k0, p.k1, z4);
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
}

}  // namespace pconv
