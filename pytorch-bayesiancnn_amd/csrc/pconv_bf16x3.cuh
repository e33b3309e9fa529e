// fp32-accurate contraction on the 16-bit matrix pipe, RANGE-FREE ("split bf16", round 4; replaces round 3's split-fp16 form and
// its operand windows / amax plumbing): the pixel-major implicit GEMM of pconv_body.cuh with every fp32 operand element a cut
// into THREE bf16 pieces while its tile is staged,
//     hi = bf16(a)   mid = bf16(a - hi)   lo = bf16(a - hi - mid)          (round to nearest even; both differences are exact)
// a = hi + mid + lo EXACTLY (3 x 8 significand bits; bf16 has fp32's exponent range, so no scaling, no window, no saturation:
// weights of 1e-4, variances of 1e-9 and gradients of 1e-6 split as exactly as O(1) activations; only elements below 2^-110,
// whose lo piece would be a bf16 subnormal, lose bits), and every product taken as
//     lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi       on v_mfma_f32_32x32x16_bf16, fp32 accumulation, small terms first;
// the three dropped terms (mid*lo, lo*mid, lo*lo) are < 2^-23 |a b|, the size of the rounding of one fp32 product.  Six
// instructions of 32 cycles do the work of sixteen 64-cycle v_mfma_f32_32x32x2_f32: 3/8 of the fp32 pipe time at the nominal rates.
// Inputs, weights, bias and outputs stay fp32 in HBM in the layouts of bbb_conv2d_chwn_fwd: the kernel is a drop-in for that
// launch (same descriptor, work units and x_unit_div included); results differ from the fp32 fmaf chain by rounding, not bit
// for bit, which is why it is a mode (ops.gemm_mode = "bf16x3") and not the silent default.
// Split ACTIVATION format ("S3", template flags XPRE / OSPLIT): the kernel is VALU-bound on the split itself -- every image row
// element is cut into pieces again by every channel-tile workgroup that stages it (24 elements per thread and tile against 768
// cycles of matrix work per wave).  So between the layers of a step activations may travel already split: three bf16 planes per
// slab, [draw][3][C][H][W][B], written ONCE by the producing launch's epilogue (OSPLIT), pooled plane-wise
// (bbb_maxpool_chwn_s3), and staged by the consumer with plain 16-byte loads and no arithmetic (XPRE).  hi + mid + lo is still
// the exact fp32 value, so the numbers do not change: S3 is a storage format of the same fp32 activations (6 bytes per element).
// Same decomposition: workgroup = one output pixel, 64 channels, 128 images, in-bounds taps only, k tables in LDS; staging
// moves 16-byte vectors on the LDS side (a thread owns 8 adjacent channels of one k / 8 adjacent images of one row), both MFMA
// operands come from ds_read_b64_tr_b16, the epilogue goes through an LDS transpose to 16-byte stores.
#pragma once
#include "pconv_body.cuh"

namespace pconv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short f16s4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (a0, a1) -> packed (hi, mid, lo) bf16 PAIRS, element 0 in the low half of each word.  Written on pairs so that it compiles to
// 3 v_cvt_pk_bf16_f32 (round to nearest even) + 2 shifts + 2 ands (bf16 -> fp32 is a 16-bit shift) + 2 v_pk_add_f32: 4.5 VALU
// instructions per element -- the kernel is VALU-bound on exactly this work (left to the scalar form hipcc spent 8 per element).
__device__ __forceinline__ void split3_pair(f32x2 a, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
    const f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xFFFF0000u)};
    const f32x2 r1 = a - hf;                                     // exact
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xFFFF0000u)};
    const f32x2 r2 = r1 - mf;                                    // exact
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MT, bool XPRE, bool OSPLIT>
__global__ __launch_bounds__(kThreads) void pconv_bf16x3_kernel(const PConvArgs p) {
    constexpr int BM = 128 * MT;
    constexpr int LDXH = BM + 32;              // 16-bit elements per image row: 64 B mod 256 -> the 4 rows of a transpose read hit disjoint banks
    constexpr int LDWH = BN + 32;              // 192 B: rows at 0 / 192 / 128 / 64 mod 256
    constexpr int XL = BM / 8, XRPP = kThreads / XL, XPASS = BK / XRPP;        // a thread moves 8 adjacent images of a row
    __shared__ __attribute__((aligned(16))) unsigned short Xp[3 * BK * LDXH];    // hi, mid, lo planes ([k][b]); epilogue staging afterwards
    __shared__ __attribute__((aligned(16))) unsigned short Wp[3 * BK * LDWH];    // hi, mid, lo planes ([k][n], columns swizzled)
    __shared__ int32_t kt_w[2][KCH];
    __shared__ int32_t kt_x[2][KCH];
    unsigned short* const Xh = Xp;
    unsigned short* const Xm = Xp + BK * LDXH;
    unsigned short* const Xl = Xp + 2 * BK * LDXH;
    unsigned short* const Wh = Wp;
    unsigned short* const Wm = Wp + BK * LDWH;
    unsigned short* const Wl = Wp + 2 * BK * LDWH;

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int64_t item = (int64_t)xcd * p.per_xcd + (bid >> 3);
    const int64_t item_end = (int64_t)(xcd + 1) * p.per_xcd;
    if (item >= item_end || item >= (int64_t)p.G * p.Mtiles) return;
    const int g = (int)(item / p.Mtiles);
    const int j = (int)(item - (int64_t)g * p.Mtiles);
    const int e = g / p.Ntiles;
    const int ue = p.unit_off + e;
    const int ew = p.unit_div > 1 ? ue / p.unit_div : e;
    const int ex = p.x_div > 1 ? (e + p.x_off) / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
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
    const int wm = wave * 32 * MT;                                   // 4 waves side by side along the images, 64 channels each

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    constexpr uint32_t kWInv = 0x7FFFFFF0u;
    const uint32_t kXInv = p.x_inv;
    // XPRE: p.x is a bf16 S3 tensor (x_ds / x_ps count bf16 elements); one descriptor per plane, sized like one plane, so that
    // the invalid-row marker (half the fp32 one: offsets are in 2-byte elements here) stays out of range
    const unsigned short* const xb16 = reinterpret_cast<const unsigned short*>(p.x) + (XPRE ? (int64_t)ex * p.x_ds : 0);
    const int x_bytes = (int)((int64_t)p.Cin * p.H * p.W * p.B * (XPRE ? 2 : 4));
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        XPRE ? (void*)const_cast<unsigned short*>(xb16) : (void*)const_cast<float*>(p.x + (int64_t)ex * p.x_ds), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(xb16 + (XPRE ? p.x_ps : 0)), 0, XPRE ? x_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(xb16 + (XPRE ? 2 * p.x_ps : 0)), 0, XPRE ? x_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w + (int64_t)ew * p.w_ds), 0, (int)((int64_t)p.Cout * p.Kp * 4), 0x00020000);

    // loaders.  Weights: lane -> k (32 per half-wave), 8 ADJACENT channel rows per thread, so that both pieces of the thread's
    // 8 elements go to LDS as one 16-byte write per plane.  The 32 lanes of a half-wave write 32 rows at the same column: at a
    // 192-byte pitch that is 4 banks, so columns are XOR-swizzled in 8-element granules by the row's group of four, (k >> 2) & 7;
    // a transpose read -- 4 consecutive rows of ONE group, 4 consecutive elements -- sees the same permutation on all its rows.
    const int wkl = tid & 31, wng = (tid >> 5) * 8;
    const int wswz = ((wkl >> 2) & 7) << 3;
    const int xb8 = (tid % XL) * 8, xkr = tid / XL;
    const uint32_t xcol = (uint32_t)(b0 + xb8) * (XPRE ? 2u : 4u);
    uint32_t wrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wrow[i] = (uint32_t)(n0 + wng + i) * (uint32_t)p.Kp * 4u;

    // One register stage + one LDS stage, as pconv_body.cuh: loads for tile t+1 are issued before tile t's MFMAs and written to
    // LDS after them.
    constexpr int XV = (XPRE ? 3 : 2) * XPASS;           // 16-byte vectors of image data per thread and tile
    float wregA[8];
    u32x4 xregA[XV];

    const float inv_nrq = nrq > 0 ? 1.0f / (float)nrq : 0.0f;
    const float inv_nq = nq > 0 ? 1.0f / (float)nq : 0.0f;
    auto fill_chunk = [&](int chunk) {
        const int k = chunk * KCH + tid;
        uint32_t wo = kWInv, xo = kXInv;
        if (k < Keff) {
            int ci = (int)((float)k * inv_nrq);
            int rq = k - ci * nrq;
            if (rq < 0) { --ci; rq += nrq; } else if (rq >= nrq) { ++ci; rq -= nrq; }
            int rr = (int)((float)rq * inv_nq);
            int qq = rq - rr * nq;
            if (qq < 0) { --rr; qq += nq; } else if (qq >= nq) { ++rr; qq -= nq; }
            const int r = r_lo + rr, q = q_lo + qq;
            wo = (uint32_t)(ci * p.khkw + r * p.kw + q) * 4u;
            xo = (uint32_t)((ci * p.H + ihb + r * p.dh) * p.W + iwb + q * p.dw) * (uint32_t)p.B * 4u;
        }
        kt_w[chunk & 1][tid] = (int32_t)wo;
        kt_x[chunk & 1][tid] = (int32_t)xo;
    };
    auto load_tile = [&](int tile, float (&wreg)[8], u32x4 (&xreg)[XV]) {
        const int buf = (tile / TPC) & 1;
        const int kb = (tile % TPC) * BK;
        const uint32_t wob = (uint32_t)kt_w[buf][kb + wkl];
        uint32_t xo[XPASS];
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps)       // (the table holds fp32 byte offsets of a row: half of that in a bf16 plane)
            xo[ps] = (XPRE ? ((uint32_t)kt_x[buf][kb + xkr + ps * XRPP] >> 1) : (uint32_t)kt_x[buf][kb + xkr + ps * XRPP]) + xcol;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) {
            if constexpr (XPRE) {
                xreg[3 * ps] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps], 0, 0));
                xreg[3 * ps + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs1, xo[ps], 0, 0));
                xreg[3 * ps + 2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs2, xo[ps], 0, 0));
            } else {
                xreg[2 * ps] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps], 0, 0));
                xreg[2 * ps + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps] + 16u, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wrow[i] + wob, 0, 0));
    };
    // the split: the three pieces of every staged element go to LDS, [k][n] / [k][b] planes of 16-bit elements, 16 bytes per write
    auto store_tile = [&](float (&wreg)[8], u32x4 (&xreg)[XV]) {
        u32x4 h, m, l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t a0, a1, a2;
            split3_pair(f32x2{wreg[2 * i], wreg[2 * i + 1]}, a0, a1, a2);
            h[i] = a0; m[i] = a1; l[i] = a2;
        }
        *reinterpret_cast<u32x4*>(&Wh[wkl * LDWH + (wng ^ wswz)]) = h;
        *reinterpret_cast<u32x4*>(&Wm[wkl * LDWH + (wng ^ wswz)]) = m;
        *reinterpret_cast<u32x4*>(&Wl[wkl * LDWH + (wng ^ wswz)]) = l;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) {
            if constexpr (XPRE) {                    // already split by the producer: straight to the planes
                h = xreg[3 * ps]; m = xreg[3 * ps + 1]; l = xreg[3 * ps + 2];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t a0, a1, a2;
                    const f32x4 v = __builtin_bit_cast(f32x4, xreg[2 * ps + (c >> 1)]);
                    split3_pair((c & 1) ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]}, a0, a1, a2);
                    h[c] = a0; m[c] = a1; l[c] = a2;
                }
            }
            *reinterpret_cast<u32x4*>(&Xh[(xkr + ps * XRPP) * LDXH + xb8]) = h;
            *reinterpret_cast<u32x4*>(&Xm[(xkr + ps * XRPP) * LDXH + xb8]) = m;
            *reinterpret_cast<u32x4*>(&Xl[(xkr + ps * XRPP) * LDXH + xb8]) = l;
        }
    };

    f32x16 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    const int lrow = lane & 31, lk = lane >> 5;
    // transpose read (ds_read_b64_tr_b16) of a [k][column] plane: 16-lane group -> (k half, 16-column block); the lane supplies
    // row (tt >> 2) of 4, columns 4*(tt & 3)..+3, and receives its column's 4 rows; two reads = the 8 k of its MFMA operand
    const int tg = lane >> 4, tt = lane & 15;
    const int tr_row = 8 * (tg >> 1) + (tt >> 2), tr_col = 16 * (tg & 1) + 4 * (tt & 3);
    typedef __attribute__((address_space(3))) f16s4* lds_s4_ptr;
    // LDS element offsets of this lane's reads (loop invariant: one LDS stage): [k16 step][first / second 4 rows]
    int xoff[BK / 16][2], woff[BK / 16][2][2];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = kk * 16 + tr_row + 4 * h;
            xoff[kk][h] = row * LDXH + wm + tr_col;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) woff[kk][h][nt] = row * LDWH + ((nt * 32 + tr_col) ^ (((row >> 2) & 7) << 3));
        }
    auto tr8 = [&](const unsigned short* plane, int o0, int o1) {
        const f16s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(plane + o0));
        const f16s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(plane + o1));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 ah[2], am[2], al[2], bh[MT], bm[MT], bl[MT];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                ah[nt] = tr8(Wh, woff[kk][0][nt], woff[kk][1][nt]);
                am[nt] = tr8(Wm, woff[kk][0][nt], woff[kk][1][nt]);
                al[nt] = tr8(Wl, woff[kk][0][nt], woff[kk][1][nt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bh[mt] = tr8(Xh, xoff[kk][0] + mt * 32, xoff[kk][1] + mt * 32);
                bm[mt] = tr8(Xm, xoff[kk][0] + mt * 32, xoff[kk][1] + mt * 32);
                bl[mt] = tr8(Xl, xoff[kk][0] + mt * 32, xoff[kk][1] + mt * 32);
            }
            // all eighteen LDS reads of the step in flight before its first MFMA (left alone, hipcc feeds each MFMA just in time)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {        // small terms first
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[nt], bh[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nt], bl[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[nt], bm[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[nt], bh[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nt], bm[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nt], bh[mt], acc[nt][mt], 0, 0, 0);
                }
        }
    };

    if (ntiles > 0) {
        fill_chunk(0);
        __syncthreads();
        load_tile(0, wregA, xregA);
        if (KCH < Keff) fill_chunk(1);
        store_tile(wregA, xregA);
        __syncthreads();                                             // tile 0 in LDS, table chunk 1 visible
        for (int t = 0; t < ntiles; ++t) {
            const bool more = (t + 1) < ntiles;
            if (more) load_tile(t + 1, wregA, xregA);
            // decode chunk c+1 early in chunk c (c >= 1; chunk 1 is decoded in the prologue)
            if ((t % TPC) == 1 && t / TPC >= 1 && (t / TPC + 1) * KCH < Keff) fill_chunk(t / TPC + 1);
            mma_tile();
            __syncthreads();
            if (more) store_tile(wregA, xregA);
            __syncthreads();
        }
    }

    // ---- epilogue: every wave transposes its 32 x 32 tiles through LDS (the image planes are free now) so that a lane ends
    //      up with 4 consecutive images of one channel: 4 bias loads and 4 16-byte stores per tile ----
    const int HoWo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : p.w), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    // OSPLIT: y is a bf16 S3 tensor [draw][3][Cout][Ho][Wo][B] (y_ds / y_ps count bf16 elements): one descriptor per plane
    unsigned short* const yb16 = reinterpret_cast<unsigned short*>(p.y) + (OSPLIT ? (int64_t)e * p.y_ds : 0);
    const int y_bytes = (int)((int64_t)p.Cout * HoWo * p.B * (OSPLIT ? 2 : 4));
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        OSPLIT ? (void*)yb16 : (void*)(p.y + (int64_t)e * p.y_ds), 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs1 = __builtin_amdgcn_make_buffer_rsrc(yb16 + (OSPLIT ? p.y_ps : 0), 0, OSPLIT ? y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs2 = __builtin_amdgcn_make_buffer_rsrc(yb16 + (OSPLIT ? 2 * p.y_ps : 0), 0, OSPLIT ? y_bytes : 0, 0x00020000);
    static_assert(3 * BK * LDXH * 2 >= 4 * 32 * 36 * 4, "epilogue staging must fit in the image planes");
    float* const T = reinterpret_cast<float*>(Xp) + wave * (32 * 36);       // [32 channels][36] floats, wave-private
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lk) * 36 + lrow] = acc[nt][mt][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nl = (lane >> 3) + 8 * i, b4 = (lane & 7) * 4;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(&T[nl * 36 + b4]);
                const int n = n0 + nt * 32 + nl, b = b0 + wm + mt * 32 + b4;
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = bbb::apply_act(v4[c] + bv, p.act);
                const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * (OSPLIT ? 2u : 4u) : kOOB;
                if constexpr (OSPLIT) {              // the next layer's operand pieces, written once here instead of per staging
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 h2, m2, l2;
                    uint32_t a0, a1, a2;
                    split3_pair(f32x2{o[0], o[1]}, a0, a1, a2);
                    h2[0] = a0; m2[0] = a1; l2[0] = a2;
                    split3_pair(f32x2{o[2], o[3]}, a0, a1, a2);
                    h2[1] = a0; m2[1] = a1; l2[1] = a2;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t, h2), yrs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t, m2), yrs1, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t, l2), yrs2, off, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, o), yrs, off, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next tile overwrites T
        }
}

}  // namespace pconv
