// fp32-accurate contraction on the 16-bit matrix pipe ("split fp16"): the pixel-major implicit GEMM of pconv_body.cuh with
// every fp32 operand element a cut into two fp16 pieces while its tile is staged,
//     hi = fp16(a)   lo = fp16(a - hi)        (a = hi + lo up to 2^-22 |a|; a pre-scaled by a power of two, see kScaleW / kScaleX)
// and every product taken as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the dropped lo*lo term is
// < 2^-22 |a b|).  Three instructions of 32 cycles do the work of sixteen 64-cycle v_mfma_f32_32x32x2_f32: 3/16 of the fp32 pipe
// time at the nominal rates (measured instruction ceilings on one box: 1952 vs 141 TFLOP/s, profiles/r03_notes.md section 11).
// Inputs, weights, bias and outputs stay fp32 in HBM in the layouts of bbb_conv2d_chwn_fwd: the kernel is a drop-in for that
// launch; results differ from the fp32 fmaf chain by rounding (measured against the float64 oracle: 1.2-2.4e-7 of sum |w||x|,
// the fp32 kernel 0.8-3.9e-7 on the same cases), not bit for bit.
// Same decomposition: workgroup = one output pixel, 64 channels, BM = 128 * MT images, in-bounds taps only, k tables in LDS.
// With the matrix phase of a tile down to 384 cycles per wave everything else shows (ablation, profiles/r03_notes.md section 12:
// loads, split + LDS writes, MFMAs, epilogue and the bare loop each cost 15-25 % of a launch): staging moves 16-byte vectors on
// the LDS side (a thread owns 8 adjacent channels of one k / 8 adjacent images of one row), the epilogue goes through an LDS
// transpose to 16-byte stores, and registers are kept low enough for 4 workgroups per CU.
#pragma once
#include "pconv_body.cuh"

namespace pconv {

// Operand window.  The matrix instruction flushes fp16 SUBNORMAL inputs to zero (measured: tests/test_gpu_f16x2.py, the
// out-of-window case), and the lo piece of an element below 0.125 is subnormal: such an element is carried by hi alone, 2^-11
// relative.  Operands are therefore pre-scaled by exact powers of two while they are split -- weights by 2^10, activations by
// 2^6, accumulators scaled back by 2^-16 in the epilogue (exact) -- so that the window of full accuracy is
//     1.2e-4 <= |w| < 64      2e-3 <= |x| < 1024
// (Bayesian-CNN weights are O(0.01-1), softplus / ReLU activations and [0, 1) images O(0.1-10)).  Elements below the window
// degrade gracefully -- they are small, and lose at most 2^-11 of themselves.  Measured against the float64 oracle, Gaussian
// tensors, error relative to sum|w||x|: typical |x| / |w| of 3 / 0.2 or 100 / 8 -> 1.2-2.4e-7 (the fp32 kernel: 1.3-3.9e-7);
// 0.25 / 0.01 (a few per cent of the elements under the window) -> 0.4-3.2e-6; 0.004 / 0.0003 (typical magnitude under the
// window) -> 1e-5..1e-4.  Elements above the window saturate at 65504 / scale instead of turning into inf.  Both scales can follow
// the data instead (x_amax / w_amax / y_amax below): every launch publishes max|y| for the next one, and sampled weights have an
// analytic bound; the fixed scales are the fallback of callers that pass no bounds.
constexpr float kF16Max = 65504.0f;          // (default scales: weights 2^10, activations 2^6 -- in the kernel)
constexpr unsigned kAmaxSlots = 64;          // x_amax / y_amax are arrays of this many floats; their maximum is the bound

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short f16s4 __attribute__((ext_vector_type(4)));

// x_amax / w_amax / y_amax (optional): 64 device floats whose maximum is an upper bound of max|x| (max|w|) -- that operand's scale
// then becomes the power of two that puts the bound just under 2^14 instead of the fixed 2^6 (2^10) -- and 64 device floats into
// which this launch max-reduces |y| (atomics on the bits of non-negative floats), i.e. the x_amax of the next layer: per-step,
// data-dependent, deterministic.  (For sampled weights a bound that needs no pass over them: max(|mu| + 6.66 sigma) -- Box-Muller
// on 32-bit uniforms cannot exceed sqrt(-2 ln 2^-32) = 6.66.)
template <int MT>
__global__ __launch_bounds__(kThreads) void pconv_f16x2_kernel(const PConvArgs p, const float* __restrict__ x_amax,
                                                               const float* __restrict__ w_amax, float* __restrict__ y_amax) {
    constexpr int BM = 128 * MT;
    constexpr int LDXH = BM + 32;              // 16-bit elements per image row: 64 B mod 256 -> the 4 rows of a transpose read hit disjoint banks
    constexpr int LDWH = BN + 32;              // 192 B: rows at 0 / 192 / 128 / 64 mod 256
    constexpr int XL = BM / 8, XRPP = kThreads / XL, XPASS = BK / XRPP;        // a thread moves 8 adjacent images of a row
    __shared__ __attribute__((aligned(16))) _Float16 Xp[2 * BK * LDXH];          // hi plane, lo plane ([k][b]); epilogue staging afterwards
    __shared__ __attribute__((aligned(16))) _Float16 Wp[2 * BK * LDWH];          // hi plane, lo plane ([k][n], columns swizzled)
    __shared__ int32_t kt_w[2][KCH];
    __shared__ int32_t kt_x[2][KCH];
    _Float16* const Xh = Xp;
    _Float16* const Xl = Xp + BK * LDXH;
    _Float16* const Wh = Wp;
    _Float16* const Wl = Wp + BK * LDWH;

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
    const int ex = p.x_div > 1 ? e / p.x_div : (p.x_mod > 0 ? ue % p.x_mod : e);
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

    // operand scales: fixed (2^6 / 2^10), or from a bound of the tensor's maximum: the power of two that puts it just under 2^14
    int shx = 6, shw = 10;
    auto shift_for = [&](const float* amax, int dflt) {
        const uint32_t mb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, bbb::wave_max(amax[lane])));
        const int ef = (int)((mb >> 23) & 0xffu);                    // biased exponent: the maximum is < 2^(ef - 126)
        if (ef == 0 || ef == 255) return dflt;                       // zero / inf / NaN: unknown
        const int sh = 14 - (ef - 126);
        return sh < -40 ? -40 : (sh > 40 ? 40 : sh);
    };
    if (x_amax != nullptr) shx = shift_for(x_amax, shx);
    if (w_amax != nullptr) shw = shift_for(w_amax, shw);
    const float sx = __builtin_bit_cast(float, (uint32_t)(127 + shx) << 23);
    const float sw = __builtin_bit_cast(float, (uint32_t)(127 + shw) << 23);
    const float unscale = __builtin_bit_cast(float, (uint32_t)(127 - shx - shw) << 23);      // 1 / (sx * sw), exact

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    constexpr uint32_t kWInv = 0x7FFFFFF0u;
    const uint32_t kXInv = p.x_inv;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x + (int64_t)ex * p.x_ds), 0, (int)((int64_t)p.Cin * p.H * p.W * p.B * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w + (int64_t)ew * p.w_ds), 0, (int)((int64_t)p.Cout * p.Kp * 4), 0x00020000);

    // loaders.  Weights: lane -> k (32 per half-wave), 8 ADJACENT channel rows per thread, so that both pieces of the thread's
    // 8 elements go to LDS as one 16-byte write per plane.  The 32 lanes of a half-wave write 32 rows at the same column: at a
    // 192-byte pitch that is 4 banks, so columns are XOR-swizzled in 8-element granules by the row's group of four, (k >> 2) & 7;
    // a transpose read -- 4 consecutive rows of ONE group, 4 consecutive elements -- sees the same permutation on all its rows.
    const int wkl = tid & 31, wng = (tid >> 5) * 8;
    const int wswz = ((wkl >> 2) & 7) << 3;
    const int xb8 = (tid % XL) * 8, xkr = tid / XL;
    const uint32_t xcol = (uint32_t)(b0 + xb8) * 4u;
    uint32_t wrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wrow[i] = (uint32_t)(n0 + wng + i) * (uint32_t)p.Kp * 4u;

    // One register stage + one LDS stage, as pconv_body.cuh: loads for tile t+1 are issued before tile t's MFMAs and written to
    // LDS after them.  (Measured: a second register stage -- two tiles of loads in flight -- costs a workgroup per CU in
    // registers and is 10-15 % slower on every layer; occupancy hides the latency better.)
    float wregA[8];
    f32x4 xregA[2 * XPASS];

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
    auto load_tile = [&](int tile, float (&wreg)[8], f32x4 (&xreg)[2 * XPASS]) {
        const int buf = (tile / TPC) & 1;
        const int kb = (tile % TPC) * BK;
        const uint32_t wob = (uint32_t)kt_w[buf][kb + wkl];
        uint32_t xo[XPASS];
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) xo[ps] = (uint32_t)kt_x[buf][kb + xkr + ps * XRPP] + xcol;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) {
            xreg[2 * ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps], 0, 0));
            xreg[2 * ps + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[ps] + 16u, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) wreg[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, wrow[i] + wob, 0, 0));
    };
    // the split: both pieces of every staged element go to LDS, [k][n] / [k][b] planes of 16-bit elements, 16 bytes per write
    auto store_tile = [&](float (&wreg)[8], f32x4 (&xreg)[2 * XPASS]) {
        f16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = __builtin_amdgcn_fmed3f(wreg[i] * sw, -kF16Max, kF16Max);
            h[i] = (_Float16)v;
            l[i] = (_Float16)(v - (float)h[i]);
        }
        *reinterpret_cast<f16x8*>(&Wh[wkl * LDWH + (wng ^ wswz)]) = h;
        *reinterpret_cast<f16x8*>(&Wl[wkl * LDWH + (wng ^ wswz)]) = l;
#pragma unroll
        for (int ps = 0; ps < XPASS; ++ps) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float v = __builtin_amdgcn_fmed3f(xreg[2 * ps + (c >> 2)][c & 3] * sx, -kF16Max, kF16Max);
                h[c] = (_Float16)v;
                l[c] = (_Float16)(v - (float)h[c]);
            }
            *reinterpret_cast<f16x8*>(&Xh[(xkr + ps * XRPP) * LDXH + xb8]) = h;
            *reinterpret_cast<f16x8*>(&Xl[(xkr + ps * XRPP) * LDXH + xb8]) = l;
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
    auto tr8 = [&](const _Float16* plane, int o0, int o1) {
        const f16s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(plane + o0));
        const f16s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(plane + o1));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma_tile = [&]() {
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            f16x8 ah[2], al[2], bh[MT], bl[MT];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                ah[nt] = tr8(Wh, woff[kk][0][nt], woff[kk][1][nt]);
                al[nt] = tr8(Wl, woff[kk][0][nt], woff[kk][1][nt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bh[mt] = tr8(Xh, xoff[kk][0] + mt * 32, xoff[kk][1] + mt * 32);
                bl[mt] = tr8(Xl, xoff[kk][0] + mt * 32, xoff[kk][1] + mt * 32);
            }
            // all twelve LDS reads of the step in flight before its first MFMA (left alone, hipcc feeds each MFMA just in time:
            // eight exposed LDS round trips per tile)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[nt], bh[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nt], bl[mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[nt], bh[mt], acc[nt][mt], 0, 0, 0);
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
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        p.y + (int64_t)e * p.y_ds, 0, (int)((int64_t)p.Cout * HoWo * p.B * 4), 0x00020000);
    static_assert(2 * BK * LDXH * 2 >= 4 * 32 * 36 * 4, "epilogue staging must fit in the image planes");
    float* const T = reinterpret_cast<float*>(Xp) + wave * (32 * 36);       // [32 channels][36] floats, wave-private
    float ymax = 0.0f;
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
                for (int c = 0; c < 4; ++c) o[c] = bbb::apply_act(v4[c] * unscale + bv, p.act);
                const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 4u : kOOB;
                if (off != kOOB) ymax = fmaxf(ymax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, o), yrs, off, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next tile overwrites T
        }
    if (y_amax != nullptr) {
        // thousands of waves max-reduce: into kAmaxSlots words, not one (10 k atomics on one address cost 56 us on conv1; spread
        // over 64 they disappear).  Bits of non-negative floats order like unsigned integers (an inf / NaN output ends up as a
        // huge bound: scale floor).
        ymax = bbb::wave_max(ymax);
        if (lane == 0 && ymax > 0.0f)
            __hip_atomic_fetch_max(reinterpret_cast<unsigned int*>(y_amax) + ((blockIdx.x * 4u + (unsigned)wave) & (kAmaxSlots - 1)),
                                   __builtin_bit_cast(unsigned int, ymax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace pconv
