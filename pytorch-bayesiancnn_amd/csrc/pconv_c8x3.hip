// Split-bf16 contraction over MFMA-READY operands (round 6): the pixel-major implicit GEMM of pconv_bf16x3.cuh -- fp32 accuracy on
// v_mfma_f32_32x32x16_bf16 through a = hi + mid + lo, six products per fp32 product, fp32 accumulation, small terms first -- with
// both operands laid out in memory the way the matrix instruction wants them, so that the k loop holds no operand arithmetic, no
// transposing LDS reads and no LDS traffic at all on the image side:
//   * activations travel between the layers channel-interleaved AND already split ("c8 S3"): three bf16 planes per slab,
//     [slab][3][C / 8][H][W][B][8] -- the 8 channels 8g .. 8g + 7 of an image are 16 adjacent bytes of a plane, which is exactly
//     one lane's B operand (8 consecutive k of one column): ONE 16-byte load per plane, 32 images = 512 contiguous bytes;
//   * sampled weights come TAP-MAJOR from the parameter pass (bbb_segment_t::w_tm_cin: fp32 [draw][cout][kh*kw][cin]), so that the
//     in-bounds taps of a pixel are runs of cin consecutive k -- padding taps are skipped exactly as in the fp32 kernel -- and a
//     32-k tile of 64 channels is 64 runs of 128 contiguous bytes; the tile is cut into its three bf16 pieces ONCE per workgroup
//     while it is staged (8 elements per thread and tile against 48 matrix instructions per wave) and lands in LDS as [n][k] rows
//     of 64 bytes with their four 16-byte pieces XOR-swizzled by the row: the A operand of a lane is one conflict-free ds_read_b128.
// Workgroup = one output pixel x 64 channels x 128 * MT images; every wave owns 64 channels x 32 * MT images (2 x MT accumulator
// tiles), so a 16-byte image fragment feeds 12 / 8 / 4 (hi / mid / lo plane) of the step's 24 * MT matrix instructions and a
// weight fragment 3 * MT of them.  Per k16 step and wave (MT = 2): 6 ds_read_b128 + 6 global 16-byte loads for 24 MFMAs (768 pipe
// cycles); pconv_bf16x3_kernel needed 18 transposing LDS reads, 9 LDS writes and the split arithmetic for 12.
// Replaces F.conv2d / F.linear of layers/BBB/BBBConv.py:77, layers/BBB/BBBLinear.py:70 (same contraction; not bit-identical to the
// fp32 fmaf chain: a mode, ops.gemm_mode = "bf16x3").
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"
#include "pconv_args.h"
#include "pconv_bf16x3.cuh"

namespace pconv {

#ifndef C8X3_ABLATE
#define C8X3_ABLATE 0            // experiments only (profiles/experiments/c8x3_ablate.sh): 1 no epilogue math, 2 no barriers, 3 no image loads,
#endif                           // 4 no weight staging, 5 no matrix instructions, 6 weight loads but no cut / LDS writes (1-3, 5: the generic loop)
constexpr int C8_LDA = 32;               // bf16 elements per LDS weight row: 32 k = four 16-byte pieces, no padding -- piece c of row n sits
                                         // at position c ^ ((n >> 2) & 3): the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15, 20-27} ...,
                                         // one piece index) then cover all 64 banks, and a writer's 8-lane group fills 128 contiguous bytes
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// NT x MT: 32-channel x 32-image accumulator tiles per wave.  The workgroup's weight tile is 32 * NT channel rows; its four waves own
//   * (plain form) 32 * MT images each of one output pixel: 128 * MT images per workgroup;
//   * (POOLP, "parallel window") one pixel each of a 2 x 2 pooling window, the SAME 32 * MT images: the launch also applies the
//     activation and MaxPool2d(2, 2) -- the four waves' activated tiles meet in LDS, the element-wise maximum is cut into its
//     three pieces and stored.  For layers without padding (all four pixels walk the same taps, so the weight tile serves all of
//     them); bit for bit the plain launch followed by bbb_maxpool_c8s3(2, 2).
// The contraction walks the in-bounds taps in (r, q) order and the channels of a tap in 16-channel steps, two steps (32 k) per
// weight tile; Cin % 16 == 0 (a tile may straddle two taps: the k runs of a tap are adjacent in a tap-major row, and a thread's
// 8-k piece never straddles).  Every output element sees the same sequence of matrix instructions whatever NT, MT and the form.
// LRT (local reparameterisation, layers/BBB_LRT/BBBConv.py:62-87): TWO contractions per output element -- act_mu over (x, W_mu) and
// act_var over (x^2, W_sigma^2) -- with two weight tiles, two sets of image fragments and two sets of accumulators walking the same
// taps, and the sampling epilogue out = act(act_mu + b_mu + sqrt(1e-16 + act_var + b_var) * eps), eps = the noise element of the
// output's canonical [B][Cout][Ho][Wo] index in stream (seed, call0 + draw, stream_id): exactly what pconv_gemm's LRT epilogue
// draws.  Input and output slabs hold SIX planes: the three pieces of the values, then the three pieces of their squares (the
// next layer's second operand, squared here in fp32, once per element).
template <int NT, int MT, bool OUTF32, bool POOLP, bool LRT = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NT * MT * (LRT ? 2 : 1) >= 8 ? 2 : 1)))      // (8 tiles: 128 + 128 registers)
void pconv_c8x3_kernel(const PConvArgs p) {
    constexpr int SETS = LRT ? 2 : 1;
    static_assert(!(LRT && POOLP), "LRT layers pool in a launch of their own");
    constexpr int BNW = 32 * NT;                                     // channels per workgroup
    constexpr int BMW = 32 * MT;                                     // images per wave
    constexpr int BM = POOLP ? BMW : 4 * BMW;                        // images per workgroup
    constexpr int PLANE = BNW * C8_LDA;                              // one plane of one stage
    constexpr int WPASS = (BNW + 63) / 64;                           // 64 channel rows per staging pass
    static_assert(!(OUTF32 && POOLP), "the pooled form writes c8 S3");
    constexpr int kWpBytes = 2 * SETS * 3 * PLANE * 2;
    constexpr int kLdsBytes = (POOLP && kWpBytes < 16384) ? 16384 : kWpBytes;       // (POOLP: the window exchange needs 16 KB)
    __shared__ __attribute__((aligned(16))) unsigned short Wp[kLdsBytes / 2];        // [stage][set][plane][n][k]

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
    const int n0 = (g - e * p.Ntiles) * BNW;
    const int pix = j / p.nbt;                                       // output pixel (POOLP: pooled pixel)
    const int b0 = (j - pix * p.nbt) * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = lane >> 5;

    int oh, ow;
    if constexpr (POOLP) {
        const int pw2 = p.Wo >> 1;
        const int poh = pix / pw2, pow_ = pix - poh * pw2;
        oh = 2 * poh + (wave >> 1);
        ow = 2 * pow_ + (wave & 1);
    } else {
        oh = pix / p.Wo;
        ow = pix - oh * p.Wo;
    }
    const int ihb = oh * p.sh - p.ph, iwb = ow * p.sw - p.pw;
    int r_lo = ihb < 0 ? (-ihb + p.dh - 1) / p.dh : 0;
    int q_lo = iwb < 0 ? (-iwb + p.dw - 1) / p.dw : 0;
    int r_hi = (p.H - 1 - ihb) >= 0 ? (p.H - 1 - ihb) / p.dh + 1 : 0;
    int q_hi = (p.W - 1 - iwb) >= 0 ? (p.W - 1 - iwb) / p.dw + 1 : 0;
    r_hi = r_hi < p.kh ? r_hi : p.kh;
    q_hi = q_hi < p.kw ? q_hi : p.kw;
    // Input rows / columns the caller declares all-zero (PConvArgs::vh0 .. vw1: the materialised padding of a space-to-depth block
    // image): their taps are exact zeros and are left out.  Plain form: they leave the walk.  POOLP: the four waves share one walk
    // (the weight tile is the workgroup's), so a wave only skips the loads and matrix instructions of the steps that are zero for
    // ITS pixel.
    int my_r_lo = r_lo, my_r_hi = r_hi, my_q_lo = q_lo, my_q_hi = q_hi;
    {
        const int a0 = p.vh0 - ihb, a1 = p.vh1 - 1 - ihb, c0 = p.vw0 - iwb, c1 = p.vw1 - 1 - iwb;
        const int rl = a0 > 0 ? (a0 + p.dh - 1) / p.dh : 0, rh = a1 >= 0 ? a1 / p.dh + 1 : 0;
        const int ql = c0 > 0 ? (c0 + p.dw - 1) / p.dw : 0, qh = c1 >= 0 ? c1 / p.dw + 1 : 0;
        my_r_lo = rl > my_r_lo ? rl : my_r_lo;
        my_r_hi = rh < my_r_hi ? rh : my_r_hi;
        my_q_lo = ql > my_q_lo ? ql : my_q_lo;
        my_q_hi = qh < my_q_hi ? qh : my_q_hi;
    }
    if constexpr (!POOLP) { r_lo = my_r_lo; r_hi = my_r_hi; q_lo = my_q_lo; q_hi = my_q_hi; }
    const int nr = r_hi > r_lo ? r_hi - r_lo : 0;
    const int nq = q_hi > q_lo ? q_hi - q_lo : 0;                    // (POOLP: no padding -- the same for the four waves)
    const int n16 = p.Cin >> 4;                                      // 16-channel steps per tap
    const int nsteps = nr * nq * n16;
    const int ntiles = (nsteps + 1) >> 1;

    constexpr uint32_t kOOB = 0xFFFFFFF0u;
    // weights: fp32, tap-major rows of Kp elements; thread (channel row tid / 4 [+ 64 per pass], 8 consecutive k of the tile)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w + (int64_t)ew * p.w_ds), 0, (int)((int64_t)p.Cout * p.Kp * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(LRT ? p.w2 + (int64_t)ew * p.w_ds : p.w), 0, (int)((int64_t)p.Cout * p.Kp * 4), 0x00020000);
    const int wn = tid >> 2, wk = (tid & 3) * 8;
    const uint32_t wrow = ((uint32_t)(n0 + wn) * (uint32_t)p.Kp) * 4u;                      // rows >= Cout: out of range, read as 0
    const uint32_t wpass = 64u * (uint32_t)p.Kp * 4u;
    // images: one descriptor over the slab's three planes; lane (image lrow of the wave's block, k half lk)
    const unsigned short* const xb16 = reinterpret_cast<const unsigned short*>(p.x) + (int64_t)ex * p.x_ds;
    const uint32_t xplane = (uint32_t)(p.x_ps * 2);                  // bytes per plane
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(xb16), 0, (int)((uint32_t)(3 * SETS) * xplane), 0x00020000);
    const uint32_t img_b = (uint32_t)p.B * 16u;                      // bytes per (channel group, position)
    const uint32_t grp_b = (uint32_t)(p.H * p.W) * img_b;            // bytes per channel group of 8
    // Out-of-range addressing without selects: every slab is < 1 GiB (checked by the launcher), an image column past the batch
    // carries 0x80000000 in its lane offset and a step past the last one 0x40000000 in its uniform offset -- any sum holding one
    // of the two lies in [0x40000000, 2^32) and reads zeros (no wrap: the in-range parts stay below 0x40000000).
    constexpr uint32_t kLaneInv = 0x80000000u, kStepInv = 0x40000000u;
    uint32_t xlane[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int b = b0 + (POOLP ? 0 : wave * BMW) + mt * 32 + lrow;
        xlane[mt] = b < p.B ? (uint32_t)b * 16u + (uint32_t)lk * grp_b : kLaneInv;
    }

    // image-side cursor (wave-uniform): the 16-channel step AHEAD of the ones being multiplied -- tap (rr, qq) of the pixel's
    // in-bounds rectangle, 16-channel block c16
    int sx_rr = 0, sx_qq = 0, sx_c16 = 0, sx_left = nsteps;
    bool sx_live = false;                                            // the step the last call described holds non-zero products
    auto next_step_offset = [&]() -> uint32_t {
        sx_live = false;
        if (sx_left <= 0) return kStepInv;
        const int r = r_lo + sx_rr, q = q_lo + sx_qq;
        uint32_t o = (uint32_t)(sx_c16 * 2) * grp_b + (uint32_t)((ihb + r * p.dh) * p.W + iwb + q * p.dw) * img_b;
        sx_live = !POOLP || ((r >= my_r_lo) & (r < my_r_hi) & (q >= my_q_lo) & (q < my_q_hi));
        if (!sx_live) o = kStepInv;
        --sx_left;
        if (++sx_c16 == n16) {
            sx_c16 = 0;
            if (++sx_qq == nq) { sx_qq = 0; ++sx_rr; }
        }
        return o;
    };
    // weight-side cursor (per thread: the four 8-k pieces of a tile may lie in two taps): piece (tid & 3) of the tile ahead
    int wc_rr, wc_qq, wc_ch;
    {
        const int tapi = wk / p.Cin;                                 // Cin >= 16: 0 or 1
        wc_ch = wk - tapi * p.Cin;
        wc_rr = nq > 0 ? tapi / nq : 0;
        wc_qq = nq > 0 ? tapi - wc_rr * nq : 0;
    }
    auto next_piece_offset = [&]() -> uint32_t {
        const uint32_t o = wc_rr < nr ? (uint32_t)(((r_lo + wc_rr) * p.kw + q_lo + wc_qq) * p.Cin + wc_ch) * 4u : kStepInv;
        wc_ch += 32;
        while (wc_ch >= p.Cin) {
            wc_ch -= p.Cin;
            if (++wc_qq >= nq) { wc_qq = 0; ++wc_rr; }
        }
        return o;
    };

    f32x4 wreg[SETS][WPASS][2];
    bf16x8 bfr[2][SETS][MT][3];
    // (unconditional loads with out-of-range offsets instead of branches: with branches around them the compiler drains every
    // outstanding load at every step)
    auto wload = [&]() {
        const uint32_t o = wrow + next_piece_offset();
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
            for (int ps = 0; ps < WPASS; ++ps) {
                wreg[st][ps][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(st ? wrs2 : wrs, o + (uint32_t)ps * wpass, 0, 0));
                wreg[st][ps][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(st ? wrs2 : wrs, o + (uint32_t)ps * wpass + 16u, 0, 0));
            }
    };
    bool live_next[2] = {false, false};
    auto bload = [&](int s) {
        const uint32_t u = next_step_offset();
        live_next[s] = sx_live;
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bfr[s][st][mt][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrs, xlane[mt] + u + (uint32_t)(3 * st + pl) * xplane, 0, 0));
    };
    constexpr int STAGE = SETS * 3 * PLANE;                          // 16-bit elements per stage
    auto wstore = [&](int stage) {
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
        for (int ps = 0; ps < WPASS; ++ps) {
            if ((BNW % 64) != 0 && ps == WPASS - 1 && wn >= (BNW % 64)) continue;   // 96-row tiles: the last pass holds 32 rows
            u32x4 h, m, l;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t a0, a1, a2;
                const f32x4 v = wreg[st][ps][i >> 1];
                split3_pair((i & 1) ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]}, a0, a1, a2);
                h[i] = a0; m[i] = a1; l[i] = a2;
            }
            unsigned short* const base = Wp + stage * STAGE + st * (3 * PLANE) + (wn + 64 * ps) * C8_LDA + (((tid & 3) ^ ((wn >> 2) & 3)) << 3);
            *reinterpret_cast<u32x4*>(base) = h;
            *reinterpret_cast<u32x4*>(base + PLANE) = m;
            *reinterpret_cast<u32x4*>(base + 2 * PLANE) = l;
        }
    };

    f32x16 acc[SETS][NT][MT];
#pragma unroll
    for (int st = 0; st < SETS; ++st)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < MT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[st][t][u][r] = 0.0f;

    auto step = [&](int stage, int s) {
        // channel tiles in pairs (a pair's 6 operand fragments live at a time); small terms first; the accumulators of a pair
        // take turns, so consecutive matrix instructions are independent
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
        for (int n2 = 0; n2 < NT; n2 += 2) {
            const unsigned short* const base = Wp + stage * STAGE + st * (3 * PLANE) + lrow * C8_LDA + (((2 * s + lk) ^ ((lrow >> 2) & 3)) << 3);
            constexpr int kPairMax = 2;
            bf16x8 a[kPairMax][3];
#pragma unroll
            for (int nt = 0; nt < kPairMax; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    if (n2 + nt < NT) a[nt][pl] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + (n2 + nt) * 32 * C8_LDA);
#define C8X3_TERM(PA, PB)                                                                                           \
            _Pragma("unroll") for (int nt = 0; nt < kPairMax; ++nt)                                                 \
                _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                   \
                    if (n2 + nt < NT)                                                                               \
                        acc[st][n2 + nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[nt][PA], bfr[s][st][mt][PB], acc[st][n2 + nt][mt], 0, 0, 0);
            C8X3_TERM(2, 0)
            C8X3_TERM(0, 2)
            C8X3_TERM(1, 1)
            C8X3_TERM(1, 0)
            C8X3_TERM(0, 1)
            C8X3_TERM(0, 0)
#undef C8X3_TERM
        }
    };

    // NT = 2, plain form: the weight fragments of a step are fetched from LDS while the step before is multiplied -- a plane's
    // registers are free as soon as its last term is issued (lo after the first term, mid after the fourth, hi after the sixth), so
    // the six 16-byte reads of step 1 ride under step 0's matrix instructions and only a tile's first reads are waited for
    // (measured -1.2 % per launch, profiles/r06_notes.md).
    if constexpr (NT == 2 && !POOLP && !LRT) {
      if (ntiles > 0) {
        bf16x8 af[2][3];                                             // the step's weight fragments: [channel tile][plane]
        auto aread = [&](int stage, int s, int pl) {
            const unsigned short* const base = Wp + stage * STAGE + lrow * C8_LDA + (((2 * s + lk) ^ ((lrow >> 2) & 3)) << 3) + pl * PLANE;
            af[0][pl] = *reinterpret_cast<const bf16x8*>(base);
            af[1][pl] = *reinterpret_cast<const bf16x8*>(base + 32 * C8_LDA);
        };
#define C8X3_T(PA, PB, S)                                                                                            \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                             \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                        \
                acc[0][nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[nt][PA], bfr[S][0][mt][PB], acc[0][nt][mt], 0, 0, 0);
        wload();
        bload(0);
        bload(1);
        wstore(0);
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const int stage = t & 1;
            aread(stage, 0, 2); aread(stage, 0, 0); aread(stage, 0, 1);
            if (C8X3_ABLATE != 4) wload();
            __builtin_amdgcn_sched_barrier(0);
            C8X3_T(2, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            aread(stage, 1, 2);                                      // step 1's lo fragments: their registers are free
            __builtin_amdgcn_sched_barrier(0);
            C8X3_T(0, 2, 0) C8X3_T(1, 1, 0) C8X3_T(1, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            aread(stage, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            C8X3_T(0, 1, 0) C8X3_T(0, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            aread(stage, 1, 0);
            bload(0);
            __builtin_amdgcn_sched_barrier(0);
            C8X3_T(2, 0, 1) C8X3_T(0, 2, 1) C8X3_T(1, 1, 1) C8X3_T(1, 0, 1) C8X3_T(0, 1, 1) C8X3_T(0, 0, 1)
            __builtin_amdgcn_sched_barrier(0);
            bload(1);
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE == 4) {
            } else if (C8X3_ABLATE == 6) {                           // (experiment: the weight loads without the cut / the LDS writes)
                asm volatile("" :: "v"(wreg[0][0][0]), "v"(wreg[0][0][1]));
            } else {
                wstore(stage ^ 1);
            }
            __syncthreads();
        }
#undef C8X3_T
      }
    } else if (ntiles > 0) {
        wload();
        bload(0);
        bload(1);
        wstore(0);
        __syncthreads();
        bool live0 = live_next[0], live1 = live_next[1];
        for (int t = 0; t < ntiles; ++t) {
            const int stage = t & 1;
            // issue order pinned with scheduling barriers: left alone, the compiler sinks every load below the step's matrix
            // instructions (right in front of its first use), i.e. out of their shadow
            if (C8X3_ABLATE != 4) wload();                           // (past the last tile: out-of-range offsets, zeros)
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE != 5 && live0) step(stage, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE != 3) { bload(0); live0 = live_next[0]; }
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE != 5 && live1) step(stage, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE != 3) { bload(1); live1 = live_next[1]; }
            __builtin_amdgcn_sched_barrier(0);
            if (C8X3_ABLATE != 4) wstore(stage ^ 1);                 // (after the last tile: zeros into a stage nobody reads)
            if (C8X3_ABLATE != 2) __syncthreads();
        }
    }

    // ---- epilogue ----
    const int HoWo = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias ? p.bias + (int64_t)ew * p.b_ds : p.w), 0, p.bias ? p.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>((LRT && p.bias2) ? p.bias2 + (int64_t)ew * p.b_ds : p.w), 0, (LRT && p.bias2) ? p.Cout * 4 : 0, 0x00020000);
    // one output element from its contraction result(s): BBB = act(a + bias); LRT = act(a_mu + b_mu + sqrt(1e-16 + a_var + b_var) * eps)
    // with eps the noise element of the output's canonical [B][Cout][Ho][Wo] index (pconv_body.cuh's LRT epilogue, same stream)
    const uint32_t lrt_call = LRT ? p.call0 + (p.call_dev ? *p.call_dev : 0u) + (uint32_t)ew : 0u;
    const int lrt_b0 = LRT ? p.b_off + (p.unit_div > 1 ? (ue % p.unit_div) * p.B : 0) : 0;
    auto finish = [&](float a_mu, float a_var, float b_mu, float b_var, int n, int b) -> float {
        float v = a_mu + b_mu;
        if constexpr (LRT) {
            const float var = 1e-16f + (a_var + b_var);
            if (p.sample) {
                const uint64_t idx = (uint64_t)(((int64_t)(b + lrt_b0) * p.Cout + n) * HoWo + pix);
                float z4[4];
                bbb::normal4(idx >> 2, p.stream_id, lrt_call, p.k0, p.k1, z4);
                const int c = (int)(idx & 3);
                const float z = c == 0 ? z4[0] : c == 1 ? z4[1] : c == 2 ? z4[2] : z4[3];
                v = v + __builtin_amdgcn_sqrtf(var) * z;
            }
        }
        return bbb::apply_act(v, p.act);
    };
    if constexpr (OUTF32) {
        // fp32 batch-innermost output [slab][cout][ho][wo][B] (the logits layer): a lane's accumulator registers are channels of ONE
        // image, a wave-store is 32 consecutive images of a channel row
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
            p.y + (int64_t)e * p.y_ds, 0, (int)((int64_t)p.Cout * HoWo * p.B * 4), 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (uint32_t)n * 4u, 0, 0));
                const float bv2 = LRT ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs2, (uint32_t)n * 4u, 0, 0)) : 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int b = b0 + wave * BMW + mt * 32 + lrow;
                    const float o = finish(acc[0][nt][mt][r], acc[SETS - 1][nt][mt][r], bv, bv2, n, b);
                    const uint32_t off = ((b < p.B) & (n < p.Cout)) ? (uint32_t)(((int64_t)n * HoWo + pix) * p.B + b) * 4u : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), yrs, off, 0, 0);
                }
            }
    } else {
        // c8 S3 output [slab][3][cout / 8][ho][wo][B][8]: a lane holds four consecutive channels of its image = 8 adjacent bytes of
        // each plane; a wave-store covers 32 images x 16 bytes = 512 contiguous bytes.  The next layer's operand pieces are cut
        // here, once per element.
        unsigned short* const yb16 = reinterpret_cast<unsigned short*>(p.y) + (int64_t)e * p.y_ds;
        const uint32_t yplane = (uint32_t)(p.y_ps * 2);
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb16, 0, (int)((uint32_t)(3 * SETS) * yplane), 0x00020000);
        auto store4 = [&](const float (&o)[4], int n, int b, int opix, int opixels, uint32_t plane0 = 0u) {   // channels n + 4 lk .. + 3 of image b
            uint32_t h0, m0, l0, h1, m1, l1;
            if (C8X3_ABLATE == 1) {
                h0 = m0 = l0 = __builtin_bit_cast(uint32_t, o[0] + o[1]); h1 = m1 = l1 = __builtin_bit_cast(uint32_t, o[2] + o[3]);
            } else {
                split3_pair(f32x2{o[0], o[1]}, h0, m0, l0);
                split3_pair(f32x2{o[2], o[3]}, h1, m1, l1);
            }
            const uint32_t off = ((b < p.B) & (n < p.Cout))
                ? (uint32_t)((((int64_t)(n >> 3) * opixels + opix) * p.B + b) * 16 + lk * 8) + plane0 * yplane : kOOB;
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{h0, h1}, yrs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{m0, m1}, yrs, off == kOOB ? kOOB : off + yplane, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{l0, l1}, yrs, off == kOOB ? kOOB : off + 2u * yplane, 0, 0);
        };
        if constexpr (!POOLP) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int n = n0 + nt * 32 + 8 * r4;                 // first channel of the group of 8 (this lane: + 4 lk .. + 3)
                    const f32x4 bq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n + 4 * lk) * 4u, 0, 0));
                    f32x4 bq2 = {0.0f, 0.0f, 0.0f, 0.0f};
                    if constexpr (LRT) bq2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs2, (uint32_t)(n + 4 * lk) * 4u, 0, 0));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int b = b0 + wave * BMW + mt * 32 + lrow;
                        float o[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            o[c] = C8X3_ABLATE == 1 ? acc[0][nt][mt][4 * r4 + c]
                                                    : finish(acc[0][nt][mt][4 * r4 + c], acc[SETS - 1][nt][mt][4 * r4 + c], bq[c], bq2[c], n + 4 * lk + c, b);
                        store4(o, n, b, pix, HoWo);
                        if constexpr (LRT) {                             // the next layer's second operand: the squares, once per element
                            // (mul_rn: the ROUNDED fp32 square is what gets cut -- left to the compiler the product contracts into
                            // the split's first subtraction as an fma and the pieces no longer sum to the fp32 square the fp32 path uses)
                            const float q[4] = {bbb::mul_rn(o[0], o[0]), bbb::mul_rn(o[1], o[1]), bbb::mul_rn(o[2], o[2]), bbb::mul_rn(o[3], o[3])};
                            store4(q, n, b, pix, HoWo, 3u);
                        }
                    }
                }
        } else {
            // 2 x 2 window: every wave activates its pixel's tile; per (channel tile, image tile) the four waves' 16 registers meet
            // in LDS ([wave][r4][lane] 16-byte vectors: conflict-free both ways) and wave w finishes channel group r4 = w
            f32x4* const ex4 = reinterpret_cast<f32x4*>(Wp);             // (the k loop's last barrier is behind every wave)
            const int opixels = HoWo >> 2;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int n = n0 + nt * 32 + 8 * r4;
                        const f32x4 bq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (uint32_t)(n + 4 * lk) * 4u, 0, 0));
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = bbb::apply_act(acc[0][nt][mt][4 * r4 + c] + bq[c], p.act);
                        ex4[(wave * 4 + r4) * 64 + lane] = o;
                    }
                    __syncthreads();
                    f32x4 v = ex4[(0 * 4 + wave) * 64 + lane];
#pragma unroll
                    for (int w2 = 1; w2 < 4; ++w2) {
                        const f32x4 u = ex4[(w2 * 4 + wave) * 64 + lane];
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], u[c]);
                    }
                    const float o4[4] = {v[0], v[1], v[2], v[3]};
                    store4(o4, n0 + nt * 32 + 8 * wave, b0 + mt * 32 + lrow, pix, opixels);
                    __syncthreads();
                }
            }
        }
    }
}

// fp32 batch-innermost [slabs][C][HW][B] <-> c8 S3 [slabs][3][C / 8][HW][B][8] (exact in both directions): the boundary of a chain of
// pconv_c8x3 launches (the first layer's fp32 output; a flatten that has to pass through the reference's NCHW order).  One thread
// per (slab, channel group, position, image): 8 coalesced 4-byte accesses on the fp32 side, one 16-byte vector per plane.
// pv = vectors per plane = (C / 8) * HW * B.
template <bool SQ>
__global__ __launch_bounds__(256) void c8s3_from_chwn_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t total,
                                                             int64_t pv, int64_t HW, int B) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t slab = i / pv, in = i - slab * pv;
    const int b = (int)(in % B);
    const int64_t t = in / B;
    const int64_t pos = t % HW, grp = t / HW;
    const float* xp = x + slab * pv * 8 + ((grp * 8) * HW + pos) * B + b;
    u32x4 h, m, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t a0, a1, a2;
        split3_pair(f32x2{xp[(int64_t)(2 * c) * HW * B], xp[(int64_t)(2 * c + 1) * HW * B]}, a0, a1, a2);
        h[c] = a0; m[c] = a1; l[c] = a2;
    }
    u32x4* yp = reinterpret_cast<u32x4*>(y) + slab * (SQ ? 6 : 3) * pv + in;
    yp[0] = h; yp[pv] = m; yp[2 * pv] = l;
    if constexpr (SQ) {                                              // six-plane slabs of the LRT chain: the squares' pieces behind
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a0, a1, a2;
            const float u0 = xp[(int64_t)(2 * c) * HW * B], u1 = xp[(int64_t)(2 * c + 1) * HW * B];
            split3_pair(f32x2{bbb::mul_rn(u0, u0), bbb::mul_rn(u1, u1)}, a0, a1, a2);
            h[c] = a0; m[c] = a1; l[c] = a2;
        }
        yp[3 * pv] = h; yp[4 * pv] = m; yp[5 * pv] = l;
    }
}

template <int NPL>      // planes per slab: 3, or 6 (the LRT chain's slabs: the values are the first three)
__global__ __launch_bounds__(256) void c8s3_to_chwn_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int64_t total,
                                                           int64_t pv, int64_t HW, int B) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t slab = i / pv, in = i - slab * pv;
    const int b = (int)(in % B);
    const int64_t t = in / B;
    const int64_t pos = t % HW, grp = t / HW;
    const u32x4* xp = reinterpret_cast<const u32x4*>(x) + slab * NPL * pv + in;
    const u32x4 h = xp[0], m = xp[pv], l = xp[2 * pv];
    float* yp = y + slab * pv * 8 + ((grp * 8) * HW + pos) * B + b;
#pragma unroll
    for (int c = 0; c < 4; ++c) {        // hi + mid is exact (16 significant bits), + lo gives back the fp32 value exactly
        yp[(int64_t)(2 * c) * HW * B] = (__builtin_bit_cast(float, h[c] << 16) + __builtin_bit_cast(float, m[c] << 16)) + __builtin_bit_cast(float, l[c] << 16);
        yp[(int64_t)(2 * c + 1) * HW * B] = (__builtin_bit_cast(float, h[c] & 0xFFFF0000u) + __builtin_bit_cast(float, m[c] & 0xFFFF0000u)) +
                                           __builtin_bit_cast(float, l[c] & 0xFFFF0000u);
    }
}

// [slabs][rows][cin][taps] -> [slabs][rows][taps][cin]: the tap-major weight layout from a dense one (test / boundary helper; the
// product path gets tap-major rows straight from the parameter pass, bbb_segment_t::w_tm_cin)
__global__ __launch_bounds__(256) void w_tap_major_kernel(const float* __restrict__ w, float* __restrict__ out, int64_t total, int cin, int taps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // output index
    if (i >= total) return;
    const int ci = (int)(i % cin);
    const int64_t t = i / cin;
    const int tp = (int)(t % taps);
    const int64_t row = t / taps;
    out[i] = w[(row * cin + ci) * taps + tp];
}

// Space-to-depth forms of a strided layer's operands, so that a layer the c8x3 kernel cannot take directly (AlexNet conv1: 3
// channels, 11 x 11 taps, stride 4, padding 5) becomes one it can: with m = ceil(k / s) the layer is an m x m convolution of stride 1
// without padding over the block image
//     x'[(c s + dy) s + dx][bh][bw] = xpad[c][s bh + dy][s bw + dx]          (xpad: the input behind `pad` zero rows / columns)
// with weights w'[(mr, mq)][(c s + dy) s + dx] = w[c][s mr + dy][s mq + dx] (zero where s mr + dy >= k or s mq + dx >= k) and
// C' = C s^2 rounded up to a multiple of 16 (zero channels): the same products as the strided layer plus exact zeros.  The zeros
// of the padding are materialised here, so every window walks the same taps -- which is what the pooled form needs.
// s2d_c8s3_kernel: caller's NCHW fp32 batch [nblk * Bs][C][H][W] -> c8 S3 [nblk][3][C' / 8][Hb][Wb][Bs][8]; one thread per 16-byte
// vector (8 channels of one image at one block position).  A 256-thread block covers 16 block columns x 16 images of one (block,
// channel group, block row): a wave reads 4 images x 16 columns -- per image a contiguous run of 16 * s floats per (channel, dy)
// row instead of one 16-byte piece from each of 64 images (602 KB apart on 224 x 224 inputs) -- and stores 64 contiguous bytes
// (4 images) per column and plane.
template <bool SQ>
__global__ __launch_bounds__(256) void s2d_c8s3_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t pv,
                                                      int C, int H, int W, int s, int pad, int Hb, int Wb, int Bs, int nbt) {
    const int bwt = blockIdx.x / nbt, bt = blockIdx.x - bwt * nbt;
    const int bw = bwt * 16 + (threadIdx.x & 15), b = bt * 16 + (threadIdx.x >> 4);
    const int bh = blockIdx.y % Hb, grp = blockIdx.y / Hb;
    const int64_t blk = blockIdx.z;
    if (bw >= Wb || b >= Bs) return;
    const float* xi = x + (blk * Bs + b) * (int64_t)C * H * W;
    float v[8];
    int cd = (grp * 8) / s, dx = grp * 8 - cd * s;                  // channel c' = cd * s + dx, cd = c * s + dy
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int c = cd / s, dy = cd - c * s;
        const int ih = s * bh + dy - pad, iw = s * bw + dx - pad;
        v[c8] = (c < C && ih >= 0 && ih < H && iw >= 0 && iw < W) ? xi[((int64_t)c * H + ih) * W + iw] : 0.0f;
        if (++dx == s) { dx = 0; ++cd; }
    }
    u32x4 h, m, l;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t a0, a1, a2;
        split3_pair(f32x2{v[2 * c], v[2 * c + 1]}, a0, a1, a2);
        h[c] = a0; m[c] = a1; l[c] = a2;
    }
    u32x4* yp = reinterpret_cast<u32x4*>(y) + blk * (SQ ? 6 : 3) * pv + (((int64_t)grp * Hb + bh) * Wb + bw) * Bs + b;
    yp[0] = h; yp[pv] = m; yp[2 * pv] = l;
    if constexpr (SQ) {                                              // LRT first layer: the squares' pieces behind the values'
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t a0, a1, a2;
            split3_pair(f32x2{bbb::mul_rn(v[2 * c], v[2 * c]), bbb::mul_rn(v[2 * c + 1], v[2 * c + 1])}, a0, a1, a2);
            h[c] = a0; m[c] = a1; l[c] = a2;
        }
        yp[3 * pv] = h; yp[4 * pv] = m; yp[5 * pv] = l;
    }
}

// w [rows][C][k][k] -> out [rows][m * m][Cp] (tap-major rows of the space-to-depth layer); one thread per output element
__global__ __launch_bounds__(256) void w_s2d_tap_major_kernel(const float* __restrict__ w, float* __restrict__ out, int64_t total, int C, int k,
                                                             int s, int m, int Cp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cp = (int)(i % Cp);
    int64_t t = i / Cp;
    const int mq = (int)(t % m); t /= m;
    const int mr = (int)(t % m);
    const int64_t row = t / m;
    const int dx = cp % s, dy = (cp / s) % s, c = cp / (s * s);
    const int r = s * mr + dy, q = s * mq + dx;
    out[i] = (c < C && r < k && q < k) ? w[((row * C + c) * k + r) * k + q] : 0.0f;
}

}  // namespace pconv

using namespace pconv;

namespace {
template <int NT, int MT>
void launch_c8x3(bool of32, bool poolp, bool lrt, dim3 grid, hipStream_t st, const PConvArgs& a) {
    const dim3 block(kThreads);
    if constexpr (NT == 2 && MT == 1) {        // (LRT: two accumulator sets -- 32 images per wave; 64 would spill)
        if (lrt) {
            if (of32) hipLaunchKernelGGL((pconv_c8x3_kernel<NT, MT, true, false, true>), grid, block, 0, st, a);
            else      hipLaunchKernelGGL((pconv_c8x3_kernel<NT, MT, false, false, true>), grid, block, 0, st, a);
            return;
        }
    }
    if (poolp)     hipLaunchKernelGGL((pconv_c8x3_kernel<NT, MT, false, true>), grid, block, 0, st, a);
    else if (of32) hipLaunchKernelGGL((pconv_c8x3_kernel<NT, MT, true, false>), grid, block, 0, st, a);
    else           hipLaunchKernelGGL((pconv_c8x3_kernel<NT, MT, false, false>), grid, block, 0, st, a);
}
}  // namespace

namespace {
struct LrtNoise { const float* w_var; const float* b_var; uint64_t seed; uint32_t call0, stream_id; int sample; const uint32_t* call_dev; };

int c8x3_launch(const bbb_conv_desc_t* d, const void* x, const float* w, const float* bias, void* y, uint32_t flags, const LrtNoise* lrt,
                void* stream) {
    constexpr uint32_t kKnown = BBB_C8X3_OUT_F32 | BBB_C8X3_TILE128 | BBB_C8X3_TILE256 | BBB_C8X3_POOL | BBB_C8X3_NT_MASK | BBB_C8X3_ZERO_MASK;
    if (d == nullptr || x == nullptr || w == nullptr || y == nullptr || (flags & ~kKnown) != 0 ||
        ((flags & BBB_C8X3_TILE128) && (flags & BBB_C8X3_TILE256)))
        return BBB_EINVAL;
    if (d->batch <= 0 || d->cin <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 ||
        d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0 || d->dil_h <= 0 || d->dil_w <= 0 || d->draws <= 0 || d->act < 0 || d->act > 2 ||
        d->pool != 0 || d->w_row_pitch != 0 || d->w_tap_major != 0)
        return BBB_EINVAL;
    const bool of32 = (flags & BBB_C8X3_OUT_F32) != 0;
    const bool poolp = (flags & BBB_C8X3_POOL) != 0;
    const int nt_force = (int)((flags & BBB_C8X3_NT_MASK) >> BBB_C8X3_NT_SHIFT);
    if (nt_force == 1 || nt_force > 4 || (poolp && of32)) return BBB_EINVAL;
    if (lrt != nullptr) {
        // LRT: one (mu, sigma^2) weight pair for every slab; 64-channel tiles; pooling is a launch of its own (bbb_maxpool_c8s3_sq)
        if (lrt->w_var == nullptr || poolp || (nt_force != 0 && nt_force != 2) || d->w_draw_stride != 0 || d->b_draw_stride != 0 ||
            ((bias == nullptr) != (lrt->b_var == nullptr)))
            return BBB_EINVAL;
        if ((((uintptr_t)lrt->w_var) & 15u) != 0 || (((uintptr_t)lrt->b_var) & (of32 ? 3u : 15u)) != 0) return BBB_EALIGN;
    }
    const int sets = lrt != nullptr ? 2 : 1;
    if (d->cin % 16 != 0 || d->batch % 4 != 0 || (!of32 && d->cout % 8 != 0)) return BBB_ESHAPE;
    const int ho = (d->h + 2 * d->pad_h - d->dil_h * (d->kh - 1) - 1) / d->stride_h + 1;
    const int wo = (d->w + 2 * d->pad_w - d->dil_w * (d->kw - 1) - 1) / d->stride_w + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    // the pooled form: no padding (the four pixels of a window walk the same taps), even output height and width
    if (poolp && (d->pad_h != 0 || d->pad_w != 0 || ho % 2 != 0 || wo % 2 != 0)) return BBB_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0 || (((uintptr_t)w) & 15u) != 0 || (((uintptr_t)bias) & (of32 ? 3u : 15u)) != 0) return BBB_EALIGN;
    PConvArgs a = {};
    a.B = d->batch; a.Cin = d->cin; a.H = d->h; a.W = d->w; a.Cout = d->cout; a.kh = d->kh; a.kw = d->kw;
    a.sh = d->stride_h; a.sw = d->stride_w; a.ph = d->pad_h; a.pw = d->pad_w; a.dh = d->dil_h; a.dw = d->dil_w;
    a.Ho = ho; a.Wo = wo; a.K = d->cin * d->kh * d->kw; a.Kp = a.K; a.khkw = d->kh * d->kw; a.act = d->act;
    a.pool = poolp ? 1 : 0;
    {   // rows / columns of the input the caller declares all-zero (BBB_C8X3_ZERO_* nibbles of flags)
        const int lh = (int)((flags >> 8) & 15u), lw = (int)((flags >> 12) & 15u), th = (int)((flags >> 16) & 15u), tw = (int)((flags >> 20) & 15u);
        if (lh + th >= a.H || lw + tw >= a.W) return BBB_EINVAL;
        a.vh0 = lh; a.vh1 = a.H - th; a.vw0 = lw; a.vw1 = a.W - tw;
    }
    a.x_ps = (int64_t)a.Cin * a.H * a.W * a.B;
    a.y_ps = (int64_t)a.Cout * ho * wo * a.B / (poolp ? 4 : 1);
    // slabs are addressed through 32-bit buffer offsets (three -- LRT: six -- planes of 2-byte elements, or fp32 outputs)
    if (6 * sets * a.x_ps >= 0x3FFF0000LL || 6 * sets * a.y_ps >= 0x3FFF0000LL || ((int64_t)a.Cout + 128) * a.K * 4 >= 0x3FFF0000LL) return BBB_ESHAPE;
    if (d->x_draw_stride != 0 && d->x_draw_stride < 3 * sets * a.x_ps) return BBB_EINVAL;
    if (d->w_draw_stride % 4 != 0 || (!of32 && d->b_draw_stride % 4 != 0) || d->x_draw_stride % 8 != 0) return BBB_EALIGN;
    a.x_ds = d->x_draw_stride; a.w_ds = d->w_draw_stride; a.b_ds = d->b_draw_stride;
    a.y_ds = of32 ? a.y_ps : 3 * sets * a.y_ps;
    if (d->unit_div < 0 || d->unit_off < 0 || d->x_unit_mod < 0) return BBB_EINVAL;
    if (d->unit_div > 1 && d->unit_off >= d->unit_div) return BBB_EINVAL;
    if (d->x_unit_mod > 0 && d->x_unit_mod != d->unit_div) return BBB_EINVAL;
    if (d->x_unit_div < 0 || d->x_unit_off < 0 || (d->x_unit_div > 1 && (d->unit_div > 1 || d->x_unit_off >= d->x_unit_div)) ||
        (d->x_unit_div <= 1 && d->x_unit_off != 0))
        return BBB_EINVAL;
    a.unit_div = d->unit_div; a.unit_off = d->unit_div > 1 ? d->unit_off : 0; a.x_mod = d->x_unit_mod;
    a.x_div = d->x_unit_div; a.x_off = d->x_unit_off;
    a.x = static_cast<const float*>(x); a.w = w; a.bias = bias; a.y = static_cast<float*>(y);
    if (lrt != nullptr) {
        a.w2 = lrt->w_var; a.bias2 = lrt->b_var;
        a.k0 = (uint32_t)lrt->seed; a.k1 = (uint32_t)(lrt->seed >> 32); a.call0 = lrt->call0; a.stream_id = lrt->stream_id;
        a.sample = lrt->sample ? 1 : 0; a.call_dev = lrt->call_dev; a.b_off = d->b_offset;
    }
    const int64_t pixels = (int64_t)ho * wo / (poolp ? 4 : 1);
    // Tile shape (the MFMA sequence per output element, hence every output bit, does not depend on it).  Channels per workgroup
    // 32 * NT: NT = 2 unless forced -- wider tiles halve the image fragments' trips through the vector memory path per matrix
    // instruction but hold 96 / 128 accumulation registers (two waves per SIMD instead of three), and measured slower on every
    // AlexNet layer but conv3 (profiles/r06_notes.md: conv2 504 / 515 / 603 us for NT = 2 / 3 / 4 at 40 slabs).  Images per wave
    // 32 * MT: 64, or 32 when the launch would otherwise leave the chip less than ~two rounds of workgroups.
    int nt = 2, mt = 2;
    {
        static const int env_nt = [] { const char* s = getenv("BBB_C8X3_NT"); return s ? atoi(s) : 0; }();   // (experiments)
        if (nt_force) nt = nt_force;
        else if (env_nt >= 2 && env_nt <= 4) nt = env_nt;
        const int64_t bm2 = (poolp ? 32 : 128) * 2;
        const int64_t items2 = (int64_t)d->draws * ((a.Cout + 32 * nt - 1) / (32 * nt)) * pixels * ((a.B + bm2 - 1) / bm2);
        if (flags & BBB_C8X3_TILE128) mt = 1;
        else if (flags & BBB_C8X3_TILE256) mt = 2;
        else if (a.B <= bm2 / 2 || items2 < 1024) mt = 1;
        if (lrt != nullptr) mt = 1;                                  // two accumulator sets: 32 images per wave
    }
    const int bnw = 32 * nt, bm = (poolp ? 32 : 128) * mt;
    a.Ntiles = (a.Cout + bnw - 1) / bnw;
    a.G = a.Ntiles * d->draws;
    a.nbt = (a.B + bm - 1) / bm;
    const int64_t mtiles = pixels * a.nbt;
    if (mtiles > 0x7fffffffLL) return BBB_ESHAPE;
    a.Mtiles = (int)mtiles;
    const int64_t per = ((int64_t)a.G * mtiles + 7) / 8;
    if (8 * per > 0x7fffffffLL) return BBB_ESHAPE;
    a.per_xcd = (int32_t)per;
    const dim3 grid((unsigned)(8 * per));
    hipStream_t st = (hipStream_t)stream;
    const bool is_lrt = lrt != nullptr;
    switch (nt * 10 + mt) {
        case 21: launch_c8x3<2, 1>(of32, poolp, is_lrt, grid, st, a); break;
        case 22: launch_c8x3<2, 2>(of32, poolp, is_lrt, grid, st, a); break;
        case 31: launch_c8x3<3, 1>(of32, poolp, false, grid, st, a); break;
        case 32: launch_c8x3<3, 2>(of32, poolp, false, grid, st, a); break;
        case 41: launch_c8x3<4, 1>(of32, poolp, false, grid, st, a); break;
        default: launch_c8x3<4, 2>(of32, poolp, false, grid, st, a); break;
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bbb_conv2d_c8x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w, const float* bias, void* y, uint32_t flags,
                                   void* stream) {
    return c8x3_launch(d, x, w, bias, y, flags, nullptr, stream);
}

extern "C" int bbb_lrt_conv2d_c8x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w_mu, const float* w_var, const float* b_mu,
                                       const float* b_var, void* y, uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                                       const uint32_t* call_dev, uint32_t flags, void* stream) {
    if (w_var == nullptr) return BBB_EINVAL;
    const LrtNoise n = {w_var, b_var, seed, call0, stream_id, sample, call_dev};
    return c8x3_launch(d, x, w_mu, b_mu, y, flags, &n, stream);
}

extern "C" int bbb_c8s3_convert(const void* src, void* dst, int64_t slabs, int channels, int64_t positions, int batch, int to_c8s3,
                                void* stream) {
    if (src == nullptr || dst == nullptr || slabs <= 0 || channels <= 0 || positions <= 0 || batch <= 0) return BBB_EINVAL;
    if (channels % 8 != 0) return BBB_ESHAPE;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) != 0) return BBB_EALIGN;
    const int64_t pv = (int64_t)(channels / 8) * positions * batch, total = slabs * pv;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    // to_c8s3: 1 = fp32 -> three planes, 2 = fp32 -> six planes (values + squares: the LRT chain), 0 / 3 = three / six planes -> fp32
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    const float* sf = static_cast<const float*>(src);
    const unsigned short* su = static_cast<const unsigned short*>(src);
    if (to_c8s3 == 1)      hipLaunchKernelGGL(c8s3_from_chwn_kernel<false>, grid, block, 0, st, sf, static_cast<unsigned short*>(dst), total, pv, positions, batch);
    else if (to_c8s3 == 2) hipLaunchKernelGGL(c8s3_from_chwn_kernel<true>, grid, block, 0, st, sf, static_cast<unsigned short*>(dst), total, pv, positions, batch);
    else if (to_c8s3 == 0) hipLaunchKernelGGL(c8s3_to_chwn_kernel<3>, grid, block, 0, st, su, static_cast<float*>(dst), total, pv, positions, batch);
    else if (to_c8s3 == 3) hipLaunchKernelGGL(c8s3_to_chwn_kernel<6>, grid, block, 0, st, su, static_cast<float*>(dst), total, pv, positions, batch);
    else return BBB_EINVAL;
    return (int)hipGetLastError();
}

namespace {
int s2d_launch(const float* x, void* y, int64_t blocks, int batch, int channels, int h, int w, int k, int stride, int pad, bool sq, void* stream) {
    if (x == nullptr || y == nullptr || blocks <= 0 || batch <= 0 || channels <= 0 || h <= 0 || w <= 0 || k <= 0 || stride <= 0 || pad < 0)
        return BBB_EINVAL;
    if ((((uintptr_t)x) & 3u) != 0 || (((uintptr_t)y) & 15u) != 0) return BBB_EALIGN;
    const int ho = (h + 2 * pad - k) / stride + 1, wo = (w + 2 * pad - k) / stride + 1;
    if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
    const int m = (k + stride - 1) / stride;
    const int hb = ho + m - 1, wb = wo + m - 1;
    const int cp = (channels * stride * stride + 15) / 16 * 16;
    const int64_t pv = (int64_t)(cp / 8) * hb * wb * batch;
    const int nbt = (batch + 15) / 16;
    const int64_t gx = (int64_t)((wb + 15) / 16) * nbt, gy = (int64_t)(cp / 8) * hb;
    if (gx > 0x7fffffffLL || gy > 65535 || blocks > 65535) return BBB_ESHAPE;
    if (sq) hipLaunchKernelGGL(s2d_c8s3_kernel<true>, dim3((unsigned)gx, (unsigned)gy, (unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                               static_cast<unsigned short*>(y), pv, channels, h, w, stride, pad, hb, wb, batch, nbt);
    else    hipLaunchKernelGGL(s2d_c8s3_kernel<false>, dim3((unsigned)gx, (unsigned)gy, (unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                               static_cast<unsigned short*>(y), pv, channels, h, w, stride, pad, hb, wb, batch, nbt);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int bbb_s2d_c8s3(const float* x, void* y, int64_t blocks, int batch, int channels, int h, int w, int k, int stride, int pad,
                            void* stream) {
    return s2d_launch(x, y, blocks, batch, channels, h, w, k, stride, pad, false, stream);
}

extern "C" int bbb_s2d_c8s3sq(const float* x, void* y, int64_t blocks, int batch, int channels, int h, int w, int k, int stride, int pad,
                              void* stream) {
    return s2d_launch(x, y, blocks, batch, channels, h, w, k, stride, pad, true, stream);
}

extern "C" int bbb_w_s2d_tap_major(const float* w, float* out, int64_t rows, int channels, int k, int stride, void* stream) {
    if (w == nullptr || out == nullptr || rows <= 0 || channels <= 0 || k <= 0 || stride <= 0) return BBB_EINVAL;
    if ((((uintptr_t)w | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int m = (k + stride - 1) / stride;
    const int cp = (channels * stride * stride + 15) / 16 * 16;
    const int64_t total = rows * m * m * cp;
    const int64_t nb = (total + 255) / 256;
    if (nb > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(w_s2d_tap_major_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w, out, total, channels, k, stride, m, cp);
    return (int)hipGetLastError();
}

extern "C" int bbb_w_tap_major(const float* w, float* out, int64_t rows, int cin, int taps, void* stream) {
    if (w == nullptr || out == nullptr || rows <= 0 || cin <= 0 || taps <= 0) return BBB_EINVAL;
    if ((((uintptr_t)w | (uintptr_t)out) & 3u) != 0) return BBB_EALIGN;
    const int64_t total = rows * cin * taps;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return BBB_ESHAPE;
    hipLaunchKernelGGL(w_tap_major_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, out, total, cin, taps);
    return (int)hipGetLastError();
}
