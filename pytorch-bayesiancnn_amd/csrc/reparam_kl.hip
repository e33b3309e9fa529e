// Fused reparameterisation + KL for Bayes-by-Backprop layers on gfx950 (MI355X).
//
// One pass over (mu, rho) of up to 16 parameter tensors and E Monte-Carlo draws:
//   sigma = log1p(exp(rho));  w[e] = mu + sigma * eps[e]  (eps from on-chip Philox4x32-7);
//   KL term of the reference's call order (prior as "q", posterior as "p"), reduced per thread in
//   fp64 -> wave shuffles -> LDS -> one partial per 1024-element chunk, published BEFORE the block starts its draws;
//   one extra block of the same launch waits for all partials and adds them in index order (bitwise reproducible),
//   hidden behind the other blocks' draw loops.  HBM traffic per element: 8 B read + 4 B written per draw; eps and
//   KL cost none.
// Replaces layers/BBB/BBBConv.py:63-70,79-83, layers/BBB/BBBLinear.py:56-63,72-76,
// layers/BBB_LRT/BBBConv.py:64-69,83-87, layers/BBB_LRT/BBBLinear.py:58-63,75-79, metrics.py:27-29,
// layers/misc.py:20-23 of the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = kThreads * 4;                         // elements per (block, group slot): 4 per thread
// A KL partial slot that has not been published yet holds this bit pattern (an all-ones NaN; the scratch is filled with
// 0xFF bytes once by the host, and every launch re-arms the slots it consumed).
constexpr unsigned long long kUnpublished = 0xFFFFFFFFFFFFFFFFull;
constexpr int kSpinBudget = 1 << 20;                         // bounded wait (~0.1 s): a lost partial gives NaN, never a hang

typedef bbb::bbb_f32x4 f32x4;
using bbb::bbb_f32x2;
using bbb::bbb_f32x4;

struct ReparamArgs {
    bbb_segment_t seg[BBB_MAX_SEGMENTS];
    float* grad_mu[BBB_MAX_SEGMENTS];    // backward only
    float* grad_rho[BBB_MAX_SEGMENTS];   // backward only
    int32_t chunk_begin[BBB_MAX_SEGMENTS + 1];
    int32_t nseg;
    int32_t draws;
    float prior_mu, prior_sigma;
    uint32_t k0, k1, call0, flags;
    unsigned long long* partials;   // [chunks] per-chunk KL partials (fp64 bits); kUnpublished between launches
    float* out32;          // KL sum (fp32) or NULL
    double* out64;         // KL sum (fp64) or NULL
    int32_t n_chunks;      // publishers = partials to add up
    int32_t sum_block;     // index of the block that only sums the partials (-1: no KL wanted)
    int32_t n_small;       // leading blocks that own ONE draw of one of the last chunks (see the launcher); 0 = none
    int32_t small_chunk0;  // first chunk handled by small blocks
    const float* gkl;
    const uint32_t* call_dev;
    // tap-major fp32 outputs (bbb_segment_t::w_tm_cin): the draws of such a segment are produced by `tm_blocks` extra blocks at the
    // FRONT of the grid, each owning tm_cg (row, input channel) pairs = tm_cg * taps consecutive source elements
    int32_t tm_begin[BBB_MAX_SEGMENTS + 1];
    int32_t tm_cg[BBB_MAX_SEGMENTS];
    int32_t tm_blocks;
};

// KL term in the reference's form (metrics.py:28 with the call-site argument order):
//   0.5 * (2*log(sig_p/sig_q) - 1 + (sig_q/sig_p)^2 + ((mu_p - mu_q)/sig_p)^2),  q = prior scalars, p = posterior.
// One Newton-refined reciprocal instead of three IEEE divisions and the hardware log2 instead of logf: each term
// is within ~1e-7 absolute / 2 ulp relative of the torch expression, far inside the 1e-6 relative bound on the SUM
// (terms are O(100) and the errors are unbiased).  l2s0 = log2(prior sigma), is0 = 1 / prior sigma.
__device__ __forceinline__ float kl_term(float mu, float sigma, float mu0, float sig0, float l2s0, float is0, bool textbook) {
    const float l2 = __builtin_amdgcn_logf(sigma) - l2s0;            // log2(sigma / sig0)
    float t;
    if (!textbook) {
        const float is = bbb::rcp_newton(sigma);
        const float a = sig0 * is;
        const float b = (mu - mu0) * is;
        t = fmaf(1.3862943611198906f, l2, -1.0f) + a * a + b * b;
    } else {  // KL(q||p): q = posterior, p = prior (opt-in)
        const float a = sigma * is0;
        const float b = (mu0 - mu) * is0;
        t = fmaf(-1.3862943611198906f, l2, -1.0f) + a * a + b * b;
    }
    return 0.5f * t;
}

// kl_term on four elements with packed math (same operations per element, same bits); returns the terms, unsummed.
__device__ __forceinline__ bbb_f32x4 kl_term4(bbb_f32x4 mu, bbb_f32x4 sigma, float mu0, float sig0, float l2s0, float is0, bool textbook) {
    using bbb::pk_fma;
    using bbb::pk_splat;
    bbb_f32x4 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bbb_f32x2 m = bbb_f32x2{mu[2 * h], mu[2 * h + 1]};
        const bbb_f32x2 sg = bbb_f32x2{sigma[2 * h], sigma[2 * h + 1]};
        const bbb_f32x2 l2 = bbb_f32x2{__builtin_amdgcn_logf(sg.x), __builtin_amdgcn_logf(sg.y)} - pk_splat(l2s0);
        bbb_f32x2 t;
        if (!textbook) {
            const bbb_f32x2 r = bbb_f32x2{__builtin_amdgcn_rcpf(sg.x), __builtin_amdgcn_rcpf(sg.y)};
            const bbb_f32x2 is = pk_fma(pk_fma(-sg, r, pk_splat(1.0f)), r, r);          // rcp_newton
            const bbb_f32x2 a = pk_splat(sig0) * is;
            const bbb_f32x2 b = (m - pk_splat(mu0)) * is;
            t = pk_fma(pk_splat(1.3862943611198906f), l2, pk_splat(-1.0f)) + a * a + b * b;
        } else {
            const bbb_f32x2 a = sg * pk_splat(is0);
            const bbb_f32x2 b = (pk_splat(mu0) - m) * pk_splat(is0);
            t = pk_fma(pk_splat(-1.3862943611198906f), l2, pk_splat(-1.0f)) + a * a + b * b;
        }
        t = t * pk_splat(0.5f);
        out[2 * h] = t.x;
        out[2 * h + 1] = t.y;
    }
    return out;
}

// w = mu + eps*sigma as ONE fused multiply-add in every kernel of this file (so the result does not depend on which
// kernel variant a launch takes).  The reference rounds twice (layers/BBB/BBBConv.py:65: a multiply, then an add); the
// FMA is within half an ulp of that, three orders of magnitude below the 2e-5 the hardware sin/cos/log leave in eps.
__device__ __forceinline__ float sample_w(float mu, float z, float sigma) { return fmaf(z, sigma, mu); }

__device__ __forceinline__ uint16_t bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

__device__ __forceinline__ int find_segment(const ReparamArgs& a, int chunk) {
    int s = 0;
    while (s + 1 < a.nseg && chunk >= a.chunk_begin[s + 1]) ++s;
    return s;
}

// KL across blocks without a second launch, without fences and without counters.  Every chunk's owner reduces its terms
// (fp64: thread -> wave shuffles -> LDS) and lane 0 PUBLISHES the partial with ONE 8-byte write-through (agent-scope,
// `sc1`) store into the chunk's slot -- a data-tagged granule: the slot holds kUnpublished until then, and an aligned
// 8-byte store is observed whole.  No drain, no ticket: the publisher does not wait for anything.  (Round-2 history,
// measured at model size: a release fence per block = `buffer_wbl2` of the whole dirty L2 -> 155 us; store + drain + ONE
// atomic counter -> 33 us, the counter word taking ~88 increments/us; 64 striped counters -> 14 us fixed; this -> see
// profiles/r02_notes.md.)
__device__ __forceinline__ void publish_partial(const ReparamArgs& a, int chunk, double kl_acc, double* sm /* [kThreads] */) {
    const int tid = threadIdx.x;
    kl_acc = bbb::wave_sum(kl_acc);
    const int lane = tid & (bbb::kWave - 1), wv = tid / bbb::kWave;
    if (lane == 0) sm[wv] = kl_acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < kThreads / bbb::kWave; ++i) t += sm[i];
        unsigned long long bits = __builtin_bit_cast(unsigned long long, t);
        if (bits == kUnpublished) bits = 0x7FF8000000000000ull;      // a NaN partial stays a NaN, but never looks unpublished
        __hip_atomic_store(a.partials + chunk, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The summing block (last block index of the launch, so every publisher has been dispatched before it): thread t waits
// for, consumes and re-arms slots t, t+256, ... IN THAT ORDER, then an LDS tree -- the fixed summation order of round 1's
// separate finish kernel, so the KL bits did not change and do not depend on timing.  Slots are read with agent-scope
// (cache-bypassing) loads, the counterpart of the publishers' write-through stores.
__device__ __forceinline__ void sum_partials(const ReparamArgs& a, double* sm /* [kThreads] */) {
    const int tid = threadIdx.x;
    double t = 0.0;
    int budget = kSpinBudget;
    for (int i = tid; i < a.n_chunks; i += kThreads) {
        unsigned long long v = __hip_atomic_load(a.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (v == kUnpublished && budget > 0) {
            __builtin_amdgcn_s_sleep(4);
            --budget;
            v = __hip_atomic_load(a.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        t += __builtin_bit_cast(double, v);                       // a slot that never arrived adds NaN
        __hip_atomic_store(a.partials + i, kUnpublished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    sm[tid] = t;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        if (a.out32) *a.out32 = (float)sm[0];
        if (a.out64) *a.out64 = sm[0];
    }
}

// Tap-major fp32 weights for bbb_conv2d_c8x3_fwd (segment field w_tm_cin = C, w_taps = T): w[draw][row][tap][ci] instead of the
// tensor's own [row][ci][tap].  The noise element of a weight is its index in the tensor's OWN order and one Philox call covers
// four consecutive ones, so a thread still owns 4 consecutive SOURCE elements; the block owns `cg` consecutive (row, ci) pairs =
// cg * T consecutive source elements (cg % 8 == 0, cg * T <= 1024: 40 channels of a 5 x 5 layer, 112 of a 3 x 3 one), every
// draw goes through LDS once -- written in source order (one 16-byte write per thread), read with stride T (odd: conflict-light)
// as 4 consecutive channels of one tap -- and leaves as 16-byte stores, runs of cg * 4 contiguous bytes per tap.  Double-buffered:
// one barrier per draw.  Sigma / KL of these elements are NOT computed here: the segment's ordinary 1024-element chunks do that
// (and skip the draws), so the KL partials, their slots and the summation order are exactly those of a dense launch.
template <bool NT>
__device__ __forceinline__ void tm_block(const ReparamArgs& a, int tb, float* lds /* [2][kChunk] */) {
    int s = 0;
    while (s + 1 < a.nseg && tb >= a.tm_begin[s + 1]) ++s;
    const bbb_segment_t sg = a.seg[s];
    const int T = (int)sg.w_taps, C = (int)sg.w_tm_cin, cg = a.tm_cg[s];
    const int64_t rc_total = sg.n / T;                           // rows * C
    const int64_t rc0 = (int64_t)(tb - a.tm_begin[s]) * cg;
    const int nrc = (rc_total - rc0) < cg ? (int)(rc_total - rc0) : cg;
    const int cnt = nrc * T;                                     // source elements of this block (multiple of 8)
    const int64_t i0 = rc0 * T;
    const int tid = threadIdx.x, j = 4 * tid;
    const bool active = j < cnt;                                 // cnt / 4 threads: one source group AND one destination quad each
    const uint32_t call0 = a.call0 + (a.call_dev ? *a.call_dev : 0u);
    f32x4 m4 = {0.f, 0.f, 0.f, 0.f}, r4 = m4;
    if (active) {
        m4 = *reinterpret_cast<const f32x4*>(sg.mu + i0 + j);
        r4 = *reinterpret_cast<const f32x4*>(sg.rho + i0 + j);
    }
    const uint64_t g = (uint64_t)(i0 + j) >> 2;
    float z[4];
    bbb::normal4(g, sg.stream_id, call0, a.k0, a.k1, z);         // first draw's noise while the loads are in flight
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 sig = bbb::softplus_ref4(r4);
    // destination quad of this thread: tap-slowest, so that consecutive lanes store consecutive 16-byte pieces
    const int nq4 = nrc >> 2;
    const int tap = active ? tid / nq4 : 0, quad = active ? tid - tap * nq4 : 0;
    const int64_t rc = rc0 + 4 * quad;
    const int64_t row = rc / C;
    const int ci = (int)(rc - row * C);
    float* wp = sg.w + row * (int64_t)T * C + (int64_t)tap * C + ci;
    const int src = 4 * quad * T + tap;
    for (int e = 0;;) {
        float* buf = lds + (e & 1) * kChunk;
        if (active) {
            f32x4 w4;
#pragma unroll
            for (int c = 0; c < 4; ++c) w4[c] = sample_w(m4[c], z[c], sig[c]);
            *reinterpret_cast<f32x4*>(buf + j) = w4;
        }
        __syncthreads();
        if (active) {
            const f32x4 o = {buf[src], buf[src + T], buf[src + 2 * T], buf[src + 3 * T]};
            if (NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(wp));
            else    *reinterpret_cast<f32x4*>(wp) = o;
        }
        if (++e >= a.draws) break;
        wp += sg.draw_stride;
        bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e, a.k0, a.k1, z);
    }
}

// The hot instantiation: on-chip noise, dense fp32 outputs.  Straight-line draw loop: no loads, no waits -- a thread's (mu, rho) vectors are loaded
// once, sigma and the KL term are computed once (and published), and every draw is one Philox call, two Box-Muller pairs,
// four FMAs and one 16-byte store; a thread may have all of its draws' stores in flight at once.  The tail of an aligned tensor (n % 4 != 0) still LOADS a full
// 16-byte vector -- the aligned granule contains valid bytes, so it cannot fault -- and stores its 1..3 elements one by one.
// GPT = 16-byte groups per thread (1 at model size, 4 for tensors beyond ~16M elements); NT = non-temporal stores of w
// (streaming sizes: the GEMM will not find w in cache anyway).
// Block kinds (b = blockIdx.x): b == sum_block: sum_partials only.  b < n_small: ONE draw of one of the last chunks
// (see the launcher: it evens out a launch that is slightly larger than one round of resident blocks).  Otherwise: a whole
// chunk, all draws.
template <int GPT, bool NT>
__global__ __launch_bounds__(kThreads) void reparam_kl_fast_kernel(const ReparamArgs a) {
    __shared__ double sm[kThreads];
    __shared__ __attribute__((aligned(16))) float tm_lds[2 * kChunk];
    if ((int)blockIdx.x < a.tm_blocks) { tm_block<NT>(a, (int)blockIdx.x, tm_lds); return; }
    const int b = (int)blockIdx.x - a.tm_blocks;
    if (b == a.sum_block) { sum_partials(a, sm); return; }
    int chunk, e_lo, e_hi;
    if (b < a.n_small) {
        const int q = b / a.draws;
        chunk = a.small_chunk0 + q;
        e_lo = b - q * a.draws;
        e_hi = e_lo + 1;
    } else {
        chunk = b - a.n_small;
        e_lo = 0;
        e_hi = a.draws;
    }
    const int s = find_segment(a, chunk);
    const bbb_segment_t sg = a.seg[s];
    const int64_t base = (int64_t)(chunk - a.chunk_begin[s]) * (kChunk * GPT);
    const float mu0 = a.prior_mu, sig0 = a.prior_sigma;
    const bool textbook = (a.flags & BBB_KL_TEXTBOOK) != 0;
    const bool sq = (a.flags & BBB_SIGMA_SQUARED) != 0;
    const bool first = e_lo == 0;                               // this block owns the chunk's KL partial and sigma output
    // 16-byte accesses need aligned bases (and draw strides that keep them aligned); a tensor that does not qualify (a
    // 10-element bias: draw stride 10) moves element by element -- decided per segment, uniform over the block
    const bool ld_vec = ((((uintptr_t)sg.mu | (uintptr_t)sg.rho) & 15u) == 0);
    const bool st_vec = (((uintptr_t)sg.w & 15u) == 0) && ((sg.draw_stride & 3) == 0);
    const bool sg_vec = (((uintptr_t)sg.sigma & 15u) == 0);
    const bool want_kl = a.partials != nullptr && first;
    const uint32_t call0 = a.call0 + (a.call_dev ? *a.call_dev : 0u);
    const float l2s0 = __builtin_amdgcn_logf(sig0), is0 = 1.0f / sig0;

    f32x4 m4[GPT], sig[GPT], r4[GPT];
    int cnt[GPT];
#pragma unroll
    for (int it = 0; it < GPT; ++it) {                          // all loads first
        const int64_t i0 = base + ((int64_t)it * kThreads + threadIdx.x) * 4;
        const int64_t left = sg.n - i0;
        const int c = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
        cnt[it] = c;
        r4[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        m4[it] = r4[it];
        if (c > 0 && ld_vec) {
            m4[it] = *reinterpret_cast<const f32x4*>(sg.mu + i0);
            r4[it] = *reinterpret_cast<const f32x4*>(sg.rho + i0);
        } else if (c > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < c) { m4[it][j] = sg.mu[i0 + j]; r4[it][j] = sg.rho[i0 + j]; }
        }
    }
    // The first draw's noise does not depend on (mu, rho): generate it while the loads are in flight (the blocks of a launch
    // start together, so without this every wave of the chip sits in the same load-latency hole at the same time).
    float z0[4] = {0.f, 0.f, 0.f, 0.f};
    if (GPT == 1 && sg.w != nullptr && cnt[0] > 0) {
        const uint64_t g = (uint64_t)(base + (int64_t)threadIdx.x * 4) >> 2;
        bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e_lo, a.k0, a.k1, z0);
        __builtin_amdgcn_sched_barrier(0);                      // keep it ahead of the first use of the loaded vectors
    }
    double kl_acc = 0.0;
#pragma unroll
    for (int it = 0; it < GPT; ++it) {
        const int64_t i0 = base + ((int64_t)it * kThreads + threadIdx.x) * 4;
        const int c = cnt[it];
        float klf = 0.0f;
        sig[it] = bbb::softplus_ref4(r4[it]);
        if (want_kl) {
            const f32x4 kt = kl_term4(m4[it], sig[it], mu0, sig0, l2s0, is0, textbook);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < c) klf += kt[j];
        }
        kl_acc += (double)klf;
        if (first && c > 0 && sg.sigma != nullptr) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = sq ? sig[it][j] * sig[it][j] : sig[it][j];
            if (c == 4 && sg_vec) *reinterpret_cast<f32x4*>(sg.sigma + i0) = o;
            else for (int j = 0; j < c; ++j) sg.sigma[i0 + j] = o[j];
        }
    }
    if (a.partials != nullptr && first) publish_partial(a, chunk, kl_acc, sm);     // block-uniform condition
    if (sg.w == nullptr || sg.w_tm_cin != 0) return;                               // (tap-major outputs: the tm blocks draw them)
#pragma unroll
    for (int it = 0; it < GPT; ++it) {
        const int c = cnt[it];
        if (c == 0) continue;
        const int64_t i0 = base + ((int64_t)it * kThreads + threadIdx.x) * 4;
        const uint64_t g = (uint64_t)i0 >> 2;
        float* wp = sg.w + i0 + (int64_t)e_lo * sg.draw_stride;
        float z[4];
        if (GPT == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = z0[j];
        } else {
            bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e_lo, a.k0, a.k1, z);
        }
        // (the store flavour is loop-invariant: two copies of the loop instead of a branch, exec-mask juggling and selects in
        // every iteration -- hipcc does not unswitch it; measured 20.3-20.6 -> 19.2-19.4 us per 10-draw launch, A/B on one box)
        if (c == 4 && st_vec) {                                  // (unrolling by two removes six register moves and measures 6 % SLOWER)
            for (int e = e_lo;;) {                               // noise for draw e+1 is generated at the end of iteration e
                f32x4 w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = sample_w(m4[it][j], z[j], sig[it][j]);
                if (NT) __builtin_nontemporal_store(w4, reinterpret_cast<f32x4*>(wp));
                else    *reinterpret_cast<f32x4*>(wp) = w4;
                if (++e >= e_hi) break;
                wp += sg.draw_stride;
                bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e, a.k0, a.k1, z);
            }
        } else {
            for (int e = e_lo;;) {
                for (int j = 0; j < c; ++j) wp[j] = sample_w(m4[it][j], z[j], sig[it][j]);
                if (++e >= e_hi) break;
                wp += sg.draw_stride;
                bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e, a.k0, a.k1, z);
            }
        }
    }
}

// GPT = groups of 4 elements per thread.  1 for model-sized launches (enough blocks as it is); 4 for very large
// tensors, where 4x fewer, longer blocks with all their loads issued up front stream HBM better.
template <int GPT>
__global__ __launch_bounds__(kThreads) void reparam_kl_fwd_kernel(const ReparamArgs a) {
    constexpr int kGroupsPerThread = GPT;
    __shared__ double sm[kThreads];
    if ((int)blockIdx.x == a.sum_block) { sum_partials(a, sm); return; }
    const int chunk = blockIdx.x;
    const int s = find_segment(a, chunk);
    const bbb_segment_t sg = a.seg[s];
    const int64_t base = (int64_t)(chunk - a.chunk_begin[s]) * (kChunk * GPT);
    const float mu0 = a.prior_mu, sig0 = a.prior_sigma;
    const bool textbook = (a.flags & BBB_KL_TEXTBOOK) != 0;
    const bool sq = (a.flags & BBB_SIGMA_SQUARED) != 0;
    const bool want_kl = a.partials != nullptr;
    const uint32_t call0 = a.call0 + (a.call_dev ? *a.call_dev : 0u);
    const float l2s0 = __builtin_amdgcn_logf(sig0), is0 = 1.0f / sig0;
    // vector path needs 16-byte aligned rows for every draw
    const bool aligned = ((((uintptr_t)sg.mu | (uintptr_t)sg.rho | (uintptr_t)sg.w | (uintptr_t)sg.sigma |
                            (uintptr_t)sg.eps) & 15u) == 0) && ((sg.draw_stride & 3) == 0);

    double kl_acc = 0.0;
#pragma unroll
    for (int it = 0; it < kGroupsPerThread; ++it) {
        const int64_t i0 = base + ((int64_t)it * kThreads + threadIdx.x) * 4;
        if (i0 >= sg.n) break;
        const uint64_t g = (uint64_t)i0 >> 2;
        const int cnt = (sg.n - i0) >= 4 ? 4 : (int)(sg.n - i0);
        float mu[4], rho[4], sigma[4];
        if (aligned && cnt == 4) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(sg.mu + i0);
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(sg.rho + i0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = m4[j]; rho[j] = r4[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mu[j] = j < cnt ? sg.mu[i0 + j] : 0.0f;
                rho[j] = j < cnt ? sg.rho[i0 + j] : 0.0f;
            }
        }
        float klf = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sigma[j] = bbb::softplus_ref(rho[j]);
            if (want_kl && j < cnt) klf += kl_term(mu[j], sigma[j], mu0, sig0, l2s0, is0, textbook);
        }
        kl_acc += (double)klf;
        if (sg.sigma != nullptr) {
            if (aligned && cnt == 4) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = sq ? sigma[j] * sigma[j] : sigma[j];
                *reinterpret_cast<f32x4*>(sg.sigma + i0) = o;
            } else {
                for (int j = 0; j < cnt; ++j) sg.sigma[i0 + j] = sq ? sigma[j] * sigma[j] : sigma[j];
            }
        }
        // bf16 weight rows (w_row_len != 0): destination of each of the 4 elements inside one draw's matrix -- rows of rl
        // elements at a pitch rounded up to 8, optionally transposed to tap-major order; the element that ends a row also
        // zeroes the row's pad.  Independent of the draw, so computed once.
        int64_t bf_dst[4] = {0, 0, 0, 0}, bf_rl = 0, bf_pitch = 0, bf_pad_row[4] = {-1, -1, -1, -1};
        bool bf_vec = false;
        if (sg.w != nullptr && sg.w_row_len != 0) {
            bf_rl = sg.w_row_len;
            bf_pitch = (bf_rl + 7) & ~(int64_t)7;
            const uint32_t taps = sg.w_taps > 1 ? sg.w_taps : 1u, cin = (uint32_t)bf_rl / taps;
            const int64_t row0 = i0 / bf_rl;
            uint32_t col = (uint32_t)(i0 - row0 * bf_rl);
            int64_t row = row0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t ci = col / taps, tp = col - ci * taps;            // taps == 1: ci = col, tp = 0
                bf_dst[j] = row * bf_pitch + (int64_t)(tp * cin + ci);
                if (j < cnt && col == (uint32_t)bf_rl - 1 && bf_pitch != bf_rl) bf_pad_row[j] = row;   // every row that ends here
                if (++col == (uint32_t)bf_rl) { col = 0; ++row; }
            }
            bf_vec = taps == 1 && cnt == 4 && bf_dst[3] == bf_dst[0] + 3 && ((bf_rl & 3) == 0) &&
                     (((uintptr_t)sg.w & 7u) == 0) && ((sg.draw_stride & 3) == 0);
        }
        if (sg.w != nullptr) {
            for (int e = 0; e < a.draws; ++e) {
                float z[4];
                const int64_t o = (int64_t)e * sg.draw_stride + i0;
                if (sg.eps != nullptr) {
                    const int64_t oe = sg.w_row_len != 0 ? (int64_t)e * sg.n + i0 : o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = j < cnt ? sg.eps[oe + j] : 0.0f;
                } else {
                    bbb::normal4(g, sg.stream_id, call0 + (uint32_t)e, a.k0, a.k1, z);
                }
                if (sg.w_row_len != 0) {
                    // bf16 weights for the bf16 GEMM (destinations computed once, above the draw loop)
                    uint16_t* wb = reinterpret_cast<uint16_t*>(sg.w) + (int64_t)e * sg.draw_stride;
                    if (bf_vec) {
                        uint32_t lo = (uint32_t)bf16_bits(sample_w(mu[0], z[0], sigma[0])) | ((uint32_t)bf16_bits(sample_w(mu[1], z[1], sigma[1])) << 16);
                        uint32_t hi = (uint32_t)bf16_bits(sample_w(mu[2], z[2], sigma[2])) | ((uint32_t)bf16_bits(sample_w(mu[3], z[3], sigma[3])) << 16);
                        *reinterpret_cast<uint2*>(wb + bf_dst[0]) = make_uint2(lo, hi);
                    } else {
                        for (int j = 0; j < cnt; ++j) wb[bf_dst[j]] = bf16_bits(sample_w(mu[j], z[j], sigma[j]));
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (bf_pad_row[j] >= 0)
                            for (int64_t c = bf_rl; c < bf_pitch; ++c) wb[bf_pad_row[j] * bf_pitch + c] = 0;
                } else if (aligned && cnt == 4) {
                    f32x4 w4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = sample_w(mu[j], z[j], sigma[j]);
                    *reinterpret_cast<f32x4*>(sg.w + o) = w4;   // plain store: the GEMM re-reads w from L2 / Infinity Cache
                } else {
                    for (int j = 0; j < cnt; ++j) sg.w[o + j] = sample_w(mu[j], z[j], sigma[j]);
                }
            }
        }
    }

    if (want_kl) publish_partial(a, chunk, kl_acc, sm);
}

__global__ __launch_bounds__(kThreads) void reparam_kl_bwd_kernel(const ReparamArgs a) {
    constexpr int kGroupsPerThread = 1;
    const int chunk = blockIdx.x;
    const int s = find_segment(a, chunk);
    const bbb_segment_t sg = a.seg[s];
    float* gmu_out = a.grad_mu[s];
    float* grho_out = a.grad_rho[s];
    const int64_t base = (int64_t)(chunk - a.chunk_begin[s]) * kChunk;
    const float mu0 = a.prior_mu, sig0 = a.prior_sigma;
    const bool textbook = (a.flags & BBB_KL_TEXTBOOK) != 0;
    const float gkl = a.gkl ? *a.gkl : 0.0f;
    const bool sq = (a.flags & BBB_SIGMA_SQUARED) != 0;
    const bool mean_only = (a.flags & BBB_GW_MEAN_ONLY) != 0;   // gw is a gradient w.r.t. mu itself: nothing of it reaches rho
    const float* __restrict__ gsig = sg.sigma;                  // backward: gradient w.r.t. the sigma (or sigma^2) output, or NULL
    const bool aligned = ((((uintptr_t)sg.mu | (uintptr_t)sg.rho | (uintptr_t)sg.w | (uintptr_t)sg.eps |
                            (uintptr_t)gmu_out | (uintptr_t)grho_out) & 15u) == 0) && ((sg.draw_stride & 3) == 0);
#pragma unroll
    for (int it = 0; it < kGroupsPerThread; ++it) {
        const int64_t i0 = base + ((int64_t)it * kThreads + threadIdx.x) * 4;
        if (i0 >= sg.n) break;
        const uint64_t g = (uint64_t)i0 >> 2;
        const int cnt = (sg.n - i0) >= 4 ? 4 : (int)(sg.n - i0);
        float mu[4], rho[4];
        if (aligned && cnt == 4) {
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(sg.mu + i0);
            const f32x4 r4 = *reinterpret_cast<const f32x4*>(sg.rho + i0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { mu[j] = m4[j]; rho[j] = r4[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mu[j] = j < cnt ? sg.mu[i0 + j] : 0.0f;
                rho[j] = j < cnt ? sg.rho[i0 + j] : 0.0f;
            }
        }
        float acc_mu[4] = {0, 0, 0, 0}, acc_sig[4] = {0, 0, 0, 0};
        if (sg.w != nullptr) {
            for (int e = 0; e < a.draws; ++e) {
                float z[4], gw[4];
                const int64_t o = (int64_t)e * sg.draw_stride + i0;
                if (mean_only) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = 0.0f;
                } else if (sg.eps != nullptr) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = j < cnt ? sg.eps[o + j] : 0.0f;
                } else {
                    bbb::normal4(g, sg.stream_id, a.call0 + (a.call_dev ? *a.call_dev : 0u) + (uint32_t)e, a.k0, a.k1, z);
                }
                if (aligned && cnt == 4) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(sg.w + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) gw[j] = g4[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) gw[j] = j < cnt ? sg.w[o + j] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc_mu[j] += gw[j]; acc_sig[j] += gw[j] * z[j]; }
            }
        }
        float gm[4], gr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sigma = bbb::softplus_ref(rho[j]);
            const float sgm = 1.0f / (1.0f + expf(-rho[j]));     // d sigma / d rho
            const float d = mu[j] - mu0;
            const float is = 1.0f / sigma;
            float kmu, ksig;
            if (!textbook) {
                kmu = d * is * is;
                ksig = is - (sig0 * sig0 + d * d) * is * is * is;
            } else {
                kmu = d / (sig0 * sig0);
                ksig = sigma / (sig0 * sig0) - is;
            }
            // the sigma output of the forward (the LRT layers' variance operand): d sigma / d rho = sigmoid(rho),
            // d sigma^2 / d rho = 2 sigma sigmoid(rho)
            const float gs = (gsig != nullptr && j < cnt) ? gsig[i0 + j] * (sq ? 2.0f * sigma : 1.0f) : 0.0f;
            gm[j] = acc_mu[j] + gkl * kmu;
            gr[j] = (acc_sig[j] + gkl * ksig + gs) * sgm;
        }
        if (aligned && cnt == 4) {
            f32x4 a4, b4;
#pragma unroll
            for (int j = 0; j < 4; ++j) { a4[j] = gm[j]; b4[j] = gr[j]; }
            *reinterpret_cast<f32x4*>(gmu_out + i0) = a4;
            *reinterpret_cast<f32x4*>(grho_out + i0) = b4;
        } else {
            for (int j = 0; j < cnt; ++j) { gmu_out[i0 + j] = gm[j]; grho_out[i0 + j] = gr[j]; }
        }
    }
}

__global__ __launch_bounds__(kThreads) void eps_dump_kernel(float* out, int64_t n, int64_t start, uint32_t k0, uint32_t k1,
                                                            uint32_t call, uint32_t stream_id) {
    const int64_t g0 = start >> 2;
    const int64_t g = g0 + (int64_t)blockIdx.x * kThreads + threadIdx.x;
    float z[4];
    bbb::normal4((uint64_t)g, stream_id, call, k0, k1, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = g * 4 + j - start;
        if (i >= 0 && i < n) out[i] = z[j];
    }
}

int fill_args(ReparamArgs& a, const bbb_segment_t* segs, int nseg, int draws, bool bwd, int gpt = 1) {
    if (segs == nullptr || nseg <= 0 || nseg > BBB_MAX_SEGMENTS || draws <= 0) return BBB_EINVAL;
    int chunks = 0;
    for (int s = 0; s < nseg; ++s) {
        const bbb_segment_t& g = segs[s];
        if (g.mu == nullptr || g.rho == nullptr || g.n <= 0) return BBB_EINVAL;
        if (g.draw_stride < g.n && draws > 1 && (g.w || g.eps)) return BBB_EINVAL;
        if ((((uintptr_t)g.mu | (uintptr_t)g.rho | (g.w_row_len ? 0 : (uintptr_t)g.w) | (uintptr_t)g.sigma | (uintptr_t)g.eps) & 3u) != 0)
            return BBB_EALIGN;
        if (g.w_row_len != 0 && (bwd || g.w == nullptr || g.n % g.w_row_len != 0 || ((uintptr_t)g.w & 1u))) return BBB_EINVAL;
        if (g.w_tm_cin != 0) {      // fp32 tap-major output: dense Philox-sampled fp32 segment, [rows][C][T] with C % 8 == 0, 16-byte aligned
            if (bwd || g.w_row_len != 0 || g.w_taps < 2 || g.w_taps > 128 || g.w_tm_cin % 8 != 0 || g.w == nullptr || g.eps != nullptr ||
                g.n % ((int64_t)g.w_tm_cin * g.w_taps) != 0)
                return BBB_EINVAL;
            if ((((uintptr_t)g.mu | (uintptr_t)g.rho | (uintptr_t)g.w) & 15u) != 0 || (g.draw_stride & 3) != 0) return BBB_EALIGN;
        } else if (g.w_taps > 1 && (g.w_row_len == 0 || g.w_row_len % g.w_taps != 0)) return BBB_EINVAL;
        a.seg[s] = g;
        a.chunk_begin[s] = chunks;
        chunks += (int)((g.n + (int64_t)kChunk * gpt - 1) / ((int64_t)kChunk * gpt));
        (void)bwd;
    }
    for (int s = nseg; s <= BBB_MAX_SEGMENTS; ++s) a.chunk_begin[s] = chunks;
    a.nseg = nseg;
    a.draws = draws;
    return chunks;
}

}  // namespace

extern "C" int64_t bbb_reparam_partials(const bbb_segment_t* segs, int nseg) {
    if (segs == nullptr || nseg <= 0 || nseg > BBB_MAX_SEGMENTS) return BBB_EINVAL;
    int64_t chunks = 0;
    for (int s = 0; s < nseg; ++s) chunks += (segs[s].n + kChunk - 1) / kChunk;
    return chunks;
}

// The fast kernel applies when every segment is a dense fp32, Philox-sampled tensor (alignment is handled per segment).
static bool fast_path_ok(const bbb_segment_t* segs, int nseg) {
    for (int s = 0; s < nseg; ++s)
        if (segs[s].eps != nullptr || segs[s].w_row_len != 0) return false;
    return true;
}

// 256-thread blocks the current device keeps resident at once: 8 per CU (<= 64 VGPRs, 2 KB LDS; 32 waves per CU).
static int resident_blocks() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 2048;
    if (cached[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev] = cus * 8;
    }
    return cached[dev];
}

extern "C" int bbb_reparam_kl_fwd(const bbb_segment_t* segs, int nseg, int draws, float prior_mu, float prior_sigma,
                                  uint64_t seed, uint32_t call0, uint32_t flags, double* kl_partials, float* kl_out,
                                  double* kl_out64, const uint32_t* call_dev, void* stream) {
    ReparamArgs a = {};
    int64_t total = 0;
    if (segs != nullptr && nseg > 0 && nseg <= BBB_MAX_SEGMENTS)
        for (int s = 0; s < nseg; ++s) total += segs[s].n;
    const bool big = total > (int64_t)kChunk * 16384;               // > 16M elements: longer blocks, streaming stores
    const int gpt = big ? 4 : 1;
    const int chunks = fill_args(a, segs, nseg, draws, false, gpt);
    if (chunks < 0) return chunks;
    const bool want_kl = (kl_out != nullptr) || (kl_out64 != nullptr);
    if (want_kl && (kl_partials == nullptr || ((uintptr_t)kl_partials & 7u) != 0)) return BBB_EINVAL;
    if (!(prior_sigma > 0.0f)) return BBB_EINVAL;
    a.prior_mu = prior_mu;
    a.prior_sigma = prior_sigma;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.call0 = call0;
    a.flags = flags;
    a.partials = want_kl ? reinterpret_cast<unsigned long long*>(kl_partials) : nullptr;
    a.out32 = kl_out;
    a.out64 = kl_out64;
    a.call_dev = call_dev;
    a.n_chunks = chunks;
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(kThreads);
#ifdef BBB_FORCE_GENERIC_REPARAM     // timing experiments: the general kernel on inputs the fast one would take
    const bool fast = false;
#else
    const bool fast = fast_path_ok(segs, nseg);
#endif
    if (fast) {
        // tap-major segments: their draws come from extra blocks at the front of the grid (tm_block)
        int tmb = 0;
        for (int s = 0; s < nseg; ++s) {
            a.tm_begin[s] = tmb;
            a.tm_cg[s] = 0;
            if (segs[s].w_tm_cin != 0) {
                const int64_t rc_total = segs[s].n / segs[s].w_taps;
                int64_t cg = (kChunk / (int)segs[s].w_taps) & ~7;
                if (cg > rc_total) cg = rc_total;
                a.tm_cg[s] = (int)cg;
                const int64_t nb = (rc_total + cg - 1) / cg;
                if (tmb + nb > 0x3fffffffLL) return BBB_ESHAPE;
                tmb += (int)nb;
            }
        }
        for (int s = nseg; s <= BBB_MAX_SEGMENTS; ++s) a.tm_begin[s] = tmb;
        a.tm_blocks = tmb;
        // A launch a little larger than one round of resident blocks (the model-sized case: 2137 chunks on 2048 slots) would
        // run its excess blocks alone at the end, one wave per SIMD, for a whole 10-draw block time.  Instead the LAST
        // `excess` chunks are cut into one block per draw and put FIRST in the grid: they finish early, the whole-chunk
        // blocks behind them fill the freed slots, and the launch ends with every SIMD still sharing work.  (The block of
        // draw 0 owns the chunk's KL partial and sigma output; partial indices = chunk indices, so the KL sum is unchanged.)
        const int slots = resident_blocks();
        int excess = 0;
        if (gpt == 1 && draws > 1 && chunks > slots && chunks <= 3 * slots && tmb == 0) excess = chunks % slots;
        a.n_small = excess * draws;
        a.small_chunk0 = chunks - excess;
        const int blocks = a.n_small + (chunks - excess);
        a.sum_block = want_kl ? blocks : -1;
        const dim3 grid(tmb + blocks + (want_kl ? 1 : 0));
        // Store flavour of w (measured, AlexNet's 12 tensors, us per launch, plain / non-temporal): E=4 14.6 / 12.2, E=10
        // 21.7 / 19.2, E=25 40.0 / 42.5 -- with plain stores the launch ends with up to 32 MB of dirty L2 lines to write back
        // at the kernel boundary; streaming them out as they are produced wins until the launch is long enough to hide that.
        const bool nt = big || draws <= 16;
        if (big)     hipLaunchKernelGGL((reparam_kl_fast_kernel<4, true>), grid, block, 0, st, a);
        else if (nt) hipLaunchKernelGGL((reparam_kl_fast_kernel<1, true>), grid, block, 0, st, a);
        else         hipLaunchKernelGGL((reparam_kl_fast_kernel<1, false>), grid, block, 0, st, a);
    } else {
        for (int s = 0; s < nseg; ++s)
            if (segs[s].w_tm_cin != 0) return BBB_EINVAL;                 // tap-major outputs: Philox-sampled dense launches only
        a.sum_block = want_kl ? chunks : -1;
        const dim3 grid(chunks + (want_kl ? 1 : 0));
        if (big) hipLaunchKernelGGL(reparam_kl_fwd_kernel<4>, grid, block, 0, st, a);
        else     hipLaunchKernelGGL(reparam_kl_fwd_kernel<1>, grid, block, 0, st, a);
    }
    return (int)hipGetLastError();
}

extern "C" int bbb_reparam_kl_bwd(const bbb_segment_t* segs, int nseg, int draws, float prior_mu, float prior_sigma,
                                  uint64_t seed, uint32_t call0, uint32_t flags, const float* gkl,
                                  float* const* grad_mu, float* const* grad_rho, const uint32_t* call_dev, void* stream) {
    ReparamArgs a = {};
    const int chunks = fill_args(a, segs, nseg, draws, true);
    if (chunks < 0) return chunks;
    if (grad_mu == nullptr || grad_rho == nullptr || !(prior_sigma > 0.0f)) return BBB_EINVAL;
    for (int s = 0; s < nseg; ++s) {
        if (grad_mu[s] == nullptr || grad_rho[s] == nullptr) return BBB_EINVAL;
        if ((((uintptr_t)grad_mu[s] | (uintptr_t)grad_rho[s]) & 3u) != 0) return BBB_EALIGN;
        a.grad_mu[s] = grad_mu[s];
        a.grad_rho[s] = grad_rho[s];
    }
    a.prior_mu = prior_mu;
    a.prior_sigma = prior_sigma;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.call0 = call0;
    a.flags = flags;
    a.gkl = gkl;
    a.call_dev = call_dev;
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(chunks), dim3(kThreads), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int bbb_eps_dump(float* out, int64_t n, int64_t start, uint64_t seed, uint32_t call, uint32_t stream_id, void* stream) {
    if (out == nullptr || n <= 0 || start < 0) return BBB_EINVAL;
    const int64_t groups = ((start + n + 3) >> 2) - (start >> 2);
    const int blocks = (int)((groups + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(eps_dump_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, out, n, start,
                       (uint32_t)seed, (uint32_t)(seed >> 32), call, stream_id);
    return (int)hipGetLastError();
}
