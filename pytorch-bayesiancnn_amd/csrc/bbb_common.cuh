// Device-side helpers shared by the gfx950 kernels: Philox4x32-7, Box-Muller on the hardware
// transcendental units, wave64 reductions.  CDNA4 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbb {

constexpr int kWave = 64;

// Philox4x32-7 (Random123; 7 rounds is the published minimum that passes BigCrush, 10 is its default safety margin).
// One call yields four 32-bit words = noise for four consecutive elements.  Measured on gfx950 (scratch/r2/valu_rate.hip,
// 8 waves per SIMD): v_mad_u64_u32 4.95, v_xor_b32 4.4, transcendentals 8.2 cycles per wave-instruction -- a round is
// 2 wide multiplies + 4 xors = 27.5 cycles per wave, so every round dropped is 6 % of the whole parameter pass.
#ifndef BBB_PHILOX_ROUNDS
#define BBB_PHILOX_ROUNDS 7      // the noise contract (include/bbb_hip.h); other values exist for timing experiments only
#endif
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < BBB_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Two uniform words -> two N(0,1) (Box-Muller).  Each word's top 23 bits become the mantissa of a float in [1, 2) with ONE
// v_alignbit_b32 ({0x7F, x} >> 9 = 0x3F800000 | x >> 9):
//   a = 1 + (xa >> 9) * 2^-23,  u1 = 2 - a  in (0, 1]   (exact; u1 = 1 gives z = 0, the smallest u1 = 2^-23 gives 5.65 sigma)
//   t = 1 + (xb >> 9) * 2^-23,  the angle in revolutions -- v_sin_f32 / v_cos_f32 take revolutions and are 1-periodic, so the
//                               integer part costs nothing and 2*pi*u2 is never formed.
// 5 plain + 4 transcendental instructions per pair (the round-1 form -- shift, add, convert, scale -- was 9 + 4).
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& z0, float& z1) {
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7Fu, xa, 9u));
    const float t = __builtin_bit_cast(float, __builtin_amdgcn_alignbit(0x7Fu, xb, 9u));
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(2.0f - a));  // sqrt(-2 ln2 log2(u1))
    z0 = r * __builtin_amdgcn_cosf(t);
    z1 = r * __builtin_amdgcn_sinf(t);
}

// eps for the four elements of group g of stream (seed, call, stream_id).
__device__ __forceinline__ void normal4(uint64_t g, uint32_t stream_id, uint32_t call,
                                        uint32_t k0, uint32_t k1, float (&z)[4]) {
    uint32_t x[4];
    philox4x32((uint32_t)g, (uint32_t)(g >> 32), stream_id, call, k0, k1, x);
    box_muller(x[0], x[1], z[0], z[1]);
    box_muller(x[2], x[3], z[2], z[3]);
}

// softplus as the reference writes it, log1p(exp(rho)) (layers/BBB/BBBConv.py:64); rho > 20 returns rho,
// which is the same fp32 value and does not overflow where the reference's exp does (rho > ~88).
// This is the per-element cost of the parameter pass, so it is built from the hardware exp2 / rcp units with the
// rounding errors folded back in (<= ~2 ulp end to end) instead of the ~100-instruction libm pair:
//   exp(rho)   = 2^hi * (1 + lo*ln2),  hi = rho*log2e rounded, lo = the product's rounding error (+ log2e's own)
//   log1p(t)   = 2*atanh(z), z = t/(2+t), odd series to z^9 (z <= 0.13 for t <= 0.3, truncation < 1e-9 relative)
// t > 0.3 (rho > -1.2, i.e. sigma > 0.26: far from any BBB posterior) takes the libm path.
__device__ __forceinline__ float exp_fast_accurate(float x) {
    const float hi = x * 1.4426950408889634f;
    float lo = fmaf(x, 1.4426950408889634f, -hi);
    lo = fmaf(x, 1.925963033500011e-8f, lo);
    const float r = __builtin_amdgcn_exp2f(hi);
    return fmaf(r, lo * 0.6931471805599453f, r);
}

__device__ __forceinline__ float rcp_newton(float v) {       // 1/v to < 1 ulp
    const float r = __builtin_amdgcn_rcpf(v);
    return fmaf(fmaf(-v, r, 1.0f), r, r);
}

__device__ __forceinline__ float softplus_ref(float rho) {
    if (rho > 20.0f) return rho;
    if (rho > -1.2f || rho < -80.0f) return log1pf(expf(rho));
    const float t = exp_fast_accurate(rho);
    const float d = 2.0f + t;
    const float rd = __builtin_amdgcn_rcpf(d);
    float z = t * rd;
    z = fmaf(fmaf(-z, d, t), rd, z);                           // correctly rounded t / (2 + t)
    const float z2 = z * z;
    const float p = fmaf(z2, fmaf(z2, fmaf(z2, fmaf(z2, 2.0f / 9.0f, 2.0f / 7.0f), 2.0f / 5.0f), 2.0f / 3.0f), 2.0f);
    return z * p;
}

// The same softplus on FOUR elements with packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32 handle two lanes' worth per
// instruction): identical operation sequence per element, hence identical bits, at ~half the VALU issue slots -- the
// once-per-element part of the parameter pass (sigma + KL) is ~15 % of its VALU time at 10 draws.  Elements outside the
// fast range of softplus_ref take the scalar function (rare: sigma > 0.26 or rho < -80).
typedef float bbb_f32x2 __attribute__((ext_vector_type(2)));
typedef float bbb_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bbb_f32x2 pk_fma(bbb_f32x2 a, bbb_f32x2 b, bbb_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ bbb_f32x2 pk_splat(float v) { return bbb_f32x2{v, v}; }

__device__ __forceinline__ bbb_f32x2 softplus_fast2(bbb_f32x2 x) {      // valid for -80 <= x <= -1.2
    const bbb_f32x2 hi = x * pk_splat(1.4426950408889634f);
    bbb_f32x2 lo = pk_fma(x, pk_splat(1.4426950408889634f), -hi);
    lo = pk_fma(x, pk_splat(1.925963033500011e-8f), lo);
    const bbb_f32x2 r = bbb_f32x2{__builtin_amdgcn_exp2f(hi.x), __builtin_amdgcn_exp2f(hi.y)};
    const bbb_f32x2 t = pk_fma(r, lo * pk_splat(0.6931471805599453f), r);
    const bbb_f32x2 d = pk_splat(2.0f) + t;
    const bbb_f32x2 rd = bbb_f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    bbb_f32x2 z = t * rd;
    z = pk_fma(pk_fma(-z, d, t), rd, z);
    const bbb_f32x2 z2 = z * z;
    const bbb_f32x2 p = pk_fma(z2, pk_fma(z2, pk_fma(z2, pk_fma(z2, pk_splat(2.0f / 9.0f), pk_splat(2.0f / 7.0f)), pk_splat(2.0f / 5.0f)),
                                   pk_splat(2.0f / 3.0f)), pk_splat(2.0f));
    return z * p;
}

__device__ __forceinline__ bbb_f32x4 softplus_ref4(bbb_f32x4 rho) {
    const bbb_f32x2 a = softplus_fast2(bbb_f32x2{rho.x, rho.y});
    const bbb_f32x2 b = softplus_fast2(bbb_f32x2{rho.z, rho.w});
    bbb_f32x4 s = bbb_f32x4{a.x, a.y, b.x, b.y};
    bool out = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) out |= !(rho[j] <= -1.2f && rho[j] >= -80.0f);
    if (__builtin_expect(out, 0)) {
        asm volatile("" ::: "memory");      // not speculatable: keeps this a real (almost never taken) branch instead of
#pragma unroll                              // four libm softplus evaluations executed under a select
        for (int j = 0; j < 4; ++j)
            if (!(rho[j] <= -1.2f && rho[j] >= -80.0f)) { asm volatile(""); s[j] = softplus_ref(rho[j]); }
    }
    return s;
}

// Epilogue activations.  Softplus(beta=1, threshold=20) on the hardware exp2 / log2 units:
// log(1 + e^v) = ln2 * log2(1 + 2^(v*log2 e)); absolute error <= ~1e-7 (v < -16.6 flushes to 0 instead of e^v < 6e-8).
// a * b rounded to fp32, as an instruction of its own: hipcc contracts a product into a following add / subtract (fma) by default,
// which is wrong wherever the ROUNDED product is the value (the squares of the LRT chain: their pieces must sum to the fp32 x * x the
// fp32 path computes).  The pragma keeps the `contract` flag off this multiply, so no later pass may fuse it.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) {
        const float t = __builtin_amdgcn_exp2f(fminf(v, 20.0f) * 1.4426950408889634f);
        const float sp = 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + t);
        return v > 20.0f ? v : sp;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
    return v;
}

__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

}  // namespace bbb
