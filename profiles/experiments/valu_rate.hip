// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction per SIMD, by opcode.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASM1)                                                                  \
__global__ __launch_bounds__(256) void k_##NAME(unsigned* out, int iters, unsigned s) {   \
    unsigned r0 = threadIdx.x + 1, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19; \
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;                                     \
    for (int i = 0; i < iters; ++i) {                                                     \
        ASM1 ASM1 ASM1 ASM1                                                               \
    }                                                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ (unsigned)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7); \
}

#define A1(op, r) asm volatile(op " %0, %0, %1" : "+v"(r) : "v"(s));
#define A1S(op, r) asm volatile(op " %0, %1, %0" : "+v"(r) : "s"(s));
#define U1(op, r) asm volatile(op " %0, %0" : "+v"(r));

#define X_XOR  A1S("v_xor_b32", r0) A1S("v_xor_b32", r1) A1S("v_xor_b32", r2) A1S("v_xor_b32", r3) A1S("v_xor_b32", r4) A1S("v_xor_b32", r5) A1S("v_xor_b32", r6) A1S("v_xor_b32", r7)
#define X_ADD  A1S("v_add_u32", r0) A1S("v_add_u32", r1) A1S("v_add_u32", r2) A1S("v_add_u32", r3) A1S("v_add_u32", r4) A1S("v_add_u32", r5) A1S("v_add_u32", r6) A1S("v_add_u32", r7)
#define X_MULLO A1S("v_mul_lo_u32", r0) A1S("v_mul_lo_u32", r1) A1S("v_mul_lo_u32", r2) A1S("v_mul_lo_u32", r3) A1S("v_mul_lo_u32", r4) A1S("v_mul_lo_u32", r5) A1S("v_mul_lo_u32", r6) A1S("v_mul_lo_u32", r7)
#define X_MULHI A1S("v_mul_hi_u32", r0) A1S("v_mul_hi_u32", r1) A1S("v_mul_hi_u32", r2) A1S("v_mul_hi_u32", r3) A1S("v_mul_hi_u32", r4) A1S("v_mul_hi_u32", r5) A1S("v_mul_hi_u32", r6) A1S("v_mul_hi_u32", r7)
#define X_MUL24 A1S("v_mul_u32_u24", r0) A1S("v_mul_u32_u24", r1) A1S("v_mul_u32_u24", r2) A1S("v_mul_u32_u24", r3) A1S("v_mul_u32_u24", r4) A1S("v_mul_u32_u24", r5) A1S("v_mul_u32_u24", r6) A1S("v_mul_u32_u24", r7)
#define X_FMUL A1S("v_mul_f32", r0) A1S("v_mul_f32", r1) A1S("v_mul_f32", r2) A1S("v_mul_f32", r3) A1S("v_mul_f32", r4) A1S("v_mul_f32", r5) A1S("v_mul_f32", r6) A1S("v_mul_f32", r7)
#define M64(q) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(q) : "v"((unsigned)q), "s"(s) : "vcc");
#define X_MAD64 M64(q0) M64(q1) M64(q2) M64(q3) M64(q4) M64(q5) M64(q6) M64(q7)
#define X_LOG U1("v_log_f32", r0) U1("v_log_f32", r1) U1("v_log_f32", r2) U1("v_log_f32", r3) U1("v_log_f32", r4) U1("v_log_f32", r5) U1("v_log_f32", r6) U1("v_log_f32", r7)
#define X_EXP U1("v_exp_f32", r0) U1("v_exp_f32", r1) U1("v_exp_f32", r2) U1("v_exp_f32", r3) U1("v_exp_f32", r4) U1("v_exp_f32", r5) U1("v_exp_f32", r6) U1("v_exp_f32", r7)
#define X_SIN U1("v_sin_f32", r0) U1("v_sin_f32", r1) U1("v_sin_f32", r2) U1("v_sin_f32", r3) U1("v_sin_f32", r4) U1("v_sin_f32", r5) U1("v_sin_f32", r6) U1("v_sin_f32", r7)
#define X_COS U1("v_cos_f32", r0) U1("v_cos_f32", r1) U1("v_cos_f32", r2) U1("v_cos_f32", r3) U1("v_cos_f32", r4) U1("v_cos_f32", r5) U1("v_cos_f32", r6) U1("v_cos_f32", r7)
#define X_SQRT U1("v_sqrt_f32", r0) U1("v_sqrt_f32", r1) U1("v_sqrt_f32", r2) U1("v_sqrt_f32", r3) U1("v_sqrt_f32", r4) U1("v_sqrt_f32", r5) U1("v_sqrt_f32", r6) U1("v_sqrt_f32", r7)
#define X_RCP U1("v_rcp_f32", r0) U1("v_rcp_f32", r1) U1("v_rcp_f32", r2) U1("v_rcp_f32", r3) U1("v_rcp_f32", r4) U1("v_rcp_f32", r5) U1("v_rcp_f32", r6) U1("v_rcp_f32", r7)
#define X_CVT U1("v_cvt_f32_u32", r0) U1("v_cvt_f32_u32", r1) U1("v_cvt_f32_u32", r2) U1("v_cvt_f32_u32", r3) U1("v_cvt_f32_u32", r4) U1("v_cvt_f32_u32", r5) U1("v_cvt_f32_u32", r6) U1("v_cvt_f32_u32", r7)
#define PK(q) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q));
#define X_PKMUL PK(q0) PK(q1) PK(q2) PK(q3) PK(q4) PK(q5) PK(q6) PK(q7)
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r) : "v"(s));
#define X_FMA FMA(r0) FMA(r1) FMA(r2) FMA(r3) FMA(r4) FMA(r5) FMA(r6) FMA(r7)
#define AB(r) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(r));
#define X_ALIGN AB(r0) AB(r1) AB(r2) AB(r3) AB(r4) AB(r5) AB(r6) AB(r7)

DEFK(xor, X_XOR) DEFK(add, X_ADD) DEFK(mullo, X_MULLO) DEFK(mulhi, X_MULHI) DEFK(mul24, X_MUL24) DEFK(fmul, X_FMUL)
DEFK(mad64, X_MAD64) DEFK(log, X_LOG) DEFK(exp, X_EXP) DEFK(sin, X_SIN) DEFK(cos, X_COS) DEFK(sqrt, X_SQRT) DEFK(rcp, X_RCP)
DEFK(cvt, X_CVT) DEFK(pkmul, X_PKMUL) DEFK(fma, X_FMA) DEFK(align, X_ALIGN)

typedef void (*kern_t)(unsigned*, int, unsigned);

int main() {
    struct { const char* name; kern_t k; } ks[] = {
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mul_u32_u24", k_mul24},
        {"v_mul_f32", k_fmul}, {"v_fma_f32", k_fma}, {"v_pk_mul_f32", k_pkmul}, {"v_mad_u64_u32", k_mad64}, {"v_alignbit_b32", k_align},
        {"v_cvt_f32_u32", k_cvt}, {"v_log_f32", k_log}, {"v_exp_f32", k_exp}, {"v_sin_f32", k_sin}, {"v_cos_f32", k_cos},
        {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;   // Hz (nominal)
    printf("device %s, %d CUs, nominal clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int blocks = cus * 8;                // 8 workgroups x 4 waves per CU = 8 waves per SIMD
    unsigned* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (auto& kk : ks) {
        for (int waves_per_simd : {1, 8}) {
            const int nb = cus * waves_per_simd;
            hipLaunchKernelGGL(kk.k, dim3(nb), dim3(256), 0, 0, out, 10, 12345u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(kk.k, dim3(nb), dim3(256), 0, 0, out, iters, 12345u);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double winst_per_simd = (double)iters * 32.0 * waves_per_simd;     // wave-instructions issued on each SIMD
            printf("%-16s waves/SIMD=%d  %8.3f ms  -> %.2f cycles per wave-instruction at nominal clock\n", kk.name, waves_per_simd, ms,
                   ms * 1e-3 * clk / winst_per_simd);
        }
    }
    return 0;
}
