// Kernel-argument block shared by the pixel-major GEMM variants (pconv_gemm.hip, pconv_bf16.hip).
#pragma once
#include <stdint.h>

struct PConvArgs {
    const float* x;
    const float* w;
    const float* w2;
    const float* bias;
    const float* bias2;
    float* y;
    float* y_mu;
    float* y_var;
    const float* eps_ext;
    int64_t x_ds, w_ds, b_ds, y_ds;
    int32_t B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
    int32_t K, Kp, khkw, act, sample;   // Kp: bf16 weight row pitch (pconv_bf16.hip)
    int32_t Mtiles, nbt, Ntiles, G, per_xcd;
    uint32_t k0, k1, call0, stream_id;
    int32_t wtap;        // bf16: weight rows are tap-major ((r, q, ci) order)
    uint32_t x_inv;      // byte offset that marks an invalid image row: x_inv + any column offset is out of range and does not wrap
    const uint32_t* call_dev;
    int32_t unit_div, unit_off, x_mod, b_off;   // work units (include/bbb_hip.h, bbb_conv_desc_t); unit_div <= 1: slab = draw
    int32_t ksplit;      // split contraction (pconv_body.cuh, SPLIT): k ranges per item, 0 / 1 = none
    int32_t* tickets;    // [items] arrival counters, zero between launches
    float* part;         // [items][ksplit][sets][64][BM] partial accumulator tiles
    int32_t px_run;      // pconv_bf16_smallk_kernel: consecutive output pixels per workgroup
    int64_t x_ps, y_ps;  // pconv_bf16x3 S3 operands: elements between the hi / mid / lo planes of an activation slab
    int32_t pool;        // bbb_conv_desc_t::pool: the launch also applies MaxPool2d(2, 2) to the activated output (pconv_body.cuh, POOL)
    int32_t x_div, x_off; // bbb_conv_desc_t::x_unit_div / x_unit_off: output slab e reads input slab (e + x_off) / x_div (x_div <= 1: slab e)
    int32_t y_f32;       // pconv_bf16_fewout_kernel: y is fp32 (the logits layer) instead of bf16
    int32_t vh0, vh1, vw0, vw1;   // pconv_c8x3: input rows [vh0, vh1) x columns [vw0, vw1) may hold non-zero data (the rest is declared zero)
    int32_t y_c8;        // pconv_bf16.hip: y is written channel-interleaved, [cout / 8][ho][wo][B][8] (BBB_BF16_OUT_C8)
};
