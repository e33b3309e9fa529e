/*
 * bbb_hip.h -- C ABI of libbbb_hip.so, the MI355X (gfx950) Bayes-by-Backprop hot path.
 *
 * The upstream project (kumar-shridhar/PyTorch-BayesianCNN) is pure Python and has no FFI; this
 * header is the native boundary its `layers` package would bind.  Each entry point names the
 * upstream lines it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data unless it says "host";
 *   - nothing is allocated, freed or retained; outputs are caller-owned buffers;
 *   - `stream` is a hipStream_t (0 = the null stream); every call only enqueues work on it;
 *   - return value: 0 on success, a negative BBB_E* code for argument errors, or the positive
 *     hipError_t of a failed launch.  No call aborts or throws.
 *   - tensors are contiguous: activations NCHW, conv weights [Cout, Cin, kh, kw], linear weights
 *     [out, in], all row-major (the reference's own layouts).
 *
 * Noise contract (replaces torch.empty(shape).normal_(0,1) on the CPU generator,
 * layers/BBB/BBBConv.py:63,68; layers/BBB_LRT/BBBConv.py:78): element i of noise stream
 * (seed, call, stream_id) is output (i & 3) of
 *     Philox4x32-7(counter = {lo32(i >> 2), i >> 34, stream_id, call}, key = {lo32(seed), hi32(seed)})
 * (Random123; 7 rounds = its published BigCrush-passing minimum) pushed through Box-Muller pairs
 * (x0,x1)->(z0,z1), (x2,x3)->(z2,z3) with 23-bit mantissas
 *     u1 = 1 - (xa >> 9) * 2^-23 in (0,1],  u2 = (xb >> 9) * 2^-23 in [0,1),  z = sqrt(-2 ln u1) * {cos,sin}(2 pi u2).
 * (ABI <= 3 used Philox4x32-10 and 24-bit uniforms; the stream definition is part of the ABI version.)
 * Monte-Carlo draw j of an ensemble uses call = call0 + j, so a batched E-draw launch and E
 * single-draw launches produce the same numbers, and any GPU can materialise any draw.
 */
#ifndef BBB_HIP_H
#define BBB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BBB_ABI_VERSION 13
#define BBB_MAX_SEGMENTS 16

#define BBB_EINVAL (-1)   /* bad argument (null pointer, non-positive size, too many segments) */
#define BBB_EALIGN (-2)   /* pointer not 4-byte aligned */
#define BBB_ESHAPE (-3)   /* inconsistent convolution geometry */

/* One parameter tensor pair (mu, rho) of a Bayesian layer: W or bias. */
typedef struct bbb_segment {
    const float* mu;      /* [n]  W_mu / bias_mu */
    const float* rho;     /* [n]  W_rho / bias_rho */
    float* w;             /* out [draws][n] at w + e*draw_stride: mu + softplus(rho)*eps; NULL = no sampling */
    float* sigma;         /* out [n]: softplus(rho), or its square when BBB_SIGMA_SQUARED; NULL = skip */
    const float* eps;     /* test entry: external noise [draws][n] (same stride as w); NULL = on-chip Philox */
    int64_t n;            /* elements */
    int64_t draw_stride;  /* elements between consecutive draws of w / eps (>= n) */
    uint32_t stream_id;   /* noise stream of this tensor */
    uint32_t w_row_len;   /* 0: w is fp32, dense.  > 0: w is BF16 (round-to-nearest-even), a matrix with rows of
                             w_row_len elements stored at a pitch of w_row_len rounded up to 8 elements -- the operand
                             layout of bbb_conv2d_chwn_bf16_fwd; draw_stride then counts bf16 elements, the pad columns
                             are written as zeros, and `eps` (fp32) keeps the dense [draws][n] layout with stride n */
    uint32_t w_taps;      /* bf16 rows only.  0 / 1: row elements keep the tensor's own order ([cin][kh][kw]).  T > 1: the
                             row is a [cin][T] matrix (T = kh*kw) and is written TRANSPOSED, [T][cin] ("tap-major":
                             element (ci, t) lands at column t*cin + ci), the BBB_BF16_W_TAP_MAJOR operand layout */
    uint32_t w_tm_cin;    /* fp32 w only (w_row_len == 0; ABI 11).  C > 0: the tensor is [rows][C][T] with T = w_taps (2..128), C % 8 == 0,
                             and w is written TAP-MAJOR, fp32 [draws][rows][T][C] -- the weight operand of bbb_conv2d_c8x3_fwd.  Noise
                             elements, sigma and KL are those of the dense launch (the noise of a weight is keyed by its index in
                             the tensor's own order); mu, rho and w 16-byte aligned, draw_stride % 4 == 0, on-chip noise only. */
} bbb_segment_t;

#define BBB_SIGMA_SQUARED 1u   /* flags: write sigma^2 (the LRT variance operand) instead of sigma */
#define BBB_KL_TEXTBOOK   2u   /* flags: KL(q||p) instead of the reference's swapped-argument form */
#define BBB_GW_MEAN_ONLY  4u   /* bbb_reparam_kl_bwd flags: the `w` gradients reach mu only (no eps term into rho): an LRT layer's
                                * d loss / d W_mu folded into the KL backward instead of a separate accumulation per tensor */

/*
 * Fused reparameterisation + KL over up to BBB_MAX_SEGMENTS tensors and `draws` Monte-Carlo draws in
 * ONE pass over (mu, rho).  Replaces, per layer and per draw: the CPU normal_ + H2D copy, log1p(exp(rho)),
 * mu + eps*sigma (layers/BBB/BBBConv.py:63-70, BBBLinear.py:56-63), the sigma / sigma**2 of the LRT
 * layers (layers/BBB_LRT/BBBConv.py:64-69, BBBLinear.py:58-63) and metrics.calculate_kl as called by
 * kl_loss() (metrics.py:27-29 via layers/BBB/BBBConv.py:79-83), plus the Python sum over layers
 * (layers/misc.py:20-23).
 *   segs        host array of nseg descriptors (copied into the kernel arguments)
 *   kl_partials device scratch, at least bbb_reparam_partials(segs, nseg) doubles, 8-byte aligned, FILLED WITH 0xFF
 *               BYTES before the first launch that uses it ("not published yet"); every launch leaves the slots it
 *               used in that state again.  One scratch buffer per stream (launches that may overlap must not share one).
 *   kl_out      device float: sum over all segments of the KL term (NULL = no KL)
 *   kl_out64    optional device double with the same sum (NULL = skip)
 *   call_dev    optional DEVICE uint32 added to call0 when the kernel runs (NULL = 0).  This is what lets a
 *               captured hipGraph draw fresh noise on every replay: the graph also contains the increment.
 * KL does not depend on eps; the sum is reduced in a fixed order in fp64 (bitwise reproducible): every 1024-element
 * chunk publishes an fp64 partial before its draws start and one extra block of the same launch adds them up in index
 * order -- one launch, no separate finish kernel, the reduction hidden behind the sampling work.
 */
int bbb_reparam_kl_fwd(const bbb_segment_t* segs, int nseg, int draws,
                       float prior_mu, float prior_sigma,
                       uint64_t seed, uint32_t call0, uint32_t flags,
                       double* kl_partials, float* kl_out, double* kl_out64,
                       const uint32_t* call_dev, void* stream);

/* Number of doubles of scratch bbb_reparam_kl_fwd needs for these segments, host-only helper. */
int64_t bbb_reparam_partials(const bbb_segment_t* segs, int nseg);

/*
 * Backward of bbb_reparam_kl_fwd (what autograd derives for the reference):
 *   grad_mu  = sum_e gw[e] + gkl * (mu - mu0) / sigma^2
 *   grad_rho = (sum_e gw[e]*eps[e] + gkl * (1/sigma - sigma0^2/sigma^3 - (mu-mu0)^2/sigma^3)) * sigmoid(rho)
 * eps is regenerated from (seed, call0 + e, stream_id), never stored.  In each segment `w` holds the
 * incoming gradient gw [draws][n] (may be NULL = 0), `eps` optional external noise, and `sigma` (may be NULL = 0) the
 * incoming gradient gs [n] w.r.t. the forward's sigma output -- sigma^2 with BBB_SIGMA_SQUARED in flags, the LRT layers'
 * variance operand (layers/BBB_LRT/BBBConv.py:64-69) -- which adds gs * sigmoid(rho), resp. gs * 2 sigma * sigmoid(rho), to
 * grad_rho.
 * grad_mu / grad_rho are arrays of nseg device pointers given on the host.  gkl: device float
 * (d loss / d kl), NULL = 0.  call_dev: as in bbb_reparam_kl_fwd (a captured training step regenerates the forward's noise).
 */
int bbb_reparam_kl_bwd(const bbb_segment_t* segs, int nseg, int draws,
                       float prior_mu, float prior_sigma,
                       uint64_t seed, uint32_t call0, uint32_t flags,
                       const float* gkl, float* const* grad_mu, float* const* grad_rho,
                       const uint32_t* call_dev, void* stream);

/* One parameter tensor of an Adam step: all four arrays hold n fp32 elements on the device. */
typedef struct bbb_adam_segment {
    float* param;        /* updated in place */
    const float* grad;
    float* exp_avg;      /* first moment, updated in place */
    float* exp_avg_sq;   /* second moment, updated in place */
    int64_t n;
} bbb_adam_segment_t;

/*
 * Multi-tensor Adam (training extension): the optimizer.step() of main_bayesian.py:58 for up to BBB_MAX_SEGMENTS
 * tensors in one launch, torch.optim.Adam semantics (amsgrad off, no weight decay), operation order as torch's:
 *   m += (g - m)(1 - beta1);  v = v*beta2 + (1 - beta2) g^2;  p -= lr/(1 - beta1^step) * m / (sqrt(v)/sqrt(1 - beta2^step) + eps)
 * `step` is the 1-based count of this update.  Hyper-parameters are doubles: derived scalars (1 - beta, the bias
 * corrections) are formed in double and rounded to fp32 once, as torch does.  step_dev (optional): DEVICE float holding the
 * step count (torch's capturable-Adam convention); when given it overrides `step` and the bias corrections are computed on
 * the device, so a captured hipGraph advances correctly on every replay.  lr_dev (optional, only with step_dev): DEVICE
 * float holding the learning rate, read at run time instead of `lr` -- a scheduler (the reference uses ReduceLROnPlateau,
 * main_bayesian.py:118) can then change the rate of an already captured step.
 */
int bbb_adam_step(const bbb_adam_segment_t* segs, int nseg, double lr, double beta1, double beta2, double eps,
                  int64_t step, const float* step_dev, const float* lr_dev, void* stream);

/* Test entry: materialise n elements of a noise stream starting at element `start`. */
int bbb_eps_dump(float* out, int64_t n, int64_t start, uint64_t seed, uint32_t call, uint32_t stream_id, void* stream);

/* Geometry of one conv2d / linear contraction, batched over Monte-Carlo draws. */
typedef struct bbb_conv_desc {
    int32_t batch;        /* B images per draw */
    int32_t cin, h, w;    /* input  [draws|1][B][cin][h][w]  (linear: h = w = 1) */
    int32_t cout, kh, kw; /* weight [draws|1][cout][cin][kh][kw] */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
    int32_t draws;        /* E: independent weight sets / output slabs */
    int64_t x_draw_stride; /* elements between draws of x (0 = every draw reads the same x) */
    int64_t w_draw_stride; /* elements between draws of w (0 = shared weights) */
    int64_t b_draw_stride; /* elements between draws of bias (0 = shared) */
    int32_t act;          /* fused epilogue: 0 none, 1 ReLU, 2 Softplus(beta=1, threshold=20) */
    /* Work units for ensemble sharding (batch-innermost entry points only; all zero = every slab is a whole draw).
     * With unit_div = S > 1 the `draws` slabs of a launch are UNITS u = unit_off + e of the draw-major grid
     * (Monte-Carlo draw, batch slice): draw = u / S, slice = u % S, `batch` = images per slice.  Slab e then reads
     * weight / bias set (u / S) - (unit_off / S) ... i.e. (unit_off % S + e) / S relative to the first set passed in,
     * writes output slab e, and reads input slab e -- or, with x_unit_mod = S (a layer whose input is the same for every
     * draw), input slab u % S of an [S][cin][h][w][batch] tensor.  unit_off is passed already reduced modulo S. */
    int32_t unit_div;
    int32_t unit_off;
    int32_t x_unit_mod;
    int32_t w_row_pitch;  /* fp32 batch-innermost entry only: elements between consecutive output-channel rows of w (0 = dense,
                             cin*kh*kw).  Lets a launch contract over a K-slice of a wider matrix in place (split-K over draws). */
    int32_t b_offset;     /* LRT noise only: global index of local image 0 (batch-parallel shards), added to the image index
                             that keys the activation noise; a unit's slice adds slice * batch on top */
    int32_t x_unit_div;   /* batch-innermost entry points (ABI 8).  D > 1: output slab e reads INPUT slab e / D (x holds draws / D
                             slabs at x_draw_stride) -- several Monte-Carlo steps in one launch: slab e = step e / D, draw e % D,
                             the first layer's input being the same for all D draws of a step.  Weight / bias set: e, as usual.
                             0 / 1: input slab e.  Not combined with work units (unit_div > 1). */
    int32_t x_unit_off;   /* with x_unit_div = D > 1: output slab e reads input slab (e + x_unit_off) / D, 0 <= x_unit_off < D -- a
                             rank's share of a GROUP of steps starts in the middle of a step (draws x_unit_off .. D-1 of its first
                             step); x then holds ceil((draws + x_unit_off) / D) slabs and draws need not be a multiple of D. */
    int32_t pool;         /* 1 (bbb_conv2d_chwn_fwd only): the layer is followed by [activation ->] MaxPool2d(kernel 2, stride 2)
                             and the launch writes the POOLED map y[draws][cout][ho/2][wo/2][batch] -- one workgroup walks the four
                             conv pixels of a pooling window and keeps the running maximum of act(conv + bias) in registers; bit
                             for bit bbb_maxpool_chwn(bbb_conv2d_chwn_fwd(...), 2, 2).  Needs even ho and wo and a layer without a
                             split contraction (bbb_conv2d_chwn_splitk_scratch reports k_split 1), else BBB_EINVAL; items are four
                             times fewer and four times longer, so it pays for launches that still spread evenly over the chip
                             (the callers decide: bbb_hip/ops.py pool_fusion_ok).  0: none.
                             bbb_conv2d_chwn_bf16_fwd (ABI 9): 1 = MaxPool2d(2, 2), (k << 8) | s = MaxPool2d(k, s); admitted: 2 / 2
                             and 3 / 2, for first layers with a row pitch <= 128 (weights in registers: 3Conv3FC conv1, LeNet
                             conv1) -> y [draws][cout][hp][wp][B]; the maximum is taken over the fp32 contraction results, then
                             bias, activation and the bf16 rounding once per pooled pixel (non-decreasing: the same values as
                             bbb_maxpool_chwn_bf16 of the unfused launch). */
    int32_t w_tap_major;  /* fp32 batch-innermost entries (bbb_conv2d_chwn_fwd, _splitk_fwd, the LRT forms; ABI 13): 1 = w rows are
                             [kh][kw][cin] instead of [cin][kh][kw] AND the contraction runs tap-major (in-bounds taps outermost, cin
                             innermost) -- the same products, summed in another order.  Training extension: the role-swapped weight
                             gradient reads the output gradient [cout][ho][wo][batch] in place as such a weight operand (no
                             transposed copy).  Elsewhere it must be 0. */
} bbb_conv_desc_t;

/*
 * y[e] = conv2d(x[e], w[e], bias[e]) on the fp32 matrix cores (implicit im2col GEMM, groups = 1).
 * Replaces F.conv2d / F.linear in layers/BBB/BBBConv.py:77 and layers/BBB/BBBLinear.py:70.
 * y: [draws][B][cout][ho][wo], contiguous.  bias may be NULL.
 */
int bbb_conv2d_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias,
                   float* y, void* stream);

/*
 * Local-reparameterisation layer in one launch (layers/BBB_LRT/BBBConv.py:71-81, BBBLinear.py:65-73):
 *   act_mu  = conv(x, w_mu, b_mu);  act_var = 1e-16 + conv(x*x, w_var, b_var)
 *   y       = act_mu + sqrt(act_var) * eps          (sample != 0)   |   act_mu   (sample == 0)
 * Both contractions share one staged x tile (squared in registers); eps is generated in the epilogue,
 * element index ((b*cout + c)*ho + oh)*wo + ow of stream (seed, call0 + e, stream_id).  Here d->draws
 * slabs of x are independent samples sharing w_mu / w_var (w_draw_stride must be 0).
 *   eps_ext   test entry: external noise with y's layout (NULL = Philox)
 *   act_mu_out / act_var_out: optional copies of the two moments (NULL = skip)
 */
int bbb_lrt_conv2d_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                       const float* b_mu, const float* b_var, float* y,
                       float* act_mu_out, float* act_var_out, const float* eps_ext,
                       uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                       const uint32_t* call_dev, void* stream);

/*
 * Batch-innermost ("CHWN") variants used by the batched Monte-Carlo ensemble path.
 *   x: [draws|1][cin][h][w][B]   y: [draws][cout][ho][wo][B]   (B = d->batch, B % 4 == 0, 16-byte aligned)
 * Same contraction and same results as bbb_conv2d_fwd / bbb_lrt_conv2d_fwd (weights keep the reference's
 * [cout][cin][kh][kw] layout); one workgroup per (output pixel, 64 channels, 64|128 images) and kernel taps
 * that fall into the zero padding are skipped instead of multiplied.  LRT eps is indexed by the canonical
 * NCHW element index, so both layouts consume the identical noise stream.
 */
int bbb_conv2d_chwn_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias,
                        float* y, void* stream);
/* The same launch with the contraction on the 16-bit matrix pipe at fp32 accuracy, RANGE-FREE ("split bf16", ABI 8): every fp32
 * operand element is cut into hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid) (a = hi + mid + lo exactly; bf16 has
 * fp32's exponent range: no operand scales, windows or saturation) and every product is
 * lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the dropped terms are
 * < 2^-23 |a b|.  Same descriptor as bbb_conv2d_chwn_fwd (work units, x_unit_div, w_row_pitch included); not bit-identical to it.
 * Opt-in (ops.gemm_mode = "bf16x3").  Operands must be finite.
 * flags = 0: x and y are the fp32 tensors of bbb_conv2d_chwn_fwd (operands are cut while their tiles are staged).
 * BBB_S3_IN / BBB_S3_OUT: x / y travel in the split ACTIVATION format "S3" -- three bf16 planes per slab,
 * [draws|1][3][C][H][W][B] (batch % 8 == 0, 16-byte aligned; d->x_draw_stride then counts bf16 elements, 3*C*H*W*B per slab or
 * 0 = shared) -- the exact fp32 values stored as their three pieces, written once by the producing launch and staged by the
 * consumer without arithmetic (the kernel is bound by the VALU work of the split otherwise).  bbb_maxpool_chwn_s3 pools S3
 * tensors; bbb_s3_convert converts fp32 <-> S3 (both exact). */
#define BBB_S3_IN  1u
#define BBB_S3_OUT 2u
int bbb_conv2d_chwn_bf16x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w, const float* bias, void* y, uint32_t flags,
                               void* stream);
/* nn.MaxPool2d(k, s) on an S3 tensor [slabs][3][channels][h][w][batch] -> [slabs][3][channels][ho][wo][batch]: the S3 form of what
 * bbb_maxpool_chwn computes on the fp32 values. */
int bbb_maxpool_chwn_s3(const void* x, void* y, int64_t slabs, int channels, int h, int w, int batch, int k, int s, void* stream);
/* fp32 [slabs][n] -> S3 [slabs][3][n] (to_s3 != 0) or back (to_s3 == 0); n % 8 == 0; exact in both directions. */
int bbb_s3_convert(const void* src, void* dst, int64_t slabs, int64_t n, int to_s3, void* stream);
/*
 * The split-bf16 contraction over MFMA-READY operands (ABI 11): same arithmetic as bbb_conv2d_chwn_bf16x3_fwd (three bf16 pieces per
 * fp32 element, six products per fp32 product, fp32 accumulation, in-bounds taps only), with both operands laid out the way
 * v_mfma_f32_32x32x16_bf16 wants them, so that the k loop holds no operand arithmetic, no transposing LDS reads and no LDS traffic
 * on the image side:
 *   x: "c8 S3" activations, bf16 [draws|1][3][cin / 8][h][w][B][8] -- the hi / mid / lo pieces (exact: hi + mid + lo is the fp32
 *      value) of the 8 channels 8g .. 8g + 7 of an image are 16 adjacent bytes of a plane = one lane's B operand.  cin % 16 == 0,
 *      B % 4 == 0, 16-byte aligned; d->x_draw_stride counts bf16 elements (3 * cin * h * w * B per slab, or 0 = shared).
 *   w: fp32, TAP-MAJOR rows [draws|1][cout][kh * kw][cin] -- what bbb_reparam_kl_fwd writes for a segment with w_tm_cin = cin,
 *      w_taps = kh * kw (bbb_w_tap_major converts a dense tensor); 16-byte aligned, draw strides multiples of 4 elements.
 *   y: c8 S3 [draws][3][cout / 8][ho][wo][B][8] (cout % 8 == 0), or with BBB_C8X3_OUT_F32 the fp32 batch-innermost tensor
 *      [draws][cout][ho][wo][B] of bbb_conv2d_chwn_fwd (the logits layer).
 * Work units / x_unit_div as in bbb_conv2d_chwn_fwd; d->pool and d->w_row_pitch must be 0.  bbb_maxpool_chwn_s3 pools c8 S3 tensors
 * (channels = cin / 8, batch = 8 * B: the window maximum is element-wise on 16-byte vectors); bbb_c8s3_convert converts
 * fp32 batch-innermost [slabs][channels][positions][batch] <-> c8 S3 (exact both ways).  Replaces F.conv2d / F.linear of
 * layers/BBB/BBBConv.py:77, layers/BBB/BBBLinear.py:70; every slab (three planes) must stay below 1 GiB.
 * BBB_C8X3_TILE128 / _TILE256 force the images per workgroup and the BBB_C8X3_NT bits the channels per workgroup (default: by layer
 * and launch size); the MFMA sequence per output element, hence every output bit, is the same for all of them.
 */
#define BBB_C8X3_OUT_F32 1u
#define BBB_C8X3_TILE128 2u
#define BBB_C8X3_TILE256 4u
#define BBB_C8X3_POOL 8u          /* the launch also applies MaxPool2d(2, 2) to the activated output ("parallel window": the four waves of a
                                     workgroup own the four pixels of a window).  pad == 0, even ho and wo, c8 S3 output
                                     [draws][3][cout / 8][ho / 2][wo / 2][B][8]; bit for bit the plain launch + bbb_maxpool_chwn_s3(2, 2).
                                     TILE128 / TILE256 then mean 32 / 64 images per workgroup. */
#define BBB_C8X3_NT_SHIFT 4       /* bits 4..6: force the channels per workgroup, 32 * NT with NT = 2, 3 or 4 (0: by layer and launch size) */
#define BBB_C8X3_NT_MASK 0x70u
/* bits 8..23: four nibbles = leading rows, leading columns, trailing rows, trailing columns of the INPUT map that are all zero (the
 * materialised padding of a bbb_s2d_c8s3 block image): their products are exact zeros and are not computed -- same result as
 * computing them, less work.  A promise of the caller: non-zero data there is silently ignored. */
#define BBB_C8X3_ZERO_LEAD_H(n) (((uint32_t)(n) & 15u) << 8)
#define BBB_C8X3_ZERO_LEAD_W(n) (((uint32_t)(n) & 15u) << 12)
#define BBB_C8X3_ZERO_TRAIL_H(n) (((uint32_t)(n) & 15u) << 16)
#define BBB_C8X3_ZERO_TRAIL_W(n) (((uint32_t)(n) & 15u) << 20)
#define BBB_C8X3_ZERO_MASK 0x00FFFF00u
int bbb_conv2d_c8x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w, const float* bias, void* y, uint32_t flags,
                        void* stream);
/* to_c8s3: 1 = fp32 -> c8 S3, 0 = back; 2 = fp32 -> six planes (values + squares), 3 = six planes -> fp32 (the LRT chain's slabs) */
int bbb_c8s3_convert(const void* src, void* dst, int64_t slabs, int channels, int64_t positions, int batch, int to_c8s3, void* stream);
/*
 * The LRT form of the same contraction (ABI 12; layers/BBB_LRT/BBBConv.py:62-87, BBBLinear.py:56-79 in the split-bf16 mode):
 *   act_mu = conv(x, w_mu) + b_mu,  act_var = 1e-16 + conv(x^2, w_var) + b_var,  y = act(act_mu + sqrt(act_var) * eps)
 * with both contractions on v_mfma_f32_32x32x16_bf16 (two weight tiles, two sets of image fragments and accumulators per
 * workgroup) and eps drawn in the epilogue exactly as bbb_lrt_conv2d_chwn_fwd draws it (element index of the output's canonical
 * [B][cout][ho][wo] slab, stream (seed, call0 + draw, stream_id); d->b_offset and work units key the GLOBAL image index).
 *   x, y: SIX-plane c8 S3 slabs [draws|1][6][c / 8][h][w][B][8] -- the three pieces of the values, then the three pieces of their
 *      squares (squared in fp32 by whoever writes the slab: this launch's epilogue, bbb_maxpool_chwn_s3sq, bbb_s2d_c8s3sq,
 *      bbb_c8s3_convert mode 2); d->x_draw_stride counts bf16 elements (6 * cin * h * w * B per slab, or 0 = shared).
 *      With BBB_C8X3_OUT_F32, y is the fp32 tensor [draws][cout][ho][wo][B] (the logits layer).
 *   w_mu, w_var: fp32 tap-major [cout][kh * kw][cin], shared by the slabs (d->w_draw_stride == d->b_draw_stride == 0).
 * flags: BBB_C8X3_OUT_F32, BBB_C8X3_TILE128 / _TILE256, BBB_C8X3_ZERO_*; no pooled form.  sample == 0: y = act(act_mu).
 * bbb_maxpool_chwn_s3sq pools six-plane slabs (maximum over the values; the squares are those of the pooled values);
 * bbb_s2d_c8s3sq is bbb_s2d_c8s3 with the squares' planes behind; bbb_c8s3_convert modes 2 / 3 convert fp32 <-> six planes.
 */
int bbb_lrt_conv2d_c8x3_fwd(const bbb_conv_desc_t* d, const void* x, const float* w_mu, const float* w_var, const float* b_mu,
                            const float* b_var, void* y, uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                            const uint32_t* call_dev, uint32_t flags, void* stream);
int bbb_maxpool_chwn_s3sq(const void* x, void* y, int64_t slabs, int channels, int h, int w, int batch, int k, int s, void* stream);
int bbb_s2d_c8s3sq(const float* x, void* y, int64_t blocks, int batch, int channels, int h, int w, int k, int stride, int pad, void* stream);
/*
 * Space-to-depth operands (ABI 12): a strided layer with few input channels (AlexNet conv1: 3 channels, 11 x 11, stride 4, padding 5 --
 * models/BayesianModels/BayesianAlexNet.py:33) as a layer bbb_conv2d_c8x3_fwd can take.  With m = ceil(k / stride) it is the m x m
 * convolution, stride 1, no padding, of the block image x'[(c s + dy) s + dx][bh][bw] = xpad[c][s bh + dy][s bw + dx] with weights
 * w'[(mr, mq)][(c s + dy) s + dx] = w[c][s mr + dy][s mq + dx] (zero outside the k x k taps); C' = channels * stride^2 rounded up to a
 * multiple of 16.  Same products as F.conv2d(x, w, stride, padding) of layers/BBB/BBBConv.py:77 plus exact zeros.
 * bbb_s2d_c8s3: the caller's NCHW fp32 batch [blocks * batch][channels][h][w] -> c8 S3 [blocks][3][C' / 8][ho + m - 1][wo + m - 1][batch][8]
 * (one block per batch slice / per step of a launch); bbb_w_s2d_tap_major: w [rows][channels][k][k] -> [rows][m * m][C'].
 */
int bbb_s2d_c8s3(const float* x, void* y, int64_t blocks, int batch, int channels, int h, int w, int k, int stride, int pad, void* stream);
int bbb_w_s2d_tap_major(const float* w, float* out, int64_t rows, int channels, int k, int stride, void* stream);
/* w [rows][cin][taps] -> out [rows][taps][cin] (rows = draws * cout): the tap-major weight layout from the reference's. */
int bbb_w_tap_major(const float* w, float* out, int64_t rows, int cin, int taps, void* stream);
int bbb_lrt_conv2d_chwn_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                            const float* b_mu, const float* b_var, float* y,
                            float* act_mu_out, float* act_var_out, const float* eps_ext,
                            uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                            const uint32_t* call_dev, void* stream);

/*
 * Split-contraction forms of the two batch-innermost entry points above (ABI 6; a property of the LAYER since ABI 8).
 * A draw of a late AlexNet layer is a few dozen workgroups, each bounded by its serial k loop (conv4: 48 tiles of 32 k on 128 of
 * the 256 CUs).  For such layers -- at most 16 (output pixel, 64-channel tile) groups and at least 16 k tiles; the plan depends on
 * the layer's GEOMETRY only, never on batch, draws or work units -- the contraction of every output tile is cut into k_split = 2..4
 * consecutive k ranges whose partial sums are added in range order: ((p0 + p1) + p2) + p3.  Two executions of that one
 * definition, chosen by launch size, same bits:
 *   - across workgroups (launches of <= 384 items of 64 images, when scratch is given): each range is its own workgroup, partial
 *     accumulator tiles go to `scratch`, the last arriver adds them in range order, applies bias / activation / the LRT sampling
 *     step and stores y;
 *   - inside one workgroup (any other launch; no scratch needed): the k loop restarts its accumulator at every range boundary.
 * So one draw computed alone, the same draw inside a 10-draw launch, as a work unit of a sharded step or as one of several steps
 * per launch is the same number, bit for bit.  Against the UNSPLIT chain of bbb_conv2d_chwn_fwd the partial sums round
 * separately: ~1e-7 relative, 1-2e-6 of max|logit| through a model.
 *   bbb_conv2d_chwn_splitk_scratch: the layer's plan -> *k_split (1 = no split: same as the plain entry point) and the scratch
 *     bytes THIS launch (d->draws, d->batch) needs for the cross-workgroup form (0: it runs the in-workgroup form).
 *     lrt != 0: sizes for the LRT entry point (two accumulator sets).
 *   scratch: device memory, 256-byte aligned, ZERO-FILLED before its first use (arrival tickets live at its start and every
 *     launch leaves them zero again); one scratch buffer per stream in flight; NULL = always the in-workgroup form.
 * k_split must be the planned value (BBB_EINVAL otherwise); k_split <= 1 behaves exactly like the plain entry point.
 */
int64_t bbb_conv2d_chwn_splitk_scratch(const bbb_conv_desc_t* d, int lrt, int32_t* k_split);
int bbb_conv2d_chwn_splitk_fwd(const bbb_conv_desc_t* d, const float* x, const float* w, const float* bias, float* y,
                               int k_split, void* scratch, int64_t scratch_bytes, void* stream);
int bbb_lrt_conv2d_chwn_splitk_fwd(const bbb_conv_desc_t* d, const float* x, const float* w_mu, const float* w_var,
                                   const float* b_mu, const float* b_var, float* y,
                                   float* act_mu_out, float* act_var_out, const float* eps_ext,
                                   uint64_t seed, uint32_t call0, uint32_t stream_id, int sample,
                                   const uint32_t* call_dev, int k_split, void* scratch, int64_t scratch_bytes, void* stream);

/* nn.MaxPool2d(kernel_size=k, stride=s) (no padding, floor mode; models/BayesianModels/BayesianAlexNet.py:37)
 * on batch-innermost planes: x [planes][h][w][B] -> y [planes][(h-k)/s+1][(w-k)/s+1][B]. */
int bbb_maxpool_chwn(const float* x, float* y, int64_t planes, int h, int w, int batch, int k, int s, void* stream);

/*
 * Training extension: backward of [fused activation -> nn.MaxPool2d(k, s)] on batch-innermost planes (k = 0: activation only).
 *   y      [planes][h][w][B]   the layer's ACTIVATED output (what the forward GEMM stored)
 *   g_out  [planes][hp][wp][B] gradient w.r.t. the pooled output (k = 0: same shape as y)
 *   g_pre  [planes][h][w][B]   gradient w.r.t. the pre-activation: the first maximum of every window receives that window's
 *                              gradient (torch's max_pool2d backward), times act'(.) recovered from y (Softplus: 1 - exp(-y)).
 * Gather formulation: deterministic, handles overlapping windows (k > s).  out_plane_pitch (elements, multiple of 4; 0 = dense
 * h*w*B): pitch between the planes of g_pre -- the first layer's gradient is written straight into the padded-row matrix its
 * weight-gradient GEMM reads (rows of exactly 2^n bytes would all fall into one memory channel).
 */
int bbb_pool_act_bwd_chwn(const float* g_out, const float* y, float* g_pre, int64_t planes, int h, int w, int batch,
                          int k, int s, int act, int64_t out_plane_pitch, void* stream);

/*
 * The same pass for a local-reparameterisation layer (layers/BBB_LRT/BBBConv.py:71-81: out = act_mu + sqrt(act_var) * eps, then
 * the activation, then the pool): besides g_mu = the gradient w.r.t. act_mu (what bbb_pool_act_bwd_chwn calls g_pre) it writes
 *   g_var = g_mu * (v - act_mu) / (2 * act_var),   v = the pre-activation recovered from y (Softplus: y + log(1 - exp(-y)) below
 *           the threshold of 20; ReLU: y, and no gradient where y == 0; act = 0: y itself)
 * -- the gradient w.r.t. act_var, since sqrt(act_var) * eps = v - act_mu.  act_mu / act_var: [moment_planes][h][w][B] with
 * moment_planes == planes, or a divisor of it when consecutive groups of moment_planes planes (the draws of a first layer whose
 * input and weights every draw shares) were sampled from ONE pair of moments.  g_mu and g_var share out_plane_pitch.
 * g_out2 / x_out (both or neither; ABI 13): the incoming gradient is g_out + 2 * x_out * g_out2 -- the two input gradients of the
 * LRT layer above (through its mean and its variance weights) and this layer's output x_out [x_planes][hp][wp][B] (x_planes divides
 * planes), combined on the fly with bbb_lrt_glue mode 1's arithmetic instead of by a launch of its own.
 */
int bbb_lrt_pool_act_bwd_chwn(const float* g_out, const float* y, const float* act_mu, const float* act_var, float* g_mu,
                              float* g_var, int64_t planes, int64_t moment_planes, int h, int w, int batch, int k, int s, int act,
                              int64_t out_plane_pitch, const float* g_out2, const float* x_out, int64_t x_planes, void* stream);

/*
 * E noise draws from ONE pair of LRT moments, batch-innermost: y[e] = act(act_mu + sqrt(act_var) * eps[e]) with eps exactly
 * as bbb_lrt_conv2d_chwn_fwd would have drawn it for draw e (layers/BBB_LRT/BBBConv.py:78-81).  For an LRT layer whose input
 * is the same for every draw (the first layer of a model) this replaces E identical pairs of contractions by one.
 *   act_mu, act_var: [channels][pixels][batch] (the act_mu_out / act_var_out of a draws = 1, sample = 0 launch)
 *   y: [draws][channels][pixels][batch]
 *   b_offset: global index of local image 0 (batch-parallel shards; bbb_conv_desc_t::b_offset of the GEMM form), ABI 6
 */
int bbb_lrt_sample_chwn(const float* act_mu, const float* act_var, float* y, int draws, int channels, int pixels, int batch,
                        int b_offset, int act, uint64_t seed, uint32_t call0, uint32_t stream_id, const uint32_t* call_dev,
                        void* stream);

/*
 * The LRT sampling step alone, reference layout: y = act_mu + sqrt(act_var) * eps over `draws` contiguous slabs of n
 * elements (layers/BBB_LRT/BBBConv.py:78-81), eps = element i of noise stream (seed, call0 + e, stream_id) -- the same
 * numbers bbb_lrt_conv2d_fwd draws.  Used by the training path, which computes the two moments with split-K launches.
 */
int bbb_lrt_sample_nchw(const float* act_mu, const float* act_var, float* y, int64_t n, int draws, uint64_t seed,
                        uint32_t call0, uint32_t stream_id, const uint32_t* call_dev, void* stream);

/*
 * BF16 storage variants of the batch-innermost path (BASELINE.json configs[1]: "BBB layers, bf16"); fp32 accumulation,
 * fp32 bias and fp32 epilogue (bias + activation), one rounding (nearest-even) when a value is stored as bf16.
 *   x: [draws|1][cin][h][w][B] bf16   (B % 8 == 0, 16-byte aligned, draw strides multiples of 8 elements)
 *   w: [draws|1][cout][Kp] bf16       K = cin*kh*kw, Kp = K rounded up to 8, pad columns zero -- what bbb_reparam_kl_fwd
 *                                     writes for a segment with w_row_len = K.  Column order inside a row: the reference's
 *                                     (ci, r, q), or with BBB_BF16_W_TAP_MAJOR (needs cin % 8 == 0) (r, q, ci) = a segment
 *                                     written with w_taps = kh*kw.  Tap-major rows let the kernel skip the kernel taps
 *                                     that fall into the zero padding (as the fp32 kernel does) and still move weights as
 *                                     16-byte vectors; with the reference order those taps are multiplied by zeros.
 *   y: [draws][cout][ho][wo][B]       bf16, or fp32 with BBB_BF16_OUT_F32 (the logits layer feeding bbb_mc_tail)
 * d->w_draw_stride counts bf16 elements (cout*Kp per draw when dense).  Same contraction as bbb_conv2d_chwn_fwd
 * (layers/BBB/BBBConv.py:77, BBBLinear.py:70) on v_mfma_f32_32x32x16_bf16.  BBB layers only (no LRT variant).
 * Channel-interleaved activations (ABI 10): [draws][C / 8][h][w][B][8] -- the 8 channels 8g .. 8g + 7 of an image are 16 adjacent
 * bytes, so a lane's MFMA operand (8 consecutive channels of one image) is ONE 16-byte load and a conv layer needs neither LDS
 * staging nor transposing reads for its images.  Same element count and draw strides as the batch-innermost tensor; same values.
 *   BBB_BF16_X_C8    x is in that layout.  Accepted where the library has a kernel that reads it: tap-major rows of 32 input
 *                    channels, 5 x 5 taps, stride 1, dilation 1, padding < 5, bf16 output (3Conv3FC conv2: a strip of three output
 *                    pixels per workgroup, every input position fetched once for all the pixels it is a tap of); BBB_EINVAL else.
 *   BBB_BF16_OUT_C8  y is written in that layout (cout % 8 == 0).  Accepted by the pooled first-layer forms (d->pool != 0) and
 *                    together with BBB_BF16_X_C8; BBB_EINVAL else.
 * Bit for bit the results of the batch-innermost forms (the same MFMA sequence per output element).
 */
#define BBB_BF16_OUT_F32      1u
#define BBB_BF16_W_TAP_MAJOR  2u
#define BBB_BF16_X_C8         4u   /* x is channel-interleaved: [draws|1][cin / 8][h][w][B][8] (ABI 10) */
#define BBB_BF16_OUT_C8       8u   /* y is written channel-interleaved: [draws][cout / 8][ho][wo][B][8] (ABI 10) */
int bbb_conv2d_chwn_bf16_fwd(const bbb_conv_desc_t* d, const void* x, const void* w, const float* bias, void* y,
                             uint32_t flags, void* stream);
/* nn.MaxPool2d(k, s) on [planes][h][w][B] bf16 (B % 8 == 0); exact (max commutes with the rounding). */
int bbb_maxpool_chwn_bf16(const void* x, void* y, int64_t planes, int h, int w, int batch, int k, int s, void* stream);
/* fp32 [batch][plane] (an NCHW tensor, plane = C*H*W) -> bf16 [plane][batch] (batch-innermost), nearest-even. */
int bbb_nchw_to_chwn_bf16(const float* x, void* y, int batch, int64_t plane, void* stream);
/* The same for `slices` batches stored back to back (ABI 7): x [slices][batch][plane] -> y [slices][plane][batch], one launch
 * (a rank's batch slices; the batches of several one-draw steps that share a launch). */
int bbb_nchw_to_chwn_bf16_slices(const float* x, void* y, int batch, int64_t plane, int slices, void* stream);

/*
 * Monte-Carlo tail (main_bayesian.py:49,53 / :78,80 + utils.py:14-22): per draw log_softmax over
 * classes, then log-sum-exp over the local draws.
 *   logits [draws][B][C]  ->  lse [B][C] = log sum_e exp(log_softmax(logits[e])[b][c])
 * log_outputs = lse - log(E_total) (applied when mean_over > 0: lse - log(mean_over)).
 * Ranks of an ensemble-sharded job combine their lse blocks with one more log-sum-exp.
 */
int bbb_mc_tail(const float* logits, int draws, int batch, int classes, int mean_over,
                float* lse_out, void* stream);

/* bbb_mc_tail for logits stored batch-innermost, [draws][C][B] (output of the batched ensemble path); lse_out is [B][C]. */
int bbb_mc_tail_cb(const float* logits, int draws, int batch, int classes, int mean_over,
                   float* lse_out, void* stream);
/* Backward of bbb_mc_tail_cb (training extension, ABI 12; loss.backward() of main_bayesian.py:57 through log_softmax + logmeanexp):
 * g_logits [draws][classes][batch] from g_lse [batch][classes] and the forward's lse [batch][classes]. */
int bbb_mc_tail_cb_bwd(const float* logits, const float* lse, const float* g_lse, float* g_logits, int draws, int batch, int classes,
                       int mean_over, void* stream);

/* Training extension: the ELBO of one Monte-Carlo step on the device (metrics.py:7-14: nll_loss(log_outputs, target, mean) *
 * train_size + beta * kl; main_bayesian.py:55-56).  lse [batch][classes] = bbb_mc_tail_cb's output, target [batch] int64 (entries
 * outside [0, classes) contribute nothing), kl / loss_out device scalars; beta_dev != NULL: the KL weight is read from the device
 * (a captured step's run-time value) instead of `beta`.  One block, fixed summation order. */
int bbb_elbo_cb_fwd(const float* lse, const int64_t* target, const float* kl, float beta, const float* beta_dev, float train_size,
                    int batch, int classes, float* loss_out, void* stream);
/* Its backward in one launch: g_logits [draws][classes][batch] = d loss / d logits through log_softmax + logmeanexp (as
 * bbb_mc_tail_cb_bwd with g_lse = -(train_size / batch) * g_loss at the target class), g_kl (device scalar, may be NULL) =
 * beta * g_loss. */
int bbb_elbo_cb_bwd(const float* logits, const float* lse, const int64_t* target, const float* g_loss, float beta,
                    const float* beta_dev, float train_size, float* g_logits, float* g_kl, int draws, int batch, int classes,
                    int mean_over, void* stream);

/* The same for a rank's WORK UNITS (bbb_conv_desc_t: unit u = unit_off + e is draw u / slices, batch slice u % slices):
 * logits [units][C][batch_slice] -> lse_out [slices * batch_slice][C]; image b of slice s gets the log-sum-exp over the local
 * units of that slice, -inf where the rank holds none (the ranks' blocks are then combined by one more log-sum-exp). */
int bbb_mc_tail_units(const float* logits, int units, int slices, int unit_off, int batch_slice, int classes, int mean_over,
                      float* lse_out, void* stream);

/* bbb_mc_tail_units + the end of a captured Monte-Carlo step in the same launch (ABI 6): kl_out = kl_in * kl_scale (kl_in: the KL
 * of ONE forward from bbb_reparam_kl_fwd; the reference sums it once per forward, main_bayesian.py:76-77; kl_out NULL = skip) and
 * *counter += counter_add (the device-side call counter that bbb_reparam_kl_fwd / the LRT entry points read through `call_dev`:
 * all of a step's readers run before its tail, so the next replay of the graph draws fresh noise; counter NULL = skip). */
int bbb_mc_tail_units_step(const float* logits, int units, int slices, int unit_off, int batch_slice, int classes, int mean_over,
                           float* lse_out, const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter,
                           uint32_t counter_add, void* stream);

/* Several Monte-Carlo steps in one set of launches (ABI 8): logits [groups * draws][C][batch] hold `groups` consecutive steps of
 * `draws` forwards each, step g on its own batch (slab g * draws + j = draw j of step g; main_bayesian.py:73-80 run `groups`
 * times) -> lse_out [groups * batch][C], block g = the log-sum-exp over step g's draws (minus log(mean_over) when > 0).
 * kl_in / kl_scale / kl_out / counter / counter_add as in bbb_mc_tail_units_step (all NULL / 0 = plain tail). */
int bbb_mc_tail_groups_step(const float* logits, int groups, int draws, int batch, int classes, int mean_over,
                            float* lse_out, const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter,
                            uint32_t counter_add, void* stream);

/* A RANK'S SHARE of a group of steps (ABI 8): the `slabs` local draws are draws first_off, first_off + 1, ... of the draw-major
 * enumeration (step g, draw j) -> g * draws + j of `steps` consecutive local steps (the share may start and end in the middle of a
 * step): lse_out [steps * batch][C], block k = the log-sum-exp over the LOCAL draws of local step k (no mean; -inf where it holds none);
 * the ranks' blocks are combined by one more log-sum-exp.  kl / counter arguments as in bbb_mc_tail_units_step. */
int bbb_mc_tail_share_step(const float* logits, int slabs, int steps, int draws, int first_off, int batch, int classes,
                           float* lse_out, const float* kl_in, float kl_scale, float* kl_out, uint32_t* counter,
                           uint32_t counter_add, void* stream);

/*
 * Uncertainty decomposition over `draws` stochastic forwards (uncertainty_estimation.py:37-58 per image, :61-102 per
 * batch): logits [draws][B][C] -> pred = mean logits, epistemic = mean (p_hat - p_bar)^2, aleatoric = p_bar - mean p_hat^2,
 * each [B][C]; p_hat = softmax(logits), or softplus(logits)/sum when `normalized`.
 */
int bbb_uncertainty(const float* logits, int draws, int batch, int classes, int normalized,
                    float* pred, float* epistemic, float* aleatoric, void* stream);

/* [rows][cols] -> [cols][rows] (e.g. an NCHW batch [B][C*H*W] into the batch-innermost [C*H*W][B] layout). */
int bbb_transpose2d(const float* in, float* out, int64_t rows, int64_t cols, void* stream);

/* Batched strided transpose: out[i1*out_b1 + i2*out_b2 + c*out_col + r] = in[i1*in_b1 + i2*in_b2 + r*in_row + c] for r < rows,
 * c < cols, i1 < nb1, i2 < nb2 (nb1*nb2 <= 65535).  Training extension: operand layouts of the role-swapped weight gradient. */
int bbb_transpose_batched(const float* in, float* out, int rows, int cols, int nb1, int nb2, int64_t in_b1, int64_t in_b2,
                          int64_t in_row, int64_t out_b1, int64_t out_b2, int64_t out_col, void* stream);

/* The same with three batch dimensions nb[3] (product <= 65535; strides in_b[3] / out_b[3]) and a summed one:
 * out[i1*out_b[0] + i2*out_b[1] + i3*out_b[2] + c*out_col + r] = sum_{s < nsum, ascending} in[i1*in_b[0] + i2*in_b[1] + i3*in_b[2]
 * + s*in_sum + r*in_row + c].  Training extension: the batch chunks of a role-swapped weight gradient summed in a fixed order
 * while the taps move innermost; an output gradient written straight into the chunked weight-operand layout.  square_off != 0: every
 * output's square is written too, square_off elements behind it (x and x^2 of an LRT layer's weight gradients in one pass). */
int bbb_transpose_sum_batched(const float* in, float* out, int rows, int cols, const int32_t* nb, const int64_t* in_b,
                              const int64_t* out_b, int64_t in_row, int64_t out_col, int nsum, int64_t in_sum, int64_t square_off,
                              void* stream);

/* Training extension: out [draws][cin][cout][kh*kw] = w [draws][cout][cin][kh*kw] with the taps reversed (spatial flip +
 * channel transpose: the weights of the stride-1 input-gradient convolution). */
int bbb_flip_transpose_w(const float* w, float* out, int64_t draws, int cout, int cin, int khkw, void* stream);
/* The same over two sources: out's draws [0, draws_each) from w0, [draws_each, 2*draws_each) from w1 (the mean and variance
 * weights of a local-reparameterisation layer as one operand: both input gradients in one launch of the forward kernel). */
int bbb_flip_transpose_w_pair(const float* w0, const float* w1, float* out, int64_t draws_each, int cout, int cin, int khkw,
                              void* stream);
/* ... and for up to 16 weight sets in ONE launch (every layer's input-gradient weights of a training step): segment i flips
 * `draws` sets [cout][cin][khkw] from w0 -- or, with w1 != NULL, draws / 2 from w0 then draws / 2 from w1 -- into out. */
typedef struct {
    const float* w0;
    const float* w1;      /* NULL, or the second source (draws must then be even) */
    float* out;           /* [draws][cin][cout][khkw] */
    int64_t draws;
    int32_t cout, cin, khkw, reserved;
} bbb_flip_seg_t;
int bbb_flip_transpose_w_multi(const bbb_flip_seg_t* segs, int n_segs, void* stream);

/* Training extension: im2col of an NCHW batch x [batch][cin][h][w] (geometry from d; draws / strides / act ignored) into
 * out [ho*wo][batch][Jp], Jp = cin*kh*kw rounded up to 4 (pad columns zero): the K-major operand of the first layer's
 * weight-gradient GEMM (see bbb_hip/ops.py: conv2d_chwn_weight_grad_shared_input). */
int bbb_im2col_pbj(const float* x, float* out, const bbb_conv_desc_t* d, void* stream);

/* Training extension, the small steps between the gradient GEMMs (ABI 8; deterministic, no atomics):
 * bbb_plane_sum: out[r] = sum over o < outer, j < cols of x[o*outer_stride + r*row_pitch + j] -- bias gradients (the gradient w.r.t.
 *   a layer's pre-activation summed over pixels and images per (draw, channel) plane; outer > 1 also sums over draws: LRT biases).
 * bbb_sum_leading: out[i] = sum over o < outer of x[o*n + i], added in index order -- the draws of a first LRT layer share
 *   one pair of moments, and shared-weight gradients are summed over draws.
 * bbb_lrt_glue: mode 0: out = x*x (the variance contraction's operand, layers/BBB_LRT/BBBConv.py:73); mode 1: out[i] = a[i] +
 *   2*x[i % x_n]*b[i] -- the LRT input gradient dgrad(g_mu, mu) + 2 x dgrad(g_var, sigma^2) with x one slab or one per draw. */
int bbb_plane_sum(const float* x, float* out, int64_t outer, int64_t rows, int64_t cols, int64_t row_pitch, int64_t outer_stride,
                  void* stream);
int bbb_sum_leading(const float* x, float* out, int64_t outer, int64_t n, void* stream);
int bbb_lrt_glue(const float* a, const float* x, const float* b, float* out, int64_t n, int64_t x_n, int mode, void* stream);

/* Library / device introspection (host-only). */
int bbb_abi_version(void);
const char* bbb_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* BBB_HIP_H */
