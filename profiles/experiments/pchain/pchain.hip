// Persistent "chain" kernel: every convolution / linear contraction and every max-pool of ONE Monte-Carlo step
// (main_bayesian.py:73-80 = num_ens x ModuleWrapper.forward, layers/misc.py:16-25, for models built like
// models/BayesianModels/BayesianAlexNet.py:35-53) in ONE launch on gfx950.
//
// Why: a layer launch of a num_ens = 10 step is 1-2.5 rounds of workgroups -- ramp, tail and a kernel boundary per layer,
// and the per-layer launches of different Monte-Carlo draws cannot overlap although draw e of layer L+1 only needs draw e
// of layer L.  Here a fixed grid of resident workgroups runs a READY-FIRST list scheduler over work items (stage, slab,
// tile): one claim counter per (stage, XCD, slab) and one completion counter per (stage, slab).  To get work, wave 0 of a
// workgroup loads all counters (one lane per slab, all stages in flight at once: one memory round trip), and claims an
// item of the DEEPEST stage that has unclaimed items whose input slab is complete -- so a workgroup never waits on a
// dependency while other work is ready (an in-order queue with blocking waits was built first and measured: half of every
// workgroup's lifetime went to head-of-line blocking), and draws run staggered through the layers by themselves: as soon
// as the first draws finish a layer, freed workgroups prefer their next layer over the later draws' current one.
//   producer: output stores are write-through (sc1), 16 bytes per lane -> every wave s_waitcnt vmcnt(0) -> barrier -> ONE
//             relaxed agent-scope atomic increment of done[stage][slab];
//   consumer: an item is only claimed after done[dep][slab] was seen complete; operand loads bypass the CU's L1 (sc1).  A
//             line is only ever read after its final value was written through, and intermediate slabs are multiples of
//             128 B, so no L2 can hold a stale copy (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by
//             other CUs' stores).
// Every GEMM item runs pconv_item -- the same instruction sequence as the per-layer kernel, hence the same bits.
// Termination: a workgroup exits when every counter of its XCD's share is exhausted; an unready dependency always belongs
// to items that are claimed (running) or claimable by resident workgroups, and the grid is sized from the occupancy query;
// the idle loop is bounded anyway (a timeout sets the error word of the workspace: wrong numbers, never a hang).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bbb_hip.h"
#include "bbb_common.cuh"
#include "pconv_args.h"
#include "pconv_body.cuh"

namespace {

using namespace pconv;

constexpr int kMaxStages = BBB_CHAIN_MAX_STAGES;
constexpr int kMaxSlabs = 64;                    // slabs (Monte-Carlo draws / work units) per launch: counter row length
constexpr int kPoolV4 = 4;                       // float4 outputs per thread of a pooling item
constexpr int kPoolItem = kThreads * kPoolV4;    // float4 outputs per pooling item
// workspace layout (int32 words): [8] error word; done[stage][slab] (publish counters, never polled) from kWsDone;
// ready[xcd]: one 128-byte line per XCD holding a bitmap of complete (stage, slab) pairs -- the ONLY words idle workgroups
// poll, each XCD its own copy, so that 1024 pollers never queue up in front of the publishers' atomics (measured: polling the
// counters themselves tripled every dequeue and made a stage transition cost ~25 us); claimed[stage][xcd][slab] after it.
constexpr int kWsErr = 8, kWsDone = 32;
constexpr int kWsReady = kWsDone + kMaxStages * kMaxSlabs;        // multiple of 32 words: line aligned
constexpr int kWsClaim = kWsReady + 8 * 32;
constexpr int kWsCu = kWsClaim + kMaxStages * 8 * kMaxSlabs;      // resident-workgroup count per physical CU (1024 keys)
constexpr int kWsWords = kWsCu + 1024;

struct StageDev {                  // 176 bytes
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    int64_t x_ds, w_ds, b_ds;
    int32_t B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
    int32_t Kp, act, Mtiles, nbt, Ntiles;
    int32_t unit_div, unit_off, x_mod;
    uint32_t x_inv;
    int32_t kind;                  // 0: GEMM item, 1: max-pool item
    int32_t dep;                   // stage whose slab e this stage's slab e reads; -1: input complete before the launch
    int32_t ipd;                   // items per slab; XCD x owns items [x * chunk, min(ipd, (x + 1) * chunk)), chunk = ceil(ipd / 8)
    int32_t dep_ipd;               // items per slab of the dep stage = value of its completion counter when done
};

struct ChainArgs {
    StageDev st[kMaxStages];
    int32_t nst, slabs;
    int32_t shallow_first, prof_off;   // policy flag; PCHAIN_PROFILE builds: word offset of the per-workgroup timing block in ws
    int32_t cus_per_xcd, no_cu_balance;
    int32_t* ws;
};
static_assert(sizeof(ChainArgs) <= 4000, "kernel arguments must stay below 4 KB");

__device__ __forceinline__ void pool_item(const StageDev& st, int e, int local) {
    const int tid = threadIdx.x;
    const int B4 = st.B >> 2;
    const int64_t in_elems = (int64_t)st.Cin * st.H * st.W * st.B, out_elems = (int64_t)st.Cin * st.Ho * st.Wo * st.B;
    const int total4 = (int)(out_elems >> 2);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(st.x + (int64_t)e * in_elems), 0,
                                                                         (int)(in_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(st.y + (int64_t)e * out_elems, 0, (int)(out_elems * 4),
                                                                         0x00020000);
    const int k = st.kh, s = st.sh;
#pragma unroll
    for (int u = 0; u < kPoolV4; ++u) {
        const int i = local * kPoolItem + u * kThreads + tid;
        if (i < total4) {
            const int b4 = i % B4;
            int t = i / B4;
            const int ow = t % st.Wo;
            t /= st.Wo;
            const int oh = t % st.Ho;
            const int pl = t / st.Ho;
            const uint32_t base = (uint32_t)(((pl * st.H + oh * s) * st.W + ow * s) * B4 + b4) * 16u;
            f32x4 m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, base, 0, 16));
            for (int a = 0; a < k; ++a)
                for (int c = 0; c < k; ++c) {
                    const f32x4 v = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, base + (uint32_t)((a * st.W + c) * B4) * 16u, 0, 16));
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
                }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t, m),
                                                   yrs, (uint32_t)i * 16u, 0, 16);
        }
    }
}

// -DPCHAIN_PROFILE (experiment builds only, profiles/experiments/chain_prof.py): every workgroup adds up the 100 MHz
// s_memrealtime ticks it spends dequeuing, waiting for a dependency, running items and publishing, and writes them behind
// the completion counters of the workspace (8 x int64 per workgroup).
#ifdef PCHAIN_PROFILE
#define PROF_NOW() __builtin_amdgcn_s_memrealtime()
#define PROF_ADD(acc, t0) do { const uint64_t t1_ = __builtin_amdgcn_s_memrealtime(); acc += t1_ - t0; t0 = t1_; } while (0)
#else
#define PROF_NOW() 0
#define PROF_ADD(acc, t0) do { } while (0)
#endif

constexpr int kMaxPairs = 256;                   // (stage, slab) pairs one scan covers: 4 per lane of the scanning wave
constexpr int kScanJ = kMaxPairs / 64;

template <int BM, bool ILV>
__global__ __launch_bounds__(kThreads, 4) void pchain_kernel(const ChainArgs a) {
    __shared__ int s_pick[4];                           // stage, slab, item in slab (or stage = -1: exit)
    __shared__ int t_ipd[kMaxStages], t_dep[kMaxStages];    // per-stage scheduling facts, for per-lane lookups
    const int tid = threadIdx.x;
    [[maybe_unused]] uint64_t pt = PROF_NOW(), p_deq = 0, p_wait = 0, p_exec = 0, p_pub = 0, p_items = 0, p_waits = 0;
    [[maybe_unused]] const uint64_t p_start = pt;
    const int xcd = blockIdx.x & 7;                     // the XCD the dispatcher is observed to use (a wrong guess costs L2 hits only)
    int32_t* const err = a.ws + kWsErr;
    int32_t* const done = a.ws + kWsDone;
    int32_t* const ready = a.ws + kWsReady;
    int32_t* const claimed = a.ws + kWsClaim;
#pragma unroll
    for (int s = 0; s < kMaxStages; ++s)
        if (tid == s && s < a.nst) {
            t_ipd[s] = a.st[s].ipd;
            t_dep[s] = a.st[s].dep;
        }
    const int npairs = a.nst * a.slabs;
    // CU-aware claiming.  The hardware dispatcher spreads the workgroups of an ordinary launch evenly over the CUs; resident
    // workgroups that claim items on their own would not: with 1024 of them and 320 ready items, some CUs would run four
    // items (each at a quarter of the matrix pipe) next to idle CUs (measured: workgroups idle 40 % of their lifetime, items
    // 30 % slower).  So every workgroup learns its rank k among the workgroups resident on ITS CU (hardware ids -> one atomic
    // at start), and rank k only claims while more than k * (CUs of the XCD) ready items are left in the XCD's shares:
    // R <= 32 items run one per CU, R <= 64 two per CU, and so on.  Rank 0 always claims: progress never depends on it.
    int cu_rank = 0;
    if (tid < 64 && !a.no_cu_balance) {
        uint32_t hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const int key = (int)(((xcc & 7u) << 7) | (((hw >> 13) & 7u) << 4) | ((hw >> 8) & 15u));     // XCC | SE, SH | CU
        int r = 0;
        if (tid == 0) r = __hip_atomic_fetch_add(a.ws + kWsCu + key, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cu_rank = __builtin_amdgcn_readfirstlane(r);
    }
    for (;;) {
        __syncthreads();                                // previous item: LDS and s_pick no longer in use (first pass: tables written)
        if (tid < 64) {
            // ---- wave 0: find work.  Pair p = lane + 64 j is (stage, slab) in PRIORITY order: deepest stage first (or
            //      shallowest with the flag), slabs ascending.  All counters are loaded before any is looked at (one memory
            //      round trip); the first pair with unclaimed items of this XCD's share and a complete input slab wins. ----
            int pick_s = -1, pick_e = 0, pick_i = 0;
            for (int idle = 0;; ++idle) {
                int c[kScanJ], d[kScanJ], sh[kScanJ], st_[kScanJ];
#pragma unroll
                for (int j = 0; j < kScanJ; ++j) {
                    const int p = tid + 64 * j;
                    c[j] = 0x7fffffff;
                    d[j] = 1;                               // input slab complete? (1 = no dependency)
                    sh[j] = 0;
                    st_[j] = 0;
                    if (p < npairs) {
                        const int ps = p / a.slabs, e = p - ps * a.slabs;
                        const int s = a.shallow_first ? ps : a.nst - 1 - ps;
                        st_[j] = (s << 8) | e;
                        const int ipd = t_ipd[s], dep = t_dep[s];
                        const int chunk = (ipd + 7) >> 3;
                        int share = ipd - xcd * chunk;
                        sh[j] = share < 0 ? 0 : (share > chunk ? chunk : share);
                        c[j] = __hip_atomic_load(claimed + (s * 8 + xcd) * kMaxSlabs + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (dep >= 0) {
                            const int bit = dep * a.slabs + e;
                            d[j] = (__hip_atomic_load(ready + xcd * 32 + (bit >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (bit & 31)) & 1;
                        }
                    }
                }
                bool any_left = false;
                int R = 0;                                   // ready, unclaimed items of this XCD's shares
#pragma unroll
                for (int j = 0; j < kScanJ; ++j) R += (c[j] < sh[j] && d[j] != 0) ? sh[j] - c[j] : 0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) R += __shfl_xor(R, off, 64);
                const bool mine = R > cu_rank * a.cus_per_xcd;
#pragma unroll
                for (int j = 0; j < kScanJ; ++j) {
                    const bool left = c[j] < sh[j];
                    any_left |= __builtin_amdgcn_ballot_w64(left) != 0;
                    uint64_t mask = __builtin_amdgcn_ballot_w64(mine && left && d[j] != 0);
                    while (mask != 0 && pick_s < 0) {
                        const int src = __builtin_ctzll(mask);
                        const int se = __builtin_amdgcn_readlane(st_[j], src);
                        const int share = __builtin_amdgcn_readlane(sh[j], src);
                        const int s = se >> 8, e = se & 0xff;
                        int idx = 0;
                        if (tid == 0)
                            idx = __hip_atomic_fetch_add(claimed + (s * 8 + xcd) * kMaxSlabs + e, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        idx = __builtin_amdgcn_readfirstlane(idx);
                        if (idx < share) {
                            pick_s = s;
                            pick_e = e;
                            pick_i = xcd * ((t_ipd[s] + 7) >> 3) + idx;
                        }
                        mask &= mask - 1;                   // lost the race for this pair's last items: next pair
                    }
                }
                if (pick_s >= 0 || !any_left) break;        // got an item, or this XCD's shares are all claimed: done
                // ready work exists only after a running item completes: back off (0.5 ... 3.4 us between polls)
                if (idle < 4) __builtin_amdgcn_s_sleep(16);
                else if (idle < 16) __builtin_amdgcn_s_sleep(48);
                else __builtin_amdgcn_s_sleep(127);
                if (idle > (1 << 20)) {                     // ~ seconds: give up, flag the error, never hang
                    if (tid == 0) __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                PROF_ADD(p_wait, pt);
                ++p_waits;
            }
            if (tid == 0) {
                s_pick[0] = pick_s;
                s_pick[1] = pick_e;
                s_pick[2] = pick_i;
            }
        }
        __syncthreads();
        const int stage = __builtin_amdgcn_readfirstlane(s_pick[0]);
        const int e = __builtin_amdgcn_readfirstlane(s_pick[1]);
        const int item_in_slab = __builtin_amdgcn_readfirstlane(s_pick[2]);
        PROF_ADD(p_deq, pt);
        if (stage < 0) break;
        const StageDev& st = a.st[stage];
        if (st.kind == 0) {
            PConvArgs p = {};
            p.x = st.x; p.w = st.w; p.bias = st.bias; p.y = st.y;
            p.x_ds = st.x_ds; p.w_ds = st.w_ds; p.b_ds = st.b_ds;
            p.B = st.B; p.Cin = st.Cin; p.H = st.H; p.W = st.W; p.Cout = st.Cout; p.kh = st.kh; p.kw = st.kw;
            p.sh = st.sh; p.sw = st.sw; p.ph = st.ph; p.pw = st.pw; p.dh = st.dh; p.dw = st.dw; p.Ho = st.Ho; p.Wo = st.Wo;
            p.y_ds = (int64_t)st.Cout * st.Ho * st.Wo * st.B;
            p.K = st.Cin * st.kh * st.kw; p.Kp = st.Kp; p.khkw = st.kh * st.kw; p.act = st.act;
            p.Mtiles = st.Mtiles; p.nbt = st.nbt; p.Ntiles = st.Ntiles;
            p.x_inv = st.x_inv;
            p.unit_div = st.unit_div; p.unit_off = st.unit_off; p.x_mod = st.x_mod;
            pconv_item<BM, false, ILV, true>(p, (int64_t)e * st.ipd + item_in_slab);
        } else {
            pool_item(st, e, item_in_slab);
        }
#ifdef PCHAIN_PROFILE
        if (tid == 0 && p_items < 30) {                     // per-item log: start, end (100 MHz ticks), stage, slab
            uint32_t* lg = reinterpret_cast<uint32_t*>(a.ws + a.prof_off + 2048 * 16) + ((size_t)blockIdx.x * 32 + p_items) * 4;
            lg[0] = (uint32_t)pt; lg[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); lg[2] = (uint32_t)stage; lg[3] = (uint32_t)e;
        }
#endif
        PROF_ADD(p_exec, pt);
        ++p_items;
        // ---- publish: the write-through stores of EVERY wave have left, then one counter increment ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < 64) {
            int old = 0;
            if (tid == 0) old = __hip_atomic_fetch_add(done + stage * kMaxSlabs + e, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == st.ipd - 1 && tid < 8) {             // the slab's last item: mark it complete in every XCD's bitmap
                const int bit = stage * a.slabs + e;
                __hip_atomic_fetch_or(ready + tid * 32 + (bit >> 5), 1 << (bit & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        PROF_ADD(p_pub, pt);
    }
#ifdef PCHAIN_PROFILE
    if (tid == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.ws + a.prof_off) + (size_t)blockIdx.x * 8;
        o[0] = p_deq; o[1] = p_wait; o[2] = p_exec; o[3] = p_pub; o[4] = p_items; o[5] = p_waits; o[6] = p_start;
        o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

template <int BM, bool ILV>
int grid_size(int device, int cap) {
    static int cached[16] = {0};
    int per_cu = (device >= 0 && device < 16) ? cached[device] : 0;
    static int cus_cached[16] = {0};
    int cus = (device >= 0 && device < 16) ? cus_cached[device] : 0;
    if (per_cu <= 0 || cus <= 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pchain_kernel<BM, ILV>, kThreads, 0) != hipSuccess) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 0;
        if (per_cu <= 0 || cus <= 0) return 0;
        if (device >= 0 && device < 16) { cached[device] = per_cu; cus_cached[device] = cus; }
    }
    if (per_cu > 4) per_cu = 4;                          // what the GEMM item is tuned for (pconv_gemm.hip)
    if (cap > 0 && per_cu > cap) per_cu = cap;
    int g = per_cu * cus;
    g -= g % 8;                                          // the same number of workgroups per XCD
    return g;
}

template <int BM, bool ILV>
int launch_chain(ChainArgs& a, int device, int cap, hipStream_t st) {
    const int grid = grid_size<BM, ILV>(device, cap);
    if (grid <= 0) return BBB_EINVAL;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 8) return BBB_EINVAL;
    a.cus_per_xcd = cus / 8;
    hipLaunchKernelGGL((pchain_kernel<BM, ILV>), dim3((unsigned)grid), dim3(kThreads), 0, st, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int64_t bbb_chain_workspace(int nstages, int slabs) {
    if (nstages <= 0 || nstages > kMaxStages || slabs <= 0 || slabs > kMaxSlabs) return 0;
    return kWsWords;
}

extern "C" int bbb_chain_fwd(const bbb_chain_stage_t* stages, int nstages, uint32_t flags, int32_t* workspace,
                             int64_t workspace_ints, void* stream) {
    if (stages == nullptr || nstages <= 0 || nstages > kMaxStages || workspace == nullptr) return BBB_EINVAL;
    const int slabs = stages[0].conv.draws;
    if (slabs <= 0) return BBB_EINVAL;
    if (slabs > kMaxSlabs) return BBB_ESHAPE;             // one scanning lane per slab: larger ensembles run per layer
    if (workspace_ints < kWsWords) return BBB_EINVAL;
    ChainArgs a = {};
    // image-tile width of the GEMM items, one for the whole launch: 128 (two accumulator chains per wave) unless even the
    // largest stage would then have fewer than 768 items (3 per CU) -- the per-layer launcher's rule, applied to the chain
    int64_t items128 = 0;
    for (int s = 0; s < nstages; ++s) {
        const bbb_conv_desc_t& d = stages[s].conv;
        if (stages[s].kind != BBB_CHAIN_CONV || d.stride_h <= 0 || d.stride_w <= 0) continue;
        const int64_t ho = (d.h + 2 * d.pad_h - d.dil_h * (d.kh - 1) - 1) / d.stride_h + 1;
        const int64_t wo = (d.w + 2 * d.pad_w - d.dil_w * (d.kw - 1) - 1) / d.stride_w + 1;
        const int64_t n = ho * wo * ((d.batch + 127) / 128) * ((d.cout + BN - 1) / BN) * slabs;
        if (n > items128) items128 = n;
    }
    const int bm = items128 < 768 ? 64 : 128;
    int64_t max_items = 0;
    for (int s = 0; s < nstages; ++s) {
        const bbb_chain_stage_t& h = stages[s];
        const bbb_conv_desc_t& d = h.conv;
        StageDev& t = a.st[s];
        if (d.draws != slabs || h.dep >= s || h.dep < -1) return BBB_EINVAL;
        if (h.x == nullptr || h.y == nullptr) return BBB_EINVAL;
        if ((((uintptr_t)h.x | (uintptr_t)h.y) & 127u) != 0) return BBB_EALIGN;
        if (d.batch <= 0 || d.batch % 32 != 0 || d.cin <= 0 || d.h <= 0 || d.w <= 0) return BBB_ESHAPE;
        t.x = h.x; t.y = h.y; t.B = d.batch; t.Cin = d.cin; t.H = d.h; t.W = d.w; t.dep = h.dep;
        if (h.kind == BBB_CHAIN_CONV) {
            if (h.w == nullptr || (((uintptr_t)h.w | (uintptr_t)h.bias) & 3u) != 0) return BBB_EINVAL;
            if (d.cout <= 0 || d.kh <= 0 || d.kw <= 0 || d.stride_h <= 0 || d.stride_w <= 0 || d.pad_h < 0 || d.pad_w < 0 ||
                d.dil_h <= 0 || d.dil_w <= 0 || d.act < 0 || d.act > 2 || d.w_row_pitch != 0 || d.b_offset != 0)
                return BBB_EINVAL;
            const int ho = (d.h + 2 * d.pad_h - d.dil_h * (d.kh - 1) - 1) / d.stride_h + 1;
            const int wo = (d.w + 2 * d.pad_w - d.dil_w * (d.kw - 1) - 1) / d.stride_w + 1;
            if (ho <= 0 || wo <= 0) return BBB_ESHAPE;
            if ((int64_t)d.cin * d.h * d.w > 0x7fffffffLL || (int64_t)d.cin * d.kh * d.kw > 0x7fffffffLL) return BBB_ESHAPE;
            if ((int64_t)d.cin * d.h * d.w * d.batch * 4 > 0xFFFE0000LL || (int64_t)d.cout * ho * wo * d.batch * 4 > 0xFFFE0000LL ||
                ((int64_t)d.cout + 64) * d.cin * d.kh * d.kw * 4 > 0x3FFFFFFFLL || (int64_t)d.batch * 4 > 0x0FFFFFFFLL)
                return BBB_ESHAPE;
            t.x_inv = (0xFFFFFFF0u - ((uint32_t)d.batch + 512u) * 4u) & ~15u;
            if ((int64_t)d.cin * d.h * d.w * d.batch * 4 > (int64_t)t.x_inv) return BBB_ESHAPE;
            if (d.unit_div < 0 || d.unit_off < 0 || d.x_unit_mod < 0) return BBB_EINVAL;
            if (d.unit_div > 1 && d.unit_off >= d.unit_div) return BBB_EINVAL;
            if (d.x_unit_mod > 0 && d.x_unit_mod != d.unit_div) return BBB_EINVAL;
            // a stage that reads another stage's output reads slab e of it: no shared / per-slice input there
            if (h.dep >= 0 && (d.x_draw_stride == 0 || d.x_unit_mod > 0)) return BBB_EINVAL;
            t.kind = 0; t.w = h.w; t.bias = h.bias;
            t.x_ds = d.x_draw_stride; t.w_ds = d.w_draw_stride; t.b_ds = d.b_draw_stride;
            t.Cout = d.cout; t.kh = d.kh; t.kw = d.kw; t.sh = d.stride_h; t.sw = d.stride_w; t.ph = d.pad_h; t.pw = d.pad_w;
            t.dh = d.dil_h; t.dw = d.dil_w; t.Ho = ho; t.Wo = wo; t.Kp = d.cin * d.kh * d.kw; t.act = d.act;
            t.unit_div = d.unit_div; t.unit_off = d.unit_div > 1 ? d.unit_off : 0; t.x_mod = d.x_unit_mod;
            t.Ntiles = (d.cout + BN - 1) / BN;
            t.nbt = (d.batch + bm - 1) / bm;
            const int64_t mt = (int64_t)ho * wo * t.nbt;
            if (mt * t.Ntiles > 0x3fffffffLL) return BBB_ESHAPE;
            t.Mtiles = (int)mt;
            t.ipd = (int)(mt * t.Ntiles);
            // an intermediate slab must be a whole number of 128-byte lines (see the header comment)
            if (((int64_t)d.cout * ho * wo * d.batch * 4) % 128 != 0) return BBB_ESHAPE;
        } else if (h.kind == BBB_CHAIN_MAXPOOL) {
            const int k = d.kh, sp = d.stride_h;
            if (k <= 0 || sp <= 0 || d.h < k || d.w < k) return BBB_ESHAPE;
            const int ho = (d.h - k) / sp + 1, wo = (d.w - k) / sp + 1;
            if ((int64_t)d.cin * d.h * d.w * d.batch * 4 > 0x7FFFFFF0LL) return BBB_ESHAPE;
            t.kind = 1; t.kh = k; t.sh = sp; t.Ho = ho; t.Wo = wo;
            const int64_t total4 = (int64_t)d.cin * ho * wo * (d.batch / 4);
            t.ipd = (int)((total4 + kPoolItem - 1) / kPoolItem);
            if (((int64_t)d.cin * ho * wo * d.batch * 4) % 128 != 0) return BBB_ESHAPE;
        } else {
            return BBB_EINVAL;
        }
        if (h.dep >= 0) {
            // the dep stage's output slab must be exactly this stage's input slab
            const StageDev& p = a.st[h.dep];
            const int64_t out = (int64_t)(p.kind == 0 ? p.Cout : p.Cin) * p.Ho * p.Wo * p.B;
            if (out != (int64_t)t.Cin * t.H * t.W * t.B || h.x != p.y) return BBB_EINVAL;
            if (t.kind == 0 && t.x_ds != out) return BBB_EINVAL;
            t.dep_ipd = p.ipd;
        }
        if ((int64_t)t.ipd * slabs > max_items) max_items = (int64_t)t.ipd * slabs;
    }
    a.nst = nstages;
    a.slabs = slabs;
    a.shallow_first = (flags & BBB_CHAIN_SHALLOW_FIRST) ? 1 : 0;
    a.no_cu_balance = (flags & 2u) ? 1 : 0;             // experiments: every resident workgroup claims whenever it can
    a.ws = workspace;
#ifdef PCHAIN_PROFILE
    a.prof_off = (kWsWords + 15) / 16 * 16;              // 64-byte aligned; the caller over-allocates
    if (workspace_ints < a.prof_off + 2048 * 16 + 1024 * 32 * 4) return BBB_EINVAL;
#endif
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return (int)hipGetLastError();
    hipStream_t st = (hipStream_t)stream;
    hipError_t rc = hipMemsetAsync(workspace, 0, (size_t)kWsWords * 4, st);
    if (rc != hipSuccess) return (int)rc;
    // launches whose largest stage is at most ~1.5 rounds of workgroups are latency-bound: staging loads interleaved with
    // the MFMAs (pconv_item's ILV form), exactly as the per-layer launcher chooses
    const bool ilv = max_items <= 1536;
    const int cap = (int)((flags >> 8) & 15u);           // experiments: fewer resident workgroups per CU
    if (bm == 128) return ilv ? launch_chain<128, true>(a, device, cap, st) : launch_chain<128, false>(a, device, cap, st);
    return ilv ? launch_chain<64, true>(a, device, cap, st) : launch_chain<64, false>(a, device, cap, st);
}
