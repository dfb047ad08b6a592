// Packed-ternary x f16/bf16 GEMM on the gfx950 matrix cores.
//
//   y[m,n] = act(clip((sum_k x[m,k] * (code(pw[n,k]) - 1)) * scale[n] + bias[n]))
//
// Design (MI355X):
//  * the activation tile goes global -> LDS directly (global_load_lds, 16 B per lane), three
//    tiles in flight, one barrier per 64-deep k step; the LDS image is XOR-swizzled through the SOURCE
//    address (LDS-DMA writes lane-linear) so the 32-row fragment reads (ds_read_b128) are
//    conflict-free;
//  * the packed weights never touch LDS: each lane loads the 8 bytes (32 weights) of its own
//    output channel for the step straight into registers and decodes them there, 2 bits ->
//    f16 {-1,0,+1,+2}, with the mantissa trick  (bits | 0x6400) = 1024 + c * 4^j  followed by one
//    packed fma  v * 4^-j - (1024 * 4^-j + 1)  -- exact, ~1.25 VALU ops per weight, reused over
//    the wave's 4 (or 2) row tiles;
//  * the k order inside a 64-step is permuted identically for both operands (lane half h takes
//    k = 32h .. 32h+31) so that each lane's weight bytes are contiguous;
//  * v_mfma_f32_32x32x16_{f16,bf16} with the weights as the "A" operand (accumulator rows = output
//    channels), float32 accumulators, fused scale / bias / clip / GELU epilogue straight from the
//    accumulator registers with 16-byte-per-lane stores (v_permlane32_swap between half-waves).
// Requires k % 64 == 0, 16-byte aligned activation rows and 8-byte aligned packed rows; any m, n
// (edge tiles clamp their loads and mask their stores).
#include "gemm_mfma_kernels.h"

#include <stdlib.h>

namespace tvl {

// Sum the split-K slabs in a fixed order (deterministic: slab 0, 1, 2, ...) and apply the fused epilogue.
// VEC = 4: one thread per four consecutive outputs of a row (16-byte slab loads, n % 4 == 0); the loads of up to 8 splits are
// issued together (they are independent; a one-load-at-a-time loop would pay the memory latency `splits` times).
template <typename TY, bool HAS_BIAS, int ACT, int VEC>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t m, int64_t n,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            float clip, int act_begin, TY* __restrict__ y, int64_t ld_y,
                                                            const TY* __restrict__ add1, int64_t ld1, const TY* __restrict__ add2, int64_t ld2) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const float c = fabsf(clip);
    const int64_t per_row = n / VEC, total = m * per_row, stride = m * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row, col = (i - r * per_row) * VEC;
        const float* p = slab + r * n + col;
        vec_t acc = *reinterpret_cast<const vec_t*>(p);   // slab 0 first: the sum is s0 + s1 + s2 + ...
        int s = 1;
        for (; s + 8 <= splits; s += 8) {
            vec_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const vec_t*>(p + (int64_t)(s + u) * stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        if (s + 4 <= splits) {
            vec_t v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const vec_t*>(p + (int64_t)(s + u) * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
            s += 4;
        }
        for (; s < splits; ++s) acc += *reinterpret_cast<const vec_t*>(p + (int64_t)s * stride);
        TY out[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float a = acc[e];
            out[e] = from_float<TY>(epilogue_ch<HAS_BIAS, ACT>(a, scale[col + e], HAS_BIAS ? bias[col + e] : 0.0f, c, (int)(col + e), act_begin));
        }
        // optional addends (tvl_ternary_matmul_residual): (y + add1) + add2, every sum rounded to y's type like the unfused sequence
        if (add1 != nullptr) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) out[e] = from_float<TY>(to_float(out[e]) + to_float(add1[r * ld1 + col + e]));
            if (add2 != nullptr) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) out[e] = from_float<TY>(to_float(out[e]) + to_float(add2[r * ld2 + col + e]));
            }
        }
        TY* dst = y + r * ld_y + col;
        if constexpr (VEC == 4) {
            if constexpr (sizeof(TY) == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(out);
            else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(out);
        } else {
            dst[0] = out[0];
        }
    }
}

template <typename TY>
static int launch_reduce(const float* slab, int splits, int64_t m, int64_t n, const float* scale, const float* bias, float clip,
                         int act, int act_begin, TY* y, int64_t ld_y, hipStream_t st) {
    // four outputs per thread when rows of the slabs and of y can be moved 16 / 8 bytes at a time
    const bool vec = n % 4 == 0 && aligned(slab, 16) && aligned(y, 4 * sizeof(TY)) && (ld_y * sizeof(TY)) % (4 * sizeof(TY)) == 0;
    int64_t g = ceil_div(vec ? m * (n / 4) : m * n, 256);
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    const dim3 grid((unsigned)g), block(256);
    const TY *add1 = nullptr, *add2 = nullptr;
    int64_t ld1 = 0, ld2 = 0;
    if (tls_residual != nullptr && !tls_residual->consumed && tls_residual->add1 != nullptr && tls_residual->y == (const void*)y &&
        tls_residual->m == m) {   // the call's addends ride in this epilogue
        add1 = (const TY*)tls_residual->add1; ld1 = tls_residual->ld1;
        add2 = (const TY*)tls_residual->add2; ld2 = tls_residual->ld2;
        tls_residual->consumed = true;
    }
#define TVL_RD(HB, ACT)                                                                                                                    \
    do {                                                                                                                                   \
        if (vec) hipLaunchKernelGGL((splitk_reduce_kernel<TY, HB, ACT, 4>), grid, block, 0, st, slab, splits, m, n, scale, bias, clip, act_begin, y, ld_y, add1, ld1, add2, ld2); \
        else hipLaunchKernelGGL((splitk_reduce_kernel<TY, HB, ACT, 1>), grid, block, 0, st, slab, splits, m, n, scale, bias, clip, act_begin, y, ld_y, add1, ld1, add2, ld2);     \
    } while (0)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_RD(true, TVL_ACT_GELU_TANH); else TVL_RD(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_RD(false, TVL_ACT_GELU_TANH); else TVL_RD(false, TVL_ACT_NONE);
    }
#undef TVL_RD
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_splitk_reduce(const float* slab, int splits, int64_t m, int64_t n, const float* scale, const float* bias, float clip,
                         int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_reduce<float>(slab, splits, m, n, scale, bias, clip, act, act_begin, (float*)y, ld_y, st);
        case TVL_F16: return launch_reduce<f16>(slab, splits, m, n, scale, bias, clip, act, act_begin, (f16*)y, ld_y, st);
        case TVL_BF16: return launch_reduce<bf16_t>(slab, splits, m, n, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, st);
        default: set_error("splitk_reduce: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

template <typename TX, typename TY, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                      const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, int splits, float* slab,
                      int w_steps, hipStream_t st) {
    const int tiles_m = (int)ceil_div(m, BM), tiles_n = (int)ceil_div(n, BN);
    const int nblk = tiles_m * tiles_n;
    const size_t lds_bytes = (size_t)NBUF * BM * ROW_BYTES;
    const dim3 block(64 * WAVES_M * WAVES_N);
    if (splits > 1) {
        const int total_steps = (int)(k / BK);
        const int steps_per_split = (int)ceil_div(total_steps, splits);
        splits = (int)ceil_div(total_steps, steps_per_split);  // no empty split
        auto kern = gemm_mfma_kernel<TX, float, BM, BN, WAVES_M, WAVES_N, false, TVL_ACT_NONE, true>;
        if (int rc = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc != TVL_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(nblk, splits), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip,
                           0, slab, n, tiles_n, nblk, steps_per_split, w_steps);
        TVL_LAUNCH_CHECK();
        return launch_reduce<TY>(slab, splits, m, n, scale, bias, clip, act, act_begin, y, ld_y, st);
    }
    const dim3 grid(nblk);
#define TVL_GM(HB, ACT)                                                                                            \
    do {                                                                                                           \
        if constexpr (BM == 256) {   /* the big tile runs on the 16x16x32 instruction */                            \
            /* 1 x 8 waves; the 2 x 4 layout (WAVES_M = 2: half the LDS reads, twice the weight decode) measured 21 % slower */ \
            const int cus = device_cu_count();                                                                      \
            const bool persist = nblk > cus;   /* more tiles than CUs: one workgroup per CU walks several (+2..7 %) */ \
            /* the persistent form runs its steps in eight phases of four row tiles (PARTS = 4: 32 fewer fragment registers), which */ \
            /* leaves room for the fourth weight register set: tiles two steps ahead there too (ViT-Base fc1 137 -> 130 us, qkv 102 -> 99) */ \
            auto k16 = persist ? gemm_mfma16_kernel<TX, TY, HB, ACT, 1, true, BIG_NB, 256, 8, false, 2, 32, 4>      \
                               : gemm_mfma16_kernel<TX, TY, HB, ACT, 1, false, BIG_NB>;                             \
            const size_t lds16 = (size_t)BIG_NB * BM * ROW_BYTES + (persist ? persist_param_lds_bytes(8) : 0);      \
            if (int rc = set_dynamic_lds_limit((const void*)k16, lds16); rc != TVL_OK) return rc;                   \
            hipLaunchKernelGGL(k16, dim3(persist ? cus : nblk), block, lds16, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, \
                               bias, clip, act_begin, y, ld_y, tiles_n, nblk, w_steps, 0);                         \
            break;                                                                                                 \
        }                                                                                                          \
        auto kern = gemm_mfma_kernel<TX, TY, BM, BN, WAVES_M, WAVES_N, HB, ACT, false>;                             \
        if (int rc = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc != TVL_OK) return rc;                 \
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act_begin, \
                           y, ld_y, tiles_n, nblk, 0, w_steps);                                                    \
    } while (0)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_GM(true, TVL_ACT_GELU_TANH); else TVL_GM(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_GM(false, TVL_ACT_GELU_TANH); else TVL_GM(false, TVL_ACT_NONE);
    }
#undef TVL_GM
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

// Tile shape and split-K factor, chosen by a small cost model (microseconds, calibrated on MI355X with forced
// configurations).  With q = workgroups per CU, a CU spends  fixed * ceil(q / occupancy)  on prologues/epilogues
// and  steps_per_workgroup * (q == 1 ? step_alone : q * step_shared)  in the k loops (step_shared = time per
// workgroup-step when the CU is shared; small tiles gain little from sharing, they are bound by the decode work).
// Split-K adds one more launch plus writing and re-reading the float32 slabs (~8 TB/s: they mostly stay in the
// Infinity Cache): it fills the chip when
// the output has few tiles, and the slab term stops it where the partial sums would cost more than idle CUs.
// kind: KIND_BIG = the 256 x 256 tile, KIND_MID = 128-row tiles of the same 16x16x32 kernel (gemm_mid.hip), KIND_SMALL = the
// 32x32x16 kernel's 64- and 32-row tiles (few rows per tile: a decoded weight feeds only one or two row tiles there)
enum { KIND_SMALL = 0, KIND_MID = 1, KIND_BIG = 2 };
struct GemmPlan { int kind, bm, bn, splits; double us; };   // us: modelled time (0 for a forced plan)
// step_light: time per k step of a workgroup alone on its CU while at most ~60 % of the CUs are busy; step_full: the same
// with every CU busy (the shared L2 / fabric path slows the tile loads down); step_shared: per workgroup-step when two or more
// share a CU.  Measured with TVL_FORCE_PLAN over M = 256 .. 4096 at N = K = 4096 and the moondream prefill shapes.
struct TileCost { int kind, bm, bn, occupancy; double fixed_us, step_light_us, step_full_us, step_shared_us; };
static const TileCost TILE_COSTS[] = {
    {KIND_BIG, 256, 256, 1, 14.0, 1.20, 1.40, 1.57},    // 8 waves, 96 / 128 KiB LDS (shared = a step of the persistent form: three LDS tiles)
    {KIND_MID, 128, 256, 1, 7.0, 0.62, 0.84, 0.84},     // 8 waves, 64 KiB LDS
    {KIND_MID, 128, 128, 2, 6.0, 0.42, 0.50, 0.49},     // 4 waves, 64 KiB LDS
    {KIND_MID, 64, 128, 4, 5.0, 0.28, 0.36, 0.30},      // 4 waves, 32 KiB LDS
    {KIND_SMALL, 32, 128, 3, 3.5, 0.32, 0.32, 0.21},
};

// TVL_FORCE_PLAN="kind:bm:bn:splits" pins the choice (tuning and tests: every tile can be timed on every shape); read once.
static bool forced_plan(GemmPlan* p) {
    static const GemmPlan forced = [] {
        GemmPlan f{-1, 0, 0, 0, 0.0};
        if (const char* e = getenv("TVL_FORCE_PLAN")) {
            if (sscanf(e, "%d:%d:%d:%d", &f.kind, &f.bm, &f.bn, &f.splits) != 4) f.kind = -1;
        }
        return f;
    }();
    if (forced.kind < 0) return false;
    *p = forced;
    return true;
}

static GemmPlan plan_gemm(int64_t m, int64_t n, int64_t k, size_t workspace_bytes) {
    const int64_t steps = k / BK;
    const int64_t max_by_ws = (int64_t)(workspace_bytes / ((size_t)m * n * sizeof(float)));
    GemmPlan best{KIND_MID, 128, 128, 1, 0.0};
    if (forced_plan(&best)) {
        if (best.splits < 0) return best;   // the two-workgroups-per-CU form of the 128 x 256 tile
        if (best.splits > max_by_ws) best.splits = max_by_ws > 1 ? (int)max_by_ws : 1;
        if (best.splits > steps) best.splits = (int)steps;
        return best;
    }
    const int cus = device_cu_count();
    double best_t = 1e30;
    for (const TileCost& tc : TILE_COSTS) {
        if (tc.bm > 32 && m <= tc.bm / 2) continue;   // more than half of the tile's rows would be padding
        // the 32-row tile (two row tiles per decoded weight): where taller tiles would be mostly padding, and up to 160 rows on layers so small
        // that 64-row tiles with 8 k ranges still leave compute units idle (128 x 1152 x 1152: 9.7 -> 8.0 us)
        if (tc.bm == 32 && m > 48 && !(m <= 160 && ceil_div(m, 64) * ceil_div(n, 128) * 8 < cus)) continue;
        const int64_t tiles = ceil_div(m, tc.bm) * ceil_div(n, tc.bn);
        for (int64_t s = 1; s <= 64; ++s) {
            if (s > 1 && (s > max_by_ws || steps / s < 2 || tc.kind == KIND_BIG)) break;
            const int64_t per = ceil_div(steps, s);
            const int64_t q = ceil_div(tiles * s, cus);
            const double load = (double)(tiles * s) / (double)cus;
            // one workgroup per CU: a step costs between its time on a lightly loaded chip (half of the CUs busy) and on a full one
            const double fill = load <= 0.5 ? 0.0 : (load >= 1.0 ? 1.0 : (load - 0.5) / 0.5);
            const double step1 = tc.step_light_us + (tc.step_full_us - tc.step_light_us) * fill;
            // the big tile's fixed part is mostly its 128 KiB-per-tile store burst: it grows with the number of tiles that end together
            const double fixed1 = tc.kind == KIND_BIG ? tc.fixed_us - 4.0 * (1.0 - (load < 1.0 ? load : 1.0)) : tc.fixed_us;
            double t = q == 1 ? fixed1 + (double)per * step1
                              : (double)ceil_div(q, tc.occupancy) * tc.fixed_us + (double)per * (double)q * tc.step_shared_us;
            // the 256 x 256 tile walks its q tiles per CU as one stream of k steps: the fixed cost once, ~4 us of epilogue per tile
            if (tc.kind == KIND_BIG && q > 1) t = tc.fixed_us + (double)q * ((double)per * tc.step_shared_us + 4.0);
            // so does the 128 x 256 tile (gemm_mid.hip): whole rounds of k loops, the last one maybe on a lightly loaded chip, and ~6 us
            // per tile for its epilogue under the next tile's first steps
            if (tc.kind == KIND_MID && tc.bn == 256 && s == 1 && tiles > cus) {
                const int64_t rounds = ceil_div(tiles, cus);
                const double last = load - (double)(rounds - 1);
                t = tc.fixed_us + (double)(rounds - 1) * (double)per * tc.step_full_us +
                    (double)per * (last <= 0.6 ? tc.step_light_us : tc.step_full_us) + load * 6.0;
            }
            if (s > 1) t += 3.0 + 3.0 * (double)s * (double)m * (double)n * 4.0 / 8.0e6;
            // 64-row tiles beyond 192 rows decode every weight for four row tiles only: they win where taller tiles leave the chip
            // half empty or need a k split (512 x 2048 x 2048: 18.7 -> 14.5 us); a small handicap settles the near-ties the other way
            if (tc.bm == 64 && m > 192) t *= 1.03;
            if (t < best_t) { best_t = t; best = GemmPlan{tc.kind, tc.bm, tc.bn, (int)s, t}; }
        }
    }
    return best;
}

// A persistent grid whose last round is mostly empty (tiles = 3.48 x the compute units runs as long as 4.0 x): the rows that fill
// whole rounds of 256 x 256 (or 128 x 256) tiles go to that tile, the remaining rows are a product of their own on smaller tiles
// (a second launch; rows are independent, so the results are the same bits).  Returns the rows of the first part and its tile
// height in *head_bm; 0 = no split.  TVL_TAIL_SPLIT=0 turns it off (tuning).
static int64_t tail_split_rows(const GemmPlan& p, int64_t m, int64_t n, int64_t k, size_t workspace_bytes, int* head_bm) {
    static const bool enabled = [] { const char* e = getenv("TVL_TAIL_SPLIT"); return !e || atoi(e) != 0; }();
    if (!enabled || p.us <= 0.0) return 0;
    const int cus = device_cu_count();
    const double per = (double)(k / BK);
    const int64_t tiles_n = ceil_div(n, 256);
    int64_t best_m1 = 0;
    double best_t = 0.97 * p.us;   // a second launch has to buy at least 3 %
    for (const int bm : {256, 128}) {
        const int64_t tiles = ceil_div(m, bm) * tiles_n, rounds = tiles / cus;
        if (rounds < 1 || tiles % cus == 0) continue;
        const int64_t m1 = (rounds * cus / tiles_n) * bm;   // rows of whole tiles that fit into `rounds` rounds
        if (m1 <= 0 || m1 >= m) continue;
        // the cost table's rows of these two tiles: one k loop per round, the epilogue under the next tile's first steps
        const TileCost& tc = TILE_COSTS[bm == 256 ? 0 : 1];
        const double t1 = bm == 256 ? (rounds == 1 ? tc.fixed_us + per * tc.step_full_us : tc.fixed_us + (double)rounds * (per * tc.step_shared_us + 4.0))
                                    : (rounds == 1 ? tc.fixed_us + per * tc.step_full_us : tc.fixed_us + (double)rounds * (per * 0.9 + 6.0));
        const double t = t1 + plan_gemm(m - m1, n, k, workspace_bytes).us + 2.0;
        if (t < best_t) { best_t = t; best_m1 = m1; *head_bm = bm; }
    }
    return best_m1;
}

template <typename TX, typename TY>
static int launch_mfma_xy(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                          const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, void* workspace,
                          size_t workspace_bytes, int w_steps, hipStream_t st) {
    const bool ws_ok = workspace != nullptr && aligned(workspace, 16);
    const GemmPlan p = plan_gemm(m, n, k, ws_ok ? workspace_bytes : 0);
    float* slab = (float*)workspace;
    constexpr int XD = std::is_same<TX, f16>::value ? TVL_F16 : TVL_BF16;
    constexpr int YD = std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16);
    int head_bm = 0;
    if (const int64_t m1 = tail_split_rows(p, m, n, k, ws_ok ? workspace_bytes : 0, &head_bm)) {
        int rc = head_bm == 256 ? launch_cfg<TX, TY, 256, 256, 1, 8>(x, m1, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, 1, slab, w_steps, st)
                                : launch_gemm_mid(x, XD, m1, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, YD, ld_y, 128, 256, 1, slab, w_steps, st);
        if (rc != TVL_OK) return rc;
        const GemmPlan p2 = plan_gemm(m - m1, n, k, ws_ok ? workspace_bytes : 0);
        const TX* x2 = x + m1 * ld_x;
        TY* y2 = y + m1 * ld_y;
        if (p2.kind == KIND_BIG)
            return launch_cfg<TX, TY, 256, 256, 1, 8>(x2, m - m1, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y2, ld_y, 1, slab, w_steps, st);
        if (p2.kind == KIND_MID)
            return launch_gemm_mid(x2, std::is_same<TX, f16>::value ? TVL_F16 : TVL_BF16, m - m1, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin,
                                   y2, std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16), ld_y, p2.bm, p2.bn,
                                   p2.splits, slab, w_steps, st);
        return launch_cfg<TX, TY, 32, 128, 1, 4>(x2, m - m1, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y2, ld_y, p2.splits, slab, w_steps, st);
    }
    if (p.kind == KIND_BIG)
        return launch_cfg<TX, TY, 256, 256, 1, 8>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, 1, slab, w_steps, st);
    if (p.kind == KIND_MID)
        return launch_gemm_mid(x, std::is_same<TX, f16>::value ? TVL_F16 : TVL_BF16, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y,
                               std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16), ld_y, p.bm, p.bn,
                               p.splits, slab, w_steps, st);
    if (p.bm == 128)   // (only by TVL_FORCE_PLAN: the 32x32x16 kernel's 128 x 128 tile, superseded by the mid-size kernel)
        return launch_cfg<TX, TY, 128, 128, 2, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, slab, w_steps, st);
    if (p.bm == 64)   // (only by TVL_FORCE_PLAN, like the 128 x 128 tile above)
        return launch_cfg<TX, TY, 64, 128, 2, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, slab, w_steps, st);
    return launch_cfg<TX, TY, 32, 128, 1, 4>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, slab, w_steps, st);
}

template <typename TX>
static int launch_mfma_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                         const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                         void* ws, size_t ws_bytes, int w_steps, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_mfma_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, ws, ws_bytes, w_steps, st);
        case TVL_F16: return launch_mfma_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, ws, ws_bytes, w_steps, st);
        case TVL_BF16: return launch_mfma_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, ws, ws_bytes, w_steps, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

// ---- float32 activations -------------------------------------------------------------------------------------
// x = hi + mid + lo with three bf16 numbers, exactly (8 + 8 + 8 significand bits): the matrix cores then compute
// sum_k w_k hi_k + sum_k w_k mid_k + sum_k w_k lo_k with exact products and float32 accumulation - a float32 dot
// product in a different summation order.  The three planes are written side by side along k ([m][3k] bf16) into the
// workspace, so the GEMM is the bf16 kernel over a k range of 3k whose weight rows wrap around every k.
// Output layout: plane p of row r starts at xs + p * plane_off + r * ld_out (side by side along k: ld_out = 3k, plane_off = k; the
// planes as three row blocks of an [3m][k] matrix: ld_out = k, plane_off = m k).
__global__ void __launch_bounds__(256) split_f32_kernel(const float* __restrict__ x, int64_t m, int64_t k, int64_t ld_x,
                                                        uint16_t* __restrict__ xs, int64_t ld_out, int64_t plane_off) {
    const int64_t per_row = k / 8, total = m * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row, c8 = u - row * per_row;
        const float4* src = reinterpret_cast<const float4*>(x + row * ld_x + 8 * c8);
        const float4 a = src[0], b = src[1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint16_t hi[8], mid[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t bits = __float_as_uint(v[i]);
            const uint32_t hb = bits & 0xFFFF0000u;
            const bool finite = (bits & 0x7F800000u) != 0x7F800000u;   // inf / nan: all of it in the first plane
            const float r1 = finite ? v[i] - __uint_as_float(hb) : 0.0f;            // exact
            const uint32_t mb = __float_as_uint(r1) & 0xFFFF0000u;
            const float r2 = r1 - __uint_as_float(mb);                                // exact, <= 8 significant bits
            hi[i] = (uint16_t)((finite ? hb : bits | ((bits & 0xFFFFu) ? 0x10000u : 0u)) >> 16);   // keep a nan a nan
            mid[i] = (uint16_t)(mb >> 16);
            lo[i] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
        uint16_t* dst = xs + row * ld_out + 8 * c8;
        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(hi);
        *reinterpret_cast<uint4*>(dst + plane_off) = *reinterpret_cast<const uint4*>(mid);
        *reinterpret_cast<uint4*>(dst + 2 * plane_off) = *reinterpret_cast<const uint4*>(lo);
    }
}

// the three planes as row blocks of a bf16 [3m][k] matrix at xs (m * 3 * k * 2 bytes), for the few-row kernels: rows r, m + r, 2m + r
// of the product are then added
int launch_split_f32_rows(const float* x, int64_t m, int64_t k, int64_t ld_x, uint16_t* xs, hipStream_t st) {
    int64_t g = ceil_div(m * (k / 8), 256);
    g = g > 256 * 16 ? 256 * 16 : g;
    hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)g), dim3(256), 0, st, x, m, k, ld_x, xs, k, m * k);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

static inline size_t split_bytes(int64_t m, int64_t k) { return ((size_t)m * 3 * k * 2 + 255) & ~(size_t)255; }

// float32 slabs of the k split the plan for m x n x k would use with unlimited workspace
static size_t slab_bytes(int64_t m, int64_t n, int64_t k) {
    const GemmPlan p = plan_gemm(m, n, k, (size_t)1 << 40);
    int head_bm = 0;
    if (const int64_t m1 = tail_split_rows(p, m, n, k, (size_t)1 << 40, &head_bm)) {   // the remainder's k split, if it has one
        const GemmPlan p2 = plan_gemm(m - m1, n, k, (size_t)1 << 40);
        return p2.splits > 1 ? (size_t)p2.splits * (m - m1) * n * sizeof(float) : 0;
    }
    return p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
}

static int describe_plan(const GemmPlan& p, int64_t m, int64_t n, char* out, size_t len) {
    const char* kind = p.kind == KIND_BIG ? "gemm_mfma16_kernel 256x256" : p.kind == KIND_MID ? "gemm_mfma16_kernel" : "gemm_mfma_kernel";
    const int cus = device_cu_count();
    const bool persist = (p.kind == KIND_BIG && ceil_div(m, 256) * ceil_div(n, 256) > cus) ||
                         (p.kind == KIND_MID && p.bn == 256 && p.splits == 1 && ceil_div(m, 128) * ceil_div(n, 256) > cus);
    if (p.kind == KIND_BIG) return snprintf(out, len, "%s%s", kind, persist ? " persistent" : "");
    if (p.splits > 1) return snprintf(out, len, "%s %dx%d split-k %d + splitk_reduce_kernel", kind, p.bm, p.bn, p.splits);
    return snprintf(out, len, "%s %dx%d%s", kind, p.bm, p.bn, persist ? " persistent" : "");
}

void gemm_mfma_describe(int x_dtype, int64_t m, int64_t n, int64_t k, size_t workspace_bytes, char* out, size_t out_len) {
    if (out_len == 0) return;
    size_t ws = workspace_bytes;
    int off = 0;
    if (x_dtype == TVL_F32) {   // three bf16 planes side by side along k
        off = snprintf(out, out_len, "split_f32_kernel + ");
        ws = ws > split_bytes(m, k) ? ws - split_bytes(m, k) : 0;
        k *= 3;
    }
    if (off < 0 || (size_t)off >= out_len) return;
    const GemmPlan p = plan_gemm(m, n, k, ws);
    int head_bm = 0;
    if (const int64_t m1 = tail_split_rows(p, m, n, k, ws, &head_bm)) {
        int w = snprintf(out + off, out_len - off, "rows 0..%lld: gemm_mfma16_kernel %dx256 persistent; rest: ", (long long)m1, head_bm);
        if (w < 0 || (size_t)(off + w) >= out_len) return;
        off += w;
        describe_plan(plan_gemm(m - m1, n, k, ws), m - m1, n, out + off, out_len - off);
        return;
    }
    describe_plan(p, m, n, out + off, out_len - off);
}

size_t gemm_mfma_workspace_bytes(int x_dtype, int64_t m, int64_t n, int64_t k) {
    if (x_dtype == TVL_F32) return split_bytes(m, k) + slab_bytes(m, n, 3 * k);
    return slab_bytes(m, n, k);
}

bool gemm_mfma_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                         int64_t ld_pw, const void* workspace, size_t workspace_bytes) {
    if (k % BK != 0 || k <= 0) return false;
    if (!aligned(pw, 8) || ld_pw % 8 != 0) return false;
    if (m > INT32_MAX || n > INT32_MAX || 3 * k > INT32_MAX) return false;
    if (ceil_div(m, 32) * ceil_div(n, 128) > INT32_MAX) return false;
    if (ld_x > (1 << 22) || ld_pw > (1 << 22)) return false;   // lane offsets inside a 256-row tile are 32-bit byte offsets
    if (x_dtype == TVL_F32) {   // needs room for the three bf16 planes
        if (!aligned(x, 16) || (ld_x * 4) % 16 != 0) return false;
        return workspace != nullptr && aligned(workspace, 256) && workspace_bytes >= split_bytes(m, k);
    }
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16) return false;
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    return true;
}

int launch_gemm_mfma(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                     int64_t ld_pw, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype,
                     int64_t ld_y, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const int w_steps = (int)(k / BK);
    if (x_dtype == TVL_F32) {
        uint16_t* xs = (uint16_t*)workspace;
        int64_t g = ceil_div(m * (k / 8), 256);
        g = g > 256 * 16 ? 256 * 16 : g;
        hipLaunchKernelGGL(split_f32_kernel, dim3((unsigned)g), dim3(256), 0, st, (const float*)x, m, k, ld_x, xs, 3 * k, k);
        TVL_LAUNCH_CHECK();
        const size_t used = split_bytes(m, k);
        return launch_mfma_x((const bf16_t*)xs, m, 3 * k, 3 * k, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y,
                             (char*)workspace + used, workspace_bytes - used, w_steps, st);
    }
    if (x_dtype == TVL_F16)
        return launch_mfma_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, w_steps, st);
    return launch_mfma_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, w_steps, st);
}

}  // namespace tvl
