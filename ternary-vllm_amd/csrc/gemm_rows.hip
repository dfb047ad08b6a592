// Packed-ternary x f16 / bf16 product for few activation rows (decode batches of 3 .. 64 rows): a one-launch kernel for small
// layers (first half of this file) and a kernel that shares the activations through LDS for larger ones (second half).
//
// The work is a weight stream: at m = 16, n = k = 4096 the packed weights are 4 MiB against 128 KiB of activations.
// A workgroup owns 16 * CT output channels and ALL rows; its waves split the k range (each wave a contiguous
// share of the 256-deep steps) and add their partial sums through LDS at the end in wave order - no slab in
// memory, no second launch, results independent of timing.
//   * weights go straight to registers, 16 bytes (64 weights) per lane and step: lane (c, g) holds weights
//     k = 64 g .. 64 g + 63 of channel c, four steps in flight per wave (plain loads: the compiler counts the waits);
//     they are streamed once, so the loads are non-temporal;
//   * v_mfma_f32_16x16x32 with the decoded weights as operand A and the activations as operand B: MFMA j of a step
//     contracts, in lane group g, the k values 64 g + 8 j .. + 7 - the same permutation of k on both operands, so
//     the activation fragment of lane (r, g) is the 16 bytes at x[r][step + 64 g + 8 j], read from L2 / the
//     Infinity Cache directly (every workgroup reads all of x; there is no reuse inside a workgroup to stage for);
//   * decode: the 2-bit -> f16 / bf16 {-1, 0, +1, +2} mantissa trick of the big kernels (exact, 1.25 VALU per weight),
//     so non-finite activations behave as in a dense product (0 * inf = NaN, +-1 * inf = +-inf);
//   * D register e of lane (c, g) is channel 4 g + e, row c: the epilogue stores 4 consecutive channels per lane.
// Requires k % 256 == 0, 16-byte aligned activation rows and packed rows.
#include "gemm_mfma_kernels.h"

#include <stdlib.h>
#include <string.h>

namespace tvl {


// One lane's share of a 16 x 16 result fragment: row gm, channels cn .. cn + 3 (epilogue, conversion, 8- or 16-byte store).
__device__ __forceinline__ void rows_store_fragment(floatx4 s, int gm, int cn, int m, int n, const float* __restrict__ scale,
                                                    const float* __restrict__ bias, bool has_bias, int act, float cl, int act_begin,
                                                    void* __restrict__ y, int y_dtype, int64_t ld_y) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        o[e] = epilogue_rt(s[e], cn + e < n ? scale[cn + e] : 0.0f, (has_bias && cn + e < n) ? bias[cn + e] : 0.0f, has_bias, act, cl, cn + e, act_begin);
    if (gm >= m) return;
    if (y_dtype == TVL_F32) {
        float* yrow = reinterpret_cast<float*>(y) + (int64_t)gm * ld_y;
        if (cn + 4 <= n && ((reinterpret_cast<uintptr_t>(yrow + cn)) & 15) == 0) *reinterpret_cast<float4*>(yrow + cn) = make_float4(o[0], o[1], o[2], o[3]);
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (cn + e < n) yrow[cn + e] = o[e];
    } else {
        uint16_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            h[e] = y_dtype == TVL_F16 ? __builtin_bit_cast(uint16_t, from_float<f16>(o[e])) : from_float<bf16_t>(o[e]).v;
        uint16_t* yrow = reinterpret_cast<uint16_t*>(y) + (int64_t)gm * ld_y;
        if (cn + 4 <= n && ((reinterpret_cast<uintptr_t>(yrow + cn)) & 7) == 0) *reinterpret_cast<uint2*>(yrow + cn) = *reinterpret_cast<const uint2*>(h);
        else
#pragma unroll
            for (int e = 0; e < 4; ++e) if (cn + e < n) yrow[cn + e] = h[e];
    }
}

template <typename TX, int MR, int CT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES)
gemm_rows_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                 const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act, int act_begin, void* __restrict__ y,
                 int y_dtype, int64_t ld_y) {
    constexpr int DEPTH = 4;                          // steps of weights in flight per wave
    __shared__ floatx4 red[WAVES][MR * CT][64];       // the waves' partial sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * CT);
    const int total = k / 256;
    const int base = total / WAVES, rem = total % WAVES;
    const int steps = base + (wave < rem ? 1 : 0);                    // this wave's share of the k steps
    const int s0 = wave * base + (wave < rem ? wave : rem);

    const uint4* wp[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        int gn = n0 + 16 * t + c;
        gn = gn < n ? gn : n - 1;
        wp[t] = reinterpret_cast<const uint4*>(pw + (int64_t)gn * ld_pw + (int64_t)s0 * 64 + 16 * g);
    }
    const uint4* xp[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        int gm = 16 * i + c;
        gm = gm < m ? gm : m - 1;
        xp[i] = reinterpret_cast<const uint4*>(x + (int64_t)gm * ld_x + (int64_t)s0 * 256 + 64 * g);
    }
    floatx4 acc[MR][CT];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const DecodeConsts dc = make_decode_consts();

    auto load_w = [&](int s, uint4 (&w)[CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t] + (int64_t)s * 4));   // 64 bytes per step
            w[t] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    // activations of one half step: MFMAs j = 4 half .. 4 half + 3
    auto load_x = [&](int s, int half, uint4 (&xf)[MR][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) xf[i][jj] = xp[i][(int64_t)s * 32 + 4 * half + jj];   // 512 bytes per row and step
    };
    auto compute_half = [&](const uint4 (&w)[CT], int half, const uint4 (&xf)[MR][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * half + jj;   // dword j >> 1 of the lane's 16 bytes, halfword j & 1
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const uint32_t d = (j >> 1) == 0 ? w[t].x : (j >> 1) == 1 ? w[t].y : (j >> 1) == 2 ? w[t].z : w[t].w;
                const uint4 a = (j & 1) ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
#pragma unroll
                for (int i = 0; i < MR; ++i) acc[i][t] = mfma16<TX>(a, xf[i][jj], acc[i][t]);
            }
        }
    };

    if (steps > 0) {
        uint4 w0[CT], w1[CT], w2[CT], w3[CT];
        uint4 xa[MR][4], xb[MR][4];
        const int last = steps - 1;
        load_w(0, w0);
        load_w(1 < last ? 1 : last, w1);
        load_w(2 < last ? 2 : last, w2);
        load_w(3 < last ? 3 : last, w3);
        load_x(0, 0, xa);
        // one step: its second half's activations are requested before the first half is multiplied, the next step's first
        // half before the second is; the weight registers are refilled for DEPTH steps later (past the end: the last step again,
        // never used)
#define TVL_ROWS_STEP(S, W)                                        \
        do {                                                       \
            load_x((S), 1, xb);                                    \
            compute_half(W, 0, xa);                                \
            load_x((S) + 1 < last ? (S) + 1 : last, 0, xa);        \
            compute_half(W, 1, xb);                                \
            load_w((S) + DEPTH < last ? (S) + DEPTH : last, W);    \
        } while (0)
        for (int s = 0; s < steps; s += DEPTH) {
            TVL_ROWS_STEP(s, w0);
            if (s + 1 < steps) TVL_ROWS_STEP(s + 1, w1);
            if (s + 2 < steps) TVL_ROWS_STEP(s + 2, w2);
            if (s + 3 < steps) TVL_ROWS_STEP(s + 3, w3);
        }
#undef TVL_ROWS_STEP
    }

    // ---- partial sums of the waves, added in wave order; wave w finishes fragments w, w + WAVES, ... --------------------
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) red[wave][i * CT + t][lane] = acc[i][t];
    __syncthreads();
    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
    for (int f = wave; f < MR * CT; f += WAVES) {
        const int i = f / CT, t = f % CT;
        floatx4 s = red[0][f][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w][f][lane];
        rows_store_fragment(s, 16 * i + c, n0 + 16 * t + 4 * g, m, n, scale, bias, has_bias, act, cl, act_begin, y, y_dtype, ld_y);
    }
}

// Every workgroup reads all of x through L2 (16 or 32 channels per workgroup), so the kernel's time grows with m * n * k:
// measured (us, n = k = 4096) 5.1 / 5.7 / 6.5 / 8.5 at m = 2 / 4 / 8 / 16 against 5.4 / 8.2 / 10.2 / 11.4 for the kernels
// that split k over workgroups and reduce in a second launch, but 25 against 19 at 16 x 8192 x 8192, where the
// activation traffic (m k n / 8 bytes) outgrows the weight stream: larger layers go to gemm_rows_shared_kernel below.  It is used
// where it wins: 3 .. 16 rows (one or two rows go to the table-lookup kernel) up to m n k = 3e8, and 17 .. 32 rows on small layers.
static bool rows_use_staged(int64_t m, int64_t n, int64_t k);
static bool rows_use_blocks(int64_t m, int64_t n, int64_t k);
// float32 activations, 3 .. 10 rows: split into three bf16 planes stacked as 3m rows (workspace), one staged launch, the planes'
// rows added in its reduction - where the staged kernel takes 3m rows (x in LDS, one round of workgroups)
static bool rows_f32_planes_ok(int64_t m, int64_t n, int64_t k, const void* workspace, size_t workspace_bytes) {
    if (m < 3 || m > 10 || k <= 0 || k % 256 != 0) return false;
    if (workspace == nullptr || !aligned(workspace, 16) || workspace_bytes < (size_t)m * 3 * k * 2) return false;
    return rows_use_staged(3 * m, n, k);
}

bool gemm_rows_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                         const void* workspace, size_t workspace_bytes) {
    if (!aligned(pw, 16) || ld_pw % 16 != 0) return false;
    if (n > INT32_MAX - 64 || k > INT32_MAX) return false;
    if (x_dtype == TVL_F32) return aligned(x, 16) && (ld_x * 4) % 16 == 0 && rows_f32_planes_ok(m, n, k, workspace, workspace_bytes);
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16) return false;
    if (m < 3 || m > 512 || k <= 0 || k % 256 != 0) return false;
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    if (m > 64) return rows_use_blocks(m, n, k);   // 65 .. 512 rows on small layers: the staged form on blocks of 64 rows
    if (m > 32) return rows_use_staged(m, n, k);   // 33 .. 64 rows: only the staged form
    const double work = (double)m * (double)n * (double)k;
    if (rows_use_staged(m, n, k)) return true;
    return m <= 16 ? work <= 3.0e8 : (double)n * (double)k <= 4.2e6;
}

template <typename TX, int MR, int CT>
static void launch_rows_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                            const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    constexpr int WAVES = 4;   // 8 and 16 waves (finer k split, more loads in flight) measured slower: the L2 path is the limit, not latency
    hipLaunchKernelGGL((gemm_rows_kernel<TX, MR, CT, WAVES>), dim3((unsigned)ceil_div(n, 16 * CT)), dim3(64 * WAVES), 0, st, x, (int)m, (int)k, ld_x, pw,
                       (int)n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y);
}

template <typename TX, int MR>
static void launch_rows_mr(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                           const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    // 16 channels per workgroup keeps the grid at n / 16 workgroups (one per CU at n = 4096); with four or more workgroups per CU,
    // 32 channels per workgroup halve the activation traffic
    if (ceil_div(n, 16) >= 4 * (int64_t)device_cu_count())
        launch_rows_cfg<TX, MR, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    else
        launch_rows_cfg<TX, MR, 1>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
}

template <typename TX>
static int launch_rows_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                         const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    if (m <= 16) launch_rows_mr<TX, 1>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    else launch_rows_mr<TX, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_rows_staged_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                                const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, int planes, hipStream_t st);

int launch_gemm_rows(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                     const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                     void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (y_dtype != TVL_F32 && y_dtype != TVL_F16 && y_dtype != TVL_BF16) {
        set_error("ternary_matmul: unsupported output dtype %d", y_dtype);
        return TVL_ERR_INVALID;
    }
    if (x_dtype == TVL_F32) {
        uint16_t* xs = reinterpret_cast<uint16_t*>(workspace);
        if (int rc = launch_split_f32_rows((const float*)x, m, k, ld_x, xs, st); rc != TVL_OK) return rc;
        return launch_rows_staged_x((const bf16_t*)xs, 3 * m, k, k, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, 3, st);
    }
    if (m > 64 || rows_use_staged(m, n, k)) {   // (more than 64 rows: gemm_rows_supported admitted them for the row-block form)
        if (x_dtype == TVL_F16) return launch_rows_staged_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, 1, st);
        return launch_rows_staged_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, 1, st);
    }
    if (x_dtype == TVL_F16) return launch_rows_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    return launch_rows_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
}

// LDS of one workgroup: the image of one phase (at most 8 * SPW steps; 4 steps with 3 or 4 row blocks, where the waves also split the
// channel tiles), reused for the partial sums
static size_t rows_staged_lds_bytes(int mr, int ct, int64_t k) {
    const int64_t steps = k / 256, spw = (mr == 1 && k > 2048) ? 2 : 1, ps = mr > 2 ? 4 : 8 * spw, phase_steps = steps < ps ? steps : ps;
    const size_t image = (size_t)phase_steps * mr * 8 * 65 * 16, sums = (size_t)8 * mr * (mr > 2 ? ct / 2 : ct) * 64 * 16;
    return image > sums ? image : sums;
}
// 16-channel tiles per workgroup: 1 while one round of workgroups covers n, else 2, 4 or 8 (two phases at 17 .. 32 rows: 1; 33 .. 64 rows: 2 or 4,
// the waves splitting them)
static int rows_staged_tiles(int64_t m, int64_t n, int64_t k) {
    if (m > 32) return ceil_div(n, 32) * ceil_div(m, 64) <= (int64_t)device_cu_count() ? 2 : 4;
    const int mr = m <= 16 ? 1 : 2;
    if (mr == 2 && k > 2048) return 1;
    // (measured beyond these: 16 x 16384 x 8192 with 4 tiles 17.4 -> 17.8 us, 16 x 28672 x 4096 with 8 tiles 14.5 -> 16.6: not used)
    const int max_ct = mr == 1 ? (k > 4096 ? 2 : (k > 2048 ? 4 : 8)) : 4;
    for (int ct = 1; ct < max_ct; ct *= 2) {
        int64_t per_cu = (int64_t)(160 * 1024) / (int64_t)(rows_staged_lds_bytes(mr, ct, k) + 1024);
        per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
        if (ceil_div(n, 16 * ct) <= per_cu * device_cu_count()) return ct;
    }
    return max_ct;
}

// ---- small layers, one launch, the activations staged once -------------------------------------------------------------------
//
// gemm_rows_kernel's fragment loads take 16 bytes per lane out of 64 different cache lines per instruction; at 16 x 4096 x 4096
// those look-ups are about half of its 8.4 us.  When x fits in LDS (16 rows x k <= 4096, 32 rows x k <= 2048; twice that k in two
// phases) the workgroup - 8 waves on 16 channels, k split 8 ways - first brings it in with whole-cache-line loads (one 16-byte
// piece per thread and 256-deep step, all steps of a phase in flight at once; written in the padded fragment layout of
// gemm_rows_shared_kernel), its weight loads - issued AFTER the activation loads, so that waiting for the latter leaves them in
// flight - arriving meanwhile.  Then every wave multiplies its steps out of LDS (the second phase's activations travelling in
// registers meanwhile) and the partial sums are added through LDS in wave order as above.
// TW = 2 (33 .. 64 rows: three or four 16-row blocks): an image of 8 steps would not fit, so a phase is 4 steps and the waves form two
// groups that split the workgroup's channel tiles - wave = (tile group, step of the phase).
template <typename TX, int MR, int CT, int SPW, int PH, int TW = 1>   // CT: 16-channel tiles per workgroup; SPW: steps per wave and phase; PH: phases
__global__ void __launch_bounds__(512)
gemm_rows_staged_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                        const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act, int act_begin, void* __restrict__ y,
                        int y_dtype, int64_t ld_y, int planes) {
    constexpr int WAVES = 8, FS = 65, IMG = 8 * FS;
    constexpr int SW = WAVES / TW, CTW = CT / TW, PS = SW * SPW;   // SW waves along the steps of a phase, CTW tiles per wave, PS steps per phase (k <= 256 * PS * PH)
    static_assert(CT % TW == 0 && (TW == 1 || TW == 2), "the waves split the channel tiles evenly");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* xbuf = reinterpret_cast<uint4*>(lds_raw);                 // [steps of one phase][MR][8 fragments][FS]
    floatx4* red = reinterpret_cast<floatx4*>(lds_raw);              // [WAVES][MR * CTW][64], over the image once it has been read
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sw = wave % SW, tg = wave / SW;                         // this wave's step slot inside a phase and its tile group
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * CT);
    if (gridDim.y > 1) {   // 65 .. 512 rows: blockIdx.y selects a block of 16 MR rows, a product of its own on the same weights
        const int r0 = (int)blockIdx.y * (16 * MR);
        x += (int64_t)r0 * ld_x;
        y = reinterpret_cast<unsigned char*>(y) + (int64_t)r0 * ld_y * (y_dtype == TVL_F32 ? 4 : 2);
        m = m - r0 < 16 * MR ? m - r0 : 16 * MR;
    }
    const int total = k / 256;
    const int nph = (total + PS - 1) / PS;                            // phases this k needs (<= PH)

    // ---- activation loads: thread t = (row t >> 5 of each 16-row block, piece t & 31) of every step of a phase
    const int st_row = threadIdx.x >> 5, st_p = threadIdx.x & 31;
    const uint4* xsrc[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        int gm = 16 * i + st_row;
        gm = gm < m ? gm : m - 1;
        xsrc[i] = reinterpret_cast<const uint4*>(x + (int64_t)gm * ld_x + 8 * st_p);
    }
    nts_u32x4 st[PS][MR];
    auto load_phase = [&](int ph) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int s = 0; s < PS; ++s) {
                int ss = ph * PS + s;
                ss = ss < total ? ss : total - 1;                    // past the end: the last step again (never read back)
                st[s][i] = *reinterpret_cast<const nts_u32x4*>(xsrc[i] + (int64_t)ss * 32);
            }
    };
    const int st_dst = (st_p & 7) * FS + (st_p >> 3) * 16 + st_row;
    // The LDS image holds min(total, PS) steps (rows_staged_lds_bytes).  With a whole phase in LDS (total >= PS) every slot exists and
    // the stores are unconditional (steps past the end of k hold the last step again and are never read); with fewer steps than a
    // phase the slots past the end do not exist and their stores are skipped.
    auto store_phase = [&](int ph) __attribute__((always_inline)) {
        if (total >= PS) {
#pragma unroll
            for (int s = 0; s < PS; ++s)
#pragma unroll
                for (int i = 0; i < MR; ++i) *reinterpret_cast<nts_u32x4*>(xbuf + (s * MR + i) * IMG + st_dst) = st[s][i];
        } else {
#pragma unroll
            for (int s = 0; s < PS; ++s)
                if (ph * PS + s < total) {
#pragma unroll
                    for (int i = 0; i < MR; ++i) *reinterpret_cast<nts_u32x4*>(xbuf + (s * MR + i) * IMG + st_dst) = st[s][i];
                }
        }
    };
    load_phase(0);
    // ---- weight loads of all of this wave's steps: step ph * PS + sw * SPW + q (past the end: the last step again, never used)
    uint4 w[PH][SPW][CTW];
#pragma unroll
    for (int t = 0; t < CTW; ++t) {
        int gn = n0 + 16 * (tg * CTW + t) + c;
        gn = gn < n ? gn : n - 1;
        const uint4* wp = reinterpret_cast<const uint4*>(pw + (int64_t)gn * ld_pw + 16 * g);
#pragma unroll
        for (int ph = 0; ph < PH; ++ph)
#pragma unroll
            for (int q = 0; q < SPW; ++q) {
                int s = ph * PS + sw * SPW + q;
                s = s < total ? s : total - 1;
                const nts_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nts_u32x4*>(wp + (int64_t)s * 4));
                w[ph][q][t] = make_uint4(v[0], v[1], v[2], v[3]);
            }
    }
    store_phase(0);
    __syncthreads();

    floatx4 acc[MR][CTW];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < CTW; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const DecodeConsts dc = make_decode_consts();
    constexpr int JB = MR > 2 ? 4 : 8;   // fragments read ahead of the multiplication (three or four row blocks: half a step, 64 registers)
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph < nph) {
            if (ph + 1 < PH && ph + 1 < nph) load_phase(ph + 1);      // the next phase's activations travel while this one is multiplied
#pragma unroll
            for (int q = 0; q < SPW; ++q) {
                if (ph * PS + sw * SPW + q < total) {
                    const uint4* img = xbuf + (sw * SPW + q) * MR * IMG + lane;
#pragma unroll
                    for (int jb = 0; jb < 8; jb += JB) {
                        uint4 xf[JB][MR];
#pragma unroll
                        for (int j = 0; j < JB; ++j)
#pragma unroll
                            for (int i = 0; i < MR; ++i) xf[j][i] = img[i * IMG + (jb + j) * FS];
#pragma unroll
                        for (int jj = 0; jj < JB; ++jj) {
                            const int j = jb + jj;
#pragma unroll
                            for (int t = 0; t < CTW; ++t) {
                                const uint4 wq = w[ph][q][t];
                                const uint32_t d = (j >> 1) == 0 ? wq.x : (j >> 1) == 1 ? wq.y : (j >> 1) == 2 ? wq.z : wq.w;
                                const uint4 a = (j & 1) ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
#pragma unroll
                                for (int i = 0; i < MR; ++i) acc[i][t] = mfma16<TX>(a, xf[jj][i], acc[i][t]);
                            }
                        }
                    }
                }
            }
            if (ph + 1 < PH && ph + 1 < nph) {
                __syncthreads();                                      // everyone has read this phase's image
                store_phase(ph + 1);
                __syncthreads();
            }
        }
    }
    __syncthreads();                                                  // the image has been read: its memory now takes the partial sums
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < CTW; ++t) red[(wave * MR * CTW + i * CTW + t) * 64 + lane] = acc[i][t];
    __syncthreads();
    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
    if (TW == 1 && planes == 3) {
        // float32 activations arrive as three bf16 planes in row blocks: rows r, m/3 + r, 2m/3 + r of the product are one output row
        const int mo = m / 3;
        for (int t = wave; t < CT; t += WAVES) {
            floatx4 s = {0.f, 0.f, 0.f, 0.f};
            if (c < mo) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const int R = pl * mo + c;
                    const floatx4* src = red + ((R >> 4) * CT + t) * 64 + g * 16 + (R & 15);
#pragma unroll
                    for (int ww = 0; ww < WAVES; ++ww) s += src[ww * MR * CT * 64];
                }
            }
            rows_store_fragment(s, c, n0 + 16 * t + 4 * g, mo, n, scale, bias, has_bias, act, cl, act_begin, y, y_dtype, ld_y);
        }
        return;
    }
    for (int f = wave; f < MR * CT; f += WAVES) {   // output fragment (row block i, tile T): the partial sums of tile group T / CTW, in wave order
        const int i = f / CT, T = f % CT;
        const floatx4* src = red + (((T / CTW) * SW) * MR * CTW + i * CTW + T % CTW) * 64 + lane;
        floatx4 s = src[0];
#pragma unroll
        for (int ww = 1; ww < SW; ++ww) s += src[ww * MR * CTW * 64];
        rows_store_fragment(s, 16 * i + c, n0 + 16 * T + 4 * g, m, n, scale, bias, has_bias, act, cl, act_begin, y, y_dtype, ld_y);
    }
}

template <typename TX, int MR, int CT, int SPW, int PH, int TW = 1>
static int launch_rows_staged_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                                  const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, int planes, hipStream_t st) {
    const size_t lds_bytes = rows_staged_lds_bytes(MR, CT, k);
    auto kern = gemm_rows_staged_kernel<TX, MR, CT, SPW, PH, TW>;
    if (int rc = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc != TVL_OK) return rc;
    const unsigned row_blocks = m > 64 ? (unsigned)ceil_div(m, 16 * MR) : 1u;   // (only the 64-row form is launched with more than 64 rows)
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(n, 16 * CT), row_blocks), dim3(512), lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip,
                       act, act_begin, y, y_dtype, ld_y, planes);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_rows_staged_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                                const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, int planes, hipStream_t st) {
    const int ct = rows_staged_tiles(m, n, k);
    if (m > 32) {   // 33 .. 64 rows, k <= 2048: three or four row blocks, phases of 4 steps, the waves split 2 or 4 channel tiles
#define TVL_STAGED_TW(MR, CT, PH) \
    return launch_rows_staged_cfg<TX, MR, CT, 1, PH, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, planes, st)
        if (k <= 2048) {
            if (m <= 48) { if (ct == 2) TVL_STAGED_TW(3, 2, 2); TVL_STAGED_TW(3, 4, 2); }   // (more than 64 rows: blocks of 64 over blockIdx.y)
            if (ct == 2) TVL_STAGED_TW(4, 2, 2);
            TVL_STAGED_TW(4, 4, 2);
        }
        if (m <= 48) { if (ct == 2) TVL_STAGED_TW(3, 2, 4); TVL_STAGED_TW(3, 4, 4); }   // k <= 4096: four phases
        if (ct == 2) TVL_STAGED_TW(4, 2, 4);
        TVL_STAGED_TW(4, 4, 4);
#undef TVL_STAGED_TW
    }
#define TVL_STAGED(MR, CT, SPW, PH) \
    return launch_rows_staged_cfg<TX, MR, CT, SPW, PH>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, planes, st)
#define TVL_STAGED_CT(MR, SPW, PH)                   \
    do {                                             \
        if (ct == 1) TVL_STAGED(MR, 1, SPW, PH);     \
        if (ct == 2) TVL_STAGED(MR, 2, SPW, PH);     \
        if (ct == 4 || MR == 2 || PH == 2) TVL_STAGED(MR, 4, SPW, PH); \
        TVL_STAGED(MR, 8, SPW, PH);                  \
    } while (0)
    if (m <= 16) {
        if (k <= 2048) TVL_STAGED_CT(1, 1, 1);
        if (k <= 4096) TVL_STAGED_CT(1, 2, 1);
        TVL_STAGED_CT(1, 2, 2);
    }
    if (k <= 2048) TVL_STAGED_CT(2, 1, 1);
    TVL_STAGED(2, 1, 1, 2);
#undef TVL_STAGED_CT
#undef TVL_STAGED
}

// x must fit in LDS in at most two phases (k <= 8192 up to 16 rows, k <= 4096 up to 32) and the grid should run in one round: every further
// round of workgroups stages x again (16 x 28672 x 4096, two rounds: 19.6 us against 14.2 for the shared-activation kernel), so a
// workgroup takes 2 or 4 channel tiles on wide layers.  Very small layers stay with gemm_rows_kernel (3.2 against 3.4 us at
// 16 x 512^2).  Measured otherwise (us, against the dispatch without this kernel):
// 16 x 4096^2 8.4 -> 5.9, 16 x 2048^2 5.1 -> 3.7, 16 x 6144 x 2048 9.3 -> 5.8, 8 x 4096^2 6.4 -> 5.7, 32 x 2048^2 8.0 -> 5.35,
// 32 x 4096^2 10.8 -> 8.4; against the shared-activation kernel 16 x 14336 x 2048 9.4 -> 6.8, 16 x 8192 x 4096 10.5 -> 7.9,
// 32 x 8192 x 2048 10.4 -> 6.6, 16 x 14336 x 4096 11.7 -> 10.8, but 16 x 32000 x 1024 6.7 -> 7.5; 4 x 6144 x 2048 6.2 -> 5.0.
// 33 .. 64 rows with k <= 2048 (three or four row blocks, phases of 4 steps, the waves split 2 or 4 channel tiles; one round of workgroups, so
// n <= 16384): 64 x 2048^2 10.5 -> 8.8, 64 x 6144 x 2048 13.0 -> 9.5, 64 x 14336 x 2048 14.5 -> 12.1, 48 x 14336 x 2048 15.2 -> 10.5,
// 40 x 6144 x 2048 12.3 -> 8.0, 64 x 1024^2 7.9 -> 5.5 against the 64-row matrix-core tile + reduce / the shared-activation kernel.
// TVL_ROWS_STAGED=0 turns it off (tuning), 2 lifts the size rules (tests).
// 3 .. 5 rows on up to 2048 channels with k > 4096: the plain kernel (four waves split k, x from global memory) beats both LDS-staged forms, whose
// staging of a long k is not repaid by so few rows and workgroups (3 x 2048 x 8192: 8.3 -> 7.4 us, 5 x 2048 x 6144: 8.1 -> 6.8)
static bool rows_few_long_k(int64_t m, int64_t n, int64_t k) { return m <= 5 && n <= 2048 && k > 4096; }
// 65 .. 512 rows, k <= 2048, few channels (the batch-256 out_proj of a Phi block, 256 x 2048 x 2048): too little work per output tile for
// the matrix-core tiles, which split k and add the slabs in a second launch.  The staged kernel takes blocks of 64 rows over
// blockIdx.y instead (x of a block = 256 KiB in two LDS phases, weights re-read from L2 by each row block), one launch, while one
// round of workgroups covers the output.  Measured (us, tiles -> row blocks): 256 x 2048^2 13.3 -> 9.4, 128 x 2048^2 11.2 -> 8.9,
// 256 x 4096 x 2048 14.8 -> 12.1, 512 x 2048^2 14.8 -> 12.5, 256 x 1024^2 9.0 -> 6.3, 320 x 1536 x 2048 13.3 -> 9.4.
// TVL_ROWS_BLOCKS=0 turns it off, 2 lifts the size rule (tuning, tests).
static bool rows_use_blocks(int64_t m, int64_t n, int64_t k) {
    static const int mode = [] { const char* e = getenv("TVL_ROWS_BLOCKS"); return e ? atoi(e) : 1; }();
    if (mode == 0 || m <= 64 || m > 512 || k > 4096 || n < 32) return false;
    const int64_t rb = ceil_div(m, 64), cus = device_cu_count();
    if (ceil_div(n, 64) * rb > cus) return false;          // one round of workgroups, one per compute unit (133 KiB of LDS)
    if (mode >= 2) return true;
    // four staging phases (k > 2048) are bound by every workgroup's pass over its rows of x through L2: they pay where the output is large
    // (256 x 3072^2 17.7 -> 15.8 us, 512 x 2048 x 4096 22.7 -> 20.2, 128 x 8192 x 4096 23.2 -> 20.1; 128 x 4096^2 15.9 -> 16.4, 96 x 4096^2 14.3 -> 15.6)
    if (k > 2048) return (double)m * (double)n >= 7.5e5;
    return k >= 1024 && (double)n * (double)k >= 5.0e5;   // (k = 512: 448 x 2048 x 512 6.9 -> 8.0 us, the tiles win)
}

static bool rows_use_staged(int64_t m, int64_t n, int64_t k) {
    static const int mode = [] { const char* e = getenv("TVL_ROWS_STAGED"); return e ? atoi(e) : 1; }();   // 2: whatever fits (tests)
    if (m > 64 || k > (m <= 16 ? 8192 : 4096)) return false;   // (17 .. 32 rows in four phases up to k = 8192: built, 12.4 - 13.9 us against 9.0 - 13.1 without)
    if (m > 32 && n < 32) return false;     // the waves of that form split at least two channel tiles
    if (mode == 2) return true;
    if (m > 32) {   // one round of workgroups (one per CU: 133 KiB of LDS) on two or four tiles each
        if (mode == 0) return false;
        // four phases (k > 2048) pay on wide layers only: 40 x 8192 x 4096 15.0 -> 13.0, 64 x 14336 x 4096 22.2 -> 19.0, 40 x 3072^2 10.9 -> 10.2, but 40 x 2048 x 4096 10.7 -> 12.1,
        // 64 x 8192 x 4096 13.6 -> 15.9
        if (k > 2048 && n < (m <= 48 ? 3072 : 10000)) return false;
        return (double)n * (double)k >= 1.0e6 && ceil_div(n, 64) <= (int64_t)device_cu_count();
    }
    const double weights = (double)n * (double)k;
    if (mode == 0 || m < 3 || weights < 1.0e6) return false;
    if (rows_few_long_k(m, n, k)) return false;
    if (k <= 1024 && weights >= 2.0e7) return false;
    const int mr = m <= 16 ? 1 : 2, ct = rows_staged_tiles(m, n, k);
    int64_t per_cu = (int64_t)(160 * 1024) / (int64_t)(rows_staged_lds_bytes(mr, ct, k) + 1024);
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    const int64_t rounds_allowed = (mr == 2 && k <= 2048) ? 2 : 1;   // at 17 .. 32 rows the alternatives are slower: two rounds still win
    return ceil_div(n, 16 * ct) <= rounds_allowed * per_cu * (int64_t)device_cu_count();
}

// ---- larger layers, up to 64 rows: the weight stream of 8 waves shares one copy of the activations in LDS --------------------
//
// Above, every wave fetches its own activation fragments from L2: 8 KiB per KiB of weights at 16 rows, in 16-byte pieces 128
// bytes apart (64 cache lines per load instruction), which bounds that kernel once the layer is large (16 x 16384 x 16384:
// 38 us, 1.7 TB/s of weights; a read-only probe with the same 16-channel x 64-byte weight pieces streams 5.5 TB/s from cold HBM -
// tools/hwtests/bw_probe.hip - so the weight pieces are not the limit).  Here a workgroup is 8 waves on 8 x 16 CT consecutive
// channels and ONE k range.  The activations of the range pass through LDS in chunks of CS steps, double-buffered:
//   * staging: wave w brings rows 2 w, 2 w + 1 of every 16-row block, half a wave per row - 32 lanes x 16 bytes, whole cache
//     lines - and writes each piece where the fragment layout wants it (fragment j = piece & 7, lane group piece >> 3; fragments
//     are padded to 65 x 16 bytes so the writes spread over the banks); a fragment is read back with one conflict-free
//     ds_read_b128 per lane, in blocks requested before the previous block is multiplied;
//   * memory returns loads in issue order, so waiting for a chunk of activations also waits for every weight load issued before
//     it: a chunk is requested two chunks (DEPTH steps) before its use and written to LDS one chunk before - the weight loads older
//     than it are then the ones the next step needs anyway, and the weight stream (DEPTH steps in flight per wave, registers) never
//     drains early;
//   * one barrier per chunk.
// k is split over blockIdx.y where the channel blocks alone do not fill the chip (float32 slabs + the deterministic reduce of
// gemm_mfma.hip).  What bounds it now is instruction issue: per 256-deep step a wave decodes 64 weights per lane (about 90
// vector-ALU instructions) for 8 matrix instructions, and the two waves of a SIMD do not overlap that work fully (per-step
// timestamps: a step takes ~800 cycles for the first wave of a SIMD and ~1600 for the second while both are in the same step);
// two workgroups per compute unit, 8 steps in flight and 32 channels per wave all measured no faster.
template <typename TX, int MR, int CT, int CS, bool SPLIT>
__global__ void __launch_bounds__(512)
gemm_rows_shared_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                        const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act, int act_begin, void* __restrict__ y,
                        int y_dtype, int64_t ld_y, int steps_per_split, float* __restrict__ slab) {
    constexpr int DEPTH = 4;          // steps of weights in flight per wave (8 measured no faster)
    constexpr int FS = 65;            // uint4 per fragment in LDS: 64 lanes + 1 of padding (spreads the staging writes over the banks)
    constexpr int IMG = 8 * FS;       // uint4 per 16-row step image
    static_assert(DEPTH % (2 * CS) == 0, "a group of DEPTH steps is an even number of chunks");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4* xbuf = reinterpret_cast<uint4*>(lds_raw);   // [2 buffers][CS steps][MR row blocks][8 fragments][FS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (128 * CT) + wave * (16 * CT);
    const int total = k / 256;
    const int s_begin = SPLIT ? (int)blockIdx.y * steps_per_split : 0;
    const int ns = SPLIT ? (total - s_begin < steps_per_split ? total - s_begin : steps_per_split) : total;
    const int last = ns - 1;

    const uint4* wp[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        int gn = n0 + 16 * t + c;
        gn = gn < n ? gn : n - 1;
        wp[t] = reinterpret_cast<const uint4*>(pw + (int64_t)gn * ld_pw + (int64_t)s_begin * 64 + 16 * g);
    }
    // staging: wave w brings rows 2 w, 2 w + 1 of every 16-row block, half a wave per row: 32 lanes x 16 bytes = the 512 contiguous
    // bytes of one step (whole cache lines).  Piece p (k = 8 p .. 8 p + 7) belongs to fragment j = p & 7, lane group p >> 3.
    const int st_row = 2 * wave + (lane >> 5), st_p = lane & 31;
    const uint4* xs[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        int gm = 16 * i + st_row;
        gm = gm < m ? gm : m - 1;
        xs[i] = reinterpret_cast<const uint4*>(x + (int64_t)gm * ld_x + (int64_t)s_begin * 256 + 8 * st_p);
    }
    const int st_dst = (st_p & 7) * FS + (st_p >> 3) * 16 + st_row;   // uint4 index inside one 16-row step image
    floatx4 acc[MR][CT];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const DecodeConsts dc = make_decode_consts();

    auto load_w = [&](int s, uint4 (&w)[CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const nts_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nts_u32x4*>(wp[t] + (int64_t)s * 4));
            w[t] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto stage_load = [&](int chunk, nts_u32x4 (&st)[CS][MR]) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < CS; ++sl) {
            int s = chunk * CS + sl;
            s = s < last ? s : last;             // past the range: the last step again (never multiplied)
#pragma unroll
            for (int i = 0; i < MR; ++i) st[sl][i] = *reinterpret_cast<const nts_u32x4*>(xs[i] + (int64_t)s * 32);
        }
    };
    auto stage_store = [&](int buf, const nts_u32x4 (&st)[CS][MR]) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < CS; ++sl)
#pragma unroll
            for (int i = 0; i < MR; ++i) *reinterpret_cast<nts_u32x4*>(xbuf + ((buf * CS + sl) * MR + i) * IMG + st_dst) = st[sl][i];
    };
    // One step: the 8 fragments are read from LDS in blocks of JB, each block requested before the previous one is multiplied (a read
    // issued right before its use costs the full LDS latency, and two waves per SIMD do not hide it).
    constexpr int JB = MR == 1 ? 4 : (MR == 2 ? 2 : 1), NBLK = 8 / JB;
    auto read_block = [&](int buf, int sl, int blk, uint4 (&xf)[JB][MR]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj)
#pragma unroll
            for (int i = 0; i < MR; ++i)
                xf[jj][i] = xbuf[((buf * CS + sl) * MR + i) * IMG + (blk * JB + jj) * FS + lane];
    };
    auto compute_block = [&](int blk, const uint4 (&w)[CT], const uint4 (&xf)[JB][MR]) __attribute__((always_inline)) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int j = blk * JB + jj;   // dword j >> 1 of the lane's 16 bytes, halfword j & 1
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const uint32_t d = (j >> 1) == 0 ? w[t].x : (j >> 1) == 1 ? w[t].y : (j >> 1) == 2 ? w[t].z : w[t].w;
                const uint4 a = (j & 1) ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
#pragma unroll
                for (int i = 0; i < MR; ++i) acc[i][t] = mfma16<TX>(a, xf[jj][i], acc[i][t]);
            }
        }
    };
    auto compute_step = [&](int buf, int sl, const uint4 (&w)[CT]) __attribute__((always_inline)) {
        uint4 xa[JB][MR], xb[JB][MR];
        read_block(buf, sl, 0, xa);
#pragma unroll
        for (int blk = 0; blk < NBLK; blk += 2) {
            read_block(buf, sl, blk + 1, xb);
            __builtin_amdgcn_sched_barrier(0);
            compute_block(blk, w, xa);
            if (blk + 2 < NBLK) read_block(buf, sl, blk + 2, xa);
            __builtin_amdgcn_sched_barrier(0);
            compute_block(blk + 1, w, xb);
        }
    };

    {
        uint4 w0[CT], w1[CT], w2[CT], w3[CT];
        // Memory returns loads in issue order, so waiting for a chunk of activations also waits for every weight load issued
        // before it.  A chunk is therefore requested TWO chunks (DEPTH steps) before its use and written to LDS one chunk before:
        // the weight loads older than it are then the ones the next step needs anyway, and the stream never drains early.
        nts_u32x4 sta[CS][MR], stb[CS][MR];   // even / odd chunks (separate arrays: one indexed array ends up in scratch memory)
        stage_load(0, sta);
        stage_load(1, stb);
        load_w(0, w0);
        load_w(1 < last ? 1 : last, w1);
        load_w(2 < last ? 2 : last, w2);
        load_w(3 < last ? 3 : last, w3);
        stage_store(0, sta);
        __syncthreads();
        // step s + Q of a group of DEPTH steps; Q is a literal, so chunk boundaries, LDS buffer and staging registers are known at
        // compile time (s is a multiple of DEPTH, DEPTH / CS is even: chunk (s + Q) / CS has the parity of Q / CS).  Past the end of
        // the range only the multiplication is skipped: loads (clamped to the last step), LDS writes and barriers run
        // unconditionally, so no register is live across a branch it is written in.
#define TVL_SHARED_STEP(Q, W)                                                                          \
        {                                                                                              \
            if ((Q) % CS == 0) stage_load((s + (Q)) / CS + 2, (((Q) / CS) & 1) ? stb : sta); \
            if (s + (Q) < ns) compute_step(((Q) / CS) & 1, (Q) % CS, W);                               \
            load_w(s + (Q) + DEPTH < last ? s + (Q) + DEPTH : last, W);                                \
            if ((Q) % CS == CS - 1) {                                                                 \
                stage_store((((Q) / CS) & 1) ^ 1, (((Q) / CS) & 1) ? sta : stb);                       \
                __syncthreads();                                                                       \
            }                                                                                          \
        }
        for (int s = 0; s < ns; s += DEPTH) {
            TVL_SHARED_STEP(0, w0)
            TVL_SHARED_STEP(1, w1)
            TVL_SHARED_STEP(2, w2)
            TVL_SHARED_STEP(3, w3)
        }
#undef TVL_SHARED_STEP
    }

    if constexpr (SPLIT) {
        float* mine = slab + (int64_t)blockIdx.y * ((int64_t)m * n);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int gm = 16 * i + c, cn = n0 + 16 * t + 4 * g;
                if (gm >= m) continue;
                if ((n & 3) == 0 && cn + 4 <= n) {
                    *reinterpret_cast<float4*>(mine + (int64_t)gm * n + cn) = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cn + e < n) mine[(int64_t)gm * n + cn + e] = acc[i][t][e];
                }
            }
    } else {
        const float cl = fabsf(clip);
        const bool has_bias = bias != nullptr;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int t = 0; t < CT; ++t)
                rows_store_fragment(acc[i][t], 16 * i + c, n0 + 16 * t + 4 * g, m, n, scale, bias, has_bias, act, cl, act_begin, y, y_dtype, ld_y);
    }
}

struct SharedPlan { int mr, cs, splits, per; double us; };   // per = steps of 256 k per split; us = modelled time

// The k split, by a small cost model fitted to measurements: 0.65 / 1.0 / 1.53 us per 256-deep step of one workgroup at 16 / 32 /
// 64 rows, the grid running in ceil(workgroups / CUs) rounds of `per` steps, a few us of launch, prologue and slab reduce.  (32
// channels per wave - half the LDS reads per weight - measured 4 % faster at 32 and 64 rows x 16384 x 16384 and slower or equal
// everywhere else: not built.)  TVL_ROWS_SHARED=<splits> pins the split, a negative value turns the kernel off (tuning).
static SharedPlan plan_rows_shared(int64_t m, int64_t n, int64_t k, size_t workspace_bytes) {
    static const int force_splits = [] { const char* e = getenv("TVL_ROWS_SHARED"); return e ? atoi(e) : 0; }();
    static const double step_us[3] = {0.65, 1.0, 1.53};
    SharedPlan best;
    best.mr = m <= 16 ? 1 : (m <= 32 ? 2 : 4);
    best.cs = best.mr == 4 ? 1 : 2;
    best.splits = 1; best.per = (int)(k / 256); best.us = 1e30;
    const int64_t steps = k / 256;
    const int cus = device_cu_count();
    const int64_t by_ws = (int64_t)(workspace_bytes / ((size_t)m * n * sizeof(float)));
    const int64_t max_s = steps / 4 > 0 ? steps / 4 : 1;             // at least 4 steps (1024 k) per workgroup
    const int64_t nblk = ceil_div(n, 128);
    for (int64_t s = 1; s <= max_s; ++s) {
        if (force_splits > 0 && s != (force_splits < max_s ? force_splits : max_s)) continue;
        if (s > 1 && s > by_ws) break;
        const int64_t per = ceil_div(steps, s), splits = ceil_div(steps, per);
        if (splits != s) continue;                                     // the same split under a smaller s
        const int64_t rounds = ceil_div(nblk * splits, (int64_t)cus);
        const double us = 4.0 + (splits > 1 ? 2.0 + 0.25 * splits : 0.0) + (double)rounds * (1.0 + per * step_us[best.mr == 1 ? 0 : (best.mr == 2 ? 1 : 2)]);
        if (us < best.us) { best.us = us; best.splits = (int)splits; best.per = (int)per; }
    }
    return best;
}

size_t gemm_rows_shared_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m > 64 || k % 256 != 0) return 0;
    const SharedPlan p = plan_rows_shared(m, n, k, (size_t)1 << 40);
    return p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
}

bool gemm_rows_shared_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                                const void* workspace, size_t workspace_bytes) {
    static const int mode = [] { const char* e = getenv("TVL_ROWS_SHARED"); return e ? atoi(e) : 0; }();
    static const bool any_size = [] { const char* e = getenv("TVL_ROWS_SHARED_ANY"); return e && atoi(e) != 0; }();   // tests: small shapes too
    if (mode < 0) return false;
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16) return false;
    if (m < 3 || m > 64 || k <= 0 || k % 256 != 0) return false;   // one or two rows: the table-lookup kernel
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    if (!aligned(pw, 16) || ld_pw % 16 != 0) return false;
    if (n > INT32_MAX - 512 || k > INT32_MAX) return false;
    if (any_size) return true;
    if (rows_use_staged(m, n, k)) return false;   // x fits in LDS and one round of workgroups covers n: the one-launch kernel above
    if (rows_few_long_k(m, n, k)) return false;
    // Measured crossovers against the kernels above / the split-K tiles, in weights (n k): e.g. us at 16 rows 15.4 -> 9.4 at
    // 14336 x 2048, 17.1 -> 10.7 at 4096 x 8192, 19.1 -> 12.8 at 8192^2, 39 -> 29 at 16384^2, but 8.5 -> 9.2 at 4096^2; at 48
    // rows 42 -> 21 at 8192^2; at 64 rows the 64-row matrix-core tile keeps 8192^2 (20.5 against 22.5) and loses 16384^2 (60 -> 52).
    const double weights = (double)n * (double)k;
    // (33 .. 48 rows: from 1.2e7 weights on, where the alternative is a 64-row matrix-core tile with a k split - 40 x 6144 x 2048: 17.0 -> 12.4,
    // 48 x 14336 x 2048: 18.2 -> 15.2; 49 .. 64 rows: from 1e8 on - 64 x 28672 x 4096: 32.8 -> 26.0, while 64 x 8192^2 stays with the tile)
    const double need = m <= 16 ? (k >= 8192 ? 1.5e7 : 2.0e7) : (m <= 32 ? 2.5e7 : (m <= 48 ? 1.2e7 : 1.0e8));
    if (weights < need) return false;
    const SharedPlan p = plan_rows_shared(m, n, k, (workspace && aligned(workspace, 16)) ? workspace_bytes : 0);
    // without the workspace for its k split the grid would underfill the chip: leave the shape to the other kernels
    // (with at most two k ranges three eighths of the chip are enough: 40 x 6144 x 2048 runs 96 workgroups and still beats the 64-row tile +
    // reduce, 12.4 against 17.0 us; 40 x 3072 x 4096, the same 96 workgroups over four k ranges, loses 13.0 against 12.3)
    const int64_t wgs = p.splits * ceil_div(n, 128);
    return wgs * 2 >= device_cu_count() || (p.splits <= 2 && wgs * 8 >= device_cu_count() * 3);
}

template <typename TX, int MR, int CS>
static int launch_shared_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                             const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, const SharedPlan& p, float* slab,
                             hipStream_t st) {
    const size_t lds_bytes = (size_t)2 * CS * MR * 8 * 65 * 16;
    constexpr int CT = 1;
    const dim3 grid((unsigned)ceil_div(n, 128 * CT), (unsigned)p.splits), block(512);
    if (p.splits > 1) {
        auto kern = gemm_rows_shared_kernel<TX, MR, CT, CS, true>;
        if (int rc = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc != TVL_OK) return rc;
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y,
                           p.per, slab);
        TVL_LAUNCH_CHECK();
        return launch_splitk_reduce(slab, p.splits, m, n, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    }
    auto kern = gemm_rows_shared_kernel<TX, MR, CT, CS, false>;
    if (int rc = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc != TVL_OK) return rc;
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, 0,
                       (float*)nullptr);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_shared_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                           const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, const SharedPlan& p, float* slab,
                           hipStream_t st) {
    if (p.mr == 1) return launch_shared_cfg<TX, 1, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, p, slab, st);
    if (p.mr == 2) return launch_shared_cfg<TX, 2, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, p, slab, st);
    return launch_shared_cfg<TX, 4, 1>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, p, slab, st);
}

int launch_gemm_rows_shared(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                            const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                            void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (y_dtype != TVL_F32 && y_dtype != TVL_F16 && y_dtype != TVL_BF16) {
        set_error("ternary_matmul: unsupported output dtype %d", y_dtype);
        return TVL_ERR_INVALID;
    }
    const bool ws_ok = workspace != nullptr && aligned(workspace, 16);
    const SharedPlan p = plan_rows_shared(m, n, k, ws_ok ? workspace_bytes : 0);
    float* slab = reinterpret_cast<float*>(workspace);
    if (x_dtype == TVL_F16)
        return launch_shared_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, p, slab, st);
    return launch_shared_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, p, slab, st);
}

}  // namespace tvl
