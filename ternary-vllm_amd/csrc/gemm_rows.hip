// Packed-ternary x f16 / bf16 product for few activation rows (decode batches of 3 .. 32 rows), ONE launch.
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

namespace tvl {

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
        const int gm = 16 * i + c, cn = n0 + 16 * t + 4 * g;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = epilogue_rt(s[e], cn + e < n ? scale[cn + e] : 0.0f, (has_bias && cn + e < n) ? bias[cn + e] : 0.0f, has_bias, act, cl, cn + e, act_begin);
        if (gm >= m) continue;
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
}

// Every workgroup reads all of x through L2 (16 or 32 channels per workgroup), so the kernel's time grows with m * n * k:
// measured (us, n = k = 4096) 5.1 / 5.7 / 6.5 / 8.5 at m = 2 / 4 / 8 / 16 against 5.4 / 8.2 / 10.2 / 11.4 for the kernels
// that split k over workgroups and reduce in a second launch, but 25 against 19 at 16 x 8192 x 8192, where the
// activation traffic (m k n / 8 bytes) outgrows the weight stream.  (Built and dropped for those layers: the four waves on
// different channels of one k range with its activations staged once in LDS, k split over workgroups, slabs + reduce launch -
// 48 us at 16 x 16384 x 16384 against 39 us for the older streaming kernel: the 16-channel x 64-byte weight pieces, half a
// cache line per channel, stream at ~2 TB/s whatever feeds the other operand.)  It is used where it wins: 3 .. 16 rows (one or two rows
// go to the table-lookup kernel) up to m n k = 3e8, and 17 .. 32 rows on small layers.
bool gemm_rows_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw) {
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16) return false;
    if (m < 3 || m > 32 || k <= 0 || k % 256 != 0) return false;
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    if (!aligned(pw, 16) || ld_pw % 16 != 0) return false;
    if (n > INT32_MAX - 64 || k > INT32_MAX) return false;
    const double work = (double)m * (double)n * (double)k;
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

int launch_gemm_rows(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                     const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    if (y_dtype != TVL_F32 && y_dtype != TVL_F16 && y_dtype != TVL_BF16) {
        set_error("ternary_matmul: unsupported output dtype %d", y_dtype);
        return TVL_ERR_INVALID;
    }
    if (x_dtype == TVL_F16) return launch_rows_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    return launch_rows_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
}

}  // namespace tvl
