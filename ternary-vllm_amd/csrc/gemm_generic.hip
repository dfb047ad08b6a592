// Shape-agnostic packed ternary matmul (any m, n, k, leading dimensions and alignment) and the
// dense-weight select-and-sum op.  The path for odd shapes and misaligned buffers; the tuned kernels in
// gemm_mfma.hip, gemm_stream.hip, gemm_skinny.hip and gemv.hip take the aligned cases.
//
// y[m,n] = act(clip((sum_k x[m,k] * (code(pw[n,k]) - 1)) * scale[n] + bias[n]))
// accumulated in float32: one fmaf per term inside each 32-deep k chunk, chunk sums added to the
// running total (two-level summation keeps the rounding error near that of a blocked BLAS).
#include "tvl_common.h"
#include "tvl_kernels.h"

namespace tvl {

constexpr int GT = 64;   // output tile 64 x 64
constexpr int GK = 32;   // k chunk (8 packed bytes per weight row)

template <typename TX, typename TY, bool HAS_BIAS, int ACT>
__global__ void __launch_bounds__(256) gemm_generic_kernel(const TX* __restrict__ x, int64_t m, int64_t k, int64_t ld_x,
                                                           const uint8_t* __restrict__ pw, int64_t n, int64_t ld_pw,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           float clip, int act_begin, TY* __restrict__ y, int64_t ld_y) {
    __shared__ float xs[GK][GT + 1];  // [k][m] : thread reads 4 rows at fixed k -> stride-1 over ty*4+i
    __shared__ float ws[GK][GT + 1];  // [k][n]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * GT, n0 = (int64_t)blockIdx.x * GT;
    float acc[4][4] = {};
    for (int64_t k0 = 0; k0 < k; k0 += GK) {
        // stage x: 64 rows x 32 k = 2048 elements, 8 per thread; consecutive threads along k
        for (int e = threadIdx.x; e < GT * GK; e += 256) {
            const int r = e / GK, c = e % GK;
            const int64_t mm = m0 + r, kk = k0 + c;
            xs[c][r] = (mm < m && kk < k) ? to_float(x[mm * ld_x + kk]) : 0.0f;
        }
        // stage w: 64 rows x 8 bytes; each thread decodes 2 bytes -> 8 values
        for (int e = threadIdx.x; e < GT * (GK / 4); e += 256) {
            const int r = e / (GK / 4), jb = e % (GK / 4);
            const int64_t nn = n0 + r, kk = k0 + 4 * jb;
            uint32_t b = 0x55;  // four zeros
            if (nn < n && kk < k) b = pw[nn * ld_pw + kk / 4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ws[4 * jb + i][r] = (float)code_value(b, i);
        }
        __syncthreads();
        float part[4][4] = {};
#pragma unroll 8
        for (int c = 0; c < GK; ++c) {
            float xv[4], wv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xv[i] = xs[c][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wv[j] = ws[c][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[i][j] = fmaf(xv[i], wv[j], part[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += part[i][j];
        __syncthreads();
    }
    const float c = fabsf(clip);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t nn = n0 + tx + 16 * j;
        if (nn >= n) continue;
        const float s = scale[nn];
        const float b = HAS_BIAS ? bias[nn] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t mm = m0 + ty * 4 + i;
            if (mm < m) y[mm * ld_y + nn] = from_float<TY>(epilogue_ch<HAS_BIAS, ACT>(acc[i][j], s, b, c, (int)nn, act_begin));
        }
    }
}

template <typename TX, typename TY>
static int launch_generic_t(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                            const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(n, GT), (unsigned)ceil_div(m, GT)), block(256);
#define TVL_GG(HB, ACT) \
    hipLaunchKernelGGL((gemm_generic_kernel<TX, TY, HB, ACT>), grid, block, 0, st, x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act_begin, y, ld_y)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_GG(true, TVL_ACT_GELU_TANH); else TVL_GG(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_GG(false, TVL_ACT_GELU_TANH); else TVL_GG(false, TVL_ACT_NONE);
    }
#undef TVL_GG
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_generic_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                            const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                            hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_generic_t(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, st);
        case TVL_F16: return launch_generic_t(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, st);
        case TVL_BF16: return launch_generic_t(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

int launch_gemm_generic(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                        int64_t ld_pw, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype,
                        int64_t ld_y, hipStream_t st) {
    if (ceil_div(m, GT) > 65535) { set_error("ternary_matmul: m = %lld too large for the generic kernel", (long long)m); return TVL_ERR_INVALID; }
    switch (x_dtype) {
        case TVL_F32: return launch_generic_x((const float*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
        case TVL_F16: return launch_generic_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
        case TVL_BF16: return launch_generic_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
        default: set_error("ternary_matmul: unsupported activation dtype %d", x_dtype); return TVL_ERR_INVALID;
    }
}

// ---- dense-weight select-and-sum (matrix_mul_tests.py:29-37) ---------------------------------
// res[m, i] = (sum_{tm[i,k] > 0} x[m,k]) - (sum_{tm[i,k] < 0} x[m,k]) : two separate float32 sums
// in ascending k, then one subtraction, like the reference's sp.sum(dim=1) - sn.sum(dim=1).
__global__ void __launch_bounds__(256) ternary_mm_kernel(const float* __restrict__ x, int64_t m, int64_t k, int64_t ld_x,
                                                         const float* __restrict__ tm, int64_t n, int64_t ld_tm,
                                                         float* __restrict__ res, int64_t ld_res) {
    __shared__ float xs[GK][GT + 1];
    __shared__ float ws[GK][GT + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * GT, n0 = (int64_t)blockIdx.x * GT;
    float pos[4][4] = {}, neg[4][4] = {};
    for (int64_t k0 = 0; k0 < k; k0 += GK) {
        for (int e = threadIdx.x; e < GT * GK; e += 256) {
            const int r = e / GK, c = e % GK;
            const int64_t kk = k0 + c;
            xs[c][r] = (m0 + r < m && kk < k) ? x[(m0 + r) * ld_x + kk] : 0.0f;
            ws[c][r] = (n0 + r < n && kk < k) ? tm[(n0 + r) * ld_tm + kk] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int c = 0; c < GK; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wv = ws[c][tx + 16 * j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = xs[c][ty * 4 + i];
                    pos[i][j] += (wv > 0.0f) ? xv : 0.0f;
                    neg[i][j] += (wv < 0.0f) ? xv : 0.0f;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t nn = n0 + tx + 16 * j;
        if (nn >= n) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t mm = m0 + ty * 4 + i;
            if (mm < m) res[mm * ld_res + nn] = pos[i][j] - neg[i][j];
        }
    }
}

int launch_ternary_mm(const float* x, int64_t m, int64_t k, int64_t ld_x, const float* tm, int64_t n, int64_t ld_tm, float* res,
                      int64_t ld_res, hipStream_t st) {
    if (ceil_div(m, GT) > 65535) { set_error("ternary_mm: m too large"); return TVL_ERR_INVALID; }
    const dim3 grid((unsigned)ceil_div(n, GT), (unsigned)ceil_div(m, GT)), block(256);
    hipLaunchKernelGGL(ternary_mm_kernel, grid, block, 0, st, x, m, k, ld_x, tm, n, ld_tm, res, ld_res);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

}  // namespace tvl
