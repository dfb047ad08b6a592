// Binarized activations x ternary weights by AND + popcount (the larq `input_quantizer="ste_sign"` mode of the
// reference's ScaledTernary layer, impl/tf/moondream/ternary.py:36, with sign_ste of binary/modules.py:5-10:
// s = +1 if x >= 0 else -1).
//
//   sum_k s_k t_k = sum_{t=+1} s - sum_{t=-1} s = 2 (popc(xb & pos) - popc(xb & neg)) - (popc(pos) - popc(neg))
//
// with one bit per element: xb = (x >= 0), pos = (t == +1), neg = (t == -1); bit i of word w is element 32 w + i.
// Everything is integer and exact; the float32 epilogue (scale, bias, clip, activation) is the matmul's.
#include "tvl_common.h"
#include "tvl_kernels.h"

namespace tvl {

// ---- x -> sign bits: one wave turns 256 consecutive elements of a row into 8 words (4 ballots) ---------------
template <typename T>
__global__ void __launch_bounds__(256) sign_bits_kernel(const T* __restrict__ x, int64_t m, int64_t k, int64_t ld_x,
                                                        uint32_t* __restrict__ xb, int64_t ld_xb) {
    const int lane = threadIdx.x & 63;
    const int64_t chunks_per_row = (k + 255) / 256, total = m * chunks_per_row;
    const int64_t words_per_row = (k + 31) / 32;
    for (int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < total; c += (int64_t)gridDim.x * 4) {
        const int64_t row = c / chunks_per_row, k0 = (c - row * chunks_per_row) * 256;
        unsigned long long mask[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t kk = k0 + 64 * q + lane;
            const bool nonneg = kk < k && to_float(x[row * ld_x + kk]) >= 0.0f;   // pad bits are 0; NaN -> 0 like (x >= 0)
            mask[q] = __ballot(nonneg);
        }
        if (lane < 8) {
            const int64_t w = k0 / 32 + lane;
            if (w < words_per_row) xb[row * ld_xb + w] = (uint32_t)(mask[lane >> 1] >> (32 * (lane & 1)));
        }
    }
}

// ---- packed 2-bit rows -> (pos, neg) bit planes: one thread per output word (8 packed bytes) ---------------------
__device__ __forceinline__ uint32_t compress_even_bits(unsigned long long v) {
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)v;
}

__global__ void __launch_bounds__(256) bitplanes_kernel(const uint8_t* __restrict__ pw, int64_t n, int64_t k, int64_t ld_pw,
                                                        uint32_t* __restrict__ pos, uint32_t* __restrict__ neg, int64_t ld_planes) {
    const int64_t words = (k + 31) / 32, bytes = (k + 3) / 4, total = n * words;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / words, w = i - row * words;
        unsigned long long v = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int64_t j = 8 * w + b;
            unsigned long long byte = j < bytes ? pw[row * ld_pw + j] : 0x55u;   // beyond the row: code 1 (value 0)
            if (j == bytes - 1 && (k & 3)) {                                     // pad codes inside the last byte -> code 1
                const unsigned keep = (1u << (2 * (k & 3))) - 1u;
                byte = (byte & keep) | (0x55u & ~keep);
            }
            v |= byte << (8 * b);
        }
        const unsigned long long hi = v >> 1, lo = v;
        pos[row * ld_planes + w] = compress_even_bits(hi);           // codes 2 and 3 (bit 1 set): +1
        neg[row * ld_planes + w] = compress_even_bits(~(hi | lo));   // code 0: -1
    }
}

// ---- popcount matmul, few rows: one wave per output channel, lanes split the words, DPP sum over the wave ------------
template <int v>
__device__ __forceinline__ int dpp_add(int x) { return x + __builtin_amdgcn_update_dpp(0, x, v, 0xF, 0xF, true); }
__device__ __forceinline__ int wave_sum(int x) {
    x = dpp_add<0x111>(x);   // row_shr:1
    x = dpp_add<0x112>(x);   // row_shr:2
    x = dpp_add<0x114>(x);   // row_shr:4
    x = dpp_add<0x118>(x);   // row_shr:8  -> lane 15 of every row of 16 holds the row's sum
    int t = __builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63);
    return t;
}

template <typename TY, int ROWS>
__global__ void __launch_bounds__(256) popcount_rows_kernel(const uint32_t* __restrict__ xb, int m, int64_t ld_xb,
                                                            const uint32_t* __restrict__ pos, const uint32_t* __restrict__ neg, int64_t n,
                                                            int64_t ld_planes, int words, const float* __restrict__ scale,
                                                            const float* __restrict__ bias, float clip, int act, int act_begin, TY* __restrict__ y, int64_t ld_y) {
    const int lane = threadIdx.x & 63;
    const int64_t ch = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (ch >= n) return;
    int p[ROWS], q[ROWS], cnt = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) p[r] = q[r] = 0;
    for (int w = lane; w < words; w += 64) {
        // once-read plane stream: non-temporal
        const uint32_t pw_ = __builtin_nontemporal_load(pos + ch * ld_planes + w), nw_ = __builtin_nontemporal_load(neg + ch * ld_planes + w);
        cnt += __popc(pw_) - __popc(nw_);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (r < m) {
                const uint32_t xw = xb[(int64_t)r * ld_xb + w];
                p[r] += __popc(xw & pw_);
                q[r] += __popc(xw & nw_);
            }
        }
    }
    const int c = wave_sum(cnt);
    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (r < m) {
            const int dot = 2 * wave_sum(p[r] - q[r]) - c;
            if (lane == 0)
                y[(int64_t)r * ld_y + ch] = from_float<TY>(epilogue_rt((float)dot, scale[ch], has_bias ? bias[ch] : 0.0f, has_bias, act, cl, (int)ch, act_begin));
        }
    }
}

// ---- popcount matmul, general: 64 x 64 output tile per workgroup, 4 x 4 outputs per thread, planes staged in LDS ------
template <typename TY>
__global__ void __launch_bounds__(256) popcount_tile_kernel(const uint32_t* __restrict__ xb, int64_t m, int64_t ld_xb,
                                                            const uint32_t* __restrict__ pos, const uint32_t* __restrict__ neg, int64_t n,
                                                            int64_t ld_planes, int words, const float* __restrict__ scale,
                                                            const float* __restrict__ bias, float clip, int act, int act_begin, TY* __restrict__ y, int64_t ld_y) {
    constexpr int WC = 32;                      // words per k chunk (1024 elements)
    __shared__ uint32_t sx[64][WC + 1], sp[64][WC + 1], sn[64][WC + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
    int acc[4][4], cnt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    }
    for (int w0 = 0; w0 < words; w0 += WC) {
        for (int i = tid; i < 64 * WC; i += 256) {
            const int r = i / WC, w = i % WC;
            const bool inw = w0 + w < words;
            sx[r][w] = (inw && m0 + r < m) ? xb[(m0 + r) * ld_xb + w0 + w] : 0u;
            sp[r][w] = (inw && n0 + r < n) ? pos[(n0 + r) * ld_planes + w0 + w] : 0u;
            sn[r][w] = (inw && n0 + r < n) ? neg[(n0 + r) * ld_planes + w0 + w] : 0u;
        }
        __syncthreads();
#pragma unroll 4
        for (int w = 0; w < WC; ++w) {
            uint32_t xv[4], pv[4], nv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { xv[i] = sx[ty * 4 + i][w]; pv[i] = sp[tx * 4 + i][w]; nv[i] = sn[tx * 4 + i][w]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt[j] += __popc(pv[j]) - __popc(nv[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += __popc(xv[i] & pv[j]) - __popc(xv[i] & nv[j]);
        }
        __syncthreads();
    }
    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + ty * 4 + i;
        if (row >= m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ch = n0 + tx * 4 + j;
            if (ch < n)
                y[row * ld_y + ch] = from_float<TY>(epilogue_rt((float)(2 * acc[i][j] - cnt[j]), scale[ch], has_bias ? bias[ch] : 0.0f, has_bias, act, cl, (int)ch, act_begin));
        }
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------------
int launch_sign_bits(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, uint32_t* xb, int64_t ld_xb, hipStream_t st) {
    int64_t g = ceil_div(m * ceil_div(k, 256), 4);
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    switch (x_dtype) {
        case TVL_F32: hipLaunchKernelGGL(sign_bits_kernel<float>, dim3((unsigned)g), dim3(256), 0, st, (const float*)x, m, k, ld_x, xb, ld_xb); break;
        case TVL_F16: hipLaunchKernelGGL(sign_bits_kernel<f16>, dim3((unsigned)g), dim3(256), 0, st, (const f16*)x, m, k, ld_x, xb, ld_xb); break;
        case TVL_BF16: hipLaunchKernelGGL(sign_bits_kernel<bf16_t>, dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)x, m, k, ld_x, xb, ld_xb); break;
        default: set_error("sign_pack_bits: unsupported dtype %d", x_dtype); return TVL_ERR_INVALID;
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_bitplanes(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint32_t* pos, uint32_t* neg, int64_t ld_planes, hipStream_t st) {
    int64_t g = ceil_div(n * ceil_div(k, 32), 256);
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    hipLaunchKernelGGL(bitplanes_kernel, dim3((unsigned)g), dim3(256), 0, st, pw, n, k, ld_pw, pos, neg, ld_planes);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TY>
static int launch_popcount_t(const uint32_t* xb, int64_t m, int64_t ld_xb, const uint32_t* pos, const uint32_t* neg, int64_t n, int64_t ld_planes,
                             int64_t k, const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, hipStream_t st) {
    const int words = (int)ceil_div(k, 32);
    if (m <= 8) {
        const dim3 grid((unsigned)ceil_div(n, 4)), block(256);
        if (m <= 2) hipLaunchKernelGGL((popcount_rows_kernel<TY, 2>), grid, block, 0, st, xb, (int)m, ld_xb, pos, neg, n, ld_planes, words, scale, bias, clip, act, act_begin, y, ld_y);
        else hipLaunchKernelGGL((popcount_rows_kernel<TY, 8>), grid, block, 0, st, xb, (int)m, ld_xb, pos, neg, n, ld_planes, words, scale, bias, clip, act, act_begin, y, ld_y);
    } else {
        const dim3 grid((unsigned)ceil_div(n, 64), (unsigned)ceil_div(m, 64)), block(256);
        hipLaunchKernelGGL((popcount_tile_kernel<TY>), grid, block, 0, st, xb, m, ld_xb, pos, neg, n, ld_planes, words, scale, bias, clip, act, act_begin, y, ld_y);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_popcount_matmul(const uint32_t* xb, int64_t m, int64_t ld_xb, const uint32_t* pos, const uint32_t* neg, int64_t n, int64_t ld_planes,
                           int64_t k, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_popcount_t(xb, m, ld_xb, pos, neg, n, ld_planes, k, scale, bias, clip, act, act_begin, (float*)y, ld_y, st);
        case TVL_F16: return launch_popcount_t(xb, m, ld_xb, pos, neg, n, ld_planes, k, scale, bias, clip, act, act_begin, (f16*)y, ld_y, st);
        case TVL_BF16: return launch_popcount_t(xb, m, ld_xb, pos, neg, n, ld_planes, k, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, st);
        default: set_error("popcount_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

}  // namespace tvl
