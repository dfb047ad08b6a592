// One-time scale fits of a dense weight matrix (binary/modules.py:189-299).
//
// Ternary: masked-mean fixed point  m0 = mean|w|;  repeat: mask = (|w| - m) > -m/2,
// m = sum(|w| mask) / count(mask)   (modules.py:284-288), per output row or for the whole matrix.
// Binary:  (sum_{w>0} w - sum_{w<0} w) / count  or  sum_{w>0} w / count  (modules.py:197-210, 236-245).
// Sums are float32 per thread, combined in a fixed order (deterministic) and finished in float64.
#include "tvl_common.h"
#include "tvl_kernels.h"

namespace tvl {

struct Acc2 {
    double s;              // sum
    unsigned long long c;  // count (or a second sum's bits, see binary fit)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of (float, unsigned) pairs; result valid in every thread
__device__ __forceinline__ void block_sum(float& s, unsigned& c, float* sm_s, unsigned* sm_c) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    s = wave_sum(s);
    c = wave_sum_u(c);
    __syncthreads();  // protect sm_* from the previous use
    if (lane == 0) { sm_s[wid] = s; sm_c[wid] = c; }
    __syncthreads();
    float ts = 0.0f; unsigned tc = 0;
    for (int i = 0; i < nw; ++i) { ts += sm_s[i]; tc += sm_c[i]; }
    s = ts; c = tc;
}

// ---- per-row ternary fit: one block per row, the row is re-read (cache resident) each pass ----
__global__ void __launch_bounds__(256) fit_ternary_rows_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                               int iters, float* __restrict__ scale, float* __restrict__ latent,
                                                               int64_t ld_latent) {
    __shared__ float sm_s[4];
    __shared__ unsigned sm_c[4];
    const int64_t row = blockIdx.x;
    const float* wr = w + row * ld_w;
    float s = 0.0f; unsigned c = 0;
    for (int64_t i = threadIdx.x; i < k; i += 256) s += fabsf(wr[i]);
    block_sum(s, c, sm_s, sm_c);
    float mval = s / (float)k;
    for (int it = 0; it < iters; ++it) {
        s = 0.0f; c = 0;
        const float th = -mval / 2.0f;
        for (int64_t i = threadIdx.x; i < k; i += 256) {
            const float a = fabsf(wr[i]);
            const bool in = (a - mval) > th;
            s += in ? a : 0.0f;
            c += in ? 1u : 0u;
        }
        block_sum(s, c, sm_s, sm_c);
        mval = s / (float)c;  // 0/0 -> NaN like the reference
    }
    if (threadIdx.x == 0) scale[row] = mval;
    if (latent) {
        const float inv = 1.0f / mval;
        float* lr = latent + row * ld_latent;
        for (int64_t i = threadIdx.x; i < k; i += 256) {
            const float v = wr[i] * inv;
            lr[i] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
        }
    }
}

// ---- whole-matrix passes: per-block partials, then a one-block finish --------------------------
// workspace layout: double state[4] | Acc2 partial[FIT_BLOCKS]
constexpr int FIT_BLOCKS = 1024;

// MODE 0: sum|w| ; MODE 1: masked sum/count against state[0] = m ; MODE 2: positive sum (s) and negative sum (c as double bits)
template <int MODE>
__global__ void __launch_bounds__(256) fit_partial_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                          const double* __restrict__ state, Acc2* __restrict__ partial) {
    __shared__ float sm_s[4];
    __shared__ unsigned sm_c[4];
    __shared__ float sm_s2[4];
    const int64_t total = n * k;
    const float mval = MODE == 1 ? (float)state[0] : 0.0f;
    const float th = -mval / 2.0f;
    float s = 0.0f, s2 = 0.0f; unsigned c = 0, c2 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / k, col = i - r * k;
        const float v = w[r * ld_w + col];
        if constexpr (MODE == 0) {
            s += fabsf(v);
        } else if constexpr (MODE == 1) {
            const float a = fabsf(v);
            const bool in = (a - mval) > th;
            s += in ? a : 0.0f;
            c += in ? 1u : 0u;
        } else {
            s += v > 0.0f ? v : 0.0f;
            s2 += v < 0.0f ? v : 0.0f;
        }
    }
    block_sum(s, c, sm_s, sm_c);
    if constexpr (MODE == 2) block_sum(s2, c2, sm_s2, sm_c);
    if (threadIdx.x == 0) {
        partial[blockIdx.x].s = (double)s;
        if constexpr (MODE == 2) partial[blockIdx.x].c = (unsigned long long)__double_as_longlong((double)s2);
        else partial[blockIdx.x].c = c;
    }
}

// FIN 0: state[0] = sum / total ; FIN 1: state[0] = sum / count ; FIN 2: state[1] = pos sum, state[2] = neg sum
template <int FIN>
__global__ void __launch_bounds__(64) fit_finish_kernel(const Acc2* __restrict__ partial, int nparts, double total, double* __restrict__ state) {
    if (threadIdx.x != 0) return;
    double s = 0.0, s2 = 0.0; unsigned long long c = 0;
    for (int i = 0; i < nparts; ++i) {
        s += partial[i].s;
        if constexpr (FIN == 2) s2 += __longlong_as_double((long long)partial[i].c);
        else c += partial[i].c;
    }
    if constexpr (FIN == 0) state[0] = (double)((float)s / (float)total);
    else if constexpr (FIN == 1) state[0] = (double)((float)s / (float)c);
    else { state[1] = s; state[2] = s2; }
}

__global__ void __launch_bounds__(256) fit_broadcast_kernel(const double* __restrict__ state, int64_t n, float* __restrict__ scale) {
    const float mval = (float)state[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) scale[i] = mval;
}

// latent = clip(w * f, -1, 1) with f = 1/state[0] (PER_ROW 0), 1/scale[row] (PER_ROW 1) or a constant (PER_ROW 2)
template <int PER_ROW>
__global__ void __launch_bounds__(256) latent_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                     const double* __restrict__ state, const float* __restrict__ scale, float f,
                                                     float* __restrict__ latent, int64_t ld_latent) {
    const int64_t total = n * k;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / k, col = i - r * k;
        float mul;
        if constexpr (PER_ROW == 0) mul = 1.0f / (float)state[0];
        else if constexpr (PER_ROW == 1) mul = 1.0f / scale[r];
        else mul = f;
        const float v = w[r * ld_w + col] * mul;
        latent[r * ld_latent + col] = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
    }
}

// ---- per-row binary sums ------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256) fit_binary_rows_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                              float* __restrict__ scale) {
    __shared__ float sm_s[4];
    __shared__ unsigned sm_c[4];
    __shared__ float sm_s2[4];
    const int64_t row = blockIdx.x;
    const float* wr = w + row * ld_w;
    float sp = 0.0f, sn = 0.0f; unsigned c = 0, c2 = 0;
    for (int64_t i = threadIdx.x; i < k; i += 256) {
        const float v = wr[i];
        sp += v > 0.0f ? v : 0.0f;
        sn += v < 0.0f ? v : 0.0f;
    }
    block_sum(sp, c, sm_s, sm_c);
    block_sum(sn, c2, sm_s2, sm_c);
    if (threadIdx.x == 0) scale[row] = (MODE == TVL_Q_SIGN ? (sp - sn) : sp) / (float)k;
}

template <int MODE>
__global__ void __launch_bounds__(256) fit_binary_uniform_finish_kernel(const double* __restrict__ state, int64_t n, double total,
                                                                        float* __restrict__ scale) {
    const float sp = (float)state[1], sn = (float)state[2];
    const float v = (MODE == TVL_Q_SIGN ? (sp - sn) : sp) / (float)total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) scale[i] = v;
}

size_t fit_workspace_bytes() { return 4 * sizeof(double) + FIT_BLOCKS * sizeof(Acc2); }

static inline int ew_grid(int64_t total) {
    int64_t g = ceil_div(total, 256);
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int launch_fit_ternary(const float* w, int64_t n, int64_t k, int64_t ld_w, int uniform, int iters, float* scale, float* latent,
                       int64_t ld_latent, void* workspace, hipStream_t st) {
    if (!uniform) {
        hipLaunchKernelGGL(fit_ternary_rows_kernel, dim3((unsigned)n), dim3(256), 0, st, w, n, k, ld_w, iters, scale, latent, ld_latent);
        TVL_LAUNCH_CHECK();
        return TVL_OK;
    }
    double* state = reinterpret_cast<double*>(workspace);
    Acc2* partial = reinterpret_cast<Acc2*>(state + 4);
    const int64_t total = n * k;
    int nb = (int)ceil_div(total, 256 * 8);
    nb = nb < 1 ? 1 : (nb > FIT_BLOCKS ? FIT_BLOCKS : nb);
    hipLaunchKernelGGL(fit_partial_kernel<0>, dim3(nb), dim3(256), 0, st, w, n, k, ld_w, state, partial);
    hipLaunchKernelGGL(fit_finish_kernel<0>, dim3(1), dim3(64), 0, st, partial, nb, (double)total, state);
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(fit_partial_kernel<1>, dim3(nb), dim3(256), 0, st, w, n, k, ld_w, state, partial);
        hipLaunchKernelGGL(fit_finish_kernel<1>, dim3(1), dim3(64), 0, st, partial, nb, (double)total, state);
    }
    hipLaunchKernelGGL(fit_broadcast_kernel, dim3(ew_grid(n)), dim3(256), 0, st, state, n, scale);
    if (latent)
        hipLaunchKernelGGL(latent_kernel<0>, dim3(ew_grid(total)), dim3(256), 0, st, w, n, k, ld_w, state, scale, 0.0f, latent, ld_latent);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_fit_binary(const float* w, int64_t n, int64_t k, int64_t ld_w, int mode, int uniform, float prescale, int natural,
                      float* scale, float* latent, int64_t ld_latent, void* workspace, hipStream_t st) {
    const int64_t total = n * k;
    if (!uniform) {
        if (mode == TVL_Q_SIGN) hipLaunchKernelGGL(fit_binary_rows_kernel<TVL_Q_SIGN>, dim3((unsigned)n), dim3(256), 0, st, w, n, k, ld_w, scale);
        else hipLaunchKernelGGL(fit_binary_rows_kernel<TVL_Q_STEP>, dim3((unsigned)n), dim3(256), 0, st, w, n, k, ld_w, scale);
    } else {
        double* state = reinterpret_cast<double*>(workspace);
        Acc2* partial = reinterpret_cast<Acc2*>(state + 4);
        int nb = (int)ceil_div(total, 256 * 8);
        nb = nb < 1 ? 1 : (nb > FIT_BLOCKS ? FIT_BLOCKS : nb);
        hipLaunchKernelGGL(fit_partial_kernel<2>, dim3(nb), dim3(256), 0, st, w, n, k, ld_w, state, partial);
        hipLaunchKernelGGL(fit_finish_kernel<2>, dim3(1), dim3(64), 0, st, partial, nb, (double)total, state);
        if (mode == TVL_Q_SIGN) hipLaunchKernelGGL(fit_binary_uniform_finish_kernel<TVL_Q_SIGN>, dim3(ew_grid(n)), dim3(256), 0, st, state, n, (double)total, scale);
        else hipLaunchKernelGGL(fit_binary_uniform_finish_kernel<TVL_Q_STEP>, dim3(ew_grid(n)), dim3(256), 0, st, state, n, (double)total, scale);
    }
    if (latent) {
        if (natural) hipLaunchKernelGGL(latent_kernel<1>, dim3(ew_grid(total)), dim3(256), 0, st, w, n, k, ld_w, (const double*)nullptr, scale, 0.0f, latent, ld_latent);
        else hipLaunchKernelGGL(latent_kernel<2>, dim3(ew_grid(total)), dim3(256), 0, st, w, n, k, ld_w, (const double*)nullptr, scale, prescale, latent, ld_latent);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

}  // namespace tvl
