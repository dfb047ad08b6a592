// One-time scale fits of a dense weight matrix (binary/modules.py:189-299).
//
// Ternary: masked-mean fixed point  m0 = mean|w|;  repeat: mask = (|w| - m) > -m/2,
// m = sum(|w| mask) / count(mask)   (modules.py:284-288), per output row or for the whole matrix.
// Binary:  (sum_{w>0} w - sum_{w<0} w) / count  or  sum_{w>0} w / count  (modules.py:197-210, 236-245).
// latent = clip(w * (1 / m), -1, 1)  (modules.py:290-292: the reference multiplies by the reciprocal).
// Sums are float32 per lane, combined in a fixed order (deterministic); whole-matrix sums finish in float64.
//
// Per-row ternary fit: ONE WAVE per row with the row held in registers (rows up to 8192 elements): the matrix is
// read from HBM once, the ten iterations run on registers with wave-level DPP reductions (no barrier), and the
// latent weights are written from the same registers.  Whole-matrix fit: one launch per pass, 16 bytes per lane and
// load, no per-element index arithmetic; every block of pass i starts by summing pass i-1's per-block partial sums
// itself (same order in every block, so all blocks see the same m) - no single-thread "finish" launches between
// passes; the matrix (64 MiB at 4096^2) stays in the Infinity Cache between passes.
#include "tvl_common.h"
#include "tvl_kernels.h"

namespace tvl {

typedef float fit_f4 __attribute__((ext_vector_type(4)));

struct Acc2 {
    double s;              // sum
    unsigned long long c;  // count (or a second sum's bits, see binary fit)
};

// ---- reductions in a fixed order --------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int fit_dpp(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }

__device__ __forceinline__ float wave_sum(float x) {   // valid in every lane
    x += __builtin_bit_cast(float, fit_dpp<0x111>(__builtin_bit_cast(int, x)));   // row_shr:1
    x += __builtin_bit_cast(float, fit_dpp<0x112>(__builtin_bit_cast(int, x)));   // row_shr:2
    x += __builtin_bit_cast(float, fit_dpp<0x114>(__builtin_bit_cast(int, x)));   // row_shr:4
    x += __builtin_bit_cast(float, fit_dpp<0x118>(__builtin_bit_cast(int, x)));   // row_shr:8 -> lane 15 of each row of 16
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 15)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 31));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 47)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 63));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ unsigned wave_sum_u(unsigned x) {
    x += (unsigned)fit_dpp<0x111>((int)x);
    x += (unsigned)fit_dpp<0x112>((int)x);
    x += (unsigned)fit_dpp<0x114>((int)x);
    x += (unsigned)fit_dpp<0x118>((int)x);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 15) + (unsigned)__builtin_amdgcn_readlane((int)x, 31) +
           (unsigned)__builtin_amdgcn_readlane((int)x, 47) + (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// block-wide (256 threads) sum of a (float, unsigned) pair; result valid in every thread
__device__ __forceinline__ void block_sum(float& s, unsigned& c, float* sm_s, unsigned* sm_c) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    s = wave_sum(s);
    c = wave_sum_u(c);
    __syncthreads();  // protect sm_* from the previous use
    if (lane == 0) { sm_s[wid] = s; sm_c[wid] = c; }
    __syncthreads();
    s = (sm_s[0] + sm_s[1]) + (sm_s[2] + sm_s[3]);
    c = (sm_c[0] + sm_c[1]) + (sm_c[2] + sm_c[3]);
}

// block-wide sum of (double, u64) pairs over 256 threads, fixed order; result valid in every thread
__device__ __forceinline__ void block_sum_d(double& s, unsigned long long& c, double* sm_s, unsigned long long* sm_c) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        c += __shfl_down(c, o, 64);
    }
    __syncthreads();
    if (lane == 0) { sm_s[wid] = s; sm_c[wid] = c; }
    __syncthreads();
    s = (sm_s[0] + sm_s[1]) + (sm_s[2] + sm_s[3]);
    c = (sm_c[0] + sm_c[1]) + (sm_c[2] + sm_c[3]);
}

__device__ __forceinline__ float clip1(float v) { return v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v); }   // NaN stays NaN

// ---- per-row ternary fit, row in registers: one wave per row, NV float4 per lane (k <= 256 NV, k % 4 == 0) ---------------
template <int NV>
__global__ void __launch_bounds__(256) fit_ternary_rows_reg_kernel(const float* __restrict__ w, int64_t n, int k, int64_t ld_w, int iters,
                                                                   float* __restrict__ scale, float* __restrict__ latent, int64_t ld_latent) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (row >= n) return;
    const float* wr = w + row * ld_w;
    fit_f4 v[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = (j * 64 + lane) * 4;
        v[j] = idx < k ? __builtin_nontemporal_load(reinterpret_cast<const fit_f4*>(wr + idx)) : (fit_f4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += fabsf(v[j][e]);
    float mval = wave_sum(s) / (float)k;
    for (int it = 0; it < iters; ++it) {
        const float th = -mval / 2.0f;
        s = 0.0f;
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // elements beyond the row are 0: (0 - m) > -m/2 is false for every m >= 0 and for NaN
                const float a = fabsf(v[j][e]);
                const bool in = (a - mval) > th;
                s += in ? a : 0.0f;
                c += in ? 1u : 0u;
            }
        mval = wave_sum(s) / (float)wave_sum_u(c);   // 0/0 -> NaN like the reference
    }
    if (lane == 0) scale[row] = mval;
    if (latent) {
        const float inv = 1.0f / mval;
        float* lr = latent + row * ld_latent;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int idx = (j * 64 + lane) * 4;
            if (idx < k)
                *reinterpret_cast<fit_f4*>(lr + idx) = (fit_f4){clip1(v[j][0] * inv), clip1(v[j][1] * inv), clip1(v[j][2] * inv), clip1(v[j][3] * inv)};
        }
    }
}

// ---- per-row ternary fit, any row length / alignment: one block per row, the row is re-read (cache resident) each pass ----
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
        for (int64_t i = threadIdx.x; i < k; i += 256) lr[i] = clip1(wr[i] * inv);
    }
}

// ---- whole-matrix passes ------------------------------------------------------------------------------------------------
// A block walks (row, 1024-element chunk) items; VEC: 16 bytes per lane (k % 4 == 0, aligned rows).
// workspace layout: Acc2 partial[2][FIT_BLOCKS] (ping-pong: pass i reads the partial sums of pass i-1 while writing its own)
constexpr int FIT_BLOCKS = 1024;
constexpr int FIT_CHUNK = 1024;

enum { PREV_NONE = 0, PREV_MEAN = 1, PREV_MASKED = 2 };

// m from the previous pass's partial sums: sum / total (PREV_MEAN) or sum / count (PREV_MASKED); same value in every block
__device__ __forceinline__ float m_from_partials(const Acc2* __restrict__ prev, int nprev, int prev_kind, double total, double* sm_d,
                                                 unsigned long long* sm_u) {
    double ps = 0.0;
    unsigned long long pc = 0;
    for (int i = threadIdx.x; i < nprev; i += 256) { ps += prev[i].s; pc += prev[i].c; }
    block_sum_d(ps, pc, sm_d, sm_u);
    return prev_kind == PREV_MEAN ? (float)ps / (float)total : (float)ps / (float)pc;
}

// MODE 0: sum|w| ; MODE 1: masked sum/count against m ; MODE 2: positive sum (s) and negative sum (c as double bits)
template <int MODE, bool VEC>
__global__ void __launch_bounds__(256) fit_pass_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                       const Acc2* __restrict__ prev, int nprev, int prev_kind, double total,
                                                       Acc2* __restrict__ partial) {
    __shared__ float sm_s[4];
    __shared__ unsigned sm_c[4];
    __shared__ float sm_s2[4];
    __shared__ double sm_d[4];
    __shared__ unsigned long long sm_u[4];
    float mval = 0.0f;
    if constexpr (MODE == 1) mval = m_from_partials(prev, nprev, prev_kind, total, sm_d, sm_u);
    const float th = -mval / 2.0f;
    float s = 0.0f, s2 = 0.0f; unsigned c = 0, c2 = 0;
    auto take = [&](float v) __attribute__((always_inline)) {
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
    };
    const int64_t cpr = (k + FIT_CHUNK - 1) / FIT_CHUNK, items = n * cpr;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t row = item / cpr, c0 = (item - row * cpr) * FIT_CHUNK + 4 * threadIdx.x;
        const float* p = w + row * ld_w + c0;
        if constexpr (VEC) {
            if (c0 < k) {
                const fit_f4 v = *reinterpret_cast<const fit_f4*>(p);
                take(v[0]); take(v[1]); take(v[2]); take(v[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < k) take(p[e]);
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

// one block: the final value from the last pass's partial sums, written to every scale[i]
// KIND PREV_MEAN / PREV_MASKED: ternary m ; KIND 3 / 4: binary (pos - neg) / total, binary01 pos / total
__global__ void __launch_bounds__(256) fit_final_kernel(const Acc2* __restrict__ prev, int nprev, int kind, double total, int64_t n,
                                                        float* __restrict__ scale) {
    __shared__ double sm_d[4];
    __shared__ unsigned long long sm_u[4];
    float v;
    if (kind == PREV_MEAN || kind == PREV_MASKED) {
        v = m_from_partials(prev, nprev, kind, total, sm_d, sm_u);
    } else {
        double sp = 0.0, sn = 0.0;
        unsigned long long zero = 0, zero2 = 0;
        for (int i = threadIdx.x; i < nprev; i += 256) { sp += prev[i].s; sn += __longlong_as_double((long long)prev[i].c); }
        block_sum_d(sp, zero, sm_d, sm_u);
        block_sum_d(sn, zero2, sm_d, sm_u);
        v = (kind == 3 ? ((float)sp - (float)sn) : (float)sp) / (float)total;
    }
    for (int64_t i = threadIdx.x; i < n; i += 256) scale[i] = v;
}

// latent = clip(w * f, -1, 1) with f = 1 / scale[row] (PER_ROW) or a constant
template <bool PER_ROW, bool VEC>
__global__ void __launch_bounds__(256) latent_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                     const float* __restrict__ scale, float f, float* __restrict__ latent, int64_t ld_latent) {
    const int64_t cpr = (k + FIT_CHUNK - 1) / FIT_CHUNK, items = n * cpr;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t row = item / cpr, c0 = (item - row * cpr) * FIT_CHUNK + 4 * threadIdx.x;
        const float mul = PER_ROW ? 1.0f / scale[row] : f;
        const float* p = w + row * ld_w + c0;
        float* q = latent + row * ld_latent + c0;
        if constexpr (VEC) {
            if (c0 < k) {
                const fit_f4 v = *reinterpret_cast<const fit_f4*>(p);
                *reinterpret_cast<fit_f4*>(q) = (fit_f4){clip1(v[0] * mul), clip1(v[1] * mul), clip1(v[2] * mul), clip1(v[3] * mul)};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < k) q[e] = clip1(p[e] * mul);
        }
    }
}

// ---- per-row binary sums: one wave per row ------------------------------------------------------------------------------
template <int MODE, bool VEC>
__global__ void __launch_bounds__(256) fit_binary_rows_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ld_w,
                                                              float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave-uniform
    if (row >= n) return;
    const float* wr = w + row * ld_w;
    float sp = 0.0f, sn = 0.0f;
    if constexpr (VEC) {
        for (int64_t i = 4 * lane; i < k; i += 256) {
            const fit_f4 v = *reinterpret_cast<const fit_f4*>(wr + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sp += v[e] > 0.0f ? v[e] : 0.0f; sn += v[e] < 0.0f ? v[e] : 0.0f; }
        }
    } else {
        for (int64_t i = lane; i < k; i += 64) {
            const float v = wr[i];
            sp += v > 0.0f ? v : 0.0f;
            sn += v < 0.0f ? v : 0.0f;
        }
    }
    sp = wave_sum(sp);
    sn = wave_sum(sn);
    if (lane == 0) scale[row] = (MODE == TVL_Q_SIGN ? (sp - sn) : sp) / (float)k;
}

size_t fit_workspace_bytes() { return 2 * FIT_BLOCKS * sizeof(Acc2); }

static inline bool vec_ok(const float* p, int64_t k, int64_t ld) { return aligned(p, 16) && k % 4 == 0 && ld % 4 == 0; }

static inline int pass_grid(int64_t n, int64_t k) {
    const int64_t items = n * ceil_div(k, (int64_t)FIT_CHUNK);
    return (int)(items < 1 ? 1 : (items > FIT_BLOCKS ? FIT_BLOCKS : items));
}
static inline int latent_grid(int64_t n, int64_t k) {
    const int64_t items = n * ceil_div(k, (int64_t)FIT_CHUNK);
    return (int)(items < 1 ? 1 : (items > 4096 ? 4096 : items));
}

template <bool PER_ROW>
static void launch_latent(const float* w, int64_t n, int64_t k, int64_t ld_w, const float* scale, float f, float* latent, int64_t ld_latent,
                          hipStream_t st) {
    const dim3 grid(latent_grid(n, k)), block(256);
    if (vec_ok(w, k, ld_w) && vec_ok(latent, k, ld_latent))
        hipLaunchKernelGGL((latent_kernel<PER_ROW, true>), grid, block, 0, st, w, n, k, ld_w, scale, f, latent, ld_latent);
    else
        hipLaunchKernelGGL((latent_kernel<PER_ROW, false>), grid, block, 0, st, w, n, k, ld_w, scale, f, latent, ld_latent);
}

template <int MODE>
static void launch_pass(const float* w, int64_t n, int64_t k, int64_t ld_w, const Acc2* prev, int nprev, int prev_kind, double total,
                        Acc2* partial, int nb, hipStream_t st) {
    if (vec_ok(w, k, ld_w))
        hipLaunchKernelGGL((fit_pass_kernel<MODE, true>), dim3(nb), dim3(256), 0, st, w, n, k, ld_w, prev, nprev, prev_kind, total, partial);
    else
        hipLaunchKernelGGL((fit_pass_kernel<MODE, false>), dim3(nb), dim3(256), 0, st, w, n, k, ld_w, prev, nprev, prev_kind, total, partial);
}

int launch_fit_ternary(const float* w, int64_t n, int64_t k, int64_t ld_w, int uniform, int iters, float* scale, float* latent,
                       int64_t ld_latent, void* workspace, hipStream_t st) {
    if (!uniform) {
        const bool in_regs = k <= 8192 && vec_ok(w, k, ld_w) && (!latent || vec_ok(latent, k, ld_latent));
        const dim3 grid4((unsigned)ceil_div(n, (int64_t)4)), block(256);
#define TVL_FR(NV) hipLaunchKernelGGL((fit_ternary_rows_reg_kernel<NV>), grid4, block, 0, st, w, n, (int)k, ld_w, iters, scale, latent, ld_latent)
        if (in_regs && k <= 1024) TVL_FR(4);
        else if (in_regs && k <= 2048) TVL_FR(8);
        else if (in_regs && k <= 4096) TVL_FR(16);
        else if (in_regs) TVL_FR(32);
        else hipLaunchKernelGGL(fit_ternary_rows_kernel, dim3((unsigned)n), block, 0, st, w, n, k, ld_w, iters, scale, latent, ld_latent);
#undef TVL_FR
        TVL_LAUNCH_CHECK();
        return TVL_OK;
    }
    Acc2* part[2] = {reinterpret_cast<Acc2*>(workspace), reinterpret_cast<Acc2*>(workspace) + FIT_BLOCKS};
    const double total = (double)n * (double)k;
    const int nb = pass_grid(n, k);
    launch_pass<0>(w, n, k, ld_w, nullptr, 0, PREV_NONE, total, part[0], nb, st);
    int cur = 0, kind = PREV_MEAN;
    for (int it = 0; it < iters; ++it) {
        launch_pass<1>(w, n, k, ld_w, part[cur], nb, kind, total, part[cur ^ 1], nb, st);
        cur ^= 1;
        kind = PREV_MASKED;
    }
    hipLaunchKernelGGL(fit_final_kernel, dim3(1), dim3(256), 0, st, part[cur], nb, kind, total, n, scale);
    if (latent) launch_latent<true>(w, n, k, ld_w, scale, 0.0f, latent, ld_latent, st);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_fit_binary(const float* w, int64_t n, int64_t k, int64_t ld_w, int mode, int uniform, float prescale, int natural,
                      float* scale, float* latent, int64_t ld_latent, void* workspace, hipStream_t st) {
    if (!uniform) {
        const dim3 grid4((unsigned)ceil_div(n, (int64_t)4)), block(256);
        const bool vec = vec_ok(w, k, ld_w);
        if (mode == TVL_Q_SIGN) {
            if (vec) hipLaunchKernelGGL((fit_binary_rows_kernel<TVL_Q_SIGN, true>), grid4, block, 0, st, w, n, k, ld_w, scale);
            else hipLaunchKernelGGL((fit_binary_rows_kernel<TVL_Q_SIGN, false>), grid4, block, 0, st, w, n, k, ld_w, scale);
        } else {
            if (vec) hipLaunchKernelGGL((fit_binary_rows_kernel<TVL_Q_STEP, true>), grid4, block, 0, st, w, n, k, ld_w, scale);
            else hipLaunchKernelGGL((fit_binary_rows_kernel<TVL_Q_STEP, false>), grid4, block, 0, st, w, n, k, ld_w, scale);
        }
    } else {
        Acc2* partial = reinterpret_cast<Acc2*>(workspace);
        const double total = (double)n * (double)k;
        const int nb = pass_grid(n, k);
        launch_pass<2>(w, n, k, ld_w, nullptr, 0, PREV_NONE, total, partial, nb, st);
        hipLaunchKernelGGL(fit_final_kernel, dim3(1), dim3(256), 0, st, partial, nb, mode == TVL_Q_SIGN ? 3 : 4, total, n, scale);
    }
    if (latent) {
        if (natural) launch_latent<true>(w, n, k, ld_w, scale, 0.0f, latent, ld_latent, st);
        else launch_latent<false>(w, n, k, ld_w, scale, prescale, latent, ld_latent, st);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

}  // namespace tvl
