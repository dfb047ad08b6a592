// Packed-ternary x f16/bf16 matmul for few activation rows (m <= 32: decode / GEMV regime).
//
// This regime is bound by streaming the 2-bit weights from HBM once (at m = 1, n = k = 4096 the packed
// weights are 98.8 % of the algorithmic bytes), so the kernel is organised around the weight stream:
//  * a workgroup (4 waves) owns 128 output channels, one 32-channel tile per wave, and a range of k;
//  * per super-step of BKS (512 or 256) k values every lane loads BKS/8 contiguous bytes of ITS OWN output
//    channel's packed row (half a 128-byte line) straight into registers - no LDS for the weights - and
//    keeps the next super-step's bytes in flight while it decodes the current ones (2 bits -> f16 with
//    v_perm / v_and_or / v_pk_fma, 10 VALU per 8 weights);
//  * the few activation rows (padded to the 32 columns of a 32x32x16 MFMA) are staged global -> LDS once
//    per workgroup and super-step by LDS-DMA (one 1-KiB piece per row), XOR-swizzled on the source
//    address so the per-lane 16-byte fragment reads are conflict-free; rows >= m are not loaded at all
//    (an accumulator column only depends on its own activation row, so garbage there is never stored);
//  * the matrix cores do the multiply-accumulate (weights = operand A), so one decode serves all rows;
//  * k is split over blockIdx.y (float32 slabs + the deterministic reduce kernel of gemm_mfma.hip) so
//    that every CU has several workgroups' weight lines in flight.
// Requires m <= 32, k % 256 == 0, 16-byte aligned activation rows, 16-byte aligned packed rows.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <type_traits>

namespace tvl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

struct SkinnyConsts { uint32_t mask01, mask23, magic; };

template <int BYTE>
__device__ __forceinline__ void sk_decode4(uint32_t d, const SkinnyConsts& c, uint32_t& r01, uint32_t& r23) {
    const uint32_t rep = __builtin_amdgcn_perm(d, d, 0x0C000C00u | BYTE | (BYTE << 16));
    const uint32_t v01 = (rep & c.mask01) | c.magic;
    const uint32_t v23 = (rep & c.mask23) | c.magic;
    const half2v m01 = {(_Float16)1.0f, (_Float16)0.25f}, a01 = {(_Float16)-1025.0f, (_Float16)-257.0f};
    const half2v m23 = {(_Float16)0.0625f, (_Float16)0.015625f}, a23 = {(_Float16)-65.0f, (_Float16)-17.0f};
    r01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v01), m01, a01));
    r23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v23), m23, a23));
}

template <typename TX, int HW>
__device__ __forceinline__ uint4 sk_decode8(uint32_t d, const SkinnyConsts& c) {
    uint4 r;
    sk_decode4<2 * HW>(d, c, r.x, r.y);
    sk_decode4<2 * HW + 1>(d, c, r.z, r.w);
    if constexpr (!__is_same(TX, f16)) {  // f16 bit patterns of {-1,0,1,2} -> bf16 patterns
        r.x |= (r.x >> 3) & 0x03800380u; r.y |= (r.y >> 3) & 0x03800380u;
        r.z |= (r.z >> 3) & 0x03800380u; r.w |= (r.w >> 3) & 0x03800380u;
    }
    return r;
}

template <typename TX>
__device__ __forceinline__ floatx16 sk_mfma(uint4 a, uint4 b, floatx16 c) {
    if constexpr (__is_same(TX, f16))
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// SPLIT: raw float32 partial sums to slab blockIdx.y of y (float [splits][m][ld_y])
template <typename TX, typename TY, int BKS, int ROWS, bool HAS_BIAS, int ACT, bool SPLIT>
__global__ void __launch_bounds__(256, ROWS <= 8 ? 4 : 2)
gemm_skinny_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                   const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act_begin, TY* __restrict__ y,
                   int64_t ld_y, int ss_per_split) {
    constexpr int ROWB = BKS * 2;          // bytes of one activation row in the LDS image
    constexpr int CPR = BKS / 8;           // 16-byte chunks per row
    constexpr int TILE_BYTES = ROWS * ROWB;  // ROWS activation rows (m <= ROWS); fragment reads of rows >= ROWS wrap around
    constexpr int NQ = BKS / 128;          // 16-byte registers of packed weights per lane and super-step
    constexpr int NMMA = BKS / 16;         // MFMAs per super-step
    constexpr int PIECES = ROWS * CPR / 64;  // LDS-DMA wave-instructions per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 * TILE_BYTES

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 128 + wid * 32;
    const int total_ss = k / BKS;
    const int ss0 = SPLIT ? (int)blockIdx.y * ss_per_split : 0;
    const int nss = SPLIT ? (total_ss - ss0 < ss_per_split ? total_ss - ss0 : ss_per_split) : total_ss;
    if constexpr (SPLIT) y += (int64_t)blockIdx.y * m * ld_y;

    SkinnyConsts dc;
    asm volatile("s_mov_b32 %0, 0x000C0003" : "=s"(dc.mask01));
    asm volatile("s_mov_b32 %0, 0x00C00030" : "=s"(dc.mask23));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(dc.magic));

    // ---- activation staging: piece p (1 KiB = 64 lanes x 16 B) covers rows [p*64/CPR, ...) -------------
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
    auto stage_a = [&](int ss, int buf) {
#pragma unroll
        for (int j = 0; j < (PIECES + 3) / 4; ++j) {
            const int p = j * 4 + wid;                       // wave-uniform piece index
            if (PIECES % 4 != 0 && p >= PIECES) break;
            const int q = p * 64 + lane;                     // 16-byte slot in the image
            const int row = q / CPR, cs = q % CPR;
            if (p * 64 / CPR < m) {                          // wave-uniform: skip pieces made only of unused rows
                const int chunk = cs ^ (row & 15);
                const int gm = row < m ? row : m - 1;
                const TX* src = x + (int64_t)gm * ld_x + (int64_t)ss * BKS + chunk * 8;
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * TILE_BYTES + p * 1024);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
            }
        }
    };

    // ---- packed weights: lane (r, h) streams bytes [ss*BKS/4 + h*BKS/8, +BKS/8) of channel n0 + r --------
    int gn = n0 + r;
    gn = gn < n ? gn : n - 1;
    const uint8_t* b_row = pw + (int64_t)gn * ld_pw + h * (BKS / 8);
    auto load_b = [&](int ss, u32x4 (&dst)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            // "+v": the destination continues the live range of the previous contents, which keeps the
            // register allocator from renaming (= copying, before the data has landed) at the loop header
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst[q]) : "v"(b_row + (int64_t)ss * (BKS / 4) + q * 16) : "memory");
    };
    auto wait_loads = [&](u32x4 (&q)[NQ]) {
        if constexpr (NQ == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : : "memory");
    };

    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    const int a_row_off = (r % ROWS) * ROWB;  // lanes of unused columns (r >= m) read some valid row: never stored
    const int sw = (r % ROWS) & 15;
    // MFMA j of a super-step: lane half h multiplies weights [h*BKS/2 + 8j, +8) with the same k of x
    auto compute = [&](int buf, const u32x4 (&bq)[NQ]) {
        const unsigned char* abase = lds + buf * TILE_BYTES + a_row_off;
#pragma unroll
        for (int j = 0; j < NMMA; ++j) {
            const int chunk = (h * (CPR / 2) + j) ^ sw;
            const uint4 af = *reinterpret_cast<const uint4*>(abase + chunk * 16);
            const uint32_t d = bq[j / 8][(j / 2) & 3];
            const uint4 bf = (j & 1) ? sk_decode8<TX, 1>(d, dc) : sk_decode8<TX, 0>(d, dc);
            acc = sk_mfma<TX>(bf, af, acc);
        }
    };

    // Pipeline: while super-step i is computed, the loads of super-step i+1 (LDS-DMA of x, register loads of
    // the weights) are in flight; they are waited for at the END of super-step i, i.e. before the loop's
    // back-edge.  A register whose load is still in flight must never cross the back-edge: at the loop header
    // the register allocator may insert copies, and a copy made before the data has landed copies garbage
    // (the compiler does not know about loads issued from inline asm).
    u32x4 bqA[NQ], bqB[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bqA[q] = bqB[q] = (u32x4){0, 0, 0, 0};
    stage_a(ss0, 0);
    load_b(ss0, bqA);
    wait_loads(bqA);
    for (int i = 0; i < nss; i += 2) {
        // ---- even super-step: data in buffer 0 / bqA -----------------------------------------------
        __builtin_amdgcn_s_barrier();  // tile i visible to all waves; buffer 1 no longer read by anyone
        {
            const int nx = i + 1 < nss ? ss0 + i + 1 : ss0 + nss - 1;  // past the end: harmless re-fetch
            stage_a(nx, 1);
            load_b(nx, bqB);
        }
        compute(0, bqA);
        wait_loads(bqB);
        if (i + 1 >= nss) break;
        // ---- odd super-step: buffer 1 / bqB ----------------------------------------------------------
        __builtin_amdgcn_s_barrier();
        {
            const int nx = i + 2 < nss ? ss0 + i + 2 : ss0 + nss - 1;
            stage_a(nx, 0);
            load_b(nx, bqA);
        }
        compute(1, bqB);
        wait_loads(bqA);
    }
    // every load has been waited for: no LDS-DMA outlives the LDS allocation, no register load is pending

    // ---- epilogue: lane owns activation row m = r; register e -> channel n0 + (e&3) + 8(e>>2) + 4h --------
    const float c = fabsf(clip);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(ld_y * (int64_t)sizeof(TY))) & 15) == 0;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int cn = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if constexpr (SPLIT) v[e] = acc[e];
        else v[e] = epilogue_ch<HAS_BIAS, ACT>(acc[e], cn < n ? scale[cn] : 0.0f, (HAS_BIAS && cn < n) ? bias[cn] : 0.0f, c, cn, act_begin);
    }
    if (r < m) {
        TY* yrow = y + (int64_t)r * ld_y;
        if (sizeof(TY) == 4 && vec_ok && n0 + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(yrow) + n0 + 8 * q + 4 * h) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        } else if (sizeof(TY) == 2 && vec_ok && n0 + 32 <= n) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                TY tmp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tmp[j] = from_float<TY>(v[4 * q + j]);
                *reinterpret_cast<uint2*>(yrow + n0 + 8 * q + 4 * h) = *reinterpret_cast<const uint2*>(tmp);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int cn = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (cn < n) yrow[cn] = from_float<TY>(v[e]);
            }
        }
    }
}

bool gemm_skinny_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw) {
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16) return false;
    if (m < 1 || m > 32 || k <= 0 || k % 256 != 0) return false;
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    if (!aligned(pw, 16) || ld_pw % 16 != 0) return false;
    if (n > INT32_MAX - 256 || k > INT32_MAX) return false;
    return true;
}

struct SkinnyPlan { int bks, splits; };
static SkinnyPlan plan_skinny(int64_t m, int64_t n, int64_t k, size_t workspace_bytes) {
    const int64_t nblk = ceil_div(n, 128);
    SkinnyPlan p;
    p.bks = (k % 512 == 0 && nblk * (k / 512) >= 512) ? 512 : 256;  // shorter super-steps when blocks are scarce
    const int64_t nss = k / p.bks;
    int64_t s = ceil_div(m <= 8 ? 2048 : 1024, nblk);               // several workgroups per CU (8 / 4 resident)
    const int64_t by_bytes = k / (16 * m) > 0 ? k / (16 * m) : 1;   // slab bytes <= packed weight bytes
    const int64_t by_ws = (int64_t)(workspace_bytes / ((size_t)m * n * sizeof(float)));
    s = s < nss ? s : nss;
    s = s < by_bytes ? s : by_bytes;
    s = s < by_ws ? s : by_ws;
    p.splits = (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
    return p;
}

size_t gemm_skinny_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    const SkinnyPlan p = plan_skinny(m, n, k, (size_t)1 << 40);
    return p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
}

template <typename TX, typename TY, int BKS, int ROWS>
static int launch_skinny_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                             const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, int splits, float* slab,
                             hipStream_t st) {
    const int nblk = (int)ceil_div(n, 128);
    const size_t lds_bytes = 2 * ROWS * BKS * 2;
    const dim3 block(256);
    if (splits > 1) {
        const int total_ss = (int)(k / BKS);
        const int per = (int)ceil_div(total_ss, splits);
        splits = (int)ceil_div(total_ss, per);
        auto kern = gemm_skinny_kernel<TX, float, BKS, ROWS, false, TVL_ACT_NONE, true>;
        if (int rc_ = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc_ != TVL_OK) return rc_;
        hipLaunchKernelGGL(kern, dim3(nblk, splits), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip,
                           0, slab, n, per);
        TVL_LAUNCH_CHECK();
        return launch_splitk_reduce(slab, splits, m, n, scale, bias, clip, act, act_begin, y, std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16), ld_y, st);
    }
#define TVL_SK(HB, ACT)                                                                                            \
    do {                                                                                                           \
        auto kern = gemm_skinny_kernel<TX, TY, BKS, ROWS, HB, ACT, false>;                                                \
        if (int rc_ = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc_ != TVL_OK) return rc_; \
        hipLaunchKernelGGL(kern, dim3(nblk, 1), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act_begin, y, ld_y, 0); \
    } while (0)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_SK(true, TVL_ACT_GELU_TANH); else TVL_SK(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_SK(false, TVL_ACT_GELU_TANH); else TVL_SK(false, TVL_ACT_NONE);
    }
#undef TVL_SK
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX, typename TY>
static int launch_skinny_xy(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                            const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, void* ws, size_t ws_bytes,
                            hipStream_t st) {
    const bool ws_ok = ws != nullptr && aligned(ws, 16);
    const SkinnyPlan p = plan_skinny(m, n, k, ws_ok ? ws_bytes : 0);
    if (m <= 8) {
        if (p.bks == 512)
            return launch_skinny_cfg<TX, TY, 512, 8>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, (float*)ws, st);
        return launch_skinny_cfg<TX, TY, 256, 8>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, (float*)ws, st);
    }
    if (p.bks == 512)
        return launch_skinny_cfg<TX, TY, 512, 32>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, (float*)ws, st);
    return launch_skinny_cfg<TX, TY, 256, 32>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p.splits, (float*)ws, st);
}

template <typename TX>
static int launch_skinny_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                           const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, void* ws,
                           size_t ws_bytes, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_skinny_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, ws, ws_bytes, st);
        case TVL_F16: return launch_skinny_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, ws, ws_bytes, st);
        case TVL_BF16: return launch_skinny_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, ws, ws_bytes, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

int launch_gemm_skinny(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                       const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    if (x_dtype == TVL_F16)
        return launch_skinny_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    return launch_skinny_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
}

}  // namespace tvl
