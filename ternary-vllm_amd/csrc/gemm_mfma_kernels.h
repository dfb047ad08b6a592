// Device code of the packed-ternary x f16/bf16 matrix-core kernels (templates; instantiated by gemm_mfma.hip and gemm_mid.hip).
//
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
#pragma once
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <type_traits>

namespace tvl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BK = 64;             // k elements per pipeline step
constexpr int ROW_BYTES = BK * 2;  // bytes of one activation row in the LDS tile
constexpr int NBUF = 3;            // activation tiles in LDS (32x32x16 kernel)
constexpr int BIG_NB = 4;          // activation tiles in LDS of the 256 x 256 kernel (128 KiB)

// ---- 2-bit -> f16 decode -------------------------------------------------------------------
// Constants live in registers the compiler cannot see through (set once per kernel by inline asm),
// so that (rep & mask) | magic becomes ONE v_and_or_b32 instead of v_and + v_or with literals.
struct DecodeConsts {
    uint32_t mask01, mask23;  // scalar registers
    uint32_t magic;           // vector register: 0x6400 in both halves (f16 1024.0)
};
__device__ __forceinline__ DecodeConsts make_decode_consts() {
    DecodeConsts c;
    asm volatile("s_mov_b32 %0, 0x000C0003" : "=s"(c.mask01));
    asm volatile("s_mov_b32 %0, 0x00C00030" : "=s"(c.mask23));
    asm volatile("v_mov_b32 %0, 0x64006400" : "=v"(c.magic));
    return c;
}

// `d` holds 16 codes; BYTE selects 4 of them (weights 4*BYTE .. 4*BYTE+3 of the dword).
// Returns two registers: (w0, w1) and (w2, w3) as packed f16.
template <int BYTE>
__device__ __forceinline__ void decode4_f16(uint32_t d, const DecodeConsts& c, uint32_t& r01, uint32_t& r23) {
    // rep = [byte, 0, byte, 0]: the selected byte in the low 8 bits of both 16-bit halves
    const uint32_t rep = __builtin_amdgcn_perm(d, d, 0x0C000C00u | BYTE | (BYTE << 16));
    const uint32_t v01 = (rep & c.mask01) | c.magic;  // lo = 1024 + c0      hi = 1024 + 4 c1
    const uint32_t v23 = (rep & c.mask23) | c.magic;  // lo = 1024 + 16 c2   hi = 1024 + 64 c3
    const half2v m01 = {(_Float16)1.0f, (_Float16)0.25f}, a01 = {(_Float16)-1025.0f, (_Float16)-257.0f};
    const half2v m23 = {(_Float16)0.0625f, (_Float16)0.015625f}, a23 = {(_Float16)-65.0f, (_Float16)-17.0f};
    const half2v o01 = __builtin_elementwise_fma(__builtin_bit_cast(half2v, v01), m01, a01);
    const half2v o23 = __builtin_elementwise_fma(__builtin_bit_cast(half2v, v23), m23, a23);
    r01 = __builtin_bit_cast(uint32_t, o01);
    r23 = __builtin_bit_cast(uint32_t, o23);
}

// f16 bit patterns of {-1,0,1,2} -> bf16 bit patterns of the same values, two per register
__device__ __forceinline__ uint32_t f16pair_to_bf16pair(uint32_t f) { return f | ((f >> 3) & 0x03800380u); }

// 8 weights (one halfword: HW = 0..1 of the dword) -> one MFMA B fragment
template <typename TX, int HW>
__device__ __forceinline__ uint4 decode8(uint32_t d, const DecodeConsts& c) {
    uint4 r;
    decode4_f16<2 * HW>(d, c, r.x, r.y);
    decode4_f16<2 * HW + 1>(d, c, r.z, r.w);
    if constexpr (sizeof(TX) == 2 && !__is_same(TX, f16)) {
        r.x = f16pair_to_bf16pair(r.x); r.y = f16pair_to_bf16pair(r.y);
        r.z = f16pair_to_bf16pair(r.z); r.w = f16pair_to_bf16pair(r.w);
    }
    return r;
}

template <typename TX>
__device__ __forceinline__ floatx16 mfma32(uint4 a, uint4 b, floatx16 c) {
    if constexpr (__is_same(TX, f16)) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// XCD-aware tile order: blocks with equal (blockIdx % 8) share an XCD (and its L2) under the
// observed round-robin placement; give each such group a contiguous run of tiles in row-major
// order so that a group streams whole activation row panels once.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return start + i;
}

// SPLIT = true: blockIdx.y selects a range of k steps; the raw float32 partial sums of that range go
// to slab blockIdx.y of `y` (float [splits][m][ld_y]) and splitk_reduce_kernel finishes the job.
template <typename TX, typename TY, int BM, int BN, int WAVES_M, int WAVES_N, bool HAS_BIAS, int ACT, bool SPLIT>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2)
gemm_mfma_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                 const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act_begin, TY* __restrict__ y,
                 int64_t ld_y, int tiles_n, int nblk, int steps_per_split, int w_steps) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;            // 32x32 MFMA tiles per wave
    constexpr int A_TILE_BYTES = BM * ROW_BYTES;
    constexpr int ROUNDS = (BM * 8) / NT;                // 16-byte pieces per thread per tile
    static_assert((BM * 8) % NT == 0, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // NBUF * A_TILE_BYTES

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wid / WAVES_N, wc = wid % WAVES_N;
    const int tile = xcd_remap(blockIdx.x, nblk);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // ---- activation staging addresses (per thread, constant over k) ----------------------
    const TX* a_src[ROUNDS];
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int p = i * NT + tid;           // 16-byte piece index inside the tile image
        const int row = p >> 3, cs = p & 7;   // LDS row, swizzled chunk
        const int chunk = cs ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < m ? gm : m - 1;
        a_src[i] = x + (int64_t)gm * ld_x + chunk * 8;
    }
    const int total_steps = k / BK;
    const int kb0 = SPLIT ? (int)blockIdx.y * steps_per_split : 0;   // first k step of this block
    const int nk = SPLIT ? (total_steps - kb0 < steps_per_split ? total_steps - kb0 : steps_per_split) : total_steps;
    if constexpr (SPLIT) {
        y += (int64_t)blockIdx.y * m * ld_y;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) a_src[i] += (int64_t)kb0 * BK;
    }
    // LDS-DMA issued from inline asm: hipcc then neither counts it nor orders the fragment reads of
    // the OTHER buffer behind it (with the builtin it drains vmcnt(0) before the first ds_read of every
    // step, which serialises load and compute).  Completion is waited for by hand: every wave's
    // s_waitcnt vmcnt(0) at the top of the next step, then the barrier.
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
    auto stage_a = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * A_TILE_BYTES + (i * NT + wid * 64) * 16);
            const TX* src = a_src[i] + (int64_t)kb * BK;
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    // ---- packed weight addresses ------------------------------------------------------------
    const uint8_t* b_src[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        int gn = n0 + wc * WN + t * 32 + r;
        gn = gn < n ? gn : n - 1;
        b_src[t] = pw + (int64_t)gn * ld_pw + 8 * h;
    }

    // ---- fragment read offsets -----------------------------------------------------------------
    const int sw = (r >> 1) & 7;
    const int a_row_off = (wr * WM + r) * ROW_BYTES;
    int a_col_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a_col_off[s] = ((4 * h + s) ^ sw) * 16;

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.0f;

    const DecodeConsts dc = make_decode_consts();

    // Pipeline (NBUF = 3 LDS tiles): at the middle of step kb every wave waits for its own pieces of
    // tile kb+1 (issued a whole step earlier), one barrier publishes them and retires all reads of
    // tile kb-1, whose buffer tile kb+2 is then loaded into.  Fragments (LDS reads and weight decode)
    // are fetched one 16-deep sub-step ahead of the MFMAs that use them, across step boundaries, so
    // the matrix pipe never drains at the barrier.
    // All loads of the loop are issued from inline asm and waited for by hand (one s_waitcnt vmcnt(0) per
    // step): hipcc's own counted waits only see the loads it emitted and would drain the LDS-DMA queue
    // early.  The wait statement names the weight registers it completes ("+v"), which orders every
    // consumer behind it.  The loop is unrolled by NBUF so that buffers and registers have static roles
    // (no rotation copies of registers whose loads are still in flight).
    uint64_t bq0[TN], bq1[TN], bq2[TN];  // packed weights (32 per lane and tile) of steps 3j, 3j+1, 3j+2
    // The packed rows hold w_steps k steps; a k range longer than that re-reads them from the start (float32
    // activations arrive as three bf16 planes side by side along k: the same weights multiply each plane).
    auto load_b = [&](int kb, uint64_t (&dst)[TN]) {
        int kw = kb0 + kb;
        kw = kw >= w_steps ? kw - w_steps : kw;
        kw = kw >= w_steps ? kw - w_steps : kw;
#pragma unroll
        for (int t = 0; t < TN; ++t)
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[t]) : "v"(b_src[t] + (int64_t)kw * (BK / 4)) : "memory");
    };
    auto wait_loads = [&](uint64_t (&q)[TN]) {
        if constexpr (TN == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]) : : "memory");
        else if constexpr (TN == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]) : : "memory");
        else static_assert(TN <= 2, "extend wait_loads");
    };
    auto read_a = [&](int buf, int s, uint4 (&af)[TM]) {
        const unsigned char* abase = lds + buf * A_TILE_BYTES + a_row_off + a_col_off[s];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(abase + i * 32 * ROW_BYTES);
    };
    auto decode_b = [&](const uint64_t (&bq)[TN], int s, uint4 (&bf)[TN]) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const uint32_t d = (s < 2) ? (uint32_t)bq[t] : (uint32_t)(bq[t] >> 32);
            bf[t] = (s & 1) ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
        }
    };
    auto mma = [&](const uint4 (&af)[TM], const uint4 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[i][t] = mfma32<TX>(bf[t], af[i], acc[i][t]);
    };

    // scheduling hint for one sub-step: each MFMA followed by up to VALU_PER_MFMA vector instructions
    auto interleave_hint = [&]() {
        constexpr int VALU_PER_MFMA = (10 * TN + 2 + TM * TN - 1) / (TM * TN) + 1;
#pragma unroll
        for (int j = 0; j < TM * TN; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);  // decode / address VALU
        }
    };

    uint4 afA[TM], bfA[TN], afB[TM], bfB[TN];
    // one step; buf = kb % NBUF is a compile-time constant, q0/q1/q2 = weights of steps kb, kb+1, kb+2
    auto step = [&](int kb, auto buf_c, uint64_t (&q0)[TN], uint64_t (&q1)[TN], uint64_t (&q2)[TN]) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr int buf1 = (buf + 1) % NBUF, buf2 = (buf + 2) % NBUF;
        // Each 16-deep sub-step: the 4 (TM) LDS reads of the NEXT sub-step's activation fragments go
        // first (pinned by sched_barrier so they cannot sink to their use), then the 8 (TM*TN) MFMAs of
        // the CURRENT sub-step are interleaved with the decode of the next weight fragments.
        read_a(buf, 1, afB);
        __builtin_amdgcn_sched_barrier(0);
        decode_b(q0, 1, bfB); mma(afA, bfA);
        interleave_hint();
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 2, afA);
        __builtin_amdgcn_sched_barrier(0);
        decode_b(q0, 2, bfA); mma(afB, bfB);
        interleave_hint();
        __builtin_amdgcn_sched_barrier(0);
        // ---- middle of the step: tile kb+1 (and its weights q1) complete, tile kb-1 retired ----
        wait_loads(q1);
        __builtin_amdgcn_s_barrier();
        // branch-free tail: past the end the loads re-fetch the last tile into a buffer nobody reads
        // again and the prefetched fragments are never used
        const int kn = kb + 2 < nk ? kb + 2 : nk - 1;
        read_a(buf, 3, afB);
        stage_a(kn, buf2); load_b(kn, q2);
        __builtin_amdgcn_sched_barrier(0);
        decode_b(q0, 3, bfB); mma(afA, bfA);
        interleave_hint();
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf1, 0, afA);
        __builtin_amdgcn_sched_barrier(0);
        decode_b(q1, 0, bfA); mma(afB, bfB);
        interleave_hint();
        __builtin_amdgcn_sched_barrier(0);
    };

    stage_a(0, 0);
    load_b(0, bq0);
    if (nk > 1) { stage_a(1, 1); load_b(1, bq1); }
    else {
#pragma unroll
        for (int t = 0; t < TN; ++t) bq1[t] = 0;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) bq2[t] = 0;
    wait_loads(bq0);
    wait_loads(bq1);
    __builtin_amdgcn_s_barrier();
    read_a(0, 0, afA);
    decode_b(bq0, 0, bfA);
    for (int kb = 0; kb < nk; kb += NBUF) {
        step(kb, std::integral_constant<int, 0>{}, bq0, bq1, bq2);
        if (kb + 1 < nk) step(kb + 1, std::integral_constant<int, 1>{}, bq1, bq2, bq0);
        if (kb + 2 < nk) step(kb + 2, std::integral_constant<int, 2>{}, bq2, bq0, bq1);
    }
    // Drain before leaving the loop: no LDS-DMA may outlive the workgroup's LDS allocation, and the
    // destination registers of the last (unused) weight loads must stay reserved until the data has
    // landed - a register freed earlier could be reused and then overwritten by the late load.
    wait_loads(bq0);
    wait_loads(bq1);
    wait_loads(bq2);

    // ---- epilogue ---------------------------------------------------------------------------------
    // The weights are the MFMA "A" operand, so a lane owns ONE activation row m = .. + r and, per
    // accumulator register group q = e >> 2, four consecutive output channels n = .. + 8q + 4h + (e & 3).
    // 16/32-bit outputs: one v_permlane32_swap per register pair hands lanes 0-31 the upper half-wave's
    // channels of group q and lanes 32-63 the lower half-wave's channels of group q+1, so every lane
    // stores 16 contiguous bytes (8x fewer store instructions than one element per lane).
    const float c = fabsf(clip);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(ld_y * (int64_t)sizeof(TY))) & 15) == 0;
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int nb = n0 + wc * WN + t * 32;  // first channel of this 32-wide tile
        float sc[16], bi[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int gn = nb + (e & 3) + 8 * (e >> 2) + 4 * h;
            sc[e] = (!SPLIT && gn < n) ? scale[gn] : 0.0f;
            bi[e] = (!SPLIT && HAS_BIAS && gn < n) ? bias[gn] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gm = m0 + wr * WM + i * 32 + r;
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                v[e] = SPLIT ? acc[i][t][e] : epilogue_ch<HAS_BIAS, ACT>(acc[i][t][e], sc[e], bi[e], c, nb + (e & 3) + 8 * (e >> 2) + 4 * h, act_begin);
            TY* yrow = y + (int64_t)(gm < m ? gm : 0) * ld_y;
            if (vec_ok && nb + 32 <= n) {
                if (gm < m) {
                    if constexpr (sizeof(TY) == 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<float4*>(yrow + nb + 8 * q + 4 * h) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                    }
                }
                if constexpr (sizeof(TY) == 2) {
                    // pack 4 channels -> 8 bytes per group, then swap halves between register pairs
                    uint2 pk[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        TY tmp[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) tmp[j] = from_float<TY>(v[4 * q + j]);
                        pk[q] = *reinterpret_cast<const uint2*>(tmp);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        // all 64 lanes take part in the swap (no divergence above this point)
                        auto sx = __builtin_amdgcn_permlane32_swap(pk[q].x, pk[q + 1].x, false, false);
                        auto sy = __builtin_amdgcn_permlane32_swap(pk[q].y, pk[q + 1].y, false, false);
                        // lanes 0-31: [own group q | upper half's group q]; lanes 32-63: [lower half's group q+1 | own q+1]
                        const uint4 out = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        if (gm < m) *reinterpret_cast<uint4*>(yrow + nb + 8 * q + 8 * h) = out;
                    }
                }
            } else if (gm < m) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int gn = nb + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (gn < n) yrow[gn] = from_float<TY>(v[e]);
                }
            }
        }
    }
}

// ---- the 256 x 256 tile on v_mfma_f32_16x16x32 -------------------------------------------------------------------
// Same workgroup tile, wave tile (256 rows x 32 channels), LDS image, staging and pipeline as gemm_mfma_kernel with
// <256, 256, 1, 8>; only the matrix instruction differs.  Cycles per flop are equal, but under matrix-core load the
// chip holds a higher clock on this shape (measured by in-process A/B on random data), so it is the one used for
// the big tile.  Operand layout: lane (c = lane & 15, g = lane >> 4) holds 8 consecutive k of weight row c
// (operand A) and of activation row c (operand B); D register e is channel 4 g + e, activation row c.
// k order inside a 64-step: lane group g owns k = 16 q .. 16 q + 15 with q = {0, 3, 1, 2}[g] (one dword of packed
// weights per 16-channel tile); MFMA group s = 0, 1 takes the dword's halfword s and the activation chunk 2 q + s.
// That assignment of q keeps the swizzled ds_read_b128 fragment reads conflict-free: the two lane groups that share
// an LDS cycle read chunk indices whose bits 1 and 2 have equal parity.
#ifdef TVL_CLOCK_PROBE
// Diagnostic builds only (tools/hwtests/gemm_clock.hip): shader-clock and 100 MHz real-time stamps at kernel entry, around the k loop
// and after the epilogue, 8 values per workgroup, into a buffer nothing else reads.
__device__ unsigned long long tvl_clock_probe[8 * 4096];
__device__ unsigned long long tvl_clock_probe_switch[2 * 4096];   // PERSIST: shader clocks spent in tile switches (stores included), and their count
__device__ unsigned long long tvl_clock_probe_steps[4 * 256];      // shader clock at the start of the first 256 steps of workgroups 0..3 (wave 0; kept in LDS behind the tiles)
#define TVL_STAMP(i) do { stamp_c[i] = __builtin_amdgcn_s_memtime(); stamp_r[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// clock-probe builds only: TVL_STEP_ABLATE removes parts of the 16x16x32 kernel's step (timing studies; results are wrong by construction):
// 1 no activation LDS-DMA, 2 no weight loads, 4 no decode, 8 no fragment reads, 16 no matrix instructions, 32 no barrier, 64 the barrier in every
// second step only (four LDS tiles)
#ifndef TVL_STEP_ABLATE
#define TVL_STEP_ABLATE 0
#endif
#define TVL_ABLATED(bit) ((TVL_STEP_ABLATE & (bit)) != 0)
#else
#define TVL_STAMP(i) do { } while (0)
#define TVL_ABLATED(bit) false
#endif

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t nts_u32x4 __attribute__((ext_vector_type(4)));

template <typename TX>
__device__ __forceinline__ floatx4 mfma16(uint4 a, uint4 b, floatx4 c) {
    if constexpr (__is_same(TX, f16)) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
}

// PERSIST: the grid has fewer workgroups than tiles; workgroup b walks tiles b, b + gridDim.x, ... as ONE stream of k
// steps: the loads of the next tile's first steps are issued during the current tile's last steps and its epilogue
// (stores are asynchronous), so the per-tile launch / prologue / drain cost is paid once per workgroup, not per tile.
// NB = LDS tiles: 3 = a tile is requested one step (~1.3 us) before its use, enough for operands that sit in the
// Infinity Cache; 4 = two steps ahead (the wait at mid-step leaves the newer group of loads in flight: counted vmcnt),
// which also covers activations that come cold from HBM.
// With 16-bit outputs it also requests the next tile's scale / bias values by LDS-DMA at every tile switch (stage_params: 512 bytes per
// wave behind the activation tiles, persist_param_lds_bytes), so a tile's epilogue starts without a trip to memory, and writes its
// tiles with plain stores.
// WAVES_M / WN_ / BM_ / WAVES: WAVES_M x (WAVES / WAVES_M) waves of (BM_ / WAVES_M) rows x WN_ channels each; every shipped form has WAVES_M = 1:
// BM_ / WAVES: the workgroup tile is BM_ rows x 32 * WAVES channels (one wave per 32 channels, every wave all BM_ rows): 256 x 256
// with 8 waves is the big tile; 256 x 128 and 128 x 128 with 4 waves serve problems with fewer rows, where a 256-wide tile
// would leave compute units idle.  SPLIT: blockIdx.y selects a range of k steps and the raw float32 partial sums go to slab
// blockIdx.y of `y` (float [splits][m][ld_y]); splitk_reduce_kernel adds the slabs in a fixed order and applies the epilogue.
// NB = LDS tiles = how far ahead a tile is requested (NB - 2 steps): a 128-row tile's step is ~0.45 us, so it needs NB = 4 where
// the 256 x 256 tile's 1.6 us step gets by with 3 (measured: 128 x 128 at M = 1024: 47.9 -> 36.5 us from NB = 3 to 4; 5 and 6 add
// nothing).  (Tried and dropped: two 4-wave groups per workgroup taking alternate k steps of one tile with an exchange of
// half the accumulators through LDS at the end - no faster than one group with NB = 4.)
// PARTS: phases per MFMA group of a step.  2 = four phases of MT / 2 row tiles; 4 = eight phases of MT / 4 (fewer fragment registers,
// which is what lets the persistent 256 x 256 form keep a fourth weight register set).

// dynamic LDS the persistent form needs beyond its NB activation tiles (the staged scale / bias values)
constexpr int persist_param_lds_bytes(int waves) { return 512 * waves; }

template <typename TX, typename TY, bool HAS_BIAS, int ACT, int WAVES_M, bool PERSIST, int NB, int BM_ = 256, int WAVES = 8, bool SPLIT = false,
          int MIN_WAVES = 2, int WN_ = 32, int PARTS = 2>
__global__ void __launch_bounds__(64 * WAVES, MIN_WAVES)
gemm_mfma16_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                   const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act_begin, TY* __restrict__ y,
                   int64_t ld_y, int tiles_n, int nblk, int w_steps, int steps_per_split) {
    static_assert(!(SPLIT && PERSIST), "a split k range belongs to one tile");
    static_assert(BM_ % 32 == 0 && (BM_ * 8) % (64 * WAVES) == 0, "tile / workgroup mismatch");
#ifdef TVL_CLOCK_PROBE
    unsigned long long stamp_c[4], stamp_r[4], switch_cycles = 0, switch_count = 0;
    int probe_step = 0;
    TVL_STAMP(0);
#endif
    constexpr int WAVES_N = WAVES / WAVES_M;
    constexpr int BM = BM_, BN = WN_ * WAVES_N, NT = 64 * WAVES;
    constexpr int WM = BM / WAVES_M, WN = WN_;            // wave tile: (BM / WAVES_M) rows x WN_ channels
    constexpr int MT = WM / 16, NTL = WN / 16;            // 16 x 16 tiles per wave: 16 x 2 or 8 x 4
    constexpr int MH = MT / 2, NH = NTL / 2;              // per phase: MH row tiles x NTL channel tiles, NH decodes
    constexpr int MP = MT / PARTS;                        // PARTS != 2: phases of MP row tiles (fewer fragment registers: more waves per SIMD)
    static_assert(MT % PARTS == 0 && (PARTS == 2 || PARTS % NTL == 0), "phase split");
    constexpr int A_TILE_BYTES = BM * ROW_BYTES;
    constexpr int ROUNDS = (BM * 8) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // NB * A_TILE_BYTES (+ persist_param_lds_bytes(WAVES) for PERSIST)
    constexpr int PARAM_OFF = NB * A_TILE_BYTES;                         // PERSIST: per wave 256 B of scale + 256 B of bias of the tile being multiplied
    [[maybe_unused]] constexpr int PROBE_OFF = PARAM_OFF + (PERSIST ? 512 * WAVES : 0);   // clock-probe builds: the step stamps
    static_assert(NB >= 3 && NB <= 6, "three to six activation tiles in LDS");
    constexpr int AHEAD = NB - 3;   // groups of loads (one step's tile pieces + weight dwords) the mid-step wait leaves in flight

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wr = wid / WAVES_N, wc = wid % WAVES_N;
    const int q = (0x2130 >> (4 * g)) & 3;
    const int all_steps = k / BK;
    const int kb0 = SPLIT ? (int)blockIdx.y * steps_per_split : 0;   // first k step of this workgroup
    const int nk = SPLIT ? (all_steps - kb0 < steps_per_split ? all_steps - kb0 : steps_per_split) : all_steps;
    if constexpr (SPLIT) y += (int64_t)blockIdx.y * m * ld_y;
    const int G = gridDim.x;
    const int my_tiles = PERSIST ? (nblk - (int)blockIdx.x + G - 1) / G : 1;
    const int total_steps = my_tiles * nk;
    int vb_l = blockIdx.x, kb_l = 0;   // load cursor: tile (as a virtual block index) and k step the next loads fetch
    int vb_c = blockIdx.x, kb_c = 0;   // the step being multiplied
    int m0, n0;                        // origin of the tile being multiplied
    {
        const int tile = xcd_remap(vb_c, nblk);
        m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) * BN;
    }

    // Load addresses = a wave-uniform 64-bit base (tile origin + k step, kept in scalar registers) + a 32-bit per-lane byte
    // offset inside the tile: half the address registers of 64-bit per-lane pointers and no vector address arithmetic per step.
    const TX* a_base;
    const uint8_t* b_base;
    uint32_t a_off[ROUNDS], b_off[NTL];
    auto set_load_tile = [&](int vb) __attribute__((always_inline)) {
        const int tile = xcd_remap(vb, nblk);
        const int lm0 = (tile / tiles_n) * BM, ln0 = (tile % tiles_n) * BN;
        a_base = x + (int64_t)lm0 * ld_x;
        b_base = pw + (int64_t)ln0 * ld_pw;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const int p = i * NT + tid;
            const int row = p >> 3, cs = p & 7;
            const int chunk = cs ^ ((row >> 1) & 7);
            int gm = lm0 + row;
            gm = gm < m ? gm : m - 1;
            a_off[i] = (uint32_t)(((int64_t)(gm - lm0) * ld_x + chunk * 8) * (int64_t)sizeof(TX));
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            int gn = ln0 + wc * WN + t * 16 + c;
            gn = gn < n ? gn : n - 1;
            b_off[t] = (uint32_t)((int64_t)(gn - ln0) * ld_pw + 4 * q);
        }
    };
    set_load_tile(vb_l);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
    auto stage_a = [&](int kb, int buf) __attribute__((always_inline)) {
        if constexpr (TVL_ABLATED(1)) return;
#pragma unroll
        for (int i = 0; i < ROUNDS; ++i) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * A_TILE_BYTES + (i * NT + wid * 64) * 16);
            const TX* sbase = a_base + (int64_t)(kb0 + kb) * BK;   // wave-uniform
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(a_off[i]), "s"(dst), "s"(sbase) : "memory");
        }
    };

    // PERSIST: the scale and bias values of the wave's channels of tile origin tn0, by LDS-DMA into the wave's own 512 bytes (one
    // dword per lane: channel tn0 + wc * WN + lane, clamped to the last channel).  Requested a whole tile before the epilogue
    // that reads them, so that epilogue does not start with a trip to memory (1 300 cycles per tile by the in-kernel clock).
    auto stage_params = [&](int tn0) __attribute__((always_inline)) {
        static_assert(WN <= 64, "one lane per channel of the wave");
        int gn = tn0 + wc * WN + lane;
        gn = gn < n ? gn : n - 1;
        const uint32_t off = (uint32_t)gn * 4u;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + PARAM_OFF + wid * 512);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(dst), "s"(scale) : "memory");
        if constexpr (HAS_BIAS)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(dst + 256), "s"(bias) : "memory");
    };

    auto load_b = [&](int kb, uint32_t (&dst)[NTL]) __attribute__((always_inline)) {
        int kw = kb0 + kb;
        kw = kw >= w_steps ? kw - w_steps : kw;
        kw = kw >= w_steps ? kw - w_steps : kw;
        if constexpr (TVL_ABLATED(2)) {
#pragma unroll
            for (int t = 0; t < NTL; ++t) asm volatile("v_mov_b32 %0, %1" : "=v"(dst[t]) : "v"(b_off[t]));
            return;
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t)
            asm volatile("global_load_dword %0, %1, %2" : "=v"(dst[t]) : "v"(b_off[t]), "s"(b_base + (int64_t)kw * (BK / 4)) : "memory");
    };
    // fetch the cursor's step into LDS buffer `buf` / registers `w`, then advance the cursor (to the next tile of this
    // workgroup after a tile's last step; past the very last step it stays there: harmless re-fetches into a buffer
    // nobody reads again, so that the number of loads per step is constant)
    auto issue_loads = [&](int buf, uint32_t (&w)[NTL]) __attribute__((always_inline)) {
        stage_a(kb_l, buf);
        load_b(kb_l, w);
        if (++kb_l == nk) {
            if (PERSIST && vb_l + G < nblk) { vb_l += G; kb_l = 0; set_load_tile(vb_l); }
            else kb_l = nk - 1;
        }
    };
    auto wait_loads = [&](uint32_t (&w)[NTL]) {
        if constexpr (NTL == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : : "memory");
    };
    // as wait_loads, but the AHEAD newest groups of loads (a group = one step's ROUNDS tile pieces + NTL weight dwords) stay in
    // flight: loads complete in issue order, so at most that many outstanding means every older load has landed
    auto wait_loads_keep_newest = [&](uint32_t (&w)[NTL]) {
        static_assert(AHEAD * (ROUNDS + NTL) <= 63, "vmcnt is a 6-bit counter");
        if constexpr (NTL == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(AHEAD * (ROUNDS + NTL)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(AHEAD * (ROUNDS + NTL)) : "memory");
    };

    auto wait_mid_step = [&](uint32_t (&w)[NTL]) __attribute__((always_inline)) {
        if constexpr (NB == 3) wait_loads(w);
        else wait_loads_keep_newest(w);
    };

    const int sw = (c >> 1) & 7;                 // (row >> 1) & 7 with row = 16 i + c
    const int a_row_off = (wr * WM + c) * ROW_BYTES;
    int a_col_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) a_col_off[s] = ((2 * q + s) ^ sw) * 16;

    floatx4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const DecodeConsts dc = make_decode_consts();

    auto read_a = [&](int buf, int s, int half, uint4 (&af)[MH]) __attribute__((always_inline)) {
        const unsigned char* abase = lds + buf * A_TILE_BYTES + a_row_off + a_col_off[s] + half * MH * 16 * ROW_BYTES;
        if constexpr (TVL_ABLATED(8)) {
#pragma unroll
            for (int i = 0; i < MH; ++i) asm volatile("" : "=v"(af[i].x), "=v"(af[i].y), "=v"(af[i].z), "=v"(af[i].w));
            return;
        }
#pragma unroll
        for (int i = 0; i < MH; ++i) af[i] = *reinterpret_cast<const uint4*>(abase + i * 16 * ROW_BYTES);
    };
    auto decode_b = [&](uint32_t d, int s) {
        if constexpr (TVL_ABLATED(4)) return make_uint4(d, d, d, d);
        return s ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
    };
    auto mma = [&](const uint4 (&af)[MH], int half, const uint4 (&bf)[NTL]) __attribute__((always_inline)) {
        if constexpr (TVL_ABLATED(16)) {
#pragma unroll
            for (int i = 0; i < MH; ++i) asm volatile("" :: "v"(af[i].x), "v"(af[i].y), "v"(af[i].z), "v"(af[i].w));
#pragma unroll
            for (int t = 0; t < NTL; ++t) asm volatile("" :: "v"(bf[t].x), "v"(bf[t].y), "v"(bf[t].z), "v"(bf[t].w));
            return;
        }
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[MH * half + i][t] = mfma16<TX>(bf[t], af[i], acc[MH * half + i][t]);
    };
    // the same in PARTS row parts per MFMA group (PARTS != 2)
    auto read_a_g = [&](int buf, int s, int part, uint4 (&af)[MP]) __attribute__((always_inline)) {
        const unsigned char* abase = lds + buf * A_TILE_BYTES + a_row_off + a_col_off[s] + part * MP * 16 * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < MP; ++i) af[i] = *reinterpret_cast<const uint4*>(abase + i * 16 * ROW_BYTES);
    };
    auto mma_g = [&](const uint4 (&af)[MP], int part, const uint4 (&bf)[NTL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[MP * part + i][t] = mfma16<TX>(bf[t], af[i], acc[MP * part + i][t]);
    };
    // decode NH of the NTL weight fragments of MFMA group s (the half `part`) from the packed dwords w
    auto decode_part = [&](const uint32_t (&w)[NTL], int s, int part, uint4 (&bf)[NTL]) {
#pragma unroll
        for (int t = 0; t < NH; ++t) bf[part * NH + t] = decode_b(w[part * NH + t], s);
    };
    auto interleave_hint = [&]() {
#pragma unroll
        for (int j = 0; j < MH * NTL; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (NH * 16 + MH * NTL - 1) / (MH * NTL), 0);  // decode / address VALU
        }
    };

    // The epilogue of a tile that lies wholly inside the output, 16-bit outputs, 16-byte aligned rows: no bounds tests, scale and
    // bias as four 16-byte loads, one row pointer advanced by a constant, and for float16 channels without activation the clip on the packed
    // halves AFTER the conversion (v_pk_minimum3_f16 / v_pk_maximum3_f16: two values per instruction).  Rounding to nearest is
    // monotone, so round(min(max(v, -c), c)) = min(max(round(v), -round(c)), round(c)) for every v including +-inf and NaN
    // (both forms propagate NaN): bit-identical to the general path below.  ~40 % fewer vector instructions per tile.
    // PERSIST: an empty asm that uses the fragments a phase has just decoded (the half `part` of bf).  Their real users sit in a later
    // basic block there (the step is split by the load cursor's tile switch and the loop's continuation test), and machine sinking
    // would move the side-effect-free decode down to them, out of its phase: 20 vector instructions in a row, the matrix pipe idle.
    auto pin_decoded = [&](int part, uint4 (&bf)[NTL]) __attribute__((always_inline)) {
        if constexpr (PERSIST) {
#pragma unroll
            for (int t = 0; t < NH; ++t) {
                uint4& f = bf[part * NH + t];
                asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w));
            }
        }
    };

    auto store_tile_full = [&]() __attribute__((always_inline)) {
        if constexpr (sizeof(TY) == 2 && !SPLIT) {
        const float cl = fabsf(clip);
        const int nb = n0 + wc * WN;
        const int noff = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        const half2_t c2 = {(_Float16)cl, (_Float16)cl};
    #pragma unroll
        for (int tp = 0; tp < NTL; tp += 2) {
            floatx4 sc[2], bi[2];
    #pragma unroll
            for (int t = 0; t < 2; ++t) {
                if constexpr (PERSIST) {   // staged by stage_params a tile ago
                    const unsigned char* par = lds + PARAM_OFF + wid * 512 + (16 * (tp + t) + 4 * g) * 4;
                    sc[t] = *reinterpret_cast<const floatx4*>(par);
                    if constexpr (HAS_BIAS) bi[t] = *reinterpret_cast<const floatx4*>(par + 256);
                } else {
                    sc[t] = *reinterpret_cast<const floatx4*>(scale + nb + 16 * (tp + t) + 4 * g);
                    if constexpr (HAS_BIAS) bi[t] = *reinterpret_cast<const floatx4*>(bias + nb + 16 * (tp + t) + 4 * g);
                }
                if constexpr (!HAS_BIAS) bi[t] = (floatx4){0.f, 0.f, 0.f, 0.f};
            }
            unsigned char* prow = reinterpret_cast<unsigned char*>(y + (int64_t)(m0 + wr * WM + c) * ld_y + nb + 16 * tp + noff);
            const int64_t row_step = 16 * ld_y * (int64_t)sizeof(TY);
            // MODE: the activation on none (0) / all (1) / some (2: tested per channel) of these 32 channels; wave-uniform
            auto rows = [&](auto mode_c) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mode_c)::value;
                constexpr bool CLIP16 = std::is_same<TY, f16>::value && MODE == 0;
    #pragma unroll
                for (int i = 0; i < MT; ++i) {
                    uint2 pk[2];
    #pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        TY tmp[4];
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v;
                            if constexpr (CLIP16) v = scale_bias<HAS_BIAS>(acc[i][tp + t][e], sc[t][e], bi[t][e]);
                            else if constexpr (MODE == 0) v = epilogue<HAS_BIAS, TVL_ACT_NONE>(acc[i][tp + t][e], sc[t][e], bi[t][e], cl);
                            else if constexpr (MODE == 1) v = epilogue<HAS_BIAS, ACT>(acc[i][tp + t][e], sc[t][e], bi[t][e], cl);
                            else v = epilogue_ch<HAS_BIAS, ACT>(acc[i][tp + t][e], sc[t][e], bi[t][e], cl, nb + 16 * (tp + t) + 4 * g + e, act_begin);
                            tmp[e] = from_float<TY>(v);
                        }
                        pk[t] = *reinterpret_cast<const uint2*>(tmp);
                        if constexpr (CLIP16) {
                            pk[t].x = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_minimum(__builtin_bit_cast(half2_t, pk[t].x), c2), -c2));
                            pk[t].y = __builtin_bit_cast(uint32_t, __builtin_elementwise_maximum(__builtin_elementwise_minimum(__builtin_bit_cast(half2_t, pk[t].y), c2), -c2));
                        }
                    }
                    auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                    auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                    const uint4 out = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    // One tile per workgroup: streaming (non-temporal) stores, +0.5 % at 4096^3.  The persistent form writes several times
                    // the bytes it reads (ViT-Base fc1: 155 MB out, 43 MB in); there plain stores, which leave through L2 as whole 128-byte
                    // lines, are 1.6 - 2.3 % faster (the write counter shows 1.4 x the output's bytes for the non-temporal form), equal elsewhere.
                    if constexpr (PERSIST) *reinterpret_cast<uint4*>(prow + i * row_step) = out;
                    else __builtin_nontemporal_store(__builtin_bit_cast(nts_u32x4, out), reinterpret_cast<nts_u32x4*>(prow + i * row_step));
                }
            };
            if constexpr (ACT == TVL_ACT_NONE) {
                rows(std::integral_constant<int, 0>{});
            } else {
                const int ch0 = __builtin_amdgcn_readfirstlane(nb + 16 * tp);
                if (ch0 >= act_begin) rows(std::integral_constant<int, 1>{});
                else if (ch0 + 32 <= act_begin) rows(std::integral_constant<int, 0>{});
                else rows(std::integral_constant<int, 2>{});
            }
        }
        }
    };

    auto store_tile = [&]() __attribute__((always_inline)) {
        // ---- epilogue: lane (c, g) owns activation row 16 i + c and channels 16 t + 4 g + e -------------------------------
        // 16-bit outputs: v_permlane16_swap hands even lane groups their neighbour's 4 channels of tile 0 and odd lane
        // groups their neighbour's 4 channels of tile 1, so every lane stores 16 contiguous bytes.
        const float cl = fabsf(clip);
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(ld_y * (int64_t)sizeof(TY))) & 15) == 0;
        if constexpr (sizeof(TY) == 2 && !SPLIT) {
            // scale / bias rows of 16 bytes need 16-byte aligned vectors too (n0, 16 t and 4 g are multiples of 4 floats)
            const bool par_ok = PERSIST || ((reinterpret_cast<uintptr_t>(scale) | (HAS_BIAS ? reinterpret_cast<uintptr_t>(bias) : 0)) & 15) == 0;
            if (vec_ok && par_ok && m0 + BM <= m && n0 + BN <= n) { store_tile_full(); return; }
        }
        const int nb = n0 + wc * WN;
        const int noff = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;   // first of the 8 channels a lane stores after the swap
    #pragma unroll
        for (int tp = 0; tp < NTL; tp += 2) {                  // pairs of 16-channel tiles
            float sc[2][4], bi[2][4];
    #pragma unroll
            for (int t = 0; t < 2; ++t)
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int gn = nb + 16 * (tp + t) + 4 * g + e;
                    sc[t][e] = (!SPLIT && gn < n) ? scale[gn] : 0.0f;
                    bi[t][e] = (!SPLIT && HAS_BIAS && gn < n) ? bias[gn] : 0.0f;
                }
    #pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int gm = m0 + wr * WM + 16 * i + c;
                float v[2][4];
    #pragma unroll
                for (int t = 0; t < 2; ++t)
    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[t][e] = SPLIT ? acc[i][tp + t][e]
                                        : epilogue_ch<HAS_BIAS, ACT>(acc[i][tp + t][e], sc[t][e], bi[t][e], cl, nb + 16 * (tp + t) + 4 * g + e, act_begin);
                TY* yrow = y + (int64_t)(gm < m ? gm : 0) * ld_y;
                if (vec_ok && nb + 16 * tp + 32 <= n) {
                    if constexpr (sizeof(TY) == 4) {
                        if (gm < m) {
    #pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                floatx4* dst = reinterpret_cast<floatx4*>(yrow + nb + 16 * (tp + t) + 4 * g);
                                if constexpr (SPLIT) *dst = (floatx4){v[t][0], v[t][1], v[t][2], v[t][3]};   // re-read at once by the reduce: keep it cached
                                else __builtin_nontemporal_store((floatx4){v[t][0], v[t][1], v[t][2], v[t][3]}, dst);
                            }
                        }
                    } else {
                        uint2 pk[2];
    #pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            TY tmp[4];
    #pragma unroll
                            for (int e = 0; e < 4; ++e) tmp[e] = from_float<TY>(v[t][e]);
                            pk[t] = *reinterpret_cast<const uint2*>(tmp);
                        }
                        // all 64 lanes take part in the swap (no divergence above this point)
                        auto sx = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                        auto sy = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                        const uint4 out = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        // streaming (non-temporal) stores: the output is written once and is far larger than L2; +2..3 % by A/B
                        if (gm < m) __builtin_nontemporal_store(__builtin_bit_cast(nts_u32x4, out), reinterpret_cast<nts_u32x4*>(yrow + nb + 16 * tp + noff));
                    }
                } else if (gm < m) {
    #pragma unroll
                    for (int t = 0; t < 2; ++t)
    #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int gn = nb + 16 * (tp + t) + 4 * g + e;
                            if (gn < n) yrow[gn] = from_float<TY>(v[t][e]);
                        }
                }
            }
        }
    };

    // packed weights (16 per lane and tile) of steps NB j, NB j + 1, ...  Separate named arrays on purpose: as one two-dimensional
    // array the register allocator inserts copies of them at the loop's back edge, while their loads are still in flight
    // (caught by the build's assembly audit).
    uint32_t bq0[NTL], bq1[NTL], bq2[NTL], bq3[NTL], bq4[NTL], bq5[NTL];
    uint4 afA[MH], afB[MH], bf0[NTL], bf1[NTL];   // bf0 / bf1: decoded weights of MFMA group s = 0 / 1
    uint4 afG0[MP], afG1[MP];                     // PARTS != 2: fragments of even / odd phases
    // PARTS != 2: one phase p of the 2 PARTS phases of a step (MFMA group s = p / PARTS, row part p % PARTS): request the next
    // phase's activation fragments, decode one weight fragment of the next MFMA group every PARTS / NTL phases, multiply
    auto phase_g = [&](auto p_c, auto buf_c, uint32_t (&w0)[NTL], uint32_t (&w1)[NTL], uint32_t (&wn)[NTL]) __attribute__((always_inline)) {
        constexpr int p = decltype(p_c)::value, buf = decltype(buf_c)::value;
        constexpr int s = p / PARTS, part = p % PARTS;
        constexpr int pn = p + 1, sn = (pn / PARTS) & 1, partn = pn % PARTS;
        constexpr int bufr = pn == 2 * PARTS ? (buf + 1) % NB : buf;
        constexpr int every = PARTS / NTL;                 // a fragment is decoded in phases part = every * t
        if constexpr ((pn & 1) == 0) read_a_g(bufr, sn, partn, afG0); else read_a_g(bufr, sn, partn, afG1);
        if constexpr (p == PARTS) issue_loads((buf + NB - 1) % NB, wn);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (part % every == 0) {
            if constexpr (s == 0) bf1[part / every] = decode_b(w0[part / every], 1);
            else bf0[part / every] = decode_b(w1[part / every], 0);
        }
        if constexpr ((p & 1) == 0) { if constexpr (s == 0) mma_g(afG0, part, bf0); else mma_g(afG0, part, bf1); }
        else { if constexpr (s == 0) mma_g(afG1, part, bf0); else mma_g(afG1, part, bf1); }
#pragma unroll
        for (int j = 0; j < MP * NTL; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, part % every == 0 ? (16 + MP * NTL - 1) / (MP * NTL) : 0, 0);
        }
        if constexpr (part % every == 0) {
            // The empty asm uses the decoded fragment HERE.  Its real users sit in a later basic block when the step is split by a
            // branch (the persistent form's cursor switch, the loop's continuation test), and machine sinking then moves the
            // side-effect-free decode down to them, out of this phase: 20 vector instructions in a row with the matrix pipe idle.
            uint4& f = s == 0 ? bf1[part / every] : bf0[part / every];
            asm volatile("" : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (p == PARTS - 1) {   // middle of the step: tile kb+1 (and its weights w1) complete, tile kb-1 retired
            wait_mid_step(w1);
            __builtin_amdgcn_s_barrier();
        }
    };
    // one step = four phases (s, half of the 16 row tiles); each phase reads the NEXT phase's activation fragments
    // first, then interleaves its 16 MFMAs with the decode of one weight fragment needed two phases later
    // w0 / w1: weights of this step and the next; wn: the registers the loads issued in this step (for step + NB - 1) fill
    auto step = [&](auto buf_c, uint32_t (&w0)[NTL], uint32_t (&w1)[NTL], uint32_t (&wn)[NTL]) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_c)::value;
        constexpr int buf1 = (buf + 1) % NB, bufn = (buf + NB - 1) % NB;
#ifdef TVL_CLOCK_PROBE
        if (tid == 0 && probe_step < 256) *reinterpret_cast<unsigned long long*>(lds + PROBE_OFF + 8 * probe_step) = __builtin_amdgcn_s_memtime();
        ++probe_step;
#endif
        if constexpr (PARTS != 2) {
            static_assert(PARTS == 2 || PARTS == 4, "two or four row parts per MFMA group");
            phase_g(std::integral_constant<int, 0>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 1>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 2>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 3>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 4>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 5>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 6>{}, buf_c, w0, w1, wn);
            phase_g(std::integral_constant<int, 7>{}, buf_c, w0, w1, wn);
        } else {
        read_a(buf, 0, 1, afB);
        __builtin_amdgcn_sched_barrier(0);
        decode_part(w0, 1, 0, bf1); mma(afA, 0, bf0);
        interleave_hint();
        pin_decoded(0, bf1);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 1, 0, afA);
        __builtin_amdgcn_sched_barrier(0);
        decode_part(w0, 1, 1, bf1); mma(afB, 1, bf0);
        interleave_hint();
        pin_decoded(1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- middle of the step: tile kb+1 (and its weights w1) complete, tile kb-1 retired ----
        wait_mid_step(w1);
        if constexpr (!TVL_ABLATED(32) && !(TVL_ABLATED(64) && (buf & 1))) __builtin_amdgcn_s_barrier();
        read_a(buf, 1, 1, afB);
        issue_loads(bufn, wn);
        __builtin_amdgcn_sched_barrier(0);
        decode_part(w1, 0, 0, bf0); mma(afA, 0, bf1);
        interleave_hint();
        pin_decoded(0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf1, 0, 0, afA);
        __builtin_amdgcn_sched_barrier(0);
        decode_part(w1, 0, 1, bf0); mma(afB, 1, bf1);
        interleave_hint();
        pin_decoded(1, bf0);
        __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PERSIST) {
            if (++kb_c == nk) {   // last step of a tile: write it out and start the next one (its first fragments are in afA / afG0 and bf0)
#ifdef TVL_CLOCK_PROBE
                const unsigned long long sw0 = __builtin_amdgcn_s_memtime();
#endif
                // the staged scale / bias were requested a tile ago, before loads that every mid-step wait since has covered; a
                // tile of one or two steps has too few such waits
                if (nk < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                store_tile();
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NTL; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
                kb_c = 0;
                vb_c += G;
                const int tile = xcd_remap(vb_c < nblk ? vb_c : nblk - 1, nblk);
                m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) * BN;
                if constexpr (sizeof(TY) == 2) stage_params(n0);   // after the epilogue's reads of the old values (program order)
                __builtin_amdgcn_sched_barrier(0);
#ifdef TVL_CLOCK_PROBE
                switch_cycles += __builtin_amdgcn_s_memtime() - sw0; ++switch_count;
#endif
            }
        }
    };

    // prologue: NB - 1 steps requested (past the last step the cursor re-fetches it, so every count below is constant)
    if constexpr (PERSIST && sizeof(TY) == 2) stage_params(n0);   // the oldest loads in flight: landed once any wait below returns
    issue_loads(0, bq0);
    issue_loads(1, bq1);
    if constexpr (NB > 3) issue_loads(2, bq2);
    if constexpr (NB > 4) issue_loads(3, bq3);
    if constexpr (NB > 5) issue_loads(4, bq4);
    if constexpr (NB == 3) {
        wait_loads(bq0);
        wait_loads(bq1);
    } else {   // NB - 1 groups issued, NB - 3 may stay in flight: the first two have landed
        wait_loads_keep_newest(bq0);
        wait_loads_keep_newest(bq1);
    }
#pragma unroll
    for (int t = 0; t < NTL; ++t) {   // the sets no load has been issued into yet
        if constexpr (NB <= 3) bq2[t] = 0;
        if constexpr (NB <= 4) bq3[t] = 0;
        if constexpr (NB <= 5) bq4[t] = 0;
        bq5[t] = 0;
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (PARTS == 2) read_a(0, 0, 0, afA);
    else read_a_g(0, 0, 0, afG0);
    decode_part(bq0, 0, 0, bf0);
    decode_part(bq0, 0, 1, bf0);
    TVL_STAMP(1);
    // unrolled by NB: LDS buffers and weight registers have static roles.  Step J multiplies with set J, decodes ahead from set
    // J + 1 and issues the loads that fill set J - 1 (all modulo NB).
#define TVL_STEP(J, W0, W1, WN) step(std::integral_constant<int, (J)>{}, W0, W1, WN)
    if constexpr (NB == 3) {
        for (int gs = 0; gs < total_steps; gs += 3) {
            TVL_STEP(0, bq0, bq1, bq2);
            if (gs + 1 < total_steps) TVL_STEP(1, bq1, bq2, bq0);
            if (gs + 2 < total_steps) TVL_STEP(2, bq2, bq0, bq1);
        }
    } else if constexpr (NB == 4) {
        for (int gs = 0; gs < total_steps; gs += 4) {
            TVL_STEP(0, bq0, bq1, bq3);
            if (gs + 1 < total_steps) TVL_STEP(1, bq1, bq2, bq0);
            if (gs + 2 < total_steps) TVL_STEP(2, bq2, bq3, bq1);
            if (gs + 3 < total_steps) TVL_STEP(3, bq3, bq0, bq2);
        }
    } else if constexpr (NB == 5) {
        for (int gs = 0; gs < total_steps; gs += 5) {
            TVL_STEP(0, bq0, bq1, bq4);
            if (gs + 1 < total_steps) TVL_STEP(1, bq1, bq2, bq0);
            if (gs + 2 < total_steps) TVL_STEP(2, bq2, bq3, bq1);
            if (gs + 3 < total_steps) TVL_STEP(3, bq3, bq4, bq2);
            if (gs + 4 < total_steps) TVL_STEP(4, bq4, bq0, bq3);
        }
    } else {
        for (int gs = 0; gs < total_steps; gs += 6) {
            TVL_STEP(0, bq0, bq1, bq5);
            if (gs + 1 < total_steps) TVL_STEP(1, bq1, bq2, bq0);
            if (gs + 2 < total_steps) TVL_STEP(2, bq2, bq3, bq1);
            if (gs + 3 < total_steps) TVL_STEP(3, bq3, bq4, bq2);
            if (gs + 4 < total_steps) TVL_STEP(4, bq4, bq5, bq3);
            if (gs + 5 < total_steps) TVL_STEP(5, bq5, bq0, bq4);
        }
    }
#undef TVL_STEP
    wait_loads(bq0);
    wait_loads(bq1);
    wait_loads(bq2);
    wait_loads(bq3);
    wait_loads(bq4);
    wait_loads(bq5);
    TVL_STAMP(2);
    if constexpr (!PERSIST) store_tile();
#ifdef TVL_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave's queue
    TVL_STAMP(3);
    if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { tvl_clock_probe[8 * blockIdx.x + i] = stamp_c[i]; tvl_clock_probe[8 * blockIdx.x + 4 + i] = stamp_r[i]; }
        tvl_clock_probe_switch[2 * blockIdx.x] = switch_cycles; tvl_clock_probe_switch[2 * blockIdx.x + 1] = switch_count;
        if (blockIdx.x < 4)
            for (int i = 0; i < 256; ++i)
                tvl_clock_probe_steps[256 * blockIdx.x + i] = i < probe_step ? *reinterpret_cast<unsigned long long*>(lds + PROBE_OFF + 8 * i) : 0;
    }
#endif
}

}  // namespace tvl
