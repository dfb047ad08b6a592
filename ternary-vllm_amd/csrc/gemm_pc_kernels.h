// Packed-ternary x f16/bf16 matrix-core kernel with SPLIT ROLES: loader waves and matrix waves (template; instantiated by gemm_pc.hip).
//
//   y[m,n] = act(clip((sum_k x[m,k] * (code(pw[n,k]) - 1)) * scale[n] + bias[n]))
//
// The kernels of gemm_mfma_kernels.h are symmetric: every wave issues its own LDS-DMA pieces, weight loads, 2-bit decode,
// fragment reads and matrix instructions in one stream.  Here a workgroup is 4 matrix waves (one per SIMD) plus NPROD loader
// waves:
//  * loader waves: activation tile global -> LDS by LDS-DMA (same swizzled image as the symmetric kernels), packed weights
//    global -> registers -> 2-bit decode -> f16/bf16 {-1,0,+1,+2} -> ds_write_b128 into an LDS tile of the SAME image
//    (row = channel), so both MFMA operands are read alike;
//  * matrix waves: ds_read_b128 fragment reads and v_mfma_f32_16x16x32 only (2 x 2 waves on a BM x 128 tile, each BM/2 rows
//    x 64 channels: 24 fragment reads per 64 matrix instructions at BM = 256), then the fused epilogue.
// One workgroup barrier per 64-deep k step, placed in the matrix waves' stream where every fragment of the step has been
// read: what it publishes is step j+1 (its activation tile has landed, its weight tile is written), what it retires is step j
// (its buffers may be overwritten).  Loaders run ahead: activation tiles 3 steps (4 LDS tiles), weight tiles one step (2).
// Requires k % 64 == 0, 16-byte aligned activation rows, 8- (4-) byte aligned packed rows; any m, n (edge tiles clamp loads,
// mask stores).
#pragma once
#include "gemm_mfma_kernels.h"

namespace tvl {

constexpr int PC_BN = 128;                           // channels per workgroup tile
constexpr int PC_NA = 4;                             // activation tiles in LDS
constexpr int PC_W_TILE_BYTES = PC_BN * ROW_BYTES;   // decoded weight tile of one step: 128 channels x 64 k x 2 B = 16 KiB
constexpr size_t pc_lds_bytes(int bm) { return (size_t)PC_NA * bm * ROW_BYTES + 2 * PC_W_TILE_BYTES; }
#ifdef TVL_PC_ABLATE   // timing-only probe builds (results wrong by construction): 1 = no tile pieces, 2 = no weight decode / stores,
constexpr int PC_ABL = TVL_PC_ABLATE;   // 4 = no matrix instructions, 8 = no fragment reads
#else
constexpr int PC_ABL = 0;
#endif

// NPROD: loader waves (4: one per SIMD beside the matrix wave; 8: two).  PERSIST: fewer workgroups than tiles, workgroup b
// walks tiles b, b + gridDim.x, ... as one stream of k steps.  SPLIT: blockIdx.y selects a range of k steps, raw float32 partial
// sums go to slab blockIdx.y of `y` (float [splits][m][ld_y]).
template <typename TX, typename TY, bool HAS_BIAS, int ACT, int BM, int NPROD, bool PERSIST, bool SPLIT>
__global__ void __launch_bounds__(64 * (4 + NPROD), (4 + NPROD) / 4)
gemm_pc_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
               const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act_begin, TY* __restrict__ y,
               int64_t ld_y, int tiles_n, int nblk, int w_steps, int steps_per_split) {
    static_assert(!(SPLIT && PERSIST), "a split k range belongs to one tile");
    static_assert(BM == 128 || BM == 256, "128- or 256-row tiles");
    static_assert(NPROD == 4 || NPROD == 8, "four or eight loader waves");
    constexpr int NA = PC_NA;
    constexpr int BN = PC_BN;
    constexpr int WM = BM / 2, WN = 64;                  // matrix-wave tile
    constexpr int MT = WM / 16, NTL = WN / 16;           // 16 x 16 tiles per matrix wave: 8 x 4 or 4 x 4
    constexpr int HP = MT / 4;                           // phases per 32-deep MFMA group: 4 row tiles x 4 channel tiles = 16 MFMAs each
    constexpr int A_TILE_BYTES = BM * ROW_BYTES;
    constexpr int W_BASE = NA * A_TILE_BYTES;            // the two decoded weight tiles follow the activation tiles
    constexpr int NPT = 64 * NPROD;                      // loader threads
    constexpr int PA = (PC_ABL & 1) ? 0 : (BM * 8) / NPT;   // LDS-DMA pieces (1 KiB each) per loader wave and step
    constexpr int WPL = (BN * 16) / NPT;                 // packed weight bytes per loader lane and step: 8 or 4
    constexpr int CPL = WPL / 2;                         // 8-weight chunks (one ds_write_b128 each) per loader lane and step
    static_assert((BM * 8) % NPT == 0, "tile / loader mismatch");
    static_assert(3 * PA + 2 <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // pc_lds_bytes(BM)
#ifdef TVL_CLOCK_PROBE
    unsigned long long stamp_c[4], stamp_r[4];
    TVL_STAMP(0);
#endif

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int all_steps = k / BK;
    const int kb0 = SPLIT ? (int)blockIdx.y * steps_per_split : 0;   // first k step of this workgroup
    const int nk = SPLIT ? (all_steps - kb0 < steps_per_split ? all_steps - kb0 : steps_per_split) : all_steps;
    const int G = gridDim.x;
    const int my_tiles = PERSIST ? (nblk - (int)blockIdx.x + G - 1) / G : 1;
    const int total_steps = my_tiles * nk;

    if (wid >= 4) {
        // =================================== loader waves =====================================================================
        // Step t's loads (its packed weights, then its PA tile pieces) are issued three steps before the matrix waves multiply it;
        // its weights are decoded into weight tile t & 1 one step before.  Memory returns loads in issue order, so one counted
        // wait per step - everything but the newest 2 PA + 1 loads - says: tile t+1 has landed and the weights of step t+2 are in
        // their registers.
        const int pid = wid - 4, ptid = tid - 256;
        int vb_l = blockIdx.x, kb_l = 0;   // load cursor: tile (as a virtual block index) and k step the next loads fetch
        const TX* a_base;
        const uint8_t* b_base;
        uint32_t a_off[PA + 1], b_off;
        // lane -> (channel, part of the step's 16 packed bytes): 8 consecutive lanes take channels 0, 2, .., 14 of a group of 16, so
        // that their ds_write_b128 of one chunk index land on 8 different 16-byte slots of the swizzled image (conflict-free)
        const int ch_in = 2 * (lane & 7) + ((lane >> 3) & 1) + (NPROD == 4 ? 16 * ((lane >> 4) & 1) : 0);
        const int ch = pid * (BN / NPROD) + ch_in;
        const int part = NPROD == 4 ? lane >> 5 : lane >> 4;       // which WPL bytes of the 16
        auto set_load_tile = [&](int vb) __attribute__((always_inline)) {
            const int tile = xcd_remap(vb, nblk);
            const int lm0 = (tile / tiles_n) * BM, ln0 = (tile % tiles_n) * BN;
            a_base = x + (int64_t)lm0 * ld_x;
            b_base = pw + (int64_t)ln0 * ld_pw;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int p = i * NPT + ptid;
                const int row = p >> 3, cs = p & 7;
                const int chunk = cs ^ ((row >> 1) & 7);
                int gm = lm0 + row;
                gm = gm < m ? gm : m - 1;
                a_off[i] = (uint32_t)(((int64_t)(gm - lm0) * ld_x + chunk * 8) * (int64_t)sizeof(TX));
            }
            int gn = ln0 + ch;
            gn = gn < n ? gn : n - 1;
            b_off = (uint32_t)((int64_t)(gn - ln0) * ld_pw + WPL * part);
        };
        set_load_tile(vb_l);
        const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
        using wreg_t = typename std::conditional<WPL == 8, uint64_t, uint32_t>::type;
        // the cursor's step: this lane's packed weights into `w`, this wave's tile pieces into LDS tile `buf`; then advance the cursor
        // (to the next tile of this workgroup after a tile's last step; past the very last step it stays: harmless re-fetches into
        // buffers nobody reads again, so that the number of loads per step is constant)
        auto issue_loads = [&](int buf, wreg_t& w) __attribute__((always_inline)) {
            int kw = kb0 + kb_l;
            kw = kw >= w_steps ? kw - w_steps : kw;
            kw = kw >= w_steps ? kw - w_steps : kw;
            const uint8_t* wsrc = b_base + (int64_t)kw * (BK / 4);   // wave-uniform
            if constexpr (WPL == 8) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(w) : "v"(b_off), "s"(wsrc) : "memory");
            else asm volatile("global_load_dword %0, %1, %2" : "=v"(w) : "v"(b_off), "s"(wsrc) : "memory");
            const TX* sbase = a_base + (int64_t)(kb0 + kb_l) * BK;    // wave-uniform
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * A_TILE_BYTES + (i * NPT + pid * 64) * 16);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(a_off[i]), "s"(dst), "s"(sbase) : "memory");
            }
            if (++kb_l == nk) {
                if (PERSIST && vb_l + G < nblk) { vb_l += G; kb_l = 0; set_load_tile(vb_l); }
                else kb_l = nk - 1;
            }
        };
        // all but the newest KEEP loads of this wave have completed; `w` is named as completed by this wait, which orders its decode behind it
        auto wait_keep = [&](auto keep_c, wreg_t& w) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w) : "n"(decltype(keep_c)::value) : "memory");
        };
        const DecodeConsts dc = make_decode_consts();
        // byte offsets of this lane's CPL chunks inside a weight tile: row = channel, chunk index XOR-swizzled like an activation row
        uint32_t w_dst[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) w_dst[i] = (uint32_t)(ch * ROW_BYTES + (((part * CPL + i) ^ ((ch >> 1) & 7)) * 16));
        auto decode_store = [&](const wreg_t& w, int wbuf) __attribute__((always_inline)) {
            if constexpr (PC_ABL & 2) { asm volatile("" :: "v"(w)); return; }
            unsigned char* wt = lds + W_BASE + wbuf * PC_W_TILE_BYTES;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const uint32_t d = (WPL == 8 && i >= 2) ? (uint32_t)((uint64_t)w >> 32) : (uint32_t)w;
                const uint4 f = (i & 1) ? decode8<TX, 1>(d, dc) : decode8<TX, 0>(d, dc);
                *reinterpret_cast<uint4*>(wt + w_dst[i]) = f;
            }
        };
        auto publish = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's weight-tile stores have been performed
            __builtin_amdgcn_s_barrier();
        };
        using K1 = std::integral_constant<int, 3 * PA + 2>;   // after three groups: the first group's weight load has completed
        using K2 = std::integral_constant<int, 2 * PA + 1>;   // steady state: everything but the newest two tiles' pieces and the newest weights
        wreg_t wq0 = 0, wq1 = 0, wq2 = 0, wq3 = 0;
        issue_loads(0, wq0);
        issue_loads(1, wq1);
        issue_loads(2, wq2);
        wait_keep(K1{}, wq0);
        decode_store(wq0, 0);
        wait_keep(K2{}, wq1);          // tile 0 has landed, the weights of step 1 are in
        publish();                     // B(-1): step 0 is readable
        // iteration j (after barrier B(j-1), which retired step j-1): request step j+3 (tile buffer and weight set (j+3) % 4), decode
        // step j+1 into weight tile (j+1) & 1, then B(j): tile j+1 landed, weights of step j+2 in their registers
        for (int gs = 0; gs < total_steps; gs += 4) {
            issue_loads(3, wq3); decode_store(wq1, 1); wait_keep(K2{}, wq2); publish();
            if (gs + 1 < total_steps) { issue_loads(0, wq0); decode_store(wq2, 0); wait_keep(K2{}, wq3); publish(); }
            if (gs + 2 < total_steps) { issue_loads(1, wq1); decode_store(wq3, 1); wait_keep(K2{}, wq0); publish(); }
            if (gs + 3 < total_steps) { issue_loads(2, wq2); decode_store(wq0, 0); wait_keep(K2{}, wq1); publish(); }
        }
        // no LDS-DMA may outlive the workgroup's LDS allocation, no register may be freed while a load into it is in flight
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(wq0), "+v"(wq1), "+v"(wq2), "+v"(wq3) : : "memory");
        return;
    }

    // =================================== matrix waves ========================================================================
    if constexpr (SPLIT) y += (int64_t)blockIdx.y * m * ld_y;
    const int c = lane & 15, g = lane >> 4;
    const int wr = wid >> 1, wc = wid & 1;
    const int q = (0x2130 >> (4 * g)) & 3;     // lane group g owns k = 16 q .. 16 q + 15 of a step (keeps the swizzled reads conflict-free)
    int vb_c = blockIdx.x, kb_c = 0;           // the tile (virtual block index) and step being multiplied
    int m0, n0;
    {
        const int tile = xcd_remap(vb_c, nblk);
        m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) * BN;
    }
    const int sw = (c >> 1) & 7;               // (row >> 1) & 7 for rows 16 i + c of either operand's tile
    int col_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) col_off[s] = ((2 * q + s) ^ sw) * 16;
    const int a_row_off = (wr * WM + c) * ROW_BYTES;
    const int w_row_off = W_BASE + (wc * WN + c) * ROW_BYTES;

    floatx4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NTL; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};

    // activation fragments of row tiles 4 h .. 4 h + 3, MFMA group s, from LDS tile abuf
    auto read_a = [&](int abuf, int s, int h, uint4 (&af)[4]) __attribute__((always_inline)) {
        if constexpr (PC_ABL & 8) { for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(af[i].x), "+v"(af[i].y), "+v"(af[i].z), "+v"(af[i].w)); return; }
        const unsigned char* base = lds + abuf * A_TILE_BYTES + a_row_off + col_off[s] + h * 4 * 16 * ROW_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4*>(base + i * 16 * ROW_BYTES);
    };
    auto read_w = [&](int wbuf, int s, uint4 (&wf)[NTL]) __attribute__((always_inline)) {
        if constexpr (PC_ABL & 8) { for (int t = 0; t < NTL; ++t) asm volatile("" : "+v"(wf[t].x), "+v"(wf[t].y), "+v"(wf[t].z), "+v"(wf[t].w)); return; }
        const unsigned char* base = lds + wbuf * PC_W_TILE_BYTES + w_row_off + col_off[s];
#pragma unroll
        for (int t = 0; t < NTL; ++t) wf[t] = *reinterpret_cast<const uint4*>(base + t * 16 * ROW_BYTES);
    };
    auto mma = [&](const uint4 (&af)[4], int h, const uint4 (&wf)[NTL]) __attribute__((always_inline)) {
        if constexpr (PC_ABL & 4) { for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(af[i].x), "v"(af[i].w), "v"(wf[i].x), "v"(wf[i].w)); return; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NTL; ++t) acc[4 * h + i][t] = mfma16<TX>(wf[t], af[i], acc[4 * h + i][t]);
    };

    auto store_tile = [&]() __attribute__((always_inline)) {
        // lane (c, g) owns activation row 16 i + c and channels 16 t + 4 g + e.  16-bit outputs: v_permlane16_swap hands even lane groups
        // their neighbour's 4 channels of tile 0 and odd lane groups their neighbour's 4 channels of tile 1: 16 contiguous bytes per lane.
        const float cl = fabsf(clip);
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(ld_y * (int64_t)sizeof(TY))) & 15) == 0;
        const int nb = n0 + wc * WN;
        const int noff = (g & 1) ? 16 + 4 * (g - 1) : 4 * g;   // first of the 8 channels a lane stores after the swap
#pragma unroll
        for (int tp = 0; tp < NTL; tp += 2) {
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

    uint4 af0[4], af1[4], wf0[NTL], wf1[NTL];   // fragments of even / odd phases; decoded weights of MFMA group 0 / 1
    // Phase p of a step (MFMA group s = p / HP, row tiles 4 h .. 4 h + 3 with h = p % HP): request the NEXT phase's fragments, then
    // multiply.  The step's last phase first waits for every fragment of the step and passes the barrier: the next step's tiles
    // are then readable, this step's may be overwritten.
    auto phase = [&](auto p_c, auto abuf_c) __attribute__((always_inline)) {
        constexpr int p = decltype(p_c)::value, abuf = decltype(abuf_c)::value;
        constexpr int s = p / HP, h = p % HP;
        constexpr bool last = p == 2 * HP - 1;
        constexpr int sn = last ? 0 : (p + 1) / HP, hn = last ? 0 : (p + 1) % HP;
        constexpr int abufn = last ? (abuf + 1) % NA : abuf;
        constexpr int wbufn = abufn & 1;
        if constexpr (last) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if constexpr ((p & 1) == 0) read_a(abufn, sn, hn, af1); else read_a(abufn, sn, hn, af0);
        if constexpr (hn == 0) { if constexpr (sn == 0) read_w(wbufn, 0, wf0); else read_w(wbufn, 1, wf1); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((p & 1) == 0) { if constexpr (s == 0) mma(af0, h, wf0); else mma(af0, h, wf1); }
        else { if constexpr (s == 0) mma(af1, h, wf0); else mma(af1, h, wf1); }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](auto abuf_c) __attribute__((always_inline)) {
        phase(std::integral_constant<int, 0>{}, abuf_c);
        phase(std::integral_constant<int, 1>{}, abuf_c);
        if constexpr (HP == 2) {
            phase(std::integral_constant<int, 2>{}, abuf_c);
            phase(std::integral_constant<int, 3>{}, abuf_c);
        }
        if constexpr (PERSIST) {
            if (++kb_c == nk) {   // last step of a tile: write it out and start the next one (its first fragments are in af0 / wf0)
                store_tile();
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int t = 0; t < NTL; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
                kb_c = 0;
                vb_c += G;
                const int tile = xcd_remap(vb_c < nblk ? vb_c : nblk - 1, nblk);
                m0 = (tile / tiles_n) * BM; n0 = (tile % tiles_n) * BN;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    __builtin_amdgcn_s_barrier();   // B(-1): step 0 is readable
    read_a(0, 0, 0, af0);
    read_w(0, 0, wf0);
    TVL_STAMP(1);
    for (int gs = 0; gs < total_steps; gs += 4) {
        step(std::integral_constant<int, 0>{});
        if (gs + 1 < total_steps) step(std::integral_constant<int, 1>{});
        if (gs + 2 < total_steps) step(std::integral_constant<int, 2>{});
        if (gs + 3 < total_steps) step(std::integral_constant<int, 3>{});
    }
    TVL_STAMP(2);
    if constexpr (!PERSIST) store_tile();
#ifdef TVL_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave's queue
    TVL_STAMP(3);
    if (tid == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { tvl_clock_probe[8 * blockIdx.x + i] = stamp_c[i]; tvl_clock_probe[8 * blockIdx.x + 4 + i] = stamp_r[i]; }
    }
#endif
}

}  // namespace tvl
