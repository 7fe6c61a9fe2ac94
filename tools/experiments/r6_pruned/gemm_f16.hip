// K3 — C = act(A . W^T + bias): the per-sample 1x1 convolutions of the render path as one MFMA GEMM.
//
// Replaces nn.Conv2d(1x1) x 13 calls (/root/reference models/CoPoNeRF.py:387-397, 404, 408, 446, 473):
// 835->832 (ReLU) ->416, 832->416, 832->128 (ReLU) ->128, 128->128 — 99.7 % of the path's FLOPs.
//
// gfx950 design
//   * v_mfma_f32_16x16x32_f16, fp16 operands, fp32 accumulate.  The N sizes of this network are 13 x 64 / 13 x 32 /
//     128, so the workgroup tile is 256 (M) x 16*NT (N) with NT = 13 (N = 832, 416) or 8 (N = 128); 32x32 tiles
//     would need per-wave N splits that 13 does not allow without 7.7 % zero padding.
//   * 8 waves = 512 threads, wave w owns rows [32w, 32w+32) x all NT column tiles.  Everything lives in the VGPR
//     file (2*NT*4 accumulators + two fragment sets, 229 registers): two waves per SIMD, no VGPR<->AGPR copies
//     (one-wave-per-SIMD variants with 4*NT*4 accumulators were measured slower — the allocator shuttles the
//     accumulators between the two register files every K step).
//   * operands are swapped (MFMA A-operand = weights, B-operand = activations) so a lane ends up holding 4
//     CONSECUTIVE output columns of one row; fp16 results are staged through LDS and leave as 16-byte
//     row-contiguous stores covering whole 416-B rows.
//   * both tiles are K-contiguous (activations (M,K), weights (N,K)) and are staged by buffer_load_dwordx4 ... lds
//     (LDS-DMA, no VGPR round trip; the descriptor's bounds check zero-fills rows past M and reads past the end)
//     as [rows][64] fp16 images.  The LDS image is lane-linear, so the bank-conflict XOR swizzle is applied to the
//     per-lane SOURCE offset (chunk c of row r lives at physical chunk c ^ ((r >> 1) & 7)) and to the fragment
//     reads: every ds_read_b128 lane group touches 16 distinct 16-B slots.
//   * asymmetric ring: 3 slots for activations (they stream from HBM, prefetched TWO stages ahead) and 2 for
//     weights (L2-resident, one stage ahead) = 96 + 53 KiB; a symmetric 3-slot ring does not fit 160 KiB.
//     The barrier is a raw s_barrier with a counted s_waitcnt vmcnt(UA4): DMA completes in issue order, so all but
//     the youngest activation pieces must have landed — __syncthreads() would drain vmcnt to 0 at every K step.
//   * two-phase software pipeline per 64-deep stage: each block of 2*NT MFMAs runs on fragments that were read
//     from LDS one phase earlier; the barrier sits between two MFMA blocks whose operands are already in
//     registers.  The DMA pieces of the next stages are issued ONE PER MFMA PAIR inside phase B (sched_barrier
//     pinned): back to back behind the barrier they stall both waves of a SIMD at once and its MFMA pipe idles.
//   * persistent workgroups (one per CU) walk the tile list; the first stage of the NEXT tile is issued before the
//     epilogue of the current one, which stages C in the two activation slots that stage does not use, so the
//     DMA latency of a tile start and the C-store phase overlap instead of adding up (875 -> 919 TFLOP/s).
//   * XCD-aware tile order: each XCD (private L2) owns a contiguous range of logical tiles and its workgroups take
//     them in lock step with the N index fastest -> the N tiles sharing one 256-row activation tile run together
//     on one L2.  rocprofv3 still shows 1.46x the algorithmic activation bytes fetched (profiles/r01_v4_traffic.json).
// Roofline: at K ~ N ~ 832 the GEMM sits on the ridge of the MI355X roofline (2*K*N/(2*(K+N)) = 416 FLOP per HBM
// byte vs 2500 TFLOP/s / 6.3 TB/s = 397): it is bounded by MFMA issue AND by streaming the (M,K) input and (M,N)
// output once.  Algorithmic FLOPs per launch = 2*M*N*K.  Measured on the 835->832 layer at M = 4.19 M rows:
// 915 TFLOP/s = 36.6 % of the 2.5 PFLOP/s spec peak, SQ_VALU_MFMA_BUSY_CYCLES = 48.9 % of the SIMD cycles (the
// clock sits near 1.85 GHz under this load); history 604 -> 765 -> 828 -> 875 -> 919 (profiles/, tools/gemm_k.py).
#include <algorithm>

#include "common.h"

namespace {

constexpr int BM = 256;
constexpr int BK = 64;                    // halves per K step = 128 B per row
constexpr int ROW_BYTES = BK * 2;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int NT>
struct Cfg {
    static constexpr int BN = NT * 16;
    static constexpr int UNITS_A = BM / 8;            // a unit = 8 rows x 128 B = one wave-wide 1 KiB DMA
    static constexpr int UNITS_B = BN / 8;
    static constexpr int UNITS = UNITS_A + UNITS_B;
    static constexpr int A_BYTES = BM * ROW_BYTES;      // one stage of activations
    static constexpr int B_BYTES = BN * ROW_BYTES;      // one stage of weights
    static constexpr int A_SLOTS = 3;                   // activations stream from HBM: prefetched two stages ahead
    static constexpr int B_SLOTS = 2;                   // weights are L2-resident: one stage ahead is enough
    // 26 weight units over 8 waves leave 6 waves one piece short: those issue an out-of-bounds (zero-fill, no
    // memory traffic) piece into a per-wave 1 KiB dump instead of branching, so the DMA stays in one basic block
    static constexpr int DUMP_BYTES = (UNITS_B % 8) ? 8 * 1024 : 0;
    static constexpr int LDS_BYTES = A_SLOTS * A_BYTES + B_SLOTS * B_BYTES + DUMP_BYTES;
};

// Persistent workgroups: the grid is one workgroup per CU and each walks a list of output tiles.  Workgroups are
// dispatched round-robin over the 8 XCDs (private L2 each): XCD x owns a contiguous range of logical tiles and its
// workgroups take them in lock step with the N index fastest, so the n_tiles workgroups that share one 256-row
// activation tile run at the same time on the same L2 instead of re-streaming the tile from HBM.
struct TileWalk {
    int begin, end, step, cur;      // logical tiles [begin, end) of this XCD, this workgroup takes begin+i, +step, ...
};
__device__ __forceinline__ TileWalk tile_walk(int total) {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    const int wg_x = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);            // workgroups on this XCD
    const int q = total >> 3, r = total & 7;
    TileWalk t;
    t.begin = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    t.end = t.begin + q + (xcd < r ? 1 : 0);
    t.step = wg_x;
    t.cur = t.begin + idx;
    return t;
}

// ---------------------------------------------------------------------------------------------
// The kernel (design notes at the top of the file).
// ---------------------------------------------------------------------------------------------
// DOT = 1: instead of storing C, store logits[m] = <fp16(C[m, :]), Q[m, :]> (one workgroup tile must span all N columns):
// the consumer of key_map_2 only needs that row dot product (attention logits), 4 bytes per row instead of 256.
// DOT = 2: additionally chain a second 128 -> 128 layer in the epilogue, C2 = W2 . fp16(act(C)) + b2, and dot THAT with
// Q (key_map -> ReLU -> key_map_2 -> logit in one kernel).  The accumulator layout of the first layer (lane = row,
// columns nt*16 + fk*4 + 0..3) serves directly as the MFMA B operand of K block p = tile pair (2p, 2p+1); the W2
// fragments are laid out in LDS for exactly that k order (k = 32p + fk*4 + e, 32p + 16 + fk*4 + e).
// DOT = 3 ("combine", training): C is the data gradient of the folded key map, d(hid) through the key path, and the epilogue
// finishes the gradient of the first layer's pre-activation while the tile is still on chip,
//     out[row, c] = hid[row, c] > 0 ? fp16(C[row, c]) + w1[row] * dh1[ray, c] + w2[row] * dh2[ray, c] : 0
// i.e. what cpn_hid_grad_combine computes from the stored C (a 7 GB write + a 7 GB read less per training step).  Q = hid
// (row stride ldq = ldc).  The read-back of the staged tile gives a lane ONE 8-column chunk (lane % 26, 52 live lanes, two rows
// per instruction), so the ray's two dhbar chunks are loaded once per tile and stay in registers.
struct CombineArgs {
    const float* w1; const float* dh1; const float* w2; const float* dh2;      // (N, R, S) weights, (rays, 1664) fp32 dhbar
    int V, R, S, ray0;
    int accum;                                                                   // OUT_F32: C += (the product + bias), then the ReLU
};
#ifndef CPN_HID_READ_NT
#define CPN_HID_READ_NT 1
#endif
template <int NT, bool OUT_F32, bool RELU, int DOT = 0>
__global__ __launch_bounds__(512) void gemm_f16_kernel(const __half* __restrict__ A, int lda,
                                                             const __half* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             void* __restrict__ Cv, int ldc, int M, int K32, int n_tiles,
                                                             int total_tiles, const __half* __restrict__ Q = nullptr,
                                                             int ldq = 0, const __half* __restrict__ W2 = nullptr,
                                                             int ldw2 = 0, const float* __restrict__ bias2 = nullptr,
                                                             CombineArgs ca = CombineArgs{}) {
    using C_ = Cfg<NT>;
    // the chained key kernel streams hid (7 GB per chunk, read once per pass): non-temporal, so that the node tables
    // of encode_hidden stay in L2 / Infinity Cache across the chunk loop
    constexpr int A_AUX = (DOT == 2 && CPN_HID_READ_NT) ? 2 : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = (K32 + 1) >> 1;
    const int nfull = K32 >> 1;

    // DMA (buffer_load ... lds): wave w moves units w, w+8, ...; unit u = image rows [8u, 8u+8).  Rows advance by
    // 64 per round, so the source swizzle ((row>>1)&7) and hence the per-lane byte offset are round-invariant:
    // ONE voffset VGPR per operand, everything else in the scalar offset.  The A descriptor ends at row
    // min(BM, M-m0): rows past M read as zero (hardware bounds check) instead of being clamped.
    constexpr int UA4 = C_::UNITS_A / 8;
    constexpr int UB = (C_::UNITS_B + 7) / 8;
    constexpr int NDMA = UA4 + UB;
    static_assert(NDMA <= NT, "one DMA piece per MFMA pair");
    const int r0 = wave * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((r0 >> 1) & 7);
    const int voff_a = (r0 * lda + lchunk * 8) * 2;
    const int voff_w = (r0 * ldw + lchunk * 8) * 2;
    char* const smem_b = smem + C_::A_SLOTS * C_::A_BYTES;
    char* const dump = smem_b + C_::B_SLOTS * C_::B_BYTES + wave * 1024;
    const int w_bytes = C_::BN * ldw * 2;

    // one DMA piece = one wave-wide 1 KiB buffer_load...lds
    auto piece_a = [&](const __amdgpu_buffer_rsrc_t& ra, int kt, int slot, int i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(smem + slot * C_::A_BYTES + wave * 1024 + i * 8192),
                                                 16, voff_a, kt * BK * 2 + i * 128 * lda, 0, A_AUX);
    };
    auto piece_b = [&](const __amdgpu_buffer_rsrc_t& rw, int kt, int slot, int i) {
        char* sbase = smem_b + slot * C_::B_BYTES + wave * 1024;
        const int kbytes = kt * BK * 2;
        if ((i + 1) * 8 <= C_::UNITS_B) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(sbase + i * 8192), 16, voff_w,
                                                     kbytes + i * 128 * ldw, 0, 0);
        } else {
            const bool live = wave + 8 * i < C_::UNITS_B;
            char* dst = live ? sbase + i * 8192 : dump;
            const int soff = live ? kbytes + i * 128 * ldw : w_bytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)dst, 16, voff_w, soff, 0, 0);
        }
    };
    auto stage_a = [&](const __amdgpu_buffer_rsrc_t& ra, int kt, int slot) {
#pragma unroll
        for (int i = 0; i < UA4; ++i) piece_a(ra, kt, slot, i);
    };
    auto stage_b = [&](const __amdgpu_buffer_rsrc_t& rw, int kt, int slot) {
#pragma unroll
        for (int i = 0; i < UB; ++i) piece_b(rw, kt, slot, i);
    };
    auto rsrc_of_a = [&](int m0) {
        const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * lda), 0, rows_valid * lda * 2, 0x00020000);
    };
    auto rsrc_of_w = [&](int n0) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(W + (size_t)n0 * ldw), 0, w_bytes, 0x00020000);
    };

    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int swz = (frow >> 1) & 7;
    const int xoff = (wave * 32 + frow) * ROW_BYTES;
    const int woff = frow * ROW_BYTES;
    const int coff0 = ((fk ^ swz) << 4), coff1 = (((4 + fk) ^ swz) << 4);

    auto load_frags = [&](const char* sa, const char* sb, int coff, half8 (&xa)[2], half8 (&wb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            xa[mt] = *reinterpret_cast<const half8*>(sa + xoff + mt * 16 * ROW_BYTES + coff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wb[nt] = *reinterpret_cast<const half8*>(sb + woff + nt * 16 * ROW_BYTES + coff);
    };

    // Software pipeline per 64-deep stage:
    //   phase A  MFMAs on fragment set 0 (stage kt, first k32)  ||  ds_reads of set 1 (stage kt, second k32)
    //   barrier  -> stage kt+1 has landed for every wave, every wave is done reading stage kt
    //   phase B  MFMAs on set 1  ||  ds_reads of set 0 for stage kt+1  ||  DMA of weights(kt+2), activations(kt+3)
    // so the barrier sits between two MFMA blocks whose operands are already in registers.
#define CPN_INTERLEAVE_READS_MFMA()                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < NT; ++q_) {                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* 1 DS read */      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); /* 2 MFMA    */      \
    }                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);

    half8* const w2l = reinterpret_cast<half8*>(smem + C_::LDS_BYTES);       // DOT == 2: [tile t][k block p][lane]
    if constexpr (DOT == 2) {
        for (int i = tid; i < NT * 4 * 64; i += 512) {
            const int l = i & 63, p = (i >> 6) & 3, t = i >> 8;
            const __half* src = W2 + (size_t)(t * 16 + (l & 15)) * ldw2 + p * 32 + (l >> 4) * 4;
            const half4 lo = *reinterpret_cast<const half4*>(src), hi = *reinterpret_cast<const half4*>(src + 16);
            w2l[i] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        __syncthreads();
    }
    TileWalk walk = tile_walk(total_tiles);
    if (walk.cur >= walk.end) return;
    // stage 0 of the first tile; later tiles get theirs issued under the previous tile's epilogue
    {
        const int t = walk.cur;
        stage_a(rsrc_of_a((t / n_tiles) * BM), 0, 0);
        stage_b(rsrc_of_w((t % n_tiles) * C_::BN), 0, 0);
    }

    for (; walk.cur < walk.end; walk.cur += walk.step) {
        const int m0 = (walk.cur / n_tiles) * BM;
        const int n0 = (walk.cur % n_tiles) * C_::BN;
        const __amdgpu_buffer_rsrc_t rsrc_a = rsrc_of_a(m0);
        const __amdgpu_buffer_rsrc_t rsrc_w = rsrc_of_w(n0);

        f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        half8 xa0[2], wb0[NT], xa1[2], wb1[NT];
        half4 qv[(DOT == 1 || DOT == 2) ? 2 : 1][(DOT == 1 || DOT == 2) ? NT : 1];   // DOT 1/2: the Q rows of this tile, in flight under the main loop
        f32x4 cdh[DOT == 3 ? 4 : 1];                       // DOT 3: dh1 / dh2 chunks (8 columns) of this wave's ray
        if constexpr (DOT == 3) {
            constexpr int CPR = C_::BN / 8;
            const int T = ca.S > 0 ? ca.V * ca.S : 1;      // S == 0: mask only (no parked parts, dh pointers are null)
            int t = (m0 + wave * 32) / T;                  // 32 | T: the wave's 32 rows belong to one ray
            const int tl = (M - 1) / T;
            t = t < tl ? t : tl;
            const int c = lane % CPR;
            const size_t off = (size_t)t * ldc + n0 + c * 8;
            cdh[0] = ca.dh1 ? *reinterpret_cast<const f32x4*>(ca.dh1 + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            cdh[1] = ca.dh1 ? *reinterpret_cast<const f32x4*>(ca.dh1 + off + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            cdh[2] = ca.dh2 ? *reinterpret_cast<const f32x4*>(ca.dh2 + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            cdh[3] = ca.dh2 ? *reinterpret_cast<const f32x4*>(ca.dh2 + off + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (DOT == 1 || DOT == 2) {
#ifndef CPN_ROWDOT_NOQ                                     /* timing-only ablation: the Q rows never loaded */
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = m0 + wave * 32 + mt * 16 + (lane & 15);
                if (ldq == 0) {
                    // Q in fragment order (CPN_ROWS_FRAG; N = 128, n0 = 0): [16-row group][32-column block p][lane = row + 16 * 8-column
                    // group][8 halves].  The 4 columns nt*16 + g*4 .. of this lane sit at p = nt >> 1, 8-column group (nt & 1) * 2 +
                    // (g >> 1), half (g & 1): the 64 lanes of one load cover 512 contiguous bytes (row-major rows: 16 x 8 bytes
                    // from 16 different rows, 64 L1 tag look-ups per instruction, 0.48 of this kernel's 1.13 ms per image)
                    const int g = lane >> 4;
                    const int last = (M - 1) >> 4;
                    int grp = (m0 + wave * 32 + mt * 16) >> 4;
                    grp = grp < last ? grp : last;
                    const __half* qb = Q + (size_t)grp * (4 * 64 * 8) + ((g >> 1) * 16 + (lane & 15)) * 8 + (g & 1) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        qv[mt][nt] = *reinterpret_cast<const half4*>(qb + ((nt >> 1) * 64 + (nt & 1) * 32) * 8);
                } else {
                    const __half* qrow = Q + (size_t)(m < M ? m : M - 1) * ldq + n0 + (lane >> 4) * 4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) qv[mt][nt] = *reinterpret_cast<const half4*>(qrow + nt * 16);
                }
            }
#else
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) qv[mt][nt] = half4{(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
#endif
        }
        // slower waves may still be reading the previous tile's C staging out of activation slots 1-2
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nk > 1) { stage_a(rsrc_a, 1, 1); stage_b(rsrc_w, 1, 1); }
        // stage 0 landed (and the previous tile's stores have left: on gfx9 they share vmcnt with the loads and may
        // retire out of order against them, so the count cannot single out the DMA); stage 1 may stay in flight
        if (nk > 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nk > 2) stage_a(rsrc_a, 2, 2);
        load_frags(smem, smem_b, coff0, xa0, wb0);

        // Main part: stages kt+2 (weights) and kt+3 (activations) exist -> branch-free body.
        //  * DMA pieces are spread between the MFMAs of phase B, one per MFMA pair: issued back to back right behind
        //    the barrier they stall BOTH waves of a SIMD at the same moment and its MFMA pipe idles.
        //  * raw s_barrier instead of __syncthreads(): the fence of __syncthreads() makes the compiler drain vmcnt
        //    to 0, which would serialise the two-stage-ahead activation prefetch behind every barrier.  What the
        //    barrier must order is spelled out instead: this wave's DMA of stage kt+1 has landed (in-order
        //    completion: all but the youngest UA4 pieces), and its ds_reads of stage kt have returned before another
        //    wave's DMA reuses the slot.
        int sa = 0;                                        // activation slot of stage kt (kt mod 3)
        int kt = 0;
        for (; kt + 3 < nk && kt < nfull; ++kt) {
            const int sa1 = (sa == 2) ? 0 : sa + 1;
            const char* acur = smem + sa * C_::A_BYTES;
            const char* anxt = smem + sa1 * C_::A_BYTES;
            const char* bcur = smem_b + (kt & 1) * C_::B_BYTES;
            const char* bnxt = smem_b + ((kt + 1) & 1) * C_::B_BYTES;
            load_frags(acur, bcur, coff1, xa1, wb1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[nt], xa0[mt], acc[mt][nt], 0, 0, 0);
            CPN_INTERLEAVE_READS_MFMA();
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(UA4) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                wb0[nt] = *reinterpret_cast<const half8*>(bnxt + woff + nt * 16 * ROW_BYTES + coff0);
                if (nt < 2) xa0[nt] = *reinterpret_cast<const half8*>(anxt + xoff + nt * 16 * ROW_BYTES + coff0);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[nt], xa1[0], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[nt], xa1[1], acc[1][nt], 0, 0, 0);
                if (nt < UB) piece_b(rsrc_w, kt + 2, kt & 1, nt);              // weight slot of stage kt just drained
                else if (nt < NDMA) piece_a(rsrc_a, kt + 3, sa, nt - UB);      // and so was its activation slot
                __builtin_amdgcn_sched_barrier(0);
            }
            sa = sa1;
        }
        // tail: the last stages, nothing (or only weights) left to fetch
        for (; kt < nfull; ++kt) {
            const int sa1 = (sa == 2) ? 0 : sa + 1;
            const char* acur = smem + sa * C_::A_BYTES;
            const char* anxt = smem + sa1 * C_::A_BYTES;
            const char* bcur = smem_b + (kt & 1) * C_::B_BYTES;
            const char* bnxt = smem_b + ((kt + 1) & 1) * C_::B_BYTES;
            load_frags(acur, bcur, coff1, xa1, wb1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[nt], xa0[mt], acc[mt][nt], 0, 0, 0);
            CPN_INTERLEAVE_READS_MFMA();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage_b(rsrc_w, kt + 2, kt & 1);
            load_frags(anxt, bnxt, coff0, xa0, wb0);       // harmless garbage after the last stage
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[nt], xa1[mt], acc[mt][nt], 0, 0, 0);
            CPN_INTERLEAVE_READS_MFMA();
            sa = sa1;
        }
        if (K32 & 1) {                                     // odd trailing k32 step (already in set 0)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[nt], xa0[mt], acc[mt][nt], 0, 0, 0);
        }

        // every wave is done with the ring: stage 0 of the NEXT tile starts now and flies under this tile's epilogue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int t = walk.cur + walk.step;
            if (t < walk.end) {
                stage_a(rsrc_of_a((t / n_tiles) * BM), 0, 0);
                stage_b(rsrc_of_w((t % n_tiles) * C_::BN), 0, 0);
            }
        }

        if constexpr (DOT == 3) {
            constexpr int RS = C_::BN * 2 + 16;
            constexpr int CPR = C_::BN / 8;
            constexpr int RPI = 64 / CPR;                              // rows per store instruction (2 at BN = 208, 4 at 128)
            constexpr int NI = 16 / RPI;
            static_assert(8 * 16 * RS <= 2 * C_::A_BYTES, "staging must fit activation slots 1-2");
            static_assert(RPI >= 1 && NI * RPI == 16, "whole rows per instruction");
            char* cw = smem + C_::A_BYTES + wave * (16 * RS);
            const int c = lane % CPR, rsel = lane / CPR;               // rsel >= RPI: idle lanes (12 of 64 at BN = 208)
            const int rl = rsel < RPI ? rsel : 0;
            const bool parts = ca.S > 0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int rbase = m0 + wave * 32 + mt * 16;
                // loads first: the mask chunks and the softmax weights of this lane's rows
                const int rb = rbase < M ? rbase : M - 16;             // M % 16 == 0 (checked by the entry points)
                size_t wbase = 0;
                if (parts) {
                    const int T = ca.V * ca.S;
                    const int t = rb / T, rem = rb - t * T;
                    const int v = rem / ca.S, s0 = rem - v * ca.S;
                    const int ray = ca.ray0 + t;
                    const int b = ray / ca.R, rr = ray - b * ca.R;
                    wbase = (((size_t)(b * ca.V + v)) * ca.R + rr) * ca.S + s0;
                }
                half8 hv[NI];
                float wa[NI], wb_[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = i * RPI + rl;
                    hv[i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(Q + (size_t)(rb + r) * ldq + n0 + c * 8));
                    wa[i] = ca.w1 ? ca.w1[wbase + r] : 0.0f;
                    wb_[i] = ca.w2 ? ca.w2[wbase + r] : 0.0f;
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    half4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = (_Float16)acc[mt][nt][i];
                    *reinterpret_cast<half4*>(cw + (lane & 15) * RS + (nt * 16 + (lane >> 4) * 4) * 2) = h;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __half* cbase = (__half*)Cv + (size_t)rbase * ldc + n0 + c * 8;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = i * RPI + rl;
                    const half8 d = *reinterpret_cast<const half8*>(cw + r * RS + c * 16);
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float a = (float)d[e];
                        a += wa[i] * cdh[e >> 2][e & 3];
                        a += wb_[i] * cdh[2 + (e >> 2)][e & 3];
                        o[e] = (float)hv[i][e] > 0.0f ? (_Float16)a : (_Float16)0.0f;
                    }
                    if (rsel < RPI && rbase + r < M) *reinterpret_cast<half8*>(cbase + (size_t)r * ldc) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        } else if constexpr (DOT != 0) {
            float dsum[2] = {0.0f, 0.0f};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = m0 + wave * 32 + mt * 16 + (lane & 15);
                f32x4 v[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    v[nt] = acc[mt][nt] + *reinterpret_cast<const f32x4*>(bias + n0 + nt * 16 + (lane >> 4) * 4);
                    if (RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[nt][i] = fmaxf(v[nt][i], 0.0f);
                    }
                }
                if constexpr (DOT == 2) {
                    half8 hb[NT / 2];
#pragma unroll
                    for (int p = 0; p < NT / 2; ++p)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            hb[p][i] = (_Float16)v[2 * p][i];
                            hb[p][4 + i] = (_Float16)v[2 * p + 1][i];
                        }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        f32x4 o2 = *reinterpret_cast<const f32x4*>(bias2 + t * 16 + (lane >> 4) * 4);
#pragma unroll
                        for (int p = 0; p < NT / 2; ++p)
                            o2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[(t * 4 + p) * 64 + lane], hb[p], o2, 0, 0, 0);
                        v[t] = o2;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) dsum[mt] += (float)(_Float16)v[nt][i] * (float)qv[mt][nt][i];
                dsum[mt] += __shfl_xor(dsum[mt], 16);
                dsum[mt] += __shfl_xor(dsum[mt], 32);
                if (lane < 16 && m < M) ((float*)Cv)[m] = dsum[mt];
            }
        } else if constexpr (!OUT_F32) {
            // fp16 epilogue through LDS (activation slots 1 and 2; slot 0 is receiving the next tile): the accumulator
            // layout gives a lane 4 consecutive columns (8 B) of one row, i.e. 32-B row segments per store; staging
            // 16 x BN per wave and reading it back row-contiguously turns them into 16-byte stores that cover whole
            // 416-B rows.  Two halves of 16 rows keep the staging inside 64 KiB.
            constexpr int RS = C_::BN * 2 + 16;                       // padded row stride: conflict-free ds_write_b64
            constexpr int CPR = C_::BN / 8;                           // 16-byte chunks per row
            static_assert(8 * 16 * RS <= 2 * C_::A_BYTES, "staging must fit activation slots 1-2");
            char* cw = smem + C_::A_BYTES + wave * (16 * RS);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + nt * 16 + (lane >> 4) * 4);
                    f32x4 v = acc[mt][nt] + bv;
                    if (RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                    }
                    half4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                    *reinterpret_cast<half4*>(cw + (lane & 15) * RS + (nt * 16 + (lane >> 4) * 4) * 2) = h;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int rbase = m0 + wave * 32 + mt * 16;
                __half* cbase = (__half*)Cv + (size_t)rbase * ldc + n0;
#pragma unroll
                for (int i = 0; i < (16 * CPR + 63) / 64; ++i) {
                    const int q = lane + 64 * i;
                    const int r = q / CPR, c = q - r * CPR;
                    if (q < 16 * CPR && rbase + r < M) {
                        const half8 val = *reinterpret_cast<const half8*>(cw + r * RS + c * 16);
                        *reinterpret_cast<half8*>(cbase + (size_t)r * ldc + c * 8) = val;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + nt * 16 + (lane >> 4) * 4;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int m = m0 + wave * 32 + mt * 16 + (lane & 15);
                    if (m >= M) continue;
                    f32x4 v = acc[mt][nt] + bv;
                    if (ca.accum) v += *reinterpret_cast<const f32x4*>((const float*)Cv + (size_t)m * ldc + n);
                    if (RELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                    }
                    *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
                }
            }
        }
    }
#undef CPN_INTERLEAVE_READS_MFMA
}

template <int NT, bool OUT_F32, bool RELU>
int launch(const __half* A, int lda, const __half* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
           int K32, hipStream_t stream, int accum = 0) {
    using C_ = Cfg<NT>;
    const size_t lds = C_::LDS_BYTES;
    auto kern = gemm_f16_kernel<NT, OUT_F32, RELU>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            cpn_set_error("cpn_gemm_f16: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int n_tiles = N / C_::BN;
    const long long total = (long long)cpn_cdiv(M, BM) * n_tiles;
    if (total >= (1LL << 31)) {
        cpn_set_error("cpn_gemm_f16: %lld output tiles exceed the 32-bit tile index", total);
        return CPN_E_SHAPE;
    }
    const int num_cu = cpn_stream_cus((void*)stream);       // persistent grid: the CUs this stream may use
    dim3 grid((unsigned)std::min<long long>(total, num_cu));       // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles, (int)total,
                       (const __half*)nullptr, 0, (const __half*)nullptr, 0, (const float*)nullptr,
                       CombineArgs{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, accum});
    CPN_LAUNCH_CHECK("cpn_gemm_f16");
    return 0;
}

// row-dot launches: N == 16*NT (one tile spans the row), logits (M) fp32 out; CHAIN adds the second 128 -> 128 layer
template <int NT, bool RELU, int DOT>
int launch_rowdot(const __half* A, int lda, const __half* W, int ldw, const float* bias, const __half* Q, int ldq,
                  float* logits, int M, int K32, const __half* W2, int ldw2, const float* bias2, hipStream_t stream) {
    using C_ = Cfg<NT>;
    const size_t lds = C_::LDS_BYTES + (DOT == 2 ? NT * 4 * 64 * 16 : 0);
    auto kern = gemm_f16_kernel<NT, false, RELU, DOT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            cpn_set_error("cpn_gemm_f16_rowdot: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const long long total = cpn_cdiv(M, BM);
    const int num_cu = cpn_stream_cus((void*)stream);       // persistent grid: the CUs this stream may use
    dim3 grid((unsigned)std::min<long long>(total, num_cu));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, A, lda, W, ldw, bias, (void*)logits, 0, M, K32, 1, (int)total, Q,
                       ldq, W2, ldw2, bias2, CombineArgs{});
    CPN_LAUNCH_CHECK("cpn_gemm_f16_rowdot");
    return 0;
}

template <int NT>
int launch_combine(const __half* A, int lda, const __half* W, int ldw, const __half* hid, int ldh, const CombineArgs& ca,
                   __half* out, int ld, int M, int N, int K32, hipStream_t stream) {
    using C_ = Cfg<NT>;
    const size_t lds = C_::LDS_BYTES;
    auto kern = gemm_f16_kernel<NT, false, false, 3>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            cpn_set_error("cpn_gemm_f16_combine: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int n_tiles = N / C_::BN;
    const long long total = (long long)cpn_cdiv(M, BM) * n_tiles;
    if (total >= (1LL << 31)) {
        cpn_set_error("cpn_gemm_f16_combine: %lld output tiles exceed the 32-bit tile index", total);
        return CPN_E_SHAPE;
    }
    const int num_cu = cpn_stream_cus((void*)stream);
    dim3 grid((unsigned)std::min<long long>(total, num_cu));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, A, lda, W, ldw, (const float*)nullptr, (void*)out, ld, M, K32, n_tiles,
                       (int)total, hid, ldh, (const __half*)nullptr, 0, (const float*)nullptr, ca);
    CPN_LAUNCH_CHECK("cpn_gemm_f16_combine");
    return 0;
}

template <int NT>
int dispatch(const __half* A, int lda, const __half* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
             int K32, int relu, int out_f32, hipStream_t s) {
    if (out_f32) {                                            // 2: accumulate onto C (fp32) before the ReLU
        return relu ? launch<NT, true, true>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s, out_f32 == 2)
                    : launch<NT, true, false>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s, out_f32 == 2);
    }
    return relu ? launch<NT, false, true>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s)
                : launch<NT, false, false>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s);
}

// ---------------------------------------------------------------------------------------------
// Few rows (the per-RAY GEMM of a small call: value_fold at M = 3 641 rays, 1/18 of an image).  The 256-row tiles above leave
// 30 workgroups on 256 CUs walking K = 1664 as 26 barrier-separated stages: 35 us for 5 GFLOP.  Here a wave owns 32 rows x
// CPN_FEW_NT column tiles, no LDS staging, no barrier, CPN_FEW_DEPTH 64-deep stages in flight (pinned with sched_barriers: left
// alone the scheduler sinks every load to just above its use).  What it took (tools/fewrows_bench.py, us at M = 3 641):
//   * operands read in the MFMA fragment layout from row-major matrices (lane = row + 16 * piece: 64 tag look-ups per
//     instruction): 52-62 - the wave waits on the L1 tag pipe, not on L2;
//   * weights pre-packed in FRAGMENT order (cpn_pack_gemm_frags: the 1 KiB of A operand (tile t, k32 step ks) contiguous,
//     a load instruction covers 8 whole lines): 30;
//   * the activation rows in the load layout as well (lane = 4 * row + piece: 16 look-ups), moved to the fragment layout with
//     4 ds_bpermute per register set: 22 (7 tiles per wave, 2-3 stages in flight), **17** (4 tiles, 4 stages) - the default.
// The accumulation order of every output element is that of gemm_f16_kernel (zero start, k32 steps in order, bias last):
// bit-identical results, so a ray's value does not depend on the size of the call it is rendered in.  Above ~ 6 000 rows the
// tiled kernel wins again (weights re-read per 32 rows).
// ---------------------------------------------------------------------------------------------
__global__ void pack_gemm_frags_kernel(const __half* __restrict__ W, int ldw, int N, int K32, half8* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // (tile, k32 step, lane)
    const long long total = (long long)(N >> 4) * K32 * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const long long q = i >> 6;
    const int ks = (int)(q % K32), t = (int)(q / K32);
    out[i] = *reinterpret_cast<const half8*>(W + (size_t)(t * 16 + (lane & 15)) * ldw + ks * 32 + (lane >> 4) * 8);
}

#ifndef CPN_FEW_NT
#define CPN_FEW_NT 4
#endif
#ifndef CPN_FEW_DEPTH
#define CPN_FEW_DEPTH 4
#endif
#ifndef CPN_FEW_RT
#define CPN_FEW_RT 2
#endif
constexpr int FEW_NT = CPN_FEW_NT, FEW_RT = CPN_FEW_RT;
__global__ __launch_bounds__(256) void gemm_f16_fewrows_kernel(const __half* __restrict__ A, int lda,
                                                               const half8* __restrict__ Wp, const float* __restrict__ bias,
                                                               float* __restrict__ C, int ldc, int M, int N, int K32, int relu) {
    constexpr int NT = FEW_NT, RT = FEW_RT, D = CPN_FEW_DEPTH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16 * RT;
    const int ntiles = N >> 4;
    const int tile0 = (blockIdx.y * 4 + wave) * NT;
    if (tile0 >= ntiles) return;
    // activation rows in the LOAD layout (lane = 4 * row + 16-byte piece: 4 adjacent lanes read 64 contiguous bytes, 16 tag
    // look-ups per instruction instead of the 64 of the fragment layout lane = row + 16 * piece, which made the kernel wait on
    // the L1 tag pipe: 1.1 us per 64-deep stage), moved to the fragment layout with 4 ds_bpermute per register set
    const int rl = lane >> 2, pl = lane & 3;
    const int to_frag = (4 * r + g) * 4;
    const __half* ap[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int row = m0 + q * 16 + rl < M ? m0 + q * 16 + rl : M - 1;
        ap[q] = A + (size_t)row * lda + pl * 8;
    }
    const half8* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tt = tile0 + t < ntiles ? tile0 + t : ntiles - 1;               // dead tiles of the last wave redo its last live one
        wp[t] = Wp + (size_t)tt * K32 * 64 + lane;
    }
    f32x4 acc[RT][NT];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nfull = K32 >> 1;                  // whole stages of two k32 steps
    half8 xa[D][2][RT], wa[D][2][NT];
    // no branch around a fetch (the compiler's vmcnt bookkeeping stays exact): a stage index past the end re-reads the last
    // stage into a slot nobody uses any more
    auto fetch = [&](int d, int st) {
        st = st < nfull ? st : nfull - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int q = 0; q < RT; ++q) xa[d][h][q] = *reinterpret_cast<const half8*>(ap[q] + (st * 2 + h) * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) wa[d][h][t] = wp[t][(size_t)(st * 2 + h) * 64];
        }
    };
    auto frag = [&](half8 v) {
        const u32x4 src = __builtin_bit_cast(u32x4, v);
        u32x4 dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute(to_frag, (int)src[i]);
        return __builtin_bit_cast(half8, dst);
    };
    auto mma = [&](int d) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            half8 xf[RT];
#pragma unroll
            for (int q = 0; q < RT; ++q) xf[q] = frag(xa[d][h][q]);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < RT; ++q)
                    acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[d][h][t], xf[q], acc[q][t], 0, 0, 0);
        }
    };
    if (nfull > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(d, d);
        int s0 = 0;
        for (; s0 + D <= nfull; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                __builtin_amdgcn_sched_barrier(0);
                mma(d);
                __builtin_amdgcn_sched_barrier(0);
                fetch(d, s0 + d + D);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
            if (s0 + d < nfull) mma(d);
    }
    if (K32 & 1) {                               // odd trailing k32 step
        half8 xl[RT], wl[NT];
#pragma unroll
        for (int q = 0; q < RT; ++q) xl[q] = *reinterpret_cast<const half8*>(ap[q] + (K32 - 1) * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) wl[t] = wp[t][(size_t)(K32 - 1) * 64];
#pragma unroll
        for (int q = 0; q < RT; ++q) xl[q] = frag(xl[q]);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < RT; ++q) acc[q][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xl[q], acc[q][t], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int m = m0 + q * 16 + r;
        if (m >= M) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (tile0 + t >= ntiles) continue;
            const int n = (tile0 + t) * 16 + g * 4;
            f32x4 v = acc[q][t] + *reinterpret_cast<const f32x4*>(bias + n);
            if (relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            *reinterpret_cast<f32x4*>(C + (size_t)m * ldc + n) = v;
        }
    }
}

}  // namespace

extern "C" int cpn_pack_gemm_frags(const uint16_t* W, int ldw, int N, int K, uint16_t* out, void* stream) {
    CPN_REQUIRE(W && out, CPN_E_ARG, "cpn_pack_gemm_frags: null pointer");
    CPN_REQUIRE(N > 0 && (N % 16) == 0 && K > 0 && (K % 32) == 0 && ldw >= K && (ldw % 8) == 0, CPN_E_SHAPE,
                "cpn_pack_gemm_frags: need N %% 16 == 0, K %% 32 == 0, ldw >= K (got N=%d K=%d ldw=%d)", N, K, ldw);
    CPN_REQUIRE(((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, CPN_E_ARG, "cpn_pack_gemm_frags: pointers must be 16-byte aligned");
    const long long total = (long long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(pack_gemm_frags_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)W, ldw, N,
                       K / 32, (half8*)out);
    CPN_LAUNCH_CHECK("cpn_pack_gemm_frags");
    return 0;
}

extern "C" int cpn_gemm_f16_fewrows(const uint16_t* A, int lda, const uint16_t* Wp, const float* bias, float* C, int ldc, int M,
                                    int N, int K, int relu, void* stream) {
    CPN_REQUIRE(A && Wp && bias && C, CPN_E_ARG, "cpn_gemm_f16_fewrows: null pointer");
    CPN_REQUIRE(M > 0 && N > 0 && (N % 16) == 0 && K > 0 && (K % 32) == 0, CPN_E_SHAPE,
                "cpn_gemm_f16_fewrows: need N %% 16 == 0 and K %% 32 == 0 (got N=%d K=%d)", N, K);
    CPN_REQUIRE(lda >= K && (lda % 8) == 0 && ldc >= N && (ldc % 4) == 0, CPN_E_SHAPE, "cpn_gemm_f16_fewrows: bad leading dimension");
    CPN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)Wp % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)bias % 16) == 0,
                CPN_E_ARG, "cpn_gemm_f16_fewrows: pointers must be 16-byte aligned");
    dim3 grid(cpn_cdiv(M, 16 * FEW_RT), cpn_cdiv(N / 16, 4 * FEW_NT));
    hipLaunchKernelGGL(gemm_f16_fewrows_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const __half*)A, lda, (const half8*)Wp, bias,
                       C, ldc, M, N, K / 32, relu);
    CPN_LAUNCH_CHECK("cpn_gemm_f16_fewrows");
    return 0;
}

extern "C" int cpn_gemm_f16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, void* C,
                            int ldc, int M, int N, int K, int relu, int out_f32, void* stream) {
    CPN_REQUIRE(A && W && bias && C, CPN_E_ARG, "cpn_gemm_f16: null pointer");
    CPN_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 32) == 0, CPN_E_SHAPE, "cpn_gemm_f16: K=%d must be a multiple of 32", K);
    // a trailing half stage over-reads up to 32 halves past K: inside the buffer that is the next row's (unused)
    // data, past the end the descriptor's bounds check returns zeros — so only K itself must fit in a row
    CPN_REQUIRE(lda >= K && ldw >= K && (lda % 8) == 0 && (ldw % 8) == 0, CPN_E_SHAPE,
                "cpn_gemm_f16: lda=%d / ldw=%d must be >= K=%d and multiples of 8 halves", lda, ldw, K);
    CPN_REQUIRE((long long)256 * lda * 2 < (1LL << 31) && (long long)N * ldw * 2 < (1LL << 31), CPN_E_SHAPE,
                "cpn_gemm_f16: tile exceeds the 32-bit buffer offset range");
    CPN_REQUIRE(ldc >= N && (ldc % 8) == 0, CPN_E_SHAPE, "cpn_gemm_f16: ldc=%d must be >= N and a multiple of 8", ldc);
    CPN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)C % 16) == 0 &&
                    ((uintptr_t)bias % 16) == 0, CPN_E_ARG, "cpn_gemm_f16: pointers must be 16-byte aligned");
    const hipStream_t s = (hipStream_t)stream;
    const __half* a = (const __half*)A;
    const __half* w = (const __half*)W;
    if (N % 208 == 0) return dispatch<13>(a, lda, w, ldw, bias, C, ldc, M, N, K / 32, relu, out_f32, s);
    if (N % 128 == 0) return dispatch<8>(a, lda, w, ldw, bias, C, ldc, M, N, K / 32, relu, out_f32, s);
    cpn_set_error("cpn_gemm_f16: N=%d is neither a multiple of 208 nor of 128", N);
    return CPN_E_SHAPE;
}

extern "C" int cpn_gemm_f16_combine(const uint16_t* dkh, int lda, const uint16_t* Wt, int ldw, const uint16_t* hid, const float* w1,
                                    const float* dh1, const float* w2, const float* dh2, int B, int V, int R, int S, int ray0,
                                    int nrays, int K, uint16_t* out, void* stream) {
    CPN_REQUIRE(dkh && Wt && hid && out && (!w1 || dh1) && (!w2 || dh2), CPN_E_ARG, "cpn_gemm_f16_combine: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && (S % 16) == 0, CPN_E_SHAPE, "cpn_gemm_f16_combine: need V == 2, S %% 16 == 0");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_SHAPE,
                "cpn_gemm_f16_combine: ray range outside B*R");
    CPN_REQUIRE(K > 0 && (K % 32) == 0 && lda >= K && ldw >= K && (lda % 8) == 0 && (ldw % 8) == 0, CPN_E_SHAPE,
                "cpn_gemm_f16_combine: bad K / leading dimension");
    const long long M = (long long)nrays * V * S;
    CPN_REQUIRE(M < (1LL << 31) && (long long)256 * lda * 2 < (1LL << 31) && (long long)1664 * ldw * 2 < (1LL << 31), CPN_E_SHAPE,
                "cpn_gemm_f16_combine: problem exceeds the 32-bit index range");
    CPN_REQUIRE(((uintptr_t)dkh % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)hid % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                    ((uintptr_t)dh1 % 16) == 0 && ((uintptr_t)dh2 % 16) == 0, CPN_E_ARG,
                "cpn_gemm_f16_combine: pointers must be 16-byte aligned");
    CombineArgs ca{w1, dh1, w2, dh2, V, R, S, ray0, 0};
    return launch_combine<13>((const __half*)dkh, lda, (const __half*)Wt, ldw, (const __half*)hid, 1664, ca, (__half*)out, 1664,
                              (int)M, 1664, K / 32, (hipStream_t)stream);
}

extern "C" int cpn_gemm_f16_masked(const uint16_t* A, int lda, const uint16_t* Wt, int ldw, const uint16_t* mask, int ldm,
                                   uint16_t* out, int ldc, int M, int N, int K, void* stream) {
    CPN_REQUIRE(A && Wt && mask && out, CPN_E_ARG, "cpn_gemm_f16_masked: null pointer");
    CPN_REQUIRE(M > 0 && (M % 16) == 0 && N > 0 && K > 0 && (K % 32) == 0, CPN_E_SHAPE,
                "cpn_gemm_f16_masked: need M %% 16 == 0 and K %% 32 == 0 (got M=%d K=%d)", M, K);
    CPN_REQUIRE(lda >= K && ldw >= K && (lda % 8) == 0 && (ldw % 8) == 0 && ldm >= N && (ldm % 8) == 0 && ldc >= N && (ldc % 8) == 0,
                CPN_E_SHAPE, "cpn_gemm_f16_masked: bad leading dimension");
    CPN_REQUIRE((long long)256 * lda * 2 < (1LL << 31) && (long long)N * ldw * 2 < (1LL << 31), CPN_E_SHAPE,
                "cpn_gemm_f16_masked: tile exceeds the 32-bit buffer offset range");
    CPN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && ((uintptr_t)mask % 16) == 0 && ((uintptr_t)out % 16) == 0,
                CPN_E_ARG, "cpn_gemm_f16_masked: pointers must be 16-byte aligned");
    CombineArgs ca{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
    if (N % 208 == 0)
        return launch_combine<13>((const __half*)A, lda, (const __half*)Wt, ldw, (const __half*)mask, ldm, ca, (__half*)out, ldc, M, N,
                                  K / 32, (hipStream_t)stream);
    if (N % 128 == 0)
        return launch_combine<8>((const __half*)A, lda, (const __half*)Wt, ldw, (const __half*)mask, ldm, ca, (__half*)out, ldc, M, N,
                                 K / 32, (hipStream_t)stream);
    cpn_set_error("cpn_gemm_f16_masked: N=%d is neither a multiple of 208 nor of 128", N);
    return CPN_E_SHAPE;
}

extern "C" int cpn_gemm_f16_rowdot(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                                   const uint16_t* Q, int ldq, float* logits, int M, int N, int K, void* stream) {
    CPN_REQUIRE(A && W && bias && Q && logits, CPN_E_ARG, "cpn_gemm_f16_rowdot: null pointer");
    CPN_REQUIRE(M > 0 && N == 128 && K > 0 && (K % 32) == 0, CPN_E_SHAPE, "cpn_gemm_f16_rowdot: need N == 128, K %% 32 == 0");
    CPN_REQUIRE(lda >= K && ldw >= K && (lda % 8) == 0 && (ldw % 8) == 0 && (ldq == 0 || (ldq >= N && (ldq % 4) == 0)), CPN_E_SHAPE,
                "cpn_gemm_f16_rowdot: bad leading dimension");
    CPN_REQUIRE((long long)256 * lda * 2 < (1LL << 31) && (long long)N * ldw * 2 < (1LL << 31), CPN_E_SHAPE,
                "cpn_gemm_f16_rowdot: tile exceeds the 32-bit buffer offset range");
    CPN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)Q % 8) == 0 &&
                    ((uintptr_t)bias % 16) == 0, CPN_E_ARG, "cpn_gemm_f16_rowdot: pointers must be aligned");
    return launch_rowdot<8, false, 1>((const __half*)A, lda, (const __half*)W, ldw, bias, (const __half*)Q, ldq, logits, M,
                                      K / 32, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int cpn_gemm_f16_chain_rowdot(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                                         const uint16_t* W2, int ldw2, const float* bias2, const uint16_t* Q, int ldq,
                                         float* logits, int M, int K, void* stream) {
    CPN_REQUIRE(A && W && bias && W2 && bias2 && Q && logits, CPN_E_ARG, "cpn_gemm_f16_chain_rowdot: null pointer");
    CPN_REQUIRE(M > 0 && K > 0 && (K % 32) == 0, CPN_E_SHAPE, "cpn_gemm_f16_chain_rowdot: K %% 32 != 0");
    CPN_REQUIRE(lda >= K && ldw >= K && (lda % 8) == 0 && (ldw % 8) == 0 && (ldq == 0 || (ldq >= 128 && (ldq % 4) == 0)) && ldw2 >= 128 &&
                    (ldw2 % 4) == 0, CPN_E_SHAPE, "cpn_gemm_f16_chain_rowdot: bad leading dimension");
    CPN_REQUIRE((long long)256 * lda * 2 < (1LL << 31) && (long long)128 * ldw * 2 < (1LL << 31), CPN_E_SHAPE,
                "cpn_gemm_f16_chain_rowdot: tile exceeds the 32-bit buffer offset range");
    CPN_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)Q % 8) == 0 && ((uintptr_t)W2 % 8) == 0 &&
                    ((uintptr_t)bias % 16) == 0 && ((uintptr_t)bias2 % 16) == 0, CPN_E_ARG,
                "cpn_gemm_f16_chain_rowdot: pointers must be aligned");
    return launch_rowdot<8, true, 2>((const __half*)A, lda, (const __half*)W, ldw, bias, (const __half*)Q, ldq, logits, M,
                                     K / 32, (const __half*)W2, ldw2, bias2, (hipStream_t)stream);
}
