// K3 — C = act(A . W^T + bias): the per-sample 1x1 convolutions of the render path as one MFMA GEMM.
//
// Replaces nn.Conv2d(1x1) x 13 calls (/root/reference models/CoPoNeRF.py:387-397, 404, 408, 446, 473):
// 835->832 (ReLU) ->416, 832->416, 832->128 (ReLU) ->128, 128->128 — 99.7 % of the path's FLOPs.
//
// gfx950 design
//   * v_mfma_f32_16x16x32_f16, fp16 operands, fp32 accumulate.  The N sizes of this network are
//     13 x 64 / 13 x 32, so the workgroup tile is 256 (M) x 16*NT (N) with NT = 13 (N = 832, 416) or 8 (N = 128);
//     32x32 tiles would need N % 32-per-wave splits that 13 does not allow without 7.7 % padding.
//   * 8 waves = 512 threads, wave w owns rows [32w, 32w+32) x all NT column tiles: 2 x NT accumulators of 4 regs.
//   * operands are swapped (A-operand = weights, B-operand = activations) so that a lane ends up holding
//     4 CONSECUTIVE output columns of one row -> 8-byte (fp16) / 16-byte (fp32) row-contiguous stores.
//   * both tiles are K-contiguous (activations (M,K), weights (N,K)): staged with global_load_lds_dwordx4
//     (no VGPR round trip) into a double-buffered [rows][64] fp16 image (128-B rows).  The LDS image is
//     lane-linear, so the bank-conflict XOR swizzle is applied to the per-lane SOURCE address
//     (16-B chunk c of row r lives at physical chunk c ^ ((r >> 1) & 7)); fragment reads apply the same
//     involution -> ds_read_b128 of 16 rows x same k-chunk touches 16 distinct 16-B slots (conflict-free).
//   * one barrier per 64-deep K step: the loads of step t+1 are issued right after the barrier that
//     publishes step t and fly under its 2 x 2 x NT MFMAs.
// Roofline: compute-bound (arithmetic intensity of the 256 x 208 tile = 115 FLOP/B of L2 traffic);
// algorithmic FLOPs per launch = 2*M*N*K.
#include "common.h"
#include <stdlib.h>

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
    static constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
};

// blockIdx.x -> (m tile, n tile).  Blocks are dispatched round-robin over the 8 XCDs (private L2 each), so the
// linear id is first remapped to give every XCD a contiguous range of logical tiles (bijective for any grid size),
// then the N tile index runs fastest: the n_tiles workgroups that share one 256-row activation tile execute
// back to back on the same XCD and hit its L2 instead of re-streaming the tile from HBM.
__device__ __forceinline__ void tile_of_block(int n_tiles, int& m_tile, int& n_tile) {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    n_tile = logical % n_tiles;
    m_tile = logical / n_tiles;
}

template <int NT, bool OUT_F32, bool RELU>
__global__ __launch_bounds__(512) void gemm_f16_kernel(const __half* __restrict__ A, int lda,
                                                       const __half* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, void* __restrict__ Cv,
                                                       int ldc, int M, int K32, int n_tiles) {
    using C_ = Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_tile, n_tile;
    tile_of_block(n_tiles, m_tile, n_tile);
    const int m0 = m_tile * BM;
    const int n0 = n_tile * C_::BN;
    const int nk = (K32 + 1) >> 1;                     // 64-deep steps; the last may hold a single k32

    // ---- per-lane DMA sources: unit u covers image rows [8u, 8u+8); lane -> (row, physical chunk).
    // Wave w moves units w, w+8, w+16, ...; the per-lane source pointer of each is fixed up to the K offset.
    constexpr int MAXU = (C_::UNITS + 7) / 8;
    const __half* dma_src[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        const int u = wave + 8 * i;
        const int r = u * 8 + (lane >> 3);             // row in the concatenated [A tile ; W tile] image
        const int lchunk = (lane & 7) ^ ((r >> 1) & 7);
        if (u < C_::UNITS_A) {
            int gr = m0 + r;
            gr = gr < M ? gr : M - 1;                  // rows past M: clamp (results discarded)
            dma_src[i] = A + (size_t)gr * lda + lchunk * 8;
        } else {
            int wr = n0 + r - BM;
            wr = u < C_::UNITS ? wr : n0;              // inactive slot of the last round
            dma_src[i] = W + (size_t)wr * ldw + lchunk * 8;
        }
    }
    auto stage = [&](int kt, int buf) {
        char* sbase = smem + buf * C_::STAGE_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (wave + 8 * i < C_::UNITS)
                __builtin_amdgcn_global_load_lds((glb_void*)(dma_src[i] + kt * BK), (lds_void*)(sbase + i * 8192), 16,
                                                 0, 0);
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: lane reads row (lane&15) of a 16-row fragment, 16-B chunk ks*4 + (lane>>4)
    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int xrow = wave * 32 + frow;                 // + 16*mt ; (row>>1)&7 only depends on frow
    const int swz = (frow >> 1) & 7;
    const int xoff = xrow * ROW_BYTES;
    const int woff = (BM + frow) * ROW_BYTES;
    const int coff0 = ((fk ^ swz) << 4), coff1 = (((4 + fk) ^ swz) << 4);

    auto load_frags = [&](const char* sbase, int coff, half8 (&xa)[2], half8 (&wb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            xa[mt] = *reinterpret_cast<const half8*>(sbase + xoff + mt * 16 * ROW_BYTES + coff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wb[nt] = *reinterpret_cast<const half8*>(sbase + woff + nt * 16 * ROW_BYTES + coff);
    };
    auto mma = [&](const half8 (&xa)[2], const half8 (&wb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt], xa[mt], acc[mt][nt], 0, 0, 0);
    };

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                               // drains this wave's DMA (vmcnt(0)) + publishes step kt
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* sbase = smem + (kt & 1) * C_::STAGE_BYTES;
        {
            half8 xa[2], wb[NT];
            load_frags(sbase, coff0, xa, wb);
            mma(xa, wb);
        }
        if (kt * 2 + 2 <= K32) {
            half8 xa[2], wb[NT];
            load_frags(sbase, coff1, xa, wb);
            mma(xa, wb);
        }
    }

    // ---- epilogue: D[n = nt*16 + (lane>>4)*4 + i][m = mt*16 + (lane&15)]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 16 + (lane >> 4) * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + wave * 32 + mt * 16 + (lane & 15);
            if (m >= M) continue;
            f32x4 v = acc[mt][nt] + bv;
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            if (OUT_F32) {
                *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
            } else {
                half4 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                *reinterpret_cast<half4*>((__half*)Cv + (size_t)m * ldc + n) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 4-wave variant: one wave per SIMD, wave tile 64 x (16*NT), up to 512 registers per lane.
// Each weight fragment read from LDS now feeds 4 MFMAs instead of 2 (LDS read traffic per MFMA drops from
// 0.58 to 0.33 KiB), and both k32 fragment sets of a 64-deep stage are in flight before the first MFMA.
// ---------------------------------------------------------------------------------------------
template <int NT, bool OUT_F32, bool RELU>
__global__ __launch_bounds__(256, 1) void gemm_f16_w4_kernel(const __half* __restrict__ A, int lda,
                                                             const __half* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             void* __restrict__ Cv, int ldc, int M, int K32, int n_tiles) {
    using C_ = Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_tile, n_tile;
    tile_of_block(n_tiles, m_tile, n_tile);
    const int m0 = m_tile * BM;
    const int n0 = n_tile * C_::BN;
    const int nk = (K32 + 1) >> 1;

    // DMA (buffer_load ... lds): wave w moves units w, w+4, ...; unit u = image rows [8u, 8u+8).  Rows advance by
    // 32 per round, so the source swizzle ((row>>1)&7) and hence the per-lane byte offset are round-invariant:
    // ONE voffset VGPR per operand, everything else in the scalar offset.  The A descriptor ends at row
    // min(BM, M-m0): rows past M read as zero (hardware bounds check) instead of being clamped.
    constexpr int MAXU = (C_::UNITS + 3) / 4;
    constexpr int UA4 = C_::UNITS_A / 4;
    const int r0 = wave * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((r0 >> 1) & 7);
    const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A + (size_t)m0 * lda), 0, rows_valid * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (size_t)n0 * ldw), 0, C_::BN * ldw * 2, 0x00020000);
    const int voff_a = (r0 * lda + lchunk * 8) * 2;
    const int voff_w = (r0 * ldw + lchunk * 8) * 2;
    auto stage = [&](int kt, int buf) {
        char* sbase = smem + buf * C_::STAGE_BYTES + wave * 1024;
        const int kbytes = kt * BK * 2;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (wave + 4 * i >= C_::UNITS) break;
            if (i < UA4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(sbase + i * 4096), 16, voff_a,
                                                         kbytes + i * 64 * lda, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(sbase + i * 4096), 16, voff_w,
                                                         kbytes + (i - UA4) * 64 * ldw, 0, 0);
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int swz = (frow >> 1) & 7;
    const int xoff = (wave * 64 + frow) * ROW_BYTES;
    const int woff = (BM + frow) * ROW_BYTES;
    const int coff0 = ((fk ^ swz) << 4), coff1 = (((4 + fk) ^ swz) << 4);

    auto load_frags = [&](const char* sbase, int coff, half8 (&xa)[4], half8 (&wb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            xa[mt] = *reinterpret_cast<const half8*>(sbase + xoff + mt * 16 * ROW_BYTES + coff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wb[nt] = *reinterpret_cast<const half8*>(sbase + woff + nt * 16 * ROW_BYTES + coff);
    };
    auto mma = [&](const half8 (&xa)[4], const half8 (&wb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt], xa[mt], acc[mt][nt], 0, 0, 0);
    };

    // Software pipeline (one wave per SIMD has no partner wave to hide LDS latency behind):
    //   phase A  MFMAs on fragment set 0 (stage kt, first k32)  ||  ds_reads of set 1 (stage kt, second k32)
    //   barrier  -> stage kt+1 has landed for every wave, every wave is done reading stage kt
    //   phase B  DMA of stage kt+2 into the buffer just freed;
    //            MFMAs on set 1  ||  ds_reads of set 0 for stage kt+1
    // so the barrier sits between two MFMA blocks whose operands are already in registers.
    // sched_group_barrier pins the interleave to 1 ds_read per 3 MFMAs (17 reads under 52 MFMAs).
#define CPN_INTERLEAVE_READS_MFMA()                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4 + NT; ++q_) {                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* 1 DS read */      \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); /* 3 MFMA    */      \
    }                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 3 * (4 + NT), 0);

    const int nfull = K32 >> 1;
    half8 xa0[4], wb0[NT], xa1[4], wb1[NT];
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, coff0, xa0, wb0);
    for (int kt = 0; kt < nfull; ++kt) {
        const char* scur = smem + (kt & 1) * C_::STAGE_BYTES;
        const char* snxt = smem + ((kt + 1) & 1) * C_::STAGE_BYTES;
        load_frags(scur, coff1, xa1, wb1);
        mma(xa0, wb0);
        CPN_INTERLEAVE_READS_MFMA();
        // the compiler does not count buffer_load...lds against the barrier: drain this wave's DMA explicitly
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 2 < nk) stage(kt + 2, kt & 1);
        load_frags(snxt, coff0, xa0, wb0);             // harmless garbage after the last stage
        mma(xa1, wb1);
        CPN_INTERLEAVE_READS_MFMA();
    }
    if (K32 & 1) mma(xa0, wb0);                        // odd trailing k32 step (already in set 0)
#undef CPN_INTERLEAVE_READS_MFMA

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 16 + (lane >> 4) * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wave * 64 + mt * 16 + (lane & 15);
            if (m >= M) continue;
            f32x4 v = acc[mt][nt] + bv;
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            if (OUT_F32) {
                *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
            } else {
                half4 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                *reinterpret_cast<half4*>((__half*)Cv + (size_t)m * ldc + n) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 4: the 8-wave kernel (wave tile 32 x 16*NT, all registers in the VGPR file: 2*NT*4 accumulators +
// two fragment sets = 224 <= 256, so two waves per SIMD and no VGPR<->AGPR shuffling) with the two-phase software
// pipeline: every MFMA block runs on fragments read one phase earlier, 1 ds_read pinned per 2 MFMAs.
// ---------------------------------------------------------------------------------------------
template <int NT, bool OUT_F32, bool RELU>
__global__ __launch_bounds__(512) void gemm_f16_p8_kernel(const __half* __restrict__ A, int lda,
                                                             const __half* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             void* __restrict__ Cv, int ldc, int M, int K32, int n_tiles, int ablate) {
    using C_ = Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_tile, n_tile;
    tile_of_block(n_tiles, m_tile, n_tile);
    const int m0 = m_tile * BM;
    const int n0 = n_tile * C_::BN;
    const int nk = (K32 + 1) >> 1;

    // DMA (buffer_load ... lds): wave w moves units w, w+4, ...; unit u = image rows [8u, 8u+8).  Rows advance by
    // 32 per round, so the source swizzle ((row>>1)&7) and hence the per-lane byte offset are round-invariant:
    // ONE voffset VGPR per operand, everything else in the scalar offset.  The A descriptor ends at row
    // min(BM, M-m0): rows past M read as zero (hardware bounds check) instead of being clamped.
    constexpr int MAXU = (C_::UNITS + 7) / 8;
    constexpr int UA4 = C_::UNITS_A / 8;
    const int r0 = wave * 8 + (lane >> 3);
    const int lchunk = (lane & 7) ^ ((r0 >> 1) & 7);
    const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A + (size_t)m0 * lda), 0, rows_valid * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (size_t)n0 * ldw), 0, C_::BN * ldw * 2, 0x00020000);
    const int voff_a = (r0 * lda + lchunk * 8) * 2;
    const int voff_w = (r0 * ldw + lchunk * 8) * 2;
    auto stage = [&](int kt, int buf) {
        char* sbase = smem + buf * C_::STAGE_BYTES + wave * 1024;
        const int kbytes = kt * BK * 2;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (wave + 8 * i >= C_::UNITS) break;
            if (i < UA4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(sbase + i * 8192), 16, voff_a,
                                                         kbytes + i * 128 * lda, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(sbase + i * 8192), 16, voff_w,
                                                         kbytes + (i - UA4) * 128 * ldw, 0, 0);
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fk = lane >> 4;
    const int swz = (frow >> 1) & 7;
    const int xoff = (wave * 32 + frow) * ROW_BYTES;
    const int woff = (BM + frow) * ROW_BYTES;
    const int coff0 = ((fk ^ swz) << 4), coff1 = (((4 + fk) ^ swz) << 4);

    auto load_frags = [&](const char* sbase, int coff, half8 (&xa)[2], half8 (&wb)[NT]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            xa[mt] = *reinterpret_cast<const half8*>(sbase + xoff + mt * 16 * ROW_BYTES + coff);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            wb[nt] = *reinterpret_cast<const half8*>(sbase + woff + nt * 16 * ROW_BYTES + coff);
    };
    auto mma = [&](const half8 (&xa)[2], const half8 (&wb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt], xa[mt], acc[mt][nt], 0, 0, 0);
    };

    // Software pipeline (one wave per SIMD has no partner wave to hide LDS latency behind):
    //   phase A  MFMAs on fragment set 0 (stage kt, first k32)  ||  ds_reads of set 1 (stage kt, second k32)
    //   barrier  -> stage kt+1 has landed for every wave, every wave is done reading stage kt
    //   phase B  DMA of stage kt+2 into the buffer just freed;
    //            MFMAs on set 1  ||  ds_reads of set 0 for stage kt+1
    // so the barrier sits between two MFMA blocks whose operands are already in registers.
    // sched_group_barrier pins the interleave to 1 ds_read per 3 MFMAs (17 reads under 52 MFMAs).
#define CPN_INTERLEAVE_READS_MFMA()                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < NT; ++q_) {                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* 1 DS read */      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); /* 2 MFMA    */      \
    }                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);

    const int nfull = K32 >> 1;
    half8 xa0[2], wb0[NT], xa1[2], wb1[NT];
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    load_frags(smem, coff0, xa0, wb0);
    load_frags(smem, coff1, xa1, wb1);
    for (int kt = 0; kt < nfull; ++kt) {
        const char* scur = smem + (kt & 1) * C_::STAGE_BYTES;
        const char* snxt = smem + ((kt + 1) & 1) * C_::STAGE_BYTES;
        if (!(ablate & 2)) load_frags(scur, coff1, xa1, wb1);
        mma(xa0, wb0);
        CPN_INTERLEAVE_READS_MFMA();
        // the compiler does not count buffer_load...lds against the barrier: drain this wave's DMA explicitly
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(ablate & 4)) __syncthreads();
        if (kt + 2 < nk && !(ablate & 1)) stage(kt + 2, kt & 1);
        if (!(ablate & 2)) load_frags(snxt, coff0, xa0, wb0);             // harmless garbage after the last stage
        mma(xa1, wb1);
        CPN_INTERLEAVE_READS_MFMA();
    }
    if (K32 & 1) mma(xa0, wb0);                        // odd trailing k32 step (already in set 0)
#undef CPN_INTERLEAVE_READS_MFMA

    if constexpr (!OUT_F32) {
        // fp16 epilogue through LDS: the accumulator layout gives a lane 4 consecutive columns (8 B) of one row, i.e.
        // 32-B row segments per store; staging the wave's 32 x BN tile in the (now idle) ring and reading it back
        // row-contiguously turns 2*NT 8-byte stores into NT 16-byte stores that cover whole 416-B rows.
        constexpr int RS = C_::BN * 2 + 16;                       // padded row stride: conflict-free ds_write_b64
        constexpr int CPR = C_::BN / 8;                           // 16-byte chunks per row
        __syncthreads();                                          // every wave is done reading the ring
        char* cw = smem + wave * (32 * RS);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0 + nt * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 v = acc[mt][nt] + bv;
                if (RELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                }
                half4 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                *reinterpret_cast<half4*>(cw + (mt * 16 + (lane & 15)) * RS + (nt * 16 + (lane >> 4) * 4) * 2) = h;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __half* cbase = (__half*)Cv + (size_t)(m0 + wave * 32) * ldc + n0;
#pragma unroll
        for (int i = 0; i < (32 * CPR + 63) / 64; ++i) {
            const int q = lane + 64 * i;
            const int r = q / CPR, c = q - r * CPR;
            if (q < 32 * CPR && m0 + wave * 32 + r < M) {
                const half8 val = *reinterpret_cast<const half8*>(cw + r * RS + c * 16);
                *reinterpret_cast<half8*>(cbase + (size_t)r * ldc + c * 8) = val;
            }
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
                if (RELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
                }
                *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ring variant: 32-deep K steps in a RING-slot LDS ring filled by buffer_load...lds with COUNTED vmcnt waits and a
// raw s_barrier (RING-1 stages in flight).  WAVES = 8 -> 256-row tile, one workgroup per CU; WAVES = 4 -> 128-row
// tile, 64.5 KiB of LDS (RING = 3), so TWO independent workgroups share a CU and one's epilogue (the C tile is
// 35 % of the kernel's time when nothing overlaps it) and prologue hide under the other's main loop.
// Row = 64 B (4 chunks of 16 B); chunk c of row r sits at physical chunk c ^ (3 * ((r >> 3) & 1)), which makes
// every ds_read_b128 lane group touch 16 distinct 16-B slots.
// ---------------------------------------------------------------------------------------------
template <int NT, int WAVES, int RING, bool OUT_F32, bool RELU>
__global__ __launch_bounds__(WAVES * 64) void gemm_f16_ring_kernel(const __half* __restrict__ A, int lda,
                                                                  const __half* __restrict__ W, int ldw,
                                                                  const float* __restrict__ bias,
                                                                  void* __restrict__ Cv, int ldc, int M, int K32,
                                                                  int n_tiles) {
    constexpr int BMR = WAVES * 32;                   // rows per workgroup tile
    constexpr int BN = NT * 16;
    constexpr int UA = BMR / 16;                      // 1-KiB DMA units (16 rows x 64 B) of the activation tile
    constexpr int UNITS = UA + NT;
    constexpr int MAXU = 2 + (NT + WAVES - 1) / WAVES;
    constexpr int STAGE = UNITS * 1024;
    constexpr int RB = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_tile, n_tile;
    tile_of_block(n_tiles, m_tile, n_tile);
    const int m0 = m_tile * BMR;
    const int n0 = n_tile * BN;

    const int urow = lane >> 2;
    const int lchunk = (lane & 3) ^ (3 * ((urow >> 3) & 1));
    const int rows_valid = (M - m0) < BMR ? (M - m0) : BMR;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A + (size_t)m0 * lda), 0, rows_valid * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (size_t)n0 * ldw), 0, BN * ldw * 2, 0x00020000);
    const int voff_a = (urow * lda + lchunk * 8) * 2;
    const int voff_w = (urow * ldw + lchunk * 8) * 2;
    const bool extra = wave + WAVES * (MAXU - 3) < NT;                 // this wave issues MAXU (else MAXU-1) DMAs
    auto stage = [&](int t) {
        char* sbase = smem + (t % RING) * STAGE + wave * 1024;
        const int kb = t * 64;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (i == MAXU - 1 && !extra) break;
            if (i < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(sbase + i * WAVES * 1024), 16, voff_a,
                                                         kb + (wave + WAVES * i) * 32 * lda, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(sbase + i * WAVES * 1024), 16, voff_w,
                                                         kb + (wave + WAVES * (i - 2)) * 32 * ldw, 0, 0);
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fpc = ((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4;
    const int xoff = (wave * 32 + frow) * RB + fpc;
    const int woff = (BMR + frow) * RB + fpc;

#pragma unroll
    for (int t = 0; t < RING - 1; ++t)
        if (t < K32) stage(t);
    for (int t = 0; t < K32; ++t) {
        // wait until this wave's DMAs of stage t have landed: the DMAs of the next RING-2 stages stay in flight
        const int newer = (K32 - 1 - t) < (RING - 2) ? (K32 - 1 - t) : (RING - 2);
        if (newer == 2) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (MAXU - 1)) : "memory");
        } else if (newer == 1) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXU - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");        // stage t visible to all waves; slot (t-1)%RING is free
        if (t + RING - 1 < K32) stage(t + RING - 1);
        const char* sbase = smem + (t % RING) * STAGE;
        half8 xa[2], wb[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xa[mt] = *reinterpret_cast<const half8*>(sbase + xoff + mt * 16 * RB);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wb[nt] = *reinterpret_cast<const half8*>(sbase + woff + nt * 16 * RB);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt], xa[mt], acc[mt][nt], 0, 0, 0);
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 16 + (lane >> 4) * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = m0 + wave * 32 + mt * 16 + (lane & 15);
            if (m >= M) continue;
            f32x4 v = acc[mt][nt] + bv;
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            if (OUT_F32) {
                *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
            } else {
                half4 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                *reinterpret_cast<half4*>((__half*)Cv + (size_t)m * ldc + n) = h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Variant 3: one wave per SIMD (4 waves, wave tile 64 x 16*NT, 4*NT accumulators) AND a 4-slot ring of 32-deep
// stages with counted vmcnt.  Every MFMA block runs on fragments that were read from LDS one phase earlier while
// the previous block executed (sched_group_barrier pins 1 ds_read per 3 MFMAs), so neither LDS latency nor the
// HBM latency of the next three stages is exposed; the barrier sits between two blocks whose operands are in
// registers already.
// ---------------------------------------------------------------------------------------------
template <int NT, bool OUT_F32, bool RELU>
__global__ __launch_bounds__(256, 1) void gemm_f16_p4_kernel(const __half* __restrict__ A, int lda,
                                                             const __half* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias,
                                                             void* __restrict__ Cv, int ldc, int M, int K32,
                                                             int n_tiles) {
    constexpr int BN = NT * 16;
    constexpr int UA = BM / 16;                       // 16 activation units (16 rows x 64 B) per stage
    constexpr int UNITS = UA + NT;
    constexpr int MAXU = (UNITS + 3) / 4;
    constexpr int STAGE = 32 * 1024;
    constexpr int RB = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_tile, n_tile;
    tile_of_block(n_tiles, m_tile, n_tile);
    const int m0 = m_tile * BM;
    const int n0 = n_tile * BN;

    const int urow = lane >> 2;
    const int lchunk = (lane & 3) ^ (3 * ((urow >> 3) & 1));
    const int rows_valid = (M - m0) < BM ? (M - m0) : BM;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(A + (size_t)m0 * lda), 0, rows_valid * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(W + (size_t)n0 * ldw), 0, BN * ldw * 2, 0x00020000);
    const int voff_a = (urow * lda + lchunk * 8) * 2;
    const int voff_w = (urow * ldw + lchunk * 8) * 2;
    const bool extra = wave + 4 * (MAXU - 1) < UNITS;                  // this wave issues MAXU (else MAXU-1) DMAs
    auto stage = [&](int t) {
        char* sbase = smem + (t & 3) * STAGE + wave * 1024;
        const int kb = t * 64;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            if (i == MAXU - 1 && !extra) break;
            if (i < UA / 4)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(sbase + i * 4096), 16, voff_a,
                                                         kb + (wave + 4 * i) * 32 * lda, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void*)(sbase + i * 4096), 16, voff_w,
                                                         kb + (wave + 4 * i - UA) * 32 * ldw, 0, 0);
        }
    };
    // wait until this wave's DMAs of stage s have landed, leaving the `newer` later stages in flight
    auto wait_stage = [&](int newer) {
        if (newer >= 2) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (MAXU - 1)) : "memory");
        } else if (newer == 1) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXU - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15;
    const int fpc = ((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4;
    const int xoff = (wave * 64 + frow) * RB + fpc;
    const int woff = (BM + frow) * RB + fpc;
    auto load_frags = [&](int t, half8 (&xa)[4], half8 (&wb)[NT]) {
        const char* sbase = smem + (t & 3) * STAGE;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xa[mt] = *reinterpret_cast<const half8*>(sbase + xoff + mt * 16 * RB);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wb[nt] = *reinterpret_cast<const half8*>(sbase + woff + nt * 16 * RB);
    };
    auto mma = [&](const half8 (&xa)[4], const half8 (&wb)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[nt], xa[mt], acc[mt][nt], 0, 0, 0);
    };
#define CPN_INTERLEAVE_READS_MFMA()                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4 + NT; ++q_) {                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                      \
    }                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 3 * (4 + NT), 0);

    half8 xa0[4], wb0[NT], xa1[4], wb1[NT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < K32) stage(t);
    {
        const int last = K32 - 1 < 3 ? K32 - 1 : 3;
        if (last >= 3) {
            if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * MAXU) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (MAXU - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_barrier" ::: "memory");
    load_frags(0, xa0, wb0);
    int t = 0;
    for (; t + 1 < K32; t += 2) {
        // ---- phase t: compute stage t (set 0), fetch fragments of stage t+1 (set 1)
        {
            const int hi = (K32 - 1 < t + 3 ? K32 - 1 : t + 3);
            wait_stage(hi - (t + 1));
            asm volatile("s_barrier" ::: "memory");
            if (t + 4 < K32) stage(t + 4);
            load_frags(t + 1, xa1, wb1);
            mma(xa0, wb0);
            CPN_INTERLEAVE_READS_MFMA();
        }
        // ---- phase t+1: compute stage t+1 (set 1), fetch fragments of stage t+2 (set 0)
        {
            const int hi = (K32 - 1 < t + 4 ? K32 - 1 : t + 4);
            wait_stage(hi >= t + 2 ? hi - (t + 2) : 0);
            asm volatile("s_barrier" ::: "memory");
            if (t + 5 < K32) stage(t + 5);
            load_frags(t + 2, xa0, wb0);               // garbage (unused) past the last stage
            mma(xa1, wb1);
            CPN_INTERLEAVE_READS_MFMA();
        }
    }
    if (t < K32) mma(xa0, wb0);                        // odd trailing stage, fragments already in set 0
#undef CPN_INTERLEAVE_READS_MFMA

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 16 + (lane >> 4) * 4;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m0 + wave * 64 + mt * 16 + (lane & 15);
            if (m >= M) continue;
            f32x4 v = acc[mt][nt] + bv;
            if (RELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            if (OUT_F32) {
                *reinterpret_cast<f32x4*>((float*)Cv + (size_t)m * ldc + n) = v;
            } else {
                half4 h;
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = (_Float16)v[i];
                *reinterpret_cast<half4*>((__half*)Cv + (size_t)m * ldc + n) = h;
            }
        }
    }
}

template <int NT, bool OUT_F32, bool RELU>
int launch(const __half* A, int lda, const __half* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
           int K32, hipStream_t stream) {
    using C_ = Cfg<NT>;
    const size_t lds = 2 * C_::STAGE_BYTES;
    static const int variant = getenv("CPN_GEMM_VARIANT") ? atoi(getenv("CPN_GEMM_VARIANT")) : 4;
    auto kern8 = gemm_f16_kernel<NT, OUT_F32, RELU>;
    auto kern4 = gemm_f16_w4_kernel<NT, OUT_F32, RELU>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)kern4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            cpn_set_error("cpn_gemm_f16: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int n_tiles = N / C_::BN;
    dim3 grid(cpn_cdiv(M, BM) * n_tiles);
    if (variant == 4) {
        auto kern_p8 = gemm_f16_p8_kernel<NT, OUT_F32, RELU>;
        static bool p8_set = false;
        if (!p8_set) {
            hipError_t e = hipFuncSetAttribute((const void*)kern_p8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { cpn_set_error("cpn_gemm_f16: LDS attr: %s", hipGetErrorString(e)); return (int)e; }
            p8_set = true;
        }
        static const int ablate = getenv("CPN_ABLATE") ? atoi(getenv("CPN_ABLATE")) : 0;
        hipLaunchKernelGGL(kern_p8, grid, dim3(512), lds, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles, ablate);
    } else if (variant == 3) {
        auto kern_p4 = gemm_f16_p4_kernel<NT, OUT_F32, RELU>;
        static bool p4_set = false;
        if (!p4_set) {
            hipError_t e = hipFuncSetAttribute((const void*)kern_p4, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32 * 1024);
            if (e != hipSuccess) { cpn_set_error("cpn_gemm_f16: LDS attr: %s", hipGetErrorString(e)); return (int)e; }
            p4_set = true;
        }
        hipLaunchKernelGGL(kern_p4, grid, dim3(256), 4 * 32 * 1024, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles);
    } else if (variant == 2 || variant == 5) {
        // 2: 8 waves x 4-slot ring (256-row tiles); 5: 4 waves x 3-slot ring (128-row tiles, two workgroups per CU)
        if (variant == 2) {
            auto kr = gemm_f16_ring_kernel<NT, 8, 4, OUT_F32, RELU>;
            constexpr int bytes = 4 * (16 + NT) * 1024;
            static bool set = false;
            if (!set) {
                hipError_t e = hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) { cpn_set_error("cpn_gemm_f16: LDS attr: %s", hipGetErrorString(e)); return (int)e; }
                set = true;
            }
            hipLaunchKernelGGL(kr, grid, dim3(512), bytes, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles);
        } else {
            auto kr = gemm_f16_ring_kernel<NT, 4, 3, OUT_F32, RELU>;
            constexpr int bytes = 3 * (8 + NT) * 1024;
            static bool set = false;
            if (!set) {
                hipError_t e = hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess) { cpn_set_error("cpn_gemm_f16: LDS attr: %s", hipGetErrorString(e)); return (int)e; }
                set = true;
            }
            dim3 grid4(cpn_cdiv(M, 128) * n_tiles);
            hipLaunchKernelGGL(kr, grid4, dim3(256), bytes, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles);
        }
    } else if (variant == 0)
        hipLaunchKernelGGL(kern8, grid, dim3(512), lds, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles);
    else
        hipLaunchKernelGGL(kern4, grid, dim3(256), lds, stream, A, lda, W, ldw, bias, C, ldc, M, K32, n_tiles);
    CPN_LAUNCH_CHECK("cpn_gemm_f16");
    return 0;
}

template <int NT>
int dispatch(const __half* A, int lda, const __half* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
             int K32, int relu, int out_f32, hipStream_t s) {
    if (out_f32) {
        return relu ? launch<NT, true, true>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s)
                    : launch<NT, true, false>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s);
    }
    return relu ? launch<NT, false, true>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s)
                : launch<NT, false, false>(A, lda, W, ldw, bias, C, ldc, M, N, K32, s);
}

}  // namespace

extern "C" int cpn_gemm_f16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, void* C,
                            int ldc, int M, int N, int K, int relu, int out_f32, void* stream) {
    CPN_REQUIRE(A && W && bias && C, CPN_E_ARG, "cpn_gemm_f16: null pointer");
    CPN_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 32) == 0, CPN_E_SHAPE, "cpn_gemm_f16: K=%d must be a multiple of 32", K);
    // every 64-deep stage is fetched whole, so rows must hold ceil(K/64)*64 readable halves
    const int kspan = ((K + 63) / 64) * 64;
    CPN_REQUIRE(lda >= kspan && ldw >= kspan && (lda % 8) == 0 && (ldw % 8) == 0, CPN_E_SHAPE,
                "cpn_gemm_f16: lda=%d / ldw=%d must be >= %d and multiples of 8 halves", lda, ldw, kspan);
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
