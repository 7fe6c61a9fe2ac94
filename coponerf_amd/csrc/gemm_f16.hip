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

template <int NT, bool OUT_F32, bool RELU>
int launch(const __half* A, int lda, const __half* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
           int K32, hipStream_t stream) {
    using C_ = Cfg<NT>;
    const size_t lds = 2 * C_::STAGE_BYTES;
    static const int variant = getenv("CPN_GEMM_VARIANT") ? atoi(getenv("CPN_GEMM_VARIANT")) : 0;
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
    if (variant == 0)
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
    CPN_REQUIRE(ldc >= N && (ldc % 4) == 0, CPN_E_SHAPE, "cpn_gemm_f16: ldc=%d must be >= N and a multiple of 4", ldc);
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
