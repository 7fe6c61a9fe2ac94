// Weight gradient of the first encoder layer (training, BASELINE config 3):
//
//   dW (832, 896) = dY^T (832, M) . X (M, 896),  M = B*R*V*S*2 = 4.2 M rows, fp16 operands, fp32 result
//
// i.e. the dW of query_encode_latent (/root/reference models/CoPoNeRF.py:437-438 under autograd): 6.3 TFLOP over 14 GB of
// operands whose contraction index is the ROW of both row-major operands.  The library runs it as a split-K "TN" GEMM at
// 500 TFLOP/s (12.5 ms, the largest single launch of the training step).  On gfx950 the row-major tiles can feed the MFMA
// directly: ds_read_b64_tr_b16 reads a 4 (rows) x 16 (columns) block of 16-bit elements from LDS and hands lane n the 4
// row-consecutive values of column n — exactly the K-packed operand layout of v_mfma_f32_16x16x32_f16 (two reads per
// fragment), so both operands go global -> LDS (16-byte rows, as stored) -> fragments with no transpose pass.
//
//   workgroup = 4 waves, output tile 208 (n) x 128 (k): every wave owns 208 x 32 = 13 x 2 MFMA tiles (104 accumulators).
//               The LDS *write* path (~80 B/clk, 13 cycles per ds_write_b128) is the scarce resource, transposing reads cost
//               2 cycles each: the tile is chosen for MFMA work per staged byte (79 FLOP/B), not for few fragment reads
//               (a first 64 x 448 tile, 56 FLOP/B, spent 70 % of its time in LDS traffic: tools/wgrad_bench.py ablations)
//   fragments = lane (fi, fg) takes rows fg*4 .. +3 and 16 + fg*4 .. +3 of a 32-row step (any row <-> k assignment works as
//               long as both operands use the same one): a 32-lane read group then touches 8 CONSECUTIVE rows, which the
//               row strides (416 B / 288 B, odd multiples of 32 B) spread over all 64 banks
//   grid      = 4 x 7 output tiles x 16 row slabs (832 x 896; 8 x 1 x 64 for the transposed 1664 x 128 key-map gradient); the
//               tiles of a slab sit on ONE XCD (workgroup id % 8) and walk the
//               slab's rows together, so each operand tile comes from HBM once and from that XCD's L2 afterwards (cached
//               loads: non-temporal ones re-fetched every tile's operands, 3x slower)
//   result    = per-slab partial sums, summed in a fixed order by a second kernel (deterministic; also applies 1/scale)
#include "common.h"

// timing-only phase ablations for tools/wgrad_bench.py (results are wrong): 1 no global loads, 2 no LDS stage writes,
// 4 no fragment reads, 8 no MFMA
#ifndef CPN_WT_ABLATE
#define CPN_WT_ABLATE 0
#endif

namespace {

constexpr int WT_ROWS = 64;                 // rows per LDS stage (two 32-row MFMA steps)
constexpr int WT_N = 208, WT_K = 128;       // output tile of a workgroup: 13 x 8 MFMA tiles
constexpr int WT_LDA = 208;                 // dY stage row stride in halves: 416 B = 13 x 32 B
constexpr int WT_LDB = 144;                 // X stage row stride in halves: 288 B = 9 x 32 B
// row slabs: as many as fill ~512 workgroup slots (2 per CU), a multiple of 8 so that every XCD holds whole slabs
static inline int wt_slabs(int N, int K) {
    const int tiles = (N / WT_N) * (K / WT_K);
    const int s = (512 / tiles) / 8 * 8;
    return s < 8 ? 8 : (s > 64 ? 64 : s);
}
constexpr int WT_ASEGS = WT_N / 8, WT_BSEGS = WT_K / 8;      // 16-byte segments per stage row

typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v lds_short4;

__device__ __forceinline__ half8 read_fragment(const _Float16* tile, int ld, int row, int col) {
    // lane (fi, fg) gets rows row .. row+3 and row+16 .. row+19 of column col - 4 * (fi & 3) + fi (header)
    const half4 lo = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(tile + row * ld + col)));
    const half4 hi = __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(tile + (row + 16) * ld + col)));
    return half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 2) void wgrad_tall_f16_kernel(const _Float16* __restrict__ dY, int ldy,
                                                                const _Float16* __restrict__ X, int ldx, long long M,
                                                                int ntn, int ntk, int nslab, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) _Float16 sa[WT_ROWS * WT_LDA];
    __shared__ __attribute__((aligned(16))) _Float16 sb[WT_ROWS * WT_LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    // workgroup -> (slab, tile): ids congruent mod 8 share an XCD; an XCD holds nslab / 8 slabs x all tiles
    const int tiles = ntn * ntk;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slab = xcd * (nslab / 8) + j / tiles, tile = j % tiles;
    const int tn = tile % ntn, tk = tile / ntn;
    const long long per = ((M + nslab - 1) / nslab + WT_ROWS - 1) / WT_ROWS * WT_ROWS;
    const long long row0 = slab * per, row1 = row0 + per < M ? row0 + per : M;

    f32x4 acc[13][2];
#pragma unroll
    for (int a = 0; a < 13; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // loaders: dY stage = 64 rows x 26 segments, 8 rows per pass by 208 of the 256 threads; X stage = 64 rows x 16 segments,
    // 16 rows per pass (constant row stride between passes: one address per operand per thread)
    u32x4 ra[8], rb[4];
    const bool aload = tid < 8 * WT_ASEGS;
    const int at = aload ? tid : tid - 8 * WT_ASEGS;            // the 48 spare threads shadow a load and drop it
    const int arow = at / WT_ASEGS, acol = (at - arow * WT_ASEGS) * 8;
    const int brow = tid >> 4, bcol = (tid & 15) * 8;
    const _Float16* pa = dY + (size_t)tn * WT_N + acol;
    const _Float16* pb = X + (size_t)tk * WT_K + bcol;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
    auto fetch = [&](long long r0) {
        if (CPN_WT_ABLATE & 1) {
#pragma unroll
            for (int p = 0; p < 8; ++p) ra[p] = u32x4{(unsigned)r0, 1u, 2u, (unsigned)p};
#pragma unroll
            for (int p = 0; p < 4; ++p) rb[p] = u32x4{(unsigned)r0, 1u, 2u, (unsigned)p};
            return;
        }
        // rows past the slab are read from its last row (always in bounds) and zeroed: no divergent loads.  The selects make
        // the compiler wait for each load right here, i.e. BEFORE the MFMA phase; moving them into stage() lets the loads fly
        // under the MFMAs (loads + fragments + MFMA without staging: 5.9 -> 4.6 ms) and yet the whole kernel gets slower
        // (9.0 -> 11.3 ms; a select-free fast path for whole stages: 10.4 ms) — measured, not understood
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long long r = r0 + arow + 8 * p;
            const u32x4 v = *reinterpret_cast<const u32x4*>(pa + (r < row1 ? r : row1 - 1) * ldy);
            ra[p] = r < row1 ? v : zero4;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long long r = r0 + brow + 16 * p;
            const u32x4 v = *reinterpret_cast<const u32x4*>(pb + (r < row1 ? r : row1 - 1) * ldx);
            rb[p] = r < row1 ? v : zero4;
        }
    };
    auto stage = [&]() {
        if (CPN_WT_ABLATE & 2) {
            if (ra[0][0] == 0x12345u && rb[3][3] == 0x777u) sa[tid] = (_Float16)1.0f;      // keeps the loads alive
            return;
        }
        if (aload) {
#pragma unroll
            for (int p = 0; p < 8; ++p) *reinterpret_cast<u32x4*>(sa + (arow + 8 * p) * WT_LDA + acol) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(sb + (brow + 16 * p) * WT_LDB + bcol) = rb[p];
    };

    // one LDS stage, the next one in registers: its loads are issued right after the stage barrier and have a whole
    // 52-MFMA phase (plus the co-resident workgroup's) to land.  A double-buffered 32-row variant (one barrier per stage,
    // loads one 26-MFMA phase ahead) was 30 % slower, and so were L2 prefetches of the lines four stages ahead
    if (row0 < row1) fetch(row0);
    for (long long r0 = row0; r0 < row1; r0 += WT_ROWS) {
        __syncthreads();                                      // the previous stage's fragments are read
        stage();
        __syncthreads();
        if (r0 + WT_ROWS < row1) fetch(r0 + WT_ROWS);         // in flight under the MFMAs below
#pragma unroll
        for (int ks = 0; ks < WT_ROWS / 32; ++ks) {
            const int row = ks * 32 + fg * 4 + (fi >> 2), c4 = 4 * (fi & 3);
            half8 fb[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
                fb[b] = (CPN_WT_ABLATE & 4) ? half8{(_Float16)(float)b, 1, 2, 3, 4, 5, 6, 7}
                                            : read_fragment(sb, WT_LDB, row, wave * 32 + b * 16 + c4);
#pragma unroll
            for (int a = 0; a < 13; ++a) {
                const half8 fa = (CPN_WT_ABLATE & 4) ? half8{(_Float16)(float)r0, 1, 2, 3, 4, 5, 6, 7}
                                                     : read_fragment(sa, WT_LDA, row, a * 16 + c4);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (CPN_WT_ABLATE & 8) acc[a][b][0] += (float)fa[b] * (float)fb[b][a & 7];
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb[b], acc[a][b], 0, 0, 0);
                }
            }
        }
    }
    // tile (a, b) of lane (fi, fg): rows n = a*16 + fg*4 + i, column k = b*16 + fi
    const int ldw = ntk * WT_K;
    float* out = part + ((size_t)slab * ntn * WT_N + (size_t)tn * WT_N) * ldw + (size_t)tk * WT_K + wave * 32;
#pragma unroll
    for (int a = 0; a < 13; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(size_t)(a * 16 + fg * 4 + i) * ldw + b * 16 + fi] = acc[a][b][i];
}

// dW = (sum over slabs, in order) / scale[0]
__global__ __launch_bounds__(256) void wgrad_tall_reduce_kernel(const float* __restrict__ part, long long n4, int nslab,
                                                                const float* __restrict__ scale, float* __restrict__ dW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
#pragma unroll 8
    for (int q = 1; q < nslab; ++q) s += reinterpret_cast<const f32x4*>(part)[q * n4 + i];
    const float inv = scale ? 1.0f / scale[0] : 1.0f;
    reinterpret_cast<f32x4*>(dW)[i] = s * inv;
}

}  // namespace

extern "C" long long cpn_wgrad_tall_scratch(int N, int K) {
    return (N < WT_N || K < WT_K || N % WT_N || K % WT_K) ? 0 : (long long)wt_slabs(N, K) * N * K;
}

extern "C" int cpn_wgrad_tall_f16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, long long M, int N, int K,
                                  const float* scale, float* part, float* dW, void* stream) {
    CPN_REQUIRE(dY && X && part && dW, CPN_E_ARG, "cpn_wgrad_tall_f16: null pointer");
    CPN_REQUIRE(M > 0 && N > 0 && K > 0 && N % WT_N == 0 && K % WT_K == 0 && ldy >= N && ldx >= K && ldy % 8 == 0 &&
                    ldx % 8 == 0, CPN_E_SHAPE,
                "cpn_wgrad_tall_f16: need N %% 208 == 0, K %% 128 == 0 and 16-byte rows (N=%d K=%d ldy=%d ldx=%d)", N, K, ldy, ldx);
    CPN_REQUIRE(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)part % 16) == 0 &&
                    ((uintptr_t)dW % 16) == 0, CPN_E_ARG, "cpn_wgrad_tall_f16: operands must be 16-byte aligned");
    const hipStream_t s = (hipStream_t)stream;
    const int ntn = N / WT_N, ntk = K / WT_K, nslab = wt_slabs(N, K);
    hipLaunchKernelGGL(wgrad_tall_f16_kernel, dim3(nslab * ntn * ntk), dim3(256), 0, s, (const _Float16*)dY, ldy,
                       (const _Float16*)X, ldx, M, ntn, ntk, nslab, part);
    const long long n4 = (long long)N * K / 4;
    hipLaunchKernelGGL(wgrad_tall_reduce_kernel, dim3((unsigned)cpn_cdiv(n4, 256)), dim3(256), 0, s, (const float*)part, n4,
                       nslab, scale, dW);
    CPN_LAUNCH_CHECK("cpn_wgrad_tall_f16");
    return 0;
}
