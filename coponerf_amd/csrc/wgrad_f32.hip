// Weight and bias gradients of the fp32 Linear layers of the training step (get_z's transformer blocks, the per-ray fp32
// layers of the render path), i.e. the dW / db of torch.nn.functional.linear under autograd
// (/root/reference models/aggregation.py:200-260 feed-forward and projection layers, models/CoPoNeRF.py:468):
//
//   dW (O, I) = dY^T (O, R) . X (R, I)        db (O) = column sums of dY        R = tokens (512 ... 32 768), O, I <= 2304
//
// The contraction index is the ROW of both row-major operands and R >> O, I: the library runs these as "TN" GEMMs with one
// workgroup per 128 x 128 output tile (a 256 x 1024 gradient is 16 workgroups on a 256-CU chip: 414 us, 41 TFLOP/s) and
// the bias sums as separate reductions.  Here the rows are split:
//
//   workgroup = 8 waves on ONE 64 x 64 output tile and one slab of rows; wave w takes the 4-row steps w, w + 8, ... of the
//               slab.  A lane (fi, fg) loads 16 bytes of row fg of the step from each operand (dY: columns o0 + 4 fi .. + 3,
//               X: i0 + 4 fi .. + 3 — a step reads 4 x 256 contiguous bytes per operand), and element e of those vectors is
//               a ready v_mfma_f32_16x16x4_f32 operand for the 16 output rows / columns {4 i + e}: 2 loads feed 16 MFMAs,
//               no LDS, no transpose (exact fp32 products, fp32 accumulation)
//   reduction = the 8 waves' accumulators are summed through LDS in a fixed order (4,5 into 0,1; 6,7 into 2,3; 2,3 into 0,1; 1 into 0),
//               wave 0 writes the slab's partial tile; a second kernel sums the slabs in slab order: deterministic
//   bias      = the tiles of the first column block also sum their dY operands (4 more accumulators per lane)
#include "common.h"

namespace {

constexpr int WF_WAVES = 8;
constexpr int WF_REGS = 68;                              // 64 accumulators + 4 bias sums per lane
constexpr int WF_STEP_ROWS = 4 * WF_WAVES;               // rows the workgroup consumes per round of steps

template <bool BIAS>
__global__ __launch_bounds__(64 * WF_WAVES, 1) void wgrad_f32_kernel(const float* __restrict__ dY, int ldy,
                                                                     const float* __restrict__ X, int ldx, long long R,
                                                                     int O, int I, int nto, int nti, long long slab_rows,
                                                                     float* __restrict__ part, float* __restrict__ bpart) {
    __shared__ float red[2 * WF_REGS * 64];                                       // 34 KiB: two waves' registers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles = nto * nti;
    const int slab = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int to = tile % nto, ti = tile / nto;
    const int o0 = to * 64, i0 = ti * 64;
    const long long r0 = slab * slab_rows;
    const long long r1 = r0 + slab_rows < R ? r0 + slab_rows : R;
    const bool ook = o0 + 4 * fi < O, iok = i0 + 4 * fi < I;
    const bool bias_tile = BIAS && ti == 0;
    // clamped addresses: lanes outside the matrix read (and discard) a valid element
    const float* ap = dY + (ook ? o0 + 4 * fi : 0);
    const float* bp = X + (iok ? i0 + 4 * fi : 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};

    const long long nsteps = (r1 - r0 + WF_STEP_ROWS - 1) / WF_STEP_ROWS;         // rounds of 8 x 4 rows
    // the zeroing of rows / columns outside the problem happens where the operands are USED: a select behind the load would
    // make the load's wait the loop's critical path
    const long long row_first = r0 + wave * 4 + fg;                                 // this lane's row of step 0
    const float* pa = ap + row_first * ldy;                                       // ... of the step being loaded
    const float* pb = bp + row_first * ldx;
    const long long sa = (long long)WF_STEP_ROWS * ldy, sb = (long long)WF_STEP_ROWS * ldx;
    long long row_ld = row_first;
    auto load = [&](f32x4& a, f32x4& b) {                                         // the next step in sequence
        const bool rok = row_ld < r1;
        a = *reinterpret_cast<const f32x4*>(rok ? pa : ap);
        b = *reinterpret_cast<const f32x4*>(rok ? pb : bp);
        pa += sa; pb += sb; row_ld += WF_STEP_ROWS;
        __builtin_amdgcn_sched_barrier(0);                                        // keep the load where it is written
    };
    auto mac = [&](long long s, f32x4 a, f32x4 b) {
        const bool rok = row_first + s * WF_STEP_ROWS < r1;
        if (!(rok && ook)) a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(rok && iok)) b = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
            for (int eb = 0; eb < 4; ++eb)
                acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ea], b[eb], acc[ea][eb], 0, 0, 0);
        if (bias_tile) bsum += a;
        __builtin_amdgcn_sched_barrier(0);
    };
    // operands of the next two steps are in flight under the 16 MFMAs (512 clocks) of the current one; three named buffers
    // (a rotation through copies would wait for the youngest load); steps past the slab's end re-read the first row of the
    // matrix and multiply zeros
    f32x4 a0, b0, a1, b1, a2, b2;
    load(a0, b0);
    load(a1, b1);
    for (long long s = 0; s < nsteps; s += 3) {
        load(a2, b2);
        mac(s, a0, b0);
        load(a0, b0);
        mac(s + 1, a1, b1);
        load(a1, b1);
        mac(s + 2, a2, b2);
    }

    // fixed-order tree over the waves
    float regs[WF_REGS];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) regs[(a * 4 + b) * 4 + r] = acc[a][b][r];
#pragma unroll
    for (int e = 0; e < 4; ++e) regs[64 + e] = bsum[e];
    // (source wave, destination wave, pairs): 4,5 -> 0,1;  6,7 -> 2,3;  2,3 -> 0,1;  1 -> 0
    constexpr int PH[4][3] = {{4, 0, 2}, {6, 2, 2}, {2, 0, 2}, {1, 0, 1}};
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int src0 = PH[ph][0], dst0 = PH[ph][1], n = PH[ph][2];
        if (wave >= src0 && wave < src0 + n) {
            float* dst = red + (size_t)(wave - src0) * WF_REGS * 64 + lane;
#pragma unroll
            for (int q = 0; q < WF_REGS; ++q) dst[q * 64] = regs[q];
        }
        __syncthreads();
        if (wave >= dst0 && wave < dst0 + n) {
            const float* src = red + (size_t)(wave - dst0) * WF_REGS * 64 + lane;
#pragma unroll
            for (int q = 0; q < WF_REGS; ++q) regs[q] += src[q * 64];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // D of tile (ea, eb), register r: output row o0 + 4 * (4 fg + r) + ea, columns i0 + 4 fi + eb
    float* dst = part + (size_t)slab * O * I;
    if (iok) {
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = o0 + 4 * (4 * fg + r) + ea;
                if (o < O)
                    *reinterpret_cast<f32x4*>(dst + (size_t)o * I + i0 + 4 * fi) =
                        f32x4{regs[(ea * 4 + 0) * 4 + r], regs[(ea * 4 + 1) * 4 + r], regs[(ea * 4 + 2) * 4 + r],
                              regs[(ea * 4 + 3) * 4 + r]};
            }
    }
    if (bias_tile) {
        // the four row groups of a step, in the order fg = 0, 1, 2, 3
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = regs[64 + e];
            const float v1 = __shfl(v, fi + 16), v2 = __shfl(v, fi + 32), v3 = __shfl(v, fi + 48);
            if (fg == 0 && ook) bpart[(size_t)slab * O + o0 + 4 * fi + e] = ((v + v1) + v2) + v3;
        }
    }
}

__global__ void wgrad_f32_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int nslab,
                                        long long n4, int O, float* __restrict__ dW, float* __restrict__ db) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n4) {
        f32x4 s = reinterpret_cast<const f32x4*>(part)[t];
        for (int k = 1; k < nslab; ++k) s += reinterpret_cast<const f32x4*>(part)[t + (long long)k * n4];
        reinterpret_cast<f32x4*>(dW)[t] = s;
    } else if (db != nullptr && t - n4 < O) {
        const long long o = t - n4;
        float s = bpart[o];
        for (int k = 1; k < nslab; ++k) s += bpart[o + (long long)k * O];
        db[o] = s;
    }
}

// slabs: enough workgroups for two per CU, at least 8 steps per wave
static inline int wf_slabs(long long R, int O, int I) {
    const long long tiles = (long long)cpn_cdiv(O, 64) * cpn_cdiv(I, 64);
    long long s = (512 + tiles - 1) / tiles;
    const long long cap = R / (8 * WF_STEP_ROWS);
    s = s > cap ? cap : s;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" long long cpn_wgrad_f32_scratch_floats(long long R, int O, int I) {
    return (long long)wf_slabs(R, O, I) * ((long long)O * I + O);
}

extern "C" int cpn_wgrad_f32(const float* dY, int ldy, const float* X, int ldx, long long R, int O, int I, float* dW,
                             float* db, float* scratch, void* stream) {
    CPN_REQUIRE(R > 0 && O > 0 && I > 0, 1, "cpn_wgrad_f32: empty problem (R=%lld O=%d I=%d)", R, O, I);
    CPN_REQUIRE(O % 4 == 0 && I % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldy >= O && ldx >= I, 1,
                "cpn_wgrad_f32: O, I and the row strides must be multiples of 4 (O=%d I=%d ldy=%d ldx=%d)", O, I, ldy, ldx);
    CPN_REQUIRE(((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dW | (uintptr_t)scratch) % 16 == 0, 1,
                "cpn_wgrad_f32: operands must be 16-byte aligned");
    const int nto = cpn_cdiv(O, 64), nti = cpn_cdiv(I, 64), nslab = wf_slabs(R, O, I);
    long long slab_rows = (R + nslab - 1) / nslab;
    slab_rows = (slab_rows + WF_STEP_ROWS - 1) / WF_STEP_ROWS * WF_STEP_ROWS;
    float* part = scratch;
    float* bpart = scratch + (size_t)nslab * O * I;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(nslab * nto * nti);
    if (db != nullptr)
        hipLaunchKernelGGL(wgrad_f32_kernel<true>, dim3(grid), dim3(64 * WF_WAVES), 0, st, dY, ldy, X, ldx, R, O, I, nto, nti,
                           slab_rows, part, bpart);
    else
        hipLaunchKernelGGL(wgrad_f32_kernel<false>, dim3(grid), dim3(64 * WF_WAVES), 0, st, dY, ldy, X, ldx, R, O, I, nto, nti,
                           slab_rows, part, bpart);
    CPN_LAUNCH_CHECK("cpn_wgrad_f32");
    const long long n4 = (long long)O * I / 4;
    hipLaunchKernelGGL(wgrad_f32_reduce_kernel, dim3(cpn_cdiv(n4 + (db ? O : 0), 256)), dim3(256), 0, st, part, bpart, nslab,
                       n4, O, dW, db);
    CPN_LAUNCH_CHECK("cpn_wgrad_f32 (reduce)");
    return 0;
}
