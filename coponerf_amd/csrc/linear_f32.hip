// K5 — exact-fp32 per-ray linear layers on the f32 MFMA (v_mfma_f32_16x16x4_f32, an fmaf chain bit for bit).
//
// Replaces nn.Conv1d encode_latent (/root/reference models/CoPoNeRF.py:468) and the light-field decoder
// lightfield.ResnetFC / ResnetBlockFC (/root/reference models/lightfield.py:52-61, 131-167): per QUERY RAY
// (not per sample) 850 -> 128 -> ... -> 3, 0.13 % of the path's FLOPs — kept in fp32 because its output is the
// rendered colour itself.
//
//   Y = act_out( act_in(X) . W^T + bias + res )      X (M,K)  W (N,K)  N <= 128, K % 16 == 0
//
// One wave = 16 rows x (16*NT) columns; operands swapped (A = W rows, B = X rows) so a lane holds 4 consecutive
// output columns of one row -> float4 stores.  Operands are read straight from L2 in the fragment layout as
// float4 (lane: row l&15, k = kb + 4*(l>>4) + e); the e-th element of every lane forms one k4 MFMA step, which
// is a fixed permutation of k shared by both operands.  No LDS: the weights (<= 213 KB) are L2-resident and the
// activations are read once.  Bound: L2 bandwidth on the W re-reads, irrelevant at 0.84 MFLOP per ray.
#include "common.h"

namespace {

// blockIdx.y selects a group of NT column tiles: at small M (the callers' 3 641-ray calls: 228 waves of NT = 8 on 1 024 SIMDs,
// each walking K as a chain of L2 round trips) the launch runs as 4 x as many waves of NT = 2 - 35 -> 13-19 us at K = 416,
// 12 -> 6-8 us at K = 128.  CPN_LIN_DEPTH k blocks are in flight per wave (2 = one block ahead of the MFMAs; 4 and 6 measured
// the same at small M and 2-17 % slower at M = 65 536, where NT = 8 then needs 202 VGPRs).  The k order of every output element
// is the same in every form: the results are bit-identical (test_small_call_forms_are_bit_identical).
#ifndef CPN_LIN_DEPTH
#define CPN_LIN_DEPTH 2
#endif
template <int NT>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ X, int ldx,
                                                         const float* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ res, int ldr,
                                                         float* __restrict__ Y, int ldy, int M, int N, int K,
                                                         int relu_in, int relu_out) {
    constexpr int D = CPN_LIN_DEPTH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x * 4 + wave) * 16;
    const int t0 = blockIdx.y * NT;                           // first 16-column tile of this workgroup
    if (m0 >= M) return;
    const int fi = lane & 15, fg = lane >> 4;
    int xr = m0 + fi;
    xr = xr < M ? xr : M - 1;
    const float* xp = X + (size_t)xr * ldx + fg * 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = (t0 + t) * 16 + fi;
        wp[t] = W + (size_t)(n < N ? n : N - 1) * ldw + fg * 4;
    }
    f32x4 xa[D], wa[D][NT];
    auto fetch = [&](int d, int kb) {
        xa[d] = *reinterpret_cast<const f32x4*>(xp + kb * 16);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wa[d][t] = *reinterpret_cast<const f32x4*>(wp[t] + kb * 16);
            if ((t0 + t) * 16 + fi >= N) wa[d][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int nkb = K >> 4;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < nkb) fetch(d, d);
    for (int kb0 = 0; kb0 < nkb; kb0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int kb = kb0 + d;
            if (kb < nkb) {
                f32x4 xv = xa[d];
                if (relu_in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = fmaxf(xv[e], 0.0f);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[d][t][e], xv[e], acc[t], 0, 0, 0);
                if (kb + D < nkb) fetch(d, kb + D);
            }
        }
    }
    const int m = m0 + fi;
    if (m >= M) return;
    const bool vec = ((N | ldy | ldr) & 3) == 0 &&               // 16-byte epilogue: a lane owns 4 consecutive columns
                     (((uintptr_t)Y | (uintptr_t)bias | (uintptr_t)res) & 15) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n0 = (t0 + t) * 16 + fg * 4;
        if (vec) {
            if (n0 >= N) continue;
            f32x4 v = acc[t];
            if (bias) v += *reinterpret_cast<const f32x4*>(bias + n0);
            if (res) v += *reinterpret_cast<const f32x4*>(res + (size_t)m * ldr + n0);
            if (relu_out) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
            }
            *reinterpret_cast<f32x4*>(Y + (size_t)m * ldy + n0) = v;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i;
            if (n >= N) continue;
            float v = acc[t][i];
            if (bias) v += bias[n];
            if (res) v += res[(size_t)m * ldr + n];
            if (relu_out) v = fmaxf(v, 0.0f);
            Y[(size_t)m * ldy + n] = v;
        }
    }
}

}  // namespace

extern "C" int cpn_linear_f32(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* res,
                              int ldr, float* Y, int ldy, int M, int N, int K, int relu_in, int relu_out,
                              void* stream) {
    CPN_REQUIRE(X && W && Y, CPN_E_ARG, "cpn_linear_f32: null pointer");
    CPN_REQUIRE(M > 0 && N > 0 && N <= 128 && K > 0 && (K % 16) == 0, CPN_E_SHAPE,
                "cpn_linear_f32: need N<=128 and K%%16==0 (got N=%d K=%d)", N, K);
    CPN_REQUIRE(ldx >= K && ldw >= K && (ldx % 4) == 0 && (ldw % 4) == 0 && ldy >= N && (!res || ldr >= N),
                CPN_E_SHAPE, "cpn_linear_f32: bad leading dimensions");
    CPN_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)W % 16) == 0, CPN_E_ARG, "cpn_linear_f32: X/W must be 16-B aligned");
    const hipStream_t s = (hipStream_t)stream;
    dim3 grid(cpn_cdiv(M, 64)), block(256);
    const int ntiles = (int)cpn_cdiv(N, 16);
    if (N <= 16)
        hipLaunchKernelGGL(linear_f32_kernel<1>, grid, block, 0, s, X, ldx, W, ldw, bias, res, ldr, Y, ldy, M, N, K,
                           relu_in, relu_out);
    else if (M <= 16384) {                                     // fewer than one wave of NT = 8 per SIMD: split the columns
        grid.y = cpn_cdiv(ntiles, 2);
        hipLaunchKernelGGL(linear_f32_kernel<2>, grid, block, 0, s, X, ldx, W, ldw, bias, res, ldr, Y, ldy, M, N, K,
                           relu_in, relu_out);
    } else
        hipLaunchKernelGGL(linear_f32_kernel<8>, grid, block, 0, s, X, ldx, W, ldw, bias, res, ldr, Y, ldy, M, N, K,
                           relu_in, relu_out);
    CPN_LAUNCH_CHECK("cpn_linear_f32");
    return 0;
}
