// Layout / packing helpers of the render path and the fp32 first layer of the attention MLPs on the training path.
//
// (Rounds 1-5 kept the bilinear multi-scale gather of /root/reference models/CoPoNeRF.py:312, 370, 384-394 here as a kernel of
// its own - cpn_gather_rows, the 835-channel encoder input as fp16 rows - beside the node-table form of the first layer that
// replaced it in round 2, and cpn_local_mlp, the row-order query MLPs that cpn_local_units replaced in round 5; both left the
// library in round 6: tools/experiments/r6_pruned/gather.hip.)
//
// Layout: feature maps are NHWC fp16 so that one bilinear tap of one level is ONE contiguous 512-B (256 ch) or 128-B (64 ch)
// segment.
#include <algorithm>

#include "common.h"
#include "taps.h"

namespace {

// ---------------------------------------------------------------------------------------------
// (N,C,h,w) fp32 -> (N,h,w,C) fp16, 32x32 LDS tile transpose over (C, h*w)
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_f16_kernel(const float* __restrict__ src, __half* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;                 // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < C) dst[((size_t)n * HW + p) * C + c] = __float2half(tile[tx][j]);
    }
}

__global__ void pack_weight_f16_kernel(const float* __restrict__ src, int n_out, int k_in,
                                       __half* __restrict__ dst, int ld) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_out * ld) return;
    const int k = (int)(idx % ld), n = (int)(idx / ld);
    dst[idx] = __float2half(k < k_in ? src[(size_t)n * k_in + k] : 0.0f);
}

// ---------------------------------------------------------------------------------------------
// first (16 -> 128) layer of query_embed / query_repeat_embed in fp32, output fp16 rows.
// One wave = 16 rows per step on the fp32 matrix cores: D(128 ch x 16 rows) = W(128 x 16) . L^T(16 x 16 rows) as 8 output
// tiles x 4 v_mfma_f32_16x16x4_f32 (the VALU form, 128 FMA per 16-byte store, ran at 20 % of the fp32 VALU peak:
// 0.54 ms per launch).  Weights are the MFMA A operand and stay in registers for the whole kernel; the rows of tile
// t = 2p+h are assigned to channels p*32 + (a/4)*8 + h*4 + a%4, so a lane ends up with 8 CONSECUTIVE channels of one
// row per tile pair (16-byte stores, 64 contiguous bytes per row and pair).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void local_hidden_kernel(
    const float* __restrict__ loc8, const float* __restrict__ coords9, const float* __restrict__ w, int ldw,
    const float* __restrict__ bias, const float* __restrict__ add, int V, int R, int S, int ray0,
    long long nrows, __half* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int a = lane & 15, fg = lane >> 4;
    f32x4 wv[8];                  // wv[t][e] = W[channel(t, a)][fg*4 + e]
    f32x4 bv[8];                  // bias of the 4 channels this lane owns in tile t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int p = t >> 1, h = t & 1;
        const int ch_a = p * 32 + (a >> 2) * 8 + h * 4 + (a & 3);
        wv[t] = *reinterpret_cast<const f32x4*>(w + (size_t)ch_a * ldw + fg * 4);
        bv[t] = *reinterpret_cast<const f32x4*>(bias + p * 32 + fg * 8 + h * 4);
    }
    const unsigned ngroups = (unsigned)((nrows + 15) >> 4);
    const unsigned wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (unsigned grp = wave_id; grp < ngroups; grp += nwaves) {
        const unsigned row = grp * 16 + a;
        const bool live = row < (unsigned)nrows;
        unsigned t_ = live ? row : (unsigned)nrows - 1;            // 32-bit: see gather_rows_kernel
        const int s = (int)(t_ % (unsigned)S); t_ /= (unsigned)S;
        const int v = (int)(t_ % (unsigned)V); t_ /= (unsigned)V;
        const unsigned ray = (unsigned)ray0 + t_;
        const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
        const size_t nr = ((size_t)(b * V + v)) * R + r;
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(loc8 + (nr * S + s) * 8);
        const f32x4 l1 = *reinterpret_cast<const f32x4*>(loc8 + (nr * S + s) * 8 + 4);
        const float* c9 = coords9 + nr * 9;
        // local_coords channel order (CoPoNeRF.py:445): ctx dir 0-2, zeros 3-5, query dir 6-8, depth 9-12, origin 13-15;
        // this lane feeds k = fg*4 .. fg*4+3
        f32x4 lv;
        if (fg == 0) lv = f32x4{l0[0], l0[1], l0[2], 0.f};
        else if (fg == 1) lv = f32x4{0.f, 0.f, c9[0], c9[1]};
        else if (fg == 2) lv = f32x4{c9[2], l0[3], l1[0], l1[1]};
        else lv = f32x4{l1[2], c9[6], c9[7], c9[8]};
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = bv[t];
            if (add) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(add + (size_t)t_ * 128 + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
                acc[t] += av;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][e], lv[e], acc[t], 0, 0, 0);
        if (live) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                half8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (_Float16)fmaxf(acc[2 * p][i], 0.0f);
                    o[4 + i] = (_Float16)fmaxf(acc[2 * p + 1][i], 0.0f);
                }
                *reinterpret_cast<half8*>(out + (size_t)row * 128 + p * 32 + fg * 8) = o;
            }
        }
    }
}

}  // namespace

extern "C" int cpn_nchw_to_nhwc_f16(const float* src, uint16_t* dst, int N, int C, int h, int w, void* stream) {
    CPN_REQUIRE(src && dst, CPN_E_ARG, "cpn_nchw_to_nhwc_f16: null pointer");
    CPN_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && N < 65536, CPN_E_SHAPE, "cpn_nchw_to_nhwc_f16: bad shape");
    const int HW = h * w;
    dim3 grid(cpn_cdiv(HW, 32), cpn_cdiv(C, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_f16_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, src, (__half*)dst, C, HW);
    CPN_LAUNCH_CHECK("cpn_nchw_to_nhwc_f16");
    return 0;
}

extern "C" int cpn_pack_weight_f16(const float* src, int n_out, int k_in, uint16_t* dst, int ld, void* stream) {
    CPN_REQUIRE(src && dst, CPN_E_ARG, "cpn_pack_weight_f16: null pointer");
    CPN_REQUIRE(n_out > 0 && k_in > 0 && ld >= k_in, CPN_E_SHAPE, "cpn_pack_weight_f16: bad shape");
    const long long total = (long long)n_out * ld;
    hipLaunchKernelGGL(pack_weight_f16_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, n_out, k_in, (__half*)dst, ld);
    CPN_LAUNCH_CHECK("cpn_pack_weight_f16");
    return 0;
}

extern "C" int cpn_local_hidden(const float* loc8, const float* coords9, const float* w, int ldw, const float* bias,
                                const float* add, int B, int V, int R, int S, int ray0, int nrays, uint16_t* out,
                                void* stream) {
    CPN_REQUIRE(loc8 && coords9 && w && bias && out, CPN_E_ARG, "cpn_local_hidden: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && ldw >= 16, CPN_E_SHAPE, "cpn_local_hidden: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_local_hidden: ray range outside B*R");
    const long long nrows = (long long)nrays * V * S;
    CPN_REQUIRE(nrows * 16 < (1LL << 31), CPN_E_SHAPE, "cpn_local_hidden: chunk too large for 32-bit indexing");
    const long long groups = cpn_cdiv(nrows, 16);                   // 16 rows per wave step, 4 waves per block
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(groups, 4), 2048);
    hipLaunchKernelGGL(local_hidden_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       loc8, coords9, w, ldw, bias, add, V, R, S, ray0, nrows, (__half*)out);
    CPN_LAUNCH_CHECK("cpn_local_hidden");
    return 0;
}
