// Backward kernels of the render path (training, BASELINE config 3).
//
// The plain GEMM gradients (dX = dY W, dW = dY^T X) go to hipBLASLt through torch.matmul on the host side; this
// unit holds the two stages that are not plain GEMMs:
//
//   cpn_attend_hidden_bwd  gradient of the joint softmax + attention-weighted hidden sum (cpn_attend_hidden),
//                          i.e. of /root/reference models/CoPoNeRF.py:450-461 / 475-485 in the folded form
//   cpn_gather_rows_bwd    gradient of the bilinear multi-scale gather w.r.t. the feature maps (scatter-add),
//                          i.e. of F.grid_sample at models/CoPoNeRF.py:312 / 370 (no coordinate gradient: all sample
//                          coordinates derive from poses only and `pt` is detached, CoPoNeRF.py:380-381, 433)
#include "common.h"

namespace {

constexpr int HC = 1664;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per ray.  T = V*S rows.
//   dw[row]   = <hid[row], dhbar> + dw_ext[row]
//   dl[row]   = w[row] * (dw[row] - sum_r w[r] dw[r]) / 11.31
//   dqa[row]  = dl[row] * qb[row] ;  dqb[row] = dl[row] * qa[row]
//   dhid[row] = w[row] * dhbar
__global__ __launch_bounds__(256) void attend_hidden_bwd_kernel(
    const __half* __restrict__ qa, const __half* __restrict__ qb, const __half* __restrict__ hid,
    const float* __restrict__ at_wt, const float* __restrict__ dhbar, const float* __restrict__ dw_ext, int V, int R,
    int S, int ray0, __half* __restrict__ dqa, __half* __restrict__ dqb, __half* __restrict__ dhid) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* dh = reinterpret_cast<float*>(smem_raw);          // HC floats: dhbar of this ray
    float* wts = dh + HC;                                    // T
    float* dl = wts + V * S;                                 // T
    float* red = dl + V * S;                                 // 4
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lray = blockIdx.x;
    const size_t row0 = (size_t)lray * T;
    const unsigned ray = (unsigned)ray0 + lray;
    const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);

    for (int c = tid; c < HC; c += 256) dh[c] = dhbar[(size_t)lray * HC + c];
    for (int row = tid; row < T; row += 256) {
        const int v = row / S, s = row - v * S;
        wts[row] = at_wt[(((size_t)(b * V + v)) * R + r) * S + s];
    }
    __syncthreads();
    // dw: one wave per row, 64 lanes x 26 elements
    float part = 0.f;
    for (int row = wave; row < T; row += 4) {
        const __half* hp = hid + (row0 + row) * HC;
        float acc = 0.f;
        for (int c = lane * 2; c < HC; c += 128) {
            const __half2 h2 = *reinterpret_cast<const __half2*>(hp + c);
            acc += __low2float(h2) * dh[c] + __high2float(h2) * dh[c + 1];
        }
        acc = wave_sum_f(acc);
        if (lane == 0) {
            float dwv = acc;
            if (dw_ext) {
                const int v = row / S, s = row - v * S;
                dwv += dw_ext[(((size_t)(b * V + v)) * R + r) * S + s];
            }
            dl[row] = dwv;
            part += wts[row] * dwv;
        }
    }
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int row = tid; row < T; row += 256) dl[row] = wts[row] * (dl[row] - dot) / 11.31f;
    __syncthreads();
    // dqa / dqb: thread = (row, 8-channel group)
    for (int i = tid; i < T * 16; i += 256) {
        const int row = i >> 4, g = i & 15;
        const half8 a = *reinterpret_cast<const half8*>(qa + (row0 + row) * 128 + g * 8);
        const half8 bq = *reinterpret_cast<const half8*>(qb + (row0 + row) * 128 + g * 8);
        const float d = dl[row];
        half8 oa, ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            oa[e] = (_Float16)(d * (float)bq[e]);
            ob[e] = (_Float16)(d * (float)a[e]);
        }
        *reinterpret_cast<half8*>(dqa + (row0 + row) * 128 + g * 8) = oa;
        *reinterpret_cast<half8*>(dqb + (row0 + row) * 128 + g * 8) = ob;
    }
    // dhid = w[row] * dhbar : thread = 8 channels, rows streamed
    if (tid < HC / 8) {
        float d8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d8[e] = dh[tid * 8 + e];
        for (int row = 0; row < T; ++row) {
            const float w = wts[row];
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)(w * d8[e]);
            *reinterpret_cast<half8*>(dhid + (row0 + row) * HC + tid * 8) = o;
        }
    }
}

// ---- bilinear taps exactly as the forward gather (gather.hip: make_taps) -----------------------------------
struct Taps {
    int off[4];
    float w[4];
};
__device__ __forceinline__ Taps make_taps(float gx, float gy, int Wl, int Hl, bool border) {
    float x = ((gx + 1.0f) * (float)Wl - 1.0f) / 2.0f;
    float y = ((gy + 1.0f) * (float)Hl - 1.0f) / 2.0f;
    if (border) {
        x = fminf(fmaxf(x, 0.0f), (float)(Wl - 1));
        y = fminf(fmaxf(y, 0.0f), (float)(Hl - 1));
    } else {
        x = fminf(fmaxf(x, -2.0f), (float)Wl + 1.0f);
        y = fminf(fmaxf(y, -2.0f), (float)Hl + 1.0f);
    }
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float fx = x - xf, fy = y - yf;
    Taps t;
    const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xi = x0 + i, yi = y0 + j;
            const bool inside = (xi >= 0) && (xi <= Wl - 1) && (yi >= 0) && (yi <= Hl - 1);
            const int xc = min(max(xi, 0), Wl - 1), yc = min(max(yi, 0), Hl - 1);
            t.off[j * 2 + i] = yc * Wl + xc;
            t.w[j * 2 + i] = inside ? wx[i] * wy[j] : 0.0f;
        }
    return t;
}

// thread = (row, 16-byte chunk < 104) like the forward; 4 taps x 8 channels of float atomics into NHWC fp32 maps
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(
    const __half* __restrict__ dxin, int ldx, int H, int W, const float* __restrict__ pixel_val,
    const float* __restrict__ sec_grid, int V, int R, int S, int ray0, long long nrows, float* __restrict__ dmap0,
    float* __restrict__ dmap1, float* __restrict__ dmap2, float* __restrict__ dmap3) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned row = gid / 104u;
    const int chunk = (int)(gid - row * 104u);
    if (row >= (unsigned)nrows) return;
    const int j = (int)(row & 1);
    unsigned t = row >> 1;
    const int s = (int)(t % (unsigned)S); t /= (unsigned)S;
    const int v = (int)(t % (unsigned)V); t /= (unsigned)V;
    const unsigned ray = (unsigned)ray0 + t;
    const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
    const size_t sidx = (((size_t)(b * V + v)) * R + r) * S + s;
    int lvl, c8;
    if (chunk < 96) { lvl = chunk >> 5; c8 = chunk & 31; } else { lvl = 3; c8 = chunk - 96; }
    const int shift = 4 - lvl - (lvl == 3);
    const int Hl = H >> shift, Wl = W >> shift;
    const int C = (lvl == 3) ? 64 : 256;
    float* base = (lvl == 0) ? dmap0 : (lvl == 1) ? dmap1 : (lvl == 2) ? dmap2 : dmap3;
    const float* g = (j == 0 ? pixel_val : sec_grid) + sidx * 2;
    const int img = b * V + (j == 0 ? v : (V - 1 - v));
    const Taps tp = make_taps(g[0], g[1], Wl, Hl, j == 0);
    const half8 d = *reinterpret_cast<const half8*>(dxin + (size_t)row * ldx + chunk * 8);
    float* m = base + (size_t)img * Hl * Wl * C + c8 * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (tp.w[k] == 0.0f) continue;
        float* p = m + (size_t)tp.off[k] * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(p + e, (float)d[e] * tp.w[k]);
    }
}

}  // namespace

extern "C" int cpn_attend_hidden_bwd(const uint16_t* qa, const uint16_t* qb, const uint16_t* hid, const float* at_wt,
                                     const float* dhbar, const float* dw_ext, int B, int V, int R, int S, int ray0,
                                     int nrays, uint16_t* dqa, uint16_t* dqb, uint16_t* dhid, void* stream) {
    CPN_REQUIRE(qa && qb && hid && at_wt && dhbar && dqa && dqb && dhid, CPN_E_ARG, "cpn_attend_hidden_bwd: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 2048, CPN_E_SHAPE, "cpn_attend_hidden_bwd: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_hidden_bwd: ray range outside B*R");
    const size_t lds = (size_t)(HC + 2 * V * S + 4) * sizeof(float);
    hipLaunchKernelGGL(attend_hidden_bwd_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream,
                       (const __half*)qa, (const __half*)qb, (const __half*)hid, at_wt, dhbar, dw_ext, V, R, S, ray0,
                       (__half*)dqa, (__half*)dqb, (__half*)dhid);
    CPN_LAUNCH_CHECK("cpn_attend_hidden_bwd");
    return 0;
}

extern "C" int cpn_gather_rows_bwd(const uint16_t* dxin, int ldx, int H, int W, const float* pixel_val,
                                   const float* sec_grid, int B, int V, int R, int S, int ray0, int nrays,
                                   float* dmap0, float* dmap1, float* dmap2, float* dmap3, void* stream) {
    CPN_REQUIRE(dxin && pixel_val && sec_grid && dmap0 && dmap1 && dmap2 && dmap3, CPN_E_ARG,
                "cpn_gather_rows_bwd: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && (H % 16) == 0 && (W % 16) == 0 && ldx >= 832,
                CPN_E_SHAPE, "cpn_gather_rows_bwd: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_gather_rows_bwd: ray range outside B*R");
    const long long nrows = (long long)nrays * V * S * 2;
    const long long total = nrows * 104;
    CPN_REQUIRE(total < (1LL << 31), CPN_E_SHAPE, "cpn_gather_rows_bwd: chunk too large for 32-bit indexing");
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)dxin, ldx, H, W, pixel_val, sec_grid, V, R, S, ray0, nrows, dmap0, dmap1, dmap2,
                       dmap3);
    CPN_LAUNCH_CHECK("cpn_gather_rows_bwd");
    return 0;
}
