// K4 — joint softmax over the V*S epipolar samples of one query ray + attention-weighted value sum.
//
// Replaces einsum('bijk,bijk->bjk') / 11.31, F.softmax over (n_context * npoints), the broadcast
// multiply-sum and the sum over views (/root/reference models/CoPoNeRF.py:450-461 and 475-485).
// The reference has NO alpha compositing: the along-ray reduction is (max, sum-exp, weighted sum), which
// maps onto wave reductions — with S = 64 each view's sample axis is exactly one 64-lane wavefront.
//
// One 256-thread workgroup per query ray: T = V*S rows.
//   phase 1  logits: 2 lanes per row, 64 fp16 products each (8 x 16-B loads per operand), fp32 accumulate
//   phase 2  block max / sum-exp through LDS (fp32), weights to LDS
//   phase 3  z[c] = sum_rows w[row] * value[row][c], c = tid and tid+256 (416 channels), coalesced row reads
// HBM/L2-bound: per ray 2 x T x 256 B (q operands) + T x 1664 B (values) = 278 KiB at T = 128.
#include <algorithm>

#include "common.h"

namespace {

constexpr int CH = 416;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void attend_kernel(const __half* __restrict__ qa, const __half* __restrict__ qb,
                                                     const float* __restrict__ value,
                                                     const float* __restrict__ zprev, int V,
                                                     int R, int S, int ray0, float* __restrict__ zout,
                                                     float* __restrict__ at_wt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wts = reinterpret_cast<float*>(smem_raw);          // T weights
    float* red = wts + V * S;                                 // 8 floats of reduction scratch
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long lray = blockIdx.x;                        // ray inside the chunk
    const size_t row0 = (size_t)lray * T;

    // ---- phase 1: logits (CoPoNeRF.py:450 / :475) — the division by 11.31 is kept a division
    float lmax = -INFINITY;
    for (int base = 0; base < T; base += 128) {
        const int row = base + (tid >> 1);
        if (row < T) {
            const int hsel = tid & 1;
            const half8* pa = reinterpret_cast<const half8*>(qa + (row0 + row) * 128 + hsel * 64);
            const half8* pb = reinterpret_cast<const half8*>(qb + (row0 + row) * 128 + hsel * 64);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const half8 a = pa[k], b = pb[k];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
            }
            acc += __shfl_xor(acc, 1);
            const float logit = acc / 11.31f;
            if (hsel == 0) wts[row] = logit;
            lmax = fmaxf(lmax, logit);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // ---- phase 2: exp / sum
    float lsum = 0.f;
    for (int row = tid; row < T; row += 256) {
        const float e = __expf(wts[row] - gmax);
        wts[row] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int row = tid; row < T; row += 256) {
        const float w = wts[row] * inv;
        wts[row] = w;
        if (at_wt) {                                           // (N,R,S) layout, n = b*V + v
            const unsigned ray = (unsigned)ray0 + (unsigned)lray;
            const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
            const int v = row / S, s = row - v * S;
            at_wt[(((size_t)(b * V + v)) * R + r) * S + s] = w;
        }
    }
    __syncthreads();
    // ---- phase 3: weighted value sum, per-view partial sums added view by view (CoPoNeRF.py:456-461)
    for (int c = tid; c < CH; c += 256) {
        float total = 0.f;
        for (int v = 0; v < V; ++v) {
            float acc = 0.f;
            const float* vp = value + (row0 + (size_t)v * S) * CH + c;
#pragma unroll 8
            for (int s = 0; s < S; ++s) acc += wts[v * S + s] * vp[(size_t)s * CH];
            if (zprev) acc += zprev[(size_t)lray * CH + c];   // round 2: the round-1 vector sits in every view slot (:481-485)
            total += acc;
        }
        zout[(size_t)lray * CH + c] = total;
    }
}

// The same round with fp32 query operands (round 5, RenderEngine(precision="f32"): every per-sample layer in fp32 - the
// reference's arithmetic - as an opt-in verification mode): logits from (rows,128) fp32 matrices, everything else as above.
__global__ __launch_bounds__(256) void attend_f32_kernel(const float* __restrict__ qa, const float* __restrict__ qb,
                                                         const float* __restrict__ value, const float* __restrict__ zprev,
                                                         int V, int R, int S, int ray0, float* __restrict__ zout,
                                                         float* __restrict__ at_wt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wts = reinterpret_cast<float*>(smem_raw);
    float* red = wts + V * S;
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long lray = blockIdx.x;
    const size_t row0 = (size_t)lray * T;
    float lmax = -INFINITY;
    for (int base = 0; base < T; base += 8) {                  // 32 lanes per row, 4 floats each
        const int row = base + (tid >> 5);
        const int rr = row < T ? row : T - 1;
        const f32x4 a = *reinterpret_cast<const f32x4*>(qa + (row0 + rr) * 128 + (tid & 31) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(qb + (row0 + rr) * 128 + (tid & 31) * 4);
        float acc = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (row < T) {
            const float logit = acc / 11.31f;
            if ((tid & 31) == 0) wts[row] = logit;
            lmax = fmaxf(lmax, logit);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int row = tid; row < T; row += 256) {
        const float e = __expf(wts[row] - gmax);
        wts[row] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int row = tid; row < T; row += 256) {
        const float w = wts[row] * inv;
        wts[row] = w;
        if (at_wt) {
            const unsigned ray = (unsigned)ray0 + (unsigned)lray;
            const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
            const int v = row / S, s = row - v * S;
            at_wt[(((size_t)(b * V + v)) * R + r) * S + s] = w;
        }
    }
    __syncthreads();
    for (int c = tid; c < CH; c += 256) {
        float total = 0.f;
        for (int v = 0; v < V; ++v) {
            float acc = 0.f;
            const float* vp = value + (row0 + (size_t)v * S) * CH + c;
#pragma unroll 8
            for (int s = 0; s < S; ++s) acc += wts[v * S + s] * vp[(size_t)s * CH];
            if (zprev) acc += zprev[(size_t)lray * CH + c];
            total += acc;
        }
        zout[(size_t)lray * CH + c] = total;
    }
}

// ---------------------------------------------------------------------------------------------
// Folded variant: there is no non-linearity between query_encode_latent_2 and latent_value
// (CoPoNeRF.py:387-404) and the softmax weights of a ray sum to 1, so
//     sum_s w_s (Wv [W2 h_s0 + b2 ; W2 h_s1 + b2] + bv)  =  (Wv_a W2 | Wv_b W2) . sum_s w_s [h_s0 ; h_s1]  +  const.
// This kernel therefore reduces the 2*832 = 1664 hidden activations per sample (fp16, post-ReLU) with the
// softmax weights; the 1664 -> 416 value projection then runs once per RAY instead of once per sample.
// thread = 8 hidden channels (16 B), rows streamed; per ray T x 3328 B read, fully coalesced.
// ---------------------------------------------------------------------------------------------
constexpr int HC = 1664;
// 1 = stream hid with non-temporal loads (the product; keeps the node tables of encode_hidden in L2 / Infinity Cache);
// 0 = plain loads (tools/mall_probe.py: does a second pass over a cache-sized range run faster?)
#ifndef CPN_ATTEND_NT
#define CPN_ATTEND_NT 1
#endif
// rows of hid in flight per thread (16 bytes each): 4 is enough when the kernel has the chip to itself (7 waves per SIMD)
#ifndef CPN_ATTEND_UNROLL
#define CPN_ATTEND_UNROLL 4
#endif

// U = rows of hid in flight per thread; one workgroup per ray (U = 4: 8 waves per SIMD cover the latency).
template <bool HAVE_LOGITS, int U>
__device__ __forceinline__ void attend_hidden_ray(const unsigned lray, float* __restrict__ wts, float* __restrict__ red,
                                                  const __half* __restrict__ qa, const __half* __restrict__ qb,
                                                  const float* __restrict__ logits, const __half* __restrict__ hid, int V,
                                                  int R, int S, int ray0, __half* __restrict__ hbar,
                                                  float* __restrict__ at_wt) {
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = (size_t)lray * T;

    float lmax = -INFINITY;
    if constexpr (HAVE_LOGITS) {                          // row dot products already formed by the producing kernel
        for (int row = tid; row < T; row += 256) {
            const float logit = logits[row0 + row] / 11.31f;
            wts[row] = logit;
            lmax = fmaxf(lmax, logit);
        }
    } else
    // 16 lanes per row, 16 bytes each: a load instruction covers 4 whole 256-byte rows of each operand (two lanes per row with 8
    // loads of 16 bytes each, 128 bytes apart, touched 64 different lines per instruction: the training step's two launches of
    // this kernel spent 42 % of their cycles with the address path stalled by the cache - TA_ADDR_STALLED_BY_TC 289 M per launch)
    for (int base = 0; base < T; base += 16) {
        const int row = base + (tid >> 4);
        const int rr = row < T ? row : T - 1;
        const half8 a = *reinterpret_cast<const half8*>(qa + (row0 + rr) * 128 + (tid & 15) * 8);
        const half8 b = *reinterpret_cast<const half8*>(qb + (row0 + rr) * 128 + (tid & 15) * 8);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)b[e];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 8);
        if (row < T) {
            const float logit = acc / 11.31f;
            if ((tid & 15) == 0) wts[row] = logit;
            lmax = fmaxf(lmax, logit);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int row = tid; row < T; row += 256) {
        const float e = __expf(wts[row] - gmax);
        wts[row] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int row = tid; row < T; row += 256) {
        const float w = wts[row] * inv;
        wts[row] = w;
        if (at_wt) {
            const unsigned ray = (unsigned)ray0 + lray;
            const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
            const int v = row / S, s = row - v * S;
            at_wt[(((size_t)(b * V + v)) * R + r) * S + s] = w;
        }
    }
    __syncthreads();
    if (tid < HC / 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const __half* hp = hid + row0 * HC + tid * 8;
        auto ld = [&](int row) {
#if CPN_ATTEND_NT
            return __builtin_nontemporal_load(reinterpret_cast<const half8*>(hp + (size_t)row * HC));
#else
            return *reinterpret_cast<const half8*>(hp + (size_t)row * HC);
#endif
        };
        int row = 0;
        for (; row + U <= T; row += U) {                  // U loads in flight, then their U x 8 FMAs in row order
            half8 h[U];
#pragma unroll
            for (int u = 0; u < U; ++u) h[u] = ld(row + u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = wts[row + u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w * (float)h[u][e];
            }
        }
        for (; row < T; ++row) {
            const half8 h = ld(row);
            const float w = wts[row];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * (float)h[e];
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
        *reinterpret_cast<half8*>(hbar + (size_t)lray * HC + tid * 8) = o;
    }
}

template <bool HAVE_LOGITS>
__global__ __launch_bounds__(256) void attend_hidden_kernel(const __half* __restrict__ qa,
                                                            const __half* __restrict__ qb,
                                                            const float* __restrict__ logits,
                                                            const __half* __restrict__ hid, int V, int R, int S,
                                                            int ray0, __half* __restrict__ hbar,
                                                            float* __restrict__ at_wt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wts = reinterpret_cast<float*>(smem_raw);
    attend_hidden_ray<HAVE_LOGITS, CPN_ATTEND_UNROLL>(blockIdx.x, wts, wts + V * S, qa, qb, logits, hid, V, R, S, ray0, hbar, at_wt);
}

// ---------------------------------------------------------------------------------------------
// "Project before you store" variant (round 5, csrc/encode_fused.hip): the folded value projection already ran per SAMPLE
// inside the encoder - val (rows, 416) fp16 = (Wv_a W2 | Wv_b W2) . [h_own ; h_other] - so a round of the attention is
//     z[c] = sum_rows w[row] * val[row][c] + c'[c]  (+ V * zprev[c] in round 2: CoPoNeRF.py:481-485)
// (the softmax weights of a ray sum to 1, so the folded constant c' enters once).  Per ray T x 832 B read instead of T x 3 328.
// A wave takes every 4th row, lanes 0..51 one 16-byte piece (8 channels) each; the four partial sums meet in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int VC = 416;
__global__ __launch_bounds__(256) void attend_value_kernel(const float* __restrict__ logits, const __half* __restrict__ val,
                                                           const float* __restrict__ vbias, const float* __restrict__ zprev,
                                                           float zprev_scale, int V, int R, int S, int ray0,
                                                           float* __restrict__ zout, float* __restrict__ at_wt) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wts = reinterpret_cast<float*>(smem_raw);          // T weights
    float* red = wts + V * S;                                 // 8 floats of reduction scratch
    float* part = red + 8;                                    // 4 x 416 partial sums
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lray = blockIdx.x;
    const size_t row0 = (size_t)lray * T;

    float lmax = -INFINITY;
    for (int row = tid; row < T; row += 256) {
        const float logit = logits[row0 + row] / 11.31f;
        wts[row] = logit;
        lmax = fmaxf(lmax, logit);
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int row = tid; row < T; row += 256) {
        const float e = __expf(wts[row] - gmax);
        wts[row] = e;
        lsum += e;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) red[4 + wave] = lsum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
    for (int row = tid; row < T; row += 256) {
        const float w = wts[row] * inv;
        wts[row] = w;
        if (at_wt) {
            const unsigned ray = (unsigned)ray0 + lray;
            const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
            const int v = row / S, s = row - v * S;
            at_wt[(((size_t)(b * V + v)) * R + r) * S + s] = w;
        }
    }
    __syncthreads();
    if (lane < VC / 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const __half* vp = val + row0 * VC + lane * 8;
        constexpr int UR = 8;                                  // rows in flight per lane
        int row = wave;
        for (; row + 4 * (UR - 1) < T; row += 4 * UR) {
            half8 h[UR];
#pragma unroll
            for (int u = 0; u < UR; ++u) h[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vp + (size_t)(row + 4 * u) * VC));
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const float w = wts[row + 4 * u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += w * (float)h[u][e];
            }
        }
        for (; row < T; row += 4) {
            const half8 h = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vp + (size_t)row * VC));
            const float w = wts[row];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * (float)h[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave * VC + lane * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int c = tid; c < VC; c += 256) {
        float z = ((part[c] + part[VC + c]) + (part[2 * VC + c] + part[3 * VC + c])) + vbias[c];
        if (zprev) z += zprev_scale * zprev[(size_t)lray * VC + c];
        zout[(size_t)lray * VC + c] = z;
    }
}

}  // namespace

extern "C" int cpn_attend_value(const float* logits, const uint16_t* val, const float* vbias, const float* zprev,
                                float zprev_scale, int B, int V, int R, int S, int ray0, int nrays, float* zout, float* at_wt,
                                void* stream) {
    CPN_REQUIRE(logits && val && vbias && zout, CPN_E_ARG, "cpn_attend_value: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 4096, CPN_E_SHAPE, "cpn_attend_value: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_value: ray range outside B*R");
    CPN_REQUIRE(((uintptr_t)val % 16) == 0, CPN_E_ARG, "cpn_attend_value: val must be 16-byte aligned");
    const size_t lds = (size_t)(V * S + 8 + 4 * VC) * sizeof(float);
    hipLaunchKernelGGL(attend_value_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream, logits, (const __half*)val, vbias,
                       zprev, zprev_scale, V, R, S, ray0, zout, at_wt);
    CPN_LAUNCH_CHECK("cpn_attend_value");
    return 0;
}

extern "C" int cpn_attend(const uint16_t* qa, const uint16_t* qb, const float* value, const float* zprev,
                          int B, int V, int R, int S, int ray0, int nrays, float* zout,
                          float* at_wt, void* stream) {
    CPN_REQUIRE(qa && qb && value && zout, CPN_E_ARG, "cpn_attend: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 4096, CPN_E_SHAPE, "cpn_attend: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend: ray range outside B*R");
    const size_t lds = (size_t)(V * S + 8) * sizeof(float);
    hipLaunchKernelGGL(attend_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream, (const __half*)qa,
                       (const __half*)qb, value, zprev, V, R, S, ray0, zout, at_wt);
    CPN_LAUNCH_CHECK("cpn_attend");
    return 0;
}

extern "C" int cpn_attend_f32(const float* qa, const float* qb, const float* value, const float* zprev, int B, int V, int R,
                              int S, int ray0, int nrays, float* zout, float* at_wt, void* stream) {
    CPN_REQUIRE(qa && qb && value && zout, CPN_E_ARG, "cpn_attend_f32: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 4096, CPN_E_SHAPE, "cpn_attend_f32: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_f32: ray range outside B*R");
    CPN_REQUIRE(((uintptr_t)qa % 16) == 0 && ((uintptr_t)qb % 16) == 0, CPN_E_ARG, "cpn_attend_f32: operands must be 16-byte aligned");
    const size_t lds = (size_t)(V * S + 8) * sizeof(float);
    hipLaunchKernelGGL(attend_f32_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream, qa, qb, value, zprev, V, R, S, ray0,
                       zout, at_wt);
    CPN_LAUNCH_CHECK("cpn_attend_f32");
    return 0;
}

extern "C" int cpn_attend_hidden(const uint16_t* qa, const uint16_t* qb, const float* logits, const uint16_t* hid, int B,
                                 int V, int R, int S, int ray0, int nrays, uint16_t* hbar, float* at_wt, void* stream) {
    CPN_REQUIRE(((qa && qb) || logits) && hid && hbar, CPN_E_ARG, "cpn_attend_hidden: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 4096, CPN_E_SHAPE, "cpn_attend_hidden: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_hidden: ray range outside B*R");
    const size_t lds = (size_t)(V * S + 8) * sizeof(float);
    hipLaunchKernelGGL(logits ? attend_hidden_kernel<true> : attend_hidden_kernel<false>, dim3(nrays), dim3(256), lds, (hipStream_t)stream, (const __half*)qa,
                       (const __half*)qb, logits, (const __half*)hid, V, R, S, ray0, (__half*)hbar, at_wt);
    CPN_LAUNCH_CHECK("cpn_attend_hidden");
    return 0;
}
