// K4 — joint softmax over the V*S epipolar samples of one query ray + attention-weighted value sum.
//
// Replaces einsum('bijk,bijk->bjk') / 11.31, F.softmax over (n_context * npoints), the broadcast
// multiply-sum and the sum over views (/root/reference models/CoPoNeRF.py:450-461 and 475-485).
// The reference has NO alpha compositing: the along-ray reduction is (max, sum-exp, weighted sum), which
// maps onto wave reductions — with S = 64 each view's sample axis is exactly one 64-lane wavefront.
//
// One 256-thread workgroup per query ray: T = V*S rows; the sum runs on the HIDDEN activations (folded form below).
// (The layer-by-layer form on the 416-wide values - cpn_attend, its fp32 sibling and the "project before you store" form
// cpn_attend_value - left the library in round 6: tools/experiments/r6_pruned/attend.hip.)
#include <algorithm>

#include "common.h"

namespace {


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

}  // namespace

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
