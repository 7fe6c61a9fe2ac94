// K6-K8 — the 4-D operators of UFC (get_z path).
//
//   cpn_conv4d_gn_relu   conv4d.Conv4d (+ MaxPool4d for stride > 1) + GroupNorm(1 group) + ReLU
//                        (/root/reference models/conv4d.py:7-30, 57-163): 63 calls per get_z
//   cpn_correlation      aggregation.correlation / correlation_token (models/aggregation.py:70-80)
//   cpn_soft_argmax_pair aggregation.soft_argmax in both directions (models/aggregation.py:119-144, 555-560)
//
// The reference spends 44 % of get_z in einops `rearrange` copies of 6-D tensors around stock Conv2d / MaxPool2d
// calls; here both separable branches, the pooling, both biases and the GroupNorm statistics are one pass over
// the volume with direct 6-D indexing — no transposed copies.  All of it is HBM/L2-bound streaming work
// (volumes 16^4 x {8,32} ch = 2-8 MB, raw correlations up to 64^4 = 67 MB): coalescing along the innermost
// support axis, one scalar-broadcast weight per (out-channel, in-channel, tap).
#include <algorithm>

#include <cstdlib>

#include "common.h"

namespace {

// GroupNorm statistics of sample b, DETERMINISTIC (round 3; the 32-slot atomicAdd scheme of round 2 summed the workgroup
// partials in whatever order they arrived: two runs of get_z differed in the last bits).  stats layout (doubles):
//   [2b], [2b+1]        sum / sum of squares of sample b, written once by the LAST workgroup of the sample to finish
//   [2B + b]            arrival counter (zero on entry), bits of an unsigned long long
//   [3B + 2(b W + w)..] partial pair of workgroup w of sample b, W = gridDim.x * gridDim.y
// Every workgroup publishes its pair with returning exchanges (performed at the device-coherent level, like the atomic
// adds they replace; the returned values are consumed, so they have completed before the counter is bumped), the
// workgroup that draws the last ticket re-reads all W pairs with device-scope loads and sums them in a FIXED order
// (thread t takes w = t, t + 256, ...; then a fixed LDS tree).  Must be called by all 256 threads of the workgroup.
__device__ __forceinline__ void gn_publish(double s1, double s2, double* __restrict__ stats, int b) {
    __shared__ int last_flag;
    __shared__ double tree[2][256];
    const unsigned W = gridDim.x * gridDim.y, me = blockIdx.x + gridDim.x * blockIdx.y, B = gridDim.z;
    unsigned long long* part = reinterpret_cast<unsigned long long*>(stats + 3 * (size_t)B + 2 * ((size_t)b * W));
    if (threadIdx.x == 0) {
        const unsigned long long o1 = __hip_atomic_exchange(part + 2 * me, (unsigned long long)__double_as_longlong(s1),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o2 = __hip_atomic_exchange(part + 2 * me + 1, (unsigned long long)__double_as_longlong(s2),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned zero;                                        // 0, but only known once both exchanges have returned
        asm volatile("v_and_b32 %0, 0, %1\n\tv_and_b32 %0, %0, %2" : "=&v"(zero) : "v"((unsigned)o1), "v"((unsigned)o2));
        unsigned long long* cnt = reinterpret_cast<unsigned long long*>(stats + 2 * (size_t)B + b);
        const unsigned long long ticket = __hip_atomic_fetch_add(cnt, 1ULL + zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = ticket == (unsigned long long)W - 1;
    }
    __syncthreads();
    if (!last_flag) return;
    double a1 = 0.0, a2 = 0.0;
    for (unsigned w = threadIdx.x; w < W; w += 256) {
        a1 += __longlong_as_double((long long)__hip_atomic_load(part + 2 * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        a2 += __longlong_as_double((long long)__hip_atomic_load(part + 2 * w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    tree[0][threadIdx.x] = a1;
    tree[1][threadIdx.x] = a2;
    __syncthreads();
    for (int n = 128; n > 0; n >>= 1) {
        if ((int)threadIdx.x < n) {
            tree[0][threadIdx.x] += tree[0][threadIdx.x + n];
            tree[1][threadIdx.x] += tree[1][threadIdx.x + n];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[2 * (size_t)b] = tree[0][0];
        stats[2 * (size_t)b + 1] = tree[1][0];
    }
}

// ------------------------------------------------------------------------------------------------
// conv4d: y[b,o,qy,qx,sy,sx] = bq[o] + bs[o]
//        + sum_{c,i,j} Wq[o,c,i,j] * Ps(x)[b,c, qy*s+i-p, qx*s+j-p, sy, sx]      (query branch, support dims pooled)
//        + sum_{c,i,j} Ws[o,c,i,j] * Pq(x)[b,c, qy, qx, sy*s+i-p, sx*s+j-p]      (support branch, query dims pooled)
// thread = one output position, blockIdx.y = output channel (uniform -> scalar weight loads), blockIdx.z = batch
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv4d_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                     const float* __restrict__ bq, const float* __restrict__ ws,
                                                     const float* __restrict__ bs, int Cin, int Hq, int Wq, int Hs,
                                                     int Ws, int k, int s, int p, int Oq, int Pq_, int Os, int Ps_,
                                                     float* __restrict__ y, double* __restrict__ stats) {
    // Oq x Pq_ = output query dims, Os x Ps_ = output support dims
    const int o = blockIdx.y, b = blockIdx.z, Cout = gridDim.y;
    const long long npos = (long long)Oq * Pq_ * Os * Ps_;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float val = 0.0f;
    const bool active = pos < npos;
    if (active) {
        int sx = (int)(pos % Ps_);
        long long t = pos / Ps_;
        const int sy = (int)(t % Os); t /= Os;
        const int qx = (int)(t % Pq_);
        const int qy = (int)(t / Pq_);
        const size_t cstride = (size_t)Hq * Wq * Hs * Ws;
        const float* xb = x + (size_t)b * Cin * cstride;
        float acc = bq[o] + bs[o];
        for (int c = 0; c < Cin; ++c) {
            const float* xc = xb + c * cstride;
            const float* wqc = wq + ((size_t)o * Cin + c) * k * k;
            const float* wsc = ws + ((size_t)o * Cin + c) * k * k;
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    // ---- query branch: conv over (Hq, Wq), support dims max-pooled by s (ceil mode)
                    const int Y = qy * s + i - p, X = qx * s + j - p;
                    if (Y >= 0 && Y < Hq && X >= 0 && X < Wq) {
                        const float* base = xc + ((size_t)Y * Wq + X) * Hs * Ws;
                        float m = -INFINITY;
                        for (int dy = 0; dy < s; ++dy)
                            for (int dx = 0; dx < s; ++dx) {
                                const int yy = sy * s + dy, xx = sx * s + dx;
                                if (yy < Hs && xx < Ws) m = fmaxf(m, base[(size_t)yy * Ws + xx]);
                            }
                        acc += wqc[i * k + j] * m;
                    }
                    // ---- support branch: conv over (Hs, Ws), query dims max-pooled by s
                    const int U = sy * s + i - p, Vv = sx * s + j - p;
                    if (U >= 0 && U < Hs && Vv >= 0 && Vv < Ws) {
                        float m = -INFINITY;
                        for (int dy = 0; dy < s; ++dy)
                            for (int dx = 0; dx < s; ++dx) {
                                const int yy = qy * s + dy, xx = qx * s + dx;
                                if (yy < Hq && xx < Wq) m = fmaxf(m, xc[(((size_t)yy * Wq + xx) * Hs + U) * Ws + Vv]);
                            }
                        acc += wsc[i * k + j] * m;
                    }
                }
        }
        val = acc;
        y[((size_t)b * Cout + o) * npos + pos] = acc;
    }
    // GroupNorm statistics of the whole (C, volume) slab of sample b, accumulated in double
    double s1 = active ? (double)val : 0.0, s2 = active ? (double)val * val : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    gn_publish(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], stats, b);
}

// Strided layers (k3 s2 on 32^4, k5 s4 on 64^4): the kernel above re-evaluates the s x s max-pool window for every tap
// (k*k*s*s reads per output, channel and branch).  With a scratch buffer the two pooled volumes are formed once
//   Ps[b,c,Y,X,sy',sx'] = max_{dy,dx<s} x[b,c,Y,X,sy'*s+dy,sx'*s+dx]     (query branch: support dims pooled)
//   Pq[b,c,qy',qx',U,V] = max_{dy,dx<s} x[b,c,qy'*s+dy,qx'*s+dx,U,V]     (support branch: query dims pooled)
// and the convolution reads k*k values per output, channel and branch.
__global__ __launch_bounds__(256) void pool_support_kernel(const float* __restrict__ x, int Hs, int Ws, int s, int Os,
                                                           int Ps_, long long planes, float* __restrict__ out) {
    // planes = B*Cin*Hq*Wq, each an (Hs,Ws) -> (Os,Ps_) max-pool with ceil mode
    const long long total = planes * Os * Ps_;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int sx = (int)(i % Ps_);
        long long t = i / Ps_;
        const int sy = (int)(t % Os);
        const long long pl = t / Os;
        const float* base = x + pl * Hs * Ws;
        float m = -INFINITY;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) {
                const int yy = sy * s + dy, xx = sx * s + dx;
                if (yy < Hs && xx < Ws) m = fmaxf(m, base[(size_t)yy * Ws + xx]);
            }
        out[i] = m;
    }
}

__global__ __launch_bounds__(256) void pool_query_kernel(const float* __restrict__ x, int Hq, int Wq, int Hs, int Ws,
                                                         int s, int Oq, int Pq_, long long bc, float* __restrict__ out) {
    // out[bc, qy', qx', U, V]; consecutive threads walk (U,V): contiguous reads of s*s planes
    const long long P = (long long)Hs * Ws, total = bc * Oq * Pq_ * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long uv = i % P;
        long long t = i / P;
        const int qx = (int)(t % Pq_); t /= Pq_;
        const int qy = (int)(t % Oq);
        const long long c = t / Oq;
        const float* base = x + c * Hq * Wq * P + uv;
        float m = -INFINITY;
        for (int dy = 0; dy < s; ++dy)
            for (int dx = 0; dx < s; ++dx) {
                const int yy = qy * s + dy, xx = qx * s + dx;
                if (yy < Hq && xx < Wq) m = fmaxf(m, base[((size_t)yy * Wq + xx) * P]);
            }
        out[i] = m;
    }
}

// same contract as conv4d_kernel, reading the pooled volumes
__global__ __launch_bounds__(256) void conv4d_pooled_kernel(const float* __restrict__ ps, const float* __restrict__ pq,
                                                            const float* __restrict__ wq, const float* __restrict__ bq,
                                                            const float* __restrict__ ws, const float* __restrict__ bs,
                                                            int Cin, int Hq, int Wq, int Hs, int Ws, int k, int s, int p,
                                                            int Oq, int Pq_, int Os, int Ps_, float* __restrict__ y,
                                                            double* __restrict__ stats) {
    const int o = blockIdx.y, b = blockIdx.z, Cout = gridDim.y;
    const long long npos = (long long)Oq * Pq_ * Os * Ps_;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float val = 0.0f;
    const bool active = pos < npos;
    if (active) {
        int sx = (int)(pos % Ps_);
        long long t = pos / Ps_;
        const int sy = (int)(t % Os); t /= Os;
        const int qx = (int)(t % Pq_);
        const int qy = (int)(t / Pq_);
        const size_t ps_c = (size_t)Hq * Wq * Os * Ps_, pq_c = (size_t)Oq * Pq_ * Hs * Ws;
        const float* psb = ps + (size_t)b * Cin * ps_c;
        const float* pqb = pq + (size_t)b * Cin * pq_c;
        float acc = bq[o] + bs[o];
        for (int c = 0; c < Cin; ++c) {
            const float* wqc = wq + ((size_t)o * Cin + c) * k * k;
            const float* wsc = ws + ((size_t)o * Cin + c) * k * k;
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    const int Y = qy * s + i - p, X = qx * s + j - p;
                    if (Y >= 0 && Y < Hq && X >= 0 && X < Wq)
                        acc += wqc[i * k + j] * psb[c * ps_c + (((size_t)Y * Wq + X) * Os + sy) * Ps_ + sx];
                    const int U = sy * s + i - p, Vv = sx * s + j - p;
                    if (U >= 0 && U < Hs && Vv >= 0 && Vv < Ws)
                        acc += wsc[i * k + j] * pqb[c * pq_c + (((size_t)qy * Pq_ + qx) * Hs + U) * Ws + Vv];
                }
        }
        val = acc;
        y[((size_t)b * Cout + o) * npos + pos] = acc;
    }
    double s1 = active ? (double)val : 0.0, s2 = active ? (double)val * val : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    gn_publish(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], stats, b);
}

// conv4d_pooled_kernel with ALL eight output channels in one thread (the strided layers of UFC are 1 -> 8 channels): the
// 2 k^2 Cin taps of a position are read once instead of once per output channel (8 workgroups per position tile re-read
// them: 52 M loads per 64^4 -> 16^4 layer, 39 us for 19 MFLOP); filters in LDS as [c][tap][branch][8].  Same tap order per
// output as conv4d_pooled_kernel (bias first, then c, i, j with the query tap before the support tap).
__global__ __launch_bounds__(256) void conv4d_pooled_c8_kernel(const float* __restrict__ ps, const float* __restrict__ pq,
                                                               const float* __restrict__ wq, const float* __restrict__ bq,
                                                               const float* __restrict__ ws, const float* __restrict__ bs,
                                                               int Cin, int Hq, int Wq, int Hs, int Ws, int k, int s, int p,
                                                               int Oq, int Pq_, int Os, int Ps_, float* __restrict__ y,
                                                               double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float wl[];        // [Cin][k*k][2][8]
    const int b = blockIdx.z, kk = k * k;
    for (int i = threadIdx.x; i < Cin * kk * 16; i += 256) {
        const int o = i & 7, br = (i >> 3) & 1, t = i >> 4;              // t = c * kk + tap
        wl[i] = (br ? ws : wq)[(size_t)o * Cin * kk + t];
    }
    __syncthreads();
    const long long npos = (long long)Oq * Pq_ * Os * Ps_;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = pos < npos;
    double s1 = 0.0, s2 = 0.0;
    if (active) {
        const int sx = (int)(pos % Ps_);
        long long t = pos / Ps_;
        const int sy = (int)(t % Os); t /= Os;
        const int qx = (int)(t % Pq_);
        const int qy = (int)(t / Pq_);
        const size_t ps_c = (size_t)Hq * Wq * Os * Ps_, pq_c = (size_t)Oq * Pq_ * Hs * Ws;
        const float* psb = ps + (size_t)b * Cin * ps_c;
        const float* pqb = pq + (size_t)b * Cin * pq_c;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = bq[o] + bs[o];
        for (int c = 0; c < Cin; ++c)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    const int Y = qy * s + i - p, X = qx * s + j - p;
                    const int U = sy * s + i - p, Vv = sx * s + j - p;
                    const bool okq = Y >= 0 && Y < Hq && X >= 0 && X < Wq, oks = U >= 0 && U < Hs && Vv >= 0 && Vv < Ws;
                    const float vq = okq ? psb[c * ps_c + (((size_t)Y * Wq + X) * Os + sy) * Ps_ + sx] : 0.0f;
                    const float vs = oks ? pqb[c * pq_c + (((size_t)qy * Pq_ + qx) * Hs + U) * Ws + Vv] : 0.0f;
                    const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + (size_t)(c * kk + i * k + j) * 16);
                    const f32x4 a0 = w4[0], a1 = w4[1], b0 = w4[2], b1 = w4[3];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // the per-channel kernel skips an out-of-range tap; adding w * 0 gives the same value
                        if (okq) { acc[e] += a0[e] * vq; acc[4 + e] += a1[e] * vq; }
                        if (oks) { acc[e] += b0[e] * vs; acc[4 + e] += b1[e] * vs; }
                    }
                }
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            y[((size_t)b * 8 + o) * npos + pos] = acc[o];
            s1 += (double)acc[o];
            s2 += (double)acc[o] * acc[o];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    gn_publish(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], stats, b);
}

// stride-1 3x3x3x3 fast path (57 of the 63 Conv4d calls of a get_z): one thread computes COUT output channels of
// its position (blockIdx.y = channel group), so every input value is read once per tap instead of once per (tap,
// output channel); the weights are staged in LDS as [cin][tap][branch][cout] and read as wave-uniform (broadcast)
// 16-byte vectors.  COUT = 8 (4 at B = 1 with 8 channels, where a 16^4 volume is 1 024 waves = ONE per SIMD): more
// channels per thread would read the inputs fewer times but do not fit the register budget of 4 waves per SIMD.
// DGRAD: the same kernel as the data gradient of the layer — input = dy, weights read IN PLACE as the transposed, flipped
// filters (w'[o][c][tap] = w[c][o][8 - tap], the forward layer's (cout_fwd = Cin here, cin_fwd = cout_total, 3, 3)
// tensors), no bias, no statistics.  The autograd wrapper used to build flipped copies, a zero bias and a statistics
// buffer per layer: ~450 small launches per training step.
template <int COUT, bool PF, bool DGRAD = false>
__global__ __launch_bounds__(256, PF ? 2 : 4) void conv4d_k3s1_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                          const float* __restrict__ bq, const float* __restrict__ ws,
                                                          const float* __restrict__ bs, int Cin, int Hq, int Wq,
                                                          int Hs, int Ws, int cout_total, float* __restrict__ y,
                                                          double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float wl[];        // Cin * 9 * 2 * COUT floats
    const int b = blockIdx.z;
    const int o0 = blockIdx.y * COUT;
    for (int i = threadIdx.x; i < Cin * 9 * 2 * COUT; i += 256) {
        const int o = o0 + i % COUT;
        int t = i / COUT;
        const int br = t & 1; t >>= 1;
        const int tap = t % 9, c = t / 9;
        wl[i] = DGRAD ? (br ? ws : wq)[((size_t)c * cout_total + o) * 9 + (8 - tap)]
                      : (br ? ws : wq)[((size_t)o * Cin + c) * 9 + tap];
    }
    __syncthreads();
    const long long npos = (long long)Hq * Wq * Hs * Ws;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = pos < npos;
    double s1 = 0.0, s2 = 0.0;
    if (active) {
        const int sx = (int)(pos % Ws);
        long long t = pos / Ws;
        const int sy = (int)(t % Hs); t /= Hs;
        const int qx = (int)(t % Wq);
        const int qy = (int)(t / Wq);
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = DGRAD ? 0.0f : bq[o0 + o] + bs[o0 + o];
        // The 18 taps of this position as 32-bit byte offsets into the batch element's (Cin, npos) block, read with
        // buffer loads (scalar base + per-lane offset; an out-of-range offset returns 0 = the zero padding).  Registers
        // are what limits this kernel: with 64-bit per-lane addresses it needed 512 VGPRs = one wave per SIMD.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(x + (size_t)b * Cin * (size_t)npos), 0, (int)((size_t)Cin * npos * 4), 0x00020000);
        int off[18];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int t = i * 3 + j;
                const int Y = qy + i - 1, X = qx + j - 1, U = sy + i - 1, Vv = sx + j - 1;
                const bool okq = Y >= 0 && Y < Hq && X >= 0 && X < Wq, oks = U >= 0 && U < Hs && Vv >= 0 && Vv < Ws;
                off[2 * t] = okq ? (((Y * Wq + X) * Hs + sy) * Ws + sx) * 4 : 0x7ffffff0;
                off[2 * t + 1] = oks ? (((qy * Wq + qx) * Hs + U) * Ws + Vv) * 4 : 0x7ffffff0;
            }
        // the 18 taps of channel c+1 are requested before the FMAs of channel c (a channel is ~1 us of load latency and
        // a few hundred cycles of math: at B = 1 nothing else hides it)
        auto load_taps = [&](int c, float (&xv)[18]) {
            const int cbase = c * (int)npos * 4;              // scalar offset of the channel
#pragma unroll
            for (int t = 0; t < 18; ++t)
                xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off[t], cbase, 0));
        };
        auto accumulate = [&](int c, const float (&xv)[18]) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4* w4 = reinterpret_cast<const f32x4*>(wl + ((c * 9 + t) * 2) * COUT);
                const float vq = xv[2 * t], vs = xv[2 * t + 1];
#pragma unroll
                for (int o4 = 0; o4 < COUT / 4; ++o4) {
                    const f32x4 a = w4[o4], bb = w4[COUT / 4 + o4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)              // two chained FMAs per output (v_pk_fma_f32 pairs)
                        acc[o4 * 4 + e] = __builtin_fmaf(bb[e], vs, __builtin_fmaf(a[e], vq, acc[o4 * 4 + e]));
                    if ((o4 & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // <= 4 weight vectors in flight
                }
            }
        };
        if (PF) {                                             // small launches (B = 1): latency, not occupancy, is the limit
            float xa[18], xb2[18];
            load_taps(0, xa);
            int c = 0;
#pragma unroll 1
            for (; c + 2 <= Cin; c += 2) {
                load_taps(c + 1, xb2);
                accumulate(c, xa);
                if (c + 2 < Cin) load_taps(c + 2, xa);
                accumulate(c + 1, xb2);
            }
            if (c < Cin) accumulate(c, xa);
        } else {                                              // large launches: 4 waves per SIMD hide the loads
#pragma unroll 1
            for (int c = 0; c < Cin; ++c) {
                float xv[18];
                load_taps(c, xv);
                accumulate(c, xv);
            }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            y[((size_t)b * cout_total + o0 + o) * npos + pos] = acc[o];
            s1 += (double)acc[o];
            s2 += (double)acc[o] * acc[o];
        }
    }
    if (DGRAD) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    gn_publish(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], stats, b);
}

// mean / 1/std of sample b from its (sum, sum of squares) pair (gn_publish), computed ONCE per workgroup and broadcast
// through LDS.
// Must be called by all threads of the block (contains a barrier).
__device__ __forceinline__ void gn_block_stats(const double* __restrict__ stats, int b, double n, float eps, float& mean,
                                               float& rstd) {
    __shared__ float mr[2];
    if (threadIdx.x == 0) {
        const double m = stats[2 * (size_t)b] / n;
        const double var = stats[2 * (size_t)b + 1] / n - m * m;
        mr[0] = (float)m;
        mr[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    mean = mr[0];
    rstd = mr[1];
}

__global__ __launch_bounds__(256) void gn_relu_kernel(const float* y, const double* __restrict__ stats,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ res, float eps, int Cout, long long npos,
                                                      float* out) {
    const int b = blockIdx.z, o = blockIdx.y;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float mean, rstd;
    gn_block_stats(stats, b, (double)Cout * (double)npos, eps, mean, rstd);
    if (pos >= npos) return;
    const size_t idx = ((size_t)b * Cout + o) * npos + pos;
    const float v = (y[idx] - mean) * rstd * gamma[o] + beta[o];
    // res: the residual the caller adds to the block's output (x + Encoder4D(x), aggregation.py:306,347-355) in the same pass
    out[idx] = res ? res[idx] + fmaxf(v, 0.0f) : fmaxf(v, 0.0f);
}

// ---- the stride-1 3x3x3x3 layer on the fp32 MFMA (round 4; the round-3 form fed every MFMA with its own 4-byte load) -------
// out[co, pos] = bias + sum over (branch, tap, ci) of w_branch[co, ci, tap] * x[ci, pos + tap_branch]: an implicit GEMM with
// M = output channels, N = positions, K = 2 * 9 * Cin on v_mfma_f32_16x16x4_f32 (an fmaf chain: fp32-exact products).
//   wave      64 consecutive positions x ALL output channels (MT tiles of 16, padded: Cout = 8 uses half a tile)
//   k step    four consecutive input channels of one (branch, tap): lane (j, g) holds a 16-byte vector of channel ci0 + g at
//             positions p0 + 4 j .. + 3, and ELEMENT e of that vector is the B operand of the output tile of positions
//             {p0 + 4 j + e}: one vector feeds 4 MT MFMAs
//   loads     per 4-channel group 9 aligned vectors for the query-pair taps (whole vectors in or out of the volume: the
//             buffer load's out-of-range zero is the padding) and 3 for the support-pair ROWS; the dx = -1 / +1 taps of a
//             row are the same vector shifted by one element, the element that crosses a lane coming from the neighbour
//             lane through a DPP row shift (zero at the ends of a row of Ws): 12 loads feed 72 MT MFMAs, and the loads of
//             the next group are in flight under them
//   A operand w[co][ci0 + g] from LDS ([branch][tap][ci][co], conflict-free 4-byte reads)
// The VALU version above spends 1 152 FMAs per position and 8 channels and re-reads every input once per 8 output channels;
// here the multiplies sit on the matrix pipe.  Needs Cin % 4 == 0, Ws in {4, 8, 16, 32, 64}, positions % 64 == 0.  DGRAD as above.
template <int MT, bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv4d_k3s1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                                  const float* __restrict__ bq, const float* __restrict__ ws,
                                                                  const float* __restrict__ bs, int Cin, int Hq, int Wq, int Hs,
                                                                  int Ws, int cout_total, float* __restrict__ y,
                                                                  double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float wl[];        // [2][9][Cin][MT*16]
    const int b = blockIdx.z;
    constexpr int CO = MT * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, g = lane >> 4;
    const long long npos = (long long)Hq * Wq * Hs * Ws;
    const long long pos = ((long long)blockIdx.x * 4 + wave) * 64 + 4 * fi;      // first of this lane's four positions
    const bool active = pos < npos;                                              // whole waves: npos % 64 == 0
    const long long pc = active ? pos : 0;
    const int sx = (int)(pc % Ws);
    long long t = pc / Ws;
    const int sy = (int)(t % Hs); t /= Hs;
    const int qx = (int)(t % Wq);
    const int qy = (int)(t / Wq);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (size_t)b * Cin * (size_t)npos), 0, (int)((size_t)Cin * npos * 4), 0x00020000);
    const int plane = (int)npos * 4;                               // bytes of one input channel
    constexpr int kOOB = 0x7ffffff0;
    int offq[9], offs[3];                                          // with this lane's channel offset g
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int Y = qy + i - 1, X = qx + j - 1;
            const bool ok = active && Y >= 0 && Y < Hq && X >= 0 && X < Wq;
            offq[i * 3 + j] = ok ? (((Y * Wq + X) * Hs + sy) * Ws + sx) * 4 + g * plane : kOOB;
        }
        const int U = sy + i - 1;
        offs[i] = (active && U >= 0 && U < Hs) ? (((qy * Wq + qx) * Hs + U) * Ws + sx) * 4 + g * plane : kOOB;
    }
    const bool row_first = sx == 0, row_last = sx + 4 == Ws;
    f32x4 acc[MT][4];                                              // [channel tile][position subset e]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = mt * 16 + g * 4 + i;
            const float bias = (DGRAD || co >= cout_total) ? 0.0f : bq[co] + bs[co];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][e][i] = bias;
        }
    auto load_group = [&](int c0, f32x4 (&vq)[9], f32x4 (&vs)[3]) {
        const int so = c0 * plane;
#pragma unroll
        for (int k = 0; k < 9; ++k) vq[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offq[k], so, 0));
#pragma unroll
        for (int k = 0; k < 3; ++k) vs[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, offs[k], so, 0));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](int c0, int k, const f32x4& v) {                // k = branch * 9 + tap
        const float* wrow = wl + ((size_t)k * Cin + c0 + g) * CO + fi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float a = wrow[mt * 16];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mt][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[e], acc[mt][e], 0, 0, 0);
        }
    };
    auto compute_group = [&](int c0, const f32x4 (&vq)[9], const f32x4 (&vs)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const f32x4 r = vs[i];
            // neighbours' edge elements: lane j - 1's element 3 and lane j + 1's element 0 (DPP row shifts inside the 16 lanes
            // of a channel group; a row of Ws never crosses them because Ws divides 64 or is a multiple of it)
            // (written as asm: with the update_dpp builtin the compiler read element 0 of the vector for BOTH shifts here)
            float left, right;
            const float r3 = r[3], r0 = r[0];
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_mov_b32_dpp %1, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(left), "=&v"(right) : "v"(r3), "v"(r0));
            const f32x4 rm = {row_first ? 0.0f : left, r[0], r[1], r[2]};       // x - 1
            const f32x4 rp = {r[1], r[2], r[3], row_last ? 0.0f : right};       // x + 1
            step(c0, i * 3 + 0, vq[i * 3 + 0]);
            step(c0, 9 + i * 3 + 0, rm);
            step(c0, i * 3 + 1, vq[i * 3 + 1]);
            step(c0, 9 + i * 3 + 1, r);
            step(c0, i * 3 + 2, vq[i * 3 + 2]);
            step(c0, 9 + i * 3 + 2, rp);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 q0[9], s0[3], q1[9], s1v[3];
    load_group(0, q0, s0);                                         // in flight while the filters are staged
    // Filters into LDS: the GLOBAL index runs with the thread index (coalesced reads, four in flight per thread), the LDS
    // index is scattered.  (Walking the LDS index instead made every thread issue up to 36 dependent 4-byte reads with a
    // stride of 9 floats: a third of the kernel's time at 32 input channels.)
    if (cout_total < CO)
        for (int i = threadIdx.x; i < 18 * Cin * CO; i += 256)
            if (i % CO >= cout_total) wl[i] = 0.0f;
    {
        const int per_branch = cout_total * Cin * 9, total = 2 * per_branch;
        for (int g0 = threadIdx.x; g0 < total; g0 += 1024) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gi = g0 + 256 * u;
                v[u] = gi < total ? (gi < per_branch ? wq[gi] : ws[gi - per_branch]) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gi = g0 + 256 * u;
                if (gi >= total) continue;
                const int br = gi >= per_branch, q = gi - br * per_branch;
                const int tapg = q % 9, mid = (q / 9) % (DGRAD ? cout_total : Cin), top = q / (9 * (DGRAD ? cout_total : Cin));
                // forward tensor [o][c][tap]; data gradient reads the forward layer's [c][o][8 - tap]
                const int o = DGRAD ? mid : top, c = DGRAD ? top : mid, tap = DGRAD ? 8 - tapg : tapg;
                wl[((size_t)(br * 9 + tap) * Cin + c) * CO + o] = v[u];
            }
        }
    }
    __syncthreads();
    for (int c0 = 0; c0 < Cin; c0 += 8) {
        if (c0 + 4 < Cin) load_group(c0 + 4, q1, s1v);
        compute_group(c0, q0, s0);
        if (c0 + 4 < Cin) {
            if (c0 + 8 < Cin) load_group(c0 + 8, q0, s0);
            compute_group(c0 + 4, q1, s1v);
        }
    }
    double s1 = 0.0, s2 = 0.0;
    if (active) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = mt * 16 + g * 4 + i;
                if (co < cout_total) {
                    const f32x4 v = {acc[mt][0][i], acc[mt][1][i], acc[mt][2][i], acc[mt][3][i]};
                    *reinterpret_cast<f32x4*>(y + ((size_t)b * cout_total + co) * npos + pos) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 += (double)v[e];
                        s2 += (double)v[e] * v[e];
                    }
                }
            }
    }
    if (DGRAD) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ double red[8];
    if (lane == 0) { red[wave * 2] = s1; red[wave * 2 + 1] = s2; }
    __syncthreads();
    gn_publish(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7], stats, b);
}

// ---- backward of GroupNorm(1 group) + ReLU, two passes over the volume ---------------------------------------
//   dz = dout * [out > 0],  yh = (y - mean_b) * rstd_b
//   pass 1: S1_b = sum dz*g_c, S2_b = sum dz*g_c*yh (over channels and positions);  dgamma_c = sum dz*yh, dbeta_c = sum dz
//   pass 2: dy = rstd_b * (dz*g_c - S1_b/n - yh*S2_b/n)
// red: (B*2 + C*2) doubles, zero on entry.
__global__ __launch_bounds__(256) void gn_relu_bwd_reduce_kernel(
    const float* __restrict__ y, const float* __restrict__ out, const float* __restrict__ dout,
    const double* __restrict__ stats, const float* __restrict__ gamma, float eps, int B, int Cout, long long npos,
    double* __restrict__ red) {
    const int b = blockIdx.z, o = blockIdx.y;
    float mean, rstd;
    gn_block_stats(stats, b, (double)Cout * (double)npos, eps, mean, rstd);
    const float g = gamma[o];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    const size_t base = ((size_t)b * Cout + o) * npos;
    if ((npos & 3) == 0) {
        // 16-byte loads, fp32 partial sums over the four elements of a load (then double): the three operand streams
        // are all this kernel does
        const long long n4 = npos >> 2;
        const f32x4* o4 = reinterpret_cast<const f32x4*>(out + base);
        const f32x4* d4 = reinterpret_cast<const f32x4*>(dout + base);
        const f32x4* y4 = reinterpret_cast<const f32x4*>(y + base);
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
            const f32x4 ov = o4[q], dv = d4[q], yv = y4[q];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dz = ov[e] > 0.0f ? dv[e] : 0.0f;
                s0 += dz;
                s1 += dz * ((yv[e] - mean) * rstd);
            }
            a[0] += (double)(s0 * g);
            a[1] += (double)(s1 * g);
            a[2] += (double)s1;
            a[3] += (double)s0;
        }
    } else {
        for (long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x; pos < npos;
             pos += (long long)gridDim.x * blockDim.x) {
            const float dz = out[base + pos] > 0.0f ? dout[base + pos] : 0.0f;
            const float yh = (y[base + pos] - mean) * rstd;
            a[0] += (double)(dz * g);
            a[1] += (double)(dz * g * yh);
            a[2] += (double)(dz * yh);
            a[3] += (double)dz;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[i] += __shfl_xor(a[i], off);
    __shared__ double sh[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int i = 0; i < 4; ++i) sh[wave][i] = a[i];
    __syncthreads();
    if (threadIdx.x < 4) {
        const double v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        double* dst = threadIdx.x < 2 ? red + b * 2 + threadIdx.x : red + 2 * B + o * 2 + (threadIdx.x - 2);
        atomicAdd(dst, v);
    }
}

__global__ __launch_bounds__(256) void gn_relu_bwd_apply_kernel(
    const float* __restrict__ y, const float* __restrict__ out, const float* __restrict__ dout,
    const double* __restrict__ stats, const float* __restrict__ gamma, float eps, int B, int Cout, long long npos,
    const double* __restrict__ red, float* __restrict__ dy, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int b = blockIdx.z, o = blockIdx.y;
    const long long pos = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
        dgamma[o] = (float)red[2 * B + o * 2];
        dbeta[o] = (float)red[2 * B + o * 2 + 1];
    }
    const double n = (double)Cout * (double)npos;
    float mean, rstd;
    gn_block_stats(stats, b, n, eps, mean, rstd);           // block-wide: before the early return
    if (pos >= npos) return;
    const float m1 = (float)(red[b * 2] / n), m2 = (float)(red[b * 2 + 1] / n);
    const size_t idx = ((size_t)b * Cout + o) * npos + pos;
    const float dz = out[idx] > 0.0f ? dout[idx] : 0.0f;
    const float yh = (y[idx] - mean) * rstd;
    dy[idx] = rstd * (dz * gamma[o] - m1 - yh * m2);
}

// ------------------------------------------------------------------------------------------------
// weight gradient of a 3x3 / stride 1 / pad 1 convolution over the LAST two dims of (B, C, G, H, W) volumes:
//   dW[o][c][i][j] = sum_{b,g,y,x} dy[b,o,g,y,x] * X[b,c,g,y+i-1,x+j-1],   db[o] = sum dy[b,o,...]
// i.e. one separable branch of Conv4d (the other branch is the same call on the volumes with the two index pairs
// swapped).  As a GEMM it is Cout x (Cin*9) with the contraction over millions of positions: MIOpen's choice for it
// ran 1.9 ms per call.  Here a workgroup walks (b,g) planes: the Cin input planes (1-pixel zero halo, so the nine
// shifted reads need no bounds logic) and the Cout gradient planes are staged in LDS; a wave owns one 16(o) x 16(c)
// output tile for all nine taps (9 accumulator tiles), dy is the MFMA A operand read once per 4 positions and reused
// by the nine v_mfma_f32_16x16x4_f32; partial sums leave once per workgroup with atomics.
// ------------------------------------------------------------------------------------------------
constexpr int WG_MAXC = 32;           // Cin, Cout <= 32
constexpr int WG_MAXP = 256;          // H * W <= 256
#ifndef CPN_WG_BLOCKS
#define CPN_WG_BLOCKS 512
#endif
constexpr int WG_BLOCKS = CPN_WG_BLOCKS;   // workgroups (= partial sums) per call
// SWAP (Cout <= 8 < Cin): the kernel is handed (dy, x, Cout, Cin) in place of (x, dy, Cin, Cout) - the gradient planes take
// the halo and the two-taps-per-column packing, since dW[o][c][tap] = sum_p dy[o][p - tap] x[c][p] is the same sum with the
// roles exchanged and the tap mirrored (8 - tap) - and writes its partial sums in the caller's (o, c, tap) order; the bias
// sum is then taken from the centre tap of the haloed operand.
template <int CI, int CO, bool SWAP = false>   // channel counts rounded up to 8 or 32: the staging registers of one plane
__global__ __launch_bounds__(256) void conv_wgrad_planes_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, int Cin, int Cout, int G, int H, int W, int nplanes,
    float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float wgl[];
    const int P = H * W, HP = (H + 2) * (W + 2);
    const int XS = HP + ((HP & 31) == 5 ? 0 : ((37 - (HP & 31)) & 31));     // plane stride = 5 (mod 32)
    const int DS = P + ((P & 31) == 4 ? 0 : ((36 - (P & 31)) & 31));        // row stride   = 4 (mod 32)
    const int CT = (Cin + 15) >> 4, MT = (Cout + 15) >> 4;
    float* xs = wgl;                                  // [CT*16][XS]   haloed input planes
    float* ds = wgl + CT * 16 * XS;                   // [MT*16][DS]   gradient planes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < CT * 16 * XS + MT * 16 * DS; i += 256) wgl[i] = 0.0f;     // halo + padding rows stay zero

    const int npairs = MT * CT;                       // 1, 2 or 4 output tiles
    const int pair = wave % npairs, kpart = wave / npairs, kparts = 4 / npairs;
    const int mt = pair % MT, ct = pair / MT;
    const int ln = lane & 15, lk = lane >> 4;
    // Cin <= 8 (CI == 8): the 16 columns of the MFMA's B operand hold 8 input channels x TWO taps (column n = channel n & 7
    // of tap 2 t + (n >> 3)), so the nine taps take five MFMAs per k step instead of nine with half of every tile empty
    // (these launches ran at twice their MFMA bound, which the padding had doubled: round 6)
    constexpr bool PACK = CI == 8;
    constexpr int NACC = PACK ? 5 : 9;
    const int tapsel = PACK ? (ln >> 3) : 0;
    f32x4 acc[NACC];
    int boff[NACC];                                   // PACK: offset of tap 2 t + tapsel in the haloed plane (the tenth slot re-reads tap 8: discarded)
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tap = min(2 * t + tapsel, 8);
        boff[t] = (tap / 3) * (W + 2) + tap % 3;
    }
    float bsum = 0.0f;
    const int ksteps = (P + 3) >> 2;
    const int k_lo = kpart * ksteps / kparts, k_hi = (kpart + 1) * ksteps / kparts;
    const float* arow = ds + (mt * 16 + ln) * DS;
    const float* brow = xs + (PACK ? (ln & 7) : ct * 16 + ln) * XS;
    // staging: P <= 256, so thread tid owns position tid of every plane
    const bool stg = tid < P;
    const int sy = stg ? tid / W : 0, sx = stg ? tid - sy * W : 0;
    const int hoff = (sy + 1) * (W + 2) + sx + 1;
    const int y0 = (k_lo * 4 + lk) / W, x0 = (k_lo * 4 + lk) - y0 * W;

    // a plane's Cin + Cout values of this thread's position travel through registers: all loads of a plane are issued
    // together, and the next plane's are in flight under the MFMA loop of the current one (the first version loaded and
    // stored channel by channel: Cin + Cout serialized HBM round trips per plane, 55 - 100 us per call for operands that
    // stream in 2 - 8 us)
    float vx[CI], vd[CO];
    const size_t cs = (size_t)G * P;
    auto load_plane = [&](int pl) {
        const int b = pl / G, g = pl - b * G;
        const float* xp = x + (((size_t)b * Cin) * G + g) * P + (stg ? tid : 0);
        const float* dp = dy + (((size_t)b * Cout) * G + g) * P + (stg ? tid : 0);
#pragma unroll
        for (int c = 0; c < CI; ++c) vx[c] = xp[(c < Cin ? c : Cin - 1) * cs];        // surplus slots repeat the last channel
#pragma unroll
        for (int o = 0; o < CO; ++o) vd[o] = dp[(o < Cout ? o : Cout - 1) * cs];
    };
    if ((int)blockIdx.x < nplanes) load_plane(blockIdx.x);
    for (int pl = blockIdx.x; pl < nplanes; pl += gridDim.x) {
        __syncthreads();                              // previous plane fully consumed (and the zero fill done)
        if (stg) {
#pragma unroll
            for (int c = 0; c < CI; ++c)
                if (c < Cin) xs[c * XS + hoff] = vx[c];
#pragma unroll
            for (int o = 0; o < CO; ++o)
                if (o < Cout) ds[o * DS + tid] = vd[o];
        }
        __syncthreads();
        if (pl + (int)gridDim.x < nplanes) load_plane(pl + gridDim.x);
        // operands of k step ks for this lane: gradient value av (rows are zero-padded past P: no predicate) and the 9
        // taps of the haloed input plane (positions past P read tap window (0,0) — finite values times av = 0).  The
        // next step's 10 LDS reads are issued before the current step's 9 MFMAs (predicated reads + a wait before every
        // MFMA ran this loop at a quarter of the MFMA rate)
        int yy = y0, xx = x0;
        bool bvalid = false;                           // SWAP: the position of the step just read lies inside the plane
        auto read_step = [&](int ks, float& av, float (&bv)[NACC]) {
            const int pos = ks * 4 + lk;
            av = arow[pos];
            if constexpr (SWAP) bvalid = pos < P;
            const float* bp = brow + (pos < P ? yy * (W + 2) + xx : 0);
            if constexpr (PACK) {
#pragma unroll
                for (int t = 0; t < 5; ++t) bv[t] = bp[boff[t]];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) bv[i * 3 + j] = bp[i * (W + 2) + j];
            }
            xx += 4;
            const bool wrap = xx >= W;                 // W >= 4: at most one row per step
            xx -= wrap ? W : 0;
            yy += wrap ? 1 : 0;
        };
        float av, bv[NACC];
        if (k_lo < k_hi) read_step(k_lo, av, bv);
        for (int ks = k_lo; ks < k_hi; ++ks) {
            if constexpr (SWAP) bsum += (bvalid && tapsel == 0) ? bv[2] : 0.0f;     // centre tap (4 = 2 * 2 + 0) of the gradient plane
            else bsum += av;
            float an = 0.0f, bn[NACC];
#pragma unroll
            for (int t = 0; t < NACC; ++t) bn[t] = 0.0f;
            if (ks + 1 < k_hi) read_step(ks + 1, an, bn);
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[t], acc[t], 0, 0, 0);
            av = an;
#pragma unroll
            for (int t = 0; t < NACC; ++t) bv[t] = bn[t];
        }
    }
    // partial sums of this workgroup: [Cout*Cin*9 weights | Cout biases]; D[m = lk*4 + e][n = ln]: m -> output
    // channel, n -> input channel.  Waves that split the positions of one tile (kparts > 1) combine with LDS atomics.
    const int nw = Cout * Cin * 9;
    const int nbias = SWAP ? Cin : Cout;              // the caller's output channels
    float* mine = partial + (size_t)blockIdx.x * (nw + nbias);
    __syncthreads();
    float* red = wgl;                                 // reuse LDS: nw + nbias floats <= 32*32*9 + 32
    for (int i = tid; i < nw + nbias; i += 256) red[i] = 0.0f;
    __syncthreads();
    const int c = PACK ? (ln & 7) : ct * 16 + ln;
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = mt * 16 + lk * 4 + e;
            const int tap = PACK ? 2 * t + tapsel : t;
            if (o < Cout && c < Cin && tap < 9) {
                // SWAP: (o, c) are (input channel, output channel) of the caller's problem, the tap is mirrored
                const size_t at = SWAP ? ((size_t)c * Cout + o) * 9 + (8 - tap) : ((size_t)o * Cin + c) * 9 + tap;
                if (kparts > 1) atomicAdd(red + at, acc[t][e]);
                else red[at] = acc[t][e];
            }
        }
    if constexpr (SWAP) {                             // bias of the caller's output channel c' = ln (tapsel 0 lanes), once per k part
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (mt == 0 && lk == 0 && ln < Cin) atomicAdd(red + nw + ln, bsum);
    } else if (ct == 0) {                             // bias: lanes (ln = o, lk) hold disjoint position subsets
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        const int o = mt * 16 + ln;
        if (lk == 0 && o < Cout) atomicAdd(red + nw + o, bsum);
    }
    __syncthreads();
    for (int i = tid; i < nw + nbias; i += 256) mine[i] = red[i];
}

__global__ __launch_bounds__(1024) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int nblocks, int nw,
                                                                 int Cout, float* __restrict__ dw,
                                                                 float* __restrict__ db) {
    // 64 outputs per workgroup, the partial sums of the nblocks producers split over 16 waves, 8 loads in flight per thread
    // (4 waves walking 64 partials each one load at a time took 17 us per call)
    __shared__ float sh[16][64];
    const int j = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + j;
    float sacc = 0.0f;
    if (i < nw + Cout) {
        const size_t ld = (size_t)(nw + Cout);
        int b = part;
        for (; b + 16 * 7 < nblocks; b += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 16 * u) * ld + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) sacc += v[u];
        }
        for (; b < nblocks; b += 16) sacc += partial[(size_t)b * ld + i];
    }
    sh[part][j] = sacc;
    __syncthreads();
    if (part == 0 && i < nw + Cout) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += sh[q][j];
        if (i < nw) dw[i] = v;
        else if (db) db[i - nw] = v;
    }
}

// weight / bias gradient of a depthwise 3x3, stride 1, pad 1 convolution (the DWConv of the UFC feed-forward blocks,
// models/aggregation.py): dw[c][i][j] = sum_{n,y,x} dy[n,c,y,x] * X[n,c,y+i-1,x+j-1],  db[c] = sum dy[n,c].
// Ten sums over N*H*W values per channel — MIOpen's batched-GEMM weight-gradient took 1.9 ms for it.
// One workgroup per channel.
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              int N, int C, int H, int W, float* __restrict__ dw,
                                                              float* __restrict__ db) {
    const int c = blockIdx.x, P = H * W;
    float a[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int idx = threadIdx.x; idx < N * P; idx += 256) {
        const int n = idx / P, pos = idx - n * P;
        const int yy = pos / W, xx = pos - yy * W;
        const size_t base = ((size_t)n * C + c) * P;
        const float g = dy[base + pos];
        a[9] += g;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int Y = yy + i - 1, X = xx + j - 1;
                if (Y >= 0 && Y < H && X >= 0 && X < W) a[i * 3 + j] += g * x[base + (size_t)Y * W + X];
            }
    }
    __shared__ float sh[4][10];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[t] += __shfl_xor(a[t], off);
        if (lane == 0) sh[wave][t] = a[t];
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const float v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (threadIdx.x < 9) dw[c * 9 + threadIdx.x] = v;
        else if (db) db[c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// The same depthwise convolution directly on the TOKEN layout (B, L = H*W, C) the feed-forward blocks work in
// (aggregation.py:18-28 transposes to (B,C,H,W), runs the grouped Conv2d and transposes back).  With the channel
// innermost the reads are coalesced float4 rows: no transposed copies on either side,
// and the library's depthwise kernel (0.76 ms for 4 x 1024 x 64 x 64, 12x the streaming time) is not needed.
//   flip = 0: y = b + sum_ij w[c][i][j] * x[p + (i-1, j-1)]        (forward)
//   flip = 1: y =     sum_ij w[c][2-i][2-j] * x[p + (i-1, j-1)]    (data gradient: correlation with the flipped taps)
// ------------------------------------------------------------------------------------------------
// thread = (run of DW_RUN consecutive x positions of one image row, 4 channels): the 36 taps of its channels are loaded
// once, and the three input rows slide through registers (3 new 16-byte loads per output instead of 9)
constexpr int DW_RUN = 8;
__global__ __launch_bounds__(256) void dwconv3x3_tokens_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, int H, int W, int C4,
                                                               long long total, int flip, float* __restrict__ y) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4 = (int)(idx % C4);
    long long t = idx / C4;
    const int runs = (W + DW_RUN - 1) / DW_RUN;
    const int xr = (int)(t % runs); t /= runs;
    const int yy = (int)(t % H);
    const long long bimg = t / H;
    const int C = C4 * 4, x0 = xr * DW_RUN;
    const f32x4 b4 = bias ? *reinterpret_cast<const f32x4*>(bias + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wr[9];                                              // wr[tap][channel e]
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int e = 0; e < 4; ++e) wr[tp][e] = w[(c4 * 4 + e) * 9 + ((flip & 1) ? 8 - tp : tp)];
    const float* xb = x + (bimg * H * W) * C + c4 * 4;
    auto col = [&](int X, f32x4 (&v)[3]) {                     // the three rows of column X (zeros outside the image)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int Y = yy + i - 1;
            v[i] = (X >= 0 && X < W && Y >= 0 && Y < H) ? *reinterpret_cast<const f32x4*>(xb + ((long long)Y * W + X) * C)
                                                         : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x4 c0[3], c1[3], c2[3];
    col(x0 - 1, c0);
    col(x0, c1);
#pragma unroll
    for (int k = 0; k < DW_RUN; ++k) {
        const int X = x0 + k;
        if (X >= W) break;
        col(X + 1, c2);
        f32x4 acc = b4;
#pragma unroll
        for (int i = 0; i < 3; ++i) acc += wr[i * 3] * c0[i] + wr[i * 3 + 1] * c1[i] + wr[i * 3 + 2] * c2[i];
        if (flip & 2) {                                       // exact GELU of the feed-forward block (nn.GELU(), aggregation.py:180)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = 0.5f * acc[e] * (1.0f + erff(acc[e] * 0.70710678118654752440f));
        }
        *reinterpret_cast<f32x4*>(y + ((bimg * H + yy) * W + X) * C + c4 * 4) = acc;
#pragma unroll
        for (int i = 0; i < 3; ++i) { c0[i] = c1[i]; c1[i] = c2[i]; }
    }
}

// weight / bias gradient in the token layout: block = (image row, 256 consecutive channels), thread = channel; the
// 3 x 3 input window slides along the row (3 new loads per position).  Each block leaves 10 partial sums per channel
// in `partial` (rows x C x 10); a second launch reduces them in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void dwconv3x3_tokens_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     int H, int W, int C, float* __restrict__ partial) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long rowid = blockIdx.x;                       // b * H + yy
    const int yy = (int)(rowid % H);
    const float* xr = x + (rowid - yy) * W * C + c;           // image base (+ channel)
    const float* dr = dy + rowid * W * C + c;
    float a[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto col = [&](int X, float (&v)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int Y = yy + i - 1;
            v[i] = (X >= 0 && X < W && Y >= 0 && Y < H) ? xr[((long long)Y * W + X) * C] : 0.0f;
        }
    };
    float c0[3], c1[3], c2[3];
    col(-1, c0);
    col(0, c1);
#pragma unroll 4
    for (int X = 0; X < W; ++X) {
        col(X + 1, c2);
        const float g = dr[(long long)X * C];
        a[9] += g;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i * 3] += g * c0[i];
            a[i * 3 + 1] += g * c1[i];
            a[i * 3 + 2] += g * c2[i];
            c0[i] = c1[i];
            c1[i] = c2[i];
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) partial[(rowid * C + c) * 10 + t] = a[t];
}

__global__ __launch_bounds__(1024) void dwconv3x3_tokens_wgrad_reduce_kernel(const float* __restrict__ partial, long long rows,
                                                                             int C, float* __restrict__ dw,
                                                                             float* __restrict__ db) {
    // 64 (channel, term) outputs per workgroup, the rows' partial sums split over 16 waves with 8 loads in flight per thread,
    // combined in a fixed order (round 6: one thread per output walking all B*H rows one load at a time, on C*10/256 = 40
    // workgroups, took 70 us per call - 2.4 x the kernel that produces the partials)
    __shared__ float sh[16][64];
    const int j = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + j;
    const size_t ld = (size_t)C * 10;
    float sacc = 0.0f;
    if (i < C * 10) {
        long long r = part;
        for (; r + 16 * 7 < rows; r += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + 16 * u) * ld + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) sacc += v[u];
        }
        for (; r < rows; r += 16) sacc += partial[(size_t)r * ld + i];
    }
    sh[part][j] = sacc;
    __syncthreads();
    if (part == 0 && i < C * 10) {
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sh[q][j];
        const int c = i / 10, t = i - c * 10;
        if (t < 9) dw[c * 9 + t] = acc;
        else if (db) db[c] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// correlation: tokens (B, L, C) -> x / (||x|| + eps), then C[b] = S_n[b] . T_n[b]^T with the exact-f32 MFMA
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long long rows, int C, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) ss += xr[c] * xr[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float d = sqrtf(ss) + eps;
    for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = xr[c] / d;
}

// VJP of the row normalisation y = x / (|x| + eps): dx = dy / (|x| + eps) - y (y . dy) / max(|x|, tiny) — one wave per row,
// the row in registers (C <= 1024).  Ten ATen launches per call in the training step's correlation backward before.
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ dy, float* __restrict__ dx,
                                                              long long rows, int C, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t base = (size_t)row * C;
    float xv[16], yv[16], gv[16];
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            xv[i] = x[base + c]; yv[i] = y[base + c]; gv[i] = dy[base + c];
            ss += xv[i] * xv[i];
            dot += yv[i] * gv[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); dot += __shfl_xor(dot, off); }
    const float r = sqrtf(ss);
    const float inv = 1.0f / (r + eps), k = dot / fmaxf(r, 1e-30f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + 64 * i;
        if (c < C) dx[base + c] = gv[i] * inv - yv[i] * k;
    }
}

// C (M x N) = A (M x K) . B (N x K)^T per batch; wave = 16 rows x 16*NT cols, block = 64 rows; K % 16 == 0.
// NT = 8 for large problems (fewer re-reads of B), NT = 2 when 128-column tiles would leave most of the chip idle
// (a 256 x 256 correlation is 8 workgroups at NT = 8); the operands of k block kb+16 are requested before the MFMAs of kb.
template <int NT>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                          float* __restrict__ Cm, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int m0 = (blockIdx.y * 4 + wave) * 16, n0 = blockIdx.x * 16 * NT;
    if (m0 >= M) return;
    const float* Ab = A + (size_t)b * M * K;
    const float* Bb = Bm + (size_t)b * N * K;
    float* Cb = Cm + (size_t)b * M * N;
    const int fi = lane & 15, fg = lane >> 4;
    int ar = m0 + fi;
    ar = ar < M ? ar : M - 1;
    const float* ap = Ab + (size_t)ar * K + fg * 4;
    const float* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 16 + fi;
        bp[t] = Bb + (size_t)(n < N ? n : N - 1) * K + fg * 4;       // columns past N are computed and not stored
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int kb, f32x4& av, f32x4 (&bv)[NT]) {
        av = *reinterpret_cast<const f32x4*>(ap + kb);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp[t] + kb);
    };
    auto step = [&](const f32x4& av, const f32x4 (&bv)[NT]) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t][e], av[e], acc[t], 0, 0, 0);
    };
    f32x4 a0, a1, b0[NT], b1[NT];
    load(0, a0, b0);
    int kb = 0;
    for (; kb + 32 <= K; kb += 32) {
        load(kb + 16, a1, b1);
        step(a0, b0);
        if (kb + 32 < K) load(kb + 32, a0, b0);
        step(a1, b1);
    }
    if (kb < K) step(a0, b0);
    const int m = m0 + fi;
    if (m >= M) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nb = n0 + t * 16 + fg * 4;
        if (nb + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<f32x4*>(Cb + (size_t)m * N + nb) = acc[t];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (nb + i < N) Cb[(size_t)m * N + nb + i] = acc[t][i];
        }
    }
}

// The same product for LARGE problems (the 4096 x 4096 x 256 correlations of the finest level): a wave owns a 64 x 64 output
// tile — 4 row tiles x 4 column tiles, 64 MFMAs per 8 operand loads where the kernel above has 32 per 9 — and a workgroup of
// four waves a 128 x 128 block.  Column tile t holds the columns {n0 + 4 j + t}, so a lane ends up with four CONSECUTIVE
// columns per row and stores 16-byte vectors.  Loads of the next k block are pinned in front of the MFMAs of this one.
// M, N multiples of 128, K of 16.  62 -> ~95 TFLOP/s on 4096 x 4096 x 256 (exact fp32 products, same k order per output).
__global__ __launch_bounds__(256, 2) void gemm_nt_f32_tile64_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                                    float* __restrict__ Cm, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128 + (wave & 1) * 64;
    const float* Ab = A + (size_t)b * M * K;
    const float* Bb = Bm + (size_t)b * N * K;
    float* Cb = Cm + (size_t)b * M * N;
    const int fi = lane & 15, fg = lane >> 4;
    const float* ap = Ab + (size_t)(m0 + fi) * K + fg * 4;                  // row tile a: + 16 a rows
    const float* bp = Bb + (size_t)(n0 + 4 * fi) * K + fg * 4;              // column tile t: + t rows
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int kb, f32x4 (&av)[4], f32x4 (&bv)[4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = *reinterpret_cast<const f32x4*>(ap + (size_t)a * 16 * K + kb);
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp + (size_t)t * K + kb);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](const f32x4 (&av)[4], const f32x4 (&bv)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[t][e], av[a][e], acc[a][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x4 a0[4], b0[4], a1[4], b1[4];
    load(0, a0, b0);
    int kb = 0;
    for (; kb + 32 <= K; kb += 32) {
        load(kb + 16, a1, b1);
        step(a0, b0);
        if (kb + 32 < K) load(kb + 32, a0, b0);
        step(a1, b1);
    }
    if (kb < K) step(a0, b0);
    // operands swapped (A operand = B rows): D row index = column sub-index, D column = A row fi.  Register r of tile (a, t)
    // in lane (fi, fg): C[m0 + 16 a + fi][n0 + 4 (4 fg + r) + t]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(Cb + (size_t)(m0 + 16 * a + fi) * N + n0 + 4 * (4 * fg + r)) =
                f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
}

// ------------------------------------------------------------------------------------------------
// soft-argmax with temperature beta over the 4-D correlation c (B, S = h*h, T = h*h)
//   rows: for every source pixel s, softmax over t   -> expected (x, y) of the target   (t_to_s maps)
//   cols: for every target pixel t, softmax over s   -> expected (x, y) of the source   (s_to_t maps)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lin11(int i, int n) { return -1.0f + (2.0f / (float)(n - 1)) * (float)i; }

__global__ __launch_bounds__(256) void soft_argmax_rows_kernel(const float* __restrict__ c, int h, float beta,
                                                               float* __restrict__ out) {
    const int T = h * h;
    const int b = blockIdx.y, s = blockIdx.x;
    const float* row = c + ((size_t)b * T + s) * T;
    __shared__ float red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, row[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f, sx = 0.f, sy = 0.f;
    {   // (x, y) of the position advanced by the stride instead of t % h and t / h per element
        int x = threadIdx.x % h, y = threadIdx.x / h;
        const int dx = 256 % h, dy = 256 / h;
        for (int t = threadIdx.x; t < T; t += 256) {
            const float e = expf((row[t] - m) / beta);
            se += e;
            sx += e * lin11(x, h);
            sy += e * lin11(y, h);
            x += dx; y += dy;
            if (x >= h) { x -= h; ++y; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off);
        sx += __shfl_xor(sx, off);
        sy += __shfl_xor(sy, off);
    }
    if (lane == 0) { red[4 + wave] = se; red[8 + wave] = sx; red[12 + wave] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[4] + red[5]) + (red[6] + red[7]);
        out[((size_t)b * 2 + 0) * T + s] = ((red[8] + red[9]) + (red[10] + red[11])) / tot;
        out[((size_t)b * 2 + 1) * T + s] = ((red[12] + red[13]) + (red[14] + red[15])) / tot;
    }
}

// Column-direction reductions over a (rows x T) matrix: block = CR_COLS columns x CR_GROUPS row groups (a wave reads four
// 64-byte row segments per instruction), online softmax per thread, partials merged through LDS.  With 64 columns x 4
// groups (round 1) a 4096 x 4096 matrix was 64 workgroups whose threads each walked 1024 rows: 0.39 ms for 67 MB.
constexpr int CR_COLS = 16, CR_GROUPS = 64;
__global__ __launch_bounds__(CR_COLS * CR_GROUPS) void soft_argmax_cols_kernel(const float* __restrict__ c, int h, float beta,
                                                                               float* __restrict__ out) {
    const int T = h * h;
    const int b = blockIdx.y;
    const int l = threadIdx.x % CR_COLS, g = threadIdx.x / CR_COLS;
    const int t = blockIdx.x * CR_COLS + l;
    __shared__ float part[CR_GROUPS][4][CR_COLS];
    float m = -INFINITY, se = 0.f, sx = 0.f, sy = 0.f;
    if (t < T) {
        const float* col = c + (size_t)b * T * T + t;
        int x = g % h, y = g / h;                              // (s % h, s / h), advanced by the stride
        const int dx = CR_GROUPS % h, dy = CR_GROUPS / h;
        for (int s = g; s < T; s += CR_GROUPS) {
            const float v = col[(size_t)s * T];
            if (v > m) {
                const float r = expf((m - v) / beta);
                se *= r; sx *= r; sy *= r;
                m = v;
            }
            const float e = expf((v - m) / beta);
            se += e;
            sx += e * lin11(x, h);
            sy += e * lin11(y, h);
            x += dx; y += dy;
            if (x >= h) { x -= h; ++y; }
        }
    }
    part[g][0][l] = m; part[g][1][l] = se; part[g][2][l] = sx; part[g][3][l] = sy;
    __syncthreads();
    if (g == 0 && t < T) {
        float M = -INFINITY;
        for (int q = 0; q < CR_GROUPS; ++q) M = fmaxf(M, part[q][0][l]);
        float E = 0.f, X = 0.f, Y = 0.f;
        for (int q = 0; q < CR_GROUPS; ++q) {
            const float r = part[q][0][l] == -INFINITY ? 0.0f : expf((part[q][0][l] - M) / beta);
            E += part[q][1][l] * r; X += part[q][2][l] * r; Y += part[q][3][l] * r;
        }
        out[((size_t)b * 2 + 0) * T + t] = X / E;
        out[((size_t)b * 2 + 1) * T + t] = Y / E;
    }
}

// ------------------------------------------------------------------------------------------------
// (N, P, Q) -> (N, Q, P) batched matrix transpose = swapping the (query, support) index pairs of a 4-D correlation
// volume, x.permute(0,1,4,5,2,3).contiguous() in the reference's notation (aggregation.py:275, 349 and every
// `rearrange` around conv4d.py).  32 x 32 tiles through LDS, coalesced on both sides; ATen's strided copy for this
// permutation runs at 0.8 TB/s and there are ~170 of them in a training step.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_pairs_kernel(const float* __restrict__ x, int P, int Q, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * P * Q;
    const int q0 = blockIdx.x * 32, p0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;              // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = p0 + ty + i * 8, q = q0 + tx;
        if (p < P && q < Q) tile[ty + i * 8][tx] = x[base + (size_t)p * Q + q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + ty + i * 8, p = p0 + tx;
        if (p < P && q < Q) y[base + (size_t)q * P + p] = tile[tx][ty + i * 8];
    }
}

// ------------------------------------------------------------------------------------------------
// a (1 - l) + b l as two multiplies and an add, never contracted into a multiply-add: the resize kernel and the two fused
// forms of the final correlation then produce the same bits whatever the compiler would have chosen in each of them
__device__ __forceinline__ float mix_nc(float a, float b, float l) {
#pragma clang fp contract(off)
    return a * (1.0f - l) + b * l;
}

// bilinear resize with align_corners=True of `planes` independent (h, w) images: F.interpolate as used by
// interpolate4d / forward_attention (aggregation.py:49-56, 285, 293, 299).  thread = output pixel; HBM-bound.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_ac_kernel(const float* __restrict__ src,
                                                                 float* __restrict__ dst, long long planes, int h,
                                                                 int w, int H, int W) {
    const long long total = planes * H * W;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(idx % W);
        const long long t = idx / W;
        const int Y = (int)(t % H);
        const long long pl = t / H;
        float fy, fx, ly, lx;
        int y0, x0;
        {
#pragma clang fp contract(off)                                // the fractions as torch forms them: a product, then a difference
            fy = sy * (float)Y;
            fx = sx * (float)X;
            y0 = (int)fy;
            x0 = (int)fx;
            ly = fy - (float)y0;
            lx = fx - (float)x0;
        }
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float* p = src + pl * h * w;
        const float top = mix_nc(p[y0 * w + x0], p[y0 * w + x1], lx);
        const float bot = mix_nc(p[y1 * w + x0], p[y1 * w + x1], lx);
        dst[idx] = mix_nc(top, bot, ly);
    }
}

// Adjoint of resize_bilinear_ac_kernel (what autograd needs behind F.interpolate(align_corners=True)): the gradient g on
// the (H, W) grid gathered onto the (h, w) grid, out[y][x] = sum_{Y,X} wy(Y, y) wx(X, x) g[Y][X] with the forward kernel's
// own tap / weight arithmetic.  A GATHER in a fixed order: deterministic, where the library's backward scatters with atomics.
__global__ __launch_bounds__(256) void resize_bilinear_ac_adjoint_kernel(const float* __restrict__ g, float* __restrict__ out,
                                                                         long long planes, int h, int w, int H, int W) {
    const long long total = planes * h * w;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % w);
        const long long t = idx / w;
        const int y = (int)(t % h);
        const long long pl = t / h;
        // fine rows / columns whose taps can touch coarse index y / x (conservative by one on each side, checked exactly below)
        const int Ylo = sy > 0.0f ? max(0, (int)((float)(y - 1) / sy) - 1) : 0;
        const int Yhi = sy > 0.0f ? min(H - 1, (int)((float)(y + 1) / sy) + 1) : H - 1;
        const int Xlo = sx > 0.0f ? max(0, (int)((float)(x - 1) / sx) - 1) : 0;
        const int Xhi = sx > 0.0f ? min(W - 1, (int)((float)(x + 1) / sx) + 1) : W - 1;
        const float* p = g + pl * H * W;
        float acc = 0.0f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            const float fy = sy * (float)Y;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1);
            const float ly = fy - (float)y0;
            float wy = 0.0f;
            if (y0 == y) wy += 1.0f - ly;
            if (y1 == y) wy += ly;
            if (wy == 0.0f) continue;
            float row = 0.0f;
            for (int X = Xlo; X <= Xhi; ++X) {
                const float fx = sx * (float)X;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1);
                const float lx = fx - (float)x0;
                float wx = 0.0f;
                if (x0 == x) wx += 1.0f - lx;
                if (x1 == x) wx += lx;
                row += wx * p[Y * W + X];
            }
            acc += wy * row;
        }
        out[idx] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// UFC.forward's last step (models/aggregation.py:549-553): the three levels' refined correlations, interpolate4d'ed to
// n^4, averaged.  interpolate4d (aggregation.py:49-56) is two bilinear passes (align_corners=True): over the target
// pair of dims, then over the source pair — here both passes of both coarse levels, the two adds and the division in
// ONE pass over the fine volume: each output reads 16 values of each coarse volume (cache resident: 256 KB and 4 MB
// per pair) and one of the fine one, in the arithmetic order of the separate kernels (resize_bilinear_ac_kernel twice,
// then (a + b) + c, then * (1/3) as the division by a scalar is evaluated).
// ---------------------------------------------------------------------------------------------
struct AxisTap {
    int i0, i1;
    float l;
};
__device__ __forceinline__ AxisTap axis_tap(int I, int h, float scale) {
#pragma clang fp contract(off)                                // (f - i0 must not become fma(scale, I, -i0): see mix_nc)
    const float f = scale * (float)I;
    AxisTap t;
    t.i0 = (int)f;
    t.i1 = t.i0 + (t.i0 < h - 1);
    t.l = f - (float)t.i0;
    return t;
}
__device__ __forceinline__ float blend4(float a, float b, float c, float d, float lx, float ly) {
    return mix_nc(mix_nc(a, b, lx), mix_nc(c, d, lx), ly);
}
// x (h,h,h,h) of one pair -> its interpolate4d value at (I, J, i, j) of the n^4 grid
__device__ __forceinline__ float interp4d_at(const float* __restrict__ x, int h, float sc, int I, int J, int i, int j) {
    const AxisTap ti = axis_tap(i, h, sc), tj = axis_tap(j, h, sc), tI = axis_tap(I, h, sc), tJ = axis_tap(J, h, sc);
    float y1[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float* p = x + ((size_t)((a ? tI.i1 : tI.i0) * h + (b ? tJ.i1 : tJ.i0))) * h * h;
            y1[a][b] = blend4(p[ti.i0 * h + tj.i0], p[ti.i0 * h + tj.i1], p[ti.i1 * h + tj.i0], p[ti.i1 * h + tj.i1], tj.l, ti.l);
        }
    return blend4(y1[0][0], y1[0][1], y1[1][0], y1[1][1], tJ.l, tI.l);
}
__global__ __launch_bounds__(256) void corr_mean3_kernel(const float* __restrict__ c0, int h0, const float* __restrict__ c1,
                                                         int h1, const float* __restrict__ c2, int n, int B,
                                                         float* __restrict__ out) {
    const long long per = (long long)n * n * n * n, total = per * B;
    const float s0 = n > 1 ? (float)(h0 - 1) / (float)(n - 1) : 0.0f, s1 = n > 1 ? (float)(h1 - 1) / (float)(n - 1) : 0.0f;
    const float third = 1.0f / 3.0f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long t = idx;
        const int j = (int)(t % n); t /= n;
        const int i = (int)(t % n); t /= n;
        const int J = (int)(t % n); t /= n;
        const int I = (int)(t % n);
        const long long b = t / n;
        const float u0 = interp4d_at(c0 + b * (long long)h0 * h0 * h0 * h0, h0, s0, I, J, i, j);
        const float u1 = interp4d_at(c1 + b * (long long)h1 * h1 * h1 * h1, h1, s1, I, J, i, j);
        out[idx] = ((u0 + u1) + c2[idx]) * third;
    }
}

// The same values, same arithmetic order, with the coarse planes in LDS: a workgroup owns ONE source position (I, J) of a
// pair and all n x n target positions.  Its eight source-tap planes (2 x 2 of each coarse volume) are staged with coalesced
// loads, every output then blends 2 x 16 LDS values instead of 2 x 16 global ones (the per-thread kernel above issues 537 M
// 4-byte global loads for a 64^4 volume and is bound by them: 175 us where the 134 MB it must move take 30).
__global__ __launch_bounds__(256) void corr_mean3_planes_kernel(const float* __restrict__ c0, int h0, const float* __restrict__ c1,
                                                                int h1, const float* __restrict__ c2, int n,
                                                                float* __restrict__ out) {
    extern __shared__ float planes[];                            // [4][h0*h0] then [4][h1*h1]
    const int I = blockIdx.x / n, J = blockIdx.x % n;
    const long long b = blockIdx.y;
    const float s0 = (float)(h0 - 1) / (float)(n - 1), s1 = (float)(h1 - 1) / (float)(n - 1);
    const int p0 = h0 * h0, p1 = h1 * h1;
    float* q1 = planes + 4 * p0;
    {
        const AxisTap tI = axis_tap(I, h0, s0), tJ = axis_tap(J, h0, s0);
        const float* base = c0 + b * (long long)p0 * p0;
        for (int t = threadIdx.x; t < 4 * p0; t += 256) {
            const int ab = t / p0, e = t - ab * p0;
            planes[t] = base[((size_t)(((ab >> 1) ? tI.i1 : tI.i0) * h0 + ((ab & 1) ? tJ.i1 : tJ.i0))) * p0 + e];
        }
    }
    {
        const AxisTap tI = axis_tap(I, h1, s1), tJ = axis_tap(J, h1, s1);
        const float* base = c1 + b * (long long)p1 * p1;
        for (int t = threadIdx.x; t < 4 * p1; t += 256) {
            const int ab = t / p1, e = t - ab * p1;
            q1[t] = base[((size_t)(((ab >> 1) ? tI.i1 : tI.i0) * h1 + ((ab & 1) ? tJ.i1 : tJ.i0))) * p1 + e];
        }
    }
    __syncthreads();
    const AxisTap uI0 = axis_tap(I, h0, s0), uJ0 = axis_tap(J, h0, s0), uI1 = axis_tap(I, h1, s1), uJ1 = axis_tap(J, h1, s1);
    const float third = 1.0f / 3.0f;
    const long long obase = ((b * n + I) * n + J) * (long long)n * n;
    auto at = [&](const float* pl, int h, float sc, int pp, const AxisTap& tI, const AxisTap& tJ, int i, int j) {
        const AxisTap ti = axis_tap(i, h, sc), tj = axis_tap(j, h, sc);
        float y1[4];
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) {
            const float* p = pl + ab * pp;
            y1[ab] = blend4(p[ti.i0 * h + tj.i0], p[ti.i0 * h + tj.i1], p[ti.i1 * h + tj.i0], p[ti.i1 * h + tj.i1], tj.l, ti.l);
        }
        return blend4(y1[0], y1[1], y1[2], y1[3], tJ.l, tI.l);
    };
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const float u0 = at(planes, h0, s0, p0, uI0, uJ0, i, j);
        const float u1 = at(q1, h1, s1, p1, uI1, uJ1, i, j);
        out[obase + e] = ((u0 + u1) + c2[obase + e]) * third;
    }
}

}  // namespace

extern "C" int cpn_corr_mean3(const float* c0, int h0, const float* c1, int h1, const float* c2, int n, int B, float* out,
                              void* stream) {
    CPN_REQUIRE(c0 && c1 && c2 && out, CPN_E_ARG, "cpn_corr_mean3: null pointer");
    CPN_REQUIRE(B > 0 && h0 > 1 && h1 > 1 && n > 1 && h0 <= n && h1 <= n && n <= 128, CPN_E_SHAPE, "cpn_corr_mean3: bad shape");
    const long long total = (long long)B * n * n * n * n;
    const size_t lds = (size_t)4 * ((size_t)h0 * h0 + (size_t)h1 * h1) * sizeof(float);
    if (lds <= 60 * 1024 && B < 65536) {
        hipLaunchKernelGGL(corr_mean3_planes_kernel, dim3((unsigned)(n * n), (unsigned)B), dim3(256), lds, (hipStream_t)stream, c0, h0,
                           c1, h1, c2, n, out);
        CPN_LAUNCH_CHECK("cpn_corr_mean3");
        return 0;
    }
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(total, 256), 1 << 20);
    hipLaunchKernelGGL(corr_mean3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c0, h0, c1, h1, c2, n, B, out);
    CPN_LAUNCH_CHECK("cpn_corr_mean3");
    return 0;
}

extern "C" int cpn_resize_bilinear_ac(const float* src, float* dst, long long planes, int h, int w, int H, int W,
                                      void* stream) {
    CPN_REQUIRE(src && dst, CPN_E_ARG, "cpn_resize_bilinear_ac: null pointer");
    CPN_REQUIRE(planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, CPN_E_SHAPE, "cpn_resize_bilinear_ac: bad shape");
    const long long total = planes * H * W;
    const unsigned blocks = (unsigned)(cpn_cdiv(total, 256) < 65536u ? cpn_cdiv(total, 256) : 65536u);
    hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, planes,
                       h, w, H, W);
    CPN_LAUNCH_CHECK("cpn_resize_bilinear_ac");
    return 0;
}

extern "C" int cpn_resize_bilinear_ac_adjoint(const float* g, float* out, long long planes, int h, int w, int H, int W,
                                              void* stream) {
    CPN_REQUIRE(g && out, CPN_E_ARG, "cpn_resize_bilinear_ac_adjoint: null pointer");
    CPN_REQUIRE(planes > 0 && h > 0 && w > 0 && H > 0 && W > 0, CPN_E_SHAPE, "cpn_resize_bilinear_ac_adjoint: bad shape");
    const long long total = planes * h * w;
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(total, 256), 1 << 20);
    hipLaunchKernelGGL(resize_bilinear_ac_adjoint_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, out, planes, h, w,
                       H, W);
    CPN_LAUNCH_CHECK("cpn_resize_bilinear_ac_adjoint");
    return 0;
}

// doubles of the `stats` argument of cpn_conv4d / cpn_conv4d_gn_relu (layout: gn_publish); an upper bound over the
// kernel variants' grids
extern "C" long long cpn_gn_stats_doubles(int B, int Cout, long long npos) {
    // VALU kernels: one workgroup per 256 positions and output channel (group); MFMA kernel: one per 256 positions (64 in round 3)
    const long long wg = std::max<long long>((long long)cpn_cdiv(npos, 256) * Cout, (long long)cpn_cdiv(npos, 64));
    return 3LL * B + 2LL * B * wg;
}

extern "C" long long cpn_conv4d_scratch(int B, int Cin, int Hq, int Wq, int Hs, int Ws, int s) {
    if (s <= 1) return 0;
    auto po = [&](int n) { return (long long)((n + s - 1) / s); };
    return (long long)B * Cin * ((long long)Hq * Wq * po(Hs) * po(Ws) + po(Hq) * po(Wq) * (long long)Hs * Ws);
}

extern "C" int cpn_conv4d(const float* x, const float* wq, const float* bq, const float* ws, const float* bs, int B,
                          int Cin, int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s, int p, float* y,
                          double* stats, float* scratch, void* stream) {
    CPN_REQUIRE(x && wq && bq && ws && bs && y && stats, CPN_E_ARG, "cpn_conv4d: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && Cin > 0 && Cout > 0 && Cout < 65536 && k > 0 && s > 0 && p >= 0, CPN_E_SHAPE,
                "cpn_conv4d: bad shape");
    auto co = [&](int n) { return (n + 2 * p - k) / s + 1; };
    auto po = [&](int n) { return (n + s - 1) / s; };
    const int Oq = co(Hq), Pq_ = co(Wq), Os = co(Hs), Ps_ = co(Ws);
    CPN_REQUIRE(Oq == po(Hq) && Pq_ == po(Wq) && Os == po(Hs) && Ps_ == po(Ws), CPN_E_SHAPE,
                "cpn_conv4d: conv output (%d) and pooled size (%d) of the two branches disagree", Oq, po(Hq));
    const long long npos = (long long)Oq * Pq_ * Os * Ps_;
    const hipStream_t st = (hipStream_t)stream;
    dim3 grid(cpn_cdiv(npos, 256), Cout, B);
    const size_t wbytes = (size_t)Cin * 9 * 2 * Cout * sizeof(float);
    // The MFMA form (conv4d_k3s1_mfma_kernel: 64 positions per wave, 12 vector loads per 72 MT MFMAs) where its layout rules
    // hold; CPN_CONV4D_MFMA=0 keeps the VALU kernel.  (The round-3 MFMA form — 16 positions per wave, one 4-byte load per
    // MFMA — measured 40 / 32 us per Cout = 8 / 32 layer against 24 / 34 for the VALU kernel and was opt-in.)
    static const bool use_mfma = !(getenv("CPN_CONV4D_MFMA") && getenv("CPN_CONV4D_MFMA")[0] == '0');
    const int mtiles = (Cout + 15) / 16;
    const size_t wb_mfma = (size_t)18 * Cin * mtiles * 16 * sizeof(float);
    const bool ws_ok = Ws == 4 || Ws == 8 || Ws == 16 || Ws == 32 || Ws == 64;
    if (use_mfma && k == 3 && s == 1 && p == 1 && Cin % 4 == 0 && mtiles <= 2 && wb_mfma <= 64 * 1024 && ws_ok && npos % 64 == 0 &&
        (long long)Cin * npos * 4 < 0x7ffffff0LL - 4LL * npos * 4 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0) {
        // gn_publish counts gridDim.x * gridDim.y workgroups per sample: one workgroup = 256 positions, all channels
        dim3 g2(cpn_cdiv(npos, 256), 1, B);
        if (mtiles == 1)
            hipLaunchKernelGGL((conv4d_k3s1_mfma_kernel<1, false>), g2, dim3(256), wb_mfma, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs,
                               Ws, Cout, y, stats);
        else
            hipLaunchKernelGGL((conv4d_k3s1_mfma_kernel<2, false>), g2, dim3(256), wb_mfma, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs,
                               Ws, Cout, y, stats);
    } else if (k == 3 && s == 1 && p == 1 && (Cout == 8 || Cout == 32) && wbytes <= 64 * 1024 &&
        (long long)Cin * npos * 4 < 0x7ffffff0LL) {
        // channels per thread: 8 (4 when one thread per position would leave <= 2 waves per SIMD).  All 32 channels in one
        // thread read every input once instead of four times, but the compiler cannot hold 32 accumulators + a tap's
        // weights + 18 taps under 128 VGPRs (512 VGPRs = one wave per SIMD, or a kilobyte of scratch spills when
        // forced): 8 channels fit in 124 and run four waves per SIMD.
        const bool small = (long long)B * npos <= 131072;
        const int per = (Cout == 8 && small) ? 4 : 8;
        dim3 g1(cpn_cdiv(npos, 256), Cout / per, B);
        const size_t wb = wbytes / (Cout / per);
        if (per == 4)
            hipLaunchKernelGGL((conv4d_k3s1_kernel<4, true>), g1, dim3(256), wb, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws,
                               Cout, y, stats);
        else if (small)
            hipLaunchKernelGGL((conv4d_k3s1_kernel<8, true>), g1, dim3(256), wb, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws,
                               Cout, y, stats);
        else
            hipLaunchKernelGGL((conv4d_k3s1_kernel<8, false>), g1, dim3(256), wb, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws,
                               Cout, y, stats);
    } else if (s > 1 && scratch) {
        float* psv = scratch;
        float* pqv = scratch + (size_t)B * Cin * Hq * Wq * Os * Ps_;
        const long long planes = (long long)B * Cin * Hq * Wq;
        const long long n1 = planes * Os * Ps_, n2 = (long long)B * Cin * Oq * Pq_ * Hs * Ws;
        hipLaunchKernelGGL(pool_support_kernel, dim3((unsigned)std::min<long long>(cpn_cdiv(n1, 256), 65536)), dim3(256), 0,
                           st, x, Hs, Ws, s, Os, Ps_, planes, psv);
        hipLaunchKernelGGL(pool_query_kernel, dim3((unsigned)std::min<long long>(cpn_cdiv(n2, 256), 65536)), dim3(256), 0,
                           st, x, Hq, Wq, Hs, Ws, s, Oq, Pq_, (long long)B * Cin, pqv);
        if (Cout == 8 && (size_t)Cin * k * k * 16 * sizeof(float) <= 48 * 1024)
            hipLaunchKernelGGL(conv4d_pooled_c8_kernel, dim3(cpn_cdiv(npos, 256), 1, B), dim3(256),
                               (size_t)Cin * k * k * 16 * sizeof(float), st, psv, pqv, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws, k, s, p,
                               Oq, Pq_, Os, Ps_, y, stats);
        else
            hipLaunchKernelGGL(conv4d_pooled_kernel, grid, dim3(256), 0, st, psv, pqv, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws, k, s,
                               p, Oq, Pq_, Os, Ps_, y, stats);
    } else {
        hipLaunchKernelGGL(conv4d_kernel, grid, dim3(256), 0, st, x, wq, bq, ws, bs, Cin, Hq, Wq, Hs, Ws, k, s, p, Oq,
                           Pq_, Os, Ps_, y, stats);
    }
    CPN_LAUNCH_CHECK("cpn_conv4d");
    return 0;
}

extern "C" int cpn_conv4d_dgrad(const float* dy, const float* wq, const float* ws, int B, int Cout, int Cin, int Hq, int Wq,
                                int Hs, int Ws, float* dx, void* stream) {
    CPN_REQUIRE(dy && wq && ws && dx, CPN_E_ARG, "cpn_conv4d_dgrad: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && Cin > 0 && Cout > 0 && Hq > 0 && Wq > 0 && Hs > 0 && Ws > 0, CPN_E_SHAPE,
                "cpn_conv4d_dgrad: bad shape");
    const long long npos = (long long)Hq * Wq * Hs * Ws;
    // the data gradient is a k3 s1 p1 convolution of dy (Cout channels) producing Cin channels: channel groups of 8 (4)
    CPN_REQUIRE((Cin % 4) == 0 && (long long)Cout * npos * 4 < 0x7ffffff0LL, CPN_E_SHAPE,
                "cpn_conv4d_dgrad: need Cin %% 4 == 0 and a batch element below 2 GiB (Cin=%d)", Cin);
    {
        static const bool use_mfma = !(getenv("CPN_CONV4D_MFMA") && getenv("CPN_CONV4D_MFMA")[0] == '0');
        const int mtiles = (Cin + 15) / 16;                                 // the gradient convolves dy (Cout channels) into Cin
        const size_t wb_mfma = (size_t)18 * Cout * mtiles * 16 * sizeof(float);
        const bool ws_ok = Ws == 4 || Ws == 8 || Ws == 16 || Ws == 32 || Ws == 64;
        if (use_mfma && Cout % 4 == 0 && mtiles <= 2 && wb_mfma <= 64 * 1024 && ws_ok && npos % 64 == 0 &&
            (long long)Cout * npos * 4 < 0x7ffffff0LL - 4LL * npos * 4 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dx % 16) == 0) {
            dim3 g2(cpn_cdiv(npos, 256), 1, B);
            const hipStream_t st2 = (hipStream_t)stream;
            if (mtiles == 1)
                hipLaunchKernelGGL((conv4d_k3s1_mfma_kernel<1, true>), g2, dim3(256), wb_mfma, st2, dy, wq, (const float*)nullptr, ws,
                                   (const float*)nullptr, Cout, Hq, Wq, Hs, Ws, Cin, dx, (double*)nullptr);
            else
                hipLaunchKernelGGL((conv4d_k3s1_mfma_kernel<2, true>), g2, dim3(256), wb_mfma, st2, dy, wq, (const float*)nullptr, ws,
                                   (const float*)nullptr, Cout, Hq, Wq, Hs, Ws, Cin, dx, (double*)nullptr);
            CPN_LAUNCH_CHECK("cpn_conv4d_dgrad");
            return 0;
        }
    }
    const int per = (Cin % 8) == 0 ? 8 : 4;
    const size_t wb = (size_t)Cout * 9 * 2 * per * sizeof(float);
    CPN_REQUIRE(wb <= 64 * 1024, CPN_E_SHAPE, "cpn_conv4d_dgrad: filter slice exceeds 64 KiB of LDS (Cout=%d)", Cout);
    dim3 g1(cpn_cdiv(npos, 256), Cin / per, B);
    const hipStream_t st = (hipStream_t)stream;
    if (per == 4)
        hipLaunchKernelGGL((conv4d_k3s1_kernel<4, false, true>), g1, dim3(256), wb, st, dy, wq, (const float*)nullptr, ws,
                           (const float*)nullptr, Cout, Hq, Wq, Hs, Ws, Cin, dx, (double*)nullptr);
    else
        hipLaunchKernelGGL((conv4d_k3s1_kernel<8, false, true>), g1, dim3(256), wb, st, dy, wq, (const float*)nullptr, ws,
                           (const float*)nullptr, Cout, Hq, Wq, Hs, Ws, Cin, dx, (double*)nullptr);
    CPN_LAUNCH_CHECK("cpn_conv4d_dgrad");
    return 0;
}

extern "C" int cpn_transpose_pairs(const float* x, int N, int P, int Q, float* y, void* stream) {
    CPN_REQUIRE(x && y, CPN_E_ARG, "cpn_transpose_pairs: null pointer");
    CPN_REQUIRE(N > 0 && N < 65536 && P > 0 && Q > 0, CPN_E_SHAPE, "cpn_transpose_pairs: bad shape (N=%d)", N);
    hipLaunchKernelGGL(transpose_pairs_kernel, dim3(cpn_cdiv(Q, 32), cpn_cdiv(P, 32), N), dim3(256), 0, (hipStream_t)stream, x, P,
                       Q, y);
    CPN_LAUNCH_CHECK("cpn_transpose_pairs");
    return 0;
}

extern "C" int cpn_gn_relu(const float* y, const double* stats, const float* gn_w, const float* gn_b, const float* residual,
                           float eps, int B, int C, long long npos, float* out, void* stream) {
    CPN_REQUIRE(y && stats && gn_w && gn_b && out, CPN_E_ARG, "cpn_gn_relu: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && C > 0 && C < 65536 && npos > 0, CPN_E_SHAPE, "cpn_gn_relu: bad shape");
    dim3 grid(cpn_cdiv(npos, 256), C, B);
    hipLaunchKernelGGL(gn_relu_kernel, grid, dim3(256), 0, (hipStream_t)stream, y, stats, gn_w, gn_b, residual, eps, C, npos,
                       out);
    CPN_LAUNCH_CHECK("cpn_gn_relu");
    return 0;
}

extern "C" int cpn_conv4d_gn_relu(const float* x, const float* wq, const float* bq, const float* ws, const float* bs,
                                  const float* gn_w, const float* gn_b, const float* residual, float eps, int B, int Cin,
                                  int Cout, int Hq,
                                  int Wq, int Hs, int Ws, int k, int s, int p, float* y, double* stats, float* scratch,
                                  void* stream) {
    CPN_REQUIRE(gn_w && gn_b, CPN_E_ARG, "cpn_conv4d_gn_relu: null pointer");
    int rc = cpn_conv4d(x, wq, bq, ws, bs, B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, y, stats, scratch, stream);
    if (rc) return rc;
    auto co = [&](int n) { return (n + 2 * p - k) / s + 1; };
    const long long npos = (long long)co(Hq) * co(Wq) * co(Hs) * co(Ws);
    return cpn_gn_relu(y, stats, gn_w, gn_b, residual, eps, B, Cout, npos, y, stream);
}

extern "C" int cpn_gn_relu_bwd(const float* y, const float* out, const float* dout, const double* stats,
                               const float* gn_w, float eps, int B, int C, long long npos, double* red, float* dy,
                               float* dgn_w, float* dgn_b, void* stream) {
    CPN_REQUIRE(y && out && dout && stats && gn_w && red && dy && dgn_w && dgn_b, CPN_E_ARG,
                "cpn_gn_relu_bwd: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && C > 0 && C < 65536 && npos > 0, CPN_E_SHAPE, "cpn_gn_relu_bwd: bad shape");
    const hipStream_t st = (hipStream_t)stream;
    const unsigned bx = (unsigned)std::min<long long>(cpn_cdiv(npos, 4096), 16);       // 4 x 16-byte loads per thread
    hipLaunchKernelGGL(gn_relu_bwd_reduce_kernel, dim3(bx, C, B), dim3(256), 0, st, y, out, dout, stats, gn_w, eps, B, C,
                       npos, red);
    CPN_LAUNCH_CHECK("cpn_gn_relu_bwd(reduce)");
    hipLaunchKernelGGL(gn_relu_bwd_apply_kernel, dim3(cpn_cdiv(npos, 256), C, B), dim3(256), 0, st, y, out, dout, stats,
                       gn_w, eps, B, C, npos, red, dy, dgn_w, dgn_b);
    CPN_LAUNCH_CHECK("cpn_gn_relu_bwd(apply)");
    return 0;
}

// backward of the two soft-argmax directions: with p = softmax over the reduced index of c / beta and (ox, oy) the
// forward outputs, d c = p / beta * (gx * (x - ox) + gy * (y - oy)).  Rows first (writes dc), columns second (adds).
__global__ __launch_bounds__(256) void soft_argmax_rows_bwd_kernel(const float* __restrict__ c, int h, float beta,
                                                                   const float* __restrict__ out,
                                                                   const float* __restrict__ gout,
                                                                   float* __restrict__ dc) {
    const int T = h * h;
    const int b = blockIdx.y, s = blockIdx.x;
    const float* row = c + ((size_t)b * T + s) * T;
    float* drow = dc + ((size_t)b * T + s) * T;
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, row[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) se += expf((row[t] - m) / beta);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
    if (lane == 0) red[4 + wave] = se;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + (red[6] + red[7])) * beta);
    const float ox = out[((size_t)b * 2 + 0) * T + s], oy = out[((size_t)b * 2 + 1) * T + s];
    const float gx = gout[((size_t)b * 2 + 0) * T + s], gy = gout[((size_t)b * 2 + 1) * T + s];
    for (int t = threadIdx.x; t < T; t += 256) {
        const float p = expf((row[t] - m) / beta) * inv;
        drow[t] = p * (gx * (lin11(t % h, h) - ox) + gy * (lin11(t / h, h) - oy));
    }
}

__global__ __launch_bounds__(256) void soft_argmax_cols_bwd_kernel(const float* __restrict__ c, int h, float beta,
                                                                   const float* __restrict__ out,
                                                                   const float* __restrict__ gout,
                                                                   float* __restrict__ dc) {
    const int T = h * h;
    const int b = blockIdx.y;
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + l;
    __shared__ float part[4][2][64];
    __shared__ float fin[2][64];
    float m = -INFINITY, se = 0.f;
    const float* col = c + (size_t)b * T * T + t;
    if (t < T)
        for (int s = g; s < T; s += 4) {
            const float v = col[(size_t)s * T];
            if (v > m) { se *= expf((m - v) / beta); m = v; }
            se += expf((v - m) / beta);
        }
    part[g][0][l] = m; part[g][1][l] = se;
    __syncthreads();
    if (g == 0) {
        const float M = fmaxf(fmaxf(part[0][0][l], part[1][0][l]), fmaxf(part[2][0][l], part[3][0][l]));
        float E = 0.f;
        for (int q = 0; q < 4; ++q) E += part[q][1][l] * expf((part[q][0][l] - M) / beta);
        fin[0][l] = M; fin[1][l] = 1.0f / (E * beta);
    }
    __syncthreads();
    if (t >= T) return;
    const float M = fin[0][l], inv = fin[1][l];
    const float ox = out[((size_t)b * 2 + 0) * T + t], oy = out[((size_t)b * 2 + 1) * T + t];
    const float gx = gout[((size_t)b * 2 + 0) * T + t], gy = gout[((size_t)b * 2 + 1) * T + t];
    float* dcol = dc + (size_t)b * T * T + t;
    for (int s = g; s < T; s += 4) {
        const float q = expf((col[(size_t)s * T] - M) / beta) * inv;
        dcol[(size_t)s * T] += q * (gx * (lin11(s % h, h) - ox) + gy * (lin11(s / h, h) - oy));
    }
}

// ------------------------------------------------------------------------------------------------
// dual softmax of the "fundamental-matrix" cross attention (models/backbone.py:296-330):
//   f[b,i,j] = softmax_j(a[b,i,:])[j] * softmax_i(a[b,:,j])[i]        a: (B, L, M), up to 4096 x 4096
// torch runs two softmax kernels (the strided one at ~230 GB/s) and a product.  Here: row statistics (max, sum) with
// one wave per row, column statistics with 64 coalesced columns per workgroup and an online softmax down the rows, and
// one elementwise pass f = exp(2a - rmax_i - cmax_j) / (rsum_i * csum_j).  The statistics are kept for the backward:
//   da = 2 f df - r * Srow_i - c * Scol_j,   Srow_i = sum_j f df,  Scol_j = sum_i f df,  r / c the two softmaxes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                        long long rows, int M, float* __restrict__ stat, int mode) {
    // mode 0: stat[row] = (max, sum exp(a - max));  mode 1: stat[row] = sum a*w   (w = second operand, same shape)
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* ar = a + (size_t)row * M;
    if (mode == 0) {
        float m = -INFINITY;
        for (int j = lane; j < M; j += 64) m = fmaxf(m, ar[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float se = 0.f;
        for (int j = lane; j < M; j += 64) se += expf(ar[j] - m);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) se += __shfl_xor(se, off);
        if (lane == 0) { stat[row * 2] = m; stat[row * 2 + 1] = se; }
    } else {
        const float* wr = w + (size_t)row * M;
        float sacc = 0.f;
        for (int j = lane; j < M; j += 64) sacc += ar[j] * wr[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
        if (lane == 0) stat[row] = sacc;
    }
}

// workgroup = CR_COLS columns x CR_GROUPS row groups, partials merged through LDS (see soft_argmax_cols_kernel)
__global__ __launch_bounds__(CR_COLS * CR_GROUPS) void col_stats_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                                        int L, int M, float* __restrict__ stat, int mode) {
    const int b = blockIdx.y;
    const int l = threadIdx.x % CR_COLS, g = threadIdx.x / CR_COLS;
    const int j = blockIdx.x * CR_COLS + l;
    __shared__ float part[CR_GROUPS][2][CR_COLS];
    float m = -INFINITY, se = 0.f;
    if (j < M) {
        const float* col = a + (size_t)b * L * M + j;
        if (mode == 0) {
            for (int i = g; i < L; i += CR_GROUPS) {
                const float v = col[(size_t)i * M];
                if (v > m) { se *= expf(m - v); m = v; }
                se += expf(v - m);
            }
        } else {
            const float* wc = w + (size_t)b * L * M + j;
            for (int i = g; i < L; i += CR_GROUPS) se += col[(size_t)i * M] * wc[(size_t)i * M];
        }
    }
    part[g][0][l] = m; part[g][1][l] = se;
    __syncthreads();
    if (g == 0 && j < M) {
        if (mode == 0) {
            float Mx = -INFINITY;
            for (int q = 0; q < CR_GROUPS; ++q) Mx = fmaxf(Mx, part[q][0][l]);
            float E = 0.f;
            for (int q = 0; q < CR_GROUPS; ++q)
                E += part[q][0][l] == -INFINITY ? 0.0f : part[q][1][l] * expf(part[q][0][l] - Mx);
            stat[((size_t)b * M + j) * 2] = Mx;
            stat[((size_t)b * M + j) * 2 + 1] = E;
        } else {
            float E = 0.f;
            for (int q = 0; q < CR_GROUPS; ++q) E += part[q][1][l];
            stat[(size_t)b * M + j] = E;
        }
    }
}

__global__ __launch_bounds__(256) void dual_softmax_apply_kernel(const float* __restrict__ a,
                                                                 const float* __restrict__ rstat,
                                                                 const float* __restrict__ cstat, int L, int M,
                                                                 long long total, float* __restrict__ f) {
    // a workgroup walks whole rows (row = b*L + i): the element's (row, column, batch) came out of three 64-bit divisions
    // per element before, which cost more than its two exponentials (0.18 ms for a 268 MB matrix that streams in 0.09)
    const long long rows = total / M;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long long b = row / L;
        const float rm = rstat[row * 2], rs = rstat[row * 2 + 1];
        const float* cst = cstat + b * M * 2;
        const float* ar = a + row * M;
        float* fr = f + row * M;
        for (int j = threadIdx.x; j < M; j += blockDim.x) {
            const float cm = cst[j * 2], cs = cst[j * 2 + 1];
            const float v = ar[j];
            fr[j] = (expf(v - rm) / rs) * (expf(v - cm) / cs);
        }
    }
}

__global__ __launch_bounds__(256) void dual_softmax_bwd_apply_kernel(
    const float* __restrict__ a, const float* __restrict__ rstat, const float* __restrict__ cstat,
    const float* __restrict__ df, const float* __restrict__ srow, const float* __restrict__ scol, int L, int M,
    long long total, float* __restrict__ da) {
    const long long rows = total / M;
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {        // whole rows per workgroup: see dual_softmax_apply_kernel
        const long long b = row / L;
        const float rm = rstat[row * 2], rs = rstat[row * 2 + 1], sr = srow[row];
        const float* cst = cstat + b * M * 2;
        const float* scl = scol + b * M;
        const size_t off = (size_t)row * M;
        for (int j = threadIdx.x; j < M; j += blockDim.x) {
            const float v = a[off + j];
            const float r = expf(v - rm) / rs;
            const float c = expf(v - cst[j * 2]) / cst[j * 2 + 1];
            da[off + j] = 2.0f * r * c * df[off + j] - r * sr - c * scl[j];
        }
    }
}

extern "C" long long cpn_conv_wgrad_scratch(int Cin, int Cout) {
    return (long long)WG_BLOCKS * ((long long)Cout * Cin * 9 + Cout);
}

extern "C" int cpn_conv_wgrad_planes(const float* x, const float* dy, int B, int Cin, int Cout, int G, int H, int W,
                                     float* partial, float* dw, float* db, void* stream) {
    CPN_REQUIRE(x && dy && dw && partial, CPN_E_ARG, "cpn_conv_wgrad_planes: null pointer");
    CPN_REQUIRE(B > 0 && G > 0 && H > 0 && W >= 4 && Cin > 0 && Cout > 0, CPN_E_SHAPE,
                "cpn_conv_wgrad_planes: bad shape");
    CPN_REQUIRE(Cin <= WG_MAXC && Cout <= WG_MAXC && H * W <= WG_MAXP, CPN_E_SHAPE,
                "cpn_conv_wgrad_planes: supports Cin, Cout <= %d and H*W <= %d (got %d, %d, %d)", WG_MAXC, WG_MAXP, Cin,
                Cout, H * W);
    CPN_REQUIRE((long long)B * G < (1LL << 31), CPN_E_SHAPE, "cpn_conv_wgrad_planes: too many planes");
    const int P = H * W, HP = (H + 2) * (W + 2);
    const int XS = HP + ((HP & 31) == 5 ? 0 : ((37 - (HP & 31)) & 31));
    const int DS = P + ((P & 31) == 4 ? 0 : ((36 - (P & 31)) & 31));
    const bool swap = Cout <= 8 && Cin > 8;                  // conv_wgrad_planes_kernel<8, 32, true>: the operands change places
    const int CT = ((swap ? Cout : Cin) + 15) / 16, MT = ((swap ? Cin : Cout) + 15) / 16;
    const size_t lds = std::max((size_t)CT * 16 * XS + (size_t)MT * 16 * DS, (size_t)Cout * Cin * 9 + Cout) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        for (const void* k : {(const void*)conv_wgrad_planes_kernel<8, 8>, (const void*)conv_wgrad_planes_kernel<8, 32>,
                              (const void*)conv_wgrad_planes_kernel<8, 32, true>, (const void*)conv_wgrad_planes_kernel<32, 32>}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) {
                cpn_set_error("cpn_conv_wgrad_planes: cannot reserve LDS: %s", hipGetErrorString(e));
                return (int)e;
            }
        }
        attr_set = true;
    }
    const int nplanes = B * G;
    const int blocks = std::min(nplanes, WG_BLOCKS);
    const hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks), block(256);
    if (Cin <= 8 && Cout <= 8)
        hipLaunchKernelGGL((conv_wgrad_planes_kernel<8, 8>), grid, block, lds, st, x, dy, Cin, Cout, G, H, W, nplanes, partial);
    else if (Cin <= 8)
        hipLaunchKernelGGL((conv_wgrad_planes_kernel<8, 32>), grid, block, lds, st, x, dy, Cin, Cout, G, H, W, nplanes, partial);
    else if (Cout <= 8)                               // roles exchanged: the 8 gradient planes take the two-taps-per-column form
        hipLaunchKernelGGL((conv_wgrad_planes_kernel<8, 32, true>), grid, block, lds, st, dy, x, Cout, Cin, G, H, W, nplanes, partial);
    else
        hipLaunchKernelGGL((conv_wgrad_planes_kernel<32, 32>), grid, block, lds, st, x, dy, Cin, Cout, G, H, W, nplanes, partial);
    CPN_LAUNCH_CHECK("cpn_conv_wgrad_planes");
    const int nw = Cout * Cin * 9;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cpn_cdiv(nw + Cout, 64)), dim3(1024), 0, st, partial, blocks, nw, Cout,
                       dw, db);
    CPN_LAUNCH_CHECK("cpn_conv_wgrad_planes(reduce)");
    return 0;
}

extern "C" int cpn_dwconv3x3_wgrad(const float* x, const float* dy, int N, int C, int H, int W, float* dw, float* db,
                                   void* stream) {
    CPN_REQUIRE(x && dy && dw, CPN_E_ARG, "cpn_dwconv3x3_wgrad: null pointer");
    CPN_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && (long long)N * H * W < (1LL << 31), CPN_E_SHAPE,
                "cpn_dwconv3x3_wgrad: bad shape");
    hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, dy, N, C, H, W, dw, db);
    CPN_LAUNCH_CHECK("cpn_dwconv3x3_wgrad");
    return 0;
}

extern "C" int cpn_dwconv3x3_tokens(const float* x, const float* w, const float* bias, int B, int H, int W, int C,
                                    int flip, float* y, void* stream) {
    CPN_REQUIRE(x && w && y, CPN_E_ARG, "cpn_dwconv3x3_tokens: null pointer");
    CPN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, CPN_E_SHAPE, "cpn_dwconv3x3_tokens: need C %% 4 == 0");
    CPN_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0, CPN_E_ARG,
                "cpn_dwconv3x3_tokens: pointers must be 16-byte aligned");
    const long long total = (long long)B * H * ((W + DW_RUN - 1) / DW_RUN) * (C / 4);
    CPN_REQUIRE(total / 256 < (1LL << 31), CPN_E_SHAPE, "cpn_dwconv3x3_tokens: too large");
    hipLaunchKernelGGL(dwconv3x3_tokens_kernel, dim3((unsigned)cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, H, W, C / 4, total, flip, y);
    CPN_LAUNCH_CHECK("cpn_dwconv3x3_tokens");
    return 0;
}

extern "C" long long cpn_dwconv3x3_tokens_wgrad_scratch(int B, int H, int C) { return (long long)B * H * C * 10; }

extern "C" int cpn_dwconv3x3_tokens_wgrad(const float* x, const float* dy, int B, int H, int W, int C, float* partial,
                                          float* dw, float* db, void* stream) {
    CPN_REQUIRE(x && dy && dw && partial, CPN_E_ARG, "cpn_dwconv3x3_tokens_wgrad: null pointer");
    CPN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, CPN_E_SHAPE, "cpn_dwconv3x3_tokens_wgrad: bad shape");
    const long long rows = (long long)B * H;
    dim3 grid((unsigned)rows, (unsigned)cpn_cdiv(C, 256));
    hipLaunchKernelGGL(dwconv3x3_tokens_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, dy, H, W, C, partial);
    hipLaunchKernelGGL(dwconv3x3_tokens_wgrad_reduce_kernel, dim3((unsigned)cpn_cdiv(C * 10, 64)), dim3(1024), 0,
                       (hipStream_t)stream, partial, rows, C, dw, db);
    CPN_LAUNCH_CHECK("cpn_dwconv3x3_tokens_wgrad");
    return 0;
}

extern "C" int cpn_dual_softmax(const float* a, int B, int L, int M, float* rstat, float* cstat, float* f,
                                void* stream) {
    CPN_REQUIRE(a && rstat && cstat && f, CPN_E_ARG, "cpn_dual_softmax: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && L > 0 && M > 0, CPN_E_SHAPE, "cpn_dual_softmax: bad shape");
    const hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * L, total = rows * M;
    hipLaunchKernelGGL(row_stats_kernel, dim3(cpn_cdiv(rows, 4)), dim3(256), 0, st, a, (const float*)nullptr, rows, M, rstat, 0);
    hipLaunchKernelGGL(col_stats_kernel, dim3(cpn_cdiv(M, CR_COLS), B), dim3(CR_COLS * CR_GROUPS), 0, st, a, (const float*)nullptr,
                       L, M, cstat, 0);
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(total, 256), 16384);
    hipLaunchKernelGGL(dual_softmax_apply_kernel, dim3(blocks), dim3(256), 0, st, a, rstat, cstat, L, M, total, f);
    CPN_LAUNCH_CHECK("cpn_dual_softmax");
    return 0;
}

extern "C" int cpn_dual_softmax_bwd(const float* a, const float* rstat, const float* cstat, const float* f,
                                    const float* df, int B, int L, int M, float* srow, float* scol, float* da,
                                    void* stream) {
    CPN_REQUIRE(a && rstat && cstat && f && df && srow && scol && da, CPN_E_ARG, "cpn_dual_softmax_bwd: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && L > 0 && M > 0, CPN_E_SHAPE, "cpn_dual_softmax_bwd: bad shape");
    const hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * L, total = rows * M;
    hipLaunchKernelGGL(row_stats_kernel, dim3(cpn_cdiv(rows, 4)), dim3(256), 0, st, f, df, rows, M, srow, 1);
    hipLaunchKernelGGL(col_stats_kernel, dim3(cpn_cdiv(M, CR_COLS), B), dim3(CR_COLS * CR_GROUPS), 0, st, f, df, L, M, scol, 1);
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(total, 256), 16384);
    hipLaunchKernelGGL(dual_softmax_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, a, rstat, cstat, df, srow, scol, L, M,
                       total, da);
    CPN_LAUNCH_CHECK("cpn_dual_softmax_bwd");
    return 0;
}

extern "C" int cpn_correlation(const float* src, const float* trg, int B, int L, int C, float eps, float* src_n,
                               float* trg_n, float* out, void* stream) {
    CPN_REQUIRE(src && trg && src_n && trg_n && out, CPN_E_ARG, "cpn_correlation: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && L > 0 && C > 0 && (C % 16) == 0, CPN_E_SHAPE,
                "cpn_correlation: C=%d must be a multiple of 16", C);
    const hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * L;
    if (trg == src + rows * C && trg_n == src_n + rows * C) {
        // [src ; trg] and [src_n ; trg_n] are halves of one buffer each (the batched views of UFCLayer): one launch
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cpn_cdiv(2 * rows, 4)), dim3(256), 0, st, src, src_n, 2 * rows, C, eps);
    } else {
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cpn_cdiv(rows, 4)), dim3(256), 0, st, src, src_n, rows, C, eps);
        hipLaunchKernelGGL(l2norm_rows_kernel, dim3(cpn_cdiv(rows, 4)), dim3(256), 0, st, trg, trg_n, rows, C, eps);
    }
    CPN_LAUNCH_CHECK("cpn_correlation(normalise)");
    if (L % 128 == 0 && (long long)B * (L / 128) * (L / 128) >= 512 && ((uintptr_t)out % 16) == 0) {
        dim3 grid(L / 128, L / 128, B);
        hipLaunchKernelGGL(gemm_nt_f32_tile64_kernel, grid, dim3(256), 0, st, src_n, trg_n, out, L, L, C);
    } else if ((long long)B * cpn_cdiv(L, 128) * cpn_cdiv(L, 64) >= 512) {
        dim3 grid(cpn_cdiv(L, 128), cpn_cdiv(L, 64), B);
        hipLaunchKernelGGL(gemm_nt_f32_kernel<8>, grid, dim3(256), 0, st, src_n, trg_n, out, L, L, C);
    } else {
        dim3 grid(cpn_cdiv(L, 32), cpn_cdiv(L, 64), B);
        hipLaunchKernelGGL(gemm_nt_f32_kernel<2>, grid, dim3(256), 0, st, src_n, trg_n, out, L, L, C);
    }
    CPN_LAUNCH_CHECK("cpn_correlation(gemm)");
    return 0;
}

extern "C" int cpn_l2norm_rows_bwd(const float* x, const float* y, const float* dy, long long rows, int C, float eps,
                                   float* dx, void* stream) {
    CPN_REQUIRE(x && y && dy && dx, CPN_E_ARG, "cpn_l2norm_rows_bwd: null pointer");
    CPN_REQUIRE(rows > 0 && rows < (1LL << 33) && C > 0 && C <= 1024, CPN_E_SHAPE, "cpn_l2norm_rows_bwd: need 0 < C <= 1024 (got %d)", C);
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(cpn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, rows,
                       C, eps);
    CPN_LAUNCH_CHECK("cpn_l2norm_rows_bwd");
    return 0;
}

extern "C" int cpn_soft_argmax_pair(const float* c, int B, int h, float beta, float* t_to_s, float* s_to_t,
                                    void* stream) {
    CPN_REQUIRE(c && t_to_s && s_to_t, CPN_E_ARG, "cpn_soft_argmax_pair: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && h > 1 && beta > 0.f, CPN_E_SHAPE, "cpn_soft_argmax_pair: bad shape");
    const hipStream_t st = (hipStream_t)stream;
    const int T = h * h;
    hipLaunchKernelGGL(soft_argmax_rows_kernel, dim3(T, B), dim3(256), 0, st, c, h, beta, t_to_s);
    hipLaunchKernelGGL(soft_argmax_cols_kernel, dim3(cpn_cdiv(T, CR_COLS), B), dim3(CR_COLS * CR_GROUPS), 0, st, c, h, beta,
                       s_to_t);
    CPN_LAUNCH_CHECK("cpn_soft_argmax_pair");
    return 0;
}

extern "C" int cpn_soft_argmax_pair_bwd(const float* c, int B, int h, float beta, const float* t_to_s,
                                        const float* s_to_t, const float* g_t_to_s, const float* g_s_to_t, float* dc,
                                        void* stream) {
    CPN_REQUIRE(c && t_to_s && s_to_t && g_t_to_s && g_s_to_t && dc, CPN_E_ARG, "cpn_soft_argmax_pair_bwd: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && h > 1 && beta > 0.f, CPN_E_SHAPE, "cpn_soft_argmax_pair_bwd: bad shape");
    const hipStream_t st = (hipStream_t)stream;
    const int T = h * h;
    hipLaunchKernelGGL(soft_argmax_rows_bwd_kernel, dim3(T, B), dim3(256), 0, st, c, h, beta, t_to_s, g_t_to_s, dc);
    hipLaunchKernelGGL(soft_argmax_cols_bwd_kernel, dim3(cpn_cdiv(T, 64), B), dim3(256), 0, st, c, h, beta, s_to_t,
                       g_s_to_t, dc);
    CPN_LAUNCH_CHECK("cpn_soft_argmax_pair_bwd");
    return 0;
}
