// f1 (inference): the launch-bound ends of the pose head (/root/reference models/backbone.py:209-278 positional encodings,
// models/CoPoNeRF.py:106-126,190-206 regressors + 6-D rotation + pose assembly).  At one stereo pair these are ~70 launches of
// 2-5 us kernels on a few hundred values each (elementwise ops of the closed-form K^-1 grid, three Linear+ReLU chains on ONE
// row, normalise / cross / cat of nine numbers); two kernels replace them.  fp32, the reference's operation order where the
// result depends on it (the dot products sum lane-strided partials, not in index order: 1e-7 relative).
#include "common.h"

namespace {

// (x^2, y^2, xy, x, y, 1) of the K^-1-normalised grid, index = col * n + row (getz.positional_encodings)
__global__ void pose_positional_kernel(const float* __restrict__ K, int V, float H, const float* __restrict__ lin, int n,
                                       float* __restrict__ out) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const float* Kb = K + (size_t)b * V * 16;                  // view 0 of sample b
    const float fx = Kb[0] / H, fy = Kb[5] / H, cx = Kb[2] / H, cy = Kb[6] / H;
    const float hp = cy * 2.0f, wp = cx * 2.0f;
    const float a = (fx / wp) * 2.0f, bb = (fy / hp) * 2.0f;
    const float c = (cx / wp) * 2.0f - 1.0f, d = (cy / hp) * 2.0f - 1.0f;
    const int k = idx / n, j = idx - k * n;
    const float p4 = (lin[k] - c) / a, p3 = (lin[j] - d) / bb;
    float* o = out + ((size_t)b * n * n + idx) * 6;
    o[0] = p3 * p3; o[1] = p4 * p4; o[2] = p3 * p4; o[3] = p3; o[4] = p4; o[5] = 1.0f;
}

// y[o] = sum_k relu(x[k]) W[o][k] + b[o]: a wave takes FOUR outputs at a time (their weight rows are requested together: one
// workgroup walks the layers alone, so the memory latency of a row is what a layer costs), lanes stride over k
template <int N_IN>
__device__ __forceinline__ void dense_relu_in(const float* x, const float* __restrict__ W, const float* __restrict__ b,
                                              int n_out, float* y, int wave, int lane, int nwaves) {
    constexpr int PER = (N_IN + 63) / 64;                         // k values per lane
    float xv[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int k = lane + 64 * i;
        xv[i] = k < N_IN ? fmaxf(x[k], 0.0f) : 0.0f;
    }
    for (int o0 = wave * 4; o0 < n_out; o0 += nwaves * 4) {
        float wv[4][PER];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = o0 + q < n_out ? o0 + q : n_out - 1;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int k = lane + 64 * i;
                wv[q][i] = k < N_IN ? W[(size_t)o * N_IN + k] : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < PER; ++i) s += xv[i] * wv[q][i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0 && o0 + q < n_out) y[o0 + q] = s + b[o0 + q];
        }
    }
}

// First Linear of the pose regressor at a handful of rows: y[b][o] = <W[o], x[b]> over K = 134 144 inputs, 275 MB of fp32
// weights read ONCE for all rows (the library's kernel for this M = 1 problem streams them at 1.8 TB/s: 154 us).  A workgroup
// takes one output row and one of `nsplit` chunks of K, 16-byte loads, and writes one partial per (row of x, output, chunk);
// pose_tail_kernel sums the chunks in chunk order and adds the bias.
template <int NB>
__global__ __launch_bounds__(256) void pose_gemv_kernel(const float* __restrict__ x, const float* __restrict__ W, int K, int O,
                                                        int nsplit, float* __restrict__ partial) {
    const int o = blockIdx.x, c = blockIdx.y;
    const int k4 = K / 4, per = (k4 + nsplit - 1) / nsplit;
    const int lo = c * per, hi = lo + per < k4 ? lo + per : k4;
    const f32x4* w4 = reinterpret_cast<const f32x4*>(W + (size_t)o * K);
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        const f32x4 w = w4[i];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)b * K)[i];
            acc[b] += (w[0] * v[0] + w[1] * v[1]) + (w[2] * v[2] + w[3] * v[3]);
        }
    }
    __shared__ float red[NB][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float v = acc[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[b][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        const int b = threadIdx.x;
        partial[((size_t)b * O + o) * nsplit + c] = (red[b][0] + red[b][1]) + (red[b][2] + red[b][3]);
    }
}

struct PoseTailW {
    const float *w2, *b2, *w3, *b3;                // pose_regressor[2] (256 x 512), [4] (256 x 256)
    const float *r1, *rb1, *r2, *rb2, *r3, *rb3;   // rotation_regressor Linear (64 x 128), (32 x 64), (6 x 32)
    const float *t1, *tb1, *t2, *tb2, *t3, *tb3;   // translation_regressor Linear (64 x 128), (32 x 64), (3 x 32)
};

__global__ __launch_bounds__(1024) void pose_tail_kernel(const float* __restrict__ h512, int nsplit, const float* __restrict__ bias0,
                                                         PoseTailW w, float* __restrict__ rel_pose) {
    __shared__ float s0[512], s1[256], s2[256], sr[64], st[64], sr2[32], st2[32], o9[16];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 512; i += 1024) {
        if (nsplit > 0) {                                     // chunk partials of cpn_pose_gemv, summed in chunk order, + bias
            const float* pp = h512 + ((size_t)b * 512 + i) * nsplit;
            float v = pp[0];
            for (int c = 1; c < nsplit; ++c) v += pp[c];
            s0[i] = v + bias0[i];
        } else {
            s0[i] = h512[(size_t)b * 512 + i];
        }
    }
    __syncthreads();
    dense_relu_in<512>(s0, w.w2, w.b2, 256, s1, wave, lane, 16);
    __syncthreads();
    dense_relu_in<256>(s1, w.w3, w.b3, 256, s2, wave, lane, 16);
    __syncthreads();
    // lat = relu(s2)[:128]; both regressors start with a ReLU of their own (idempotent); waves 0-7 rotation, 8-15 translation
    if (wave < 8) dense_relu_in<128>(s2, w.r1, w.rb1, 64, sr, wave, lane, 8);
    else dense_relu_in<128>(s2, w.t1, w.tb1, 64, st, wave - 8, lane, 8);
    __syncthreads();
    if (wave < 8) dense_relu_in<64>(sr, w.r2, w.rb2, 32, sr2, wave, lane, 8);
    else dense_relu_in<64>(st, w.t2, w.tb2, 32, st2, wave - 8, lane, 8);
    __syncthreads();
    if (wave < 2) dense_relu_in<32>(sr2, w.r3, w.rb3, 6, o9, wave, lane, 2);
    else if (wave == 8) dense_relu_in<32>(st2, w.t3, w.tb3, 3, o9 + 8, 0, lane, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        // Zhou et al. 6-D rotation -> rows b1, b2, b1 x b2 (F.normalize: x / max(|x|, 1e-12))
        const float a1[3] = {o9[0], o9[1], o9[2]}, a2[3] = {o9[3], o9[4], o9[5]};
        const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
        const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
        const float dt = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
        const float u[3] = {a2[0] - dt * b1[0], a2[1] - dt * b1[1], a2[2] - dt * b1[2]};
        const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
        const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
        const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
        float* o = rel_pose + (size_t)b * 16;
        o[0] = b1[0]; o[1] = b1[1]; o[2] = b1[2]; o[3] = o9[8];
        o[4] = b2[0]; o[5] = b2[1]; o[6] = b2[2]; o[7] = o9[9];
        o[8] = b3[0]; o[9] = b3[1]; o[10] = b3[2]; o[11] = o9[10];
        o[12] = 0.0f; o[13] = 0.0f; o[14] = 0.0f; o[15] = 1.0f;
    }
}

}  // namespace

extern "C" int cpn_pose_positional(const float* intrinsics, int B, int V, float H, const float* lin, int n, float* out,
                                   void* stream) {
    CPN_REQUIRE(intrinsics && lin && out && B > 0 && B < 65536 && V > 0 && n > 0 && H > 0.f, 1, "cpn_pose_positional: bad arguments");
    hipLaunchKernelGGL(pose_positional_kernel, dim3(cpn_cdiv((long long)n * n, 256), B), dim3(256), 0, (hipStream_t)stream,
                       intrinsics, V, H, lin, n, out);
    CPN_LAUNCH_CHECK("cpn_pose_positional");
    return 0;
}

extern "C" int cpn_pose_gemv(const float* x, const float* W, int B, int K, int O, int nsplit, float* partial, void* stream) {
    CPN_REQUIRE(x && W && partial && B >= 1 && B <= 4 && K > 0 && K % 4 == 0 && O > 0 && nsplit >= 1 && nsplit <= 64, 1,
                "cpn_pose_gemv: 1..4 rows, K %% 4 == 0, 1..64 chunks (got B=%d K=%d nsplit=%d)", B, K, nsplit);
    CPN_REQUIRE(((uintptr_t)x | (uintptr_t)W) % 16 == 0, 1, "cpn_pose_gemv: 16-byte alignment");
    dim3 grid((unsigned)O, (unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
    switch (B) {
        case 1: hipLaunchKernelGGL(pose_gemv_kernel<1>, grid, dim3(256), 0, st, x, W, K, O, nsplit, partial); break;
        case 2: hipLaunchKernelGGL(pose_gemv_kernel<2>, grid, dim3(256), 0, st, x, W, K, O, nsplit, partial); break;
        case 3: hipLaunchKernelGGL(pose_gemv_kernel<3>, grid, dim3(256), 0, st, x, W, K, O, nsplit, partial); break;
        default: hipLaunchKernelGGL(pose_gemv_kernel<4>, grid, dim3(256), 0, st, x, W, K, O, nsplit, partial); break;
    }
    CPN_LAUNCH_CHECK("cpn_pose_gemv");
    return 0;
}

extern "C" int cpn_pose_tail(const float* h512, int nsplit, const float* bias0, const float* const* weights, int B,
                             float* rel_pose, void* stream) {
    CPN_REQUIRE(h512 && weights && rel_pose && B > 0 && nsplit >= 0 && (nsplit == 0 || bias0), 1, "cpn_pose_tail: bad arguments");
    PoseTailW w;
    const float** dst = reinterpret_cast<const float**>(&w);
    for (int i = 0; i < CPN_POSE_TAIL_TENSORS; ++i) {
        CPN_REQUIRE(weights[i] != nullptr, 1, "cpn_pose_tail: weight %d is null", i);
        dst[i] = weights[i];
    }
    hipLaunchKernelGGL(pose_tail_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, h512, nsplit, bias0, w, rel_pose);
    CPN_LAUNCH_CHECK("cpn_pose_tail");
    return 0;
}
