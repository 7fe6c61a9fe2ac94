// VJP of the strided Conv4d layers (models/conv4d.py:57-135 with stride > 1: k3 s2 p1 on 32^4, k5 s4 p2 on 64^4 —
// Encoder4D((1, nhead), k, s, p) of UFC.embedding / UFCLayer.feat_to_corr1,2, aggregation.py:198-199, 369-371).
//
// Forward (csrc/ufc.hip): the two max-pooled volumes
//   Ps[b,c,Y,X,sy',sx'] = max_{dy,dx<s} x[b,c,Y,X,sy'*s+dy,sx'*s+dx]     (query branch: support dims pooled, ceil mode)
//   Pq[b,c,qy',qx',U,V] = max_{dy,dx<s} x[b,c,qy'*s+dy,qx'*s+dx,U,V]     (support branch: query dims pooled)
// and   y[b,o,qy,qx,sy,sx] = bq[o] + bs[o] + sum_{c,i,j} wq[o,c,i,j] Ps[b,c,qy*s+i-p,qx*s+j-p,sy,sx]
//                                              + ws[o,c,i,j] Pq[b,c,qy,qx,sy*s+i-p,sx*s+j-p].
// The reference differentiates this through autograd (two max_pool2d + two conv2d + ~10 permute copies: 0.7 / 2.6 ms per
// layer at 4 pairs, 11 ms per training step over the 8 strided layers).  Here, given dy:
//   1. the pooled volumes again, with the window position of each maximum (first maximum in scan order, NaN wins:
//      the routing max_pool2d's backward uses)                                                   2 launches
//   2. gPs, gPq: the transposed strided convolutions of dy (at most ceil(k/s)^2 taps per element)  1 launch
//   3. dx[b,c,Y,X,U,V] = [(U,V) is the maximum of its support window] gPs + [(Y,X) is the maximum of its query window] gPq
//      — one coalesced pass over the volume, no atomics                                           1 launch
//   4. weight / bias gradients: per (tap, branch) block sums over position chunks, then a fixed-order sum of the
//      chunk partials (deterministic)                                                              2 launches
#include <algorithm>

#include "common.h"

namespace {

constexpr int WG_CHUNK = 4096;             // positions per weight-gradient workgroup (16 per thread)

struct SGeo {
    int B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, Oq, Pq, Os, Ps;
    __host__ __device__ long long n_ps() const { return (long long)B * Cin * Hq * Wq * Os * Ps; }
    __host__ __device__ long long n_pq() const { return (long long)B * Cin * Oq * Pq * Hs * Ws; }
    __host__ __device__ long long npos() const { return (long long)Oq * Pq * Os * Ps; }
};

__device__ __forceinline__ bool takes_max(float v, float m) { return v > m || v != v; }

__global__ __launch_bounds__(256) void pool_support_arg_kernel(const float* __restrict__ x, SGeo g, float* __restrict__ out,
                                                               unsigned char* __restrict__ arg) {
    const unsigned total = (unsigned)g.n_ps();               // all element counts < 2^31 (checked by the launcher)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int sx = (int)(i % (unsigned)g.Ps);
        unsigned t = i / (unsigned)g.Ps;
        const int sy = (int)(t % (unsigned)g.Os);
        const unsigned pl = t / (unsigned)g.Os;              // (b, c, Y, X)
        const float* base = x + (size_t)pl * g.Hs * g.Ws;
        float m = -INFINITY;
        int a = 0;
        for (int dy = 0; dy < g.s; ++dy)
            for (int dx = 0; dx < g.s; ++dx) {
                const int yy = sy * g.s + dy, xx = sx * g.s + dx;
                if (yy < g.Hs && xx < g.Ws) {
                    const float v = base[(size_t)yy * g.Ws + xx];
                    if (takes_max(v, m)) { m = v; a = dy * g.s + dx; }
                }
            }
        out[i] = m;
        arg[i] = (unsigned char)a;
    }
}

__global__ __launch_bounds__(256) void pool_query_arg_kernel(const float* __restrict__ x, SGeo g, float* __restrict__ out,
                                                             unsigned char* __restrict__ arg) {
    const unsigned P = (unsigned)(g.Hs * g.Ws), total = (unsigned)g.n_pq();
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned uv = i % P;
        unsigned t = i / P;
        const int qx = (int)(t % (unsigned)g.Pq); t /= (unsigned)g.Pq;
        const int qy = (int)(t % (unsigned)g.Oq);
        const unsigned bc = t / (unsigned)g.Oq;
        const float* base = x + (size_t)bc * g.Hq * g.Wq * P + uv;
        float m = -INFINITY;
        int a = 0;
        for (int dy = 0; dy < g.s; ++dy)
            for (int dx = 0; dx < g.s; ++dx) {
                const int yy = qy * g.s + dy, xx = qx * g.s + dx;
                if (yy < g.Hq && xx < g.Wq) {
                    const float v = base[((size_t)yy * g.Wq + xx) * P];
                    if (takes_max(v, m)) { m = v; a = dy * g.s + dx; }
                }
            }
        out[i] = m;
        arg[i] = (unsigned char)a;
    }
}

// gPs (n_ps elements) followed by gPq (n_pq elements): one thread per element
__global__ __launch_bounds__(256) void dpool_kernel(const float* __restrict__ dy, const float* __restrict__ wq,
                                                    const float* __restrict__ ws, SGeo g, float* __restrict__ gps,
                                                    float* __restrict__ gpq) {
    const unsigned nps = (unsigned)g.n_ps(), total = nps + (unsigned)g.n_pq();
    const long long npos = g.npos();
    const int kk = g.k * g.k;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        float acc = 0.0f;
        if (e < nps) {
            unsigned t = e;
            const int sx = (int)(t % (unsigned)g.Ps); t /= (unsigned)g.Ps;
            const int sy = (int)(t % (unsigned)g.Os); t /= (unsigned)g.Os;
            const int X = (int)(t % (unsigned)g.Wq); t /= (unsigned)g.Wq;
            const int Y = (int)(t % (unsigned)g.Hq); t /= (unsigned)g.Hq;
            const int c = (int)(t % (unsigned)g.Cin);
            const int b = (int)(t / (unsigned)g.Cin);
            // taps with (Y + p - i) a non-negative multiple of s: i = ry, ry + s, ... ; qy counts down from (Y + p) / s
            const int qy0 = (Y + g.p) / g.s, ry = (Y + g.p) - qy0 * g.s, qx0 = (X + g.p) / g.s, rx = (X + g.p) - qx0 * g.s;
            for (int i = ry, qy = qy0; i < g.k && qy >= 0; i += g.s, --qy) {
                if (qy >= g.Oq) continue;
                for (int j = rx, qx = qx0; j < g.k && qx >= 0; j += g.s, --qx) {
                    if (qx >= g.Pq) continue;
                    const long long pos = (((long long)qy * g.Pq + qx) * g.Os + sy) * g.Ps + sx;
                    for (int o = 0; o < g.Cout; ++o)
                        acc += wq[((size_t)o * g.Cin + c) * kk + i * g.k + j] * dy[((size_t)b * g.Cout + o) * npos + pos];
                }
            }
            gps[e] = acc;
        } else {
            unsigned t = e - nps;
            const int V = (int)(t % (unsigned)g.Ws); t /= (unsigned)g.Ws;
            const int U = (int)(t % (unsigned)g.Hs); t /= (unsigned)g.Hs;
            const int qx = (int)(t % (unsigned)g.Pq); t /= (unsigned)g.Pq;
            const int qy = (int)(t % (unsigned)g.Oq); t /= (unsigned)g.Oq;
            const int c = (int)(t % (unsigned)g.Cin);
            const int b = (int)(t / (unsigned)g.Cin);
            const int sy0 = (U + g.p) / g.s, ru = (U + g.p) - sy0 * g.s, sx0 = (V + g.p) / g.s, rv = (V + g.p) - sx0 * g.s;
            for (int i = ru, sy = sy0; i < g.k && sy >= 0; i += g.s, --sy) {
                if (sy >= g.Os) continue;
                for (int j = rv, sx = sx0; j < g.k && sx >= 0; j += g.s, --sx) {
                    if (sx >= g.Ps) continue;
                    const long long pos = (((long long)qy * g.Pq + qx) * g.Os + sy) * g.Ps + sx;
                    for (int o = 0; o < g.Cout; ++o)
                        acc += ws[((size_t)o * g.Cin + c) * kk + i * g.k + j] * dy[((size_t)b * g.Cout + o) * npos + pos];
                }
            }
            gpq[e - nps] = acc;
        }
    }
}

// dx: one workgroup per (b, c, Y, X) plane of the input, threads over its (U, V) elements — the plane's indices are
// wave-uniform, an element costs one 32-bit division; routed by the stored window positions of the two maxima
__global__ __launch_bounds__(256) void route_dx_kernel(const float* __restrict__ gps, const float* __restrict__ gpq,
                                                       const unsigned char* __restrict__ args,
                                                       const unsigned char* __restrict__ argq, SGeo g,
                                                       float* __restrict__ dx) {
    const unsigned P = (unsigned)(g.Hs * g.Ws), nplanes = (unsigned)(g.B * g.Cin * g.Hq * g.Wq);
    for (unsigned pl = blockIdx.x; pl < nplanes; pl += gridDim.x) {
        const unsigned X = pl % (unsigned)g.Wq, t1 = pl / (unsigned)g.Wq;
        const unsigned Y = t1 % (unsigned)g.Hq, bc = t1 / (unsigned)g.Hq;
        const size_t s_base = (size_t)pl * g.Os * g.Ps;                                   // gPs / args plane of (bc, Y, X)
        const size_t q_base = (((size_t)bc * g.Oq + Y / g.s) * g.Pq + X / g.s) * P;        // gPq / argq plane
        const int q_pos = (int)((Y % g.s) * g.s + X % g.s);
        float* out = dx + (size_t)pl * P;
        for (unsigned uv = threadIdx.x; uv < P; uv += 256) {
            const unsigned U = uv / (unsigned)g.Ws, V = uv - U * (unsigned)g.Ws;
            const size_t is = s_base + (size_t)(U / g.s) * g.Ps + V / g.s;
            float v = 0.0f;
            if (args[is] == (int)((U % g.s) * g.s + V % g.s)) v += gps[is];
            if (argq[q_base + uv] == q_pos) v += gpq[q_base + uv];
            out[uv] = v;
        }
    }
}

// weight gradients: blockIdx.y = tap (branch * k*k + i*k + j), blockIdx.x = chunk of WG_CHUNK positions of the (b, pos)
// index space; a thread keeps COUT*CIN partial sums (compile-time counts: runtime-indexed accumulators would live in
// scratch memory); block sums go to part[chunk][tap][o][c].  The bias gradient (sum of dy) rides on tap 0.
template <int COUT, int CIN>
__global__ __launch_bounds__(256) void wgrad_strided_kernel(const float* __restrict__ dy, const float* __restrict__ psv,
                                                            const float* __restrict__ pqv, SGeo g, float* __restrict__ part,
                                                            float* __restrict__ partb) {
    constexpr int NA = COUT * CIN;
    const int tap = blockIdx.y, kk = g.k * g.k;
    const int br = tap / kk, i = (tap % kk) / g.k, j = tap % g.k;
    const unsigned npos = (unsigned)g.npos(), total = (unsigned)g.B * npos;
    const unsigned p0 = blockIdx.x * WG_CHUNK;
    const float* src = br == 0 ? psv : pqv;
    const unsigned cstride = br == 0 ? (unsigned)(g.Hq * g.Wq * g.Os * g.Ps) : (unsigned)(g.Oq * g.Pq * g.Hs * g.Ws);
    float acc[NA], accb[COUT];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = 0.0f;
#pragma unroll
    for (int a = 0; a < COUT; ++a) accb[a] = 0.0f;
    for (unsigned q = p0 + threadIdx.x; q < p0 + WG_CHUNK && q < total; q += 256) {
        const unsigned b = q / npos, pos = q - b * npos;
        const int sx = (int)(pos % (unsigned)g.Ps);
        unsigned t = pos / (unsigned)g.Ps;
        const int sy = (int)(t % (unsigned)g.Os); t /= (unsigned)g.Os;
        const int qx = (int)(t % (unsigned)g.Pq);
        const int qy = (int)(t / (unsigned)g.Pq);
        bool ok;
        size_t off;
        if (br == 0) {
            const int Y = qy * g.s + i - g.p, X = qx * g.s + j - g.p;
            ok = Y >= 0 && Y < g.Hq && X >= 0 && X < g.Wq;
            off = (size_t)b * CIN * cstride + (((size_t)Y * g.Wq + X) * g.Os + sy) * g.Ps + sx;
        } else {
            const int U = sy * g.s + i - g.p, V = sx * g.s + j - g.p;
            ok = U >= 0 && U < g.Hs && V >= 0 && V < g.Ws;
            off = (size_t)b * CIN * cstride + (((size_t)qy * g.Pq + qx) * g.Hs + U) * g.Ws + V;
        }
        float v[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[c] = ok ? src[off + (size_t)c * cstride] : 0.0f;
        const float* dp = dy + (size_t)b * COUT * npos + pos;
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float d = dp[(size_t)o * npos];
            accb[o] += d;
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[o * CIN + c] += d * v[c];
        }
    }
    __shared__ float red[4][NA + COUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < NA + COUT; ++a) {
        float r = a < NA ? acc[a < NA ? a : 0] : accb[a >= NA ? a - NA : 0];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) r += __shfl_xor(r, o2);
        if (lane == 0) red[wave][a] = r;
    }
    __syncthreads();
    const int ntap = gridDim.y;
    if (threadIdx.x < NA)
        part[((size_t)blockIdx.x * ntap + tap) * NA + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (tap == 0 && threadIdx.x < COUT) {
        const int a = NA + threadIdx.x;
        partb[(size_t)blockIdx.x * COUT + threadIdx.x] = (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]);
    }
}

// gwq / gws (Cout, Cin, k, k) and gb (Cout) from the chunk partials, summed in chunk order
__global__ __launch_bounds__(1024) void wgrad_strided_reduce_kernel(const float* __restrict__ part,
                                                                    const float* __restrict__ partb, int nchunk, SGeo g,
                                                                    float* __restrict__ gwq, float* __restrict__ gws,
                                                                    float* __restrict__ gb) {
    // 64 outputs per workgroup, the chunks split over 16 waves (fixed combination order); one thread per output walking all
    // chunks on five workgroups took 21 us per call
    __shared__ float sh[16][64];
    const int kk = g.k * g.k, na = g.Cout * g.Cin, ntap = 2 * kk;
    const int nw = ntap * na;
    const int j = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + j;
    float v = 0.0f;
    if (e < nw) {
        const int tap = e / na, a = e % na;
        for (int ch = p; ch < nchunk; ch += 16) v += part[((size_t)ch * ntap + tap) * na + a];
    } else if (e < nw + g.Cout) {
        for (int ch = p; ch < nchunk; ch += 16) v += partb[(size_t)ch * g.Cout + (e - nw)];
    }
    sh[p][j] = v;
    __syncthreads();
    if (p == 0 && e < nw + g.Cout) {
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sh[q][j];
        if (e < nw) {
            const int tap = e / na, a = e % na;
            const int br = tap / kk, ij = tap % kk;
            (br == 0 ? gwq : gws)[(size_t)a * kk + ij] = acc;        // a = o * Cin + c
        } else {
            gb[e - nw] = acc;
        }
    }
}

bool make_geo(int B, int Cin, int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s, int p, SGeo* g) {
    auto co = [&](int n) { return (n + 2 * p - k) / s + 1; };
    auto po = [&](int n) { return (n + s - 1) / s; };
    *g = SGeo{B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, co(Hq), co(Wq), co(Hs), co(Ws)};
    return g->Oq == po(Hq) && g->Pq == po(Wq) && g->Os == po(Hs) && g->Ps == po(Ws);
}

long long nchunks(const SGeo& g) { return ((long long)g.B * g.npos() + WG_CHUNK - 1) / WG_CHUNK; }

}  // namespace

// floats of scratch: Ps | Pq | gPs | gPq | window positions of the maxima (bytes) | weight / bias gradient partials
extern "C" long long cpn_conv4d_strided_bwd_scratch(int B, int Cin, int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s,
                                                    int p) {
    SGeo g;
    if (s <= 1 || !make_geo(B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, &g)) return 0;
    const long long n = g.n_ps() + g.n_pq();
    return 2 * n + (n + 3) / 4 + nchunks(g) * (2LL * k * k * Cout * Cin + Cout);
}

extern "C" int cpn_conv4d_strided_bwd(const float* x, const float* dy, const float* wq, const float* ws, int B, int Cin,
                                      int Cout, int Hq, int Wq, int Hs, int Ws, int k, int s, int p, float* scratch,
                                      float* dx, float* gwq, float* gws, float* gb, void* stream) {
    CPN_REQUIRE(x && dy && wq && ws && scratch, CPN_E_ARG, "cpn_conv4d_strided_bwd: null pointer");
    CPN_REQUIRE((gwq && gws && gb) || (!gwq && !gws && !gb), CPN_E_ARG,
                "cpn_conv4d_strided_bwd: weight and bias gradients are produced together");
    CPN_REQUIRE(B > 0 && Cout == 8 && (Cin == 1 || Cin == 2 || Cin == 8) && s > 1 && s * s <= 255 && k > 0 && k <= 7 && p >= 0,
                CPN_E_SHAPE, "cpn_conv4d_strided_bwd: compiled for Cout = 8, Cin in {1, 2, 8}, s > 1, k <= 7 (got Cin=%d Cout=%d k=%d s=%d)",
                Cin, Cout, k, s);
    SGeo g;
    CPN_REQUIRE(make_geo(B, Cin, Cout, Hq, Wq, Hs, Ws, k, s, p, &g), CPN_E_SHAPE,
                "cpn_conv4d_strided_bwd: conv output and pooled size of the two branches disagree");
    const hipStream_t st = (hipStream_t)stream;
    const long long nps = g.n_ps(), npq = g.n_pq(), n = nps + npq;
    CPN_REQUIRE(n < (1LL << 31) && (long long)B * Cin * Hq * Wq * Hs * Ws < (1LL << 31), CPN_E_SHAPE,
                "cpn_conv4d_strided_bwd: volume too large for 32-bit indexing");
    float* psv = scratch;
    float* pqv = psv + nps;
    float* gps = pqv + npq;
    float* gpq = gps + nps;
    unsigned char* args = reinterpret_cast<unsigned char*>(gpq + npq);
    unsigned char* argq = args + nps;
    float* part = gpq + npq + (n + 3) / 4;
    const long long nch = nchunks(g);
    float* partb = part + nch * 2LL * k * k * Cout * Cin;
    auto blocks = [](long long total) { return dim3((unsigned)std::min<long long>(cpn_cdiv(total, 256), 1 << 18)); };
    hipLaunchKernelGGL(pool_support_arg_kernel, blocks(nps), dim3(256), 0, st, x, g, psv, args);
    hipLaunchKernelGGL(pool_query_arg_kernel, blocks(npq), dim3(256), 0, st, x, g, pqv, argq);
    CPN_LAUNCH_CHECK("cpn_conv4d_strided_bwd(pool)");
    if (dx) {
        hipLaunchKernelGGL(dpool_kernel, blocks(n), dim3(256), 0, st, dy, wq, ws, g, gps, gpq);
        hipLaunchKernelGGL(route_dx_kernel, dim3((unsigned)std::min<long long>((long long)B * Cin * Hq * Wq, 1 << 20)), dim3(256), 0,
                           st, gps, gpq, args, argq, g, dx);
        CPN_LAUNCH_CHECK("cpn_conv4d_strided_bwd(dx)");
    }
    if (gwq) {
        dim3 grid((unsigned)nch, 2 * k * k);
        if (Cout == 8 && Cin == 1)
            hipLaunchKernelGGL((wgrad_strided_kernel<8, 1>), grid, dim3(256), 0, st, dy, psv, pqv, g, part, partb);
        else if (Cout == 8 && Cin == 2)
            hipLaunchKernelGGL((wgrad_strided_kernel<8, 2>), grid, dim3(256), 0, st, dy, psv, pqv, g, part, partb);
        else
            hipLaunchKernelGGL((wgrad_strided_kernel<8, 8>), grid, dim3(256), 0, st, dy, psv, pqv, g, part, partb);
        hipLaunchKernelGGL(wgrad_strided_reduce_kernel, dim3(cpn_cdiv(2LL * k * k * Cout * Cin + Cout, 64)), dim3(1024), 0, st,
                           part, partb, (int)nch, g, gwq, gws, gb);
        CPN_LAUNCH_CHECK("cpn_conv4d_strided_bwd(wgrad)");
    }
    return 0;
}
