// K9 / K10 — the two attention forms of UFCLayer on gfx950 (get_z path).
//
//   cpn_linear_attention   aggregation.LinearAttention.forward (/root/reference models/aggregation.py:84-117):
//                          phi = ELU + 1;  KV = sum_s phi(K)_s (x) V_s / L;  out = phi(Q).KV * L / (phi(Q).sum_s phi(K)_s + eps)
//   cpn_cross_attention    the cost-volume cross attention of UFCLayer.forward_cross (aggregation.py:327-328):
//                          src_attn = softmax_t(corr) . trg_v ,  trg_attn = softmax_s(corr)^T . src_v
//
// Both are tiny in FLOPs (<= 0.6 GFLOP per call) and were ~15 ATen launches each (elu, add, div, three einsums with
// their permute copies / two strided softmaxes and two einsums): HBM/launch-bound glue.  Here each is two launches
// that read every operand once in its native layout.  fp32 throughout.
#include "common.h"

namespace {

constexpr int LA_D = 32;                    // head dimension of q / k

__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x + 1.0f : expf(x); }        // elu(x) + 1, alpha = 1

// value addressing: token-major (B, L, H, Dv) [the feature branch] or channel-major (B, H, Dv, L) [the cost-volume
// branch: (B, H*Ht*Wt, fs, fs) maps are exactly that, so neither the operand nor the result needs a permute copy]
__device__ __forceinline__ size_t v_index(bool cm, int b, int l, int h, int dv, int L, int H, int Dv) {
    return cm ? (((size_t)b * H + h) * Dv + dv) * L + l : (((size_t)b * L + l) * H + h) * Dv + dv;
}

// ---- phase 1: per (b, h, 32-wide dv tile, l split): partial KV (32 x 32) and partial Ksum (32) ---------------------
// thread = (d, 4 dv columns); K and V rows of a 32-token tile are staged in LDS
// BWD (the backward's dKV / dKsum): k is phi(Q)'s source, v the output gradient, every token's value row is scaled by
// tok_scale[b,h,l] (= L Z_l) instead of 1/L and enters the row sum with weight tok_wt[b,h,l] (= dden_l) instead of 1
template <bool BWD>
__global__ __launch_bounds__(256) void linear_attention_reduce_kernel(
    const float* __restrict__ k, const float* __restrict__ v, int B, int L, int H, int Dv, int cm, int nsplit,
    const float* __restrict__ tok_scale, const float* __restrict__ tok_wt, float* __restrict__ kv_part,
    float* __restrict__ ks_part) {
    __shared__ float ks[32][LA_D + 1];
    __shared__ float vs[32][33];
    __shared__ float wt[32], sc[32];
    const int tid = threadIdx.x;
    const int dvt = blockIdx.x, h = blockIdx.y % H, b = blockIdx.y / H, sp = blockIdx.z;
    const int d = tid >> 3, c4 = (tid & 7) * 4;
    const int lper = (L + nsplit - 1) / nsplit;
    const int l0 = sp * lper, l1 = min(L, l0 + lper);
    const float invL = 1.0f / (float)L;
    float acc[4] = {0, 0, 0, 0}, ksum = 0.0f;
    for (int lt = l0; lt < l1; lt += 32) {
        if (BWD) {
            if (tid < 32) {
                const int l = lt + tid;
                sc[tid] = l < l1 ? tok_scale[((size_t)b * H + h) * L + l] : 0.0f;
                wt[tid] = l < l1 ? tok_wt[((size_t)b * H + h) * L + l] : 0.0f;
            }
            __syncthreads();
        }
        // stage 32 tokens: K (32 x 32, token-major always) and V (32 x 32 slice)
        for (int i = tid; i < 32 * 32; i += 256) {
            const int tl = i >> 5, e = i & 31;
            const int l = lt + tl;
            ks[tl][e] = l < l1 ? elu1(k[(((size_t)b * L + l) * H + h) * LA_D + e]) : 0.0f;
        }
        for (int i = tid; i < 32 * 32; i += 256) {
            // channel-major: consecutive threads walk l (contiguous); token-major: consecutive threads walk dv
            const int tl = cm ? (i & 31) : (i >> 5), e = cm ? (i >> 5) : (i & 31);
            const int l = lt + tl, dv = dvt * 32 + e;
            vs[tl][e] = (l < l1 && dv < Dv) ? v[v_index(cm, b, l, h, dv, L, H, Dv)] * (BWD ? sc[tl] : invL) : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int tl = 0; tl < 32; ++tl) {
            const float kd = ks[tl][d];
            ksum += BWD ? kd * wt[tl] : kd;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += kd * vs[tl][c4 + j];
        }
        __syncthreads();
    }
    const size_t pb = ((size_t)(b * H + h) * nsplit + sp);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dv = dvt * 32 + c4 + j;
        if (dv < Dv) kv_part[(pb * LA_D + d) * Dv + dv] = acc[j];
    }
    if (dvt == 0 && (tid & 7) == 0) ks_part[pb * LA_D + d] = ksum;
}

// ---- phase 2: out[l, h, dv] = L * (Q_l . KV[:, dv]) / (Q_l . Ksum + eps) ---------------------------------------------
// block = (b, h, 64-token tile, 64-wide dv tile); the split partials are summed in a fixed order (deterministic)
// BWD (the backward's dV): q is phi(K)'s source, kv_part is dKV and out[l, h, dv] = (phi(K)_l . dKV[:, dv]) / L
template <bool BWD>
__global__ __launch_bounds__(256) void linear_attention_apply_kernel(
    const float* __restrict__ q, const float* __restrict__ kv_part, const float* __restrict__ ks_part, int B, int L,
    int H, int Dv, int cm, int nsplit, float eps, float* __restrict__ out) {
    __shared__ float kvs[LA_D][65];
    __shared__ float kss[LA_D];
    __shared__ float qs[64][LA_D + 1];
    __shared__ float zs[64];
    const int tid = threadIdx.x;
    const int lt = blockIdx.x * 64, dv0 = blockIdx.y * 64;
    const int h = blockIdx.z % H, b = blockIdx.z / H;
    const size_t pb = (size_t)(b * H + h) * nsplit;
    for (int i = tid; i < LA_D * 64; i += 256) {
        const int d = i >> 6, e = i & 63;
        float s = 0.0f;
        if (dv0 + e < Dv)
            for (int sp = 0; sp < nsplit; ++sp) s += kv_part[((pb + sp) * LA_D + d) * Dv + dv0 + e];
        kvs[d][e] = s;
    }
    if (tid < LA_D) {
        float s = 0.0f;
        for (int sp = 0; sp < nsplit; ++sp) s += ks_part[(pb + sp) * LA_D + tid];
        kss[tid] = s;
    }
    for (int i = tid; i < 64 * LA_D; i += 256) {
        const int tl = i >> 5, d = i & 31;
        const int l = lt + tl;
        qs[tl][d] = l < L ? elu1(q[(((size_t)b * L + l) * H + h) * LA_D + d]) : 0.0f;
    }
    __syncthreads();
    if (tid < 64 && !BWD) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < LA_D; ++d) s += qs[tid][d] * kss[d];
        zs[tid] = 1.0f / (s + eps);
    }
    __syncthreads();
    const float fL = (float)L, invL = 1.0f / (float)L;
    for (int i = tid; i < 64 * 64; i += 256) {
        // channel-major output: consecutive threads walk l; token-major: consecutive threads walk dv
        const int tl = cm ? (i & 63) : (i >> 6), e = cm ? (i >> 6) : (i & 63);
        const int l = lt + tl, dv = dv0 + e;
        if (l >= L || dv >= Dv) continue;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < LA_D; ++d) s += qs[tl][d] * kvs[d][e];
        out[v_index(cm, b, l, h, dv, L, H, Dv)] = BWD ? s * invL : s * zs[tl] * fL;
    }
}

// ---- backward helper: T[l, :] = X[l, :] . M^T  (X: (L x Dv) values or output gradients, M: (32 x Dv) KV or dKV) ----------
// block = (b, h, 64-token tile); thread = (token, 8 of the 32 d).  Epilogues:
//   MODE 0 (X = dOut, M = KV, a = q):  a_l = phi(Q)_l . T_l,  Z_l = 1 / (phi(Q)_l . Ksum + eps),  dden_l = -L a_l Z_l^2,
//          dq_l = (L Z_l T_l + dden_l Ksum) phi'(q_l);  tok_scale = L Z_l and tok_wt = dden_l feed the dKV / dKsum reduce
//   MODE 1 (X = V,    M = dKV, a = k): dk_l = (T_l / L + dKsum) phi'(k_l)
// phi'(x) = 1 (x > 0) or exp(x) = phi(x)
template <int MODE>
__global__ __launch_bounds__(256) void linear_attention_tokens_kernel(
    const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ msum, const float* __restrict__ a,
    int B, int L, int H, int Dv, int cm, float eps, float* __restrict__ da, float* __restrict__ tok_scale,
    float* __restrict__ tok_wt) {
    __shared__ float xs[64][65];
    __shared__ float ms[LA_D][65];
    const int tid = threadIdx.x;
    const int lt = blockIdx.x * 64, h = blockIdx.y % H, b = blockIdx.y / H;
    const int tl = tid >> 2, d0 = (tid & 3) * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const float* mb = m + (size_t)(b * H + h) * LA_D * Dv;
    for (int dv0 = 0; dv0 < Dv; dv0 += 64) {
        for (int i = tid; i < 64 * 64; i += 256) {
            const int t = cm ? (i & 63) : (i >> 6), e = cm ? (i >> 6) : (i & 63);
            const int l = lt + t, dv = dv0 + e;
            xs[t][e] = (l < L && dv < Dv) ? x[v_index(cm, b, l, h, dv, L, H, Dv)] : 0.0f;
        }
        for (int i = tid; i < LA_D * 64; i += 256) {
            const int d = i >> 6, e = i & 63;
            ms[d][e] = dv0 + e < Dv ? mb[(size_t)d * Dv + dv0 + e] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int e = 0; e < 64; ++e) {
            const float xv = xs[tl][e];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += xv * ms[d0 + j][e];
        }
        __syncthreads();
    }
    const int l = lt + tl;
    const bool live = l < L;
    const size_t row = (((size_t)b * L + (live ? l : 0)) * H + h) * LA_D + d0;
    float raw[8], phi[8], sum8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        raw[j] = live ? a[row + j] : 0.0f;
        phi[j] = elu1(raw[j]);
        sum8[j] = msum[(size_t)(b * H + h) * LA_D + d0 + j];
    }
    float out[8];
    if (MODE == 0) {
        float at = 0.0f, den = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { at += phi[j] * acc[j]; den += phi[j] * sum8[j]; }
        at += __shfl_xor(at, 1); at += __shfl_xor(at, 2);
        den += __shfl_xor(den, 1); den += __shfl_xor(den, 2);
        const float z = 1.0f / (den + eps), fL = (float)L;
        const float dden = -fL * at * z * z;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = fL * z * acc[j] + dden * sum8[j];
        if (live && (tid & 3) == 0) {
            tok_scale[((size_t)b * H + h) * L + l] = fL * z;
            tok_wt[((size_t)b * H + h) * L + l] = dden;
        }
    } else {
        const float invL = 1.0f / (float)L;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = acc[j] * invL + sum8[j];
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) da[row + j] = out[j] * (raw[j] > 0.0f ? 1.0f : phi[j]);
}

// ---- cross attention, row direction: src_attn[b, s, h, :] = sum_t softmax_t(c[b,h,s,:])_t trg_v[b, t, h, :] ------------
// one wave per row s; the row's probabilities go through LDS, lanes 0-31 then own one output channel each
template <int C>
__global__ __launch_bounds__(256) void cross_rows_kernel(const float* __restrict__ c, const float* __restrict__ tv,
                                                         int B, int H, int S, int T, float* __restrict__ out) {
    extern __shared__ float prob[];                                   // 4 waves x T
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * 4 + wave, h = blockIdx.y % H, b = blockIdx.y / H;
    if (s >= S) return;
    const float* row = c + (((size_t)b * H + h) * S + s) * T;
    float* p = prob + wave * T;
    float m = -INFINITY;
    for (int t = lane; t < T; t += 64) { const float x = row[t]; p[t] = x; m = fmaxf(m, x); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float z = 0.0f;
    for (int t = lane; t < T; t += 64) { const float e = expf(p[t] - m); p[t] = e; z += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // lanes 0..C-1 own a channel; lanes C..2C-1 take the second half of the t range (C = 32: the whole wave works)
    const int ch = lane % C, part = lane / C, nparts = 64 / C;
    float acc = 0.0f;
    for (int t = part; t < T; t += nparts) acc += p[t] * tv[(((size_t)b * T + t) * H + h) * C + ch];
    for (int o = C; o < 64; o <<= 1) acc += __shfl_xor(acc, o);
    if (part == 0) out[(((size_t)b * S + s) * H + h) * C + ch] = acc / z;
}

// ---- cross attention, column direction: trg_attn[b, t, h, :] = sum_s softmax_s(c[b,h,:,t])_s src_v[b, s, h, :] --------
// block = XC_COLS columns x XC_GROUPS row groups: a thread walks rows s = g, g + XC_GROUPS, ... of its column with an
// online softmax (running max, sum, C-vector), the groups' partials meet in LDS.  One thread per column (round-2 first
// version) was 32 wavefronts on the whole chip for a (8, 256, 256) volume: 138 us per call.
constexpr int XC_COLS = 16, XC_GROUPS = 16;
template <int C>
__global__ __launch_bounds__(XC_COLS * XC_GROUPS) void cross_cols_kernel(const float* __restrict__ c, const float* __restrict__ sv,
                                                                        int B, int H, int S, int T, float* __restrict__ out) {
    extern __shared__ float svs[];                                    // S x C values of this (b, h), then the partials
    float* part = svs + (size_t)S * C;                                // [XC_GROUPS][C + 2][XC_COLS]
    const int l = threadIdx.x % XC_COLS, g = threadIdx.x / XC_COLS;
    const int t = blockIdx.x * XC_COLS + l, h = blockIdx.y % H, b = blockIdx.y / H;
    for (int i = threadIdx.x; i < S * C; i += XC_COLS * XC_GROUPS) svs[i] = sv[(((size_t)b * S + i / C) * H + h) * C + i % C];
    __syncthreads();
    float m = -INFINITY, z = 0.0f;
    float acc[C];
#pragma unroll
    for (int i = 0; i < C; ++i) acc[i] = 0.0f;
    if (t < T) {
        const float* col = c + ((size_t)b * H + h) * S * T + t;
        for (int s = g; s < S; s += XC_GROUPS) {
            const float x = col[(size_t)s * T];
            if (x > m) {
                const float r = expf(m - x);                          // 0 on the first row (m = -inf)
                z *= r;
#pragma unroll
                for (int i = 0; i < C; ++i) acc[i] *= r;
                m = x;
            }
            const float p = expf(x - m);
            z += p;
#pragma unroll
            for (int i = 0; i < C; ++i) acc[i] += p * svs[s * C + i];
        }
    }
    float* mine = part + (size_t)g * (C + 2) * XC_COLS + l;
    mine[0] = m;
    mine[XC_COLS] = z;
#pragma unroll
    for (int i = 0; i < C; ++i) mine[(2 + i) * XC_COLS] = acc[i];
    __syncthreads();
    // merge: thread (l, g) finishes channels g, g + XC_GROUPS, ... of column l
    if (t >= T) return;
    float M = -INFINITY;
    for (int q = 0; q < XC_GROUPS; ++q) M = fmaxf(M, part[(size_t)q * (C + 2) * XC_COLS + l]);
    float Z = 0.0f;
    float r[XC_GROUPS];
#pragma unroll
    for (int q = 0; q < XC_GROUPS; ++q) {
        const float mq = part[(size_t)q * (C + 2) * XC_COLS + l];
        r[q] = mq == -INFINITY ? 0.0f : expf(mq - M);
        Z += part[(size_t)q * (C + 2) * XC_COLS + XC_COLS + l] * r[q];
    }
    const float iz = 1.0f / Z;
    float* o = out + (((size_t)b * T + t) * H + h) * C;
    for (int i = g; i < C; i += XC_GROUPS) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < XC_GROUPS; ++q) v += part[(size_t)q * (C + 2) * XC_COLS + (2 + i) * XC_COLS + l] * r[q];
        o[i] = v * iz;
    }
}

// ---- backward of the cross attention (training) -------------------------------------------------------------------------
// With P1 = softmax_t(c), P2 = softmax_s(c), src_attn = P1 tv, trg_attn = P2^T sv and incoming gradients g1 (of src_attn),
// g2 (of trg_attn):
//   dtv[t] = sum_s P1[s,t] g1[s],   dsv[s] = sum_t P2[s,t] g2[t],
//   dc[s,t] = P1[s,t] (g1[s].tv[t] - g1[s].src_attn[s]) + P2[s,t] (sv[s].g2[t] - g2[t].trg_attn[t])
// (the library VJP re-ran both softmaxes and einsums and differentiated them: ~25 launches per call on 256 x 256 maps).
// Three launches: row / column statistics and the two subtracted dot products; a row-oriented kernel for dc and dsv; a
// column-oriented one for dtv.
constexpr int XB_C = 32;                  // channels per head
constexpr int XB_LD = XB_C + 1;           // LDS row stride (floats): threads of a wave read the same channel of 16 rows

// stats[(b*H + h)][0..2][S] = row max, 1 / row sum, g1.src_attn;  [3..5][T] = column max, 1 / column sum, g2.trg_attn
// grid (XB_PARTS, B*H): a block takes 1/XB_PARTS of the rows (one wave per row) and of the columns (32 columns x 8 row
// groups, online max / sum per thread, merged in LDS) — one block per (b, h) was 32 blocks on the chip, 100 us
constexpr int XB_PARTS = 8;
__global__ __launch_bounds__(256) void cross_bwd_stats_kernel(const float* __restrict__ c, const float* __restrict__ sa,
                                                              const float* __restrict__ ta, const float* __restrict__ g1,
                                                              const float* __restrict__ g2, int H, int S, int T,
                                                              float* __restrict__ stats) {
    __shared__ float pm[8][32], pz[8][32];
    const int bh = blockIdx.y, h = bh % H, b = bh / H, part = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* cm = c + (size_t)bh * S * T;
    float* st = stats + (size_t)bh * 3 * (S + T);
    const int sper = (S + XB_PARTS - 1) / XB_PARTS, s_lo = part * sper, s_hi = min(S, s_lo + sper);
    for (int s = s_lo + wave; s < s_hi; s += 4) {                      // rows: one wave each
        const float* row = cm + (size_t)s * T;
        float m = -INFINITY;
        for (int t = lane; t < T; t += 64) m = fmaxf(m, row[t]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float z = 0.0f;
        for (int t = lane; t < T; t += 64) z += expf(row[t] - m);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
        float r = 0.0f;
        if (lane < XB_C) {
            const size_t o = (((size_t)b * S + s) * H + h) * XB_C + lane;
            r = g1[o] * sa[o];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) r += __shfl_xor(r, o);
        if (lane == 0) { st[s] = m; st[S + s] = 1.0f / z; st[2 * S + s] = r; }
    }
    float* sc = st + 3 * S;
    const int tper = (T + XB_PARTS - 1) / XB_PARTS, t_lo = part * tper, t_hi = min(T, t_lo + tper);
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    for (int t0 = t_lo; t0 < t_hi; t0 += 32) {                         // columns: 32 at a time, 8 row groups each
        const int t = t0 + cl;
        float m = -INFINITY, z = 0.0f;
        if (t < t_hi)
            for (int s = rg; s < S; s += 8) {
                const float x = cm[(size_t)s * T + t];
                if (x > m) { z *= expf(m - x); m = x; }
                z += expf(x - m);
            }
        pm[rg][cl] = m;
        pz[rg][cl] = z;
        __syncthreads();
        if (rg == 0 && t < t_hi) {
            float M = pm[0][cl];
#pragma unroll
            for (int q = 1; q < 8; ++q) M = fmaxf(M, pm[q][cl]);
            float Z = 0.0f;
#pragma unroll
            for (int q = 0; q < 8; ++q) Z += pm[q][cl] == -INFINITY ? 0.0f : pz[q][cl] * expf(pm[q][cl] - M);
            const float* gp = g2 + (((size_t)b * T + t) * H + h) * XB_C;
            const float* ap = ta + (((size_t)b * T + t) * H + h) * XB_C;
            float r = 0.0f;
#pragma unroll
            for (int i = 0; i < XB_C; ++i) r += gp[i] * ap[i];
            sc[t] = M; sc[T + t] = 1.0f / Z; sc[2 * T + t] = r;
        }
        __syncthreads();
    }
}

// block = (b, h, 16 rows): thread (row r, column lane cl) walks t = cl, cl + 16, ...; dc written, dsv reduced over the lanes
__global__ __launch_bounds__(256) void cross_bwd_rows_kernel(const float* __restrict__ c, const float* __restrict__ sv,
                                                             const float* __restrict__ tv, const float* __restrict__ g1,
                                                             const float* __restrict__ g2, const float* __restrict__ stats,
                                                             int H, int S, int T, float* __restrict__ dc,
                                                             float* __restrict__ dsv) {
    extern __shared__ float xb[];                                     // tv and g2 of this (b, h): 2 x T x XB_LD
    float* tvs = xb;
    float* g2s = xb + (size_t)T * XB_LD;
    const int bh = blockIdx.y, h = bh % H, b = bh / H;
    for (int i = threadIdx.x; i < T * XB_C; i += 256) {
        const int t = i / XB_C, k = i - t * XB_C;
        const size_t o = (((size_t)b * T + t) * H + h) * XB_C + k;
        tvs[t * XB_LD + k] = tv[o];
        g2s[t * XB_LD + k] = g2[o];
    }
    __syncthreads();
    const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int s = blockIdx.x * 16 + r;
    const bool live = s < S;
    const int sc_ = live ? s : S - 1;
    const float* st = stats + (size_t)bh * 3 * (S + T);
    const float m1 = st[sc_], iz1 = st[S + sc_], r1 = st[2 * S + sc_];
    const float* sc = st + 3 * S;
    float gv[XB_C], sw[XB_C], acc[XB_C];
    const size_t so = (((size_t)b * S + sc_) * H + h) * XB_C;
#pragma unroll
    for (int k = 0; k < XB_C; ++k) { gv[k] = g1[so + k]; sw[k] = sv[so + k]; acc[k] = 0.0f; }
    const float* crow = c + ((size_t)bh * S + sc_) * T;
    float* drow = dc + ((size_t)bh * S + sc_) * T;
    for (int t = cl; t < T; t += 16) {
        const float x = crow[t];
        const float p1 = expf(x - m1) * iz1, p2 = expf(x - sc[t]) * sc[T + t];
        float d1 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < XB_C; ++k) {
            const float gk = g2s[t * XB_LD + k];
            d1 += gv[k] * tvs[t * XB_LD + k];
            d2 += sw[k] * gk;
            acc[k] += p2 * gk;
        }
        if (live) drow[t] = p1 * (d1 - r1) + p2 * (d2 - sc[2 * T + t]);
    }
#pragma unroll
    for (int k = 0; k < XB_C; ++k) {
        float v = acc[k];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        acc[k] = v;
    }
    if (live && cl == 0) {
#pragma unroll
        for (int k = 0; k < XB_C; ++k) dsv[so + k] = acc[k];
    }
}

// block = (b, h, 16 columns): thread (column cl, row group rg of 16) walks s = rg, rg + 16, ...; the groups meet in LDS
__global__ __launch_bounds__(256) void cross_bwd_cols_kernel(const float* __restrict__ c, const float* __restrict__ g1,
                                                             const float* __restrict__ stats, int H, int S, int T,
                                                             float* __restrict__ dtv) {
    extern __shared__ float xb[];                                     // g1 of this (b, h): S x XB_LD, then the partials
    float* g1s = xb;
    float* part = xb + (size_t)S * XB_LD;                             // [16 groups][XB_C][16 columns]
    const int bh = blockIdx.y, h = bh % H, b = bh / H;
    for (int i = threadIdx.x; i < S * XB_C; i += 256) {
        const int s = i / XB_C, k = i - s * XB_C;
        g1s[s * XB_LD + k] = g1[(((size_t)b * S + s) * H + h) * XB_C + k];
    }
    __syncthreads();
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int t = blockIdx.x * 16 + cl;
    const float* st = stats + (size_t)bh * 3 * (S + T);
    float acc[XB_C];
#pragma unroll
    for (int k = 0; k < XB_C; ++k) acc[k] = 0.0f;
    if (t < T) {
        const float* col = c + (size_t)bh * S * T + t;
        for (int s = rg; s < S; s += 16) {
            const float p1 = expf(col[(size_t)s * T] - st[s]) * st[S + s];
#pragma unroll
            for (int k = 0; k < XB_C; ++k) acc[k] += p1 * g1s[s * XB_LD + k];
        }
    }
#pragma unroll
    for (int k = 0; k < XB_C; ++k) part[(rg * XB_C + k) * 16 + cl] = acc[k];
    __syncthreads();
    // thread (cl, rg) finishes channels rg and rg + 16 of column cl
    if (t >= T) return;
    for (int k = rg; k < XB_C; k += 16) {
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += part[(q * XB_C + k) * 16 + cl];
        dtv[(((size_t)b * T + t) * H + h) * XB_C + k] = v;
    }
}

// the nsplit partial (KV, Ksum) blocks of every (b, h) summed in a fixed order into one block: the apply kernel then reads
// 4 KB per head instead of nsplit x 4 KB per workgroup (with 64 splits that was 268 MB of L2 reads per call)
__global__ __launch_bounds__(256) void linear_attention_combine_kernel(const float* __restrict__ kv_part,
                                                                       const float* __restrict__ ks_part, int nsplit, int Dv,
                                                                       long long t1, long long t2, float* __restrict__ kv_sum,
                                                                       float* __restrict__ ks_sum) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= t1 + t2) return;
    const bool second = i >= t1;
    if (second) i -= t1;
    const int per = second ? LA_D : LA_D * Dv;
    const float* part = second ? ks_part : kv_part;
    const long long bh = i / per, e = i - bh * per;
    float acc = 0.0f;
    for (int sp = 0; sp < nsplit; ++sp) acc += part[(bh * nsplit + sp) * per + e];
    (second ? ks_sum : kv_sum)[i] = acc;
}

}  // namespace

extern "C" long long cpn_linear_attention_scratch(int B, int H, int Dv, int nsplit) {
    return (long long)B * H * (nsplit + 1) * LA_D * (Dv + 1);
}


extern "C" int cpn_linear_attention(const float* q, const float* k, const float* v, int B, int L, int H, int Dv,
                                    int channel_major, float eps, int nsplit, float* scratch, float* out, void* stream) {
    CPN_REQUIRE(q && k && v && scratch && out, CPN_E_ARG, "cpn_linear_attention: null pointer");
    CPN_REQUIRE(B > 0 && L > 0 && H > 0 && Dv > 0 && nsplit > 0 && nsplit <= 64 && (long long)B * H < 65536, CPN_E_SHAPE,
                "cpn_linear_attention: bad shape");
    const hipStream_t s = (hipStream_t)stream;
    float* kv_part = scratch;
    float* ks_part = scratch + (size_t)B * H * nsplit * LA_D * Dv;
    hipLaunchKernelGGL(linear_attention_reduce_kernel<false>, dim3(cpn_cdiv(Dv, 32), B * H, nsplit), dim3(256), 0, s, k, v, B,
                       L, H, Dv, channel_major, nsplit, (const float*)nullptr, (const float*)nullptr, kv_part, ks_part);
    if (nsplit > 1) {
        // fixed-order sum of the partials, once per head instead of once per apply workgroup
        float* kv_sum = scratch + (size_t)B * H * nsplit * LA_D * (Dv + 1);
        float* ks_sum = kv_sum + (size_t)B * H * LA_D * Dv;
        const long long t1 = (long long)B * H * LA_D * Dv, t2 = (long long)B * H * LA_D;
        hipLaunchKernelGGL(linear_attention_combine_kernel, dim3((unsigned)cpn_cdiv(t1 + t2, 256)), dim3(256), 0, s, kv_part,
                           ks_part, nsplit, Dv, t1, t2, kv_sum, ks_sum);
        kv_part = kv_sum;
        ks_part = ks_sum;
    }
    hipLaunchKernelGGL(linear_attention_apply_kernel<false>, dim3(cpn_cdiv(L, 64), cpn_cdiv(Dv, 64), B * H), dim3(256), 0, s,
                       q, kv_part, ks_part, B, L, H, Dv, channel_major, nsplit > 1 ? 1 : nsplit, eps, out);
    CPN_LAUNCH_CHECK("cpn_linear_attention");
    return 0;
}

// scratch of the backward: two partial/combined (KV, Ksum) sets (the recomputed forward one and the gradient one) and the two
// per-token scalars
extern "C" long long cpn_linear_attention_bwd_scratch(int B, int L, int H, int Dv, int nsplit) {
    return 2 * cpn_linear_attention_scratch(B, H, Dv, nsplit) + 2LL * B * H * L;
}

// VJP of cpn_linear_attention: dq, dk (B, L, H, 32) and dv (layout of v) from dout (layout of out).  With P = phi(Q),
// N = phi(K), KV = sum_s N_s (x) V_s / L, Ks = sum_s N_s, Z_l = 1 / (P_l . Ks + eps), out_l = L Z_l P_l . KV:
//   T_l = dout_l . KV^T,  a_l = P_l . T_l,  dden_l = -L a_l Z_l^2,  dP_l = L Z_l T_l + dden_l Ks
//   dKV = sum_l P_l (x) (L Z_l dout_l),  dKs = sum_l dden_l P_l
//   dN_s = V_s . dKV^T / L + dKs,  dV_s = N_s . dKV / L
// (reference: autograd through models/aggregation.py:84-117.)  Launches: reduce + combine (KV, Ks again), tokens<0>, reduce<BWD>
// + combine, tokens<1>, apply<BWD>; every operand is read in its native layout.
extern "C" int cpn_linear_attention_bwd(const float* q, const float* k, const float* v, const float* dout, int B, int L, int H,
                                        int Dv, int channel_major, float eps, int nsplit, float* scratch, float* dq, float* dk,
                                        float* dv, void* stream) {
    CPN_REQUIRE(q && k && v && dout && scratch && dq && dk && dv, CPN_E_ARG, "cpn_linear_attention_bwd: null pointer");
    CPN_REQUIRE(B > 0 && L > 0 && H > 0 && Dv > 0 && nsplit > 0 && nsplit <= 64 && (long long)B * H < 65536, CPN_E_SHAPE,
                "cpn_linear_attention_bwd: bad shape");
    const hipStream_t s = (hipStream_t)stream;
    const size_t set = (size_t)cpn_linear_attention_scratch(B, H, Dv, nsplit);
    const size_t parts = (size_t)B * H * nsplit * LA_D * Dv, partsum = (size_t)B * H * nsplit * LA_D * (Dv + 1);
    const long long t1 = (long long)B * H * LA_D * Dv, t2 = (long long)B * H * LA_D;
    float* tok_scale = scratch + 2 * set;
    float* tok_wt = tok_scale + (size_t)B * H * L;
    const float* sums[2][2];
    for (int pass = 0; pass < 2; ++pass) {
        float* base = scratch + pass * set;
        float* kv_part = base;
        float* ks_part = base + parts;
        if (pass == 0)
            hipLaunchKernelGGL(linear_attention_reduce_kernel<false>, dim3(cpn_cdiv(Dv, 32), B * H, nsplit), dim3(256), 0, s, k,
                               v, B, L, H, Dv, channel_major, nsplit, (const float*)nullptr, (const float*)nullptr, kv_part,
                               ks_part);
        else
            hipLaunchKernelGGL(linear_attention_reduce_kernel<true>, dim3(cpn_cdiv(Dv, 32), B * H, nsplit), dim3(256), 0, s, q,
                               dout, B, L, H, Dv, channel_major, nsplit, (const float*)tok_scale, (const float*)tok_wt,
                               kv_part, ks_part);
        if (nsplit > 1) {
            float* kv_sum = base + partsum;
            float* ks_sum = kv_sum + t1;
            hipLaunchKernelGGL(linear_attention_combine_kernel, dim3((unsigned)cpn_cdiv(t1 + t2, 256)), dim3(256), 0, s,
                               kv_part, ks_part, nsplit, Dv, t1, t2, kv_sum, ks_sum);
            kv_part = kv_sum;
            ks_part = ks_sum;
        }
        sums[pass][0] = kv_part;
        sums[pass][1] = ks_part;
        if (pass == 0)
            hipLaunchKernelGGL(linear_attention_tokens_kernel<0>, dim3(cpn_cdiv(L, 64), B * H), dim3(256), 0, s, dout,
                               sums[0][0], sums[0][1], q, B, L, H, Dv, channel_major, eps, dq, tok_scale, tok_wt);
    }
    hipLaunchKernelGGL(linear_attention_tokens_kernel<1>, dim3(cpn_cdiv(L, 64), B * H), dim3(256), 0, s, v, sums[1][0],
                       sums[1][1], k, B, L, H, Dv, channel_major, eps, dk, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(linear_attention_apply_kernel<true>, dim3(cpn_cdiv(L, 64), cpn_cdiv(Dv, 64), B * H), dim3(256), 0, s, k,
                       sums[1][0], sums[1][1], B, L, H, Dv, channel_major, 1, eps, dv);
    CPN_LAUNCH_CHECK("cpn_linear_attention_bwd");
    return 0;
}

extern "C" int cpn_cross_attention(const float* corr, const float* src_v, const float* trg_v, int B, int H, int S, int T,
                                   int C, float* src_attn, float* trg_attn, void* stream) {
    CPN_REQUIRE(corr && src_v && trg_v && src_attn && trg_attn, CPN_E_ARG, "cpn_cross_attention: null pointer");
    CPN_REQUIRE(B > 0 && H > 0 && S > 0 && T > 0 && C == 32 && (long long)B * H < 65536 && S * C * 4 <= 64 * 1024 &&
                    T * 16 <= 64 * 1024, CPN_E_SHAPE, "cpn_cross_attention: need C == 32, S, T <= 512 (got C=%d S=%d T=%d)", C, S, T);
    const hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cross_rows_kernel<32>, dim3(cpn_cdiv(S, 4), B * H), dim3(256), (size_t)4 * T * sizeof(float), s, corr,
                       trg_v, B, H, S, T, src_attn);
    const size_t lds_cols = ((size_t)S * 32 + (size_t)XC_GROUPS * 34 * XC_COLS) * sizeof(float);      // <= 99 KiB at S = 512
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)cross_cols_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)((512 * 32 + XC_GROUPS * 34 * XC_COLS) * sizeof(float)));
        if (e != hipSuccess) {
            cpn_set_error("cpn_cross_attention: cannot reserve LDS: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(cross_cols_kernel<32>, dim3(cpn_cdiv(T, XC_COLS), B * H), dim3(XC_COLS * XC_GROUPS), lds_cols, s, corr,
                       src_v, B, H, S, T, trg_attn);
    CPN_LAUNCH_CHECK("cpn_cross_attention");
    return 0;
}

extern "C" long long cpn_cross_attention_bwd_scratch(int B, int H, int S, int T) { return 3LL * B * H * (S + T); }

// VJP of cpn_cross_attention (autograd through models/aggregation.py:327-328): g_src / g_trg are the gradients of src_attn /
// trg_attn, which are passed back in (forward outputs) for the two subtracted dot products; dcorr, dsrc_v, dtrg_v are written.
extern "C" int cpn_cross_attention_bwd(const float* corr, const float* src_v, const float* trg_v, const float* src_attn,
                                       const float* trg_attn, const float* g_src, const float* g_trg, int B, int H, int S,
                                       int T, int C, float* scratch, float* dcorr, float* dsrc_v, float* dtrg_v, void* stream) {
    CPN_REQUIRE(corr && src_v && trg_v && src_attn && trg_attn && g_src && g_trg && scratch && dcorr && dsrc_v && dtrg_v,
                CPN_E_ARG, "cpn_cross_attention_bwd: null pointer");
    CPN_REQUIRE(B > 0 && H > 0 && S > 0 && T > 0 && C == XB_C && (long long)B * H < 65536 && S <= 512 && T <= 512, CPN_E_SHAPE,
                "cpn_cross_attention_bwd: need C == 32, S, T <= 512 (got C=%d S=%d T=%d)", C, S, T);
    const hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        const int rows_lds = 2 * 512 * XB_LD * 4, cols_lds = (512 * XB_LD + 16 * XB_C * 16) * 4;
        hipError_t e = hipFuncSetAttribute((const void*)cross_bwd_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, rows_lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)cross_bwd_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, cols_lds);
        if (e != hipSuccess) {
            cpn_set_error("cpn_cross_attention_bwd: cannot reserve LDS: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(cross_bwd_stats_kernel, dim3(XB_PARTS, B * H), dim3(256), 0, s, corr, src_attn, trg_attn, g_src, g_trg, H,
                       S, T, scratch);
    hipLaunchKernelGGL(cross_bwd_rows_kernel, dim3(cpn_cdiv(S, 16), B * H), dim3(256), (size_t)2 * T * XB_LD * sizeof(float), s,
                       corr, src_v, trg_v, g_src, g_trg, (const float*)scratch, H, S, T, dcorr, dsrc_v);
    hipLaunchKernelGGL(cross_bwd_cols_kernel, dim3(cpn_cdiv(T, 16), B * H), dim3(256),
                       ((size_t)S * XB_LD + 16 * XB_C * 16) * sizeof(float), s, corr, g_src, (const float*)scratch, H, S, T, dtrg_v);
    CPN_LAUNCH_CHECK("cpn_cross_attention_bwd");
    return 0;
}

// ---- q / k of UFCLayer.forward_attention from the LOW-resolution projection (round 3) -----------------------------
// aggregation.py:276-281: q = q_proj(cat(interp(corr maps, fs), norm1(feat))) + pos_embed, same for k.  A Linear layer
// acts per position on the channels, bilinear interpolation per channel on the positions: they commute, so the 2048
// cost-volume channels are projected at their native Hs x Ws positions (16 x 16: 6 - 16x fewer FLOPs than at fs x fs) and
// the 2d projected channels are upsampled instead.  This kernel does the upsampling (align_corners=True, the arithmetic
// of cpn_resize_bilinear_ac), adds the feature half of the projection (+ bias) and the positional embedding and writes
// q and k in the (B, L, H, 32) layout the linear attention reads.
namespace {
__global__ __launch_bounds__(256) void qk_assemble_kernel(const float* __restrict__ lin, const float* __restrict__ low,
                                                          const float* __restrict__ pos, int L, int fs, int h, int w, int d,
                                                          int dim, float* __restrict__ q, float* __restrict__ k) {
    const int b = blockIdx.y;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // over (token, channel of [q | k])
    if (idx >= (long long)L * 2 * d) return;
    const int c = (int)(idx % (2 * d)), l = (int)(idx / (2 * d));
    const int Y = l / fs, X = l - Y * fs;
    const float sy = fs > 1 ? (float)(h - 1) / (float)(fs - 1) : 0.0f;
    const float sx = fs > 1 ? (float)(w - 1) / (float)(fs - 1) : 0.0f;
    const float fy = sy * (float)Y, fx = sx * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = low + ((size_t)b * 2 * d + c) * h * w;
    const float top = p[y0 * w + x0] * (1.0f - lx) + p[y0 * w + x1] * lx;
    const float bot = p[y1 * w + x0] * (1.0f - lx) + p[y1 * w + x1] * lx;
    const float up = top * (1.0f - ly) + bot * ly;
    const float v = (lin[((size_t)b * L + l) * 2 * d + c] + up) + pos[(size_t)l * dim + (c % dim)];
    if (c < d) q[((size_t)b * L + l) * d + c] = v;
    else k[((size_t)b * L + l) * d + (c - d)] = v;
}
}  // namespace

extern "C" int cpn_qk_assemble(const float* lin, const float* low, const float* pos, int B, int fs, int h, int w, int nhead,
                               int dim, float* q, float* k, void* stream) {
    CPN_REQUIRE(lin && low && pos && q && k, CPN_E_ARG, "cpn_qk_assemble: null pointer");
    CPN_REQUIRE(B > 0 && B < 65536 && fs > 0 && h > 0 && w > 0 && nhead > 0 && dim > 0, CPN_E_SHAPE, "cpn_qk_assemble: bad shape");
    const int L = fs * fs, d = nhead * dim;
    hipLaunchKernelGGL(qk_assemble_kernel, dim3(cpn_cdiv((long long)L * 2 * d, 256), B), dim3(256), 0, (hipStream_t)stream, lin,
                       low, pos, L, fs, h, w, d, dim, q, k);
    CPN_LAUNCH_CHECK("cpn_qk_assemble");
    return 0;
}

// ---- cost-volume side of UFCLayer.forward_attention at the volume's NATIVE resolution (round 3) -------------------
// aggregation.py:283-297: value_corr (B, H*Ht*Wt, Hs, Ws) is upsampled to fs x fs (U, bilinear, align_corners), run through
// the linear attention with the fs*fs queries / keys, and the message is sampled back down to Hs x Ws (D).  U and D are
// linear maps over the positions, the attention is linear in the values, so with K' = phi(k), Q' = phi(q), Z_l = 1 /
// (Q'_l . sum_m K'_m + eps):
//     KV   = sum_l K'_l (U v_low)_l^T  = (U^T K')^T v_low                      -> Kd = U^T K'        (P x 32 per head)
//     msg  = D (diag(Z) Q' KV)         = (D diag(Z) Q') KV                      -> Qd = D (Z . Q')    (P x 32 per head)
//     msg_low = Qd (Kd^T v_low)        (P x Dv), P = Hs*Ws = 256, Dv = Ht*Wt = 256: no 2 048-channel tensor at fs x fs exists
// (exact in real arithmetic; the reference's 1/L and *L cancel).  Two launches:
//   cva_kd       per (b, h, block of 32 positions): Kd as a gather (fixed order), the block's partial of M = Kd^T v_low
//   cva_out      per (b, h, 8 positions): M = sum of the partials, Qd from the 4 taps of D, out = residual + Qd M
namespace {
constexpr int CVA_D = 32;

__device__ __forceinline__ void ac_taps(int Y, int X, int n_out, int n_in, int (&idx)[4], float (&w)[4]) {
    // the 4 source positions / weights of output pixel (Y, X) of an n_in -> n_out bilinear resize (align_corners=True),
    // with the arithmetic of resize_bilinear_ac_kernel
    const float s = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
    const float fy = s * (float)Y, fx = s * (float)X;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < n_in - 1), x1 = x0 + (x0 < n_in - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    idx[0] = y0 * n_in + x0; idx[1] = y0 * n_in + x1; idx[2] = y1 * n_in + x0; idx[3] = y1 * n_in + x1;
    w[0] = (1.0f - lx) * (1.0f - ly); w[1] = lx * (1.0f - ly); w[2] = (1.0f - lx) * ly; w[3] = lx * ly;
}

// K1: workgroup = (block of 32 low-res positions, (b, h)).  Kd of its positions as a GATHER over the tokens whose bilinear
// footprint holds the position (fixed order: no atomics, the result is bit-reproducible), then its partial of
// M = Kd^T v_low (32 x Dv) and its share of sum_l phi(k_l).
constexpr int CVA_BP = 16;
__global__ __launch_bounds__(256) void cva_kd_kernel(const float* __restrict__ k, const float* __restrict__ v_low, int L, int H,
                                                     int fs, int hs, int Dv, int nblk, float* __restrict__ kvm_part,
                                                     float* __restrict__ ksum_out) {
    __shared__ float kd[CVA_BP][CVA_D];
    __shared__ float tree[8][CVA_D];
    const int P = hs * hs;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, blk = blockIdx.x;
    const int d = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float* kb = k + ((size_t)b * L * H + h) * CVA_D + d;    // token l at kb[l * H * 32]
    {
        // this block's share of sum_l phi(k_l) (tokens blk*Lb .. ): the nblk partials are added in block order by cva_out
        const int Lb = (L + nblk - 1) / nblk, l0 = blk * Lb, l1 = min(L, l0 + Lb);
        float a = 0.0f;
        for (int l = l0 + g; l < l1; l += 8) a += elu1(kb[(size_t)l * H * CVA_D]);
        tree[g][d] = a;
        __syncthreads();
        if (g == 0) {
            float t = tree[0][d];
#pragma unroll
            for (int i = 1; i < 8; ++i) t += tree[i][d];
            ksum_out[((size_t)bh * nblk + blk) * CVA_D + d] = t;
        }
    }
    const float inv_s = fs > 1 && hs > 1 ? (float)(fs - 1) / (float)(hs - 1) : 1.0f;     // tokens per low-res step
    const float sc = fs > 1 ? (float)(hs - 1) / (float)(fs - 1) : 0.0f;                    // the resize kernel's scale
    // the upsampling is separable: weight of token (Y, X) at position (y, x) = wy(Y, y) * wx(X, x), each the 1-D bilinear
    // weight of resize_bilinear_ac_kernel (y0 = (int)(sc*Y), y1 = y0 + (y0 < hs-1), ly = sc*Y - y0)
    auto w1d = [&](int T, int t) {
        const float f = sc * (float)T;
        const int t0 = (int)f, t1 = t0 + (t0 < hs - 1);
        const float l = f - (float)t0;
        return (t == t0 ? 1.0f - l : 0.0f) + (t == t1 ? l : 0.0f);
    };
    for (int pp = g; pp < CVA_BP; pp += 8) {
        const int p = blk * CVA_BP + pp;
        float acc = 0.0f;
        if (p < P) {
            const int y = p / hs, x = p - y * hs;
            const int Y0 = max(0, (int)floorf((float)(y - 1) * inv_s) - 1), Y1 = min(fs - 1, (int)ceilf((float)(y + 1) * inv_s) + 1);
            const int X0 = max(0, (int)floorf((float)(x - 1) * inv_s) - 1), X1 = min(fs - 1, (int)ceilf((float)(x + 1) * inv_s) + 1);
            for (int Y = Y0; Y <= Y1; ++Y) {
                const float wy = w1d(Y, y);
                if (wy == 0.0f) continue;
                float row = 0.0f;
                for (int X = X0; X <= X1; ++X) {
                    const float wx = w1d(X, x);
                    if (wx != 0.0f) row += wx * elu1(kb[(size_t)(Y * fs + X) * H * CVA_D]);
                }
                acc += wy * row;
            }
        }
        kd[pp][d] = acc;
    }
    __syncthreads();
    const float* vb = v_low + (size_t)bh * P * Dv;
    float* out = kvm_part + ((size_t)bh * nblk + blk) * CVA_D * Dv;
    for (int v = threadIdx.x; v < Dv; v += 256) {
        float m[CVA_D];
#pragma unroll
        for (int dd = 0; dd < CVA_D; ++dd) m[dd] = 0.0f;
        for (int pp = 0; pp < CVA_BP; ++pp) {
            const int p = blk * CVA_BP + pp;
            if (p >= P) break;
            const float xv = vb[(size_t)p * Dv + v];
#pragma unroll
            for (int dd = 0; dd < CVA_D; ++dd) m[dd] += kd[pp][dd] * xv;
        }
#pragma unroll
        for (int dd = 0; dd < CVA_D; ++dd) out[(size_t)dd * Dv + v] = m[dd];
    }
}

constexpr int CVA_PB = 8;                                         // low-res positions per workgroup of cva_out
__global__ __launch_bounds__(256) void cva_out_kernel(const float* __restrict__ q, const float* __restrict__ kvm_part,
                                                      const float* __restrict__ ksum_in, const float* __restrict__ residual,
                                                      int L, int H, int fs, int hs, int Dv, int nblk, float eps,
                                                      float* __restrict__ out) {
    __shared__ float qd[CVA_PB][CVA_D];
    __shared__ float ksum[CVA_D];
    const int P = hs * hs;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, p0 = blockIdx.x * CVA_PB;
    const float* mb = kvm_part + (size_t)bh * nblk * CVA_D * Dv;
    if (threadIdx.x < CVA_D) {
        float t = 0.0f;
        for (int kb2 = 0; kb2 < nblk; ++kb2) t += ksum_in[((size_t)bh * nblk + kb2) * CVA_D + threadIdx.x];
        ksum[threadIdx.x] = t;
    }
    __syncthreads();
    {
        // Qd[p][d] = sum_t w_t Z_{l_t} phi(q_{l_t})[d]: 8 positions x 32 features = 256 threads; Z through a 32-lane sum
        const int d = threadIdx.x & 31, pp = threadIdx.x >> 5, p = p0 + pp;
        float acc = 0.0f;
        if (p < P) {
            int idx[4];
            float w[4];
            ac_taps(p / hs, p % hs, hs, fs, idx, w);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float qv = elu1(q[(((size_t)b * L + idx[t]) * H + h) * CVA_D + d]);
                float dot = qv * ksum[d];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o);            // over the 32 features of this position
                acc += w[t] * qv / (dot + eps);
            }
        }
        qd[pp][d] = acc;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < Dv; v += 256) {
        float m[CVA_D];
#pragma unroll
        for (int d = 0; d < CVA_D; ++d) m[d] = 0.0f;
        for (int kb2 = 0; kb2 < nblk; ++kb2)                       // the position blocks' partials, fixed order
#pragma unroll
            for (int d = 0; d < CVA_D; ++d) m[d] += mb[((size_t)kb2 * CVA_D + d) * Dv + v];
#pragma unroll
        for (int pp = 0; pp < CVA_PB; ++pp) {
            const int p = p0 + pp;
            if (p >= P) break;
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < CVA_D; ++d) a += qd[pp][d] * m[d];
            const size_t o = ((size_t)bh * P + p) * Dv + v;
            out[o] = residual ? residual[o] + a : a;
        }
    }
}
}  // namespace

extern "C" long long cpn_cost_volume_attention_scratch(int B, int L, int H, int P, int Dv) {
    const long long nblk = (P + CVA_BP - 1) / CVA_BP;
    return (long long)B * H * nblk * (CVA_D * Dv + CVA_D);
}

extern "C" int cpn_cost_volume_attention(const float* q, const float* k, const float* v_low, const float* residual, int B,
                                         int fs, int H, int hs, int Dv, float eps, float* scratch, float* out, void* stream) {
    CPN_REQUIRE(q && k && v_low && scratch && out, CPN_E_ARG, "cpn_cost_volume_attention: null pointer");
    CPN_REQUIRE(B > 0 && H > 0 && (long long)B * H < 65536 && fs >= hs && hs > 0 && Dv > 0, CPN_E_SHAPE,
                "cpn_cost_volume_attention: bad shape (fs=%d hs=%d)", fs, hs);
    const int L = fs * fs, P = hs * hs, nblk = (P + CVA_BP - 1) / CVA_BP;
    float* kvm_part = scratch;
    float* ksum = scratch + (size_t)B * H * nblk * CVA_D * Dv;
    const hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cva_kd_kernel, dim3(nblk, B * H), dim3(256), 0, st, k, v_low, L, H, fs, hs, Dv, nblk, kvm_part, ksum);
    hipLaunchKernelGGL(cva_out_kernel, dim3(cpn_cdiv(P, CVA_PB), B * H), dim3(256), 0, st, q, (const float*)kvm_part,
                       (const float*)ksum, residual, L, H, fs, hs, Dv, nblk, eps, out);
    CPN_LAUNCH_CHECK("cpn_cost_volume_attention");
    return 0;
}
