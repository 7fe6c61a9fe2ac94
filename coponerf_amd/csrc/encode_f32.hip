// The reference-arithmetic mode (RenderEngine.precision = "f32") in the RESTRUCTURED formulation (round 6): the same algebra the
// fp16 default uses - node tables for the three coarse levels, the folded key / value projections, the attention-weighted sums
// on the hidden activations - with fp32 tables, fp32 blends and products, and the hidden activations
// carried as fp16 (hi, lo) PAIRS: hi = fp16(x), lo = fp16(x - hi), 22 significant bits, products against (hi, lo) weights are
// exact in the fp32 accumulators of the fp16 MFMA.  Round 5 ran this mode in the reference's own layer order (gather + nine
// exact-fp32 GEMM launches per 128-column block: 43.7 TFLOP per 256^2 x 64 image, 78 k rays/s); the algebra removes 85 % of that.
//
// Replaces, per sample (/root/reference models/CoPoNeRF.py:312, 370, 384-397): F.grid_sample x 8 + cat + query_encode_latent
// (Conv2d 835 -> 832) + ReLU, and per ray (:450-461, 475-485) the joint softmax + weighted sum, evaluated on the hidden layer.
//
//   cpn_node_features_f32   the three coarse levels sampled at every table node, fp32 (node_features_kernel of encode.hip in fp32)
//   cpn_encode_hidden_f32   hid = ReLU(sum_t a_t T32[node_t] + W[:, 768:835] . [gather_3 | tanh(pt/5)] + b) -> (hi, lo) fp16 pairs
//   cpn_attend_hidden_f32   joint softmax of <qa, qb> / 11.31 over the 2 S samples of a ray, hbar = sum_s w_s (hi_s + lo_s), fp32
//
// gfx950 notes.  cpn_encode_hidden_f32 is a persistent kernel, one 512-thread workgroup per CU; a workgroup computes HALF of the
// layer's channels (416) for its range of rows, so that its share of the K = 67 (+ bias) fp32 block - 111 KB; the whole block is
// 223 KB and fits neither LDS nor a per-batch L2 refetch (14 KB per row) - stays in LDS for the whole launch (the first version
// kept 136 weights per thread in registers: the allocator spilled them; the second ran the contraction as packed fp32 FMAs
// against broadcast LDS reads of the K operand and was bound by the LDS return path - a broadcast ds_read_b128 still delivers
// 1 KiB to the wave: 8.1 us per batch).  Seven OWNER waves run it on the fp32 MFMA (16x16x4: one 4-byte LDS read per operand
// and lane), wave w owning the channel tiles 4w .. 4w+3.  Rows come in batches of 16
// (8 samples x {own, other image}); the eighth wave is the PRODUCER: while the owners multiply batch n it resolves batch n + 1 -
// tap records, the 64 bilinear full-resolution channels and the point encodings - into the other half of a double-buffered LDS
// image, k-major = the MFMA's B-operand order.  One barrier per batch.  An owner lane (row n, channel group g) requests its 16
// table taps (16-byte loads: the four lanes of a row read 64 contiguous bytes of a node row), runs the 17 k4 steps, then blends
// the taps in and stores (hi, lo) halves.  (The first version -
// every phase by all threads, three barriers per batch, taps behind the contraction - took 20 ms per 16 384 rays under two
// co-running calls; its phases were latency chains of 2 + 2 + 6 us around 4.4 us of FMAs.)
// Bound: fp32 MFMA issue (1.9 TFLOP per 65 536-ray image against 157 TFLOP/s) at 2 waves per SIMD.
#include <algorithm>

// timing-only ablations (results are wrong when non-zero; the product builds with 0): 1 = every table tap reads node 0,
// 2 = no contraction, 4 = no stores, 8 = no table taps at all
#ifndef CPN_EF32_ABLATE
#define CPN_EF32_ABLATE 0
#endif

#include "encode_common.h"

namespace {

constexpr int EB = 16;                   // rows (ray, view, sample, image) per batch
constexpr int KX = 68;                   // 64 full-resolution channels + 3 point encodings + the bias (x = 1)
constexpr int TABF = CPN_TAB_LD;         // 832 floats per node

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void node_features_f32_kernel(const float* __restrict__ map0, const float* __restrict__ map1,
                                         const float* __restrict__ map2, int H, int W, long long total,
                                         float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int chunk = (int)(idx % 192);                                 // 3 levels x 64 chunks of 4 channels
    const long long node = idx / 192;
    const int lvl = chunk >> 6, c4 = chunk & 63;
    const NodeGrid ng{W >> 1, H >> 1};
    const long long npi = ng.nodes_per_image();
    const int img = (int)(node / npi);
    long long rem = node - (long long)img * npi;
    const bool border = rem < ng.border_nodes();
    if (!border) rem -= ng.border_nodes();
    const int nw = border ? ng.bw() : ng.zw(), pad = border ? 0 : PAD;
    const int ny = (int)(rem / nw) - pad, nx = (int)(rem % nw) - pad;
    const float gx = (float)(2 * nx - ng.Mx) / (float)ng.Mx, gy = (float)(2 * ny - ng.My) / (float)ng.My;
    const int shift = 4 - lvl;
    const int Hl = H >> shift, Wl = W >> shift;
    const Taps tp = make_taps(gx, gy, Wl, Hl, border);
    const float* m = (lvl == 0 ? map0 : lvl == 1 ? map1 : map2) + (size_t)img * Hl * Wl * 256 + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(m + (size_t)tp.off[k] * 256);
        acc += v * tp.w[k];
    }
    *reinterpret_cast<f32x4*>(out + (size_t)node * 768 + lvl * 256 + c4 * 4) = acc;
}

// node_taps (encode_common.h) with element offsets into an fp32 table
struct TapF { unsigned off[4]; float w[4]; };
__device__ __forceinline__ TapF node_taps_f32(float gx, float gy, const NodeGrid ng, bool border, unsigned base_nodes) {
    const TapRec t = node_taps(gx, gy, ng, border);
    TapF o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o.off[k] = (base_nodes + (unsigned)(t.off[k] / TAB_ROW_BYTES)) * (unsigned)TABF;
        o.w[k] = t.w[k];
    }
    return o;
}

// LDS image of one batch of 16 rows: what the producer wave resolves for the seven owner waves
struct BatchLds {
    float xsT[KX][EB];                   // K operand, k-major (16-byte aligned rows)
    unsigned toff[EB][4];                // BYTE offsets of the 4 table nodes
    float tw[EB][4];                     // their bilinear weights (0 for rows past the end)
    long long orow[EB];                  // output row (ray, view, sample) or -1
};

// The producer wave's pipeline over batches (lane l < 16 = row l of a batch; the other lanes carry channels only):
//   fetch(b)    request the sample coordinates + point encodings of batch b            (one memory latency)
//   resolve(b)  taps of both kinds from those, then REQUEST the 16 x 4 full-resolution texel rows of the batch, lane = channel
//   commit(b)   records -> LDS, the texels (landed by now) blended -> the K operand in LDS
// run as  commit(n + 1), resolve(n + 2), fetch(n + 3)  while the owners multiply batch n: no request is waited for in the
// trip that made it.  (One batch at a time, coordinates -> taps -> texels was a chain of two memory latencies per batch, and
// with the contraction on the MFMA the producer, not the owners, set the pace: 6.5 us per batch.)
struct Producer {
    const float* __restrict__ map3; const float* __restrict__ pixel_val; const float* __restrict__ sec_grid; const float* __restrict__ pe6;
    int H, W, V, R, S, ray0; long long nrows2; NodeGrid ng; unsigned npi; int l;
    // fetched
    f32x2 gc; float p0, p1, p2; long long frow; int fimg; bool fown, flive;
    // resolved
    unsigned toff[4]; float tw[4]; float m3w[4]; long long orow; float q0, q1, q2, one;
    float v[EB][4];

    __device__ __forceinline__ void fetch(long long bt) {
        if (l < EB) {
            const long long g2 = bt * EB + l;
            flive = g2 < nrows2 && g2 >= 0;
            const long long row = (flive ? g2 : 0) >> 1;
            const int j = (int)(g2 & 1);
            const int T = V * S;
            const int t = (int)(row / T);
            const int rem = (int)(row - (long long)t * T), vv = rem / S, s = rem - vv * S;
            const long long ray = (long long)ray0 + t;
            const int bb = (int)(ray / R), r = (int)(ray - (long long)bb * R);
            const size_t sidx = (((size_t)(bb * V + vv)) * R + r) * S + s;
            fown = j == 0;
            gc = *reinterpret_cast<const f32x2*>((fown ? pixel_val : sec_grid) + sidx * 2);
            const float* pe = pe6 + sidx * 6 + j * 3;
            p0 = pe[0]; p1 = pe[1]; p2 = pe[2];
            fimg = fown ? bb * V + vv : bb * V + (V - 1 - vv);
            frow = row;
        }
    }
    __device__ __forceinline__ void resolve() {
        unsigned m3o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) m3w[k] = 0.0f;
        if (l < EB) {
            const unsigned base = (unsigned)fimg * npi + (fown ? 0u : (unsigned)ng.border_nodes());
            const TapF tf = node_taps_f32(gc[0], gc[1], ng, fown, base);
            const Taps t3 = make_taps(gc[0], gc[1], W, H, fown);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                toff[k] = tf.off[k] * 4u;
                tw[k] = flive ? tf.w[k] : 0.0f;
                m3o[k] = ((unsigned)fimg * (unsigned)(H * W) + (unsigned)t3.off[k]) * 64u;
                m3w[k] = flive ? t3.w[k] : 0.0f;
            }
            orow = flive ? frow : -1;
            q0 = flive ? p0 : 0.0f; q1 = flive ? p1 : 0.0f; q2 = flive ? p2 : 0.0f; one = flive ? 1.0f : 0.0f;
        }
        const int c = l & 63;
#pragma unroll
        for (int r2 = 0; r2 < EB; ++r2)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned o = (unsigned)__shfl((int)m3o[k], r2);
                v[r2][k] = map3[(size_t)o + c];
            }
    }
    __device__ __forceinline__ void commit(BatchLds& b) {
        if (l < EB) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { b.toff[l][k] = toff[k]; b.tw[l][k] = tw[k]; }
            b.orow[l] = orow;
            b.xsT[64][l] = q0; b.xsT[65][l] = q1; b.xsT[66][l] = q2; b.xsT[67][l] = one;
        }
        const int c = l & 63;
#pragma unroll
        for (int r2 = 0; r2 < EB; ++r2) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) a += __shfl(m3w[k], r2) * v[r2][k];
            b.xsT[c][r2] = a;
        }
    }
};

constexpr int EWAVES = 8;                // 7 owner waves (26 channel tiles of 16) + 1 producer wave
constexpr int ECH = 416;                 // channels per workgroup: half of the layer
constexpr int ETILES = ECH / 16;         // 26
constexpr int WLD = 432;                 // LDS row stride of the weight block in floats (= 16 mod 32: the four k groups of an
                                         // A-operand read fall on disjoint bank halves)
constexpr int EF32_LDS = KX * WLD * 4 + 2 * (int)sizeof(BatchLds);
__global__ __launch_bounds__(64 * EWAVES, 1) void encode_hidden_f32_kernel(
    const float* __restrict__ tab, const float* __restrict__ map3, int H, int W, const float* __restrict__ pixel_val,
    const float* __restrict__ sec_grid, const float* __restrict__ pe6, const float* __restrict__ w80t, int V, int R, int S,
    int ray0, long long nrows2, long long nbatches, __half* __restrict__ hs) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* const wl = reinterpret_cast<float*>(smem_raw);               // [k][WLD]: this workgroup's half of the K = 68 block
    BatchLds* const lds = reinterpret_cast<BatchLds*>(smem_raw + KX * WLD * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave == EWAVES - 1;
    const int half = blockIdx.x & 1;                                    // channels [half * 416, half * 416 + 416)
    const int npairs = (gridDim.x + 1) >> 1, pair = blockIdx.x >> 1;
    const NodeGrid ng{W >> 1, H >> 1};
    const unsigned npi = (unsigned)ng.nodes_per_image();
    const long long per = (nbatches + npairs - 1) / npairs;
    const long long b_lo = (long long)pair * per, b_hi = (b_lo + per < nbatches) ? b_lo + per : nbatches;
    for (int i = tid; i < ECH * KX; i += 64 * EWAVES) {
        const int k = i / ECH, c = i - k * ECH;
        wl[k * WLD + c] = w80t[(size_t)k * 832 + half * ECH + c];
    }
    if (b_lo >= b_hi) return;
    Producer P{map3, pixel_val, sec_grid, pe6, H, W, V, R, S, ray0, nrows2, ng, npi, lane};
    if (producer) {
        P.fetch(b_lo);
        P.resolve();
        P.commit(lds[0]);
        P.fetch(b_lo + 1);                  // (batches past the end resolve to dead rows: nothing is stored for them)
        P.resolve();
        P.fetch(b_lo + 2);
    }
    __syncthreads();
    // MFMA roles (v_mfma_f32_16x16x4_f32, an fmaf chain bit for bit): A = weights (16 channels x 4 k), B = the K operand (4 k x 16
    // rows); lane (n = lane & 15, g = lane >> 4) supplies A[ch = n][k = g], B[k = g][row = n] and receives channels 4g .. 4g+3
    // of row n.  Wave w owns the channel tiles 4w .. 4w+3 (wave 6: two tiles).
    // A wave's four tiles interleave over its 64 channels: tile t holds the channels 16 q + 4 t + i (q, i < 4) of the wave's block,
    // so a lane's 4 x 4 results are 16 CONSECUTIVE channels (32 bytes of hi, 32 of lo; the four lanes of a row one 128-byte
    // line each) - with tiles of 16 consecutive channels a store instruction wrote 32-byte pieces per row: 20 of the kernel's
    // 53 ms per image were its stores.  Wave 6 owns 32 channels (two such tiles of 8-channel blocks).
    const int n = lane & 15, g = lane >> 4;
    const int wbase = wave * 64;                                       // first channel of the wave's block inside the half
    const int ntl = producer ? 0 : (ECH - wbase >= 64 ? 4 : 2);        // tiles the wave owns; its block is 16 * ntl channels wide
    const int cpl = ntl * 4;                                           // consecutive channels per lane
    const char* const tbase = reinterpret_cast<const char*>(tab);
    for (long long bt = b_lo; bt < b_hi; ++bt) {
        const int cur = (int)((bt - b_lo) & 1);
        if (producer) {
            // batch n + 1 is committed, n + 2 resolved, n + 3 fetched while the seven other waves multiply batch n
            P.commit(lds[cur ^ 1]);
            P.resolve();
            P.fetch(bt + 3);
        } else {
            const BatchLds& L = lds[cur];
            // the table taps of this lane's row for its 4 x 4 channels: 16 x 16-byte loads, in flight under the contraction
            f32x4 tv[4][4];
            float twn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) twn[k] = L.tw[n][k];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int tt = t < ntl ? t : 0;                          // (dead tiles of wave 6 shadow its first one)
                const unsigned cb = (unsigned)(half * ECH + wbase + g * cpl + tt * 4) * 4u;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tv[t][k] = (CPN_EF32_ABLATE & 8) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                                     : *reinterpret_cast<const f32x4*>(tbase + (((CPN_EF32_ABLATE & 1) ? 0u : L.toff[n][k]) + cb));
            }
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < ((CPN_EF32_ABLATE & 2) ? 1 : KX / 4); ++s4) {
                const float bx = L.xsT[4 * s4 + g][n];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tt = t < ntl ? t : 0;
                    const float aw = wl[(4 * s4 + g) * WLD + wbase + (n >> 2) * cpl + tt * 4 + (n & 3)];     // A row n = channel 4 q + i
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bx, acc[t], 0, 0, 0);
                }
            }
            const long long row = L.orow[n];
            const int j = n & 1;                                         // bt * EB is even: the batch's rows alternate own / other
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t >= ntl || row < 0 || ((CPN_EF32_ABLATE & 4) && acc[t][0] != 123.f)) continue;
                f32x4 v = acc[t];
#pragma unroll
                for (int k = 0; k < 4; ++k) v += tv[t][k] * twn[k];
                half4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r = fmaxf(v[i], 0.0f);
                    hi[i] = (_Float16)r;
                    lo[i] = (_Float16)(r - (float)hi[i]);
                }
                __half* o = hs + (size_t)row * 3328 + j * 832 + half * ECH + wbase + g * cpl + t * 4;
                *reinterpret_cast<half4*>(o) = hi;                      // (non-temporal stores: 156 against 120 ms per image)
                *reinterpret_cast<half4*>(o + 1664) = lo;
            }
        }
        __syncthreads();
    }
}

constexpr int HCF = 1664;

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f2(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one workgroup per ray: logits from (rows,128) fp32 operands, softmax over the V*S rows, hbar = sum_rows w (hi + lo)
__global__ __launch_bounds__(256) void attend_hidden_f32_kernel(const float* __restrict__ qa, const float* __restrict__ qb,
                                                                const __half* __restrict__ hs, int V, int R, int S, int ray0,
                                                                float* __restrict__ hbar, float* __restrict__ at_wt) {
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
    lmax = wave_max_f(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    const float gmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float lsum = 0.f;
    for (int row = tid; row < T; row += 256) {
        const float e = __expf(wts[row] - gmax);
        wts[row] = e;
        lsum += e;
    }
    lsum = wave_sum_f2(lsum);
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
    if (tid < HCF / 8) {                                       // thread = 8 hidden channels, rows streamed
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const __half* hp = hs + row0 * 3328 + tid * 8;
#pragma unroll 8
        for (int row = 0; row < T; ++row) {
            const half8 hi = __builtin_nontemporal_load(reinterpret_cast<const half8*>(hp + (size_t)row * 3328));
            const half8 lo = __builtin_nontemporal_load(reinterpret_cast<const half8*>(hp + (size_t)row * 3328 + 1664));
            const float w = wts[row];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * ((float)hi[e] + (float)lo[e]);
        }
        float* o = hbar + (size_t)lray * HCF + tid * 8;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
}

}  // namespace

extern "C" int cpn_node_features_f32(const float* map0, const float* map1, const float* map2, int H, int W, int nimg, float* feat,
                                     void* stream) {
    CPN_REQUIRE(map0 && map1 && map2 && feat, CPN_E_ARG, "cpn_node_features_f32: null pointer");
    CPN_REQUIRE(nimg > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0, CPN_E_SHAPE,
                "cpn_node_features_f32: need H,W multiples of 16 (got H=%d W=%d)", H, W);
    CPN_REQUIRE(((uintptr_t)map0 % 16) == 0 && ((uintptr_t)map1 % 16) == 0 && ((uintptr_t)map2 % 16) == 0 && ((uintptr_t)feat % 16) == 0,
                CPN_E_ARG, "cpn_node_features_f32: pointers must be 16-byte aligned");
    const NodeGrid ng{W >> 1, H >> 1};
    const long long total = (long long)nimg * ng.nodes_per_image() * 192;
    hipLaunchKernelGGL(node_features_f32_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, map0, map1, map2, H, W,
                       total, feat);
    CPN_LAUNCH_CHECK("cpn_node_features_f32");
    return 0;
}

extern "C" int cpn_encode_hidden_f32(const float* tab, const float* map3, int H, int W, const float* pixel_val, const float* sec_grid,
                                     const float* pe6, const float* w80t, int B, int V, int R, int S, int ray0, int nrays,
                                     uint16_t* hs, void* stream) {
    CPN_REQUIRE(tab && map3 && pixel_val && sec_grid && pe6 && w80t && hs, CPN_E_ARG, "cpn_encode_hidden_f32: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0, CPN_E_SHAPE,
                "cpn_encode_hidden_f32: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_encode_hidden_f32: ray range [%d,%d) outside B*R=%lld", ray0, ray0 + nrays, (long long)B * R);
    const NodeGrid ng{W >> 1, H >> 1};
    CPN_REQUIRE((long long)B * V * ng.nodes_per_image() * TABF * 4 < (1LL << 32) && (long long)B * V * H * W * 64 < (1LL << 32), CPN_E_SHAPE,
                "cpn_encode_hidden_f32: tables / map exceed the 32-bit element offsets");
    CPN_REQUIRE(((uintptr_t)tab % 8) == 0 && ((uintptr_t)w80t % 8) == 0 && ((uintptr_t)hs % 4) == 0 && ((uintptr_t)pixel_val % 8) == 0 &&
                    ((uintptr_t)sec_grid % 8) == 0, CPN_E_ARG, "cpn_encode_hidden_f32: pointers must be 8-byte aligned");
    const long long nrows2 = (long long)nrays * V * S * 2;
    const long long nbatches = (nrows2 + EB - 1) / EB;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)encode_hidden_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EF32_LDS);
        if (e != hipSuccess) {
            cpn_set_error("cpn_encode_hidden_f32: cannot reserve %d B of LDS: %s", EF32_LDS, hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int num_cu = cpn_stream_cus(stream);
    // workgroups 2p, 2p + 1 = the two channel halves of batch range p
    const unsigned grid = 2u * (unsigned)std::min<long long>(std::max(1, num_cu / 2), nbatches);
    hipLaunchKernelGGL(encode_hidden_f32_kernel, dim3(grid), dim3(64 * EWAVES), EF32_LDS, (hipStream_t)stream, tab, map3, H, W, pixel_val,
                       sec_grid, pe6, w80t, V, R, S, ray0, nrows2, nbatches, (__half*)hs);
    CPN_LAUNCH_CHECK("cpn_encode_hidden_f32");
    return 0;
}

extern "C" int cpn_attend_hidden_f32(const float* qa, const float* qb, const uint16_t* hs, int B, int V, int R, int S, int ray0,
                                     int nrays, float* hbar, float* at_wt, void* stream) {
    CPN_REQUIRE(qa && qb && hs && hbar, CPN_E_ARG, "cpn_attend_hidden_f32: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 4096, CPN_E_SHAPE, "cpn_attend_hidden_f32: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_hidden_f32: ray range outside B*R");
    CPN_REQUIRE(((uintptr_t)qa % 16) == 0 && ((uintptr_t)qb % 16) == 0 && ((uintptr_t)hs % 16) == 0 && ((uintptr_t)hbar % 16) == 0,
                CPN_E_ARG, "cpn_attend_hidden_f32: operands must be 16-byte aligned");
    const size_t lds = (size_t)(V * S + 8) * sizeof(float);
    hipLaunchKernelGGL(attend_hidden_f32_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream, qa, qb, (const __half*)hs, V, R, S,
                       ray0, hbar, at_wt);
    CPN_LAUNCH_CHECK("cpn_attend_hidden_f32");
    return 0;
}
