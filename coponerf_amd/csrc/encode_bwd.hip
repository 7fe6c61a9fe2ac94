// Backward of the first encoder layer IN ITS TABLE FORM (round 3; training, BASELINE config 3).
//
// Forward (csrc/encode.hip): hid[row] = ReLU( sum_{t<4} a_t T[node_t] + W[:,768:835] . [gather_3 (64) | tanh(pt/5) (3)] + b )
// with the node tables T = node_features . W[:, :768]^T.  Round 2 differentiated the layer in its ORIGINAL form
// (/root/reference models/CoPoNeRF.py:312, 370, 384-397): re-gather the 835-channel rows (3.0 ms), weight gradient over
// 4.2 M rows x 896 (9.1 ms), data gradient GEMM (6.2 ms), scatter of 832 columns into the four maps (9.1 ms).  In the
// table form the 768 coarse channels never exist per row:
//
//   cpn_scatter_rows_tables   dT[node] += a_t * d[row]            (832-wide rows into the fp32 table gradient; the same
//                             wave-owned LDS tiles as cpn_gather_rows_bwd: no atomic contention)
//   (host) dW[:, :768] = dT^T . node_features,  dfeat = dT . W[:, :768]      two small GEMMs over 0.28 M nodes
//   cpn_node_features_bwd     dfeat -> the three coarse maps (adjoint of cpn_node_features, as a gather: no atomics)
//   cpn_gather_tail           [gather_3 | tanh(pt/5) | 1] per row, 128 wide, for the weight / bias gradient of the
//                             K = 80 tail (the level-3 data gradient goes through cpn_gather_rows_bwd_level3)
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "taps.h"

namespace {

constexpr int PAD = CPN_NODE_PAD;
constexpr int TLD = CPN_TAB_LD;                    // 832

struct NodeGridB {                                  // same geometry as encode.hip's NodeGrid
    int Mx, My;
    __host__ __device__ int w(int kind) const { return Mx + 1 + (kind ? 2 * PAD : 0); }
    __host__ __device__ int h(int kind) const { return My + 1 + (kind ? 2 * PAD : 0); }
    __host__ __device__ long long border_nodes() const { return (long long)(Mx + 1) * (My + 1); }
    __host__ __device__ long long zeros_nodes() const { return (long long)(Mx + 1 + 2 * PAD) * (My + 1 + 2 * PAD); }
    __host__ __device__ long long per_image() const { return border_nodes() + zeros_nodes(); }
};

// table coordinates (>= 0) of the node cell of sample coordinate g: EXACTLY node_taps() of encode.hip
__device__ __forceinline__ void node_cell(float2 g, int kind, const NodeGridB ng, int& xi, int& yi, float& fx, float& fy) {
    const int pad = kind ? PAD : 0;
    float tx = (g.x + 1.0f) * (0.5f * (float)ng.Mx), ty = (g.y + 1.0f) * (0.5f * (float)ng.My);
    tx = fminf(fmaxf(tx, (float)-pad), (float)(ng.Mx + pad));
    ty = fminf(fmaxf(ty, (float)-pad), (float)(ng.My + pad));
    const int x0 = min((int)floorf(tx), ng.Mx + pad - 1), y0 = min((int)floorf(ty), ng.My + pad - 1);
    fx = tx - (float)x0;
    fy = ty - (float)y0;
    xi = x0 + pad;
    yi = y0 + pad;
}

// rows that read image `img` (same enumeration as backward.hip): idx in [0, per) own view (kind 0, pixel_val),
// [per, 2 per) other view (kind 1, sec_grid)
struct RowRefT {
    unsigned row;
    float2 g;
    int j;
};
__device__ __forceinline__ RowRefT row_of_t(int idx, int per, int S, int rlo, int b, int vi, int V, int R, int ray0,
                                            const float* __restrict__ pixel_val, const float* __restrict__ sec_grid) {
    RowRefT o;
    o.j = idx >= per;
    const int rem = idx - o.j * per;
    const int rr = rem / S, sm = rem - rr * S;
    const int r = rlo + rr;
    const int v = o.j ? (V - 1 - vi) : vi;
    const size_t sidx = (((size_t)(b * V + v)) * R + r) * S + sm;
    o.g = *reinterpret_cast<const float2*>((o.j ? sec_grid : pixel_val) + sidx * 2);
    o.row = ((((unsigned)(b * R + r - ray0)) * V + v) * S + sm) * 2 + o.j;
    return o;
}

// per 64-row chunk: the node-cell bounding box of its kind-0 rows and of its kind-1 rows (a chunk holds both kinds only
// where it straddles idx = per)
__global__ __launch_bounds__(256) void table_bbox_kernel(int H, int W, const float* __restrict__ pixel_val,
                                                         const float* __restrict__ sec_grid, int V, int R, int S, int ray0,
                                                         int nrays, int maxchunks, int nimg, int4* __restrict__ bbox) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int img = wid / maxchunks, c = wid - img * maxchunks;
    if (img >= nimg) return;
    const int b = img / V, vi = img - b * V;
    const int rlo = max(ray0, b * R) - b * R, rhi = min(ray0 + nrays, (b + 1) * R) - b * R;
    const int per = max(rhi - rlo, 0) * S, total = 2 * per;
    const int idx = c * 64 + lane;
    const NodeGridB ng{W >> 1, H >> 1};
    int box[2][4] = {{1 << 30, 1 << 30, -(1 << 30), -(1 << 30)}, {1 << 30, 1 << 30, -(1 << 30), -(1 << 30)}};
    if (idx < total) {
        const RowRefT rf = row_of_t(idx, per, S, rlo, b, vi, V, R, ray0, pixel_val, sec_grid);
        int xi, yi;
        float fx, fy;
        node_cell(rf.g, rf.j, ng, xi, yi, fx, fy);
        box[rf.j][0] = xi; box[rf.j][1] = yi; box[rf.j][2] = xi + 1; box[rf.j][3] = yi + 1;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            box[k][0] = min(box[k][0], __shfl_xor(box[k][0], o)); box[k][1] = min(box[k][1], __shfl_xor(box[k][1], o));
            box[k][2] = max(box[k][2], __shfl_xor(box[k][2], o)); box[k][3] = max(box[k][3], __shfl_xor(box[k][3], o));
        }
        if (lane == 0) bbox[((size_t)img * maxchunks + c) * 2 + k] = make_int4(box[k][0], box[k][1], box[k][2], box[k][3]);
    }
}

constexpr int TP = 8, TPY = 4;      // tile of 8 x 4 nodes
constexpr int TC = 64;              // channels per slice = lanes
constexpr int SW = TLD / TC;        // 13 waves per workgroup: one per 64-channel slice of the 832-wide rows
constexpr int QCAP = 2048;          // shared descriptor queue (32 KB)
constexpr int NB = 16;              // rows per drain batch

struct ScatterPlan {
    int kind, tiles_x, tiles, G, maxchunks;
    int scan_only;                  // timing-only ablation (CPN_SCATTER_SCAN_ONLY=1 in the environment): rows are queued, not accumulated
};

// ONE WORKGROUP of 13 waves owns one (image, kind, 8x4-node tile): wave w accumulates the 64-channel slice w of the rows
// whose cell touches the tile in its own 8 KB fp32 LDS tile (lane = channel, no atomics).  The row search is done ONCE for
// the 13 slices: 832 chunk boxes are tested per step (thread = chunk), the candidate chunks are handed out one per wave
// (lane = row) and the rows that really touch the tile go into ONE shared queue, which all 13 waves then drain for their
// slice (16 rows at a time, the next 16 gradient slices in flight) — the 13 x 128 bytes of a row are read by waves that
// run together.  With one wave per (tile, slice) (first version: 9.6 ms per step) every slice repeated the search
// (3.4 ms of the 9.6) and read its 128 bytes of a row on its own.
__global__ __launch_bounds__(64 * SW) void scatter_tables_kernel(
    const __half* __restrict__ d, int ldx, int H, int W, const float* __restrict__ pixel_val,
    const float* __restrict__ sec_grid, int V, int R, int S, int ray0, int nrays, float* __restrict__ dtab,
    ScatterPlan plan, const int4* __restrict__ bbox) {
    __shared__ float tiles_lds[SW][TP * TPY * TC];
    __shared__ uint4 queue[QCAP];
    __shared__ int cand[64 * SW];
    __shared__ int ncand[SW], wcount[SW];
    __shared__ int qn_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = tiles_lds[wave];
    int role = blockIdx.x;
    const int kind = plan.kind, G = plan.G;
    const int tidx = role % plan.tiles; role /= plan.tiles;
    const int g = role % G;
    const int img = role / G;
    const int tx0 = (tidx % plan.tiles_x) * TP, ty0 = (tidx / plan.tiles_x) * TPY;
    const NodeGridB ng{W >> 1, H >> 1};
    const int nw = ng.w(kind), nh = ng.h(kind);
    const __half* dcol = d + wave * TC + lane;

#pragma unroll
    for (int i = 0; i < TP * TPY; ++i) tile[i * TC + lane] = 0.0f;
    if (threadIdx.x == 0) qn_s = 0;

    const int b = img / V, vi = img - b * V;
    const int rlo = max(ray0, b * R) - b * R, rhi = min(ray0 + nrays, (b + 1) * R) - b * R;
    const int per = max(rhi - rlo, 0) * S, total = 2 * per;
    // chunks that can hold rows of this kind
    const int c_lo = kind ? per / 64 : 0, c_hi = kind ? (total + 63) / 64 : (per + 63) / 64;
    const int nchunks = max(c_hi - c_lo, 0);
    const int cpg = (nchunks + G - 1) / G;
    const int c_begin = c_lo + g * cpg, c_end = min(c_hi, c_lo + (g + 1) * cpg);
    const int4* boxes = bbox + (size_t)img * plan.maxchunks * 2 + kind;
    __syncthreads();

    auto drain = [&](int n) {                                  // all waves, the same n queue entries, each for its slice
        if (plan.scan_only) return;
        __half cur[NB], nxt[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) nxt[u] = dcol[(size_t)queue[min(u, n - 1)].x * ldx];
        for (int i = 0; i < n; i += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < NB; ++u) nxt[u] = dcol[(size_t)queue[min(i + NB + u, n - 1)].x * ldx];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (i + u >= n) break;
                const float du = __half2float(cur[u]);
                const uint4 q = queue[i + u];
                const int pk = __builtin_amdgcn_readfirstlane((int)q.y);
                const float hfx = __uint_as_float(q.z), hfy = __uint_as_float(q.w);
                const int hx = (pk & 255) - 1, hy = ((pk >> 8) & 255) - 1;
                float* t = tile + (hy * TP + hx) * TC + lane;
                if (pk & (1 << 16)) t[0] += du * ((1.0f - hfx) * (1.0f - hfy));
                if (pk & (2 << 16)) t[TC] += du * (hfx * (1.0f - hfy));
                if (pk & (4 << 16)) t[TP * TC] += du * ((1.0f - hfx) * hfy);
                if (pk & (8 << 16)) t[TP * TC + TC] += du * (hfx * hfy);
            }
        }
    };

    for (int cb = c_begin; cb < c_end; cb += 64 * SW) {
        // A1: thread = chunk, conservative box test; wave w compacts the candidates among its 64 chunks into cand[w][..]
        //     in chunk order (a fixed layout whatever the wave timing: the per-node sums are formed in queue order, so
        //     the result does not depend on scheduling)
        {
            const int c = cb + (int)threadIdx.x;
            bool maybe = false;
            if (c < c_end) {
                const int4 bx = boxes[(size_t)c * 2];
                maybe = (bx.z >= tx0) && (bx.x < tx0 + TP) && (bx.w >= ty0) && (bx.y < ty0 + TPY);
            }
            const unsigned long long m = __ballot(maybe);
            const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (maybe) cand[wave * 64 + pos] = c;
            if (lane == 0) ncand[wave] = (int)__builtin_popcountll(m);
        }
        __syncthreads();
        int prefix[SW + 1];
        prefix[0] = 0;
#pragma unroll
        for (int w2 = 0; w2 < SW; ++w2) prefix[w2 + 1] = prefix[w2] + ncand[w2];
        const int totalc = prefix[SW];
        // A2: the candidates in that order, 13 at a time: ONE wave evaluates the 64 rows of a chunk (lane = row); the rows
        //     that touch the tile are appended to the shared queue in (candidate, row) order
        for (int base = 0; base < totalc; base += SW) {
            const int ci = base + wave;
            uint4 desc = make_uint4(0, 0, 0, 0);
            int flags = 0;
            if (ci < totalc) {
                int slot = 0;
#pragma unroll
                for (int w2 = 1; w2 < SW; ++w2) slot += (ci >= prefix[w2]) ? 1 : 0;
                const int c = cand[slot * 64 + (ci - prefix[slot])];
                const int idx = c * 64 + lane;
                if (idx < total) {
                    const RowRefT rf = row_of_t(idx, per, S, rlo, b, vi, V, R, ray0, pixel_val, sec_grid);
                    if (rf.j == kind) {
                        int xi, yi;
                        float fx, fy;
                        node_cell(rf.g, kind, ng, xi, yi, fx, fy);
                        const int hx = xi - tx0, hy = yi - ty0;
                        if (hx >= -1 && hx < TP && hy >= -1 && hy < TPY) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int xk = xi + (k & 1), yk = yi + (k >> 1);
                                const float wk = ((k & 1) ? fx : 1.0f - fx) * ((k >> 1) ? fy : 1.0f - fy);
                                const bool in_tile = (xk >= tx0) && (xk < tx0 + TP) && (yk >= ty0) && (yk < ty0 + TPY);
                                if (in_tile && wk != 0.0f) flags |= 1 << k;
                            }
                            desc.x = rf.row;
                            desc.y = (unsigned)((hx + 1) | ((hy + 1) << 8) | (flags << 16));
                            desc.z = __float_as_uint(fx);
                            desc.w = __float_as_uint(fy);
                        }
                    }
                }
            }
            const unsigned long long mask = __ballot(flags != 0);
            if (lane == 0) wcount[wave] = (int)__builtin_popcountll(mask);
            __syncthreads();
            int off = qn_s, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < SW; ++w2) {
                const int n2 = wcount[w2];
                off += (w2 < wave) ? n2 : 0;
                tot += n2;
            }
            if (flags) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                queue[off + pos] = desc;
            }
            __syncthreads();
            if (threadIdx.x == 0) qn_s += tot;
            __syncthreads();
            if (qn_s > QCAP - 64 * SW) {                       // uniform across the workgroup: the next group always fits
                drain(qn_s);
                __syncthreads();
                if (threadIdx.x == 0) qn_s = 0;
                __syncthreads();
            }
        }
        __syncthreads();
    }
    if (qn_s) drain(qn_s);

    float* m = dtab + ((size_t)img * ng.per_image() + (kind ? ng.border_nodes() : 0)) * TLD + wave * TC + lane;
#pragma unroll 4
    for (int pix = 0; pix < TP * TPY; ++pix) {
        const float v = tile[pix * TC + lane];
        const int gy = ty0 + (pix >> 3), gx = tx0 + (pix & 7);
        if (v != 0.0f && gy < nh && gx < nw) {
            if (G > 1) atomicAdd(m + ((size_t)gy * nw + gx) * TLD, v);
            else m[((size_t)gy * nw + gx) * TLD] = v;           // the tile's only writer: dtab is zero on entry
        }
    }
}

// ---- the same scatter with the rows BUCKETED by tile first (counting sort), default since it was measured ------------
// The search above costs every tile a pass over all chunk boxes plus the evaluation of every chunk whose box touches it
// (an epipolar line's 64 samples span many tiles), in barrier-separated phases with one workgroup per CU (104 KB of LDS
// tiles).  Sorting replaces the search: (1) count the rows per (image, kind, tile), (2) exclusive scan + a work list that
// splits tiles with more than WMAX rows, (3) write row index, the four LDS cells and the four weights of every (row, tile
// it touches) into the tile's bucket, (4) a 13-wave workgroup per work item streams its bucket and accumulates.
constexpr int WMAX = 2048;          // rows per work item
constexpr int NSUB = (TP + 1) * (TPY + 1);   // a tile's bucket is ordered by the row's cell position relative to the tile (-1 .. TP-1, -1 .. TPY-1)

struct BucketGeo {
    int V, R, S, ray0, nrays, nimg, H, W;
    int tiles_x[2], tiles[2], T;    // per kind; T = max(tiles) = bucket slots per (image, kind)
};

// the tiles a row's 2x2 node footprint touches: calls f(tile, cells, weights) with, for the four taps, the LDS cell of the
// tile (byte k of `cells`; taps outside the tile or of zero weight point at the dummy cell TP*TPY with weight 0)
template <class F>
__device__ __forceinline__ void for_each_touched_tile(const RowRefT& rf, int kind, const NodeGridB ng, const BucketGeo& geo, F f) {
    int xi, yi;
    float fx, fy;
    node_cell(rf.g, kind, ng, xi, yi, fx, fy);
    const int txa = xi / TP, txb = (xi + 1) / TP, tya = yi / TPY, tyb = (yi + 1) / TPY;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx) {
            if ((tx && txb == txa) || (ty && tyb == tya)) continue;
            const int tX = tx ? txb : txa, tY = ty ? tyb : tya;
            const int tx0 = tX * TP, ty0 = tY * TPY;
            unsigned cells = 0;
            f32x4 w4;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xk = xi + (k & 1), yk = yi + (k >> 1);
                const float wk = ((k & 1) ? fx : 1.0f - fx) * ((k >> 1) ? fy : 1.0f - fy);
                const bool hit = (xk >= tx0) && (xk < tx0 + TP) && (yk >= ty0) && (yk < ty0 + TPY) && wk != 0.0f;
                cells |= (unsigned)(hit ? (yk - ty0) * TP + (xk - tx0) : TP * TPY) << (8 * k);
                w4[k] = hit ? wk : 0.0f;
                any |= hit;
            }
            if (any) f(tY * geo.tiles_x[kind] + tX, (yi - ty0 + 1) * (TP + 1) + (xi - tx0 + 1), cells, w4);
        }
}

template <bool FILL>
__global__ __launch_bounds__(256) void bucket_rows_kernel(BucketGeo geo, const float* __restrict__ pixel_val,
                                                          const float* __restrict__ sec_grid, int maxrows,
                                                          int* __restrict__ counts, const int* __restrict__ offsets,
                                                          unsigned* __restrict__ rows, unsigned* __restrict__ cellsv,
                                                          f32x4* __restrict__ wts) {
    const int img = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = img / geo.V, vi = img - b * geo.V;
    const int rlo = max(geo.ray0, b * geo.R) - b * geo.R, rhi = min(geo.ray0 + geo.nrays, (b + 1) * geo.R) - b * geo.R;
    const int per = max(rhi - rlo, 0) * geo.S, total = 2 * per;
    if (idx >= total || idx >= maxrows) return;
    const RowRefT rf = row_of_t(idx, per, geo.S, rlo, b, vi, geo.V, geo.R, geo.ray0, pixel_val, sec_grid);
    const NodeGridB ng{geo.W >> 1, geo.H >> 1};
    const int group = img * 2 + rf.j;
    for_each_touched_tile(rf, rf.j, ng, geo, [&](int tile, int sub, unsigned cells, const f32x4& w4) {
        const int slot = (group * geo.T + tile) * NSUB + sub;
        const int pos = atomicAdd(counts + slot, 1);
        if (FILL) {
            const size_t at = (size_t)offsets[slot] + pos;
            rows[at] = rf.row;
            cellsv[at] = cells;
            wts[at] = w4;
        }
    });
}

// Exclusive scan of the bucket sizes (per tile and cell position) and the work list per TILE (tile slot, first, last,
// shared?) in tile order, in three small launches: (1) a wave per tile sums its NSUB counts, (2) one workgroup scans the
// per-tile sums and work-item counts, (3) a wave per tile writes its NSUB offsets, clears its cursors and emits its work
// items.  (As ONE workgroup — every thread walking five tiles' 81 counts, thread 0 scanning 1024 partial sums, every thread
// writing 810 integers — this step took 0.72 ms of the training step for 420 k integers.)
__global__ __launch_bounds__(256) void bucket_tile_sums_kernel(const int* __restrict__ counts, int ntiles, int* __restrict__ tsum) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    int n = 0;
    for (int k = lane; k < NSUB; k += 64) n += counts[tile * NSUB + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if (lane == 0) tsum[tile] = n;
}

// tsum -> toff (exclusive row offsets per tile), woff (exclusive work-item offsets); offsets[ntiles * NSUB] = all rows
__global__ __launch_bounds__(1024) void bucket_tile_scan_kernel(const int* __restrict__ tsum, int ntiles, int* __restrict__ toff,
                                                                int* __restrict__ woff, int* __restrict__ offsets_end,
                                                                int* __restrict__ nwork) {
    __shared__ int part[1024], wpart[1024];
    const int per = (ntiles + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, ntiles);
    int s = 0, w = 0;
    for (int i = lo; i < hi; ++i) {
        const int n = tsum[i];
        s += n;
        w += (n + WMAX - 1) / WMAX;
    }
    part[threadIdx.x] = s;
    wpart[threadIdx.x] = w;
    __syncthreads();
    // inclusive Hillis-Steele scan over the 1024 partial sums
    for (int d = 1; d < 1024; d <<= 1) {
        const int a = threadIdx.x >= d ? part[threadIdx.x - d] : 0, c = threadIdx.x >= d ? wpart[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += a;
        wpart[threadIdx.x] += c;
        __syncthreads();
    }
    if (threadIdx.x == 1023) {
        *offsets_end = part[1023];
        *nwork = wpart[1023];
    }
    int a = part[threadIdx.x] - s, c = wpart[threadIdx.x] - w;                    // exclusive
    for (int i = lo; i < hi; ++i) {
        const int n = tsum[i];
        toff[i] = a;
        woff[i] = c;
        a += n;
        c += (n + WMAX - 1) / WMAX;
    }
}

__global__ __launch_bounds__(256) void bucket_tile_write_kernel(const int* __restrict__ counts, int ntiles,
                                                                const int* __restrict__ toff, const int* __restrict__ woff,
                                                                int* __restrict__ offsets, int* __restrict__ cursor,
                                                                int4* __restrict__ work) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (tile >= ntiles) return;
    static_assert(NSUB <= 128, "two counts per lane");
    // lane l owns cell positions 2 l and 2 l + 1: exclusive scan across the wave, then inside the pair
    const int k0 = 2 * lane, k1 = 2 * lane + 1;
    const int c0 = k0 < NSUB ? counts[tile * NSUB + k0] : 0, c1 = k1 < NSUB ? counts[tile * NSUB + k1] : 0;
    int incl = c0 + c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    const int a0 = toff[tile];
    const int excl = a0 + incl - (c0 + c1);
    if (k0 < NSUB) { offsets[tile * NSUB + k0] = excl; cursor[tile * NSUB + k0] = 0; }
    if (k1 < NSUB) { offsets[tile * NSUB + k1] = excl + c0; cursor[tile * NSUB + k1] = 0; }
    const int n = __shfl(incl, 63);
    int c = woff[tile];
    for (int st = lane * WMAX; st < n; st += 64 * WMAX)
        work[c + st / WMAX] = make_int4(tile, a0 + st, a0 + min(st + WMAX, n), n > WMAX);
}

// Everything about a row that is the same for the 64 lanes — its index, the four LDS cells and the four weights — is read
// through wave-uniform indices from read-only global arrays, i.e. with SCALAR loads into SGPRs: the vector unit is left
// with one address add per tap, one conversion and four FMAs per (row, wave).  (With the descriptors staged in LDS and
// unpacked by the vector unit the kernel spent 29 VALU + 11 LDS instructions per row and wave and was bound by them:
// 10 ms for 9.7 GB of reads — rocprofv3 counters of that version: VALU + LDS busy, L2 read latency only 684 clocks.)
// Round 4: a lane owns TWO channels (one 4-byte load per row), a wave 128, a workgroup 7 waves (the last one half
// idle: 832 = 6.5 x 128).  With one channel per lane every row was fetched as thirteen 128-byte requests issued by
// thirteen waves at thirteen different moments, from a row drawn at random out of 7 GB: 9.7 GB in 6.35 ms = 1.5 TB/s with
// the L1 -> L2 latency at a modest 695 clocks, L2 hit rate 9 %, no pipe busy - the DRAM pages are what that pattern wastes.
// 256-byte requests halve the number of pages touched per byte.
#ifndef CPN_BUCKET_CPL
#define CPN_BUCKET_CPL 2                                       // channels per lane: 2 (256-byte requests, 7 waves: 4.5 ms) or 4 (512-byte, 4 waves: 4.9 ms; 1 was 6.35)
#endif
constexpr int CPL = CPN_BUCKET_CPL;
constexpr int NBK = 64 / CPL;                                   // rows in flight per wave: the same 8 KiB either way
constexpr int BTC = 64 * CPL;                                  // channels per wave of bucket_accumulate_kernel
constexpr int BSW = (TLD + BTC - 1) / BTC;                     // waves per workgroup
__global__ __launch_bounds__(64 * BSW, 1) void bucket_accumulate_kernel(const __half* __restrict__ d, int ldx, BucketGeo geo,
                                                                     const int4* __restrict__ work, const int* __restrict__ nwork,
                                                                     const unsigned* __restrict__ rows,
                                                                     const unsigned* __restrict__ cellsv,
                                                                     const f32x4* __restrict__ wts, float* __restrict__ dtab) {
    typedef float fv __attribute__((ext_vector_type(CPL)));
    typedef _Float16 hv_t __attribute__((ext_vector_type(CPL)));
    typedef unsigned uv __attribute__((ext_vector_type(CPL / 2)));
    __shared__ fv tiles_lds[BSW][(TP * TPY + 1) * 64];           // 32 node cells + the dummy cell, lane = CPL channels
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    fv* tile = tiles_lds[wave] + lane;
    if ((int)blockIdx.x >= *nwork) return;      // (a persistent walk of the work list, one workgroup per CU, measured 2.7 ms slower)
    {
    const int4 item = work[blockIdx.x];
    const int group = item.x / geo.T, tidx = item.x - group * geo.T;
    const int img = group >> 1, kind = group & 1;
    const NodeGridB ng{geo.W >> 1, geo.H >> 1};
    const int nw = ng.w(kind), nh = ng.h(kind);
    const int tx0 = (tidx % geo.tiles_x[kind]) * TP, ty0 = (tidx / geo.tiles_x[kind]) * TPY;
    const int ch0 = wave * BTC + lane * CPL;
    const bool ch_ok = ch0 < TLD;                                 // the tail of the last wave has no channels (832 = 6.5 x 128)
    const uv* dcol = reinterpret_cast<const uv*>(d + (ch_ok ? ch0 : 0));                   // CPL fp16 values per row
    fv zero;
#pragma unroll
    for (int c = 0; c < CPL; ++c) zero[c] = 0.0f;
#pragma unroll
    for (int i = 0; i <= TP * TPY; ++i) tile[i * 64] = zero;

    const int first = item.y, last = item.z - 1;
    // The bucket is ordered by the rows' cell position, so consecutive rows mostly hit the SAME four cells: their sums
    // stay in registers and go to the LDS tile only when the position changes (the first version did four LDS
    // read-modify-writes per row and wave and was bound by the LDS pipe: SQ_ACTIVE_INST_LDS = 86 % of the kernel).
    unsigned held = 0xffffffffu;                                  // cells word of the run in the registers
    fv a0 = zero, a1 = zero, a2 = zero, a3 = zero;
    auto spill = [&]() {
        fv* t0 = tile + (held & 255u) * 64;
        fv* t1 = tile + ((held >> 8) & 255u) * 64;
        fv* t2 = tile + ((held >> 16) & 255u) * 64;
        fv* t3 = tile + (held >> 24) * 64;
        const fv v0 = *t0, v1 = *t1, v2 = *t2, v3 = *t3;         // the four cells of a position are distinct (or the dummy)
        *t0 = v0 + a0;
        *t1 = v1 + a1;
        *t2 = v2 + a2;
        *t3 = v3 + a3;
    };
    // Per-row descriptors (row index, cells word, four weights) are wave-uniform: lane u of the wave loads the descriptor
    // of row base + u with ordinary coalesced vector loads, two batches ahead, and v_readlane hands each row's values to
    // the scalar side when it is processed (through the scalar cache a batch of 32 rows needs 32 x 6 dwords of SGPRs).
    static_assert(NBK <= 64, "one descriptor per lane");
    struct Desc { unsigned row, cells; f32x4 w; };
    auto load_desc = [&](int base) {
        const int idx = min(base + (lane < NBK ? lane : NBK - 1), last);
        Desc dsc;
#ifdef CPN_BUCKET_ABLATE_SEQ                                   // timing only: rows read in memory order.  Round 6: 4.37 vs 4.47 ms inside
        // the training step - the random 1.6 KB rows are NOT what bounds this kernel any more (per-row issue work is: six
        // v_readlane, the address, the convert and four packed FMAs per row and wave at < 2 waves per SIMD)
        dsc.row = (unsigned)(idx % (2 * geo.nrays * geo.V * geo.S));
#else
        dsc.row = rows[idx];
#endif
        dsc.cells = cellsv[idx];
        dsc.w = wts[idx];
        // slots past the end of the item repeat its last row with ZERO weights: the row loop below then needs no early
        // exit and unrolls (with the `break` it stayed a loop: the rows of a batch were picked out of their registers
        // through s_set_gpr_idx, 28 instructions per row and wave where the unrolled form has 14 — the kernel is bound by
        // its issue slots, not by the rows' addresses: reading them in memory order changed nothing, HISTORY.md round 6)
        if (base + (lane < NBK ? lane : NBK - 1) > last) dsc.w = f32x4{0.f, 0.f, 0.f, 0.f};
        return dsc;
    };
    auto lane_u = [](unsigned v, int u) { return (unsigned)__builtin_amdgcn_readlane((int)v, u); };
    auto lane_f = [](float v, int u) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), u)); };
    const size_t ldw = (size_t)ldx / CPL;                         // row pitch in units of CPL fp16 values
    uv cur[NBK], nxt[NBK];
    Desc dcur = load_desc(first), dnxt = load_desc(first + NBK);
#pragma unroll
    for (int u = 0; u < NBK; ++u) nxt[u] = dcol[(size_t)lane_u(dcur.row, u) * ldw];
    for (int base = first; base <= last; base += NBK) {
        const Desc dfar = load_desc(base + 2 * NBK);              // descriptors of the batch after next
#pragma unroll
        for (int u = 0; u < NBK; ++u) cur[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < NBK; ++u) nxt[u] = dcol[(size_t)lane_u(dnxt.row, u) * ldw];     // rows past `last` repeat the last one
#pragma unroll
        for (int u = 0; u < NBK; ++u) {
            const hv_t hv = __builtin_bit_cast(hv_t, cur[u]);
            const fv du = __builtin_convertvector(hv, fv);
            const unsigned cells = lane_u(dcur.cells, u);
            if (cells != held) {                                  // wave-uniform
                if (held != 0xffffffffu) spill();
                held = cells;
                a0 = a1 = a2 = a3 = zero;
            }
            a0 += du * lane_f(dcur.w[0], u);
            a1 += du * lane_f(dcur.w[1], u);
            a2 += du * lane_f(dcur.w[2], u);
            a3 += du * lane_f(dcur.w[3], u);
        }
        dcur = dnxt;
        dnxt = dfar;
    }
    if (held != 0xffffffffu) spill();
    float* m = dtab + ((size_t)img * ng.per_image() + (kind ? ng.border_nodes() : 0)) * TLD + (ch_ok ? ch0 : 0);
#pragma unroll 4
    for (int pix = 0; pix < TP * TPY; ++pix) {
        const fv v = tile[pix * 64];
        const int gy = ty0 + (pix >> 3), gx = tx0 + (pix & 7);
        bool any = false;
#pragma unroll
        for (int c = 0; c < CPL; ++c) any = any || v[c] != 0.0f;
        if (ch_ok && gy < nh && gx < nw && any) {
            float* o = m + ((size_t)gy * nw + gx) * TLD;
            if (item.w) {                                          // the tile was split over several work items
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (v[c] != 0.0f) atomicAdd(o + c, v[c]);
            } else {
                *reinterpret_cast<fv*>(o) = v;                     // its only writer: dtab is zero on entry
            }
        }
    }
  }
}

// ---- adjoint of node_features_kernel as a gather: wave = one texel of one coarse level, lane = 4 channels ----------
// dmap[img, ty, tx, c] = sum over the nodes of both tables of the image whose grid_sample footprint at this level
// contains the texel, of weight * dfeat[node, lvl*256 + c].  The candidate nodes of texel tx are nx in
// [p (tx - 1/2), p (tx + 3/2)) (p = node pitch of the level's texels: 8, 4, 2) plus, at the map edge of the border table,
// everything beyond (clamped coordinates); each candidate is re-evaluated with the forward's own make_taps().
__global__ __launch_bounds__(256) void node_features_bwd_kernel(const float* __restrict__ dfeat, int H, int W, int nimg,
                                                                float* __restrict__ dmap0, float* __restrict__ dmap1,
                                                                float* __restrict__ dmap2, long long nwaves) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const long long t0 = (long long)nimg * (H >> 4) * (W >> 4), t1 = (long long)nimg * (H >> 3) * (W >> 3);
    const int lvl = wid < t0 ? 0 : (wid < t0 + t1 ? 1 : 2);
    long long rem = wid - (lvl == 0 ? 0 : (lvl == 1 ? t0 : t0 + t1));
    const int shift = 4 - lvl, Hl = H >> shift, Wl = W >> shift, p = 8 >> lvl;
    const int tx = (int)(rem % Wl); rem /= Wl;
    const int ty = (int)(rem % Hl);
    const int img = (int)(rem / Hl);
    const NodeGridB ng{W >> 1, H >> 1};
    const int texel = ty * Wl + tx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kind = 0; kind < 2; ++kind) {
        const int pad = kind ? PAD : 0, nw = ng.w(kind);
        int x_lo = max(p * tx - p, -pad), x_hi = min(p * tx + 2 * p, ng.Mx + pad);
        int y_lo = max(p * ty - p, -pad), y_hi = min(p * ty + 2 * p, ng.My + pad);
        if (!kind) {                                           // border padding: clamped coordinates pile up on the edge texels
            if (tx == 0) x_lo = 0;
            if (tx == Wl - 1) x_hi = ng.Mx;
            if (ty == 0) y_lo = 0;
            if (ty == Hl - 1) y_hi = ng.My;
        }
        const float* base = dfeat + ((size_t)img * ng.per_image() + (kind ? ng.border_nodes() : 0)) * 768 + lvl * 256 + lane * 4;
        // 64 candidate nodes are evaluated at once, one per lane (round 6: evaluated one after the other by the whole wave -
        // up to 1 250 make_taps per texel of the coarsest level - the kernel was bound by that redundant arithmetic: 0.64 ms);
        // the ones that touch the texel are then visited in the same (ny, nx) order as before, so the sums are unchanged
        const int wc = x_hi - x_lo + 1, ncand = wc * (y_hi - y_lo + 1);
        for (int c0 = 0; c0 < ncand; c0 += 64) {
            const int cand = c0 + lane;
            float wsum = 0.0f;
            int noff = 0;
            if (cand < ncand) {
                const int ny = y_lo + cand / wc, nx = x_lo + cand % wc;
                const float gx = (float)(2 * nx - ng.Mx) / (float)ng.Mx, gy = (float)(2 * ny - ng.My) / (float)ng.My;
                const Taps tp = make_taps(gx, gy, Wl, Hl, kind == 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) wsum += (tp.off[k] == texel) ? tp.w[k] : 0.0f;
                noff = (ny + pad) * nw + (nx + pad);
            }
            unsigned long long hit = __ballot(wsum != 0.0f);
            while (hit) {
                const int l = (int)__builtin_ctzll(hit);
                hit &= hit - 1;
                const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsum), l));
                const int o = __builtin_amdgcn_readlane(noff, l);
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)o * 768);
                acc += w * v;
            }
        }
    }
    float* out = (lvl == 0 ? dmap0 : lvl == 1 ? dmap1 : dmap2) + (((size_t)img * Hl + ty) * Wl + tx) * 256 + lane * 4;
    *reinterpret_cast<f32x4*>(out) = acc;
}

// ---- the K = 80 tail of the layer's input per row, 128 fp16 wide: [gather_3 (64) | tanh(pt/5) (3) | 1 | 0 x 60] ----------
// the arithmetic of encode_hidden_kernel's operand (4 texels blended in fp32 in tap order, one rounding to fp16)
__global__ __launch_bounds__(256) void gather_tail_kernel(const __half* __restrict__ map3, int H, int W,
                                                          const float* __restrict__ pixel_val,
                                                          const float* __restrict__ sec_grid, const float* __restrict__ pe6,
                                                          int V, int R, int S, int ray0, long long nrows,
                                                          __half* __restrict__ xt) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nrows * 16) return;
    const int piece = (int)(idx & 15);
    // 32-bit index arithmetic (rows < 2^31, checked by the entry): as 64-bit quotients and remainders these five were most of
    // the kernel's 265 VALU instructions per thread
    const unsigned row = (unsigned)(idx >> 4);
    const int j = (int)(row & 1u);
    unsigned t = row >> 1;
    const int s = (int)(t % (unsigned)S); t /= (unsigned)S;
    const int v = (int)(t % (unsigned)V); t /= (unsigned)V;
    const unsigned ray = t + (unsigned)ray0;
    const int b = (int)(ray / (unsigned)R), r = (int)(ray - (unsigned)b * (unsigned)R);
    const size_t sidx = (((size_t)(b * V + v)) * R + r) * S + s;
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)0.0f;
    if (piece < 8) {
        const float2 g = *reinterpret_cast<const float2*>((j ? sec_grid : pixel_val) + sidx * 2);
        const Taps t3 = make_taps(g.x, g.y, W, H, j == 0);
        const int img = b * V + (j ? V - 1 - v : v);
        const __half* m3 = map3 + (size_t)img * H * W * 64 + piece * 8;
        float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half8 tv = *reinterpret_cast<const half8*>(m3 + (size_t)(unsigned)t3.off[k] * 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) a8[e] = fmaf((float)tv[e], t3.w[k], a8[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)a8[e];
    } else if (piece == 8) {
        const float* pe = pe6 + sidx * 6 + j * 3;
        o[0] = (_Float16)pe[0]; o[1] = (_Float16)pe[1]; o[2] = (_Float16)pe[2];
        o[3] = (_Float16)1.0f;                                 // the bias column
    }
    *reinterpret_cast<half8*>(xt + (size_t)row * 128 + piece * 8) = o;
}

// ---- fp32 -> fp16 with a power-of-two scale chosen on the device: two passes over the data instead of the five of
//      abs / amax / mul / convert as separate tensor ops (the table gradient is 0.94 GB at batch 4)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long n4, unsigned* __restrict__ amax_bits) {
    float m = 0.0f;
    // four loads in flight per thread (one load per trip ran this pass at 2.2 TB/s: 0.42 ms over the 0.94 GB table gradient)
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u][0]), fabsf(v[u][1])), fmaxf(fabsf(v[u][2]), fabsf(v[u][3]))));
    }
    for (; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    // ONE atomic per workgroup: with one per wave of an 8 192-block grid the 32 768 atomics on the same word were the
    // kernel (164 us on a 33 MB tensor that streams in 10)
    __shared__ float wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        if (m == m) atomicMax(amax_bits, __float_as_uint(m));                            // non-negative floats order like their bits
    }
}

__device__ __forceinline__ float pow2_scale(unsigned amax_bits, float target) {
    const float amax = fmaxf(__uint_as_float(amax_bits), 1e-30f);
    return fminf(fmaxf(exp2f(floorf(log2f(target / amax))), 0x1p-40f), 0x1p40f);
}

__global__ __launch_bounds__(256) void scale_to_f16_kernel(const float* __restrict__ x, long long n4,
                                                           const unsigned* __restrict__ amax_bits, float target,
                                                           __half* __restrict__ y, float* __restrict__ scale_out) {
    const float s = pow2_scale(*amax_bits, target);
    if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = s; scale_out[1] = 1.0f / s; }   // a power of two: exact
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (_Float16)(v[u][e] * s);
            reinterpret_cast<half4*>(y)[i + u * stride] = o;
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(v[e] * s);
        reinterpret_cast<half4*>(y)[i] = o;
    }
}

}  // namespace

// y = fp16(x * s), s = 2^floor(log2(target / max|x|)) clamped to [2^-40, 2^40], written to scale_out[0] (1 / s to
// scale_out[1]); amax_scratch: one uint32, ZERO on entry.  n % 4 == 0, 16-byte aligned x and 8-byte aligned y.
extern "C" int cpn_scale_to_f16(const float* x, long long n, float target, uint32_t* amax_scratch, uint16_t* y,
                                float* scale_out, void* stream) {
    CPN_REQUIRE(x && amax_scratch && y && scale_out, CPN_E_ARG, "cpn_scale_to_f16: null pointer");
    CPN_REQUIRE(n > 0 && (n % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0 && target > 0.0f, CPN_E_SHAPE,
                "cpn_scale_to_f16: need n %% 4 == 0 and aligned pointers");
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(n / 4, 256), 8192);
    // four loads in flight per thread; at most two workgroups per CU: their atomics on the one word serialise (2 048 of them
    // were 10 of the kernel's 18 us on a 33 MB tensor)
    const unsigned ablocks = (unsigned)std::min<long long>(cpn_cdiv(n / 4, 1024), 512);
    hipLaunchKernelGGL(absmax_kernel, dim3(ablocks), dim3(256), 0, (hipStream_t)stream, x, n / 4, amax_scratch);
    hipLaunchKernelGGL(scale_to_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n / 4, amax_scratch, target,
                       (__half*)y, scale_out);
    CPN_LAUNCH_CHECK("cpn_scale_to_f16");
    return 0;
}

static void bucket_geo(BucketGeo& g, int H, int W, int B, int V, int R, int S, int ray0, int nrays) {
    const NodeGridB ng{W >> 1, H >> 1};
    g.V = V; g.R = R; g.S = S; g.ray0 = ray0; g.nrays = nrays; g.nimg = B * V; g.H = H; g.W = W;
    for (int k = 0; k < 2; ++k) {
        g.tiles_x[k] = (ng.w(k) + TP - 1) / TP;
        g.tiles[k] = g.tiles_x[k] * ((ng.h(k) + TPY - 1) / TPY);
    }
    g.T = std::max(g.tiles[0], g.tiles[1]);
}

// int32 entries of the scratch: [counts | cursor | offsets (+1) | nwork (4) | work items | descriptors], sized for the worst
// case (every row in four tiles) and for the search variant's chunk boxes
extern "C" long long cpn_scatter_tables_scratch(int H, int W, int B, int V, int R, int S) {
    if (H < 16 || W < 16 || B <= 0 || V <= 0 || R <= 0 || S <= 0) return -1;
    BucketGeo g;
    bucket_geo(g, H, W, B, V, R, S, 0, B * R);
    const long long ntiles = (long long)B * V * 2 * g.T, slots = ntiles * NSUB, rows = 2LL * B * V * R * S;
    const long long maxdesc = 4 * rows, maxwork = ntiles + maxdesc / WMAX + 1;
    const long long sorted = 3 * slots + 8 + 4 * maxwork + 6 * maxdesc + 16 + 3LL * ntiles;
    const long long boxes = (long long)B * V * ((2LL * R * S + 63) / 64) * 8;
    return std::max(sorted, boxes);
}

extern "C" int cpn_scatter_rows_tables(const uint16_t* d, int ldx, int H, int W, const float* pixel_val,
                                       const float* sec_grid, int B, int V, int R, int S, int ray0, int nrays, float* dtab,
                                       int32_t* scratch, void* stream) {
    CPN_REQUIRE(d && pixel_val && sec_grid && dtab && scratch, CPN_E_ARG, "cpn_scatter_rows_tables: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && (H % 16) == 0 && (W % 16) == 0 && ldx >= TLD, CPN_E_SHAPE,
                "cpn_scatter_rows_tables: bad shape");
    CPN_REQUIRE(H <= 1024 && W <= 1024, CPN_E_SHAPE, "cpn_scatter_rows_tables: maps larger than 1024 pixels a side");
    CPN_REQUIRE((ldx % CPL) == 0 && ((uintptr_t)d % (2 * CPL)) == 0 && ((uintptr_t)dtab % (4 * CPL)) == 0, CPN_E_ARG,
                "cpn_scatter_rows_tables: d must be %d-byte aligned with ldx a multiple of %d, dtab %d-byte aligned", 2 * CPL, CPL, 4 * CPL);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_scatter_rows_tables: ray range outside B*R");
    CPN_REQUIRE((long long)nrays * V * S * 2 < (1LL << 31) && ((uintptr_t)scratch % 16) == 0, CPN_E_SHAPE,
                "cpn_scatter_rows_tables: chunk too large / scratch not 16-byte aligned");
    const hipStream_t st = (hipStream_t)stream;
    const int nimg = B * V;
    static const bool use_search = getenv("CPN_SCATTER_SEARCH") != nullptr;      // the first implementation, kept for A/B timing
    if (!use_search) {
        BucketGeo g;
        bucket_geo(g, H, W, B, V, R, S, ray0, nrays);
        const int ntiles = nimg * 2 * g.T, slots = ntiles * NSUB;
        const long long rows = 2LL * B * V * R * S, maxdesc = 4 * rows, maxwork = ntiles + maxdesc / WMAX + 1;
        int* counts = scratch;
        int* cursor = counts + slots;
        int* offsets = cursor + slots;
        int* nwork = offsets + slots + 1;
        nwork += (4 - ((nwork - scratch) & 3)) & 3;                                // 16-byte alignment of what follows
        int4* work = reinterpret_cast<int4*>(nwork + 4);
        f32x4* wts = reinterpret_cast<f32x4*>(work + maxwork);                     // [maxdesc] 16-byte aligned
        unsigned* rows_a = reinterpret_cast<unsigned*>(wts + maxdesc);
        unsigned* cells_a = rows_a + maxdesc;
        hipError_t e = hipMemsetAsync(counts, 0, sizeof(int) * slots, st);
        CPN_REQUIRE(e == hipSuccess, (int)e, "cpn_scatter_rows_tables: memset failed: %s", hipGetErrorString(e));
        const int per_img = 2 * std::min(R, nrays) * S;
        dim3 grid(cpn_cdiv(per_img, 256), nimg);
        hipLaunchKernelGGL((bucket_rows_kernel<false>), grid, dim3(256), 0, st, g, pixel_val, sec_grid, per_img, counts,
                           (const int*)nullptr, (unsigned*)nullptr, (unsigned*)nullptr, (f32x4*)nullptr);
        int* tile_scratch = reinterpret_cast<int*>(cells_a + maxdesc);             // three ints per tile behind the descriptors
        int* tsum_a = tile_scratch, *toff_a = tile_scratch + ntiles, *woff_a = tile_scratch + 2 * (size_t)ntiles;
        hipLaunchKernelGGL(bucket_tile_sums_kernel, dim3(cpn_cdiv(ntiles, 4)), dim3(256), 0, st, (const int*)counts, ntiles, tsum_a);
        hipLaunchKernelGGL(bucket_tile_scan_kernel, dim3(1), dim3(1024), 0, st, (const int*)tsum_a, ntiles, toff_a, woff_a,
                           offsets + (size_t)ntiles * NSUB, nwork);
        hipLaunchKernelGGL(bucket_tile_write_kernel, dim3(cpn_cdiv(ntiles, 4)), dim3(256), 0, st, (const int*)counts, ntiles,
                           (const int*)toff_a, (const int*)woff_a, offsets, cursor, work);
        hipLaunchKernelGGL((bucket_rows_kernel<true>), grid, dim3(256), 0, st, g, pixel_val, sec_grid, per_img, cursor,
                           (const int*)offsets, rows_a, cells_a, wts);
        // the work-list length lives on the device: launch its upper bound, surplus workgroups return at once
        hipLaunchKernelGGL(bucket_accumulate_kernel, dim3((unsigned)maxwork), dim3(64 * BSW), 0, st,
                           (const __half*)d, ldx, g, (const int4*)work, (const int*)nwork, (const unsigned*)rows_a,
                           (const unsigned*)cells_a, (const f32x4*)wts, dtab);
        CPN_LAUNCH_CHECK("cpn_scatter_rows_tables");
        return 0;
    }
    int32_t* chunk_boxes = scratch;
    const int maxchunks = (int)((2LL * R * S + 63) / 64);
    const long long nwaves = (long long)nimg * maxchunks;
    hipLaunchKernelGGL(table_bbox_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, H, W, pixel_val, sec_grid, V,
                       R, S, ray0, nrays, maxchunks, nimg, (int4*)chunk_boxes);
    const NodeGridB ng{W >> 1, H >> 1};
    const long long cand = (long long)((nrays + B - 1) / B) * S;            // rows of one kind that read one image
    for (int kind = 0; kind < 2; ++kind) {
        ScatterPlan plan;
        plan.kind = kind;
        plan.scan_only = getenv("CPN_SCATTER_SCAN_ONLY") != nullptr;
        plan.maxchunks = maxchunks;
        plan.tiles_x = (ng.w(kind) + TP - 1) / TP;
        plan.tiles = plan.tiles_x * ((ng.h(kind) + TPY - 1) / TPY);
        const long long hits = cand / plan.tiles;
        plan.G = (int)std::min<long long>(64, std::max<long long>(1, (hits + 1023) / 2048));
        const long long nroles = (long long)nimg * plan.G * plan.tiles;               // workgroups of 13 waves
        CPN_REQUIRE(nroles < (1LL << 31), CPN_E_SHAPE, "cpn_scatter_rows_tables: too many roles");
        hipLaunchKernelGGL(scatter_tables_kernel, dim3((unsigned)nroles), dim3(64 * SW), 0, st, (const __half*)d, ldx, H, W,
                           pixel_val, sec_grid, V, R, S, ray0, nrays, dtab, plan, (const int4*)chunk_boxes);
    }
    CPN_LAUNCH_CHECK("cpn_scatter_rows_tables");
    return 0;
}

extern "C" int cpn_node_features_bwd(const float* dfeat, int H, int W, int nimg, float* dmap0, float* dmap1, float* dmap2,
                                     void* stream) {
    CPN_REQUIRE(dfeat && dmap0 && dmap1 && dmap2, CPN_E_ARG, "cpn_node_features_bwd: null pointer");
    CPN_REQUIRE(nimg > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0, CPN_E_SHAPE,
                "cpn_node_features_bwd: need H,W multiples of 16 (got H=%d W=%d)", H, W);
    CPN_REQUIRE(((uintptr_t)dfeat % 16) == 0 && ((uintptr_t)dmap0 % 16) == 0 && ((uintptr_t)dmap1 % 16) == 0 &&
                    ((uintptr_t)dmap2 % 16) == 0, CPN_E_ARG, "cpn_node_features_bwd: pointers must be 16-byte aligned");
    const long long nwaves = (long long)nimg * ((long long)(H >> 4) * (W >> 4) + (long long)(H >> 3) * (W >> 3) +
                                                (long long)(H >> 2) * (W >> 2));
    hipLaunchKernelGGL(node_features_bwd_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dfeat,
                       H, W, nimg, dmap0, dmap1, dmap2, nwaves);
    CPN_LAUNCH_CHECK("cpn_node_features_bwd");
    return 0;
}

extern "C" int cpn_gather_tail(const uint16_t* map3, int H, int W, const float* pixel_val, const float* sec_grid,
                               const float* pe6, int B, int V, int R, int S, int ray0, int nrays, uint16_t* xt,
                               void* stream) {
    CPN_REQUIRE(map3 && pixel_val && sec_grid && pe6 && xt, CPN_E_ARG, "cpn_gather_tail: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H > 0 && W > 0, CPN_E_SHAPE, "cpn_gather_tail: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_gather_tail: ray range outside B*R");
    CPN_REQUIRE(((uintptr_t)map3 % 16) == 0 && ((uintptr_t)xt % 16) == 0, CPN_E_ARG, "cpn_gather_tail: pointers must be 16-byte aligned");
    CPN_REQUIRE((long long)nrays * V * S * 2 < (1LL << 31), CPN_E_SHAPE, "cpn_gather_tail: more than 2^31 rows");
    const long long nrows = (long long)nrays * V * S * 2;
    hipLaunchKernelGGL(gather_tail_kernel, dim3((unsigned)cpn_cdiv(nrows * 16, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)map3, H, W, pixel_val, sec_grid, pe6, V, R, S, ray0, nrows, (__half*)xt);
    CPN_LAUNCH_CHECK("cpn_gather_tail");
    return 0;
}
