// K2+K3a — first encoder layer straight from the feature maps: hid = ReLU(query_encode_latent([gather ‖ tanh(pt/5)])).
//
// Replaces F.grid_sample x 8 + torch.cat + the 835 -> 832 1x1 convolution + ReLU
// (/root/reference models/CoPoNeRF.py:312, 370, 384-397 with the layer of :71) WITHOUT materialising the gathered
// 835-channel encoder input.
//
// (1) A 1x1 convolution and bilinear interpolation commute (both linear; the bias is added after the interpolation, so
//     this holds for 'border' and for 'zeros' padding alike; SURVEY.md §7.4 tier B):  W . sum_t a_t tex_t = sum_t a_t (W . tex_t).
// (2) The three coarse levels (256 channels each at H/16, H/8, H/4, align_corners=False) have their texel centres at
//     u = (i + 1/2) / W_l of the unit square, i.e. at the nodes t = 8i+4, 4i+2, 2i+1 of ONE grid of spacing 1/M,
//     M = W/2.  Each level's interpolant is bilinear on every cell of that grid (a bilinear function restricted to a
//     sub-rectangle of its cell is still bilinear; the per-level clamps of 'border' padding sit on the nodes 4, 2, 1
//     and the zero rims of 'zeros' padding on -4, -2, -1), hence so is their projected SUM
//         F(u) = sum_{l<3} W[:, 256l:256l+256] . grid_sample_l(u),
//     and bilinear interpolation of F's node values reproduces it EXACTLY (in real arithmetic).
// So per stereo pair and image two node tables are built once (node_features_kernel + one cpn_gemm_f16):
//     T_border (M_y+1, M_x+1, 832)   nodes 0..M          (primary gather: own image, border padding)
//     T_zeros  (M_y+9, M_x+9, 832)   nodes -4..M+4       (secondary gather: other image, zeros padding)
// = 97 GFLOP and 127 MB per 256^2 pair instead of 4.5 TFLOP per 16 384 rays, and a row of the layer becomes
//
//     hid[row] = ReLU( sum_{t<4} a_t T[node_t]  +  W[:, 768:835] . [gather_3(64 ch) ‖ tanh(pt/5)]  +  b )
//
// i.e. 4 table taps (VALU, fp32 accumulation) + a K = 96 MFMA product for the full-resolution level, whose table
// would be 13x the map (218 MB per pair) and is kept as a contraction.  Numerically the node features are rounded to
// fp16 like the gathered rows of the GEMM form were, the table entry once more, the 4-tap sum runs in fp32.
//
// gfx950 design
//   * workgroup = 128 rows = 4 adjacent rays x 16 consecutive samples x {own image, other image} of one view, 4 waves,
//     two workgroups per CU; wave w walks the four 208-channel slices of the output with 2 x 13 accumulator tiles of
//     v_mfma_f32_16x16x32_f16 (weights as the A operand: a lane holds 4 consecutive channels of one row per tile).
//   * MFMA tile mt <-> j (own / other image); column r = (sample & 3)*4 + (ray & 3): ONE load instruction covers a
//     4 x 4 patch of (sample, ray) whose taps fall on a handful of nodes -> most 64-byte requests hit lines a
//     neighbouring lane just brought into the vector L1.
//   * the channel -> (tile, register) assignment is chosen so that a lane's 52 channels of a slice are 6 x 8
//     consecutive ones (k*32 + g*8 .. +8) + 4 (192 + g*4 ..): a table tap is 6 16-byte loads + 1 8-byte load per
//     lane, the 4 lanes of a row reading 64 contiguous, 64-byte aligned bytes (tables are stored with each
//     208-channel slice padded to 224 halves for that), and the fp16 row leaves as 16-byte stores.
//   * per-row tap offsets / weights are computed once per row (not per lane) into LDS; the full-resolution level and
//     the point encoding are gathered once per row into an LDS image in MFMA B-operand order; the bias sits in LDS.
//   * weight fragments of a slice (39 KiB) stream through LDS by buffer_load ... lds under the previous slice's taps;
//     the slice loop uses raw s_barriers (no vmcnt(0)): a wave never waits for its own hid stores there.
// Bound (rocprofv3 PMC, profiles/r02_*): the texture-address / vector-L1 path (64 B/clk/CU) for the taps and the
// 7 GB hid write stream per 16 384 rays; algorithmic FLOPs of the layer it replaces 2*835*832 per row.
#include <algorithm>

#include "common.h"
#include "taps.h"

// timing-only ablations for tools/encode_ablate.py (results are wrong when non-zero; the product builds with 0):
// 1 = no table taps, 2 = no hid stores, 4 = no MFMA phase, 16 = every tap reads node 0
#ifndef CPN_ENCODE_ABLATE
#define CPN_ENCODE_ABLATE 0
#endif

namespace {

constexpr int TILE_ROWS = 128;
constexpr int NSLICE = 13;                // 832 = 13 x 64 output channels
constexpr int SLICE_CH = 64;
constexpr int NT = 4;                     // 16-channel MFMA tiles per slice
constexpr int KSTEPS = 3;                 // K = 80 = 2 x 32 level-3 channels + a 16-wide tail (3 point encodings + zeros)
constexpr int PAD = CPN_NODE_PAD;         // zero rim of the 'zeros' table, in nodes (= level-0 texel pitch / 2)
constexpr int TAB_SLICE_BYTES = SLICE_CH * 2;                  // 128: one cache line per node and slice
constexpr int TAB_ROW_BYTES = CPN_TAB_LD * 2;                  // 1664 per node, channels in natural order
constexpr int AIMG_BYTES = 2 * 4 * 2 * 1024 + 4 * 2 * 256;     // [k < 2][wave][mt][lane] half8 + [wave][mt][r] half8 (k = 2, g = 0)
constexpr int AIMG2_OFF = 2 * 4 * 2 * 1024;                    // third K step: only lane group 0 holds data (pt enc)
constexpr int TG = 4;                     // rays per tile
constexpr int TSB = 16;                   // samples per tile

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct TapRec {
    int off[4];                           // byte offsets of the 4 nodes / texels inside the (image, mode) table / map
    float w[4];
};

// acc + f32(lo / hi half of `packed`) * w with the fp16 -> fp32 conversion inside the FMA
__device__ __forceinline__ float fma_mix_lo(float acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}
__device__ __forceinline__ float fma_mix_hi(float acc, unsigned packed, float w) {
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc) : "v"(packed), "v"(w));
    return acc;
}

// node-grid geometry of one image: border table first, zeros table behind it
struct NodeGrid {
    int Mx, My;                           // W/2, H/2
    __host__ __device__ int bw() const { return Mx + 1; }
    __host__ __device__ int bh() const { return My + 1; }
    __host__ __device__ int zw() const { return Mx + 1 + 2 * PAD; }
    __host__ __device__ int zh() const { return My + 1 + 2 * PAD; }
    __host__ __device__ long long border_nodes() const { return (long long)bw() * bh(); }
    __host__ __device__ long long zeros_nodes() const { return (long long)zw() * zh(); }
    __host__ __device__ long long nodes_per_image() const { return border_nodes() + zeros_nodes(); }
};

// the 4 nodes around normalised coordinate g (grid_sample convention, [-1,1] = image) and their bilinear weights
__device__ __forceinline__ TapRec node_taps(float gx, float gy, const NodeGrid ng, bool border) {
    const int pad = border ? 0 : PAD;
    const int nw = border ? ng.bw() : ng.zw();
    float tx = (gx + 1.0f) * (0.5f * (float)ng.Mx), ty = (gy + 1.0f) * (0.5f * (float)ng.My);
    // beyond the rim the function is constant (border: clamped; zeros: 0), and |g| can reach 1e10 (geometry.py:390-391)
    tx = fminf(fmaxf(tx, (float)-pad), (float)(ng.Mx + pad));
    ty = fminf(fmaxf(ty, (float)-pad), (float)(ng.My + pad));
    const int x0 = min((int)floorf(tx), ng.Mx + pad - 1), y0 = min((int)floorf(ty), ng.My + pad - 1);
    const float fx = tx - (float)x0, fy = ty - (float)y0;
    const int base = (y0 + pad) * nw + (x0 + pad);
    TapRec t;
    t.off[0] = base * TAB_ROW_BYTES;
    t.off[1] = (base + 1) * TAB_ROW_BYTES;
    t.off[2] = (base + nw) * TAB_ROW_BYTES;
    t.off[3] = (base + nw + 1) * TAB_ROW_BYTES;
    t.w[0] = (1.0f - fx) * (1.0f - fy);
    t.w[1] = fx * (1.0f - fy);
    t.w[2] = (1.0f - fx) * fy;
    t.w[3] = fx * fy;
    return t;
}

// Which (ray, sample) a tile row is.  A tile = TG adjacent rays (same batch element) x TSB consecutive samples x
// {own, other image} of one view; wave w takes samples 4w..4w+3 of the block, MFMA column r = (sample & 3)*4 + (ray & 3).
struct RowId {
    bool live;
    int j, s, r;                          // r = ray index inside its batch element
};
__device__ __forceinline__ RowId tile_row(int row, int rgroup, int blk, int S, int R, int b, int ray0, int nrays) {
    const int w = row >> 5, rl = row & 31, rs = rl >> 1;
    RowId o;
    o.j = rl & 1;
    o.s = blk * TSB + w * 4 + (rs >> 2);
    o.r = rgroup * TG + (rs & 3);
    const long long ray = (long long)b * R + o.r;
    o.live = (o.s < S) && (o.r < R) && ray >= ray0 && ray < (long long)ray0 + nrays;
    return o;
}

__global__ __launch_bounds__(256, 2) void encode_hidden_kernel(
    const __half* __restrict__ tab, const __half* __restrict__ map3, int H, int W,
    const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, const float* __restrict__ pe6,
    const half8* __restrict__ wfrag, const float* __restrict__ bias, int V, int R, int S, int ray0, int nrays,
    int nblk, int groups_per_b, long long group0, __half* __restrict__ hid) {
    __shared__ __attribute__((aligned(16))) TapRec taps[TILE_ROWS];          // table taps
    __shared__ __attribute__((aligned(16))) TapRec taps3[TILE_ROWS];         // full-resolution level
    __shared__ __attribute__((aligned(16))) char aimg[AIMG_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_s[832];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware order: blocks go round-robin over the 8 XCDs; each XCD takes a contiguous range of tiles
    // (= neighbouring rays = overlapping node footprints), so its private L2 works on 1/8 of the tables
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, rem = nb & 7;
    const unsigned tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    const int blk = (int)(tile % (unsigned)nblk);
    const int v = (int)((tile / (unsigned)nblk) % (unsigned)V);
    const long long gq = group0 + tile / (unsigned)(nblk * V);
    const int b = (int)(gq / groups_per_b), rgroup = (int)(gq % groups_per_b);
    const NodeGrid ng{W >> 1, H >> 1};

    // ---- phase A: tap records, thread = (row, {table, full-resolution level}) ---------------------------------
    for (int i = tid; i < 832 / 4; i += 256)
        *reinterpret_cast<f32x4*>(bias_s + i * 4) = *reinterpret_cast<const f32x4*>(bias + i * 4);
    {
        const int row = tid >> 1, half = tid & 1;
        const RowId id = tile_row(row, rgroup, blk, S, R, b, ray0, nrays);
        const size_t sidx = (((size_t)(b * V + v)) * R + min(id.r, R - 1)) * S + min(id.s, S - 1);
        const float2 gc = *reinterpret_cast<const float2*>((id.j == 0 ? pixel_val : sec_grid) + sidx * 2);
        TapRec rec;
        if (half == 0) {
            rec = node_taps(gc.x, gc.y, ng, id.j == 0);
        } else {
            const Taps tp = make_taps(gc.x, gc.y, W, H, id.j == 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rec.off[k] = tp.off[k] * 128;                           // 64 fp16 channels per texel
                rec.w[k] = tp.w[k];
            }
        }
        if (!id.live) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { rec.off[k] = 0; rec.w[k] = 0.0f; }
        }
        (half == 0 ? taps : taps3)[row] = rec;
    }
    __syncthreads();

    // image of the own (j = 0, border table) and of the other (j = 1, zeros table) view of this tile
    const int img_own = b * V + v, img_oth = b * V + (V - 1 - v);
    // ---- phase B: the K = 96 operand image: level-3 gather (64 ch) ‖ tanh(pt/5) (3) ‖ zeros --------------------
    // B-operand order: fragment (k, wave, mt) is 1 KiB, lane (r, g) reads its 16 bytes at lane*16
    {
        const int row = tid >> 1, half = tid & 1;
        const int w_ = row >> 5, rl = row & 31, rs = rl >> 1, mt = rl & 1;
        const TapRec rec = taps3[row];
        const char* m3 = reinterpret_cast<const char*>(map3) + (size_t)(mt == 0 ? img_own : img_oth) * H * W * 128 + half * 64;
        half8 tv[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) tv[c][k] = *reinterpret_cast<const half8*>(m3 + (size_t)(unsigned)rec.off[k] + c * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 tq = __builtin_bit_cast(u32x4, tv[c][k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] = fma_mix_lo(acc[2 * i], tq[i], rec.w[k]);
                    acc[2 * i + 1] = fma_mix_hi(acc[2 * i + 1], tq[i], rec.w[k]);
                }
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
            // chunk = half*4 + c: K step `half`, lane group g = c
            *reinterpret_cast<half8*>(aimg + (((half * 4 + w_) * 2 + mt) * 64 + c * 16 + rs) * 16) = o;
        }
        // third K step: halves 64..66 = point encoding of this row, the rest zero: only lane group g = 0 has data
        if (half == 0) {
            const RowId id = tile_row(row, rgroup, blk, S, R, b, ray0, nrays);
            half8 p8;
#pragma unroll
            for (int e = 0; e < 8; ++e) p8[e] = (_Float16)0.0f;
            if (id.live) {
                const size_t sidx = (((size_t)(b * V + v)) * R + id.r) * S + id.s;
                const float* pe = pe6 + sidx * 6 + id.j * 3;
                p8[0] = (_Float16)pe[0]; p8[1] = (_Float16)pe[1]; p8[2] = (_Float16)pe[2];
            }
            *reinterpret_cast<half8*>(aimg + AIMG2_OFF + ((w_ * 2 + mt) * 16 + rs) * 16) = p8;
        }
    }
    __syncthreads();          // last workgroup-wide barrier: from here on the four waves run independently

    // Tap / store phase lane roles ("load layout"): lane = 4*rl + pl -> row rl, 16-byte piece pl, so that 4 ADJACENT lanes
    // read / write 64 contiguous bytes of one node / row.  The texture addresser walks a wave 4 lanes per cycle and
    // the vector L1 does one tag lookup per distinct line of such a quad: in the MFMA layout (lane = r + 16 g) a quad
    // is 4 different rows = up to 4 lookups for 64 bytes (rocprofv3: 40 L1 accesses per load instruction, TA 73 %
    // busy); here it is one.  Register contents are the same in both layouts (lane (r, g) and lane (rl = r, pl = g)
    // own the same 52 channels), so switching is one ds_bpermute per accumulator register after the MFMA phase.
    const int rl = lane >> 2, pl = lane & 3;
    const int perm_addr = (rl + 16 * pl) * 4;                // this lane takes the accumulators of MFMA lane (r = rl, g = pl)
    // rows of this lane in that phase: tile rows 32*wave + 2*rl + mt, mt = 0 (own image), 1 (other image)
    const RowId lid = tile_row(wave * 32 + 2 * rl, rgroup, blk, S, R, b, ray0, nrays);
    const size_t lrow0 = ((((size_t)b * R + lid.r - ray0) * V + v) * S + lid.s) * 2;          // + mt (only used when live)
    // (image, mode) tables: border table of the own image for mt = 0, zeros table of the other image for mt = 1
    const size_t img_bytes = (size_t)ng.nodes_per_image() * TAB_ROW_BYTES;
    const char* tbase = reinterpret_cast<const char*>(tab);
    const __amdgpu_buffer_rsrc_t trs_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(tbase + img_bytes * img_own), 0, (int)(ng.border_nodes() * TAB_ROW_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t trs_z = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(tbase + img_bytes * img_oth + (size_t)ng.border_nodes() * TAB_ROW_BYTES), 0,
        (int)(ng.zeros_nodes() * TAB_ROW_BYTES), 0x00020000);

    // tap records of the lane's two rows (they do not change from slice to slice)
    TapRec rec[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) rec[mt] = taps[wave * 32 + 2 * rl + mt];
    int vo[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[mt][k] = ((CPN_ENCODE_ABLATE & 16) ? 0 : rec[mt].off[k]) + pl * 16;

    // Slice loop, software-pipelined by one slice on both sides:
    //     issue ALL loads of the iteration (weights of slice n+1, taps of slice n)  ->  the stores of slice n-1  ->  compute slice n.
    // gfx9 has ONE counter for loads and stores (vmcnt) and they retire out of order against each other, so waiting
    // for any load that was issued AFTER a store also waits for that store to reach memory.  With the order above every
    // load a wave ever waits for is OLDER than the stores in flight: the 7 GB hid stream never stalls the wave that
    // issued it, only the end of the kernel does.  No workgroup barrier, no LDS traffic except bias + bpermute.
    half8 res[2][2];                                        // fp16 results of the previous slice, waiting to be stored
    const __half* const hrow = hid + lrow0 * 832 + pl * 8;
    auto store_slice = [&](int n) {
        if (lid.live && (!(CPN_ENCODE_ABLATE & 2) || res[0][0][0] == (_Float16)123.0f)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                __half* o = const_cast<__half*>(hrow) + mt * 832 + n * SLICE_CH;
                *reinterpret_cast<half8*>(o) = res[mt][0];
                *reinterpret_cast<half8*>(o + 32) = res[mt][1];
            }
        }
    };
    // weight fragments: [slice][k < 2][tile][lane] half8 for the 64 full-resolution channels, then
    // [slice][tile][lane] half4 for the K tail (3 point-encoding columns + zero), consumed by a 16x16x16 MFMA
    const half4* const wtail = reinterpret_cast<const half4*>(wfrag + NSLICE * 2 * NT * 64);
    auto load_w = [&](int n, half8 (&w)[2][NT]) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[k][nt] = wfrag[((n * 2 + k) * NT + nt) * 64 + lane];
    };
    // one slice: `wc` holds this slice's main fragments (fetched during the previous slice), `wn` receives the next one's
    auto slice = [&](int n, half8 (&wc)[2][NT], half8 (&wn)[2][NT]) {
        // ---- every load of this iteration first: next slice's weights, this slice's K tail and its 16 tap pieces
        if (n + 1 < NSLICE) load_w(n + 1, wn);
        half4 wt[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wt[nt] = wtail[(n * NT + nt) * 64 + lane];
        u32x4 td[2][4][2];
        if (!(CPN_ENCODE_ABLATE & 1)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const __amdgpu_buffer_rsrc_t rs = mt == 0 ? trs_b : trs_z;
                    td[mt][k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[mt][k], n * TAB_SLICE_BYTES, 0);
                    td[mt][k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[mt][k] + 64, n * TAB_SLICE_BYTES, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (n > 0) store_slice(n - 1);
        __builtin_amdgcn_sched_barrier(0);

        // ---- K = 80 contraction of the full-resolution level + point encoding, on top of the bias.  The B operands
        //      (this wave's rows) come back from LDS every slice: 24 registers that the prefetched weights need
        f32x4 acc[2][NT];
        {
            const float* bp = bias_s + n * SLICE_CH + g * 8;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt] = *reinterpret_cast<const f32x4*>(bp + (nt >> 1) * 32 + (nt & 1) * 4);
                acc[1][nt] = acc[0][nt];
            }
        }
        if (!(CPN_ENCODE_ABLATE & 4)) {
            half8 xa[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    xa[k][mt] = *reinterpret_cast<const half8*>(aimg + (((k * 4 + wave) * 2 + mt) * 64 + lane) * 16);
            half4 xt[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const half4 p4 = *reinterpret_cast<const half4*>(aimg + AIMG2_OFF + ((wave * 2 + mt) * 16 + r) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) xt[mt][e] = (g == 0) ? p4[e] : (_Float16)0.0f;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[k][nt], xa[k][0], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[k][nt], xa[k][1], acc[1][nt], 0, 0, 0);
                }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wt[nt], xt[0], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wt[nt], xt[1], acc[1][nt], 0, 0, 0);
            }
        }
        // MFMA layout -> load layout
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = acc[mt][nt][i];          // (a bit_cast applied to the vector-element lvalue itself
                    acc[mt][nt][i] = __int_as_float(         //  reads element 0 for every i with this compiler)
                        __builtin_amdgcn_ds_bpermute(perm_addr, __float_as_int(t)));
                }
        // ---- 4 table taps per row in fp32 on top of it, then ReLU and the fp16 rounding
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (!(CPN_ENCODE_ABLATE & 1)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wk = rec[mt].w[k];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const u32x4 d = td[mt][k][h];
                        f32x4* a2 = &acc[mt][2 * h];
                        a2[0][0] = fma_mix_lo(a2[0][0], d[0], wk); a2[0][1] = fma_mix_hi(a2[0][1], d[0], wk);
                        a2[0][2] = fma_mix_lo(a2[0][2], d[1], wk); a2[0][3] = fma_mix_hi(a2[0][3], d[1], wk);
                        a2[1][0] = fma_mix_lo(a2[1][0], d[2], wk); a2[1][1] = fma_mix_hi(a2[1][1], d[2], wk);
                        a2[1][2] = fma_mix_lo(a2[1][2], d[3], wk); a2[1][3] = fma_mix_hi(a2[1][3], d[3], wk);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    res[mt][h][i] = (_Float16)fmaxf(acc[mt][2 * h][i], 0.0f);
                    res[mt][h][4 + i] = (_Float16)fmaxf(acc[mt][2 * h + 1][i], 0.0f);
                }
        }
    };
    half8 wA[2][NT], wB[2][NT];
    load_w(0, wA);
    for (int n = 0; n + 1 < NSLICE; n += 2) {               // ping-pong: no register copies between slices
        slice(n, wA, wB);
        slice(n + 1, wB, wA);
    }
    slice(NSLICE - 1, wA, wB);
    store_slice(NSLICE - 1);
}

// ---- node features: the three coarse levels sampled (grid_sample semantics of the mode) at every table node -------
// out (nimg * nodes_per_image, 768) fp16: [level 0 | level 1 | level 2], thread = (node, level, 8-channel chunk)
__global__ void node_features_kernel(const __half* __restrict__ map0, const __half* __restrict__ map1,
                                     const __half* __restrict__ map2, int H, int W, long long total,
                                     __half* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int chunk = (int)(idx % 96);                                  // 3 levels x 32 chunks of 8 channels
    const long long node = idx / 96;
    const int lvl = chunk >> 5, c8 = chunk & 31;
    const NodeGrid ng{W >> 1, H >> 1};
    const long long npi = ng.nodes_per_image();
    const int img = (int)(node / npi);
    long long rem = node - (long long)img * npi;
    const bool border = rem < ng.border_nodes();
    if (!border) rem -= ng.border_nodes();
    const int nw = border ? ng.bw() : ng.zw(), pad = border ? 0 : PAD;
    const int ny = (int)(rem / nw) - pad, nx = (int)(rem % nw) - pad;
    // node t <-> u = t / M <-> normalised g = 2u - 1 (exact when M is a power of two)
    const float gx = (float)(2 * nx - ng.Mx) / (float)ng.Mx, gy = (float)(2 * ny - ng.My) / (float)ng.My;
    const int shift = 4 - lvl;
    const int Hl = H >> shift, Wl = W >> shift;
    const Taps tp = make_taps(gx, gy, Wl, Hl, border);
    const __half* m = (lvl == 0 ? map0 : lvl == 1 ? map1 : map2) + (size_t)img * Hl * Wl * 256 + c8 * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x4 tq = __builtin_bit_cast(u32x4, *reinterpret_cast<const half8*>(m + (size_t)tp.off[k] * 256));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[2 * i] = fma_mix_lo(acc[2 * i], tq[i], tp.w[k]);
            acc[2 * i + 1] = fma_mix_hi(acc[2 * i + 1], tq[i], tp.w[k]);
        }
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)acc[e];
    *reinterpret_cast<half8*>(out + (size_t)node * 768 + lvl * 256 + c8 * 8) = o;
}

// ---- weight images ------------------------------------------------------------------------------------------------
// channel of a slice that MFMA tile nt, A-operand row a computes (lane (r, g) of the result then holds a = g*4 + i)
__host__ __device__ inline int slice_channel(int nt, int a) {
    return (nt >> 1) * 32 + (a >> 2) * 8 + (nt & 1) * 4 + (a & 3);
}

// W (832, 835) fp32 -> wfrag: [slice][k < 2][nt][lane] half8 over columns 768..831 (the full-resolution level), followed
// by [slice][nt][lane] half4 over columns 832..834 + one zero (the K tail, v_mfma_f32_16x16x16_f16: lane group 0 only)
__global__ void pack_encode_frag_kernel(const float* __restrict__ W, int ldw, half8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nmain = NSLICE * 2 * NT * 64;
    if (idx < nmain) {
        const int lane = idx & 63;
        int t = idx >> 6;
        const int nt = t % NT; t /= NT;
        const int k = t % 2;
        const int n = t / 2;
        const int ch = n * SLICE_CH + slice_channel(nt, lane & 15);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)W[(size_t)ch * ldw + 768 + k * 32 + (lane >> 4) * 8 + e];
        out[idx] = o;
        return;
    }
    const int j = idx - nmain;
    if (j >= NSLICE * NT * 64) return;
    const int lane = j & 63;
    const int nt = (j >> 6) % NT, n = (j >> 6) / NT;
    const int ch = n * SLICE_CH + slice_channel(nt, lane & 15);
    half4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kk = (lane >> 4) * 4 + e;                         // K index inside the 16-wide tail
        o[e] = (_Float16)(kk < 3 ? W[(size_t)ch * ldw + 832 + kk] : 0.0f);
    }
    reinterpret_cast<half4*>(out + nmain)[j] = o;
}

// W (832, 835) fp32 -> the table projection (832, 768) fp16 over the three coarse levels (natural channel order)
__global__ void pack_table_weight_kernel(const float* __restrict__ W, int ldw, __half* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CPN_TAB_LD * 768) return;
    const int kc = idx % 768, ch = idx / 768;
    out[idx] = __float2half(W[(size_t)ch * ldw + kc]);
}

}  // namespace

extern "C" long long cpn_encode_table_nodes(int H, int W) {
    if (H < 16 || W < 16 || (H % 16) || (W % 16)) return -1;
    const NodeGrid ng{W >> 1, H >> 1};
    return ng.nodes_per_image();
}

extern "C" int cpn_pack_encode_weights(const float* W, int ldw, uint16_t* wfrag, uint16_t* wtab, void* stream) {
    CPN_REQUIRE(W && wfrag && wtab, CPN_E_ARG, "cpn_pack_encode_weights: null pointer");
    CPN_REQUIRE(ldw >= 835, CPN_E_SHAPE, "cpn_pack_encode_weights: ldw=%d < 835", ldw);
    const hipStream_t s = (hipStream_t)stream;
    const int nf = NSLICE * 2 * NT * 64 + NSLICE * NT * 64;           // main half8 fragments + half4 tail fragments
    hipLaunchKernelGGL(pack_encode_frag_kernel, dim3(cpn_cdiv(nf, 256)), dim3(256), 0, s, W, ldw, (half8*)wfrag);
    hipLaunchKernelGGL(pack_table_weight_kernel, dim3(cpn_cdiv(CPN_TAB_LD * 768, 256)), dim3(256), 0, s, W, ldw,
                       (__half*)wtab);
    CPN_LAUNCH_CHECK("cpn_pack_encode_weights");
    return 0;
}

extern "C" int cpn_node_features(const uint16_t* map0, const uint16_t* map1, const uint16_t* map2, int H, int W,
                                 int nimg, uint16_t* out, void* stream) {
    CPN_REQUIRE(map0 && map1 && map2 && out, CPN_E_ARG, "cpn_node_features: null pointer");
    CPN_REQUIRE(nimg > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0, CPN_E_SHAPE,
                "cpn_node_features: need H,W multiples of 16 (got H=%d W=%d)", H, W);
    const NodeGrid ng{W >> 1, H >> 1};
    const long long total = ng.nodes_per_image() * nimg * 96;
    CPN_REQUIRE(total / 256 < (1LL << 31), CPN_E_SHAPE, "cpn_node_features: too many nodes");
    hipLaunchKernelGGL(node_features_kernel, dim3((unsigned)cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)map0, (const __half*)map1, (const __half*)map2, H, W, total, (__half*)out);
    CPN_LAUNCH_CHECK("cpn_node_features");
    return 0;
}

extern "C" int cpn_encode_hidden(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                                 const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                                 int B, int V, int R, int S, int ray0, int nrays, uint16_t* hid, void* stream) {
    CPN_REQUIRE(tab && map3 && pixel_val && sec_grid && pe6 && wfrag && bias && hid, CPN_E_ARG,
                "cpn_encode_hidden: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0,
                CPN_E_SHAPE, "cpn_encode_hidden: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_encode_hidden: ray range [%d,%d) outside B*R=%lld", ray0, ray0 + nrays, (long long)B * R);
    const long long nrows = (long long)nrays * V * S * 2;
    const NodeGrid ng{W >> 1, H >> 1};
    CPN_REQUIRE(nrows < (1LL << 31) && ng.zeros_nodes() * TAB_ROW_BYTES < (1LL << 31) && (long long)H * W * 128 < (1LL << 31),
                CPN_E_SHAPE, "cpn_encode_hidden: chunk / per-image table too large for 32-bit offsets (%lld rows)", nrows);
    CPN_REQUIRE(((uintptr_t)tab % 16) == 0 && ((uintptr_t)map3 % 16) == 0 && ((uintptr_t)wfrag % 16) == 0 &&
                    ((uintptr_t)bias % 16) == 0 && ((uintptr_t)hid % 16) == 0, CPN_E_ARG,
                "cpn_encode_hidden: pointers must be 16-byte aligned");
    static_assert(2 * (2 * TILE_ROWS * 32 + AIMG_BYTES + 832 * 4) <= 160 * 1024, "at least two workgroups per CU");
    // ray groups: TG consecutive rays of ONE batch element (r aligned to TG), so a tile's images are uniform
    const int groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    const long long group0 = (long long)b_lo * groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    const int nblk = (int)cpn_cdiv(S, TSB);
    const long long tiles = (group1 - group0 + 1) * V * nblk;
    CPN_REQUIRE(tiles < (1LL << 31), CPN_E_SHAPE, "cpn_encode_hidden: %lld tiles exceed the grid limit", tiles);
    hipLaunchKernelGGL(encode_hidden_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)tab, (const __half*)map3, H, W, pixel_val, sec_grid, pe6, (const half8*)wfrag,
                       bias, V, R, S, ray0, nrays, nblk, groups_per_b, group0, (__half*)hid);
    CPN_LAUNCH_CHECK("cpn_encode_hidden");
    return 0;
}
