// Pieces shared by the two forms of the first encoder layer on the node tables: cpn_encode_hidden (encode.hip) and
// cpn_encode_key (encode_key.hip, the same layer with the folded key_map contraction fused behind it) - geometry of the
// node tables, tap records, the (ray, sample) <-> tile-row map, the channel order of an MFMA tile, the hid store.
// Design notes: encode.hip.
#pragma once
#include "common.h"
#include "taps.h"

namespace {

constexpr int NSLICE = 13;                // 832 = 13 x 64 output channels
constexpr int SLICE_CH = 64;
constexpr int NT = 4;                     // 16-channel MFMA tiles per slice
// K = 80 = 2 x 32 full-resolution channels (v_mfma_f32_16x16x32_f16) + a 16-wide tail (3 point encodings + zeros, 16x16x16)
constexpr int PAD = CPN_NODE_PAD;         // zero rim of the 'zeros' table, in nodes (= level-0 texel pitch / 2)
constexpr int TAB_SLICE_BYTES = SLICE_CH * 2;                  // 128: one cache line per node and slice
constexpr int TAB_ROW_BYTES = CPN_TAB_LD * 2;                  // 1664 per node, channels in natural order
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

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

// 16-byte hid store; CPN_ENCODE_STORE selects the form: 0 = write-back, 1 = nt store inside an asm (rounds 2-3), 2..5 other
// cache policy bits (experiments), 6 = compiler-emitted nt store behind a branch, 7 = unconditional nt BUFFER store that the
// compiler counts in its vmcnt bookkeeping (encode.hip; the product)
#ifndef CPN_ENCODE_STORE
#define CPN_ENCODE_STORE 7
#endif
__device__ __forceinline__ void store16(__half* p, half8 v) {
#if CPN_ENCODE_STORE == 0
    *reinterpret_cast<half8*>(p) = v;
#elif CPN_ENCODE_STORE == 1
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
#elif CPN_ENCODE_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#elif CPN_ENCODE_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif CPN_ENCODE_STORE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
#elif CPN_ENCODE_STORE == 5
    asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
#elif CPN_ENCODE_STORE == 6
    // the same nt store, but emitted by the compiler: it then COUNTS the store in its s_waitcnt vmcnt bookkeeping (an
    // asm store is invisible to it, so every "vmcnt(0)" it places for the youngest tap load also waits for the stores
    // issued behind that load)
    __builtin_nontemporal_store(v, reinterpret_cast<half8*>(p));
#endif
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

// Which (ray, sample) a row of a wave tile is.  A wave tile = 4 adjacent rays (same batch element) x 4 consecutive
// samples x {own, other image} of one view = 32 rows; MFMA column / load-layout row index c = (sample & 3)*4 + (ray & 3),
// so ONE load instruction covers a 4 x 4 patch of (sample, ray) whose taps fall on a handful of nodes.
constexpr int TG = 4;                     // rays per wave tile
constexpr int TSW = 4;                    // samples per wave tile

struct RowId {
    bool live;
    int s, r;                             // sample, ray index inside its batch element
};
__device__ __forceinline__ RowId tile_row(int c, int rgroup, int sblk, int S, int R, int b, int ray0, int nrays) {
    RowId o;
    o.s = sblk * TSW + (c >> 2);
    o.r = rgroup * TG + (c & 3);
    const long long ray = (long long)b * R + o.r;
    o.live = (o.s < S) && (o.r < R) && ray >= ray0 && ray < (long long)ray0 + nrays;
    return o;
}

// channel of a slice that MFMA tile nt, A-operand row a computes (lane (r, g) of the result then holds a = g*4 + i)
__host__ __device__ inline int slice_channel(int nt, int a) {
    return (nt >> 1) * 32 + (a >> 2) * 8 + (nt & 1) * 4 + (a & 3);
}

}  // namespace
