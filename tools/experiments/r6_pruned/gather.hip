// K2 — bilinear multi-scale feature gather into fp16 encoder-input rows, plus the layout /
// packing helpers and the tiny fp32 first layers of the attention MLPs.
//
// Replaces F.grid_sample(..., 'bilinear', 'border' | 'zeros', align_corners=False) x 8 and the
// torch.cat's around them (/root/reference models/CoPoNeRF.py:312, 370, 384-394).
//
// Layout: feature maps are NHWC fp16 so that one bilinear tap of one level is ONE contiguous
// 512-B (256 ch) or 128-B (64 ch) segment; a thread owns 8 channels (16 B) of one output row, so a
// wave reads 4 x 1 KiB fully-coalesced segments and writes 1 KiB of the row.  HBM/L2-bound:
// per output row 4 taps x 832 ch x 2 B = 6.5 KiB read (mostly L2 hits: 44.6 MB fp32 -> 22.3 MB fp16
// per pair at 256^2, neighbouring samples share texels) and 1.75 KiB written.
#include <algorithm>

#include "common.h"
#include "taps.h"

namespace {

// ---------------------------------------------------------------------------------------------
// (N,C,h,w) fp32 -> (N,h,w,C) fp16, 32x32 LDS tile transpose over (C, h*w)
// ---------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_f16_kernel(const float* __restrict__ src, __half* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;                 // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (p < HW && c < C) dst[((size_t)n * HW + p) * C + c] = __float2half(tile[tx][j]);
    }
}

__global__ void pack_weight_f16_kernel(const float* __restrict__ src, int n_out, int k_in,
                                       __half* __restrict__ dst, int ld) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n_out * ld) return;
    const int k = (int)(idx % ld), n = (int)(idx / ld);
    dst[idx] = __float2half(k < k_in ? src[(size_t)n * k_in + k] : 0.0f);
}

// 108 16-byte chunks per row: 32 | 32 | 32 (levels 0-2, 256 ch) | 8 (level 3, 64 ch) | 1 (pe) | 3 (zero)
constexpr int LANES_PER_LINE = 32;      // 4 levels x 8 lanes

// thread = (ray, view, j, level, sub) and walks the S samples of that epipolar line.  Per line the index arithmetic is
// done once; per sample a level-0..2 thread produces the 4 chunks sub, sub+8, sub+16, sub+24 of its level (8
// neighbouring lanes read / write 128 contiguous bytes) and keeps the 4 x 4 texel vectors in registers: consecutive
// samples of a line land on the same 2x2 texel quad ~60 % of the time at level 0 and ~20 % at level 1, and a tap
// whose texel did not move is not fetched again (the kernel is bound by L2->CU traffic: 6.6 KB of taps per 1.7 KB
// row).  Level-3 threads own one chunk (64 channels) and, for sub < 4, the pe / zero-pad chunk 104+sub.
__global__ __launch_bounds__(256) void gather_rows_kernel(
    const __half* __restrict__ map0, const __half* __restrict__ map1, const __half* __restrict__ map2,
    const __half* __restrict__ map3, int H, int W, const float* __restrict__ pixel_val,
    const float* __restrict__ sec_grid, const float* __restrict__ pe6, int V, int R, int S, int ray0,
    int nlines, __half* __restrict__ xin) {
    // XCD-aware order: blocks are dispatched round-robin over the 8 XCDs; giving each XCD a contiguous range of
    // lines (= neighbouring rays = overlapping texel footprints) keeps its private L2 on 1/8 of the feature maps
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, rem = nb & 7;
    const unsigned lblock = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    const unsigned gid = lblock * blockDim.x + threadIdx.x;
    const unsigned line = gid / LANES_PER_LINE;                       // = (ray_local * V + v) * 2 + j
    const int lvl = (int)(gid >> 3) & 3, sub = (int)gid & 7;
    if (line >= (unsigned)nlines) return;
    const int j = (int)(line & 1);
    unsigned t = line >> 1;
    const int v = (int)(t % (unsigned)V); t /= (unsigned)V;           // t = ray within this launch
    const unsigned ray = (unsigned)ray0 + t;
    const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
    const size_t sidx0 = (((size_t)(b * V + v)) * R + r) * S;         // first sample of the line in (N,R,S) arrays

    const int shift = 4 - lvl - (lvl == 3);                           // H/16, H/8, H/4, H
    const int Hl = H >> shift, Wl = W >> shift;
    const int C = (lvl == 3) ? 64 : 256;
    const __half* base = (lvl == 0) ? map0 : (lvl == 1) ? map1 : (lvl == 2) ? map2 : map3;
    // j = 0: own image at the epipolar sample (border); j = 1: other image at the reprojected point (zeros)
    const float2* g = reinterpret_cast<const float2*>((j == 0 ? pixel_val : sec_grid) + sidx0 * 2);
    const int img = b * V + (j == 0 ? v : (V - 1 - v));
    const __half* m = base + (size_t)img * Hl * Wl * C + sub * 8;
    // row = ((ray_local*V + v)*S + s)*2 + j
    __half* orow = xin + ((size_t)(t * V + v) * S * 2 + j) * CPN_XIN_STRIDE;
    const int ocol = (lvl == 3 ? 768 : lvl * 256) + sub * 8;
    const bool coarse = lvl < 3;

    half8 tv[4][4];                     // [chunk][tap] texel vectors of the previous sample
    int prev[4] = {-1, -1, -1, -1};
    float2 gnext = g[0];
    for (int s = 0; s < S; ++s) {
        const float2 gq = gnext;
        if (s + 1 < S) gnext = g[s + 1];              // the next coordinate is in flight while this sample's texels load
        const Taps tp = make_taps(gq.x, gq.y, Wl, Hl, j == 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (tp.off[k] != prev[k]) {
                const __half* src = m + (size_t)tp.off[k] * C;
                tv[0][k] = *reinterpret_cast<const half8*>(src);
                if (coarse) {
                    tv[1][k] = *reinterpret_cast<const half8*>(src + 64);
                    tv[2][k] = *reinterpret_cast<const half8*>(src + 128);
                    tv[3][k] = *reinterpret_cast<const half8*>(src + 192);
                }
                prev[k] = tp.off[k];
            }
        }
        __half* o = orow + (size_t)s * 2 * CPN_XIN_STRIDE;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (it > 0 && !coarse) break;
            // acc[e] = fma(f32(texel), w, acc[e]) with the fp16 -> fp32 conversion inside the FMA (v_fma_mix_f32):
            // 128 instructions per sample instead of 128 conversions + 64 packed FMAs, same bits
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 tq = __builtin_bit_cast(u32x4, tv[it][k]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * i]) : "v"(tq[i]), "v"(tp.w[k]));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                        : "+v"(acc[2 * i + 1]) : "v"(tq[i]), "v"(tp.w[k]));
                }
            }
            half8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (_Float16)acc[e];
            *reinterpret_cast<half8*>(o + ocol + it * 64) = out;
        }
        if (!coarse && sub < 4) {
            half8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = (_Float16)0.0f;
            if (sub == 0) {
                const float* pe = pe6 + (sidx0 + s) * 6 + j * 3;
                out[0] = (_Float16)pe[0]; out[1] = (_Float16)pe[1]; out[2] = (_Float16)pe[2];
            }
            *reinterpret_cast<half8*>(o + (104 + sub) * 8) = out;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// first (16 -> 128) layer of query_embed / query_repeat_embed in fp32, output fp16 rows.
// One wave = 16 rows per step on the fp32 matrix cores: D(128 ch x 16 rows) = W(128 x 16) . L^T(16 x 16 rows) as 8 output
// tiles x 4 v_mfma_f32_16x16x4_f32 (the VALU form, 128 FMA per 16-byte store, ran at 20 % of the fp32 VALU peak:
// 0.54 ms per launch).  Weights are the MFMA A operand and stay in registers for the whole kernel; the rows of tile
// t = 2p+h are assigned to channels p*32 + (a/4)*8 + h*4 + a%4, so a lane ends up with 8 CONSECUTIVE channels of one
// row per tile pair (16-byte stores, 64 contiguous bytes per row and pair).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void local_hidden_kernel(
    const float* __restrict__ loc8, const float* __restrict__ coords9, const float* __restrict__ w, int ldw,
    const float* __restrict__ bias, const float* __restrict__ add, int V, int R, int S, int ray0,
    long long nrows, __half* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int a = lane & 15, fg = lane >> 4;
    f32x4 wv[8];                  // wv[t][e] = W[channel(t, a)][fg*4 + e]
    f32x4 bv[8];                  // bias of the 4 channels this lane owns in tile t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int p = t >> 1, h = t & 1;
        const int ch_a = p * 32 + (a >> 2) * 8 + h * 4 + (a & 3);
        wv[t] = *reinterpret_cast<const f32x4*>(w + (size_t)ch_a * ldw + fg * 4);
        bv[t] = *reinterpret_cast<const f32x4*>(bias + p * 32 + fg * 8 + h * 4);
    }
    const unsigned ngroups = (unsigned)((nrows + 15) >> 4);
    const unsigned wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (unsigned grp = wave_id; grp < ngroups; grp += nwaves) {
        const unsigned row = grp * 16 + a;
        const bool live = row < (unsigned)nrows;
        unsigned t_ = live ? row : (unsigned)nrows - 1;            // 32-bit: see gather_rows_kernel
        const int s = (int)(t_ % (unsigned)S); t_ /= (unsigned)S;
        const int v = (int)(t_ % (unsigned)V); t_ /= (unsigned)V;
        const unsigned ray = (unsigned)ray0 + t_;
        const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
        const size_t nr = ((size_t)(b * V + v)) * R + r;
        const f32x4 l0 = *reinterpret_cast<const f32x4*>(loc8 + (nr * S + s) * 8);
        const f32x4 l1 = *reinterpret_cast<const f32x4*>(loc8 + (nr * S + s) * 8 + 4);
        const float* c9 = coords9 + nr * 9;
        // local_coords channel order (CoPoNeRF.py:445): ctx dir 0-2, zeros 3-5, query dir 6-8, depth 9-12, origin 13-15;
        // this lane feeds k = fg*4 .. fg*4+3
        f32x4 lv;
        if (fg == 0) lv = f32x4{l0[0], l0[1], l0[2], 0.f};
        else if (fg == 1) lv = f32x4{0.f, 0.f, c9[0], c9[1]};
        else if (fg == 2) lv = f32x4{c9[2], l0[3], l1[0], l1[1]};
        else lv = f32x4{l1[2], c9[6], c9[7], c9[8]};
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = bv[t];
            if (add) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(add + (size_t)t_ * 128 + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
                acc[t] += av;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][e], lv[e], acc[t], 0, 0, 0);
        if (live) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                half8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (_Float16)fmaxf(acc[2 * p][i], 0.0f);
                    o[4 + i] = (_Float16)fmaxf(acc[2 * p + 1][i], 0.0f);
                }
                *reinterpret_cast<half8*>(out + (size_t)row * 128 + p * 32 + fg * 8) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// query_embed / query_repeat_embed as ONE kernel: out = W2 . relu(W1 . L + b1 + add) + b2  (16 -> 128 -> 128), the
// hidden layer never leaves the registers.  Stage 1 is local_hidden_kernel's fp32 MFMA; its accumulator layout
// (lane = row ln, channels p*32 + fg*8 + 0..7) IS the B-operand layout of v_mfma_f32_16x16x32_f16 for K block p, so
// after ReLU and the fp16 rounding (the same rounding the stored hidden layer had) the second layer runs straight
// off the registers against W2 fragments held in LDS (lane-linear 16-byte slots, 32 KiB).  Output rows use the
// same channel permutation: 16-byte stores, 64 contiguous bytes per row and tile pair.
// ---------------------------------------------------------------------------------------------
// timing-only ablations for tools/local_mlp_bench.py (results are wrong when non-zero; the product builds with 0):
// 1 = no first-layer MFMAs, 2 = no second-layer MFMAs, 4 = no `add` loads, 8 = no `dot_with` loads, 16 = no stores
#ifndef CPN_LMLP_ABLATE
#define CPN_LMLP_ABLATE 0
#endif
// 1 = the dot_with rows in the load layout (lane = 4 * row + piece, 16 L1 tag look-ups per instruction instead of 64) and 16
// ds_bpermute per group to the accumulator layout: measured SLOWER here (0.293 against 0.235 ms per 16 384-ray call, tools/
// local_mlp_bench.py variant 100 = this macro at 0) - the kernel's LDS pipe already carries the W2 fragments; the product builds 0
#ifndef CPN_LMLP_DOT_LOAD_LAYOUT
#define CPN_LMLP_DOT_LOAD_LAYOUT 0
#endif
__global__ __launch_bounds__(512, 4) void local_mlp_kernel(
    const float* __restrict__ loc8, const float* __restrict__ coords9, const float* __restrict__ w1, int ldw1,
    const float* __restrict__ b1, const float* __restrict__ add, const __half* __restrict__ w2, int ldw2,
    const float* __restrict__ b2, int V, int R, int S, int ray0, long long nrows, __half* __restrict__ out,
    const __half* __restrict__ dot_with, float* __restrict__ logits_out, int frag) {
    // 8 waves share the W2 fragments; 2 workgroups per CU = 4 waves per SIMD (<= 128 VGPRs): the first layer's bias
    // rides on the unused K = 3 input slot (x = 1, exact in the fp32 MFMA), the second layer's sits in LDS, and the
    // inputs of the NEXT 16-row group are requested before the MFMAs of the current one.
    __shared__ __attribute__((aligned(16))) half8 w2l[8 * 4 * 64];        // [tile t][k block p][lane]
    __shared__ __attribute__((aligned(16))) half8 ostage[8][16 * 17];
    __shared__ __attribute__((aligned(16))) float b2s[128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = lane & 15, fg = lane >> 4;
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += 512) {
        const int l = i & 63, p = (i >> 6) & 3, t = i >> 8;
        const int ch = (t >> 1) * 32 + ((l & 15) >> 2) * 8 + (t & 1) * 4 + (l & 3);       // output channel of tile row
        w2l[i] = *reinterpret_cast<const half8*>(w2 + (size_t)ch * ldw2 + p * 32 + (l >> 4) * 8);
    }
    if (threadIdx.x < 128) b2s[threadIdx.x] = b2[threadIdx.x];
    f32x4 wv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int ch = (t >> 1) * 32 + (a >> 2) * 8 + (t & 1) * 4 + (a & 3);
        wv[t] = *reinterpret_cast<const f32x4*>(w1 + (size_t)ch * ldw1 + fg * 4);
        if (fg == 0) wv[t][3] = b1[ch];                        // K slot 3 is unused by the inputs: bias x 1.0
    }
    __syncthreads();
    const unsigned ngroups = (unsigned)((nrows + 15) >> 4);
    const unsigned wave_id = blockIdx.x * 8 + wave, nwaves = gridDim.x * 8;

    struct RowIn {
        f32x4 lv;          // this lane's 4 K entries of the 16-wide input
        unsigned rayrel;   // ray - ray0 (row of `add`)
    };
    auto fetch = [&](unsigned grp) {
        const unsigned row = grp * 16 + a;
        unsigned t_ = row < (unsigned)nrows ? row : (unsigned)nrows - 1;
        const int s = (int)(t_ % (unsigned)S); t_ /= (unsigned)S;
        const int v = (int)(t_ % (unsigned)V); t_ /= (unsigned)V;
        const unsigned ray = (unsigned)ray0 + t_;
        const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);
        const size_t nr = ((size_t)(b * V + v)) * R + r;
        const float* lp = loc8 + (nr * S + s) * 8;
        const float* c9 = coords9 + nr * 9;
        RowIn o;
        o.rayrel = t_;
        if (fg == 0) { const f32x4 l0 = *reinterpret_cast<const f32x4*>(lp); o.lv = f32x4{l0[0], l0[1], l0[2], 1.0f}; }
        else if (fg == 1) o.lv = f32x4{0.f, 0.f, c9[0], c9[1]};
        else if (fg == 2) o.lv = f32x4{c9[2], lp[3], lp[4], lp[5]};
        else o.lv = f32x4{lp[6], c9[6], c9[7], c9[8]};
        return o;
    };

    RowIn cur = fetch(wave_id < ngroups ? wave_id : 0);
    for (unsigned grp = wave_id; grp < ngroups; grp += nwaves) {
        const unsigned row = grp * 16 + a;
        const bool live = row < (unsigned)nrows;
        const RowIn nxt = fetch(grp + nwaves < ngroups ? grp + nwaves : grp);
        half8 cv[4];
        if (logits_out && !(CPN_LMLP_ABLATE & 8)) {
#if CPN_LMLP_DOT_LOAD_LAYOUT
            // load layout: lane = 4 * row + piece, 4 adjacent lanes read 64 contiguous bytes of one row (16 L1 tag look-ups per
            // instruction; in the fragment layout lane = row + 16 * piece every lane is a look-up of its own: 64)
            const unsigned lrow = grp * 16 + (lane >> 2);
            const unsigned crow = lrow < (unsigned)nrows ? lrow : (unsigned)nrows - 1;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                cv[p] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(dot_with + (size_t)crow * 128 + p * 32 + (lane & 3) * 8));
#else
            if (frag) {
                // fragment order (CPN_ROWS_FRAG): the 1 KiB a wave needs of (16-row group, 32-channel block p) is contiguous
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    cv[p] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(dot_with) + ((size_t)grp * 4 + p) * 64 + lane);
            } else {
                const unsigned crow = live ? row : (unsigned)nrows - 1;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    cv[p] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(dot_with + (size_t)crow * 128 + p * 32 + fg * 8));
            }
#endif
        }
        f32x4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (add && !(CPN_LMLP_ABLATE & 4))
                acc[t] = *reinterpret_cast<const f32x4*>(add + (size_t)cur.rayrel * 128 + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
        }
        if (!(CPN_LMLP_ABLATE & 1)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[t][e], cur.lv[e], acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] += f32x4{cur.lv[0], cur.lv[1], cur.lv[2], cur.lv[3]} * wv[t];
        }
        // hidden layer -> fp16 B operands: K block p = channels p*32 .. p*32+31, this lane holds fg*8 .. fg*8+7 of it
        half8 hb[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hb[p][i] = (_Float16)fmaxf(acc[2 * p][i], 0.0f);
                hb[p][4 + i] = (_Float16)fmaxf(acc[2 * p + 1][i], 0.0f);
            }
        f32x4 o2[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            o2[t] = *reinterpret_cast<const f32x4*>(b2s + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
            if (!(CPN_LMLP_ABLATE & 2)) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                o2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2l[(t * 4 + p) * 64 + lane], hb[p], o2[t], 0, 0, 0);
            } else {
                o2[t] += f32x4{(float)hb[t & 3][0], (float)hb[t & 3][1], (float)hb[t & 3][2], (float)hb[t & 3][3]};
            }
        }
        cur = nxt;
        if (logits_out) {
            // the consumer only needs <out[row], dot_with[row]>: form it here from the fp16-rounded outputs (the values
            // a stored row would have had) and write 4 bytes per row instead of 256
            float dsum = 0.0f;
#if CPN_LMLP_DOT_LOAD_LAYOUT
#pragma unroll
            for (int p = 0; p < 4; ++p) {                      // to the accumulator layout (lane = row + 16 * piece)
                const u32x4 src = __builtin_bit_cast(u32x4, cv[p]);
                u32x4 dst;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute((4 * a + fg) * 4, (int)src[i]);
                cv[p] = __builtin_bit_cast(half8, dst);
            }
#endif
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dsum += (float)(_Float16)o2[2 * p][i] * (float)cv[p][i];
                    dsum += (float)(_Float16)o2[2 * p + 1][i] * (float)cv[p][4 + i];
                }
            dsum += __shfl_xor(dsum, 16);
            dsum += __shfl_xor(dsum, 32);
            if (live && fg == 0 && (!(CPN_LMLP_ABLATE & 16) || dsum == 12345.678f)) logits_out[row] = dsum;
            continue;
        }
        if (frag) {
            // fragment order: the registers leave as they are, 4 stores of 1 KiB of contiguous memory each (dead rows of the
            // last group land in the buffer's padding: the caller sizes it to whole groups)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                half8 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (_Float16)o2[2 * p][i];
                    o[4 + i] = (_Float16)o2[2 * p + 1][i];
                }
                if (!(CPN_LMLP_ABLATE & 16) || (float)o[0] == 12345.0f)
                    reinterpret_cast<half8*>(out)[((size_t)grp * 4 + p) * 64 + lane] = o;
            }
            continue;
        }
        // stage the wave's 16 x 128 tile in LDS and write whole 256-byte rows (4 rows per store instruction)
        half8* stg = ostage[wave];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            half8 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (_Float16)o2[2 * p][i];
                o[4 + i] = (_Float16)o2[2 * p + 1][i];
            }
            stg[a * 17 + p * 4 + fg] = o;                       // row a, 16-byte slot p*4+fg (row stride 17 slots)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = q * 4 + (lane >> 4), slot = lane & 15;
            const unsigned orow = grp * 16 + rr;
            const half8 o = stg[rr * 17 + slot];
            if (orow < (unsigned)nrows && (!(CPN_LMLP_ABLATE & 16) || (float)o[0] == 12345.0f))
                *reinterpret_cast<half8*>(out + (size_t)orow * 128 + slot * 8) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

}  // namespace

extern "C" int cpn_nchw_to_nhwc_f16(const float* src, uint16_t* dst, int N, int C, int h, int w, void* stream) {
    CPN_REQUIRE(src && dst, CPN_E_ARG, "cpn_nchw_to_nhwc_f16: null pointer");
    CPN_REQUIRE(N > 0 && C > 0 && h > 0 && w > 0 && N < 65536, CPN_E_SHAPE, "cpn_nchw_to_nhwc_f16: bad shape");
    const int HW = h * w;
    dim3 grid(cpn_cdiv(HW, 32), cpn_cdiv(C, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_f16_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, src, (__half*)dst, C, HW);
    CPN_LAUNCH_CHECK("cpn_nchw_to_nhwc_f16");
    return 0;
}

extern "C" int cpn_pack_weight_f16(const float* src, int n_out, int k_in, uint16_t* dst, int ld, void* stream) {
    CPN_REQUIRE(src && dst, CPN_E_ARG, "cpn_pack_weight_f16: null pointer");
    CPN_REQUIRE(n_out > 0 && k_in > 0 && ld >= k_in, CPN_E_SHAPE, "cpn_pack_weight_f16: bad shape");
    const long long total = (long long)n_out * ld;
    hipLaunchKernelGGL(pack_weight_f16_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       src, n_out, k_in, (__half*)dst, ld);
    CPN_LAUNCH_CHECK("cpn_pack_weight_f16");
    return 0;
}

extern "C" int cpn_gather_rows(const uint16_t* map0, const uint16_t* map1, const uint16_t* map2,
                               const uint16_t* map3, int H, int W, const float* pixel_val, const float* sec_grid,
                               const float* pe6, int B, int V, int R, int S, int ray0, int nrays, uint16_t* xin,
                               void* stream) {
    CPN_REQUIRE(map0 && map1 && map2 && map3 && pixel_val && sec_grid && pe6 && xin, CPN_E_ARG,
                "cpn_gather_rows: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0,
                CPN_E_SHAPE, "cpn_gather_rows: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_gather_rows: ray range [%d,%d) outside B*R=%lld", ray0, ray0 + nrays, (long long)B * R);
    const long long nrows = (long long)nrays * V * S * 2;
    const long long nlines = (long long)nrays * V * 2;
    const long long total = nlines * LANES_PER_LINE;
    CPN_REQUIRE(nrows < (1LL << 31) && total < (1LL << 31) && (long long)B * R < (1LL << 31), CPN_E_SHAPE,
                "cpn_gather_rows: chunk too large for 32-bit indexing (%lld rows)", nrows);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)map0, (const __half*)map1, (const __half*)map2, (const __half*)map3, H, W,
                       pixel_val, sec_grid, pe6, V, R, S, ray0, (int)nlines, (__half*)xin);
    CPN_LAUNCH_CHECK("cpn_gather_rows");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Reference-arithmetic form of the gather (round 5, RenderEngine(precision="f32")): fp32 maps (NHWC), fp32 rows, the four
// taps summed in ATen's order (nw, ne, sw, se) without contraction into the fp16 pipeline.  A plain kernel - thread = 4
// channels of one row - for an opt-in verification mode, not for speed.  Row layout as cpn_gather_rows: 832 features |
// tanh(pt/5) (3) | zeros up to ld.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(
    const float* __restrict__ map0, const float* __restrict__ map1, const float* __restrict__ map2,
    const float* __restrict__ map3, int H, int W, const float* __restrict__ pixel_val, const float* __restrict__ sec_grid,
    const float* __restrict__ pe6, int V, int R, int S, int ray0, long long nrows2, float* __restrict__ xin, int ld) {
    const int quads = ld >> 2;                                         // 4-channel groups per row
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = gid / quads;
    if (row >= nrows2) return;
    const int c0 = (int)(gid - row * quads) * 4;
    const int j = (int)(row & 1);
    long long t = row >> 1;
    const int s = (int)(t % S); t /= S;
    const int v = (int)(t % V); t /= V;
    const long long ray = (long long)ray0 + t;
    const int b = (int)(ray / R), r = (int)(ray % R);
    const size_t sidx = (((size_t)(b * V + v)) * R + r) * S + s;
    float* o = xin + (size_t)row * ld + c0;
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    if (c0 < 832) {
        const int lvl = c0 < 768 ? c0 >> 8 : 3;
        const int cl = c0 - (lvl == 3 ? 768 : lvl * 256);
        const int shift = 4 - lvl - (lvl == 3);
        const int Hl = H >> shift, Wl = W >> shift, C = lvl == 3 ? 64 : 256;
        const float* base = lvl == 0 ? map0 : lvl == 1 ? map1 : lvl == 2 ? map2 : map3;
        const float2 g = *reinterpret_cast<const float2*>((j == 0 ? pixel_val : sec_grid) + sidx * 2);
        const int img = b * V + (j == 0 ? v : (V - 1 - v));
        const Taps tp = make_taps(g.x, g.y, Wl, Hl, j == 0);
        const float* m = base + (size_t)img * Hl * Wl * C + cl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 tex = *reinterpret_cast<const f32x4*>(m + (size_t)tp.off[k] * C);
            out += tex * tp.w[k];
        }
    } else if (c0 == 832) {
        const float* pe = pe6 + sidx * 6 + j * 3;
        out = f32x4{pe[0], pe[1], pe[2], 0.f};
    }
    *reinterpret_cast<f32x4*>(o) = out;
}

extern "C" int cpn_gather_rows_f32(const float* map0, const float* map1, const float* map2, const float* map3, int H, int W,
                                   const float* pixel_val, const float* sec_grid, const float* pe6, int B, int V, int R, int S,
                                   int ray0, int nrays, float* xin, int ld, void* stream) {
    CPN_REQUIRE(map0 && map1 && map2 && map3 && pixel_val && sec_grid && pe6 && xin, CPN_E_ARG, "cpn_gather_rows_f32: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0 && ld >= 836 && (ld % 4) == 0,
                CPN_E_SHAPE, "cpn_gather_rows_f32: need V==2, H,W multiples of 16, ld >= 836 and a multiple of 4");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_gather_rows_f32: ray range outside B*R");
    const long long nrows2 = (long long)nrays * V * S * 2;
    const long long total = nrows2 * (ld >> 2);
    CPN_REQUIRE(total / 256 + 1 < (1LL << 31), CPN_E_SHAPE, "cpn_gather_rows_f32: chunk too large");
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, map0, map1,
                       map2, map3, H, W, pixel_val, sec_grid, pe6, V, R, S, ray0, nrows2, xin, ld);
    CPN_LAUNCH_CHECK("cpn_gather_rows_f32");
    return 0;
}

extern "C" int cpn_local_hidden(const float* loc8, const float* coords9, const float* w, int ldw, const float* bias,
                                const float* add, int B, int V, int R, int S, int ray0, int nrays, uint16_t* out,
                                void* stream) {
    CPN_REQUIRE(loc8 && coords9 && w && bias && out, CPN_E_ARG, "cpn_local_hidden: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && ldw >= 16, CPN_E_SHAPE, "cpn_local_hidden: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_local_hidden: ray range outside B*R");
    const long long nrows = (long long)nrays * V * S;
    CPN_REQUIRE(nrows * 16 < (1LL << 31), CPN_E_SHAPE, "cpn_local_hidden: chunk too large for 32-bit indexing");
    const long long groups = cpn_cdiv(nrows, 16);                   // 16 rows per wave step, 4 waves per block
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(groups, 4), 2048);
    hipLaunchKernelGGL(local_hidden_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       loc8, coords9, w, ldw, bias, add, V, R, S, ray0, nrows, (__half*)out);
    CPN_LAUNCH_CHECK("cpn_local_hidden");
    return 0;
}

extern "C" int cpn_local_mlp(const float* loc8, const float* coords9, const float* w1, int ldw1, const float* b1,
                             const float* add, const uint16_t* w2, int ldw2, const float* b2, int B, int V, int R, int S,
                             int ray0, int nrays, uint16_t* out, const uint16_t* dot_with, float* logits_out,
                             int rows_frag, void* stream) {
    CPN_REQUIRE(loc8 && coords9 && w1 && b1 && w2 && b2 && (out || (dot_with && logits_out)), CPN_E_ARG,
                "cpn_local_mlp: null pointer");
    CPN_REQUIRE(!logits_out || dot_with, CPN_E_ARG, "cpn_local_mlp: logits_out needs dot_with");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && ldw1 >= 16 && ldw2 >= 128 && (ldw2 % 8) == 0, CPN_E_SHAPE,
                "cpn_local_mlp: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_local_mlp: ray range outside B*R");
    const long long nrows = (long long)nrays * V * S;
    CPN_REQUIRE(nrows * 16 < (1LL << 31), CPN_E_SHAPE, "cpn_local_mlp: chunk too large for 32-bit indexing");
    const long long groups = cpn_cdiv(nrows, 16);
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(groups, 8), 1024);
    hipLaunchKernelGGL(local_mlp_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, loc8, coords9, w1, ldw1, b1, add,
                       (const __half*)w2, ldw2, b2, V, R, S, ray0, nrows, (__half*)out, (const __half*)dot_with, logits_out,
                       rows_frag);
    CPN_LAUNCH_CHECK("cpn_local_mlp");
    return 0;
}
