// K2+K3a — first encoder layer straight from the feature maps: hid = ReLU(query_encode_latent([gather ‖ tanh(pt/5)])).
//
// Replaces F.grid_sample x 8 + torch.cat + the 835 -> 832 1x1 convolution + ReLU
// (/root/reference models/CoPoNeRF.py:312, 370, 384-397 with the layer of :71) WITHOUT materialising the gathered
// 835-channel encoder input: a 1x1 convolution and bilinear interpolation commute (both are linear, the bias is
// added after the interpolation, so it holds for 'border' and for 'zeros' padding alike; SURVEY.md §7.4 tier B):
//
//     W . sum_t a_t tex_t  =  sum_t a_t (W . tex_t)
//
// The three coarse levels (256 channels each at H/16, H/8, H/4) are therefore projected ONCE per stereo pair
// through their column block of the weight (cpn_gemm_f16 on the NHWC maps: 4.6 GFLOP per 256^2 pair instead of
// 4.5 TFLOP per 16 384 rays) into fp16 tables P_l[texel][832], and a row of the layer becomes
//
//     hid[row] = ReLU( sum_{l<3} sum_{t<4} a_{l,t} P_l[tex_{l,t}]  +  W[:, 768:835] . [gather_3(64 ch) ‖ tanh(pt/5)]  +  b )
//
// i.e. 12 table taps (VALU, fp32 accumulation) + a K = 96 MFMA product for the full-resolution level, whose table
// would be 13x the map (218 MB per pair) and is kept as a contraction.  Numerically the table form is equivalent
// to rounding the gathered features to fp16 (one fp16 rounding per tap of a 256-term sum instead of one per
// channel): rms error of the pre-activation 1.39e-4 either way on N(0,1) features.
//
// gfx950 design
//   * workgroup = 128 consecutive rows (one (ray, view) at S = 64: 64 samples x {own image, other image}), 4 waves,
//     two workgroups per CU; wave w owns rows 32w .. 32w+31 = 16 samples x 2 and walks the four 208-channel slices
//     of the output (2 x 13 accumulator tiles of v_mfma_f32_16x16x32_f16, weights as the A operand: a lane holds 4
//     consecutive channels of one row per tile).
//   * MFMA tile mt <-> j (own / other image), column r <-> sample: lane (r, g) owns rows (s, 0) and (s, 1).
//   * the channel -> (tile, register) assignment is chosen so that a lane's 52 channels of a slice are 6 x 8
//     consecutive ones (k*32 + g*8 .. +8) + 4 (192 + g*4 ..): a table tap is then 6 16-byte loads + 1 8-byte load
//     per lane, the 4 lanes of a row reading 64 contiguous, 64-byte aligned bytes (the tables are stored with
//     each 208-channel slice padded to 224 halves for that), and the fp16 row leaves as 16-byte stores.
//   * per-row tap offsets / weights are computed once per row (not per lane) into LDS; the full-resolution level
//     and the point encoding are gathered once per row into an LDS image in MFMA B-operand order.
//   * weight fragments of a slice (39 KiB) stream through LDS by buffer_load ... lds while the previous slice's
//     table taps are being accumulated.
// Bound: vector L1 / texture-address throughput of the tap loads (12 x 1.75 KiB per row, mostly L1/L2 hits: the
// tables of a pair are 19 MB) and the 7 GB hid write stream; algorithmic FLOPs of the layer it replaces
// 2*835*832 per row.
#include <algorithm>

#include "common.h"
#include "taps.h"

// timing-only ablations for tools/encode_ablate.py (results are wrong when non-zero; the product builds with 0):
// 1 = no table taps, 2 = no hid stores, 4 = no MFMA phase, 8 = taps of level 0 only, 16 = every tap reads texel 0
#ifndef CPN_ENCODE_ABLATE
#define CPN_ENCODE_ABLATE 0
#endif

namespace {

constexpr int TILE_ROWS = 128;
constexpr int NSLICE = 4;                 // 832 = 4 x 208 output channels
constexpr int SLICE_CH = 208;
constexpr int NT = 13;                    // 16-channel MFMA tiles per slice
constexpr int KSTEPS = 3;                 // K = 96 = 64 level-3 channels + 3 point encodings + zeros
constexpr int TAB_SLICE_BYTES = CPN_TAB_SLICE * 2;             // 448: 384 main + 64 tail (4 x {8 B used, 8 B pad})
constexpr int TAB_ROW_BYTES = CPN_TAB_LD * 2;                  // 1792 per texel
constexpr int TAPS_BYTES = TILE_ROWS * 4 * 32;                 // [row][level]{int off[4]; float w[4]}
constexpr int AIMG_BYTES = 2 * 4 * 2 * 1024 + 4 * 2 * 256;     // [k < 2][wave][mt][lane] half8 + [wave][mt][r] half8 (k = 2, g = 0)
constexpr int AIMG2_OFF = 2 * 4 * 2 * 1024;                    // third K step: only lane group 0 holds data (pt enc)
constexpr int WIMG_BYTES = KSTEPS * NT * 1024;                 // [k][nt][lane] half8
constexpr int LDS_BYTES = TAPS_BYTES + AIMG_BYTES + WIMG_BYTES + 832 * 4;

typedef __attribute__((address_space(3))) void lds_void;

struct TapRec {
    int off[4];
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
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Which (ray, sample) a tile row is.  A tile = TG adjacent rays x TSB consecutive samples x {own, other image} of one
// view = 128 rows; wave w takes samples 4w..4w+3 of the block and MFMA column r = (sample & 3)*4 + (ray & 3), so ONE
// load instruction covers a 4 x 4 patch of (sample, ray): adjacent query pixels have almost the same epipolar line
// and adjacent samples sit <= 1.5 texels apart, i.e. the 16 rows of an instruction share a handful of texels and
// most of its 64-byte requests hit lines that a neighbouring lane just brought into the vector L1 (rays of a
// full-image render are row-major pixels; any other ray order is still correct, only less cache friendly).
constexpr int TG = 4;                     // rays per tile
constexpr int TSB = 16;                   // samples per tile

struct RowId {
    bool live;
    int j, s;
    unsigned rayl;                        // ray inside this launch
};
__device__ __forceinline__ RowId tile_row(int row, unsigned group, int blk, int S, unsigned nrays) {
    const int w = row >> 5, rl = row & 31, rs = rl >> 1;
    RowId o;
    o.j = rl & 1;
    o.s = blk * TSB + w * 4 + (rs >> 2);
    o.rayl = group * TG + (rs & 3);
    o.live = (o.s < S) && (o.rayl < nrays);
    return o;
}

__global__ __launch_bounds__(256, 2) void encode_hidden_kernel(
    const __half* __restrict__ tab0, const __half* __restrict__ tab1, const __half* __restrict__ tab2,
    long long tab0_bytes, long long tab1_bytes, long long tab2_bytes, const __half* __restrict__ map3, int H, int W,
    const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, const float* __restrict__ pe6,
    const half8* __restrict__ wfrag, const float* __restrict__ bias, int V, int R, int S, int ray0, unsigned nrays,
    int nblk, __half* __restrict__ hid) {
    // three separate LDS objects (not one dynamic array): the compiler then knows that reads of the tap records do
    // not alias the in-flight buffer_load ... lds of the weight image and does not drain vmcnt in front of them
    __shared__ __attribute__((aligned(16))) TapRec taps[TILE_ROWS * 4];
    __shared__ __attribute__((aligned(16))) char aimg[AIMG_BYTES];
    __shared__ __attribute__((aligned(16))) char wimg[WIMG_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_s[832];     // keeps the slice loop free of global loads up to the taps

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // XCD-aware order: blocks go round-robin over the 8 XCDs; each XCD takes a contiguous range of tiles
    // (= neighbouring rays = overlapping texel footprints), so its private L2 works on 1/8 of the tables
    const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, q = nb >> 3, rem = nb & 7;
    const unsigned tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    const int blk = (int)(tile % (unsigned)nblk);
    const int v = (int)((tile / (unsigned)nblk) % (unsigned)V);
    const unsigned group = tile / (unsigned)(nblk * V);

    // weight fragments of slice n -> LDS (lane-linear 1 KiB pieces, wave w moves pieces w, w+4, ...)
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)wfrag, 0, NSLICE * WIMG_BYTES, 0x00020000);
    auto stage_w = [&](int n) {
#pragma unroll
        for (int i = 0; i < (KSTEPS * NT + 3) / 4; ++i) {
            const int piece = wave + 4 * i;
            if (piece < KSTEPS * NT)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(wimg + piece * 1024), 16, lane * 16,
                                                         n * WIMG_BYTES + piece * 1024, 0, 0);
        }
    };

    // ---- phase A: tap records, thread = (row, level pair) ----------------------------------------------------
    for (int i = tid; i < 832 / 4; i += 256)
        *reinterpret_cast<f32x4*>(bias_s + i * 4) = *reinterpret_cast<const f32x4*>(bias + i * 4);
    {
        const int row = tid >> 1, half = tid & 1;
        const RowId id = tile_row(row, group, blk, S, nrays);
        const unsigned ray = (unsigned)ray0 + (id.rayl < nrays ? id.rayl : nrays - 1);
        const int b = (int)(ray / (unsigned)R), rr = (int)(ray % (unsigned)R);
        const size_t sidx = (((size_t)(b * V + v)) * R + rr) * S + (id.s < S ? id.s : S - 1);
        const float2 gq = *reinterpret_cast<const float2*>((id.j == 0 ? pixel_val : sec_grid) + sidx * 2);
        const int img = b * V + (id.j == 0 ? v : (V - 1 - v));
#pragma unroll
        for (int li = 0; li < 2; ++li) {
            const int lvl = half * 2 + li;
            const int shift = 4 - lvl - (lvl == 3);
            const int Hl = H >> shift, Wl = W >> shift;
            const Taps tp = make_taps(gq.x, gq.y, Wl, Hl, id.j == 0);
            const int entry = (lvl == 3) ? 128 : TAB_ROW_BYTES;         // bytes per texel
            TapRec rec;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rec.off[k] = (img * Hl * Wl + tp.off[k]) * entry;
                rec.w[k] = id.live ? tp.w[k] : 0.0f;
            }
            taps[row * 4 + lvl] = rec;
        }
    }
    __syncthreads();
    stage_w(0);               // lands under phase B

    // ---- phase B: the K = 96 operand image: level-3 gather (64 ch) ‖ tanh(pt/5) (3) ‖ zeros --------------------
    // B-operand order: fragment (k, wave, mt) is 1 KiB, lane (r, g) reads its 16 bytes at lane*16
    {
        const int row = tid >> 1, half = tid & 1;
        const int w_ = row >> 5, rl = row & 31, rs = rl >> 1, mt = rl & 1;
        const TapRec rec = taps[row * 4 + 3];
        const char* m3 = reinterpret_cast<const char*>(map3) + half * 64;
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
            const RowId id = tile_row(row, group, blk, S, nrays);
            half8 p8;
#pragma unroll
            for (int e = 0; e < 8; ++e) p8[e] = (_Float16)0.0f;
            if (id.live) {
                const unsigned ray = (unsigned)ray0 + id.rayl;
                const int b = (int)(ray / (unsigned)R), rr = (int)(ray % (unsigned)R);
                const size_t sidx = (((size_t)(b * V + v)) * R + rr) * S + id.s;
                const float* pe = pe6 + sidx * 6 + id.j * 3;
                p8[0] = (_Float16)pe[0]; p8[1] = (_Float16)pe[1]; p8[2] = (_Float16)pe[2];
            }
            *reinterpret_cast<half8*>(aimg + AIMG2_OFF + ((w_ * 2 + mt) * 16 + rs) * 16) = p8;
        }
    }
    __syncthreads();          // (drains vmcnt: slice 0 of the weights has landed too)

    half8 xa[KSTEPS][2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            xa[k][mt] = *reinterpret_cast<const half8*>(aimg + (((k * 4 + wave) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const half8 p8 = *reinterpret_cast<const half8*>(aimg + AIMG2_OFF + ((wave * 2 + mt) * 16 + r) * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) xa[2][mt][e] = (g == 0) ? p8[e] : (_Float16)0.0f;
    }

    // rows of this lane: tile rows 32*wave + 2*r + mt, mt = 0 (own image), 1 (other image)
    const RowId lid = tile_row(wave * 32 + 2 * r, group, blk, S, nrays);
    const size_t lrow0 = (((size_t)lid.rayl * V + v) * S + lid.s) * 2;                 // + mt
    const __amdgpu_buffer_rsrc_t trs0 = __builtin_amdgcn_make_buffer_rsrc((void*)tab0, 0, (int)tab0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t trs1 = __builtin_amdgcn_make_buffer_rsrc((void*)tab1, 0, (int)tab1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t trs2 = __builtin_amdgcn_make_buffer_rsrc((void*)tab2, 0, (int)tab2_bytes, 0x00020000);

    for (int n = 0; n < NSLICE; ++n) {
        f32x4 acc[2][NT];
        // bias of the lane's 52 channels (natural channel order)
        {
            const float* bp = bias_s + n * SLICE_CH;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                acc[0][2 * k] = *reinterpret_cast<const f32x4*>(bp + k * 32 + g * 8);
                acc[0][2 * k + 1] = *reinterpret_cast<const f32x4*>(bp + k * 32 + g * 8 + 4);
            }
            acc[0][12] = *reinterpret_cast<const f32x4*>(bp + 192 + g * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[1][nt] = acc[0][nt];
        }
        // ---- K = 96 contraction of the full-resolution level + point encoding
#pragma unroll
        for (int k = 0; k < ((CPN_ENCODE_ABLATE & 4) ? 0 : KSTEPS); ++k)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8 wb = *reinterpret_cast<const half8*>(wimg + ((k * NT + nt) * 64 + lane) * 16);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[k][0], acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb, xa[k][1], acc[1][nt], 0, 0, 0);
            }
        // every wave has read this slice's fragments.  Raw s_barrier, not __syncthreads(): the fence of the latter
        // drains vmcnt to 0, i.e. it would wait here for the previous slice's hid stores to reach memory
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (n + 1 < NSLICE) stage_w(n + 1);     // the next slice lands under the table taps below

        // ---- 12 table taps per row, accumulated in fp32 on top of the MFMA result.  Order: 128-byte line (two
        //      16-byte pieces per lane) outermost, the four taps inside: the taps of neighbouring rows that fall on
        //      the same texel request the same cache line back to back
        const int col_off = n * TAB_SLICE_BYTES + g * 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int trow = wave * 32 + 2 * r + mt;
#pragma unroll
            for (int lvl = 0; lvl < ((CPN_ENCODE_ABLATE & 1) ? 0 : (CPN_ENCODE_ABLATE & 8) ? 1 : 3); ++lvl) {
                const TapRec rec = taps[trow * 4 + lvl];
                const __amdgpu_buffer_rsrc_t rs = lvl == 0 ? trs0 : (lvl == 1 ? trs1 : trs2);
                int vo[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) vo[k] = ((CPN_ENCODE_ABLATE & 16) ? 0 : rec.off[k]) + col_off;
#pragma unroll
                for (int cp = 0; cp < 3; ++cp) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x4 d0 = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[k] + cp * 128, 0, 0);
                        const u32x4 d1 = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[k] + cp * 128 + 64, 0, 0);
                        const float wk = rec.w[k];
                        f32x4* a4 = &acc[mt][4 * cp];
                        a4[0][0] = fma_mix_lo(a4[0][0], d0[0], wk); a4[0][1] = fma_mix_hi(a4[0][1], d0[0], wk);
                        a4[0][2] = fma_mix_lo(a4[0][2], d0[1], wk); a4[0][3] = fma_mix_hi(a4[0][3], d0[1], wk);
                        a4[1][0] = fma_mix_lo(a4[1][0], d0[2], wk); a4[1][1] = fma_mix_hi(a4[1][1], d0[2], wk);
                        a4[1][2] = fma_mix_lo(a4[1][2], d0[3], wk); a4[1][3] = fma_mix_hi(a4[1][3], d0[3], wk);
                        a4[2][0] = fma_mix_lo(a4[2][0], d1[0], wk); a4[2][1] = fma_mix_hi(a4[2][1], d1[0], wk);
                        a4[2][2] = fma_mix_lo(a4[2][2], d1[1], wk); a4[2][3] = fma_mix_hi(a4[2][3], d1[1], wk);
                        a4[3][0] = fma_mix_lo(a4[3][0], d1[2], wk); a4[3][1] = fma_mix_hi(a4[3][1], d1[2], wk);
                        a4[3][2] = fma_mix_lo(a4[3][2], d1[3], wk); a4[3][3] = fma_mix_hi(a4[3][3], d1[3], wk);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x2 dt = __builtin_amdgcn_raw_buffer_load_b64(rs, vo[k] + 384, 0, 0);
                    const float wk = rec.w[k];
                    acc[mt][12][0] = fma_mix_lo(acc[mt][12][0], dt[0], wk);
                    acc[mt][12][1] = fma_mix_hi(acc[mt][12][1], dt[0], wk);
                    acc[mt][12][2] = fma_mix_lo(acc[mt][12][2], dt[1], wk);
                    acc[mt][12][3] = fma_mix_hi(acc[mt][12][3], dt[1], wk);
                }
            }
            // ---- ReLU, fp16, store (natural channel order, 16-byte pieces, 64 contiguous bytes per row and k)
            if (lid.live && (!(CPN_ENCODE_ABLATE & 2) || acc[mt][0][0] == 123.456f)) {
                __half* orow = hid + (lrow0 + mt) * 832 + n * SLICE_CH;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    half8 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o[i] = (_Float16)fmaxf(acc[mt][2 * c][i], 0.0f);
                        o[4 + i] = (_Float16)fmaxf(acc[mt][2 * c + 1][i], 0.0f);
                    }
                    *reinterpret_cast<half8*>(orow + c * 32 + g * 8) = o;
                }
                half4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = (_Float16)fmaxf(acc[mt][12][i], 0.0f);
                *reinterpret_cast<half4*>(orow + 192 + g * 4) = o4;
            }
        }
        // The next slice's fragments are in LDS: this wave's DMA pieces were issued BEFORE its tap loads, loads
        // retire in order and every tap load has been consumed above, so only the barrier is needed — no vmcnt(0)
        // that would also wait for the hid stores just issued
        if (n + 1 < NSLICE) __builtin_amdgcn_s_barrier();
    }
}

// ---- weight images ------------------------------------------------------------------------------------------------
// channel of a slice that MFMA tile nt, A-operand row a computes (lane (r, g) of the result then holds a = g*4 + i)
__host__ __device__ inline int slice_channel(int nt, int a) {
    return nt < 12 ? (nt >> 1) * 32 + (a >> 2) * 8 + (nt & 1) * 4 + (a & 3) : 192 + a;
}

// W (832, 835) fp32 -> wfrag [slice][k][nt][lane] half8 over columns 768..834 (K padded to 96)
__global__ void pack_encode_frag_kernel(const float* __restrict__ W, int ldw, half8* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NSLICE * KSTEPS * NT * 64) return;
    const int lane = idx & 63;
    int t = idx >> 6;
    const int nt = t % NT; t /= NT;
    const int k = t % KSTEPS;
    const int n = t / KSTEPS;
    const int ch = n * SLICE_CH + slice_channel(nt, lane & 15);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = k * 32 + (lane >> 4) * 8 + e;
        o[e] = (_Float16)(kk < 67 ? W[(size_t)ch * ldw + 768 + kk] : 0.0f);
    }
    out[idx] = o;
}

// W (832, 835) fp32 -> the table projection of level l: (CPN_TAB_LD, 256) fp16, table column c' -> channel
//   slice n = c' / 224, q = c' % 224:  q < 192 -> n*208 + q ;  else u = q - 192: (u & 7) < 4 -> n*208 + 192 + (u>>3)*4 + (u&7)
//   (the 4 lanes of a row read 8 bytes each at 16-byte pitch), else a zero row
__global__ void pack_table_weight_kernel(const float* __restrict__ W, int ldw, int lvl, __half* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= CPN_TAB_LD * 256) return;
    const int kc = idx & 255, cp = idx >> 8;
    const int n = cp / CPN_TAB_SLICE, qq = cp % CPN_TAB_SLICE;
    int ch = -1;
    if (qq < 192) ch = n * SLICE_CH + qq;
    else {
        const int u = qq - 192;
        if ((u & 7) < 4) ch = n * SLICE_CH + 192 + (u >> 3) * 4 + (u & 7);
    }
    out[idx] = __float2half(ch >= 0 ? W[(size_t)ch * ldw + lvl * 256 + kc] : 0.0f);
}

}  // namespace

extern "C" int cpn_pack_encode_weights(const float* W, int ldw, uint16_t* wfrag, uint16_t* wtab0, uint16_t* wtab1,
                                       uint16_t* wtab2, void* stream) {
    CPN_REQUIRE(W && wfrag && wtab0 && wtab1 && wtab2, CPN_E_ARG, "cpn_pack_encode_weights: null pointer");
    CPN_REQUIRE(ldw >= 835, CPN_E_SHAPE, "cpn_pack_encode_weights: ldw=%d < 835", ldw);
    const hipStream_t s = (hipStream_t)stream;
    const int nf = NSLICE * KSTEPS * NT * 64;
    hipLaunchKernelGGL(pack_encode_frag_kernel, dim3(cpn_cdiv(nf, 256)), dim3(256), 0, s, W, ldw, (half8*)wfrag);
    uint16_t* tabs[3] = {wtab0, wtab1, wtab2};
    for (int l = 0; l < 3; ++l)
        hipLaunchKernelGGL(pack_table_weight_kernel, dim3(cpn_cdiv(CPN_TAB_LD * 256, 256)), dim3(256), 0, s, W, ldw, l,
                           (__half*)tabs[l]);
    CPN_LAUNCH_CHECK("cpn_pack_encode_weights");
    return 0;
}

extern "C" int cpn_encode_hidden(const uint16_t* tab0, const uint16_t* tab1, const uint16_t* tab2, const uint16_t* map3,
                                 int H, int W, const float* pixel_val, const float* sec_grid, const float* pe6,
                                 const uint16_t* wfrag, const float* bias, int B, int V, int R, int S, int ray0,
                                 int nrays, uint16_t* hid, void* stream) {
    CPN_REQUIRE(tab0 && tab1 && tab2 && map3 && pixel_val && sec_grid && pe6 && wfrag && bias && hid, CPN_E_ARG,
                "cpn_encode_hidden: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0,
                CPN_E_SHAPE, "cpn_encode_hidden: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_encode_hidden: ray range [%d,%d) outside B*R=%lld", ray0, ray0 + nrays, (long long)B * R);
    const long long nrows = (long long)nrays * V * S * 2;
    const long long nimg = (long long)B * V;
    const long long t2 = nimg * (H / 4) * (W / 4) * TAB_ROW_BYTES, t1 = t2 / 4, t0 = t2 / 16;
    CPN_REQUIRE(nrows < (1LL << 31) && t2 < (1LL << 31) && nimg * H * W * 128 < (1LL << 31), CPN_E_SHAPE,
                "cpn_encode_hidden: chunk / tables too large for 32-bit offsets (%lld rows, %lld table bytes)", nrows, t2);
    CPN_REQUIRE(((uintptr_t)tab0 % 16) == 0 && ((uintptr_t)tab1 % 16) == 0 && ((uintptr_t)tab2 % 16) == 0 &&
                    ((uintptr_t)map3 % 16) == 0 && ((uintptr_t)wfrag % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                    ((uintptr_t)hid % 16) == 0, CPN_E_ARG, "cpn_encode_hidden: pointers must be 16-byte aligned");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    const int nblk = (int)cpn_cdiv(S, TSB);
    const long long tiles = (long long)cpn_cdiv(nrays, TG) * V * nblk;
    CPN_REQUIRE(tiles < (1LL << 31), CPN_E_SHAPE, "cpn_encode_hidden: %lld tiles exceed the grid limit", tiles);
    hipLaunchKernelGGL(encode_hidden_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)tab0, (const __half*)tab1, (const __half*)tab2, t0, t1, t2, (const __half*)map3, H, W,
                       pixel_val, sec_grid, pe6, (const half8*)wfrag, bias, V, R, S, ray0, (unsigned)nrays, nblk,
                       (__half*)hid);
    CPN_LAUNCH_CHECK("cpn_encode_hidden");
    return 0;
}
