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
// gfx950 design (v6; the counter trail of v1..v6 is DESIGN.md §4.1)
//   * persistent workgroups, one per CU, 16 waves (4 per SIMD, <= 128 VGPRs).  All K = 80 weight fragments of the layer
//     (13 slices x [2 x half8 + a half4 tail] per lane and tile = 123 KiB; the bias rides in the tail as an fp16
//     (hi, lo) pair against two constant-one operand entries) are loaded into LDS ONCE per workgroup; one
//     __syncthreads() after that and none in the tile loop.
//   * unit of work = a 16-row WAVE tile: 4 adjacent rays x 4 consecutive samples of one (view, image).  The wave-tile
//     range is split over the XCDs that received a workgroup (blockIdx % 8), each XCD's workgroups walk their range in
//     lock step, so the 4 x 4 (sample, ray) patches a private L2 sees at one time are neighbours on the epipolar lines.
//   * two register layouts of the same 16 x 64 tile: the "load layout" (lane = 4*row + piece: the 4 lanes of a quad
//     read 64 contiguous, 64-byte aligned bytes of one row - the texture addresser takes a wave 4 lanes per cycle, so
//     a quad = one tag lookup) for table taps and level-3 gather, and the MFMA layout (lane = row + 16*group) for the
//     contraction; ds_bpermute switches between them (operands once per tile, accumulators once per slice).
//   * per tile: tap offsets / weights (node_taps), the 64 full-resolution channels (4 texels, fp32 blend) and
//     tanh(pt/5) are computed in registers; then 13 slices of 64 channels: 8 tap loads of slice n are issued, THEN
//     the stores of slice n-1 (gfx9 has one vmcnt: a wait on a load younger than a store waits for the store), then
//     3 MFMAs per 16-channel tile on the LDS-resident weights (v_mfma_f32_16x16x32_f16 x 2 + v_mfma_f32_16x16x16f16
//     for the tail; C = inline 0) run while the taps are in flight, then the 4-tap fp32 blend (v_fma_mix_f32 takes
//     the fp16 table entry directly), v_cvt_pk_f16_f32 + v_pk_max_f16 (ReLU).
//   * hid leaves as NON-TEMPORAL stores of WHOLE 128-byte lines: neighbouring quads swap one 64-byte half with two
//     DPP row shifts so that 8 consecutive lanes cover one line.  Write-back stores let the 7 GB stream evict the
//     tables from L2 (2.7 ms), nt stores of 64-byte pieces run at 3.6 TB/s, whole lines at 6.2 TB/s (tools/write_bw.py).
// Bound (rocprofv3 PMC, profiles/r02_*): the 7 GB hid write stream per 16 384 rays (1.1 ms at the 6.2 TB/s the part
// sustains for this pattern) behind VALU (4-tap blend) + MFMA + LDS issue; algorithmic FLOPs of the layer it
// replaces: 2*835*832 per row.
#include <algorithm>

// timing-only ablations for tools/encode_ablate.py (results are wrong when non-zero; the product builds with 0):
// 1 = no table taps, 2 = no hid stores, 4 = no MFMA phase, 16 = every tap reads node 0, 32 = stores wrap into a 1.7 MB window
#ifndef CPN_ENCODE_ABLATE
#define CPN_ENCODE_ABLATE 0
#endif
// 1 = the table taps of the next 64-channel slice are issued before the current slice is computed
#ifndef CPN_ENCODE_PREFETCH
#define CPN_ENCODE_PREFETCH 0
#endif
// images (own / other) per wave tile: 2 = 32-row wave tiles, 8 waves per CU; 1 = 16-row wave tiles, 16 waves per CU
#ifndef CPN_ENCODE_MT
#define CPN_ENCODE_MT 1
#endif

#include "encode_common.h"

namespace {

constexpr int WMAIN_HALF8 = NSLICE * 2 * NT * 64;              // [slice][k < 2][tile][lane] half8: 104 KiB
constexpr int WTAIL_HALF4 = NSLICE * NT * 48;                  // [slice][tile][K group < 3][A-operand row] half4: 19.5 KiB
constexpr int MTN = CPN_ENCODE_MT;
#ifndef CPN_ENCODE_WAVES
#define CPN_ENCODE_WAVES (16 / CPN_ENCODE_MT)
#endif
constexpr int ENC_WAVES = CPN_ENCODE_WAVES;

// Persistent kernel: one workgroup per CU keeps ALL weight fragments of the K = 80 contraction in LDS for the whole
// launch (123 KiB: they are the same for every row; as per-wave L2 loads they were 35 % of the bytes through the
// texture path, which rocprofv3 showed 82 % busy), and every wave walks its own sequence of wave tiles with no
// workgroup barrier after the prologue.  Nothing else lives in LDS: the per-row tap records and the K = 80 operand
// (full-resolution gather + point encoding) are produced in registers, in the load layout, and moved to the MFMA
// layout with ds_bpermute.
__global__ __launch_bounds__(64 * ENC_WAVES, 1) void encode_hidden_kernel(
    const __half* __restrict__ tab, const __half* __restrict__ map3, int H, int W,
    const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, const float* __restrict__ pe6,
    const half8* __restrict__ wfrag, const float* __restrict__ bias, int V, int R, int S, int ray0, int nrays,
    int nsblk, int groups_per_b, long long group0, long long nwtiles, __half* __restrict__ hid) {
    __shared__ __attribute__((aligned(16))) half8 wmain[WMAIN_HALF8];
    __shared__ __attribute__((aligned(16))) half4 wtail_s[WTAIL_HALF4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < WMAIN_HALF8; i += 64 * ENC_WAVES) wmain[i] = wfrag[i];
    {
        // K tail (16 wide): k = 0..2 the point encoding, k = 3 and 4 the bias as an fp16 (hi, lo) pair against two
        // constant 1.0 entries of the operand (fp32-accurate; the accumulators then start from the inline constant 0),
        // k >= 8 zero.  In memory [slice][tile][lane] half4 with only the weights filled in; lane groups 2, 3 share one
        // zero image in LDS.
        const half4* tsrc = reinterpret_cast<const half4*>(wfrag + WMAIN_HALF8);
        for (int i = tid; i < WTAIL_HALF4; i += 64 * ENC_WAVES) {
            const int f = i / 48, l = i - f * 48;
            half4 t = tsrc[f * 64 + l];
            if (l < 32) {
                const float bv = bias[(f / NT) * SLICE_CH + slice_channel(f % NT, l & 15)];
                const _Float16 hi = (_Float16)bv;
                if (l < 16) t[3] = hi;
                else t[0] = (_Float16)(bv - (float)hi);
            }
            wtail_s[i] = t;
        }
    }
    __syncthreads();          // the only workgroup-wide barrier: from here on the eight waves run independently

    const int r = lane & 15, g = lane >> 4;                   // MFMA layout: column (row of the tile) r, K / channel group g
    const int rl = lane >> 2, pl = lane & 3;                  // load layout: row rl, 16-byte piece pl (see below)
    const int tail_lane = min(g, 2) * 16 + r;                 // K-tail fragment: groups 2 and 3 read the shared zero image
    const int to_ll = (rl + 16 * pl) * 4;                     // ds_bpermute address: this lane takes MFMA lane (r = rl, g = pl)
    const int to_mfma = (4 * r + g) * 4;                      //                      this lane takes load-layout lane (rl = r, pl = g)
    const NodeGrid ng{W >> 1, H >> 1};
    const size_t img_bytes = (size_t)ng.nodes_per_image() * TAB_ROW_BYTES;
    const char* const tbase = reinterpret_cast<const char*>(tab);
    const char* const m3base = reinterpret_cast<const char*>(map3);

    // XCD-aware, CU-local order: blocks go round-robin over the 8 XCDs; XCD x owns a contiguous range of wave tiles and
    // its workgroups take them in lock step, the 8 waves of a workgroup 8 consecutive ones (= neighbouring samples of
    // the same 4 rays, then the next rays): one CU's L1 and one XCD's L2 see overlapping node footprints
    const unsigned nbk = gridDim.x, nx = nbk < 8 ? nbk : 8;        // tile ranges: one per XCD that received a block
    const unsigned xcd = blockIdx.x % nx, wgx = blockIdx.x / nx;
    const unsigned wg_on_xcd = nbk / nx + (xcd < nbk % nx ? 1 : 0);
    const long long q = nwtiles / nx, rem = nwtiles % nx;
    const long long x_begin = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const long long x_end = x_begin + q + (xcd < rem ? 1 : 0);

    for (long long wt = x_begin + (long long)wgx * ENC_WAVES + wave; wt < x_end; wt += (long long)wg_on_xcd * ENC_WAVES) {
        // ---- decode the wave tile: (ray group, view, block of 4 samples)
        const long long wu = MTN == 2 ? wt : (wt >> 1);
        const int j0 = MTN == 2 ? 0 : (int)(wt & 1);          // image of tile row block 0 (wave-uniform)
        const int sblk = (int)(wu % nsblk);
        const int v = (int)((wu / nsblk) % V);
        const long long gq = group0 + wu / ((long long)nsblk * V);
        const int b = (int)(gq / groups_per_b), rgroup = (int)(gq % groups_per_b);
        const int img_own = b * V + v, img_oth = b * V + (V - 1 - v);

        // ---- per-row records in the LOAD layout (lane = 4*rl + pl: 4 adjacent lanes = 64 contiguous bytes of one row;
        //      the texture addresser walks a wave 4 lanes per cycle and the L1 does one tag lookup per distinct line of
        //      such a quad).  mt = 0: own image (border padding, pixel_val), mt = 1: other image (zeros, sec_grid).
        const RowId lid = tile_row(rl, rgroup, sblk, S, R, b, ray0, nrays);
        const size_t sidx_l = (((size_t)(b * V + v)) * R + min(lid.r, R - 1)) * S + min(lid.s, S - 1);
        TapRec rec[MTN];
        half8 xl[2][MTN];                                     // K = 64 operand pieces of this lane's row, load layout
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            const bool own = (j0 + mt) == 0;
            const float2 gc = *reinterpret_cast<const float2*>((own ? pixel_val : sec_grid) + sidx_l * 2);
            rec[mt] = node_taps(gc.x, gc.y, ng, own);
            const Taps t3 = make_taps(gc.x, gc.y, W, H, own);
            const char* m3 = m3base + (size_t)(own ? img_own : img_oth) * H * W * 128 + pl * 16;
            u32x4 tv[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    tv[k][t] = *reinterpret_cast<const u32x4*>(m3 + (size_t)(unsigned)t3.off[t] * 128 + k * 64);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        a8[2 * i] = fma_mix_lo(a8[2 * i], tv[k][t][i], t3.w[t]);
                        a8[2 * i + 1] = fma_mix_hi(a8[2 * i + 1], tv[k][t][i], t3.w[t]);
                    }
#pragma unroll
                for (int e = 0; e < 8; ++e) xl[k][mt][e] = lid.live ? (_Float16)a8[e] : (_Float16)0.0f;
            }
            if (!lid.live) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { rec[mt].off[t] = 0; rec[mt].w[t] = 0.0f; }
            }
        }
        // ---- the same operand in the MFMA layout (B operand: lane (r, g) holds K = g*8 .. g*8+7 of row r)
        half8 xa[2][MTN];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
                const u32x4 src = __builtin_bit_cast(u32x4, xl[k][mt]);
                u32x4 dst;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned sv = src[i];
                    dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute(to_mfma, (int)sv);
                }
                xa[k][mt] = __builtin_bit_cast(half8, dst);
            }
        // K tail: tanh(pt/5) of the row (3 values), lane group 0 only
        const RowId mid = tile_row(r, rgroup, sblk, S, R, b, ray0, nrays);
        half4 xt[MTN];
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[mt][e] = (_Float16)0.0f;
            if (g == 0 && mid.live) {
                const float* pe = pe6 + ((((size_t)(b * V + v)) * R + mid.r) * S + mid.s) * 6 + (j0 + mt) * 3;
                xt[mt][0] = (_Float16)pe[0]; xt[mt][1] = (_Float16)pe[1]; xt[mt][2] = (_Float16)pe[2];
                xt[mt][3] = (_Float16)1.0f;                   // x bias (hi)
            }
            if (g == 1 && mid.live) xt[mt][0] = (_Float16)1.0f;   // x bias (lo)
        }

        int vo[MTN][4];
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
            for (int k = 0; k < 4; ++k) vo[mt][k] = ((CPN_ENCODE_ABLATE & 16) ? 0 : rec[mt].off[k]) + pl * 16;
        // (image, mode) tables: border table of the own image for j = 0, zeros table of the other image for j = 1
        __amdgpu_buffer_rsrc_t trs[MTN];
#pragma unroll
        for (int mt = 0; mt < MTN; ++mt) {
            const bool own = (j0 + mt) == 0;
            const char* tb = own ? tbase + img_bytes * img_own
                                 : tbase + img_bytes * img_oth + (size_t)ng.border_nodes() * TAB_ROW_BYTES;
            trs[mt] = __builtin_amdgcn_make_buffer_rsrc(
                (void*)tb, 0, (int)((own ? ng.border_nodes() : ng.zeros_nodes()) * TAB_ROW_BYTES), 0x00020000);
        }
        // output rows of this lane (load layout): ((ray, view, sample), j = mt)
        // Stores: the 4 lanes of a quad hold 64 contiguous bytes of a row (pieces pl of each half), but a non-temporal
        // store stream only runs at the full rate when an instruction writes WHOLE 128-byte lines (tools/write_bw.py:
        // 6.2 TB/s against 3.6 TB/s for 64-byte pieces).  The even quad 2m (row c = 2m) and the odd quad 2m+1
        // (row c + 1 = the neighbouring ray of the same sample) therefore swap one half each with two DPP row shifts:
        // instruction A writes row 2m (even quad: its half 0, odd quad: half 1 of row 2m), instruction B row 2m+1.
        const int qodd = (lane >> 2) & 1;
        const RowId lidA = tile_row(rl & ~1, rgroup, sblk, S, R, b, ray0, nrays);
        const RowId lidB = tile_row(rl | 1, rgroup, sblk, S, R, b, ray0, nrays);
        auto out_row = [&](const RowId& id) {
            const size_t row = ((((size_t)b * R + id.r - ray0) * V + v) * S + id.s) * 2;       // + j (only used when live)
            return hid + ((CPN_ENCODE_ABLATE & 32) ? (row & 1023) : row) * 832 + (pl + 4 * qodd) * 8;  // 32: L2-resident window
        };
        __half* const hrowA = out_row(lidA);
        __half* const hrowB = out_row(lidB);
#if CPN_ENCODE_STORE == 7
        // The same whole-line nt stores as BUFFER stores through a per-tile descriptor, issued unconditionally: a dead row's
        // offset lies past the descriptor's range and the hardware drops the write.  Two things follow.  (1) No branch
        // around the store, so (2) the compiler, which does not see a store inside an `asm` (and must assume the fewest
        // outstanding operations at the join behind a conditional one), now COUNTS the two stores of a slice in its
        // s_waitcnt vmcnt bookkeeping: the last tap of slice n is waited for with vmcnt(2), not vmcnt(0), i.e. the stores
        // of slice n-1 stay in flight under the blend of slice n and are only retired by the (in-order) wait for the
        // taps of slice n+1, a whole slice later.  With the asm stores every wave drained its own 2 KB of stores once
        // per slice: 16 waves x 2 KB in flight per CU, 8 MB on the chip = 2.5 us of store latency at the 3.3 TB/s the
        // kernel reached - it was bound by store LATENCY, not by the write bandwidth (6.2 TB/s for this pattern).
        const long long tile_row0 = ((((long long)b * R + (long long)rgroup * TG - ray0) * V + v) * S + (long long)sblk * TSW) * 2;
        constexpr int kOOB = 0x7ffffff0;
        // (the tile's base is wave-uniform; the 64-bit index arithmetic above runs on the vector unit, so say so)
        const unsigned long long hb = (unsigned long long)(hid + tile_row0 * 832);
        const unsigned long long hbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)hb);
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)hbu, 0, (int)(((TG - 1) * V * S + TSW) * 2 * 1664), 0x00020000);
        auto out_off = [&](const RowId& id) {
            const int rel = (((id.r - rgroup * TG) * V * S + (id.s - sblk * TSW)) * 2) * 1664 + (pl + 4 * qodd) * 16;
            return id.live ? rel : kOOB;
        };
        const int hoffA = out_off(lidA), hoffB = out_off(lidB);
#endif

        // Slice loop, software-pipelined by one slice on the store side:
        //     issue the 16 tap loads of slice n  ->  issue the stores of slice n-1  ->  compute slice n.
        // gfx9 has ONE counter for loads and stores (vmcnt) and they retire out of order against each other, so waiting
        // for any load that was issued AFTER a store also waits for that store to reach memory.  With this order every
        // load a wave waits for is OLDER than the stores in flight: the 7 GB hid stream never stalls the wave that
        // issued it.
        half8 res[MTN][2];                                    // fp16 results of the previous slice, waiting to be stored
        auto store_slice = [&](int n) {
            if (!(CPN_ENCODE_ABLATE & 2) || res[0][0][0] == (_Float16)123.0f) {
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt) {
                    const u32x4 h0 = __builtin_bit_cast(u32x4, res[mt][0]), h1 = __builtin_bit_cast(u32x4, res[mt][1]);
                    u32x4 sa, sb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned a0 = h0[i], a1 = h1[i];
                        // even quads (banks 0, 2) take half 0 of lane + 4 into B; odd quads (banks 1, 3) half 1 of lane - 4 into A
                        sb[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a1, (int)a0, 0x104, 0xF, 0x5, false);
                        sa[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a0, (int)a1, 0x114, 0xF, 0xA, false);
                    }
                    const int co = (j0 + mt) * 832 + n * SLICE_CH;
#if CPN_ENCODE_STORE == 7
                    __builtin_amdgcn_raw_buffer_store_b128(sa, hrs, hoffA, co * 2, 2);          // aux 2 = nt
                    __builtin_amdgcn_raw_buffer_store_b128(sb, hrs, hoffB, co * 2, 2);
                    // two wait states before anything may write the stores' data registers: the compiler does not insert
                    // them behind a 16-byte buffer store with an SGPR offset, and gfx950 needs them (encode_key.hip)
                    asm volatile("s_nop 1" ::: "memory");
#else
                    if (lidA.live) store16(hrowA + co, __builtin_bit_cast(half8, sa));
                    if (lidB.live) store16(hrowB + co, __builtin_bit_cast(half8, sb));
#endif
                }
            }
        };
        typedef u32x4 TapData[MTN][4][2];
        auto issue_taps = [&](int n, TapData& td) {
            if (!(CPN_ENCODE_ABLATE & 1)) {
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const __amdgpu_buffer_rsrc_t rs = trs[mt];
                        td[mt][k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[mt][k], n * TAB_SLICE_BYTES, 0);
                        td[mt][k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[mt][k] + 64, n * TAB_SLICE_BYTES, 0);
                    }
            }
        };
        auto compute_slice = [&](int n, TapData& td) {

            // ---- K = 80 contraction of the full-resolution level + point encoding, on top of the bias; weights from LDS
            f32x4 acc[MTN][NT];
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (!(CPN_ENCODE_ABLATE & 4)) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const half8 wf = wmain[((n * 2 + k) * NT + nt) * 64 + lane];
#pragma unroll
                        for (int mt = 0; mt < MTN; ++mt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xa[k][mt], acc[mt][nt], 0, 0, 0);
                    }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const half4 wq = wtail_s[(n * NT + nt) * 48 + tail_lane];
#pragma unroll
                    for (int mt = 0; mt < MTN; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wq, xt[mt], acc[mt][nt], 0, 0, 0);
                }
            }
            // MFMA layout -> load layout
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = acc[mt][nt][i];      // (a bit_cast applied to the vector-element lvalue itself
                        acc[mt][nt][i] = __int_as_float(     //  reads element 0 for every i with this compiler)
                            __builtin_amdgcn_ds_bpermute(to_ll, __float_as_int(t)));
                    }
            // ---- 4 table taps per row in fp32 on top of it, then ReLU and the fp16 rounding
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) {
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
                // fp16 rounding two at a time (v_cvt_pk_f16_f32), ReLU on the packed pair (v_pk_max_f16): rounding is
                // monotonic and keeps the sign, so this equals max(x, 0) followed by the rounding
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 pk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4& src = acc[mt][2 * h + (q >> 1)];
                        const f32x2v two = {src[2 * (q & 1)], src[2 * (q & 1) + 1]};
                        half2v hv = __builtin_convertvector(two, half2v);
                        hv = __builtin_elementwise_max(hv, (half2v){(_Float16)0.0f, (_Float16)0.0f});
                        pk[q] = __builtin_bit_cast(unsigned, hv);
                    }
                    res[mt][h] = __builtin_bit_cast(half8, pk);
                }
            }
        };
#if CPN_ENCODE_PREFETCH
        // taps of slice n+1 in flight while slice n is computed (two register sets, 32 more VGPRs): the table reads that
        // miss L2 take ~500 clocks from a warm Infinity Cache but ~1550 from HBM behind the kernel's own 7 GB write
        // stream once other kernels have replaced the tables there (profiles/r03_pmc_encode_hidden_hot_vs_flushed.json)
        TapData ta, tb;
        issue_taps(0, ta);
        for (int n = 0; n < NSLICE; n += 2) {
            if (n + 1 < NSLICE) issue_taps(n + 1, tb);
            __builtin_amdgcn_sched_barrier(0);
            if (n > 0) store_slice(n - 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_slice(n, ta);
            if (n + 1 < NSLICE) {
                if (n + 2 < NSLICE) issue_taps(n + 2, ta);
                __builtin_amdgcn_sched_barrier(0);
                store_slice(n);
                __builtin_amdgcn_sched_barrier(0);
                compute_slice(n + 1, tb);
            }
        }
        store_slice(NSLICE - 1);
#else
        for (int n = 0; n < NSLICE; ++n) {
            TapData td;
            issue_taps(n, td);
            __builtin_amdgcn_sched_barrier(0);
            if (n > 0) store_slice(n - 1);
            __builtin_amdgcn_sched_barrier(0);
            compute_slice(n, td);
        }
        store_slice(NSLICE - 1);
#endif
    }
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
    // ray groups: TG consecutive rays of ONE batch element (r aligned to TG), so a wave tile's images are uniform
    const int groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    const long long group0 = (long long)b_lo * groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    const int nsblk = (int)cpn_cdiv(S, TSW);
    const long long nwtiles = (group1 - group0 + 1) * V * nsblk * (2 / MTN);
    const int num_cu = cpn_stream_cus((void*)stream);       // persistent grid: the CUs this stream may use
    const unsigned grid = (unsigned)std::min<long long>(num_cu, cpn_cdiv(nwtiles, ENC_WAVES));      // persistent: one workgroup per CU
    hipLaunchKernelGGL(encode_hidden_kernel, dim3(grid), dim3(64 * ENC_WAVES), 0, (hipStream_t)stream,
                       (const __half*)tab, (const __half*)map3, H, W, pixel_val, sec_grid, pe6, (const half8*)wfrag,
                       bias, V, R, S, ray0, nrays, nsblk, groups_per_b, group0, nwtiles, (__half*)hid);
    CPN_LAUNCH_CHECK("cpn_encode_hidden");
    return 0;
}
