// ROUND-4 FORM, kept as the timing reference of tools/ef_check.py (exports renamed *_r4; not part of the product library).
// K2+K3a+K3c — the first encoder layer on the node tables (encode.hip) WITH the folded key_map contraction behind it:
//     hid[row]  = ReLU(query_encode_latent([gather ‖ tanh(pt/5)]))                       (written once, for the two hidden sums)
//     kh[sample] = ReLU( (Wk_a W2 | Wk_b W2) . [hid_own ; hid_other] + c' )                (128 wide, fp16)
// i.e. /root/reference models/CoPoNeRF.py:312, 370, 384-397 and :404-407 (key_map after query_encode_latent_2, folded as
// DESIGN.md §4.3 describes).  Round 3 ran the second line as its own kernel (cpn_gemm_f16_chain_rowdot) that read the
// 27.9 GB of hid back from HBM: one of the four passes over hid.  Here the 64-channel slice of hid a wave has just
// produced — still in registers — becomes the K-panel of the 1664 -> 128 MFMA contraction; `kh` (256 B per sample
// instead of 3 328) goes to cpn_gemm_f16_rowdot for key_map_2 and the logit.
//
// What the fusion has to solve is the WEIGHT traffic: the folded key matrix is 128 x 1664 fp16 = 416 KiB, every 16-row
// wave tile needs the 16 KiB that belong to its (image, slice) and LDS is full (the K = 80 fragments of the first layer
// take 123.5 KiB).  So, unlike encode.hip, the waves of a workgroup run the slice loop IN LOCK STEP: a two-slot ring of
// 16 KiB in LDS holds the key weights of the slice in flight; one s_barrier per slice (right behind the issue of the
// slice's tap loads) says "everybody has finished with slot (q-1) & 1 and everybody's share of slot q & 1 has landed",
// after which each wave DMAs its 1 KiB pieces of slice q + 1 (buffer_load ... lds straight from the row-major weight matrix: an MFMA A fragment is 16 contiguous bytes per
// lane) and runs the 16 key MFMAs of slice q against its own 16 x 64 piece of hid.  L2 -> LDS traffic: 16 KiB per
// workgroup and slice against 96-128 KiB of table taps.  A wave owns BOTH images of its 4 rays x 4 samples (own image
// first, then the other): the key accumulators (8 tiles x 4 = 32 VGPRs) run over K = 2 x 832.
//   * 12 waves (3 per SIMD, <= 168 VGPRs): the accumulators do not fit the 128 of encode.hip's 16 waves.
//   * hid is moved from the load layout to the MFMA B-operand layout with 8 ds_bpermute per slice (fp16 pairs).
//   * every wave of a workgroup executes the same number of slice steps (dead units at the end of a range do dummy
//     work): the barrier count must match.
// Everything else (tap records, level-3 gather, K = 80 MFMA, 4-tap blend, whole-line nt stores) is encode.hip's.
#include <algorithm>

// timing-only ablations (results are wrong when non-zero): 1 = no table taps, 2 = no hid stores, 4 = no K = 80 MFMA,
// 8 = no key MFMA (no LDS reads of the ring either), 64 = no ring traffic (no DMA, no barrier: weights are garbage),
// 16 / 32 = every key / K = 80 MFMA of a slice on the SAME LDS fragment (the MFMAs stay, 16 -> 1 / 12 -> 2 LDS reads),
// 128 = no kh stores
#ifndef CPN_EK_ABLATE
#define CPN_EK_ABLATE 0
#endif
#ifndef CPN_EK_WAVES
#define CPN_EK_WAVES 12
#endif

#include "encode_common.h"

namespace {

constexpr int WMAIN_HALF8 = NSLICE * 2 * NT * 64;              // [slice][k < 2][tile][lane] half8: 104 KiB
constexpr int WTAIL_HALF4 = NSLICE * NT * 48;                  // [slice][tile][K group < 3][A-operand row] half4: 19.5 KiB
constexpr int EK_WAVES_DEFAULT = CPN_EK_WAVES;
constexpr int EK_WAVES_BESIDE = 8;                             // the form that leaves room on its CU (cpn_encode_key_beside)
constexpr int KT = 8;                                          // 16-wide output tiles of the key layer (128)
constexpr int KPIECES = KT * 2;                                // 1 KiB fragments (tile, k step) per slice
constexpr int KSLOT_HALF8 = KPIECES * 64;                      // one ring slot: 16 KiB
constexpr int KSTEPS = 2 * NSLICE;                             // slice steps per unit: both images
constexpr int KLD = 2 * CPN_TAB_LD;                            // row length of the folded key matrix (1664)

// GROUP = 0 (the product): the K = 80 fragments of the first layer stay resident in LDS (123.5 KiB) and the key weights go
// through a two-slot ring of one slice each: ONE BARRIER PER SLICE.  GROUP = G > 0 (measured, slower): nothing is resident;
// a slice's K = 80 block (8 KiB + its tail and bias, from the pre-packed `k80blk`) travels with its key weights (26 KiB per
// slice), the ring has two halves of G slices each and the waves meet once per G slices - with G = 3 (156 KiB) they may
// drift up to three slices apart in between.  That was meant to give back the overlap between waves that lock step costs,
// and the barrier count did fall 3x, but 12.4 ms per image (G = 3) / 13.6 (G = 1) against 11.7 resident say the cost sits in
// the L2 -> LDS weight stream itself, 26 KiB instead of 16 per workgroup and slice, not in the barriers.
constexpr int STEP_BYTES = 26 * 1024;                          // streamed form: 16 KiB key + 8 KiB K=80 main + 2 KiB tail (1.5 used)
constexpr int K80_BLOCK_BYTES = 10 * 1024;                     // one slice of k80blk: main fragments, tail fragments with the bias, pad

// CPN_EK_REGS_FOR > 0: register budget of that many waves per SIMD even where the workgroup has fewer (8 waves = 2 per SIMD
// with the 168-register budget of 3 leave a third of each SIMD's register file to the waves of ANOTHER kernel)
#ifndef CPN_EK_REGS_FOR
#define CPN_EK_REGS_FOR 0
#endif
#if CPN_EK_REGS_FOR > 0
#define EK_OCC __attribute__((amdgpu_waves_per_eu(CPN_EK_REGS_FOR, CPN_EK_REGS_FOR)))
#else
#define EK_OCC
#endif

// KPACKED: the key matrix comes pre-packed in ring-piece order ([slice step][piece = tile * 2 + k][lane][8 halves], group 4 of
// cpn_encode_key): a DMA piece is 1 KiB of contiguous memory.  From the row-major matrix a piece is 16 rows x 64 bytes - 64 L1
// tag look-ups per instruction, and the 16 pieces per slice added two thirds to the look-ups of the 96 tap loads of a workgroup
// in a kernel that sits on that pipe.
template <int GROUP, int EK_WAVES, bool KPACKED = false>
__global__ __launch_bounds__(64 * EK_WAVES, 1) EK_OCC void encode_key_kernel(
    const __half* __restrict__ tab, const __half* __restrict__ map3, int H, int W,
    const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, const float* __restrict__ pe6,
    const half8* __restrict__ wfrag, const float* __restrict__ bias, const __half* __restrict__ k80blk,
    const __half* __restrict__ kw, const float* __restrict__ kbias, int V, int R, int S, int ray0, int nrays, int nsblk,
    int groups_per_b, long long group0, long long nunits, __half* __restrict__ hid, __half* __restrict__ kh) {
    constexpr bool STREAM = GROUP > 0;
    __shared__ __attribute__((aligned(16))) half8 wmain[STREAM ? 1 : WMAIN_HALF8];
    __shared__ __attribute__((aligned(16))) half4 wtail_s[STREAM ? 1 : WTAIL_HALF4];
    __shared__ __attribute__((aligned(16))) half8 kring[STREAM ? 2 * GROUP * (STEP_BYTES / 16) : 2 * KSLOT_HALF8];
    __shared__ __attribute__((aligned(16))) half8 dump[STREAM ? 64 : 1];         // target of the refill slots that carry no piece

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (!STREAM) {
    for (int i = tid; i < WMAIN_HALF8; i += 64 * EK_WAVES) wmain[i] = wfrag[i];
    {
        const half4* tsrc = reinterpret_cast<const half4*>(wfrag + WMAIN_HALF8);          // K tail + bias: see encode.hip
        for (int i = tid; i < WTAIL_HALF4; i += 64 * EK_WAVES) {
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
    }

    const int r = lane & 15, g = lane >> 4;                   // MFMA layout: column (row of the tile) r, K / channel group g
    const int rl = lane >> 2, pl = lane & 3;                  // load layout: row rl, 16-byte piece pl
    const int tail_lane = min(g, 2) * 16 + r;
    const int to_ll = (rl + 16 * pl) * 4;                     // ds_bpermute address: this lane takes MFMA lane (r = rl, g = pl)
    const int to_mfma = (4 * r + g) * 4;                      //                      this lane takes load-layout lane (rl = r, pl = g)
    const NodeGrid ng{W >> 1, H >> 1};
    const size_t img_bytes = (size_t)ng.nodes_per_image() * TAB_ROW_BYTES;
    const char* const tbase = reinterpret_cast<const char*>(tab);
    const char* const m3base = reinterpret_cast<const char*>(map3);

    // ---- the key-weight ring.  Fragment (t, k) of slice step q = (image j, slice n): lane (a = lane & 15, g) holds
    //      Wk[t*16 + a][j*832 + n*64 + k*32 + g*8 .. +8] = 16 contiguous bytes of the row-major matrix.
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kw, 0, 128 * KLD * 2, 0x00020000);
    const int kvoff = (r * KLD + g * 8) * 2;
    auto ring_fill = [&](int step_in_unit, int slot) {
        if (CPN_EK_ABLATE & 64) return;
        const int j = step_in_unit >= NSLICE ? 1 : 0, n = step_in_unit - j * NSLICE;
        for (int p = wave; p < KPIECES; p += EK_WAVES) {                       // wave-uniform trip count
            const int t = p >> 1, k = p & 1;
            if constexpr (KPACKED)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    krs, (lds_void*)(reinterpret_cast<char*>(kring) + slot * (KSLOT_HALF8 * 16) + p * 1024), 16, lane * 16,
                    (step_in_unit * KPIECES + p) * 1024, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    krs, (lds_void*)(reinterpret_cast<char*>(kring) + slot * (KSLOT_HALF8 * 16) + p * 1024), 16, kvoff,
                    ((t * 16) * KLD + j * CPN_TAB_LD + n * SLICE_CH + k * 32) * 2, 0, 0);
        }
    };
    // streamed form: group gi = slice steps [gi*GROUP, gi*GROUP + GROUP) of this workgroup, into ring half gi & 1
    const __amdgpu_buffer_rsrc_t k80rs = __builtin_amdgcn_make_buffer_rsrc((void*)k80blk, 0, NSLICE * K80_BLOCK_BYTES, 0x00020000);
    auto group_fill = [&](int gi, int total_steps) {
        if (CPN_EK_ABLATE & 64) return;
        constexpr int PPS = STEP_BYTES / 1024;                                 // 1 KiB pieces per slice step
        for (int pp = wave; pp < PPS * (STREAM ? GROUP : 1); pp += EK_WAVES) {     // wave-uniform trip count
            const int si = pp / PPS, p = pp - si * PPS;
            const int st = gi * GROUP + si;
            if (st >= total_steps) break;
            const int qq = st % KSTEPS, j = qq >= NSLICE ? 1 : 0, n = qq - j * NSLICE;
            lds_void* dst = (lds_void*)(reinterpret_cast<char*>(kring) + ((gi & 1) * GROUP + si) * STEP_BYTES + p * 1024);
            if (p < KPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, dst, 16, kvoff,
                                                         (((p >> 1) * 16) * KLD + j * CPN_TAB_LD + n * SLICE_CH + (p & 1) * 32) * 2, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(k80rs, dst, 16, lane * 16, n * K80_BLOCK_BYTES + (p - KPIECES) * 1024, 0, 0);
        }
    };
    if constexpr (!STREAM) ring_fill(0, 0);

    // XCD-aware order as in encode.hip, in UNITS = (4 rays, view, 4 samples) x both images; the waves of a workgroup take
    // consecutive units and every wave runs the same number of iterations
    const unsigned nbk = gridDim.x, nx = nbk < 8 ? nbk : 8;
    const unsigned xcd = blockIdx.x % nx, wgx = blockIdx.x / nx;
    const unsigned wg_on_xcd = nbk / nx + (xcd < nbk % nx ? 1 : 0);
    const long long q = nunits / nx, rem = nunits % nx;
    const long long x_begin = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const long long x_end = x_begin + q + (xcd < rem ? 1 : 0);
    const long long per_iter = (long long)wg_on_xcd * EK_WAVES;
    const int iters = (int)((x_end - x_begin + per_iter - 1) / per_iter);
    int gstep = 0;                                            // slice steps this workgroup has started (ring slot = parity)
    const int total_steps = iters * KSTEPS;
    if constexpr (STREAM) group_fill(0, total_steps);
    __syncthreads();          // resident form: K = 80 fragments in place; streamed form: group 0 has landed (vmcnt(0) + barrier)
    // Lock step costs the overlap BETWEEN waves: with one barrier site every wave is in the same phase of a slice at the
    // same time (memory wait, then 3 waves' blends, then 3 waves' MFMAs on each SIMD, one after the other: 12.1 ms per
    // image against 9.5 without the barrier).  The s_barrier only counts arrivals, so the waves may take it at DIFFERENT
    // points of their slice: "early" waves right behind the issue of their tap loads, the others in front of their key
    // MFMAs.  Both sites keep what the ring needs - a wave arrives at barrier q only after its key MFMAs of step q - 1
    // and after its own pieces of slot q have landed, and it reads slot q / refills slot q + 1 only behind barrier q -
    // but the two groups now run about half a slice apart: one blends while the other multiplies.
    // CPN_EK_PHASES: 1 = every wave at the late site, 2 = waves 4-7 (one of the three waves of each SIMD) early.
#ifndef CPN_EK_PHASES
#define CPN_EK_PHASES 2
#endif
    // CPN_EK_PHASES 4: a third site behind the K = 80 MFMAs for waves 8-11, so that the three waves of a SIMD are in three
    // different phases (measured: see DESIGN.md 4.1b)
    const bool early_sync = (CPN_EK_PHASES == 2 || CPN_EK_PHASES == 4) ? ((wave >> 2) == 1) : (CPN_EK_PHASES == 3);
    const bool mid_sync = CPN_EK_PHASES == 4 && (wave >> 2) == 2;

    for (int it = 0; it < iters; ++it) {
        const long long uu_raw = x_begin + (long long)it * per_iter + (long long)wgx * EK_WAVES + wave;
        const bool unit_live = uu_raw < x_end;
        const long long uu = unit_live ? uu_raw : x_begin;    // a dead unit walks a live unit's addresses with every row masked
        const int sblk = (int)(uu % nsblk);
        const int v = (int)((uu / nsblk) % V);
        const long long gq = group0 + uu / ((long long)nsblk * V);
        const int b = (int)(gq / groups_per_b), rgroup = (int)(gq % groups_per_b);
        const int img_own = b * V + v, img_oth = b * V + (V - 1 - v);

        RowId lid = tile_row(rl, rgroup, sblk, S, R, b, ray0, nrays);
        RowId mid = tile_row(r, rgroup, sblk, S, R, b, ray0, nrays);
        lid.live = lid.live && unit_live;
        mid.live = mid.live && unit_live;
        const size_t sidx_l = (((size_t)(b * V + v)) * R + min(lid.r, R - 1)) * S + min(lid.s, S - 1);
        const int qodd = (lane >> 2) & 1;
        RowId lidA = tile_row(rl & ~1, rgroup, sblk, S, R, b, ray0, nrays);
        RowId lidB = tile_row(rl | 1, rgroup, sblk, S, R, b, ray0, nrays);
        lidA.live = lidA.live && unit_live;
        lidB.live = lidB.live && unit_live;
        // hid stores: unconditional nt buffer stores through a per-tile descriptor (encode.hip, CPN_ENCODE_STORE 7)
        const long long tile_row0 = ((((long long)b * R + (long long)rgroup * TG - ray0) * V + v) * S + (long long)sblk * TSW) * 2;
        constexpr int kOOB = 0x7ffffff0;
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

        f32x4 kacc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) kacc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll 1
        for (int j0 = 0; j0 < 2; ++j0) {
            // ---- per-row records of image j0 in the LOAD layout (j0 = 0: own image, border table, pixel_val;
            //      j0 = 1: other image, zeros table, sec_grid)
            const bool own = j0 == 0;
            TapRec rec;
            half8 xl[2];
            {
                const float2 gc = *reinterpret_cast<const float2*>((own ? pixel_val : sec_grid) + sidx_l * 2);
                rec = node_taps(gc.x, gc.y, ng, own);
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
                    for (int e = 0; e < 8; ++e) xl[k][e] = lid.live ? (_Float16)a8[e] : (_Float16)0.0f;
                }
                if (!lid.live) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { rec.off[t] = 0; rec.w[t] = 0.0f; }
                }
            }
            half8 xa[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const u32x4 src = __builtin_bit_cast(u32x4, xl[k]);
                u32x4 dst;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned sv = src[i];
                    dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute(to_mfma, (int)sv);
                }
                xa[k] = __builtin_bit_cast(half8, dst);
            }
            half4 xt;
#pragma unroll
            for (int e = 0; e < 4; ++e) xt[e] = (_Float16)0.0f;
            if (g == 0 && mid.live) {
                const float* pe = pe6 + ((((size_t)(b * V + v)) * R + mid.r) * S + mid.s) * 6 + j0 * 3;
                xt[0] = (_Float16)pe[0]; xt[1] = (_Float16)pe[1]; xt[2] = (_Float16)pe[2];
                xt[3] = (_Float16)1.0f;                       // x bias (hi)
            }
            if (g == 1 && mid.live) xt[0] = (_Float16)1.0f;   // x bias (lo)

            int vo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) vo[k] = rec.off[k] + pl * 16;
            const char* tb = own ? tbase + img_bytes * img_own
                                 : tbase + img_bytes * img_oth + (size_t)ng.border_nodes() * TAB_ROW_BYTES;
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)tb, 0, (int)((own ? ng.border_nodes() : ng.zeros_nodes()) * TAB_ROW_BYTES), 0x00020000);

            half8 res[2];                                     // fp16 results of the previous slice, waiting to be stored
            // `drop`: the call before the first slice stores nothing (offsets out of range) — but it IS two store
            // instructions, so every trip of the slice loop issues the same operations and the compiler's vmcnt counts stay exact
            auto store_slice = [&](int n, bool drop) {
                if (CPN_EK_ABLATE & 2) return;
                const u32x4 h0 = __builtin_bit_cast(u32x4, res[0]), h1 = __builtin_bit_cast(u32x4, res[1]);
                u32x4 sa, sb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a0 = h0[i], a1 = h1[i];
                    sb[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a1, (int)a0, 0x104, 0xF, 0x5, false);
                    sa[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a0, (int)a1, 0x114, 0xF, 0xA, false);
                }
                const int co = __builtin_amdgcn_readfirstlane((j0 * 832 + n * SLICE_CH) * 2);       // scalar offset operand
                // gfx950 hazard the compiler does not cover: a VALU write to a data VGPR of a 16-byte buffer store in the
                // one or two issue slots behind it reaches memory in some lanes (LLVM assumes the "store > 8 bytes, then VALU
                // write of the data" hazard away when the store has an SGPR offset; with the select of the second store's
                // offset scheduled between the two stores, dword 0 of lanes 12-15 of every row of 16 held that offset).
                // So: both offsets exist before the first store, and two wait states follow the second one.
                int oa = drop ? kOOB : hoffA, ob = drop ? kOOB : hoffB;
                asm volatile("" : "+v"(oa), "+v"(ob));
                __builtin_amdgcn_raw_buffer_store_b128(sa, hrs, oa, co, 2);          // aux 2 = nt
                __builtin_amdgcn_raw_buffer_store_b128(sb, hrs, ob, co, 2);
                asm volatile("s_nop 1" ::: "memory");
            };
            res[0] = res[1] = half8{0, 0, 0, 0, 0, 0, 0, 0};

#pragma unroll 1
            for (int n = 0; n < NSLICE; ++n) {
                if constexpr (STREAM) {
                    // once per GROUP slices: everybody has finished the previous group (its ring half may be refilled with the
                    // group after this one) and everybody's pieces of THIS group, issued a whole group ago, have landed
                    // (in-order vmcnt; only the two stores of the previous slice may still be in flight)
                    if (gstep % GROUP == 0 && !(CPN_EK_ABLATE & 64)) {
                        if (CPN_EK_ABLATE & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const char* const sbase = reinterpret_cast<const char*>(kring) +
                                          (STREAM ? (((gstep / (STREAM ? GROUP : 1)) & 1) * GROUP + gstep % (STREAM ? GROUP : 1)) * STEP_BYTES : 0);
                const half8* const wm = STREAM ? reinterpret_cast<const half8*>(sbase + KPIECES * 1024) : wmain + n * 2 * NT * 64;
                const half4* const wt = STREAM ? reinterpret_cast<const half4*>(sbase + KPIECES * 1024 + 8192) : wtail_s + n * NT * 48;
                u32x4 td[4][2];
                if (!(CPN_EK_ABLATE & 1)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        td[k][0] = __builtin_amdgcn_raw_buffer_load_b128(trs, vo[k], n * TAB_SLICE_BYTES, 0);
                        td[k][1] = __builtin_amdgcn_raw_buffer_load_b128(trs, vo[k] + 64, n * TAB_SLICE_BYTES, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STREAM) {
                    // The refill of the other ring half (the NEXT group's 26 x GROUP pieces) is spread over this group's steps:
                    // every wave issues exactly FPW pieces per step, BEHIND its tap loads - a fixed number of operations,
                    // so the compiler's vmcnt counts for the taps stay exact and leave the pieces (and the stores behind
                    // them) in flight; issued in front of the taps, as one burst per group, every wave's next tap wait also
                    // waited for its pieces (12.9 ms per image against 12.1 for the resident form).  Slots past the group's
                    // last piece read out of range (no memory traffic) into a 1 KiB dump.
                    constexpr int PPS = STEP_BYTES / 1024, FPW = (PPS + EK_WAVES - 1) / EK_WAVES;
                    const int gi = gstep / GROUP + 1, sg = gstep % GROUP;
                    if (!(CPN_EK_ABLATE & 64)) {
#pragma unroll
                        for (int f = 0; f < FPW; ++f) {
                            const int pp = (sg * FPW + f) * EK_WAVES + wave;
                            const int si = pp / PPS, p = pp - si * PPS;
                            const int st = gi * GROUP + si;
                            const bool real = pp < PPS * GROUP && st < total_steps;
                            const int qq = st % KSTEPS, j = qq >= NSLICE ? 1 : 0, nn = qq - j * NSLICE;
                            lds_void* dst = (lds_void*)(real ? reinterpret_cast<char*>(kring) + ((gi & 1) * GROUP + si) * STEP_BYTES + p * 1024
                                                             : reinterpret_cast<char*>(dump));
                            const bool key = p < KPIECES;
                            const int so = !real ? 0x7ffffff0
                                                 : key ? (((p >> 1) * 16) * KLD + j * CPN_TAB_LD + nn * SLICE_CH + (p & 1) * 32) * 2
                                                       : nn * K80_BLOCK_BYTES + (p - KPIECES) * 1024;
                            // one instruction either way (the descriptor and the per-lane offset are selected, not branched on)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(key || !real ? krs : k80rs, dst, 16, key || !real ? kvoff : lane * 16,
                                                                     so, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- ring.  The barrier sits HERE, behind the issue of the step's 8 tap loads, where every wave is about to
                //      wait for memory anyway (in front of the key MFMAs it cost 2.6 of 12 ms per image): it says (a) every
                //      wave has finished the key MFMAs of the previous step, so slot (gstep + 1) & 1 may be refilled, and
                //      (b) every wave's pieces of THIS step's slot have landed - each wave issued them one step ago and
                //      makes sure of its own with the counted wait (in-order vmcnt: all but the 2 stores and 8 taps issued
                //      since).  The stores of the previous slice follow the refill, so that they stay the youngest
                //      operations in flight and no wait of this step has to cover them.
                const int step_in_unit = j0 * NSLICE + n;
                auto ring_sync = [&](bool counted_wait) {
                    if (CPN_EK_ABLATE & 64) return;
                    if (counted_wait) {
                        if (CPN_EK_ABLATE & (1 | 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    const bool last = (it == iters - 1) && (step_in_unit == KSTEPS - 1);
                    if (!last) ring_fill(step_in_unit == KSTEPS - 1 ? 0 : step_in_unit + 1, (gstep + 1) & 1);
                };
                if (!STREAM && early_sync) ring_sync(true);
                __builtin_amdgcn_sched_barrier(0);
                store_slice(n > 0 ? n - 1 : 0, n == 0);
                __builtin_amdgcn_sched_barrier(0);

                // ---- K = 80 contraction of the full-resolution level + point encoding + bias; weights from LDS
                f32x4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (!(CPN_EK_ABLATE & 4)) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[((CPN_EK_ABLATE & 32) ? 0 : (k * NT + nt) * 64) + lane], xa[k], acc[nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(wt[((CPN_EK_ABLATE & 32) ? 0 : nt * 48) + tail_lane], xt, acc[nt], 0, 0, 0);
                }
                if (!STREAM && mid_sync) {                    // (taps and stores of this step are the 10 operations issued since)
                    __builtin_amdgcn_sched_barrier(0);
                    ring_sync(true);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = acc[nt][i];
                        acc[nt][i] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_ll, __float_as_int(t)));
                    }
                // ---- 4 table taps per row in fp32 on top of it, ReLU, fp16
                if (!(CPN_EK_ABLATE & 1)) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float wk = rec.w[k];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const u32x4 d = td[k][h];
                            f32x4* a2 = &acc[2 * h];
                            a2[0][0] = fma_mix_lo(a2[0][0], d[0], wk); a2[0][1] = fma_mix_hi(a2[0][1], d[0], wk);
                            a2[0][2] = fma_mix_lo(a2[0][2], d[1], wk); a2[0][3] = fma_mix_hi(a2[0][3], d[1], wk);
                            a2[1][0] = fma_mix_lo(a2[1][0], d[2], wk); a2[1][1] = fma_mix_hi(a2[1][1], d[2], wk);
                            a2[1][2] = fma_mix_lo(a2[1][2], d[3], wk); a2[1][3] = fma_mix_hi(a2[1][3], d[3], wk);
                        }
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4 pk;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x4& src = acc[2 * h + (qq >> 1)];
                        const f32x2v two = {src[2 * (qq & 1)], src[2 * (qq & 1) + 1]};
                        half2v hv = __builtin_convertvector(two, half2v);
                        hv = __builtin_elementwise_max(hv, (half2v){(_Float16)0.0f, (_Float16)0.0f});
                        pk[qq] = __builtin_bit_cast(unsigned, hv);
                    }
                    res[h] = __builtin_bit_cast(half8, pk);
                }
                // ---- this slice of hid as the B operand of the key layer: K = 8 consecutive channels per lane and k step
                half8 xb[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 src = __builtin_bit_cast(u32x4, res[k]);
                    u32x4 dst;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned sv = src[i];
                        dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute(to_mfma, (int)sv);
                    }
                    xb[k] = __builtin_bit_cast(half8, dst);
                }
                // (second site of the ring barrier: the waves that did not take it behind the tap issue.  In-order vmcnt:
                // their pieces of this step's slot, issued one step ago before this step's taps, have landed.)
                if (!STREAM && !early_sync && !mid_sync) ring_sync(false);
                __builtin_amdgcn_sched_barrier(0);
                if (!(CPN_EK_ABLATE & 8)) {
                    const half8* slot = STREAM ? reinterpret_cast<const half8*>(sbase) : kring + (gstep & 1) * KSLOT_HALF8;
#pragma unroll
                    for (int t = 0; t < KT; ++t)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            kacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(slot[((CPN_EK_ABLATE & 16) ? 0 : (t * 2 + k) * 64) + lane], xb[k], kacc[t], 0, 0, 0);
                }
                ++gstep;
            }
            store_slice(NSLICE - 1, false);
        }

        // ---- kh = fp16(ReLU(acc + c')): lane (r, g) holds outputs t*16 + g*4 .. +4 of row r.  Stored as they are (8 stores of
        //      8 bytes per lane, 16 rows x 4 pieces per instruction) the 2.1 GB of kh cost 0.38 of the kernel's 11.5 ms (ablation
        //      128).  v_permlane16_swap (gfx950) trades the odd 16-lane rows of tile t with the even rows of tile t + 1: lane
        //      (r, g) then holds 8 consecutive outputs of tile t + (g & 1) - 4 stores of 16 bytes, 64 contiguous bytes per row.
        {
            unsigned hw[KT][2];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(kbias + t * 16 + g * 4);
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)fmaxf(kacc[t][i] + bv[i], 0.0f);
                const u32x2 w2 = __builtin_bit_cast(u32x2, o);
                hw[t][0] = w2[0];
                hw[t][1] = w2[1];
            }
            const size_t srow = (((size_t)b * R + mid.r - ray0) * V + v) * S + mid.s;
            __half* dst = kh + srow * 128 + (g & 1) * 16 + (g >> 1) * 8;
            // (ablation 128: a condition that never holds at run time keeps the accumulators - and the key MFMAs - alive)
            const bool store = mid.live && (!(CPN_EK_ABLATE & 128) || kacc[0][0] == 12345.678f);
#pragma unroll
            for (int q = 0; q < KT / 2; ++q) {
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(hw[2 * q][0], hw[2 * q + 1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(hw[2 * q][1], hw[2 * q + 1][1], false, false);
                const u32x4 piece = {lo[0], hi[0], lo[1], hi[1]};
                if (store) *reinterpret_cast<u32x4*>(dst + q * 32) = piece;
            }
        }
    }
}

}  // namespace

static int encode_key_launch(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                             const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                             const uint16_t* k80blk, int group, int waves, const uint16_t* kw, const float* kbias, int B, int V,
                             int R, int S, int ray0, int nrays, uint16_t* hid, uint16_t* kh, void* stream, const char* who) {
    CPN_REQUIRE(tab && map3 && pixel_val && sec_grid && pe6 && wfrag && bias && kw && kbias && hid && kh, CPN_E_ARG,
                "%s: null pointer", who);
    CPN_REQUIRE((group == 0 || group == 1 || group == 3 || group == 4) &&
                    (group == 0 || group == 4 || (k80blk && ((uintptr_t)k80blk % 16) == 0)), CPN_E_ARG,
                "%s: group must be 0, 1, 3 or 4 (got %d) and needs k80blk for 1 and 3", who, group);
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0,
                CPN_E_SHAPE, "%s: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", who, H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "%s: ray range [%d,%d) outside B*R=%lld", who, ray0, ray0 + nrays, (long long)B * R);
    const long long nrows = (long long)nrays * V * S * 2;
    const NodeGrid ng{W >> 1, H >> 1};
    CPN_REQUIRE(nrows < (1LL << 31) && ng.zeros_nodes() * TAB_ROW_BYTES < (1LL << 31) && (long long)H * W * 128 < (1LL << 31) &&
                    (long long)TG * V * S * 2 * 1664 < (1LL << 31),
                CPN_E_SHAPE, "%s: chunk / per-image table too large for 32-bit offsets (%lld rows)", who, nrows);
    CPN_REQUIRE(((uintptr_t)tab % 16) == 0 && ((uintptr_t)map3 % 16) == 0 && ((uintptr_t)wfrag % 16) == 0 &&
                    ((uintptr_t)bias % 16) == 0 && ((uintptr_t)hid % 16) == 0 && ((uintptr_t)kw % 16) == 0 &&
                    ((uintptr_t)kbias % 16) == 0 && ((uintptr_t)kh % 8) == 0, CPN_E_ARG,
                "%s: pointers must be 16-byte aligned", who);
    const int groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    const long long group0 = (long long)b_lo * groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    const int nsblk = (int)cpn_cdiv(S, TSW);
    const long long nunits = (group1 - group0 + 1) * V * nsblk;
    const int num_cu = cpn_stream_cus((void*)stream);
    const unsigned grid = (unsigned)std::min<long long>(num_cu, cpn_cdiv(nunits, waves));
    auto kern = waves == EK_WAVES_BESIDE ? encode_key_kernel<0, EK_WAVES_BESIDE>
                : group == 4             ? encode_key_kernel<0, EK_WAVES_DEFAULT, true>
                : group == 0             ? encode_key_kernel<0, EK_WAVES_DEFAULT>
                : group == 1             ? encode_key_kernel<1, EK_WAVES_DEFAULT>
                                         : encode_key_kernel<3, EK_WAVES_DEFAULT>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), 0, (hipStream_t)stream,
                       (const __half*)tab, (const __half*)map3, H, W, pixel_val, sec_grid, pe6, (const half8*)wfrag,
                       bias, (const __half*)k80blk, (const __half*)kw, kbias, V, R, S, ray0, nrays, nsblk, groups_per_b,
                       group0, nunits, (__half*)hid, (__half*)kh);
    CPN_LAUNCH_CHECK(who);
    return 0;
}

extern "C" int cpn_encode_key_r4(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                              const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                              const uint16_t* k80blk, int group, const uint16_t* kw, const float* kbias, int B, int V, int R,
                              int S, int ray0, int nrays, uint16_t* hid, uint16_t* kh, void* stream) {
    return encode_key_launch(tab, map3, H, W, pixel_val, sec_grid, pe6, wfrag, bias, k80blk, group, EK_WAVES_DEFAULT, kw, kbias, B, V,
                             R, S, ray0, nrays, hid, kh, stream, "cpn_encode_key");
}

// The same kernel (resident form) with 8 waves per workgroup instead of 12: 2 waves x 160 VGPRs per SIMD and 155.5 KiB of LDS
// leave 192 registers per SIMD and 4.5 KiB of LDS on every CU to the waves of ANOTHER kernel - cpn_attend_hidden of the
// chunks before this one (53 VGPRs, 544 B), launched on a second stream, then runs UNDER this launch instead of behind it
// (coponerf_amd/render.py: the slot schedule; tools/coresident_probe.py).  Alone it is 5-7 % slower than the 12-wave form.
// hid / kh are bit-identical to cpn_encode_key's.
extern "C" int cpn_encode_key_beside_r4(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                                     const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                                     const uint16_t* kw, const float* kbias, int B, int V, int R, int S, int ray0, int nrays,
                                     uint16_t* hid, uint16_t* kh, void* stream) {
    return encode_key_launch(tab, map3, H, W, pixel_val, sec_grid, pe6, wfrag, bias, nullptr, 0, EK_WAVES_BESIDE, kw, kbias, B, V, R, S,
                             ray0, nrays, hid, kh, stream, "cpn_encode_key_beside");
}
