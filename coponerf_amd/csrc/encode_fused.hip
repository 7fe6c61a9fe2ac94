// K2+K3a+K3c(+K3b) — the first encoder layer on the node tables (encode.hip) with the folded layers BEHIND it fed from
// registers, several 16-row units per wave:
//     hid[row]    = ReLU(query_encode_latent([gather ‖ tanh(pt/5)]))                         (/root/reference models/CoPoNeRF.py:312, 370, 384-397)
//     kh[sample]  = ReLU( (Wk_a W2 | Wk_b W2) . [hid_own ; hid_other] + c' )                  (:404-407 key_map after query_encode_latent_2, folded: DESIGN.md 4.3)
//     val[sample] =       (Wv_a W2 | Wv_b W2) . [hid_own ; hid_other]                         (:404 latent_value, folded; NV = 26 only)
//
// Round 4's cpn_encode_key gave every wave ONE unit (4 rays x 4 samples x both images): each 1 KiB weight fragment read from
// LDS fed a single MFMA, 28 ds_read_b128 per 28 MFMAs and slice, and the 12 waves of a workgroup, in lock step behind the
// weight ring, were all in the same phase at the same time.  Here a wave owns UNITS units: a fragment is read once and
// multiplied against every unit's B operand (the A fragment stays in registers: 28 / UNITS LDS reads per unit and slice),
// the units' dependency chains are independent instruction streams inside one wave (the tap blend of one under the MFMAs
// of the other), and the taps of slice n + 1 are issued BEFORE the key / value MFMAs of slice n, the stores of slice n
// right behind them (the youngest operations in flight: no wait of the next slice has to cover them).
//
//   NV = 0  ("key" form):  hid is written (the two hidden sums read it), kh for key_map_2 + the logit; the K = 80 fragments
//           of the first layer stay resident in LDS (123.5 KiB), the key weights go through a two-slot ring of 16 KiB.
//   NV = 26 ("project before you store"): the 416-wide value projection runs per SAMPLE on the slice that is still in
//           registers; hid never reaches HBM: 832 + 256 bytes per sample leave instead of 3 328 + 256, and both attention
//           rounds read 832 bytes per sample (cpn_attend_value).  Per slice step 68 KiB of key + value fragments and the
//           slice's 9.5 KiB K = 80 block stream through a two-slot ring (2 x 78 KiB: nothing is resident); a wave holds
//           UNITS x 34 accumulator tiles (272 registers at UNITS = 2: one wave per SIMD).
// Ring protocol (one s_barrier per slice step q, every wave the same number of steps): a wave arrives at barrier q only
// after its MFMAs of step q - 1 have been issued (their LDS reads have returned) and its own DMA pieces of slot q & 1 have
// landed (counted vmcnt: every operation issued since is younger); behind barrier q it refills slot (q + 1) & 1 and may
// read slot q & 1.  MFMA shapes and the k order of every accumulator are those of encode.hip / encode_key.hip: hid and kh
// are bit-identical to cpn_encode_hidden + cpn_gemm_f16(key_fold).
#include <algorithm>

// timing-only ablations (results are wrong when non-zero): 1 = no table taps, 2 = no hid / val stores, 4 = no K = 80 MFMA,
// 8 = no key / value MFMA (no LDS reads of the ring either), 64 = no ring traffic (no DMA, no barrier), 256 = no DMA (barrier
// kept), 512 = no barrier (DMA kept: racy), 1024 = only the first 16 pieces of a step are fetched
#ifndef CPN_EF_ABLATE
#define CPN_EF_ABLATE 0
#endif

#ifndef CPN_EF_PRIO
#define CPN_EF_PRIO 0          // > 0: s_setprio of that value around the MFMA phases of a step
#endif
#ifndef CPN_EF_STAGE
#define CPN_EF_STAGE 0         // key form; 0: the ring is refilled by LDS-DMA (buffer_load ... lds); 1: through registers (buffer_load + ds_write)
#endif
#ifndef CPN_EF_STAGE_PROJECT
#define CPN_EF_STAGE_PROJECT 2
#endif
#ifndef CPN_EF_ACC_F16
#define CPN_EF_ACC_F16 1       // 1: the K = 80 accumulators change layout as fp16 pairs (8 ds_bpermute instead of 16); 0: as fp32 (rounds 4-5)
#endif
#ifndef CPN_EF_FRAG_DEPTH
#define CPN_EF_FRAG_DEPTH 4
#endif

#include "encode_common.h"

namespace {

constexpr int WMAIN_HALF8 = NSLICE * 2 * NT * 64;              // [slice][k < 2][tile][lane] half8: 104 KiB
constexpr int WTAIL_HALF4 = NSLICE * NT * 48;                  // [slice][tile][K group < 3][A-operand row] half4: 19.5 KiB
constexpr int KT = 8;                                          // 16-wide output tiles of the key layer (128)
constexpr int KSTEPS = 2 * NSLICE;                             // slice steps per unit: both images
constexpr int K80_PIECES = 10;                                 // the streamed K = 80 block of a slice: 8 KiB main + 1.5 KiB tail, 10 KiB

// s_waitcnt that waits for vmcnt <= N only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// SITE: where a wave takes the ring barrier of a step: 0 = at the top of the step (in front of the K = 80 MFMAs; the only
// legal site when the K = 80 block itself comes through the ring), 1 = in front of the key / value MFMAs, 2 = waves
// 4-7 (one of the two waves of each SIMD) at the top, the others in front of the key MFMAs (s_barrier only counts arrivals: the two groups then run
// half a step apart, one blends while the other multiplies - DESIGN.md 4.1b)
template <int WAVES, int UNITS, int NV, int SITE>
__global__ __launch_bounds__(64 * WAVES, 1) void encode_fused_kernel(
    const __half* __restrict__ tab, const __half* __restrict__ map3, int H, int W,
    const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, const float* __restrict__ pe6,
    const half8* __restrict__ wfrag, const float* __restrict__ bias, const __half* __restrict__ wring,
    const float* __restrict__ kbias, int V, int R, int S, int ray0, int nrays, int nsblk,
    int groups_per_b, long long group0, long long nunits, __half* __restrict__ hid, __half* __restrict__ kh,
    __half* __restrict__ val, int kh_units) {
    constexpr bool PROJECT = NV > 0;
    constexpr bool STORE_HID = !PROJECT;
    constexpr int NTOT = KT + NV;                              // accumulator tiles per unit
    constexpr int WPIECES = NTOT * 2;                          // 1 KiB fragments (tile, k step) per slice step
    constexpr int PIECES = WPIECES + (PROJECT ? K80_PIECES : 0);
    constexpr int SLOT_BYTES = PIECES * 1024;
    constexpr int FPW = (PIECES + WAVES - 1) / WAVES;          // DMA instructions per wave and step (the same for every wave)
    constexpr int FD = CPN_EF_FRAG_DEPTH;                      // weight fragments in flight from LDS ahead of their MFMAs
    constexpr bool NEED_DUMP = true;                            // (also the target of the last step's dead refill)
    static_assert(FD >= 4, "the K = 80 phase issues its 4 tail fragments from the last FD main slots");
    static_assert(SITE == 0 || !PROJECT, "the streamed K = 80 block needs the barrier at the top of the step");
    constexpr int U = UNITS;

    __shared__ __attribute__((aligned(16))) half8 wmain[PROJECT ? 1 : WMAIN_HALF8];
    __shared__ __attribute__((aligned(16))) half4 wtail_s[PROJECT ? 1 : WTAIL_HALF4];
    __shared__ __attribute__((aligned(16))) half8 kring[2 * SLOT_BYTES / 16];
    __shared__ __attribute__((aligned(16))) half8 dump[NEED_DUMP ? 64 : 1];     // target of the DMA slots that carry no piece

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if constexpr (!PROJECT) {
        for (int i = tid; i < WMAIN_HALF8; i += 64 * WAVES) wmain[i] = wfrag[i];
        const half4* tsrc = reinterpret_cast<const half4*>(wfrag + WMAIN_HALF8);          // K tail + bias: see encode.hip
        for (int i = tid; i < WTAIL_HALF4; i += 64 * WAVES) {
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

    const int r = lane & 15, g = lane >> 4;                   // MFMA layout: column (row of the tile) r, K / channel group g
    const int rl = lane >> 2, pl = lane & 3;                  // load layout: row rl, 16-byte piece pl
    const int tail_lane = min(g, 2) * 16 + r;
    const int to_ll = (rl + 16 * pl) * 4;                     // ds_bpermute address: this lane takes MFMA lane (r = rl, g = pl)
    const int to_mfma = (4 * r + g) * 4;                      //                      this lane takes load-layout lane (rl = r, pl = g)
    const int qodd = (lane >> 2) & 1;
    const NodeGrid ng{W >> 1, H >> 1};
    const size_t img_bytes = (size_t)ng.nodes_per_image() * TAB_ROW_BYTES;
    const char* const tbase = reinterpret_cast<const char*>(tab);
    const char* const m3base = reinterpret_cast<const char*>(map3);

    // ---- the weight ring.  `wring` holds, per slice step q = (image j, slice n), the step's slot image: fragment (t, k) =
    //      1 KiB [lane = a + 16 g][8] of W'[16 t + a][832 j + 64 n + 32 k + 8 g .. +8] (t < 8: key, t >= 8: value), then
    //      (PROJECT) the slice's K = 80 block.  Every DMA piece is 1 KiB of contiguous memory.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)wring, 0, KSTEPS * SLOT_BYTES, 0x00020000);
    // (`live` = false: the same FPW instructions with every source out of range, into the dump - the step count of a wave's
    // DMA operations never depends on a branch, so the compiler's vmcnt bookkeeping for the taps stays exact)
    auto ring_fill = [&](int step_in_unit, int slot, bool live) {
        if (CPN_EF_ABLATE & (64 | 256)) return;
#pragma unroll
        for (int f = 0; f < FPW; ++f) {
            const int p = f * WAVES + wave;                    // wave-uniform
            const bool real = live && p < ((CPN_EF_ABLATE & 1024) ? 16 : PIECES);
            lds_void* dst = real ? (lds_void*)(reinterpret_cast<char*>(kring) + slot * SLOT_BYTES + p * 1024)
                                 : (lds_void*)reinterpret_cast<char*>(dump);
            const int so = real ? step_in_unit * SLOT_BYTES + p * 1024 : 0x7ffffff0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, lane * 16, so, 0, 0);
        }
    };
    // STAGE forms of the refill: ordinary buffer loads into registers, ds_write_b128 into the slot behind the barrier that frees
    // it, instead of LDS-DMA.  Measured (profiles/r05_project_before_store.json, tools/ef_check.py): the project form's refill
    // of 78 KiB per slice step costs 10-13 of its 22-24 ms EITHER way (21.7-22.1 ms through registers, 23.3-24.0 by DMA; 12.2
    // with the barrier but no refill, 8.8 with neither), and it is not the instruction count (16 live pieces of 78: 20.6) nor
    // every CU asking its L2 for the same lines at once (slice order rotated per workgroup: no change).  The key form's 16 KiB
    // per step costs 1.5-1.8 of its 11 ms in both forms (DMA 11.1, registers 11.5): DMA stays its default.
    //   CPN_EF_STAGE 1: the pieces of step q + 1 sit in registers from barrier q - 1 to barrier q (FPW x 4 registers for good: the
    //       key form's 2 pieces);  2: requested and written INSIDE step q in two halves - behind the barrier / behind the blend,
    //       behind the taps of q + 1 / two thirds into the key-value MFMAs (the project form's 10 pieces: 20 registers, not 40).
    constexpr int STAGE = PROJECT ? CPN_EF_STAGE_PROJECT : CPN_EF_STAGE;
    constexpr int FH = (FPW + 1) / 2;                          // pieces of the first half (STAGE 2)
    u32x4 stage[STAGE == 2 ? FH : FPW];
    // pieces [f0, f1) of a slot image -> stage[0 ..)
    auto stage_load = [&](int step_in_unit, int f0, int f1) {
        if (CPN_EF_ABLATE & (64 | 256)) return;
#pragma unroll
        for (int f = f0; f < f1; ++f) {
            const int p = f * WAVES + wave;                    // wave-uniform; pieces past the slot read out of range (zeros)
            stage[f - f0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, p < PIECES ? step_in_unit * SLOT_BYTES + p * 1024 : 0x7ffffff0, 0);
        }
    };
    auto stage_store = [&](int slot, int f0, int f1) {
        if (CPN_EF_ABLATE & (64 | 256)) return;
#pragma unroll
        for (int f = f0; f < f1; ++f) {
            const int p = f * WAVES + wave;
            char* dst = p < PIECES ? reinterpret_cast<char*>(kring) + slot * SLOT_BYTES + p * 1024 : reinterpret_cast<char*>(dump);
            *reinterpret_cast<u32x4*>(dst + lane * 16) = stage[f - f0];
        }
    };

    // XCD-aware order as in encode.hip, in UNITS = (4 rays, view, 4 samples) x both images; a wave takes U consecutive
    // units, the waves of a workgroup consecutive groups of U, and every wave runs the same number of iterations
    const unsigned nbk = gridDim.x, nx = nbk < 8 ? nbk : 8;
    const unsigned xcd = blockIdx.x % nx, wgx = blockIdx.x / nx;
    const unsigned wg_on_xcd = nbk / nx + (xcd < nbk % nx ? 1 : 0);
    const long long q = nunits / nx, rem = nunits % nx;
    const long long x_begin = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const long long x_end = x_begin + q + (xcd < rem ? 1 : 0);
    const long long per_iter = (long long)wg_on_xcd * WAVES * U;
    const int iters = (int)((x_end - x_begin + per_iter - 1) / per_iter);
    int gstep = 0;                                            // slice steps this workgroup has started (ring slot = parity)
    if constexpr (STAGE == 1) {
        stage_load(0, 0, FPW);
        stage_store(0, 0, FPW);
        stage_load(1, 0, FPW);
    } else if constexpr (STAGE == 2) {
        stage_load(0, 0, FH);
        stage_store(0, 0, FH);
        stage_load(0, FH, FPW);
        stage_store(0, FH, FPW);
    } else {
        ring_fill(0, 0, true);
    }
    __syncthreads();          // the resident K = 80 fragments are in place (the ring's first slot is waited for at barrier 0)

    const bool top_site = SITE == 0 || (SITE == 2 && ((wave >> 2) & 1)) ||   // waves w, w + 4, w + 8 share a SIMD
                          (SITE == 3 && (wave >> 2) != 0);

    for (int it = 0; it < iters; ++it) {
        // ---- the units of this iteration
        bool ulive[U];
        int ub[U], uv[U], urg[U], usb[U];
        RowId lid[U], mid[U];
        size_t sidx_l[U];
        int hoffA[U], hoffB[U];
        long long trow0[U], uidx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long uu_raw = x_begin + (long long)it * per_iter + ((long long)wgx * WAVES + wave) * U + u;
            uidx[u] = uu_raw;
            ulive[u] = uu_raw < x_end;
            const long long uu = ulive[u] ? uu_raw : x_begin;  // a dead unit walks a live unit's addresses with every row masked
            usb[u] = (int)(uu % nsblk);
            uv[u] = (int)((uu / nsblk) % V);
            const long long gq = group0 + uu / ((long long)nsblk * V);
            ub[u] = (int)(gq / groups_per_b);
            urg[u] = (int)(gq % groups_per_b);
            lid[u] = tile_row(rl, urg[u], usb[u], S, R, ub[u], ray0, nrays);
            mid[u] = tile_row(r, urg[u], usb[u], S, R, ub[u], ray0, nrays);
            lid[u].live = lid[u].live && ulive[u];
            mid[u].live = mid[u].live && ulive[u];
            sidx_l[u] = (((size_t)(ub[u] * V + uv[u])) * R + min(lid[u].r, R - 1)) * S + min(lid[u].s, S - 1);
            trow0[u] = ((((long long)ub[u] * R + (long long)urg[u] * TG - ray0) * V + uv[u]) * S + (long long)usb[u] * TSW) * 2;
        }
        // hid stores: unconditional nt buffer stores (encode.hip, CPN_ENCODE_STORE 7) through ONE descriptor per iteration
        // based at unit 0's tile (the units of a wave are consecutive: a few MB apart at most); dead rows' offsets are out
        // of range and the hardware drops them
        constexpr int kOOB = 0x7ffffff0;
        const unsigned long long hb = (unsigned long long)(hid + trow0[0] * 832);
        const unsigned long long hbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(hb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)hb);
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)hbu, 0, 0x7ff00000, 0x00020000);
        if constexpr (STORE_HID) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const RowId a = tile_row(rl & ~1, urg[u], usb[u], S, R, ub[u], ray0, nrays);
                const RowId b2 = tile_row(rl | 1, urg[u], usb[u], S, R, ub[u], ray0, nrays);
                const long long delta = ulive[u] ? (trow0[u] - trow0[0]) * 1664 : 0;
                auto out_off = [&](const RowId& id) {
                    const int rel = (((id.r - urg[u] * TG) * V * S + (id.s - usb[u] * TSW)) * 2) * 1664 + (pl + 4 * qodd) * 16;
                    return (id.live && ulive[u]) ? (int)delta + rel : kOOB;
                };
                hoffA[u] = out_off(a);
                hoffB[u] = out_off(b2);
            }
        }

        f32x4 kacc[U][NTOT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NTOT; ++t) kacc[u][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

#pragma unroll 1
        for (int j0 = 0; j0 < 2; ++j0) {
            // ---- per-row records of image j0 in the LOAD layout (j0 = 0: own image, border table, pixel_val;
            //      j0 = 1: other image, zeros table, sec_grid)
            const bool own = j0 == 0;
            float tw[U][4];
            int vo[U][4];
            half8 xa[U][2];
            half4 xt[U];
            const char* tb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int img_own = ub[u] * V + uv[u], img_oth = ub[u] * V + (V - 1 - uv[u]);
                const float2 gc = *reinterpret_cast<const float2*>((own ? pixel_val : sec_grid) + sidx_l[u] * 2);
                TapRec rec = node_taps(gc.x, gc.y, ng, own);
                const Taps t3 = make_taps(gc.x, gc.y, W, H, own);
                const char* m3 = m3base + (size_t)(own ? img_own : img_oth) * H * W * 128 + pl * 16;
                u32x4 tv[2][4];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        tv[k][t] = *reinterpret_cast<const u32x4*>(m3 + (size_t)(unsigned)t3.off[t] * 128 + k * 64);
                half8 xl[2];
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
                    for (int e = 0; e < 8; ++e) xl[k][e] = lid[u].live ? (_Float16)a8[e] : (_Float16)0.0f;
                }
                if (!lid[u].live) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) { rec.off[t] = 0; rec.w[t] = 0.0f; }
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32x4 src = __builtin_bit_cast(u32x4, xl[k]);
                    u32x4 dst;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned sv = src[i];
                        dst[i] = (unsigned)__builtin_amdgcn_ds_bpermute(to_mfma, (int)sv);
                    }
                    xa[u][k] = __builtin_bit_cast(half8, dst);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) xt[u][e] = (_Float16)0.0f;
                if (g == 0 && mid[u].live) {
                    const float* pe = pe6 + ((((size_t)(ub[u] * V + uv[u])) * R + mid[u].r) * S + mid[u].s) * 6 + j0 * 3;
                    xt[u][0] = (_Float16)pe[0]; xt[u][1] = (_Float16)pe[1]; xt[u][2] = (_Float16)pe[2];
                    xt[u][3] = (_Float16)1.0f;                // x bias (hi)
                }
                if (g == 1 && mid[u].live) xt[u][0] = (_Float16)1.0f;   // x bias (lo)
#pragma unroll
                for (int k = 0; k < 4; ++k) { vo[u][k] = rec.off[k] + pl * 16; tw[u][k] = rec.w[k]; }
                tb[u] = own ? tbase + img_bytes * img_own : tbase + img_bytes * img_oth + (size_t)ng.border_nodes() * TAB_ROW_BYTES;
            }
            const int tab_bytes = (int)((own ? ng.border_nodes() : ng.zeros_nodes()) * TAB_ROW_BYTES);

            u32x4 td[U][4][2];
            auto issue_taps = [&](int n) {
                if (CPN_EF_ABLATE & 1) return;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const unsigned long long tp = (unsigned long long)tb[u];
                    const unsigned long long tpu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tp >> 32)) << 32) |
                                                   (unsigned)__builtin_amdgcn_readfirstlane((int)tp);
                    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)tpu, 0, tab_bytes, 0x00020000);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        td[u][k][0] = __builtin_amdgcn_raw_buffer_load_b128(trs, vo[u][k], n * TAB_SLICE_BYTES, 0);
                        td[u][k][1] = __builtin_amdgcn_raw_buffer_load_b128(trs, vo[u][k] + 64, n * TAB_SLICE_BYTES, 0);
                    }
                }
            };
            issue_taps(0);
            __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
            for (int n = 0; n < NSLICE; ++n) {
                const int step_in_unit = j0 * NSLICE + n;
                // ---- ring barrier of this step (see the header).  Everything a wave has issued since its DMA pieces of this
                //      step's slot - the taps of this step (top site) or the next one (late site), and the hid stores behind them -
                //      is younger than the pieces: a counted wait leaves exactly those in flight.
                auto ring_sync = [&](bool top) {
                    if (CPN_EF_ABLATE & 64) return;
                    constexpr int TAPS = (CPN_EF_ABLATE & 1) ? 0 : 8 * U, STORES = (STORE_HID && !(CPN_EF_ABLATE & 2)) ? 2 * U : 0;
                    if constexpr (STAGE == 0) {
                        if (top || n + 1 < NSLICE) wait_vm<TAPS + STORES>();
                        else wait_vm<STORES>();                // late site, last slice of an image: no taps were issued behind the pieces
                    }
                    if (!(CPN_EF_ABLATE & 512)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if constexpr (STAGE == 1) {
                        // behind barrier q: the slot of step q - 1 is free - step q + 1's pieces (in registers since the previous
                        // step) go in, step q + 2's are requested (the slot images repeat with period KSTEPS)
                        stage_store((gstep + 1) & 1, 0, FPW);
                        stage_load((step_in_unit + 2) % KSTEPS, 0, FPW);
                    } else if constexpr (STAGE == 2) {
                        stage_load((step_in_unit + 1) % KSTEPS, 0, FH);
                    } else {
                        const bool last = (it == iters - 1) && (step_in_unit == KSTEPS - 1);
                        ring_fill((step_in_unit + 1) % KSTEPS, (gstep + 1) & 1, !last);
                    }
                };
                if (top_site) ring_sync(true);
                __builtin_amdgcn_sched_barrier(0);
                const char* const sbase = reinterpret_cast<const char*>(kring) + (gstep & 1) * SLOT_BYTES;
                const half8* const wm = PROJECT ? reinterpret_cast<const half8*>(sbase + WPIECES * 1024) : wmain + n * 2 * NT * 64;
                const half4* const wt = PROJECT ? reinterpret_cast<const half4*>(sbase + WPIECES * 1024 + 8192) : wtail_s + n * NT * 48;

                // ---- K = 80 contraction of the full-resolution level + point encoding + bias: a fragment serves all units
                f32x4 acc[U][NT];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[u][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (!(CPN_EF_ABLATE & 4)) {
                    // fragment i < 8: main (k = i / 4, tile i % 4); the LDS reads run FD fragments ahead of the MFMAs
                    half8 am[FD];
#pragma unroll
                    for (int d = 0; d < FD; ++d) am[d] = wm[d * 64 + lane];
                    half4 at[NT];
#pragma unroll
                    for (int i = 0; i < NT && i + 2 * NT < FD; ++i) at[i] = wt[i * 48 + tail_lane];     // (FD > 8 only)
#pragma unroll
                    for (int i = 0; i < 2 * NT; ++i) {
                        const half8 a = am[i % FD];
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            acc[u][i % NT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xa[u][i / NT], acc[u][i % NT], 0, 0, 0);
                        if (i + FD < 2 * NT) am[i % FD] = wm[(i + FD) * 64 + lane];
                        else if (i + FD - 2 * NT < NT) at[i + FD - 2 * NT] = wt[(i + FD - 2 * NT) * 48 + tail_lane];
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x16f16(at[nt], xt[u], acc[u][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
                    for (int i = 0; i < 2 * NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, U, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NT * U, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                half8 res[U][2];                               // fp16 results of this slice in the load layout
                half8 xb[U][2];                                // ... and as the B operand of the key / value layer
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (CPN_EF_ABLATE & 2048) {
                        // (timing only: the accumulators stay where they are)
                    } else {
#if CPN_EF_ACC_F16
                    // the K = 80 partial sums cross to the load layout as fp16 PAIRS (round 6): 8 ds_bpermute per unit and slice
                    // instead of 16 on the kernel's busiest pipe.  One more rounding (2^-11 of the partial sum, which is one of
                    // five summands of a value that is rounded to fp16 two instructions later); hid is no longer the bits of the
                    // round-2 kernel, the bound that counts is the one against the oracle (tests/test_gpu_parity.py)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const f32x2v two = {acc[u][nt][2 * i], acc[u][nt][2 * i + 1]};
                            const half2v hv = __builtin_convertvector(two, half2v);
                            const unsigned got = (unsigned)__builtin_amdgcn_ds_bpermute(to_ll, (int)__builtin_bit_cast(unsigned, hv));
                            const half2v back = __builtin_bit_cast(half2v, got);
                            acc[u][nt][2 * i] = (float)back[0];
                            acc[u][nt][2 * i + 1] = (float)back[1];
                        }
#else
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float t = acc[u][nt][i];
                            acc[u][nt][i] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_ll, __float_as_int(t)));
                        }
#endif
                    }
                    // ---- 4 table taps per row in fp32 on top of it, ReLU, fp16
                    if (!(CPN_EF_ABLATE & 1)) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float wk = tw[u][k];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const u32x4 d = td[u][k][h];
                                f32x4* a2 = &acc[u][2 * h];
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
                            const f32x4& src = acc[u][2 * h + (qq >> 1)];
                            const f32x2v two = {src[2 * (qq & 1)], src[2 * (qq & 1) + 1]};
                            half2v hv = __builtin_convertvector(two, half2v);
                            hv = __builtin_elementwise_max(hv, (half2v){(_Float16)0.0f, (_Float16)0.0f});
                            pk[qq] = __builtin_bit_cast(unsigned, hv);
                        }
                        res[u][h] = __builtin_bit_cast(half8, pk);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const u32x4 src = __builtin_bit_cast(u32x4, res[u][k]);
                        u32x4 dst;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const unsigned sv = src[i];
                            dst[i] = (CPN_EF_ABLATE & 4096) ? sv : (unsigned)__builtin_amdgcn_ds_bpermute(to_mfma, (int)sv);
                        }
                        xb[u][k] = __builtin_bit_cast(half8, dst);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- the taps of the NEXT slice go out before this slice's key / value MFMAs, this slice's hid stores behind them
                if constexpr (STAGE == 2) stage_store((gstep + 1) & 1, 0, FH);
                if (n + 1 < NSLICE) issue_taps(n + 1);
                if constexpr (STAGE == 2) stage_load((step_in_unit + 1) % KSTEPS, FH, FPW);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (STORE_HID) {
                    if (!(CPN_EF_ABLATE & 2)) {
                        const int co = __builtin_amdgcn_readfirstlane((j0 * 832 + n * SLICE_CH) * 2);       // scalar offset operand
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const u32x4 h0 = __builtin_bit_cast(u32x4, res[u][0]), h1 = __builtin_bit_cast(u32x4, res[u][1]);
                            u32x4 sa, sb;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const unsigned a0 = h0[i], a1 = h1[i];
                                sb[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a1, (int)a0, 0x104, 0xF, 0x5, false);
                                sa[i] = (unsigned)__builtin_amdgcn_update_dpp((int)a0, (int)a1, 0x114, 0xF, 0xA, false);
                            }
                            // gfx950 hazard (encode_key.hip): both offsets exist before the first store, two wait states follow the second
                            int oa = hoffA[u], ob = hoffB[u];
                            asm volatile("" : "+v"(oa), "+v"(ob));
                            __builtin_amdgcn_raw_buffer_store_b128(sa, hrs, oa, co, 2);          // aux 2 = nt
                            __builtin_amdgcn_raw_buffer_store_b128(sb, hrs, ob, co, 2);
                            asm volatile("s_nop 1" ::: "memory");
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!top_site) ring_sync(false);
                __builtin_amdgcn_sched_barrier(0);
                // ---- key (and value) layer on this slice: K = 64 of the 2 x 832; a fragment serves all units
                if (CPN_EF_PRIO) __builtin_amdgcn_s_setprio(CPN_EF_PRIO);
                if (!(CPN_EF_ABLATE & 8)) {
                    // fragment i = k * NTOT + t (k outer: two MFMAs on one accumulator are NTOT * U instructions apart)
                    const half8* slot = reinterpret_cast<const half8*>(sbase);
                    constexpr int NF = 2 * NTOT;
                    half8 af[FD];
#pragma unroll
                    for (int d = 0; d < FD; ++d) af[d] = slot[((d % NTOT) * 2 + d / NTOT) * 64 + lane];
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        const half8 a = af[i % FD];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            // PROJECT: 2 x 34 accumulator tiles are 272 registers, 16 more than there are AGPRs, and with AGPRs in
                            // play the compiler selects the AGPR-destination form for EVERY MFMA: the surplus tiles would be
                            // shuttled through v_accvgpr moves around each of their MFMAs (150 moves per step).  The key tiles'
                            // MFMAs are therefore written with VGPR accumulators by hand.
                            if (PROJECT && (i % NTOT) < KT)
                                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(kacc[u][i % NTOT]) : "v"(a), "v"(xb[u][i / NTOT]));
                            else
                                kacc[u][i % NTOT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xb[u][i / NTOT], kacc[u][i % NTOT], 0, 0, 0);
                        }
                        if (i + FD < NF) af[i % FD] = slot[(((i + FD) % NTOT) * 2 + (i + FD) / NTOT) * 64 + lane];
                        if (STAGE == 2 && i == (2 * NF) / 3) stage_store((gstep + 1) & 1, FH, FPW);
                    }
                    if constexpr (!PROJECT) {
                        __builtin_amdgcn_sched_group_barrier(0x100, FD, 0);
#pragma unroll
                        for (int i = 0; i < NF - FD; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, U, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, FD * U, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (CPN_EF_PRIO) __builtin_amdgcn_s_setprio(0);
                ++gstep;
            }
        }

        // ---- kh = fp16(ReLU(acc + c')), val = fp16(acc): lane (r, g) holds outputs t*16 + g*4 .. +4 of row r.
        //      v_permlane16_swap (gfx950) trades the odd 16-lane rows of tile t with the even rows of tile t + 1: lane (r, g)
        //      then holds 8 consecutive outputs of tile t + (g & 1) - stores of 16 bytes, 64 contiguous bytes per row and pair.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t srow = (((size_t)ub[u] * R + mid[u].r - ray0) * V + uv[u]) * S + mid[u].s;
            const bool store = mid[u].live;
            unsigned hw[NTOT][2];
#pragma unroll
            for (int t = 0; t < NTOT; ++t) {
                half4 o;
                if (t < KT) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(kbias + t * 16 + g * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)fmaxf(kacc[u][t][i] + bv[i], 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)kacc[u][t][i];
                }
                const u32x2 w2 = __builtin_bit_cast(u32x2, o);
                hw[t][0] = w2[0];
                hw[t][1] = w2[1];
            }
            // kh_units: the (rows, 128) matrix in UNIT order (include/coponerf_hip.h) - [unit][32-column block q][lane = row of the
            // unit + 16 * 8-column group][8]: the swapped piece of lane (r, g) is the 8-column group fg = 2 (g & 1) + (g >> 1) of
            // block q, i.e. exactly the MFMA B fragment the per-sample kernels behind this one read (cpn_local_units), and the
            // wave's store is 1 KiB of contiguous memory; otherwise row-major rows (64 contiguous bytes per row and store)
            __half* kdst = kh_units ? kh + ((size_t)uidx[u] * 4 * 64 + (r + 16 * (2 * (g & 1) + (g >> 1)))) * 8
                                    : kh + srow * 128 + (g & 1) * 16 + (g >> 1) * 8;
            const bool kstore = kh_units ? ulive[u] : store;
            const int kstep = kh_units ? 64 * 8 : 32;
#pragma unroll
            for (int qq = 0; qq < KT / 2; ++qq) {
                const u32x2 lo = __builtin_amdgcn_permlane16_swap(hw[2 * qq][0], hw[2 * qq + 1][0], false, false);
                const u32x2 hi = __builtin_amdgcn_permlane16_swap(hw[2 * qq][1], hw[2 * qq + 1][1], false, false);
                const u32x4 piece = {lo[0], hi[0], lo[1], hi[1]};
                if (kstore) *reinterpret_cast<u32x4*>(kdst + qq * kstep) = piece;
            }
            if constexpr (PROJECT) {
                __half* vdst = val + srow * (NV * 16) + (g & 1) * 16 + (g >> 1) * 8;
#pragma unroll
                for (int qq = 0; qq < NV / 2; ++qq) {
                    const int t0 = KT + 2 * qq;
                    const u32x2 lo = __builtin_amdgcn_permlane16_swap(hw[t0][0], hw[t0 + 1][0], false, false);
                    const u32x2 hi = __builtin_amdgcn_permlane16_swap(hw[t0][1], hw[t0 + 1][1], false, false);
                    const u32x4 piece = {lo[0], hi[0], lo[1], hi[1]};
                    if (store && !(CPN_EF_ABLATE & 2)) *reinterpret_cast<u32x4*>(vdst + qq * 32) = piece;
                }
            }
        }
    }
}

}  // namespace

// compile-time shape of the two product forms (tools/ef_ablate.py builds other shapes into tools/_build)
#ifndef CPN_EF_KEY_WAVES
#define CPN_EF_KEY_WAVES 12
#endif
#ifndef CPN_EF_KEY_UNITS
#define CPN_EF_KEY_UNITS 1
#endif
#ifndef CPN_EF_KEY_SITE
#define CPN_EF_KEY_SITE 2
#endif
#ifndef CPN_EF_PROJ_WAVES
#define CPN_EF_PROJ_WAVES 8
#endif
#ifndef CPN_EF_PROJ_UNITS
#define CPN_EF_PROJ_UNITS 1
#endif

static int encode_fused_launch(bool project, const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                               const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                               const uint16_t* wring, const float* kbias, int B, int V, int R, int S, int ray0, int nrays,
                               uint16_t* hid, uint16_t* kh, uint16_t* val, int kh_units, void* stream, const char* who) {
    CPN_REQUIRE(tab && map3 && pixel_val && sec_grid && pe6 && wring && kbias && kh, CPN_E_ARG, "%s: null pointer", who);
    CPN_REQUIRE(project ? (val != nullptr) : (wfrag && bias && hid), CPN_E_ARG, "%s: null pointer", who);
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && W >= 16 && (H % 16) == 0 && (W % 16) == 0,
                CPN_E_SHAPE, "%s: need V==2 and H,W multiples of 16 (got H=%d W=%d V=%d)", who, H, W, V);
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "%s: ray range [%d,%d) outside B*R=%lld", who, ray0, ray0 + nrays, (long long)B * R);
    const long long nrows = (long long)nrays * V * S * 2;
    const NodeGrid ng{W >> 1, H >> 1};
    constexpr int UMAX = CPN_EF_KEY_UNITS > CPN_EF_PROJ_UNITS ? CPN_EF_KEY_UNITS : CPN_EF_PROJ_UNITS;
    CPN_REQUIRE(nrows < (1LL << 31) && ng.zeros_nodes() * TAB_ROW_BYTES < (1LL << 31) && (long long)H * W * 128 < (1LL << 31) &&
                    (long long)(UMAX + 1) * TG * V * S * 2 * 1664 < 0x7ff00000LL,
                CPN_E_SHAPE, "%s: chunk / per-image table too large for 32-bit offsets (%lld rows)", who, nrows);
    CPN_REQUIRE(((uintptr_t)tab % 16) == 0 && ((uintptr_t)map3 % 16) == 0 && ((uintptr_t)wfrag % 16) == 0 &&
                    ((uintptr_t)bias % 16) == 0 && ((uintptr_t)hid % 16) == 0 && ((uintptr_t)wring % 16) == 0 &&
                    ((uintptr_t)kbias % 16) == 0 && ((uintptr_t)kh % 16) == 0 && ((uintptr_t)val % 16) == 0, CPN_E_ARG,
                "%s: pointers must be 16-byte aligned", who);
    const int groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    const long long group0 = (long long)b_lo * groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    const int nsblk = (int)cpn_cdiv(S, TSW);
    const long long nunits = (group1 - group0 + 1) * V * nsblk;
    const int num_cu = cpn_stream_cus((void*)stream);
    if (project) {
        // ("project before you store", cpn_encode_project: measured in round 5 - it loses, DESIGN.md / HISTORY.md - and out of
        // the library since round 6; the kernel template still carries the form: tools/ef_check.py builds it)
        cpn_set_error("%s: the project form is not part of this build", who);
        return CPN_E_ARG;
    } else {
        constexpr int WV = CPN_EF_KEY_WAVES, UN = CPN_EF_KEY_UNITS;
        const unsigned grid = (unsigned)std::min<long long>(num_cu, cpn_cdiv(nunits, WV * UN));
        hipLaunchKernelGGL((encode_fused_kernel<WV, UN, 0, CPN_EF_KEY_SITE>), dim3(grid), dim3(64 * WV), 0, (hipStream_t)stream,
                           (const __half*)tab, (const __half*)map3, H, W, pixel_val, sec_grid, pe6, (const half8*)wfrag, bias,
                           (const __half*)wring, kbias, V, R, S, ray0, nrays, nsblk, groups_per_b, group0, nunits,
                           (__half*)hid, (__half*)kh, (__half*)val, kh_units);
    }
    CPN_LAUNCH_CHECK(who);
    return 0;
}

extern "C" int cpn_encode_key(const uint16_t* tab, const uint16_t* map3, int H, int W, const float* pixel_val,
                               const float* sec_grid, const float* pe6, const uint16_t* wfrag, const float* bias,
                               const uint16_t* kwring, const float* kbias, int B, int V, int R, int S, int ray0, int nrays,
                               uint16_t* hid, uint16_t* kh, int kh_units, void* stream) {
    return encode_fused_launch(false, tab, map3, H, W, pixel_val, sec_grid, pe6, wfrag, bias, kwring, kbias, B, V, R, S, ray0,
                               nrays, hid, kh, nullptr, kh_units, stream, "cpn_encode_key");
}

// units (4 rays x 4 samples of one view) a launch over rays [ray0, ray0 + nrays) walks: the row count / 16 of its unit-order outputs
extern "C" long long cpn_encode_units(int B, int R, int S, int ray0, int nrays) {
    if (B <= 0 || R <= 0 || S <= 0 || ray0 < 0 || nrays <= 0 || (long long)ray0 + nrays > (long long)B * R) return -1;
    const int groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    const long long group0 = (long long)b_lo * groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    return (group1 - group0 + 1) * 2 * (long long)cpn_cdiv(S, TSW);
}
