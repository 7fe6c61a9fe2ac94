// K3b/K4 logits in UNIT order (round 5) — the per-sample query / key tails of the two attention rounds on the 16-row units the
// first-layer kernel works in (4 rays x 4 samples of one view, csrc/encode_fused.hip), so that its 128-wide key hidden layer
// reaches them as ready MFMA B fragments and the key / coords_embed matrices never exist in row-major form:
//
//   mode 0 (round 1, /root/reference models/CoPoNeRF.py:408, 446, 450):
//       ce    = query_embed_2(ReLU(query_embed(local_coords)))                         -> ce_u (unit order, for round 2)
//       key   = key_map_2(kh)            kh = cpn_encode_key's unit-order output
//       logit = <fp16(key), fp16(ce)>                                                   -> logits[row] (row order, for the sums)
//     replaces cpn_local_mlp (coords_embed) + cpn_gemm_f16_rowdot (key_map_2 + logit): one pass instead of two, kh read as
//     1 KiB fragments, 2.1 GB of coords_embed written once and read once (round 2) instead of written once and read twice.
//   mode 1 (round 2, :472-475):
//       q2    = query_repeat_embed_2(ReLU(W_l . local_coords + b + add[ray]))           add = W_z . encode_latent(z_local)
//       logit = <fp16(q2), ce_u>                                                        -> logits[row]
//     = cpn_local_mlp's logits form with the unit row map.
//   mode 2 (round 2 with coords_embed RECOMPUTED): mode 1 without its 256-byte-per-sample read of ce_u - the two query_embed
//     layers are evaluated again from the local coordinates the kernel reads anyway (56 more MFMAs per unit, the same
//     instructions in the same order as mode 0: the same bits), and mode 0 is then called with ce_u = NULL and stores nothing:
//     2.1 GB less written and 2.1 GB less read per 65 536-ray image, both kernels were bound by that traffic.
// Unit order of a (rows, 128) fp16 matrix X: [unit][32-column block p][lane = c + 16 fg][8] holds X[row(unit, c)][32 p + 8 fg .. +8],
// c = (sample & 3) * 4 + (ray & 3) — a wave's access to one block is 1 KiB of contiguous memory and IS the B operand of
// v_mfma_f32_16x16x32_f16 for the k block p.  Structure (8 waves share the layer-2 fragments in LDS, the first layer's bias on
// the unused K slot, inputs of the next unit requested before the MFMAs of the current one): cpn_local_mlp's (csrc/gather.hip).
#include <algorithm>

#include "encode_common.h"

// timing-only ablations (tools/lu_check.py; results are wrong when non-zero): 1 = the other operand (kh / ce fragments) is not
// loaded, 2 = no `add` rows, 4 = the per-row inputs are fetched once (no loads inside the loop), 8 = no stores,
// 16 = no 128 -> 128 layers (their LDS reads and MFMAs)
#ifndef CPN_LU_ABLATE
#define CPN_LU_ABLATE 0
#endif

namespace {

struct UnitGeo {
    int V, R, S, ray0, nrays, nsblk, groups_per_b;
    long long group0, nunits;
};

// MODE 2: (w1, b1, add, w2, b2) are the round-2 query layers as in mode 1; (w1b, b1b) = query_embed, (wk2, bk2) = query_embed_2
// two accumulator tiles (channels 8 fg .. + 4 and + 4 .. + 8 of a 32-block) -> one fp16 B-operand / fragment register quad, as
// PACKED conversions (v_cvt_pk_f16_f32, round to nearest even) and a packed ReLU behind the rounding (the same value as rounding
// behind the ReLU: rounding is monotone and keeps the sign) - written element by element the compiler emitted a v_max_f32, a
// v_cvt_f16_f32 and half a v_perm_b32 per value, and the kernel's SIMDs were issue-bound (VALU 51 % + MFMA 44 % of their cycles)
template <bool RELU>
__device__ __forceinline__ half8 pack_tiles(const f32x4& lo, const f32x4& hi) {
    half8 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4& src = q < 2 ? lo : hi;
        const f32x2v two = {src[2 * (q & 1)], src[2 * (q & 1) + 1]};
        half2v hv = __builtin_convertvector(two, half2v);
        if (RELU) hv = __builtin_elementwise_max(hv, (half2v){(_Float16)0.0f, (_Float16)0.0f});
        out[2 * q] = hv[0];
        out[2 * q + 1] = hv[1];
    }
    return out;
}

#ifndef CPN_LU_UNITS
#define CPN_LU_UNITS 2         // units per wave in modes 0 and 2 (mode 1 keeps one: 128 registers, four waves per SIMD)
#endif
template <int MODE, int U>
__global__ __launch_bounds__(512, (MODE == 2 || U > 1) ? 2 : 4) void local_units_kernel(
    const float* __restrict__ loc8, const float* __restrict__ coords9, const float* __restrict__ w1, int ldw1,
    const float* __restrict__ b1, const float* __restrict__ add, const __half* __restrict__ w2, int ldw2,
    const float* __restrict__ b2, const __half* __restrict__ wk2, int ldwk2, const float* __restrict__ bk2,
    const float* __restrict__ w1b, int ldw1b, const float* __restrict__ b1b,
    const __half* __restrict__ kh_u, UnitGeo geo, __half* __restrict__ ce_u, const f32x4* __restrict__ lv_u,
    float* __restrict__ logits) {
    __shared__ __attribute__((aligned(16))) half8 w2l[8 * 4 * 64];                       // [tile t][k block p][lane]
    __shared__ __attribute__((aligned(16))) half8 wkl[MODE != 1 ? 8 * 4 * 64 : 1];      // key_map_2 (mode 0) / query_embed_2 (mode 2), same layout
    __shared__ __attribute__((aligned(16))) float b2s[128];
    __shared__ __attribute__((aligned(16))) float bks[128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = lane & 15, fg = lane >> 4;
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += 512) {
        const int l = i & 63, p = (i >> 6) & 3, t = i >> 8;
        const int ch = (t >> 1) * 32 + ((l & 15) >> 2) * 8 + (t & 1) * 4 + (l & 3);       // output channel of tile row
        w2l[i] = *reinterpret_cast<const half8*>(w2 + (size_t)ch * ldw2 + p * 32 + (l >> 4) * 8);
        if constexpr (MODE != 1) wkl[i] = *reinterpret_cast<const half8*>(wk2 + (size_t)ch * ldwk2 + p * 32 + (l >> 4) * 8);
    }
    if (threadIdx.x < 128) {
        b2s[threadIdx.x] = b2[threadIdx.x];
        bks[threadIdx.x] = MODE != 1 ? bk2[threadIdx.x] : 0.0f;
    }
    // First layer (K = 16, fp32 weights and inputs) on the fp16 MFMA as a hi / lo split - w = wh + wl, x = xh + xl (each part
    // an fp16), w . x = wh xh + wh xl + wl xh up to 2^-22 |w x| - three v_mfma_f32_16x16x16_f16 of 8 cycles per tile instead
    // of four v_mfma_f32_16x16x4_f32 of 32: the fp32 MFMA runs at 1/16 of the fp16 rate and was two thirds of this
    // kernel's matrix time (cpn_local_mlp keeps the fp32 form).
    // (the A fragments live in LDS - [set][tile][lane] half4, sets: hi, lo (, query_embed's hi, lo in mode 2) - and are read where
    // they are used: held in registers they were 32 (64) of the 128 a wave may have at four waves per SIMD)
    __shared__ __attribute__((aligned(8))) half4 w1s[MODE == 2 ? 4 : 2][8][64];
    if (wave == 0) {
        auto split = [&](const float* wsrc, int ldw, const float* bsrc, int ch, half4& hi, half4& lo) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(wsrc + (size_t)ch * ldw + fg * 4);
            if (fg == 0) wv[3] = bsrc[ch];                     // K slot 3 is unused by the inputs: bias x 1.0
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hi[i] = (_Float16)wv[i];
                lo[i] = (_Float16)(wv[i] - (float)hi[i]);
            }
        };
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int ch = (t >> 1) * 32 + (a >> 2) * 8 + (t & 1) * 4 + (a & 3);
            half4 hi, lo;
            split(w1, ldw1, b1, ch, hi, lo);
            w1s[0][t][lane] = hi;
            w1s[1][t][lane] = lo;
            if constexpr (MODE == 2) {
                split(w1b, ldw1b, b1b, ch, hi, lo);
                w1s[2][t][lane] = hi;
                w1s[3][t][lane] = lo;
            }
        }
    }
    __syncthreads();
    const unsigned nunits = (unsigned)geo.nunits;
    const unsigned wave_id = blockIdx.x * 8 + wave, nwaves = gridDim.x * 8;
    const int V = geo.V, R = geo.R, S = geo.S;

    struct RowIn {
        f32x4 lv;          // this lane's 4 K entries of the 16-wide input
        unsigned rayrel;   // ray - ray0 (row of `add`)
        long long srow;    // row of the (rows, .) arrays in row order; -1: dead row of the unit
    };
    // row c of unit uu: the map of encode_fused.hip (uu = ((ray group - group0) * V + v) * nsblk + sample block)
    auto fetch = [&](unsigned uu) {
        const int sblk = (int)(uu % (unsigned)geo.nsblk);
        const int v = (int)((uu / (unsigned)geo.nsblk) % (unsigned)V);
        const long long gq = geo.group0 + uu / ((unsigned)geo.nsblk * (unsigned)V);
        const int b = (int)(gq / geo.groups_per_b), rgroup = (int)(gq % geo.groups_per_b);
        const RowId id = tile_row(a, rgroup, sblk, S, R, b, geo.ray0, geo.nrays);
        const int r = min(id.r, R - 1), s = min(id.s, S - 1);
        const size_t nr = ((size_t)(b * V + v)) * R + r;
        const float* lp = loc8 + (nr * S + s) * 8;
        const float* c9 = coords9 + nr * 9;
        RowIn o;
        const long long rayrel = (long long)b * R + r - geo.ray0;
        o.rayrel = (unsigned)max(0LL, min(rayrel, (long long)geo.nrays - 1));
        o.srow = id.live ? (rayrel * V + v) * S + s : -1;
        if (lv_u) {
            // the lane's four inputs as cpn_sample_geometry packed them: ONE coalesced 1 KiB read per unit instead of five
            // scattered ones (3 - 16 bytes per lane from 16 rows and 4 rays: 0.26 of mode 0's 0.9 ms, 0.5 of mode 2's 1.2 ms,
            // tools/lu_check.py).  Rows that do not exist (R or S no multiple of 4) hold whatever the buffer held: their
            // MFMA columns are their own and nothing of them is stored
            o.lv = __builtin_nontemporal_load(lv_u + (size_t)uu * 64 + lane);
            return o;
        }
        if (fg == 0) { const f32x4 l0 = *reinterpret_cast<const f32x4*>(lp); o.lv = f32x4{l0[0], l0[1], l0[2], 1.0f}; }
        else if (fg == 1) o.lv = f32x4{0.f, 0.f, c9[0], c9[1]};
        else if (fg == 2) o.lv = f32x4{c9[2], lp[3], lp[4], lp[5]};
        else o.lv = f32x4{lp[6], c9[6], c9[7], c9[8]};
        return o;
    };

    // A 128 -> 128 layer on the wave's U units: o[u][t] = bias + sum_p W(t, p) . b[u][p].  The 32 fragments are read from LDS FD
    // ahead of the MFMAs that use them, k block outer (two MFMAs on one accumulator are 8 U instructions apart), and every
    // fragment is multiplied against ALL the wave's units: with one unit per wave the 1 KiB fragment read (4 clocks of the CU's
    // one LDS pipe) feeds a single 16-clock MFMA, and four SIMDs asking for one each saturate that pipe exactly when the matrix
    // cores would (measured floor with no global loads at all: 0.55 / 0.67 ms for 0.26 / 0.30 ms of MFMA).  Each accumulator
    // still receives bias, p = 0, 1, 2, 3 in that order: the results do not change.
    auto layer128 = [&](const half8* wfr, const float* bias_s, const half8 (&b)[U][4], f32x4 (&o)[U][8]) {
        constexpr int NF = 32, FD = 4;
        if (CPN_LU_ABLATE & 16) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) o[u][t] = f32x4{b[u][t & 3][0], b[u][t & 3][1], b[u][t & 3][2], b[u][t & 3][3]};
            return;
        }
        half8 af[FD];
#pragma unroll
        for (int d = 0; d < FD; ++d) af[d] = wfr[(((d & 7) * 4) + (d >> 3)) * 64 + lane];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_s + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
#pragma unroll
            for (int u = 0; u < U; ++u) o[u][t] = bv;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int t = i & 7, pblk = i >> 3;
#pragma unroll
            for (int u = 0; u < U; ++u) o[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i % FD], b[u][pblk], o[u][t], 0, 0, 0);
            if (i + FD < NF) af[i % FD] = wfr[((((i + FD) & 7) * 4) + ((i + FD) >> 3)) * 64 + lane];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, FD + 8, 0);      // the first FD fragments + the 8 bias reads
#pragma unroll
        for (int i = 0; i < NF - FD; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, U, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, FD * U, 0);
    };
    // the first layer (K = 16 as three hi / lo MFMAs per tile) of set `ws` (0: w1 / b1, 2: w1b / b1b) on the U units
    auto layer16 = [&](int ws, const half4 (&xh)[U], const half4 (&xl)[U], f32x4 (&o)[U][8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const half4 wh = w1s[ws][t][lane], wl = w1s[ws + 1][t][lane];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(wl, xh[u], o[u][t], 0, 0, 0);
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xl[u], o[u][t], 0, 0, 0);
                o[u][t] = __builtin_amdgcn_mfma_f32_16x16x16f16(wh, xh[u], o[u][t], 0, 0, 0);
            }
        }
    };
    auto add_rows = [&](unsigned rayrel, f32x4 (&dst)[8]) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            dst[t] = (CPN_LU_ABLATE & 2) ? f32x4{0.f, 0.f, 0.f, 0.f}
                                         : *reinterpret_cast<const f32x4*>(add + (size_t)rayrel * 128 + (t >> 1) * 32 + fg * 8 + (t & 1) * 4);
    };

    // a wave takes the units wave_id + (it * U + u) * nwaves; past the end it walks the last unit again with nothing stored
    auto unit_of = [&](unsigned first, int u) { return min(first + (unsigned)u * nwaves, nunits - 1); };
    RowIn cur[U];
    constexpr bool ADD_AHEAD = MODE == 2 && U == 1;            // (two units per wave leave no registers for it, and need it less)
    f32x4 addn[ADD_AHEAD ? U : 1][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        cur[u] = fetch(unit_of(wave_id, u));
        if constexpr (ADD_AHEAD) add_rows(cur[u].rayrel, addn[u]);
    }
    for (unsigned uu = wave_id; uu < nunits; uu += U * nwaves) {
        RowIn nxt[U];
        unsigned un[U];
        bool ulive[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            un[u] = unit_of(uu, u);
            ulive[u] = uu + (unsigned)u * nwaves < nunits;
            nxt[u] = (CPN_LU_ABLATE & 4) ? cur[u] : fetch(unit_of(uu + U * nwaves, u));
        }
        // the other operand of the dot product, as B fragments / accumulator-layout rows: 4 x 1 KiB of contiguous memory
        // (requesting it a unit ahead was measured: no change - the kernel is not waiting for it)
        half8 cv[U][4];
        if constexpr (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    cv[u][p] = (CPN_LU_ABLATE & 1) ? half8{} : __builtin_nontemporal_load(
                        reinterpret_cast<const half8*>(MODE == 0 ? kh_u : ce_u) + ((size_t)un[u] * 4 + p) * 64 + lane);
        }
        f32x4 acc[U][8];
        half4 xh[U], xl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (ADD_AHEAD) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[u][t] = addn[u][t];        // requested an iteration ahead (1.17 -> 0.91 ms)
                add_rows(nxt[u].rayrel, addn[u]);
            } else if constexpr (MODE != 0) {
                add_rows(cur[u].rayrel, acc[u]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[u][i] = (_Float16)cur[u].lv[i];
                xl[u][i] = (_Float16)(cur[u].lv[i] - (float)xh[u][i]);
            }
        }
        if constexpr (MODE == 2) {
            // coords_embed of these units, exactly as mode 0 forms it (same instructions, same order: the bits mode 0 would have stored)
            f32x4 ab[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) ab[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            layer16(2, xh, xl, ab);
            half8 hq[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) hq[u][p] = pack_tiles<true>(ab[u][2 * p], ab[u][2 * p + 1]);
            f32x4 oq[U][8];
            layer128(wkl, bks, hq, oq);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) cv[u][p] = pack_tiles<false>(oq[u][2 * p], oq[u][2 * p + 1]);
        }
        layer16(0, xh, xl, acc);
        // hidden layer -> fp16 B operands: K block p = channels p*32 .. p*32+31, this lane holds fg*8 .. fg*8+7 of it
        half8 hb[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < 4; ++p) hb[u][p] = pack_tiles<true>(acc[u][2 * p], acc[u][2 * p + 1]);
        f32x4 o2[U][8];
        layer128(w2l, b2s, hb, o2);
        long long srow[U];
        float dsum[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            srow[u] = ulive[u] ? cur[u].srow : -1;
            cur[u] = nxt[u];
            dsum[u] = 0.0f;
        }
        if constexpr (MODE == 0) {
            // coords_embed leaves in unit order as it is (the accumulator layout is the fragment layout), and meets the key:
            // key_map_2 on the B fragments of kh, in the same accumulator layout
            half8 ce[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    ce[u][p] = pack_tiles<false>(o2[u][2 * p], o2[u][2 * p + 1]);
                    if (ce_u && ulive[u] && !(CPN_LU_ABLATE & 8))                     // NULL: round 2 recomputes it (mode 2)
                        reinterpret_cast<half8*>(ce_u)[((size_t)un[u] * 4 + p) * 64 + lane] = ce[u][p];
                }
            f32x4 k2[U][8];
            layer128(wkl, bks, cv, k2);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const half8 kp = pack_tiles<false>(k2[u][2 * p], k2[u][2 * p + 1]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dsum[u] += (float)kp[e] * (float)ce[u][p][e];
                }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const half8 qp = pack_tiles<false>(o2[u][2 * p], o2[u][2 * p + 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dsum[u] += (float)qp[i] * (float)cv[u][p][i];
                        dsum[u] += (float)qp[4 + i] * (float)cv[u][p][4 + i];
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = dsum[u];
            d += __shfl_xor(d, 16);
            d += __shfl_xor(d, 32);
            if (srow[u] >= 0 && fg == 0 && (!(CPN_LU_ABLATE & 8) || d == 12345.678f)) logits[srow[u]] = d;
        }
    }
}

}  // namespace

extern "C" int cpn_local_units(int mode, const float* loc8, const float* coords9, const float* w1, int ldw1, const float* b1,
                               const float* add, const uint16_t* w2, int ldw2, const float* b2, const uint16_t* wk2, int ldwk2,
                               const float* bk2, const float* w1b, int ldw1b, const float* b1b, const uint16_t* kh_u, int B,
                               int V, int R, int S, int ray0, int nrays, uint16_t* ce_u, const float* lv_u, float* logits,
                               void* stream) {
    CPN_REQUIRE(mode >= 0 && mode <= 2, CPN_E_ARG, "cpn_local_units: mode must be 0, 1 or 2 (got %d)", mode);
    CPN_REQUIRE(loc8 && coords9 && w1 && b1 && w2 && b2 && logits, CPN_E_ARG, "cpn_local_units: null pointer");
    CPN_REQUIRE(mode == 0 ? (wk2 && bk2 && kh_u) : mode == 1 ? (add && ce_u) : (add && wk2 && bk2 && w1b && b1b), CPN_E_ARG,
                "cpn_local_units: null pointer for mode %d", mode);
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && ldw1 >= 16 && ldw2 >= 128 && (ldw2 % 8) == 0 &&
                    (mode == 1 || (ldwk2 >= 128 && (ldwk2 % 8) == 0)) && (mode != 2 || ldw1b >= 16), CPN_E_SHAPE,
                "cpn_local_units: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_local_units: ray range outside B*R");
    CPN_REQUIRE(((uintptr_t)ce_u % 16) == 0 && ((uintptr_t)kh_u % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)wk2 % 16) == 0 &&
                    ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w1b % 16) == 0 && ((uintptr_t)lv_u % 16) == 0,
                CPN_E_ARG, "cpn_local_units: fp16 operands and first-layer weights must be 16-byte aligned");
    UnitGeo geo;
    geo.V = V; geo.R = R; geo.S = S; geo.ray0 = ray0; geo.nrays = nrays;
    geo.nsblk = (int)cpn_cdiv(S, TSW);
    geo.groups_per_b = (int)cpn_cdiv(R, TG);
    const int b_lo = ray0 / R, b_hi = (ray0 + nrays - 1) / R;
    geo.group0 = (long long)b_lo * geo.groups_per_b + (ray0 - b_lo * R) / TG;
    const long long group1 = (long long)b_hi * geo.groups_per_b + (ray0 + nrays - 1 - b_hi * R) / TG;
    geo.nunits = (group1 - geo.group0 + 1) * V * geo.nsblk;
    CPN_REQUIRE(geo.nunits * 16 < (1LL << 31), CPN_E_SHAPE, "cpn_local_units: chunk too large for 32-bit indexing");
    // lv_u covers the whole (B, R, S) problem in unit order (cpn_sample_geometry); this launch's units start at its first ray group
    const f32x4* lv_chunk = lv_u ? reinterpret_cast<const f32x4*>(lv_u) + (size_t)geo.group0 * V * geo.nsblk * 64 : nullptr;
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(geo.nunits, 8), mode == 1 ? 1024 : (CPN_LU_UNITS > 1 ? 256 : 512));
    if (mode == 0)
        hipLaunchKernelGGL((local_units_kernel<0, CPN_LU_UNITS>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, loc8, coords9, w1, ldw1, b1, add,
                           (const __half*)w2, ldw2, b2, (const __half*)wk2, ldwk2, bk2, w1b, ldw1b, b1b, (const __half*)kh_u, geo,
                           (__half*)ce_u, lv_chunk, logits);
    else if (mode == 1)
        hipLaunchKernelGGL((local_units_kernel<1, 1>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, loc8, coords9, w1, ldw1, b1, add,
                           (const __half*)w2, ldw2, b2, (const __half*)wk2, ldwk2, bk2, w1b, ldw1b, b1b, (const __half*)kh_u, geo,
                           (__half*)ce_u, lv_chunk, logits);
    else
        hipLaunchKernelGGL((local_units_kernel<2, CPN_LU_UNITS>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, loc8, coords9, w1, ldw1, b1, add,
                           (const __half*)w2, ldw2, b2, (const __half*)wk2, ldwk2, bk2, w1b, ldw1b, b1b, (const __half*)kh_u, geo,
                           (__half*)ce_u, lv_chunk, logits);
    CPN_LAUNCH_CHECK("cpn_local_units");
    return 0;
}
