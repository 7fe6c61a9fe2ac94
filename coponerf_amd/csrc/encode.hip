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
