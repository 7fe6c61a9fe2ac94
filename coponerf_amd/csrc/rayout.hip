// Per-RAY tail of CoPoNeRF.forward in two launches (round 3): the light-field decoder with the output masking, and the
// auxiliary per-ray outputs.  A full-image render is 18 forward() calls of <= 3 641 rays in the reference's callers
// (/root/reference test.py:176-190), where the 11 + 1 launches of the layer-by-layer decoder and the ~30 ATen launches
// of the auxiliary outputs cost more host time than the call's per-sample kernels take on the GPU.
//
//   cpn_lightfield_decode  replaces lightfield.ResnetFC.forward (/root/reference models/lightfield.py:131-167; block:
//                          :52-61) on [coords(18) | z_local ; z_local] (models/CoPoNeRF.py:547-560) and the white
//                          background for rays without overlap (:562-566)
//   cpn_ray_outputs        replaces models/CoPoNeRF.py:493-541 (argmax sample, attention-weighted expected point, its
//                          depth in the query camera, reprojections into both context views, flow / cycle-mask lookups:
//                          utils_training/utils.py:52-69,140-170,260-276, geometry.py:395-406)
#include "common.h"

namespace {

// ---- packed decoder weights (floats), built once per parameter version by the host (render.py) ----------------
constexpr int OFF_IN_W = 0;                          // lin_in   [128][32]   (18 columns used)
constexpr int OFF_IN_B = OFF_IN_W + 128 * 32;        //          [128]
constexpr int OFF_BLK = OFF_IN_B + 128;
constexpr int BLK_ZW = 0;                            // lin_z[k] [128][416]  (the two 416-column halves summed)
constexpr int BLK_ZB = BLK_ZW + 128 * 416;
constexpr int BLK_0W = BLK_ZB + 128;                 // fc_0     [128][128]
constexpr int BLK_0B = BLK_0W + 128 * 128;
constexpr int BLK_1W = BLK_0B + 128;                 // fc_1     [128][128]
constexpr int BLK_1B = BLK_1W + 128 * 128;
constexpr int BLK_SZ = BLK_1B + 128;
constexpr int OFF_OUT_W = OFF_BLK + 3 * BLK_SZ;      // lin_out  [16][128]   (rows 3..15 zero)
constexpr int OFF_OUT_B = OFF_OUT_W + 16 * 128;      //          [16]
constexpr int PACK_SZ = OFF_OUT_B + 16;
static_assert(PACK_SZ == CPN_LIGHTFIELD_PACK_FLOATS, "decoder weight pack layout and header disagree");

constexpr int XS_LD = 132;                           // LDS row stride (floats) of the 16 x 128 layer state

// acc[t] = sum_k W[n_base + 16 t + fi][k] * x[k]   on v_mfma_f32_16x16x4_f32, operands swapped (A = weights, B = rays)
// exactly as cpn_linear_f32 orders them: k block kb covers k = 16 kb + 4 fg + e, one MFMA step per e.
// W is stored in FRAGMENT order (round 4; the host packs it, include/coponerf_hip.h): the 1 KiB of A operand (tile, k block)
// contiguous, lane l's four floats at l * 16 bytes - a load instruction covers 8 whole lines.  Row-major weights read in the
// fragment layout (lane = row + 16 * k group: 16 rows x 16 bytes) cost 64 L1 tag look-ups per instruction, and the 272
// weight loads of a wave made the decoder wait on the tag pipe, not on the MFMA (552 us per 65 536 rays at 42 % of the fp32
// MFMA rate; tools/fewrows_bench.py found the same limit in the few-row GEMM).
template <int NT, class LoadX>
__device__ __forceinline__ void mm_f32(LoadX load_x, const float* __restrict__ W, int nkb, int n_base, int lane, f32x4 (&acc)[NT]) {
    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        wp[t] = W + ((size_t)((n_base >> 4) + t) * nkb * 64 + lane) * 4;
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 xa = load_x(0), wa[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wa[t] = *reinterpret_cast<const f32x4*>(wp[t]);
    for (int kb = 0; kb < nkb; ++kb) {
        f32x4 xn = xa, wn[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) wn[t] = wa[t];
        if (kb + 1 < nkb) {                                        // next block's operands in flight under the MFMAs
            xn = load_x(kb + 1);
#pragma unroll
            for (int t = 0; t < NT; ++t) wn[t] = *reinterpret_cast<const f32x4*>(wp[t] + (kb + 1) * 256);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][e], xa[e], acc[t], 0, 0, 0);
        xa = xn;
#pragma unroll
        for (int t = 0; t < NT; ++t) wa[t] = wn[t];
    }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
    return v;
}

// One workgroup = 16 rays, 4 waves; wave w owns output columns [32 w, 32 w + 32) of every 128-wide layer.  The layer
// state crosses waves through LDS (16 x 128 fp32), the weights come from L2 in the fragment layout.  The arithmetic
// (operation order included) is that of the layer-by-layer cpn_linear_f32 chain.
__global__ __launch_bounds__(256) void lightfield_decode_kernel(const float* __restrict__ coords9,
                                                                const float* __restrict__ zl,
                                                                const float* __restrict__ wp,
                                                                const uint8_t* __restrict__ overlaps, int B, int R,
                                                                float* __restrict__ rgb, float* __restrict__ valid,
                                                                float* __restrict__ rgb_raw) {
    __shared__ __attribute__((aligned(16))) float xs[16 * XS_LD];
    __shared__ __attribute__((aligned(16))) float ns[16 * XS_LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const long long nray = (long long)B * R;
    long long ray = (long long)blockIdx.x * 16 + fi;
    const bool live = ray < nray;
    ray = live ? ray : nray - 1;
    const int b = (int)(ray / R), r = (int)(ray - (long long)b * R);
    const int n_base = wave * 32;
    auto bias4 = [&](int off, int t) { return *reinterpret_cast<const f32x4*>(wp + off + n_base + 16 * t + fg * 4); };
    auto from_lds = [&](const float* buf) {
        return [=](int kb) { return relu4(*reinterpret_cast<const f32x4*>(buf + fi * XS_LD + kb * 16 + fg * 4)); };
    };
    auto to_lds = [&](float* buf, const f32x4 (&v)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(buf + fi * XS_LD + n_base + 16 * t + fg * 4) = v[t];
    };

    f32x4 x[2], acc[2];
    // lin_in on the 18 ray coordinates of the two views (CoPoNeRF.py:547-549: coords of view 0 | view 1), K padded to 32
    mm_f32<2>([&](int kb) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kb * 16 + fg * 4 + e;
            const int vw = k >= 9 ? 1 : 0;
            v[e] = k < 18 ? coords9[(((size_t)b * 2 + vw) * R + r) * 9 + (k - 9 * vw)] : 0.0f;
        }
        return v;
    }, wp + OFF_IN_W, 2, n_base, lane, acc);
#pragma unroll
    for (int t = 0; t < 2; ++t) x[t] = acc[t] + bias4(OFF_IN_B, t);

    // z rows in the LOAD layout (lane = 4 * ray + 16-byte piece: 4 adjacent lanes read 64 contiguous bytes of one row, 16 tag
    // look-ups per instruction instead of 64), moved to the B-operand layout (lane = ray + 16 * k group) with 4 ds_bpermute
    long long ray_l = (long long)blockIdx.x * 16 + (lane >> 2);
    ray_l = ray_l < nray ? ray_l : nray - 1;
    const float* zrow = zl + (size_t)ray_l * 416 + (lane & 3) * 4;
    const int to_b = (4 * fi + fg) * 4;
    auto load_z = [&](int kb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zrow + kb * 16);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(to_b, __float_as_int(v[e])));
        return o;
    };
    for (int k = 0; k < 3; ++k) {
        const float* blk = wp + OFF_BLK + (size_t)k * BLK_SZ;
        // x = x + lin_z[k](z)                                                           (lightfield.py:150-156)
        mm_f32<2>(load_z, blk + BLK_ZW, 26, n_base, lane, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = acc[t];
            v += bias4(OFF_BLK + k * BLK_SZ + BLK_ZB, t);
            v += x[t];
            x[t] = v;
        }
        to_lds(xs, x);
        __syncthreads();
        // net = fc_0(relu(x));  x = x + fc_1(relu(net))                                  (lightfield.py:52-61)
        mm_f32<2>(from_lds(xs), blk + BLK_0W, 8, n_base, lane, acc);
        f32x4 net[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) net[t] = acc[t] + bias4(OFF_BLK + k * BLK_SZ + BLK_0B, t);
        to_lds(ns, net);
        __syncthreads();
        mm_f32<2>(from_lds(ns), blk + BLK_1W, 8, n_base, lane, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 v = acc[t];
            v += bias4(OFF_BLK + k * BLK_SZ + BLK_1B, t);
            v += x[t];
            x[t] = v;
        }
    }
    to_lds(xs, x);                                    // every wave is past its last read of xs (barrier after ns)
    __syncthreads();
    if (wave != 0) return;
    // lin_out(relu(x)) -> 3 channels, then white where no context view sees the ray (CoPoNeRF.py:562-566)
    f32x4 o[1];
    mm_f32<1>(from_lds(xs), wp + OFF_OUT_W, 8, 0, lane, o);
    if (fg != 0 || !live) return;
    f32x4 raw = o[0] + *reinterpret_cast<const f32x4*>(wp + OFF_OUT_B);
    const bool any = overlaps[((size_t)b * 2 + 0) * R + r] || overlaps[((size_t)b * 2 + 1) * R + r];
    const float vm = any ? 1.0f : 0.0f;
    valid[ray] = vm;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) rgb[ray * 3 + ch] = raw[ch] * vm + 1.0f * (1.0f - vm);
    if (rgb_raw) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rgb_raw[ray * 3 + ch] = raw[ch];
    }
}

// ---- auxiliary per-ray outputs --------------------------------------------------------------------------------
// per-pair constants (CPN_RAYC_STRIDE floats), built on the host next to the camera block
//   0..3   row 2 of inv(query cam2world)          4..12  inv(K_query[:3,:3])     13..21 K_ctx0[:3,:3]   22..30 K_ctx1[:3,:3]
//   31..46 Tq[:,0] (query camera -> frame of context view 0)                     47..62 Tq[:,1]
__device__ __forceinline__ void reproject(float u, float v, float depth, const float* __restrict__ Ki,
                                          const float* __restrict__ Kj, const float* __restrict__ T, float* out2) {
    // utils.py:140-170: to_homogeneous, K_i^-1, scale by depth, rigid transform, from_homogeneous (w + 1e-6), K_j
    float p[3], q[4], rr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) p[i] = (u * Ki[3 * i] + v * Ki[3 * i + 1] + Ki[3 * i + 2]) * depth;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = p[0] * T[4 * i] + p[1] * T[4 * i + 1] + p[2] * T[4 * i + 2] + T[4 * i + 3];
    const float qw = q[3] + 1e-6f;
    const float c0 = q[0] / qw, c1 = q[1] / qw, c2 = q[2] / qw;
#pragma unroll
    for (int i = 0; i < 3; ++i) rr[i] = c0 * Kj[3 * i] + c1 * Kj[3 * i + 1] + c2 * Kj[3 * i + 2];
    const float rw = rr[2] + 1e-6f;
    out2[0] = rr[0] / rw;
    out2[1] = rr[1] / rw;
}

// float -> int64 the way Tensor.long() does it on the device (truncation; NaN / out-of-range saturate like the
// hardware conversion), then the 0..255 tests and clamps of utils.py:52-69,260-276
__device__ __forceinline__ long long trunc_ll(float x) { return (long long)x; }

// one wave per query ray: lanes stride the samples of both views
__global__ __launch_bounds__(256) void ray_outputs_kernel(const float* __restrict__ at_wt, const float* __restrict__ pt,
                                                          const float* __restrict__ uv, long long uv_bstride,
                                                          const float* __restrict__ rayc,
                                                          const uint8_t* __restrict__ mask2,
                                                          const float* __restrict__ flow_up, int B, int R, int S,
                                                          long long* __restrict__ at_max, float* __restrict__ depth_ray,
                                                          float* __restrict__ t1, float* __restrict__ t2,
                                                          uint8_t* __restrict__ mask_c2, uint8_t* __restrict__ match,
                                                          float* __restrict__ c2_to_c1) {
    const int lane = threadIdx.x & 63;
    const long long ray = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= (long long)B * R) return;
    const int b = (int)(ray / R), r = (int)(ray - (long long)b * R);
    float ex[3] = {0.f, 0.f, 0.f};
    for (int v = 0; v < 2; ++v) {
        const size_t base = (((size_t)b * 2 + v) * R + r) * S;
        float best = -INFINITY, sx = 0.f, sy = 0.f, sz = 0.f;
        int bi = 0x7fffffff;
        for (int s = lane; s < S; s += 64) {
            const float w = at_wt[base + s];
            // argmax: first index of the maximum; a NaN counts as the maximum (torch.argmax)
            const bool better = (w > best) || (w != w && best == best);
            if (better || bi == 0x7fffffff) { best = w; bi = s; }
            const float* p = pt + (base + s) * 3;
            sx += w * fminf(fmaxf(p[0], -100.f), 100.f);
            sy += w * fminf(fmaxf(p[1], -100.f), 100.f);
            sz += w * fminf(fmaxf(p[2], -100.f), 100.f);
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const float ob = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            const bool onan = ob != ob, mnan = best != best;
            const bool take = (onan && !mnan) || (onan == mnan && (ob > best || (ob == best && oi < bi))) ||
                              (onan && mnan && oi < bi);
            if (take) { best = ob; bi = oi; }
            sx += __shfl_xor(sx, off);
            sy += __shfl_xor(sy, off);
            sz += __shfl_xor(sz, off);
        }
        if (lane == 0) at_max[((size_t)b * 2 + v) * R + r] = bi;
        ex[0] += sx;
        ex[1] += sy;
        ex[2] += sz;
    }
    if (lane != 0) return;
    const float* c = rayc + (size_t)b * CPN_RAYC_STRIDE;
    const float depth = c[0] * ex[0] + c[1] * ex[1] + c[2] * ex[2] + c[3];         // geometry.py:395-406, z row only
    const float u = uv[(size_t)b * uv_bstride + (size_t)r * 2], vv = uv[(size_t)b * uv_bstride + (size_t)r * 2 + 1];
    float a[2], d[2];
    reproject(u, vv, depth, c + 4, c + 13, c + 31, a);
    reproject(u, vv, depth, c + 4, c + 22, c + 47, d);
    t1[ray * 2] = a[0];
    t1[ray * 2 + 1] = a[1];
    t2[ray * 2] = d[0];
    t2[ray * 2 + 1] = d[1];
    depth_ray[ray] = fminf(fmaxf(depth, 0.f), 10.f);                               // CoPoNeRF.py:529 (NaN stays NaN below)
    if (depth != depth) depth_ray[ray] = depth;
    const long long lx = trunc_ll(d[0]), ly = trunc_ll(d[1]);
    mask_c2[ray] = (lx >= 0 && lx < 256 && ly >= 0 && ly < 256) ? 1 : 0;
    const int kx = (int)(lx < 0 ? 0 : (lx > 255 ? 255 : lx)), ky = (int)(ly < 0 ? 0 : (ly > 255 ? 255 : ly));
    match[ray] = mask2[((size_t)b * 256 + ky) * 256 + kx];
    const float* fu = flow_up + (size_t)b * 2 * 65536;
    c2_to_c1[ray * 2] = (float)kx + fu[ky * 256 + kx];
    c2_to_c1[ray * 2 + 1] = (float)ky + fu[65536 + ky * 256 + kx];
}

}  // namespace

extern "C" int cpn_lightfield_decode(const float* coords9, const float* z_local, const float* wpack,
                                     const uint8_t* overlaps, int B, int V, int R, float* rgb, float* valid,
                                     float* rgb_raw, void* stream) {
    CPN_REQUIRE(coords9 && z_local && wpack && overlaps && rgb && valid, CPN_E_ARG, "cpn_lightfield_decode: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0, CPN_E_SHAPE, "cpn_lightfield_decode: need B>0, V==2, R>0 (got %d,%d,%d)", B, V, R);
    CPN_REQUIRE(((uintptr_t)z_local % 16) == 0 && ((uintptr_t)wpack % 16) == 0, CPN_E_ARG,
                "cpn_lightfield_decode: z_local / wpack must be 16-B aligned");
    hipLaunchKernelGGL(lightfield_decode_kernel, dim3(cpn_cdiv((long long)B * R, 16)), dim3(256), 0, (hipStream_t)stream,
                       coords9, z_local, wpack, overlaps, B, R, rgb, valid, rgb_raw);
    CPN_LAUNCH_CHECK("cpn_lightfield_decode");
    return 0;
}

extern "C" int cpn_ray_outputs(const float* at_wt, const float* pt, const float* uv, long long uv_batch_stride,
                               const float* rayc, const uint8_t* mask2, const float* flow_up, int B, int V, int R, int S,
                               long long* at_wt_max, float* depth_ray, float* t_to_c1, float* t_to_c2, uint8_t* mask_c2,
                               uint8_t* match_mask, float* c2_to_c1, void* stream) {
    CPN_REQUIRE(at_wt && pt && uv && rayc && mask2 && flow_up && at_wt_max && depth_ray && t_to_c1 && t_to_c2 && mask_c2 &&
                    match_mask && c2_to_c1, CPN_E_ARG, "cpn_ray_outputs: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && uv_batch_stride >= 2LL * R, CPN_E_SHAPE,
                "cpn_ray_outputs: need B>0, V==2, R>0, S>0, uv_batch_stride >= 2R (got %d,%d,%d,%d,%lld)", B, V, R, S,
                uv_batch_stride);
    hipLaunchKernelGGL(ray_outputs_kernel, dim3(cpn_cdiv((long long)B * R, 4)), dim3(256), 0, (hipStream_t)stream, at_wt, pt,
                       uv, uv_batch_stride, rayc, mask2, flow_up, B, R, S, at_wt_max, depth_ray, t_to_c1, t_to_c2, mask_c2,
                       match_mask, c2_to_c1);
    CPN_LAUNCH_CHECK("cpn_ray_outputs");
    return 0;
}
