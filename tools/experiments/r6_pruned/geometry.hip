// K1 / K1b — epipolar geometry of the render path, one thread per ray-view / per sample.
//
// COMPILE WITH -ffp-contract=off.  Every expression below is a sequence of IEEE-754
// add/sub/mul/div/sqrt in exactly the order of oracle/render_ref.py (which restates
// /root/reference models/epipolar.py:175-253, utils_training/geometry.py:98-162,236-245,313-324,374-393,
// utils_training/utils.py:99-108,242-245, models/CoPoNeRF.py:259-309,428-445), so that pixel_val,
// the secondary sample coordinates and therefore every bilinear tap index are bit-identical
// to the oracle's.  No FMA contraction, no fast-math, correctly rounded division and sqrt
// (hipcc default).  HBM-bound elementwise work: 36 B/ray in, 56 B/ray-view out (K1);
// 88 B/sample out (K1b) — the float64 island is ~150 flop/sample, far under the fp64 VALU roof.
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ bool in_bounds(float x, float y) {
    const float lo = -1e-6f, hi = 1.000001f;     // epipolar.py:28-35 (float32 of 1 + 1e-6)
    return (x >= lo) && (y >= lo) && (x <= hi) && (y <= hi);
}

struct Hit { float t, x, y; bool valid; };

// intersection of the projected ray with the frame line {x,y}[dim] == value (epipolar.py:74-122)
__device__ __forceinline__ Hit frame_hit(const float* Kn, const float o[3], const float d[3], int dim, float value) {
    const int other = 1 - dim;
    const float fs = Kn[dim * 3 + dim], fo = Kn[other * 3 + other];
    const float cs = Kn[dim * 3 + 2], co = Kn[other * 3 + 2];
    const float os = o[dim], oo = o[other], ds = d[dim], dd = d[other];
    const float oz = o[2], dz = d[2];
    const float c = (value - cs) / fs;
    const float t = (c * oz - os) / (ds - c * dz);
    const float num = fo * (oo * (c * dz - ds) + dd * (os - c * oz));
    const float den = dz * os - ds * oz;
    const float coord_other = co + num / den;
    Hit h;
    h.t = t;
    h.x = dim == 0 ? value : coord_other;
    h.y = dim == 0 ? coord_other : value;
    const float z = oz + t * dz;
    h.valid = in_bounds(h.x, h.y) && (z > -1e-6f);
    return h;
}

// first-index arg-min / arg-max over candidates, invalid t replaced by +-inf (epipolar.py:125-149)
__device__ __forceinline__ Hit pick(const Hit* c, bool take_max) {
    const float fill = take_max ? -INFINITY : INFINITY;
    Hit b = c[0];
    b.t = c[0].valid ? c[0].t : fill;
    for (int i = 1; i < 4; ++i) {
        const float ti = c[i].valid ? c[i].t : fill;
        const bool better = take_max ? (ti > b.t) : (ti < b.t);
        if (better) { b = c[i]; b.t = ti; }
    }
    return b;
}

__device__ __forceinline__ void project_point(const float* Kn, const float p[3], float& x, float& y) {
    const float q = p[2] + 1e-8f;                 // epipolar.py:23-26
    const float h0 = p[0] / q, h1 = p[1] / q, h2 = p[2] / q;
    x = (Kn[0] * h0 + Kn[1] * h1) + Kn[2] * h2;
    y = (Kn[3] * h0 + Kn[4] * h1) + Kn[5] * h2;
}

__device__ __forceinline__ float scrub(float v, float repl) { return isfinite(v) ? v : repl; }

__device__ __forceinline__ void normalize3(float v[3]) {
    float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    n = (n < 1e-12f) ? 1e-12f : n;                // clamp_min keeps NaN
    v[0] = v[0] / n; v[1] = v[1] / n; v[2] = v[2] / n;
}

__global__ void project_rays_kernel(const float* __restrict__ cam, const float* __restrict__ uv, long long uv_bstride,
                                    int B, int V, int R, float* __restrict__ coords9,
                                    float* __restrict__ seg, uint8_t* __restrict__ overlaps) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * V * R;
    if (idx >= total) return;
    const int r = (int)(idx % R);
    const int n = (int)(idx / R);
    const int b = n / V;
    const float* c = cam + (size_t)n * CPN_CAM_STRIDE;
    const float* T = c + CPN_CAM_TQ;
    const float fx = c[CPN_CAM_KQ], fy = c[CPN_CAM_KQ + 1], cx = c[CPN_CAM_KQ + 2], cy = c[CPN_CAM_KQ + 3];
    const float* Kn = c + CPN_CAM_KN;
    const float u = uv[(size_t)b * uv_bstride + (size_t)r * 2 + 0], v = uv[(size_t)b * uv_bstride + (size_t)r * 2 + 1];

    // Pluecker embedding of the query ray in this context frame (geometry.py:236-245)
    const float one = 1.0f;
    const float xl = (u - cx) / fx * one, yl = (v - cy) / fy * one;
    float o[3], d[3], m[3];
    for (int i = 0; i < 3; ++i) {
        const float w = ((T[i * 4 + 0] * xl + T[i * 4 + 1] * yl) + T[i * 4 + 2] * one) + T[i * 4 + 3] * one;
        o[i] = T[i * 4 + 3];
        d[i] = w - o[i];
    }
    normalize3(d);
    m[0] = o[1] * d[2] - o[2] * d[1];
    m[1] = o[2] * d[0] - o[0] * d[2];
    m[2] = o[0] * d[1] - o[1] * d[0];
    float* c9 = coords9 + (size_t)idx * 9;
    c9[0] = d[0]; c9[1] = d[1]; c9[2] = d[2];
    c9[3] = m[0]; c9[4] = m[1]; c9[5] = m[2];
    c9[6] = o[0]; c9[7] = o[1]; c9[8] = o[2];

    // clip the projection of o + t d, t in [0, inf), to the unit image square (epipolar.py:175-253)
    Hit cand[4] = {frame_hit(Kn, o, d, 0, 0.0f), frame_hit(Kn, o, d, 0, 1.0f),
                   frame_hit(Kn, o, d, 1, 0.0f), frame_hit(Kn, o, d, 1, 1.0f)};
    const Hit fmin = pick(cand, false), fmax = pick(cand, true);
    const bool depth_zero = o[2] < 1e-6f;
    const bool at_camera = sqrtf((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) < 1e-6f;
    float p0[3] = {at_camera ? d[0] : o[0], at_camera ? d[1] : o[1], at_camera ? d[2] : o[2]};
    float x0, y0, xi, yi;
    project_point(Kn, p0, x0, y0);
    const bool v0 = in_bounds(x0, y0) && (p0[2] > -1e-6f) && !(depth_zero && !at_camera);
    project_point(Kn, d, xi, yi);
    const bool vi = in_bounds(xi, yi) && (d[2] > -1e-6f);

    const float xmin = v0 ? x0 : fmin.x, ymin = v0 ? y0 : fmin.y;
    const float xmax = vi ? xi : fmax.x, ymax = vi ? yi : fmax.y;
    const bool okmin = v0 ? true : fmin.valid, okmax = vi ? true : fmax.valid;
    float* sg = seg + (size_t)idx * 4;
    sg[0] = scrub((xmin - 0.5f) * 2.0f, 0.0f);    // CoPoNeRF.py:279-285
    sg[1] = scrub((ymin - 0.5f) * 2.0f, 0.0f);
    sg[2] = scrub((xmax - 0.5f) * 2.0f, 0.0f);
    sg[3] = scrub((ymax - 0.5f) * 2.0f, 0.0f);
    overlaps[idx] = (okmin && okmax) ? 1 : 0;
}

__device__ __forceinline__ float nan_to_num0(float v) {          // torch.nan_to_num(x, 0)
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return v;
}

// UNITS = false: one thread per (image, ray, sample) in index order.  UNITS = true (lv_u given): a workgroup is 4 adjacent rays x
// 64 consecutive samples of one image = 16 whole units of the unit-order copy, which it stages in LDS and writes as one
// contiguous 16 KiB run (written from the index-order threads it was 64 sixteen-byte pieces at a 64-byte stride per store
// instruction: 0.14 ms per image on top of the kernel's 0.23); threads past R or S repeat their neighbour's row.
template <bool UNITS>
__global__ __launch_bounds__(256) void sample_geometry_kernel(const float* __restrict__ cam, const float* __restrict__ coords9,
                                       const float* __restrict__ seg, const float* __restrict__ interval,
                                       int N, int R, int S, int H, int W,
                                       float* __restrict__ pixel_val, float* __restrict__ pt_out,
                                       float* __restrict__ sec_grid, float* __restrict__ pe6,
                                       float* __restrict__ loc8, float* __restrict__ lv_u, int V) {
    __shared__ __attribute__((aligned(16))) f32x4 stage[UNITS ? 16 * 64 : 1];
    long long idx;
    int s, n, ur = 0, usl = 0, urg = 0, usc = 0;
    long long nr;
    bool ulive = true;
    if constexpr (UNITS) {
        const int nsc = (S + 63) >> 6, gpb = (R + 3) >> 2;
        usc = (int)(blockIdx.x % nsc);
        urg = (int)((blockIdx.x / nsc) % gpb);
        n = (int)(blockIdx.x / ((long long)nsc * gpb));
        ur = threadIdx.x >> 6;
        usl = threadIdx.x & 63;
        const int r_raw = urg * 4 + ur, s_raw = usc * 64 + usl;
        ulive = r_raw < R && s_raw < S;
        s = min(s_raw, S - 1);
        nr = (long long)n * R + min(r_raw, R - 1);
        idx = nr * S + s;
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[threadIdx.x + 256 * i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
    } else {
        idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        const long long total = (long long)N * R * S;
        if (idx >= total) return;
        s = (int)(idx % S);
        nr = idx / S;
        n = (int)(nr / R);
    }
    const float* c = cam + (size_t)n * CPN_CAM_STRIDE;
    const float* sg = seg + (size_t)nr * 4;
    const float* c9 = coords9 + (size_t)nr * 9;

    // sample on the segment (CoPoNeRF.py:304-307)
    const float iv = interval[s];
    const float pvx = sg[0] + (sg[2] - sg[0]) * iv;
    const float pvy = sg[1] + (sg[3] - sg[1]) * iv;
    pixel_val[idx * 2 + 0] = pvx;
    pixel_val[idx * 2 + 1] = pvy;

    // context-camera ray through the sample pixel (geometry.py:100-109)
    const float fx = c[CPN_CAM_KC], fy = c[CPN_CAM_KC + 1], cx = c[CPN_CAM_KC + 2], cy = c[CPN_CAM_KC + 3];
    const float px = (pvx + 1.0f) / 2.0f * (float)(W - 1);
    const float py = (pvy + 1.0f) / 2.0f * (float)(H - 1);
    const float one = 1.0f;
    const float xl = (px - cx) / fx * one, yl = (py - cy) / fy * one;
    const float* M = c + CPN_CAM_M;
    float oc[3], dc[3], mc[3];
    for (int i = 0; i < 3; ++i) {
        const float w = ((M[i * 4 + 0] * xl + M[i * 4 + 1] * yl) + M[i * 4 + 2] * one) + M[i * 4 + 3] * one;
        oc[i] = M[i * 4 + 3];
        dc[i] = w - oc[i];
    }
    normalize3(dc);
    mc[0] = oc[1] * dc[2] - oc[2] * dc[1];
    mc[1] = oc[2] * dc[0] - oc[0] * dc[2];
    mc[2] = oc[0] * dc[1] - oc[1] * dc[0];

    // closest point on the query line to the context line, float64 (geometry.py:132-162)
    const double l1[3] = {c9[0], c9[1], c9[2]}, m1[3] = {c9[3], c9[4], c9[5]};
    const double l2[3] = {dc[0], dc[1], dc[2]}, m2[3] = {mc[0], mc[1], mc[2]};
    double c12[3], l2c[3], mt[3];
    c12[0] = l1[1] * l2[2] - l1[2] * l2[1];
    c12[1] = l1[2] * l2[0] - l1[0] * l2[2];
    c12[2] = l1[0] * l2[1] - l1[1] * l2[0];
    l2c[0] = l2[1] * c12[2] - l2[2] * c12[1];
    l2c[1] = l2[2] * c12[0] - l2[0] * c12[2];
    l2c[2] = l2[0] * c12[1] - l2[1] * c12[0];
    mt[0] = m1[1] * l2c[2] - m1[2] * l2c[1];
    mt[1] = m1[2] * l2c[0] - m1[0] * l2c[2];
    mt[2] = m1[0] * l2c[1] - m1[1] * l2c[0];
    const double sdot = (m2[0] * c12[0] + m2[1] * c12[1]) + m2[2] * c12[2];
    const double nn = sqrt((c12[0] * c12[0] + c12[1] * c12[1]) + c12[2] * c12[2]);
    const double cd = nn * nn + 1e-12;
    float p[3];
    for (int i = 0; i < 3; ++i) {
        const double v = (-mt[i] + sdot * l1[i]) / cd;
        p[i] = (float)(isfinite(v) ? v : 0.0);                     // geometry.py:126-129
    }
    pt_out[idx * 3 + 0] = p[0]; pt_out[idx * 3 + 1] = p[1]; pt_out[idx * 3 + 2] = p[2];

    // the point in its own frame and in the other view's frame (utils.py:99-108)
    const float* Ao = c + CPN_CAM_AOWN;
    const float* At = c + CPN_CAM_AOTH;
    float own[3], oth[3];
    for (int i = 0; i < 3; ++i) {
        own[i] = ((p[0] * Ao[i * 4 + 0] + p[1] * Ao[i * 4 + 1]) + p[2] * Ao[i * 4 + 2]) + one * Ao[i * 4 + 3];
        oth[i] = ((p[0] * At[i * 4 + 0] + p[1] * At[i * 4 + 1]) + p[2] * At[i * 4 + 2]) + one * At[i * 4 + 3];
    }
    // reprojection into the other image (geometry.py:374-393, utils.py:242-245)
    const float fxo = c[CPN_CAM_KO], fyo = c[CPN_CAM_KO + 1], cxo = c[CPN_CAM_KO + 2], cyo = c[CPN_CAM_KO + 3];
    const float zz = oth[2] + 1e-12f;
    const float xp = scrub(fxo * oth[0] / zz + cxo, 1e10f);
    const float yp = scrub(fyo * oth[1] / zz + cyo, 1e10f);
    sec_grid[idx * 2 + 0] = (xp / (float)(W - 1)) * 2.0f - 1.0f;
    sec_grid[idx * 2 + 1] = (yp / (float)(H - 1)) * 2.0f - 1.0f;

    // point encodings (CoPoNeRF.py:375-394)
    float* pe = pe6 + idx * 6;
    for (int i = 0; i < 3; ++i) {
        pe[i] = tanhf(nan_to_num0(own[i]) / 5.0f);
        pe[3 + i] = tanhf(nan_to_num0(oth[i]) / 5.0f);
    }

    // per-sample part of local_coords (CoPoNeRF.py:411-445; geometry.py:313-324)
    float cr[3] = {xl, yl, one};
    normalize3(cr);
    const float dx = p[0] - c9[6], dy = p[1] - c9[7], dz = p[2] - c9[8];
    const float depth = scrub(sqrtf((dx * dx + dy * dy) + dz * dz), 1000000.0f);
    float* lc = loc8 + idx * 8;
    lc[0] = cr[0]; lc[1] = cr[1]; lc[2] = cr[2];
    lc[3] = tanhf(depth);
    lc[4] = tanhf(depth / 10.0f);
    lc[5] = tanhf(depth / 100.0f);
    lc[6] = tanhf(depth / 1000.0f);
    lc[7] = 0.0f;
    if constexpr (UNITS) {
        // the same 16 inputs (local_coords, CoPoNeRF.py:411-445) once more, in the UNIT order cpn_local_units multiplies in
        // (include/coponerf_hip.h): lane c + 16 fg of unit ((b * ceil(R/4) + r/4) * V + v) * ceil(S/4) + s/4, c = (s & 3) * 4 +
        // (r & 3), holds K entries 4 fg .. 4 fg + 3 - a wave of that kernel then reads its unit's inputs as ONE 1 KiB line
        // instead of five scattered 4 - 16 byte accesses per lane.  Slots of rays >= R / samples >= S are written as zeros.
        if (ulive) {
            f32x4* dst = stage + (usl >> 2) * 64 + ((usl & 3) * 4 + ur);
            dst[0] = f32x4{lc[0], lc[1], lc[2], 1.0f};        // (the 1.0 multiplies the first layer's bias)
            dst[16] = f32x4{0.0f, 0.0f, c9[0], c9[1]};
            dst[32] = f32x4{c9[2], lc[3], lc[4], lc[5]};
            dst[48] = f32x4{lc[6], c9[6], c9[7], c9[8]};
        }
        __syncthreads();
        const int b = n / V, v = n - b * V, nsblk = (S + 3) >> 2;
        const long long unit0 = ((((long long)b * ((R + 3) >> 2) + urg) * V + v) * nsblk) + usc * 16;
        const int nlanes = min(16, nsblk - usc * 16) * 64;
        f32x4* out = reinterpret_cast<f32x4*>(lv_u) + unit0 * 64;
        for (int i = threadIdx.x; i < nlanes; i += 256) out[i] = stage[i];
    }
}

__global__ void mask_rgb_kernel(const float* __restrict__ rgb_raw, int ld, const uint8_t* __restrict__ overlaps,
                                int B, int V, int R, float* __restrict__ rgb, float* __restrict__ valid) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * R) return;
    const int r = (int)(idx % R), b = (int)(idx / R);
    bool any = false;
    for (int v = 0; v < V; ++v) any = any || overlaps[((size_t)b * V + v) * R + r];
    const float vm = any ? 1.0f : 0.0f;
    valid[idx] = vm;
    for (int ch = 0; ch < 3; ++ch)                               // CoPoNeRF.py:563
        rgb[idx * 3 + ch] = rgb_raw[idx * ld + ch] * vm + 1.0f * (1.0f - vm);
}

}  // namespace

extern "C" int cpn_project_rays(const float* cam, const float* uv, long long uv_batch_stride, int B, int V, int R,
                                float* coords9, float* seg, uint8_t* overlaps, void* stream) {
    CPN_REQUIRE(cam && uv && coords9 && seg && overlaps, CPN_E_ARG, "cpn_project_rays: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0, CPN_E_SHAPE, "cpn_project_rays: need B>0, V==2, R>0 (got %d,%d,%d)", B, V, R);
    CPN_REQUIRE(uv_batch_stride >= 2LL * R, CPN_E_SHAPE, "cpn_project_rays: uv_batch_stride %lld < 2R", uv_batch_stride);
    const long long total = (long long)B * V * R;
    hipLaunchKernelGGL(project_rays_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       cam, uv, uv_batch_stride, B, V, R, coords9, seg, overlaps);
    CPN_LAUNCH_CHECK("cpn_project_rays");
    return 0;
}

extern "C" int cpn_sample_geometry(const float* cam, const float* coords9, const float* seg, const float* interval,
                                   int B, int V, int R, int S, int H, int W, float* pixel_val, float* pt,
                                   float* sec_grid, float* pe6, float* loc8, float* lv_u, void* stream) {
    CPN_REQUIRE(cam && coords9 && seg && interval && pixel_val && pt && sec_grid && pe6 && loc8, CPN_E_ARG,
                "cpn_sample_geometry: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H > 1 && W > 1, CPN_E_SHAPE, "cpn_sample_geometry: bad shape");
    CPN_REQUIRE(((uintptr_t)lv_u % 16) == 0, CPN_E_ARG, "cpn_sample_geometry: lv_u must be 16-byte aligned");
    const long long total = (long long)B * V * R * S;
    if (lv_u) {
        const long long blocks = (long long)B * V * cpn_cdiv(R, 4) * cpn_cdiv(S, 64);
        CPN_REQUIRE(blocks < (1LL << 31), CPN_E_SHAPE, "cpn_sample_geometry: too many workgroups");
        hipLaunchKernelGGL(sample_geometry_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           cam, coords9, seg, interval, B * V, R, S, H, W, pixel_val, pt, sec_grid, pe6, loc8, lv_u, V);
    } else {
        hipLaunchKernelGGL(sample_geometry_kernel<false>, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           cam, coords9, seg, interval, B * V, R, S, H, W, pixel_val, pt, sec_grid, pe6, loc8, lv_u, V);
    }
    CPN_LAUNCH_CHECK("cpn_sample_geometry");
    return 0;
}

extern "C" int cpn_mask_rgb(const float* rgb_raw, int ld, const uint8_t* overlaps, int B, int V, int R,
                            float* rgb, float* valid, void* stream) {
    CPN_REQUIRE(rgb_raw && overlaps && rgb && valid, CPN_E_ARG, "cpn_mask_rgb: null pointer");
    CPN_REQUIRE(B > 0 && V > 0 && R > 0 && ld >= 3, CPN_E_SHAPE, "cpn_mask_rgb: bad shape");
    hipLaunchKernelGGL(mask_rgb_kernel, dim3(cpn_cdiv((long long)B * R, 256)), dim3(256), 0, (hipStream_t)stream,
                       rgb_raw, ld, overlaps, B, V, R, rgb, valid);
    CPN_LAUNCH_CHECK("cpn_mask_rgb");
    return 0;
}
