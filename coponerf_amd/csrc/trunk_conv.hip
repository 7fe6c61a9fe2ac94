// f3 (inference): the deep half of the ResNet-34 trunk — 3x3 / 1x1 convolutions of layer3 (256 channels at H/8) and layer4
// (512 at H/16) with their BatchNorm, residual and ReLU (/root/reference models/backbone.py:10-102; torchvision BasicBlock).
//
// At one stereo pair these layers are 2.4 GFLOP each over 2 048 (512) output positions: the library's fp32 Winograd kernels
// find 128 (64) tiles of work on a 256-CU chip and take 50 (90) us per layer + one more launch for the normalisation.  Here
// the contraction index (tap, input channel) is split over waves AND workgroups:
//
//   layout     activations NHWC fp32 (N, H, W, C) between these layers, weights packed [tap][ci][co] (cpn_pack_conv_weight)
//   workgroup  8 waves on ONE tile of 64 output channels x 64 output positions and one slab of the (tap, 16-channel block)
//              iterations; a wave takes a contiguous run of the slab's iterations
//   iteration  4 x 16-byte loads of the weights (rows ci0 + 4 kg + e, 64 consecutive co) and 4 of the activations (16
//              consecutive ci of 4 x 16 positions, zero outside the image) feed 64 v_mfma_f32_16x16x4_f32: element e of an
//              activation vector and weight row e form one k4 step (a fixed permutation of ci shared by both operands);
//              exact fp32 products, fp32 accumulation, no LDS in the loop
//   reduction  the 8 waves' accumulators are summed through LDS in a fixed order, the slab's partial tile goes to scratch, and
//              the epilogue kernel sums the slabs in slab order and applies batch norm (the reference's expression
//              (x - mean) / sqrt(var + eps) * w + b), residual and ReLU — deterministic, one launch per layer more than the
//              convolution itself (the library path: convolution + cpn_bn_act)
#include "common.h"

namespace {

constexpr int TC_WAVES = 8;

struct ConvGeo {
    int N, Hin, Win, Cin, Hout, Wout, Cout, ksize, stride, pad;
};

__global__ __launch_bounds__(64 * TC_WAVES, 1) void trunk_conv_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                      ConvGeo g, int ntc, int ntp, int iters_total,
                                                                      int iters_per_slab, float* __restrict__ part) {
    __shared__ float red[2 * 64 * 64];                                            // 32 KiB: two waves' accumulators
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles = ntc * ntp;
    const int slab = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int co0 = (tile % ntc) * 64, p0 = (tile / ntc) * 64;
    const int P = g.N * g.Hout * g.Wout;
    const int cblocks = g.Cin / 16;

    // this lane's four output positions (one per position group)
    int pn[4], py[4], px[4];
    bool pok[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int p = p0 + 16 * pg + fi;
        pok[pg] = p < P;
        const int pc = pok[pg] ? p : 0;
        pn[pg] = pc / (g.Hout * g.Wout);
        const int rem = pc - pn[pg] * g.Hout * g.Wout;
        py[pg] = rem / g.Wout;
        px[pg] = rem - py[pg] * g.Wout;
    }
    const bool cok = co0 + 4 * fi < g.Cout;
    const float* wcol = wp + (cok ? co0 + 4 * fi : 0);

    f32x4 acc[4][4];                                                              // [co sub-tile ea][position group]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // contiguous run of iterations for this wave
    const int s_begin = slab * iters_per_slab;
    const int s_end = s_begin + iters_per_slab < iters_total ? s_begin + iters_per_slab : iters_total;
    const int per_wave = (s_end - s_begin + TC_WAVES - 1) / TC_WAVES;
    const int it0 = s_begin + wave * per_wave;
    const int it1 = it0 + per_wave < s_end ? it0 + per_wave : s_end;

    int cur_tap = -1;
    long long xoff[4];
    bool xok[4];
    auto set_tap = [&](int tap) {
        const int ky = tap / g.ksize, kx = tap - ky * g.ksize;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            const int iy = py[pg] * g.stride + ky - g.pad, ix = px[pg] * g.stride + kx - g.pad;
            xok[pg] = pok[pg] && iy >= 0 && iy < g.Hin && ix >= 0 && ix < g.Win;
            xoff[pg] = xok[pg] ? (((long long)pn[pg] * g.Hin + iy) * g.Win + ix) * g.Cin + 4 * fg : 4 * fg;
        }
        cur_tap = tap;
    };
    // The zeroing of taps outside the image happens where the operands are USED (a select behind the load would put the
    // load's latency on the loop's critical path); lanes outside the tile's channels / positions compute values nobody stores.
    f32x4 wa[2][4], xb[2][4];
    unsigned okm[2];
    auto load = [&](int it, f32x4 (&a)[4], f32x4 (&b)[4], unsigned& ok) {
        const int tap = it / cblocks, ci0 = (it - tap * cblocks) * 16;
        if (tap != cur_tap) set_tap(tap);
        const float* wr = wcol + ((size_t)tap * g.Cin + ci0 + 4 * fg) * g.Cout;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = *reinterpret_cast<const f32x4*>(wr + (size_t)e * g.Cout);
        ok = 0;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            b[pg] = *reinterpret_cast<const f32x4*>(x + xoff[pg] + ci0);
            ok |= (xok[pg] ? 1u : 0u) << pg;
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mac = [&](const f32x4 (&a)[4], f32x4 (&b)[4], unsigned ok) {
#pragma unroll
        for (int pg = 0; pg < 4; ++pg)
            if (!((ok >> pg) & 1)) b[pg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ea = 0; ea < 4; ++ea)
#pragma unroll
                for (int pg = 0; pg < 4; ++pg)
                    acc[ea][pg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e][ea], b[pg][e], acc[ea][pg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (it0 < it1) {
        load(it0, wa[0], xb[0], okm[0]);
        for (int it = it0; it < it1; it += 2) {
            if (it + 1 < it1) load(it + 1, wa[1], xb[1], okm[1]);
            mac(wa[0], xb[0], okm[0]);
            if (it + 1 < it1) {
                if (it + 2 < it1) load(it + 2, wa[0], xb[0], okm[0]);
                mac(wa[1], xb[1], okm[1]);
            }
        }
    }

    // fixed-order tree over the waves: 4,5 -> 0,1;  6,7 -> 2,3;  2,3 -> 0,1;  1 -> 0
    float regs[64];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) regs[(a * 4 + b) * 4 + r] = acc[a][b][r];
    constexpr int PH[4][3] = {{4, 0, 2}, {6, 2, 2}, {2, 0, 2}, {1, 0, 1}};
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int src0 = PH[ph][0], dst0 = PH[ph][1], n = PH[ph][2];
        if (wave >= src0 && wave < src0 + n) {
            float* dst = red + (size_t)(wave - src0) * 64 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 64; ++q) dst[q * 64] = regs[q];
        }
        __syncthreads();
        if (wave >= dst0 && wave < dst0 + n) {
            const float* src = red + (size_t)(wave - dst0) * 64 * 64 + lane;
#pragma unroll
            for (int q = 0; q < 64; ++q) regs[q] += src[q * 64];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // D of (ea, pg), register r: channel co0 + 4 * (4 fg + r) + ea, position p0 + 16 pg + fi -> 4 consecutive channels per store
    float* dst = part + (size_t)slab * P * g.Cout;
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int p = p0 + 16 * pg + fi;
        if (p >= P) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * fg + 4 * r;
            if (co < g.Cout)
                *reinterpret_cast<f32x4*>(dst + (size_t)p * g.Cout + co) =
                    f32x4{regs[(0 * 4 + pg) * 4 + r], regs[(1 * 4 + pg) * 4 + r], regs[(2 * 4 + pg) * 4 + r],
                          regs[(3 * 4 + pg) * 4 + r]};
        }
    }
}

// y = act((sum_slab part - mean) / sqrt(var + eps) * w + b + res): NHWC out, and NCHW as well where the map leaves the trunk
__global__ void trunk_conv_epilogue_kernel(const float* __restrict__ part, int nslab, int P, int C, int HW,
                                           const float* __restrict__ mean, const float* __restrict__ var,
                                           const float* __restrict__ w, const float* __restrict__ b, float eps,
                                           const float* __restrict__ res, int relu, float* __restrict__ out_nhwc,
                                           float* __restrict__ out_nchw) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = (long long)P * C / 4;
    if (t >= n4) return;
    const int c = (int)((t * 4) % C);
    const long long p = (t * 4) / C;
    f32x4 s = reinterpret_cast<const f32x4*>(part)[t];
    for (int k = 1; k < nslab; ++k) s += reinterpret_cast<const f32x4*>(part)[t + (long long)k * n4];
    const f32x4 m = *reinterpret_cast<const f32x4*>(mean + c), v = *reinterpret_cast<const f32x4*>(var + c);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c), bb = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = (s[e] - m[e]) / sqrtf(v[e] + eps) * ww[e] + bb[e];
    if (res) y += reinterpret_cast<const f32x4*>(res)[t];
    if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], 0.0f);
    }
    if (out_nhwc) reinterpret_cast<f32x4*>(out_nhwc)[t] = y;
    if (out_nchw) {
        const long long n = p / HW, hw = p - n * HW;
#pragma unroll
        for (int e = 0; e < 4; ++e) out_nchw[((n * C) + c + e) * HW + hw] = y[e];
    }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, float* __restrict__ wp) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;           // index into wp [tap][ci][co]
    if (t >= (long long)taps * Cin * Cout) return;
    const int co = (int)(t % Cout);
    const int ci = (int)((t / Cout) % Cin);
    const int tap = (int)(t / ((long long)Cout * Cin));
    wp[t] = w[((size_t)co * Cin + ci) * taps + tap];
}

static inline int tc_slabs(int tiles, int iters) {
    int s = (256 + tiles - 1) / tiles;                        // one workgroup per CU
    const int cap = iters / (TC_WAVES * 4);                   // at least 4 iterations per wave
    s = s > cap ? cap : s;
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int cpn_pack_conv_weight(const float* w, int Cout, int Cin, int ksize, float* wp, void* stream) {
    CPN_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), 1, "cpn_pack_conv_weight: bad arguments");
    const long long n = (long long)ksize * ksize * Cin * Cout;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(cpn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin,
                       ksize * ksize, wp);
    CPN_LAUNCH_CHECK("cpn_pack_conv_weight");
    return 0;
}

extern "C" long long cpn_trunk_conv_scratch_floats(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride) {
    const int pad = ksize / 2;
    const int Hout = (Hin + 2 * pad - ksize) / stride + 1, Wout = (Win + 2 * pad - ksize) / stride + 1;
    const long long P = (long long)N * Hout * Wout;
    const int tiles = (int)(cpn_cdiv(Cout, 64) * cpn_cdiv(P, 64));
    return (long long)tc_slabs(tiles, ksize * ksize * Cin / 16) * P * Cout;
}

extern "C" int cpn_trunk_conv_bn_act(const float* x, const float* wp, int N, int Hin, int Win, int Cin, int Cout, int ksize,
                                     int stride, const float* mean, const float* var, const float* bn_w, const float* bn_b,
                                     float eps, const float* res, int relu, float* out_nhwc, float* out_nchw,
                                     float* scratch, void* stream) {
    CPN_REQUIRE(x && wp && mean && var && bn_w && bn_b && scratch && (out_nhwc || out_nchw), 1, "cpn_trunk_conv_bn_act: null pointer");
    CPN_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && N > 0 && Hin > 0 && Win > 0, 1,
                "cpn_trunk_conv_bn_act: 1x1 / 3x3 kernels with stride 1 / 2 only (got k=%d s=%d)", ksize, stride);
    CPN_REQUIRE(Cin % 16 == 0 && Cout % 4 == 0, 1, "cpn_trunk_conv_bn_act: Cin %% 16 and Cout %% 4 must be 0 (got %d, %d)", Cin, Cout);
    CPN_REQUIRE(((uintptr_t)x | (uintptr_t)wp | (uintptr_t)scratch | (uintptr_t)res | (uintptr_t)out_nhwc | (uintptr_t)mean |
                 (uintptr_t)var | (uintptr_t)bn_w | (uintptr_t)bn_b) % 16 == 0, 1, "cpn_trunk_conv_bn_act: 16-byte alignment");
    const int pad = ksize / 2;
    ConvGeo g{N, Hin, Win, Cin, (Hin + 2 * pad - ksize) / stride + 1, (Win + 2 * pad - ksize) / stride + 1, Cout, ksize, stride, pad};
    const long long P = (long long)N * g.Hout * g.Wout;
    CPN_REQUIRE(P * Cout < (1LL << 31) && (long long)N * Hin * Win * Cin < (1LL << 40), 1, "cpn_trunk_conv_bn_act: map too large");
    const int ntc = cpn_cdiv(Cout, 64), ntp = cpn_cdiv(P, 64);
    const int iters = ksize * ksize * Cin / 16;
    const int nslab = tc_slabs(ntc * ntp, iters);
    const int per_slab = (iters + nslab - 1) / nslab;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(trunk_conv_kernel, dim3((unsigned)(nslab * ntc * ntp)), dim3(64 * TC_WAVES), 0, st, x, wp, g, ntc, ntp,
                       iters, per_slab, scratch);
    CPN_LAUNCH_CHECK("cpn_trunk_conv_bn_act");
    hipLaunchKernelGGL(trunk_conv_epilogue_kernel, dim3(cpn_cdiv(P * Cout / 4, 256)), dim3(256), 0, st, scratch, nslab, (int)P,
                       Cout, g.Hout * g.Wout, mean, var, bn_w, bn_b, eps, res, relu, out_nhwc, out_nchw);
    CPN_LAUNCH_CHECK("cpn_trunk_conv_bn_act (epilogue)");
    return 0;
}
