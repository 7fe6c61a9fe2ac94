// f3 — conv_map: the 7x7, 3 -> 64 convolution that produces the full-resolution feature level of the render path
// (/root/reference models/CoPoNeRF.py:69, 182-187) straight from the (N, H, W, 3) image of the input dict.
//
// Fused: (rgb + 1) / 2 -> ImageNet normalisation (utils_training/utils.py:247-257) -> 7x7 / stride 1 / pad 3 -> + bias.
// The zero padding applies to the NORMALISED image, as in the reference (the pad value is 0 after normalisation).
// Outputs: the NCHW fp32 map the caller contract wants (z[3]) and, optionally, the NHWC fp16 copy the render path
// gathers from (cpn_encode_hidden), so the separate layout pass over the largest level disappears.
// One thread = one pixel x 64 output channels (64 fp32 accumulators), 16 x 16 pixel tile, the 22 x 22 x 3 input
// window and the 64 x 147 weights in LDS (weights are read as LDS broadcasts).  9408 MAC per pixel: fp32 VALU work,
// ~0.1 ms per 256^2 stereo pair — MIOpen picked its naive kernel for this shape (0.75 ms per image).
#include <algorithm>

#include "common.h"

namespace {

constexpr int CM_OUT = 64, CM_K = 7, CM_P = 3, CM_T = 16, CM_W = CM_T + 2 * CM_P;

__global__ __launch_bounds__(256) void conv_map7x7_kernel(const float* __restrict__ rgb, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int H, int W,
                                                          float* __restrict__ out_nchw, __half* __restrict__ out_nhwc) {
    __shared__ float win[3][CM_W][CM_W + 1];
    __shared__ float wl[3 * CM_K * CM_K][CM_OUT];             // [tap][out channel]
    const int tid = threadIdx.x;
    const int n = blockIdx.z, y0 = blockIdx.y * CM_T, x0 = blockIdx.x * CM_T;
    for (int i = tid; i < CM_OUT * 3 * CM_K * CM_K; i += 256) {
        const int oc = i / (3 * CM_K * CM_K), tap = i % (3 * CM_K * CM_K);        // w is (64, 3, 7, 7)
        wl[tap][oc] = w[i];
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int i = tid; i < CM_W * CM_W; i += 256) {
        const int wy = i / CM_W, wx = i % CM_W;
        const int y = y0 + wy - CM_P, x = x0 + wx - CM_P;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = 0.0f;
            if (in) v = ((rgb[(((size_t)n * H + y) * W + x) * 3 + c] + 1.0f) / 2.0f - mean[c]) / stdv[c];
            win[c][wy][wx] = v;
        }
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float acc[CM_OUT];
#pragma unroll
    for (int o = 0; o < CM_OUT; ++o) acc[o] = bias[o];
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < CM_K; ++i)
#pragma unroll
            for (int j = 0; j < CM_K; ++j) {
                const float v = win[c][ty + i][tx + j];
                const float* wr = wl[(c * CM_K + i) * CM_K + j];
#pragma unroll
                for (int o = 0; o < CM_OUT; ++o) acc[o] += v * wr[o];
            }
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) return;
    const size_t pix = (size_t)y * W + x;
#pragma unroll
    for (int o = 0; o < CM_OUT; ++o) out_nchw[((size_t)n * CM_OUT + o) * H * W + pix] = acc[o];
    if (out_nhwc) {
        __half* d = out_nhwc + ((size_t)n * H * W + pix) * CM_OUT;
#pragma unroll
        for (int o8 = 0; o8 < CM_OUT / 8; ++o8) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)acc[o8 * 8 + e];
            *reinterpret_cast<half8*>(d + o8 * 8) = h;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Inference BatchNorm (+ residual) (+ ReLU) of the ResNet-34 trunk (models/backbone.py:10-102 via torchvision's
// BasicBlock: bn(conv(x)), `out += identity`, relu) as ONE pass: y = act((x - mean[c]) * invstd[c] * w[c] + b[c] + res).
// The library spends three launches on it (batch norm, add, clamp): 85 of get_z's launches -> 36.
// Thread = 4 consecutive pixels of one (n, c) plane (HW % 4 == 0), NCHW contiguous.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_kernel(const float* x, const float* __restrict__ res,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                     int C, int HW4, long long total4, int relu, float* y) {     // y may alias x
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW4) % C);
        const float m = mean[c], invstd = 1.0f / sqrtf(var[c] + eps), g = w[c], beta = b[c];
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (res) r = reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float o = (v[e] - m) * invstd * g + beta;
            if (res) o += r[e];
            v[e] = (relu && o < 0.0f) ? 0.0f : o;                  // NaN stays NaN, as torch.relu keeps it
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

}  // namespace

extern "C" int cpn_bn_act(const float* x, const float* res, const float* mean, const float* var, const float* w,
                          const float* b, float eps, int N, int C, int HW, int relu, float* y, void* stream) {
    CPN_REQUIRE(x && mean && var && w && b && y, CPN_E_ARG, "cpn_bn_act: null pointer");
    CPN_REQUIRE(N > 0 && C > 0 && HW > 0 && (HW % 4) == 0, CPN_E_SHAPE, "cpn_bn_act: need HW %% 4 == 0 (got %d)", HW);
    CPN_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!res || ((uintptr_t)res % 16) == 0), CPN_E_ARG,
                "cpn_bn_act: tensors must be 16-byte aligned");
    const long long total4 = (long long)N * C * (HW / 4);
    const unsigned blocks = (unsigned)std::min<long long>(cpn_cdiv(total4, 256), 1 << 16);
    hipLaunchKernelGGL(bn_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, res, mean, var, w, b, eps, C, HW / 4,
                       total4, relu, y);
    CPN_LAUNCH_CHECK("cpn_bn_act");
    return 0;
}

extern "C" int cpn_conv_map7x7(const float* rgb, const float* w, const float* bias, int N, int H, int W, float* out_nchw,
                               uint16_t* out_nhwc_f16, void* stream) {
    CPN_REQUIRE(rgb && w && bias && out_nchw, CPN_E_ARG, "cpn_conv_map7x7: null pointer");
    CPN_REQUIRE(N > 0 && N < 65536 && H > 0 && W > 0, CPN_E_SHAPE, "cpn_conv_map7x7: bad shape");
    hipLaunchKernelGGL(conv_map7x7_kernel, dim3(cpn_cdiv(W, CM_T), cpn_cdiv(H, CM_T), N), dim3(256), 0, (hipStream_t)stream,
                       rgb, w, bias, H, W, out_nchw, (__half*)out_nhwc_f16);
    CPN_LAUNCH_CHECK("cpn_conv_map7x7");
    return 0;
}
