// f4 — device side of the input pipeline: uint8 frames -> the float tensors of the model's input dict.
//
// The reference's dataset (/root/reference data/realestate10k_dataio.py:333-441) decodes a compressed per-scene .npz,
// square-crops every 256 x 455 frame on the host (utils_training/data_util.py:116-121), converts it to float32
// (`rgb.astype(np.float32) / 127.5 - 1`, :353, :437) and ships 2.4 MB of floats per sample through the DataLoader.
// Here the frames stay uint8 until they are on the GPU (coponerf_amd/shards.py: pre-decoded, pre-resized shards,
// pinned staging): one kernel crops, normalises the two context frames and gathers the query colours at the selected
// pixels.  HBM-trivial (0.35 MB in, 1.6 MB out per sample); it exists to keep 8 GPUs from waiting on host cores.
// Arithmetic = the reference's, bit for bit: float(u8) / 127.5f - 1.0f (IEEE division and subtraction).
#include "common.h"

namespace {

__device__ __forceinline__ float norm_u8(uint8_t v) { return (float)v / 127.5f - 1.0f; }

// ctx: thread = one cropped context pixel (all 3 channels); query: thread = one selected ray
__global__ void prepare_input_kernel(const uint8_t* __restrict__ frames, int Hs, int Ws, int y0, int x0, int H, int W,
                                     int B, int R, const int* __restrict__ ray_pix, float* __restrict__ ctx_rgb,
                                     float* __restrict__ qry_rgb) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nctx = (long long)B * 2 * H * W;
    if (idx < nctx) {
        const int x = (int)(idx % W);
        long long t = idx / W;
        const int y = (int)(t % H); t /= H;
        const int v = (int)(t % 2), b = (int)(t / 2);
        const uint8_t* s = frames + ((((size_t)b * 3 + v) * Hs + (y0 + y)) * Ws + (x0 + x)) * 3;
        float* d = ctx_rgb + idx * 3;
        d[0] = norm_u8(s[0]); d[1] = norm_u8(s[1]); d[2] = norm_u8(s[2]);
        return;
    }
    const long long q = idx - nctx;
    if (q >= (long long)B * R) return;
    const int b = (int)(q / R);
    const int pix = ray_pix[q];                                   // y * W + x in the CROPPED query frame
    const int y = pix / W, x = pix % W;
    const uint8_t* s = frames + ((((size_t)b * 3 + 2) * Hs + (y0 + y)) * Ws + (x0 + x)) * 3;
    float* d = qry_rgb + q * 3;
    d[0] = norm_u8(s[0]); d[1] = norm_u8(s[1]); d[2] = norm_u8(s[2]);
}

}  // namespace

extern "C" int cpn_prepare_input(const uint8_t* frames_u8, int B, int Hs, int Ws, int y0, int x0, int H, int W, int R,
                                 const int32_t* ray_pix, float* ctx_rgb, float* qry_rgb, void* stream) {
    CPN_REQUIRE(frames_u8 && ray_pix && ctx_rgb && qry_rgb, CPN_E_ARG, "cpn_prepare_input: null pointer");
    CPN_REQUIRE(B > 0 && R > 0 && H > 0 && W > 0 && y0 >= 0 && x0 >= 0 && y0 + H <= Hs && x0 + W <= Ws, CPN_E_SHAPE,
                "cpn_prepare_input: crop (%d,%d)+(%d,%d) outside the %dx%d frame", y0, x0, H, W, Hs, Ws);
    const long long total = (long long)B * 2 * H * W + (long long)B * R;
    hipLaunchKernelGGL(prepare_input_kernel, dim3(cpn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, frames_u8, Hs, Ws,
                       y0, x0, H, W, B, R, ray_pix, ctx_rgb, qry_rgb);
    CPN_LAUNCH_CHECK("cpn_prepare_input");
    return 0;
}
