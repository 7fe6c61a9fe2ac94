// The Adam update of the training step for ALL parameter tensors in ONE launch.
//
// Replaces torch.optim.Adam.step() of the reference's loop (/root/reference train.py:102-105, wrapper.py:149-151; default
// betas / eps, no weight decay, no amsgrad).  The library's fused multi-tensor form still needs ~20 launches for the 636
// tensors of this model (its kernel arguments hold a few dozen tensor addresses each) and runs at 1.7 ms for 1 GB of
// traffic; here the addresses live in a device table the host refreshes per step (the gradient tensors are new every step):
//
//   segs[t]   = { param, grad, offset of the tensor's moments in the two flat state buffers, numel, step size lr / (1 - b1^k),
//                 1 / sqrt(1 - b2^k) }    (k = number of updates THIS tensor has received: tensors without a gradient are skipped
//                 and keep their count, as torch.optim.Adam does)
//   blocks[b] = { tensor, first element }: 2048 elements per 256-thread block, built once
//
// Arithmetic per element (fp32, the operation order of the library's fused kernel):
//   m += (g - m) (1 - b1);   v = b2 v + (1 - b2) g g;   p -= step_size * m / (sqrt(v) * inv_sqrt_bc2 + eps)
// `gscale` (device scalar or NULL) multiplies the gradient first: the clip coefficient can stay on the device.
// `gate` (device scalar or NULL): 0 skips the whole update — the finite-gradient guard without a host read; the per-tensor update
// counts then live on the device too (`counts_in` / `counts_out`, swapped by the caller every step) and the bias corrections
// are formed from them in the kernel.
#include "common.h"

namespace {

struct AdamSeg {
    float* p;
    const float* g;
    long long off;
    int n;
    float step_size;
    float inv_sqrt_bc2;
    int pad_[3];
};
static_assert(sizeof(AdamSeg) == CPN_ADAM_SEG_BYTES, "AdamSeg layout is part of the ABI");

constexpr int ADAM_CHUNK = 2048;

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamSeg* __restrict__ segs, const int2* __restrict__ blocks,
                                                        float* __restrict__ m_all, float* __restrict__ v_all,
                                                        const float* __restrict__ gscale, const float* __restrict__ gate,
                                                        const int* __restrict__ counts_in, int* __restrict__ counts_out,
                                                        double lr, double b1d, double b2d, float b1, float b2, float omb1,
                                                        float omb2, float eps) {
    const int2 blk = blocks[blockIdx.x];
    AdamSeg sg = segs[blk.x];
    const bool update = sg.g != nullptr && !(gate != nullptr && *gate == 0.0f);
    if (counts_in != nullptr) {
        // update counts on the device (the host never learns whether a gated step happened): this step reads counts_in and
        // the first block of every tensor writes counts_out; bias corrections from the tensor's own count, in double
        const int k = counts_in[blk.x] + (update ? 1 : 0);
        if (blk.y == 0 && threadIdx.x == 0) counts_out[blk.x] = k;
        if (!update) return;
        sg.step_size = (float)(lr / (1.0 - pow(b1d, (double)k)));
        sg.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(b2d, (double)k)));
    } else if (!update) {
        return;
    }
    const float gs = gscale ? *gscale : 1.0f;
    float* __restrict__ p = sg.p;
    const float* __restrict__ g = sg.g;
    float* __restrict__ m = m_all + sg.off;
    float* __restrict__ v = v_all + sg.off;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gs;
        mm = mm + (gg - mm) * omb1;
        vv = b2 * vv + omb2 * gg * gg;
        const float denom = sqrtf(vv) * sg.inv_sqrt_bc2 + eps;
        pp -= sg.step_size * mm / denom;
    };
    const int end = blk.y + ADAM_CHUNK < sg.n ? blk.y + ADAM_CHUNK : sg.n;
    const bool vec = (((uintptr_t)p | (uintptr_t)g) & 15) == 0;                  // state offsets are multiples of 4 floats
    if (vec) {
        for (int i = blk.y + 4 * threadIdx.x; i + 3 < end; i += 1024) {
            f32x4 pp = *reinterpret_cast<f32x4*>(p + i), mm = *reinterpret_cast<f32x4*>(m + i),
                  vv = *reinterpret_cast<f32x4*>(v + i);
            const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pp[e], me = mm[e], ve = vv[e];
                upd(pe, gg[e], me, ve);
                pp[e] = pe; mm[e] = me; vv[e] = ve;
            }
            *reinterpret_cast<f32x4*>(p + i) = pp;
            *reinterpret_cast<f32x4*>(m + i) = mm;
            *reinterpret_cast<f32x4*>(v + i) = vv;
        }
        const int tail = blk.y + (end - blk.y) / 4 * 4 + threadIdx.x;
        if (tail < end) upd(p[tail], g[tail], m[tail], v[tail]);
    } else {
        for (int i = blk.y + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

}  // namespace

extern "C" int cpn_adam_chunk(void) { return ADAM_CHUNK; }

extern "C" int cpn_adam_step(const void* segs, const int* blocks, int nblocks, float* exp_avg, float* exp_avg_sq,
                             const float* gscale, const float* gate, const int* counts_in, int* counts_out, double lr,
                             double beta1, double beta2, double eps, void* stream) {
    CPN_REQUIRE(nblocks > 0 && segs && blocks && exp_avg && exp_avg_sq, 1, "cpn_adam_step: empty table");
    CPN_REQUIRE(((uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)segs) % 16 == 0, 1,
                "cpn_adam_step: state buffers and the segment table must be 16-byte aligned");
    CPN_REQUIRE((counts_in == nullptr) == (counts_out == nullptr) && (counts_in == nullptr || counts_in != counts_out), 1,
                "cpn_adam_step: counts_in / counts_out come as a pair of DIFFERENT arrays (or both NULL)");
    hipLaunchKernelGGL(adam_step_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const AdamSeg*)segs,
                       (const int2*)blocks, exp_avg, exp_avg_sq, gscale, gate, counts_in, counts_out, lr, beta1, beta2,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)eps);                    // 1 - beta in double: 1 - 0.999f is off by 1.3e-5
    CPN_LAUNCH_CHECK("cpn_adam_step");
    return 0;
}
