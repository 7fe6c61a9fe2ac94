// Backward kernels of the render path (training, BASELINE config 3).
//
// The plain GEMM gradients (dX = dY W, dW = dY^T X) go to hipBLASLt through torch.matmul on the host side; this
// unit holds the two stages that are not plain GEMMs:
//
//   cpn_attend_hidden_bwd  gradient of the joint softmax + attention-weighted hidden sum (cpn_attend_hidden),
//                          i.e. of /root/reference models/CoPoNeRF.py:450-461 / 475-485 in the folded form
//   cpn_gather_rows_bwd    gradient of the bilinear multi-scale gather w.r.t. the feature maps (scatter-add),
//                          i.e. of F.grid_sample at models/CoPoNeRF.py:312 / 370 (no coordinate gradient: all sample
//                          coordinates derive from poses only and `pt` is detached, CoPoNeRF.py:380-381, 433)
#include <algorithm>

#include "common.h"
#include "taps.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Weight gradients of the 128-wide per-sample layers: dW (128, K) = dY^T (128, M) . X (M, K) with M = millions of rows
// and K = 128 (fp16 operands) or 16 (fp32 operands).  69 / 8.6 GFLOP over 1.07 GB of operands: pure streaming work,
// but the library's GEMM heuristics pick split-K kernels that run these at 24 TFLOP/s (2.9 ms and 2.2 ms per call,
// 13 ms per training step).  Here a workgroup streams 64-row tiles of both operands through LDS (coalesced 16-byte
// loads), the contraction index (the row) is put on the MFMA K axis by reading the fragments column-wise from the
// row-major LDS image (2-byte LDS reads: slow per element, irrelevant at 0.27 FLOP per byte), partial sums leave with
// one atomic pass per workgroup.  db[n] = sum_m dY[m][n] falls out of the same loads.
// ---------------------------------------------------------------------------------------------
constexpr int WS_ROWS = 64;                 // rows per tile
constexpr int WS_LD = 128 + 2;              // padded row stride (halves): the 4 K groups of a fragment hit different banks

__global__ __launch_bounds__(256) void wgrad_skinny_f16_kernel(const __half* __restrict__ dY, const __half* __restrict__ X,
                                                               int ldx, long long M, float* __restrict__ dW,
                                                               float* __restrict__ db) {
    __shared__ __attribute__((aligned(16))) __half sy[WS_ROWS * WS_LD];
    __shared__ __attribute__((aligned(16))) __half sx[WS_ROWS * WS_LD];
    __shared__ float sdb[16][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int lr = tid >> 4, lc = (tid & 15) * 8;            // loader: row lr (+16 per pass), 8 columns from lc
    f32x4 acc[2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long ntiles = (M + WS_ROWS - 1) / WS_ROWS;
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long long m0 = t * WS_ROWS;
        half8 vy[4], vx[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long long m = m0 + lr + 16 * p;
            if (m < M) {
                vy[p] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(dY + m * 128 + lc));
                vx[p] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(X + m * ldx + lc));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { vy[p][e] = (_Float16)0.0f; vx[p][e] = (_Float16)0.0f; }
            }
        }
        __syncthreads();                                      // previous tile's fragments are read
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // rows are 260 bytes apart: 4-byte stores
            unsigned* py = reinterpret_cast<unsigned*>(sy + (lr + 16 * p) * WS_LD + lc);
            unsigned* px = reinterpret_cast<unsigned*>(sx + (lr + 16 * p) * WS_LD + lc);
            const u32x4 uy = __builtin_bit_cast(u32x4, vy[p]), ux = __builtin_bit_cast(u32x4, vx[p]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { py[e] = uy[e]; px[e] = ux[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)vy[p][e];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < WS_ROWS / 32; ++ks) {
            // fragment element e of lane (fi, fg): row ks*32 + fg*8 + e, column tile*16 + fi
            half8 fa[2], fb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int row = ks * 32 + fg * 8 + e;
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a][e] = sy[row * WS_LD + (wave * 2 + a) * 16 + fi];
#pragma unroll
                for (int b = 0; b < 8; ++b) fb[b][e] = sx[row * WS_LD + b * 16 + fi];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 8; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
    // D tile (a, b): lane (fi, fg) holds rows n = (wave*2+a)*16 + fg*4 + i, column k = b*16 + fi
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                atomicAdd(dW + ((wave * 2 + a) * 16 + fg * 4 + i) * 128 + b * 16 + fi, acc[a][b][i]);
    if (db) {
        // threads with the same (tid & 15) own the same 8 columns: reduce the 16 row groups through LDS
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bsum[e];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            bsum[e] = v;
        }
        __syncthreads();
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sdb[lane][e] = 0.0f;
        }
        __syncthreads();
        if (fg == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(&sdb[fi][e], bsum[e]);
        }
        __syncthreads();
        if (tid < 128) atomicAdd(db + tid, sdb[tid >> 3][tid & 7]);
    }
}

// Backward of cpn_local_hidden (out = fp16(relu(W . L(row) + b + add[ray]))) in ONE pass over the incoming gradient:
//   d[row, n] = out[row, n] > 0 ? ds[row, n] / scale : 0          (ds fp16, carries the pass's power-of-two scale)
//   dW[n, k] += sum_rows d[row, n] * L[row, k],  db[n] += sum_rows d,  dadd[ray, n] = sum over the ray's V*S rows
// L(row) is rebuilt from loc8 / coords9 exactly as the forward kernel builds it (channel order of CoPoNeRF.py:445).  The
// autograd form materialised d as fp32 (1 GB), the 16-wide input rows (134 MB), and ran a GEMM and two reductions over
// them.  block = `rays_per_block` rays; thread = 4 output rows n x one of 8 row lanes; a ray's 8 lane sums meet in LDS.
__global__ __launch_bounds__(256) void local_hidden_bwd_kernel(
    const __half* __restrict__ ds, const __half* __restrict__ out, const float* __restrict__ loc8,
    const float* __restrict__ coords9, const float* __restrict__ scale, int V, int R, int S, int nrays, int rays_per_block,
    float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dadd) {
    // Round 6: on the fp32 matrix cores.  C[n][k] += sum_rows d[row][n] * L[row][k] is a GEMM with the rows as its contraction
    // index: 64 rows of masked d (fp16, still scaled) and of L are staged in LDS, a wave owns two of the eight 16-channel
    // tiles and issues 2 v_mfma_f32_16x16x4_f32 per 4 rows.  Columns 3..5 of L are zeros by construction (CoPoNeRF.py:445):
    // column 3 carries ONES here, so C[n][3] is the sum of d over the rows - per ray that is dadd, over everything db - and
    // dW[:, 3] is left at zero.  (The VALU form, 32 packed FMAs per row and thread, ran at 0.5 ms per launch, a third of it
    // issue and the rest waiting.)
    constexpr int CH = 64;                                       // rows per staged chunk
    constexpr int DLD = 128 + 8;                                 // halves per staged row of d
    __shared__ __half dsh[CH * DLD];
    __shared__ float lsh[CH * 17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const float inv = 1.0f / scale[0];
    const int rpr = V * S;
    const int srow = tid >> 2, sq = tid & 3;                     // staging: thread = (row of the chunk, quarter of its 128 channels)
    f32x4 blk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int ray_lo = blockIdx.x * rays_per_block;
    const int ray_hi = min(ray_lo + rays_per_block, nrays);
    half8 dv[4], ov[4];
    auto load_chunk = [&](int ray, int m0) {
        const size_t row = (size_t)ray * rpr + min(m0 + srow, rpr - 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dv[u] = *reinterpret_cast<const half8*>(ds + row * 128 + sq * 32 + u * 8);
            ov[u] = *reinterpret_cast<const half8*>(out + row * 128 + sq * 32 + u * 8);
        }
    };
    const int nchunk = (rpr + CH - 1) / CH;
    if (ray_lo < ray_hi) load_chunk(ray_lo, 0);
    for (int ray = ray_lo; ray < ray_hi; ++ray) {
        const int b = ray / R, r = ray - b * R;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int c = 0; c < nchunk; ++c) {
            const int m0 = c * CH;
            const bool live = m0 + srow < rpr;
            __syncthreads();                                     // the previous chunk has been consumed
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                half8 d8;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    d8[e] = (live && (float)ov[u][e] > 0.0f) ? dv[u][e] : (_Float16)0.0f;
                *reinterpret_cast<half8*>(dsh + srow * DLD + sq * 32 + u * 8) = d8;
            }
            {   // L(row), four of its sixteen entries per thread (the forward kernel's construction; k = 3 carries the ones)
                f32x4 lv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (live) {
                    const int m = m0 + srow;
                    const int v = m / S, sm = m - v * S;
                    const size_t nr = ((size_t)(b * V + v)) * R + r;
                    const float* lp = loc8 + (nr * S + sm) * 8;
                    const float* c9 = coords9 + nr * 9;
                    if (sq == 0) lv = f32x4{lp[0], lp[1], lp[2], 1.0f};
                    else if (sq == 1) lv = f32x4{0.f, 0.f, c9[0], c9[1]};
                    else if (sq == 2) lv = f32x4{c9[2], lp[3], lp[4], lp[5]};
                    else lv = f32x4{lp[6], c9[6], c9[7], c9[8]};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) lsh[srow * 17 + sq * 4 + e] = lv[e];
            }
            __syncthreads();
            // the next chunk's rows are in flight under this chunk's MFMAs
            if (c + 1 < nchunk) load_chunk(ray, m0 + CH);
            else if (ray + 1 < ray_hi) load_chunk(ray + 1, 0);
#pragma unroll 4
            for (int k4 = 0; k4 < CH / 4; ++k4) {
                const int row = k4 * 4 + lk;
                const float bval = lsh[row * 17 + li];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float aval = (float)dsh[row * DLD + (wave * 2 + t) * 16 + li];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aval, bval, acc[t], 0, 0, 0);
                }
            }
        }
        // D[m = 4 lk + i][n = li]: m -> channel of the tile, n -> column of L
        if (dadd && li == 3) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) dadd[(size_t)ray * 128 + (wave * 2 + t) * 16 + lk * 4 + i] = acc[t][i] * inv;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) blk[t] += acc[t];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = (wave * 2 + t) * 16 + lk * 4 + i;
            if (li == 3) atomicAdd(db + n, blk[t][i] * inv);
            else atomicAdd(dW + n * 16 + li, blk[t][i] * inv);
        }
}

}  // namespace

extern "C" int cpn_local_hidden_bwd(const uint16_t* ds, const uint16_t* out, const float* loc8, const float* coords9,
                                    const float* scale, int B, int V, int R, int S, float* dW, float* db, float* dadd,
                                    void* stream) {
    CPN_REQUIRE(ds && out && loc8 && coords9 && scale && dW && db, CPN_E_ARG, "cpn_local_hidden_bwd: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0, CPN_E_SHAPE, "cpn_local_hidden_bwd: bad shape");
    CPN_REQUIRE(((uintptr_t)ds % 8) == 0 && ((uintptr_t)out % 8) == 0 && ((uintptr_t)loc8 % 16) == 0, CPN_E_ARG,
                "cpn_local_hidden_bwd: operands must be aligned");
    const int nrays = B * R;
    const int rpb = std::max(1, (nrays + 2047) / 2048);
    hipLaunchKernelGGL(local_hidden_bwd_kernel, dim3((nrays + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)ds, (const __half*)out, loc8, coords9, scale, V, R, S, nrays, rpb, dW, db, dadd);
    CPN_LAUNCH_CHECK("cpn_local_hidden_bwd");
    return 0;
}

extern "C" int cpn_wgrad_skinny_f16(const uint16_t* dY, const uint16_t* X, int ldx, long long M, float* dW, float* db,
                                    void* stream) {
    CPN_REQUIRE(dY && X && dW, CPN_E_ARG, "cpn_wgrad_skinny_f16: null pointer");
    CPN_REQUIRE(M > 0 && ldx >= 128 && (ldx % 8) == 0, CPN_E_SHAPE, "cpn_wgrad_skinny_f16: bad shape (ldx=%d)", ldx);
    CPN_REQUIRE(((uintptr_t)dY % 16) == 0 && ((uintptr_t)X % 16) == 0, CPN_E_ARG,
                "cpn_wgrad_skinny_f16: operands must be 16-byte aligned");
    const long long ntiles = (M + WS_ROWS - 1) / WS_ROWS;
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 1024);
    hipLaunchKernelGGL(wgrad_skinny_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __half*)dY,
                       (const __half*)X, ldx, M, dW, db);
    CPN_LAUNCH_CHECK("cpn_wgrad_skinny_f16");
    return 0;
}


extern "C" long long cpn_gather_bwd_chunks(int R, int S);

namespace {

constexpr int HC = 1664;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup per ray.  T = V*S rows.
//   dw[row]   = <hid[row], dhbar> + dw_ext[row]
//   dl[row]   = w[row] * (dw[row] - sum_r w[r] dw[r]) / 11.31
//   dqa[row]  = dl[row] * qb[row] ;  dqb[row] = dl[row] * qa[row]
//   dhid[row] = w[row] * dhbar
__global__ __launch_bounds__(256) void attend_hidden_bwd_kernel(
    const __half* __restrict__ qa, const __half* __restrict__ qb, const __half* __restrict__ hid,
    const float* __restrict__ at_wt, const float* __restrict__ dhbar, const float* __restrict__ dw_ext, int V, int R,
    int S, int ray0, __half* __restrict__ dqa, __half* __restrict__ dqb, __half* __restrict__ dhid,
    const __half* __restrict__ dqb_acc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* dh = reinterpret_cast<float*>(smem_raw);          // HC floats: dhbar of this ray
    float* wts = dh + HC;                                    // T
    float* dl = wts + V * S;                                 // T
    float* red = dl + V * S;                                 // 4
    const int T = V * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lray = blockIdx.x;
    const size_t row0 = (size_t)lray * T;
    const unsigned ray = (unsigned)ray0 + lray;
    const int b = (int)(ray / (unsigned)R), r = (int)(ray % (unsigned)R);

    for (int c = tid; c < HC; c += 256) dh[c] = dhbar[(size_t)lray * HC + c];
    for (int row = tid; row < T; row += 256) {
        const int v = row / S, s = row - v * S;
        wts[row] = at_wt[(((size_t)(b * V + v)) * R + r) * S + s];
    }
    __syncthreads();
    // dw[row] = <hid[row, :], dhbar>: one wave per row, 16-byte loads (a lane owns the 8-channel chunks lane, lane + 64, ...
    // of the 208 in a row and keeps their dhbar values in registers), two rows in flight per wave.  (4-byte loads, one
    // row at a time and dhbar re-read from LDS per element ran this pass at 3 TB/s: 2.4 ms per call.)
    constexpr int NCH = HC / 8;                              // 208 chunks per row
    float dreg[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) dreg[k][e] = (lane + 64 * k < NCH) ? dh[(lane + 64 * k) * 8 + e] : 0.0f;
    auto row_dot = [&](const half8 (&h)[4]) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += (float)h[k][e] * dreg[k][e];
        return wave_sum_f(acc);
    };
    auto load_row = [&](int row, half8 (&h)[4]) {
        const __half* hp = hid + (row0 + min(row, T - 1)) * HC;
#pragma unroll
        for (int k = 0; k < 4; ++k)                          // chunk 208.. of the last group: re-read chunk lane (weight 0)
            h[k] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(hp + ((lane + 64 * k < NCH) ? lane + 64 * k : lane) * 8));
    };
    float part = 0.f;
    auto finish = [&](int row, float acc) {
        if (lane == 0 && row < T) {
            float dwv = acc;
            if (dw_ext) {
                const int v = row / S, s = row - v * S;
                dwv += dw_ext[(((size_t)(b * V + v)) * R + r) * S + s];
            }
            dl[row] = dwv;
            part += wts[row] * dwv;
        }
    };
    for (int row = wave; row < T; row += 8) {
        half8 ha[4], hb[4];
        load_row(row, ha);
        load_row(row + 4, hb);
        finish(row, row_dot(ha));
        finish(row + 4, row_dot(hb));
    }
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const float dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int row = tid; row < T; row += 256) dl[row] = wts[row] * (dl[row] - dot) / 11.31f;
    __syncthreads();
    // dqa / dqb: thread = (row, 8-channel group)
    for (int i = tid; i < T * 16; i += 256) {
        const int row = i >> 4, g = i & 15;
        const half8 a = *reinterpret_cast<const half8*>(qa + (row0 + row) * 128 + g * 8);
        const half8 bq = *reinterpret_cast<const half8*>(qb + (row0 + row) * 128 + g * 8);
        const float d = dl[row];
        half8 oa, ob;
        if (dqb_acc) {                      // qb is shared with another attention round: its gradient is summed here
            const half8 pq = *reinterpret_cast<const half8*>(dqb_acc + (row0 + row) * 128 + g * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                oa[e] = (_Float16)(d * (float)bq[e]);
                ob[e] = (_Float16)((float)pq[e] + d * (float)a[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                oa[e] = (_Float16)(d * (float)bq[e]);
                ob[e] = (_Float16)(d * (float)a[e]);
            }
        }
        *reinterpret_cast<half8*>(dqa + (row0 + row) * 128 + g * 8) = oa;
        *reinterpret_cast<half8*>(dqb + (row0 + row) * 128 + g * 8) = ob;
    }
    // dhid = w[row] * dhbar : thread = 8 channels, rows streamed (skipped when the caller combines the hidden-activation
    // gradients of all consumers in cpn_hid_grad_combine instead of materialising one 7 GB tensor per consumer)
    if (dhid && tid < HC / 8) {
        float d8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) d8[e] = dh[tid * 8 + e];
        for (int row = 0; row < T; ++row) {
            const float w = wts[row];
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)(w * d8[e]);
            *reinterpret_cast<half8*>(dhid + (row0 + row) * HC + tid * 8) = o;
        }
    }
}

// d(pre-activation of the first encoder layer) from ALL consumers of `hid` in one pass:
//     out[row, c] = hid[row, c] > 0 ? dkey[row, c] + w1[row] * dh1[ray, j*832 + c] + w2[row] * dh2[ray, j*832 + c] : 0
// hid feeds the key path (a GEMM, gradient dkey) and the two attention-weighted hidden sums (gradients w_i (x) dhbar_i,
// rank one per ray).  Autograd used to materialise the three 7 GB tensors, add them twice and mask the sum (84 GB of
// traffic per step at 4 x 4096 rays); this kernel reads hid and dkey once and writes the masked sum once (21 GB).
// thread = (row, 8-channel chunk); all gradient operands carry the pass's power-of-two scale (train_fns.GradScale).
__global__ __launch_bounds__(256) void hid_grad_combine_kernel(
    const __half* __restrict__ dkey, const __half* __restrict__ hid, const float* __restrict__ w1,
    const float* __restrict__ dh1, const float* __restrict__ w2, const float* __restrict__ dh2, int V, int R, int S,
    int ray0, long long nchunks, __half* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nchunks) return;
    const int c8 = (int)(idx % 104);
    long long row = idx / 104;
    const int j = (int)(row & 1);
    long long t = row >> 1;
    const int s = (int)(t % S); t /= S;
    const int v = (int)(t % V); t /= V;                             // t = ray inside this launch
    const long long ray = ray0 + t;
    const int b = (int)(ray / R), r = (int)(ray % R);
    const size_t widx = (((size_t)(b * V + v)) * R + r) * S + s;
    const half8 h = *reinterpret_cast<const half8*>(hid + (size_t)row * 832 + c8 * 8);
    float acc[8];
    if (dkey) {
        const half8 d = *reinterpret_cast<const half8*>(dkey + (size_t)row * 832 + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = (float)d[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    }
    const size_t hoff = (size_t)t * HC + j * 832 + c8 * 8;
    if (w1) {
        const float w = w1[widx];
        const f32x4 a = *reinterpret_cast<const f32x4*>(dh1 + hoff), bq = *reinterpret_cast<const f32x4*>(dh1 + hoff + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += w * a[e]; acc[4 + e] += w * bq[e]; }
    }
    if (w2) {
        const float w = w2[widx];
        const f32x4 a = *reinterpret_cast<const f32x4*>(dh2 + hoff), bq = *reinterpret_cast<const f32x4*>(dh2 + hoff + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += w * a[e]; acc[4 + e] += w * bq[e]; }
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)h[e] > 0.0f ? (_Float16)acc[e] : (_Float16)0.0f;
    *reinterpret_cast<half8*>(out + (size_t)row * 832 + c8 * 8) = o;
}

// Scatter-add of the gather gradient without atomic contention.
// Global fp32 atomics collapse here (millions of rows land on a few hundred coarse pixels: 316 ms per training step)
// and ds_add_f32 is ~10x slower than a plain LDS write on gfx950 (measured, tools/gbwd_bench.py), so accumulation is
// made exclusive instead: ONE WAVE owns one (image, level, 8x4-pixel tile, 64-channel slice) accumulator in LDS
// (8 KB fp32; with 8x8 tiles LDS held the kernel at 2 waves per SIMD: 12.5 vs 9.0 ms per training step) with lane =
// channel, so its read-modify-writes need no atomics at all.
//   pass 1 (gather_bbox_kernel): per 64-row chunk and level, the pixel bounding box of the rows' 2x2 footprints.
//   pass 2 (gather_rows_bwd_kernel): the wave
//     A) tests 64 chunk boxes at a time against its tile (lane = chunk), and for the chunks that may touch it
//        evaluates the rows (lane = row) and queues a 16-byte descriptor per row that really does,
//     B) drains the queue NB rows at a time: the 128-byte channel slices of dxin for the NEXT NB rows are in flight
//        while the current NB are accumulated.
// Coarse levels are additionally split G ways over the rows; tiles are flushed once with global atomics.
#ifndef CPN_GBWD_TPY
#define CPN_GBWD_TPY 4
#endif
constexpr int TP = 8;        // tile width (pixels)
constexpr int TPY = CPN_GBWD_TPY;   // tile height: 8 x TPY x 64 ch x 4 B of LDS per wave decide the occupancy
constexpr int TC = 64;       // channels per slice = lanes
constexpr int QCAP = 128;    // descriptor queue entries per wave
constexpr int NB = 16;       // rows per drain batch
constexpr int WAVES = 2;     // waves (= roles) per workgroup
struct GatherBwdPlan {
    int lvl, tiles_x, tiles, slices, G;                    // roles of this launch: (img, g, tile, slice), slice fastest
    int maxchunks;                                         // chunk slots per image in the bbox scratch
    int col0;                                              // first column of the level's channels in a gradient row
};

// the rows that read image `img`: idx in [0, 2*nr*S) -> (first j=0: own view at pixel_val, then j=1: other view at
// sec_grid); same enumeration in both passes
struct RowRef {
    unsigned row;      // row of dxin
    float2 g;          // normalised sample coordinate
    int j;
};
__device__ __forceinline__ RowRef row_of(int idx, int per, int S, bool s_pow2, int s_shift, int rlo, int b, int vi, int V,
                                         int R, int ray0, const float* __restrict__ pixel_val,
                                         const float* __restrict__ sec_grid) {
    RowRef o;
    o.j = idx >= per;
    const int rem = idx - o.j * per;
    const int rr = s_pow2 ? (rem >> s_shift) : (rem / S);
    const int sm = rem - rr * S;
    const int r = rlo + rr;
    const int v = o.j ? (V - 1 - vi) : vi;
    const size_t sidx = (((size_t)(b * V + v)) * R + r) * S + sm;
    o.g = *reinterpret_cast<const float2*>((o.j ? sec_grid : pixel_val) + sidx * 2);
    o.row = ((((unsigned)(b * R + r - ray0)) * V + v) * S + sm) * 2 + o.j;
    return o;
}
// pixel-space sample position at a level, with the clamps of make_taps
__device__ __forceinline__ void level_xy(float2 g, int j, float fW, float fH, float& x, float& y) {
    x = ((g.x + 1.0f) * fW - 1.0f) / 2.0f;
    y = ((g.y + 1.0f) * fH - 1.0f) / 2.0f;
    if (!j) {
        x = fminf(fmaxf(x, 0.0f), fW - 1.0f);
        y = fminf(fmaxf(y, 0.0f), fH - 1.0f);
    } else {
        x = fminf(fmaxf(x, -2.0f), fW + 1.0f);
        y = fminf(fmaxf(y, -2.0f), fH + 1.0f);
    }
}

// Conservative pixel boxes of the rows' footprints at ONE level, per 16-row QUARTER of a 64-row chunk (round 6).  A chunk
// is one ray's samples along its epipolar line: the box of the whole line covers some 300 tiles of which 40 hold samples,
// and the tiles' scan evaluated every row of every chunk whose box they touched (that scan, not the accumulation, was the
// kernel's time: 1.58 ms for a gigabyte of rows); a quarter line's box is a sixteenth of the area.
__global__ __launch_bounds__(256) void gather_bbox_kernel(
    int H, int W, const float* __restrict__ pixel_val, const float* __restrict__ sec_grid, int V, int R, int S,
    int ray0, int nrays, int maxchunks, int nimg, int lvl, int4* __restrict__ bbox) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int img = wid / maxchunks, c = wid - img * maxchunks;
    if (img >= nimg) return;
    const int b = img / V, vi = img - b * V;
    const int rlo = max(ray0, b * R) - b * R, rhi = min(ray0 + nrays, (b + 1) * R) - b * R;
    const int per = max(rhi - rlo, 0) * S, total = 2 * per;
    const int idx = c * 64 + lane;
    const bool live = idx < total;
    const int shift = 4 - lvl - (lvl == 3);
    int x0 = 1 << 30, y0 = 1 << 30, x1 = -(1 << 30), y1 = -(1 << 30);
    if (live) {
        const RowRef rf = row_of(idx, per, S, (S & (S - 1)) == 0, 31 - __builtin_clz(S), rlo, b, vi, V, R, ray0, pixel_val, sec_grid);
        float x, y;
        level_xy(rf.g, rf.j, (float)(W >> shift), (float)(H >> shift), x, y);
        x0 = (int)floorf(x); y0 = (int)floorf(y);
        x1 = x0 + 1; y1 = y0 + 1;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {                          // within the 16 lanes of a quarter
        x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
        x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
    }
    if ((lane & 15) == 0) bbox[((size_t)img * maxchunks + c) * 4 + (lane >> 4)] = make_int4(x0, y0, x1, y1);
}

__global__ __launch_bounds__(64 * WAVES) void gather_rows_bwd_kernel(
    const __half* __restrict__ dxin, int ldx, int H, int W, const float* __restrict__ pixel_val,
    const float* __restrict__ sec_grid, int V, int R, int S, int ray0, int nrays, float* __restrict__ dmap0,
    float* __restrict__ dmap1, float* __restrict__ dmap2, float* __restrict__ dmap3, GatherBwdPlan plan,
    int nroles, const int4* __restrict__ bbox) {
    __shared__ float tiles_lds[WAVES][TP * TPY * TC];
    __shared__ uint4 queue_lds[WAVES][QCAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = tiles_lds[wave];
    uint4* queue = queue_lds[wave];
    // waves that run at the same time share (img, g): they read neighbouring 128-byte slices of the SAME rows
    int role = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);
    if (role >= nroles) return;
    const int lvl = plan.lvl, G = plan.G;
    const int slice = role % plan.slices; role /= plan.slices;
    const int tidx = role % plan.tiles; role /= plan.tiles;
    const int g = role % G;
    const int img = role / G;
    const int tx0 = (tidx % plan.tiles_x) * TP, ty0 = (tidx / plan.tiles_x) * TPY;
    const int shift = 4 - lvl - (lvl == 3);
    const int Hl = H >> shift, Wl = W >> shift;
    const int C = (lvl == 3) ? 64 : 256;
    const __half* dcol = dxin + plan.col0 + slice * TC + lane;
    float* base = (lvl == 0) ? dmap0 : (lvl == 1) ? dmap1 : (lvl == 2) ? dmap2 : dmap3;

#pragma unroll
    for (int i = 0; i < TP * TPY; ++i) tile[i * TC + lane] = 0.0f;

    const int b = img / V, vi = img - b * V;
    const int rlo = max(ray0, b * R) - b * R, rhi = min(ray0 + nrays, (b + 1) * R) - b * R;
    const int per = max(rhi - rlo, 0) * S, total = 2 * per;
    const int nchunks = (total + 63) >> 6;
    const int cpg = (nchunks + G - 1) / G;
    const int c_end = min(nchunks, (g + 1) * cpg);
    const bool s_pow2 = (S & (S - 1)) == 0;
    const int s_shift = 31 - __builtin_clz(S);
    const float fW = (float)Wl, fH = (float)Hl;
    const int4* boxes = bbox + (size_t)img * plan.maxchunks * 4;          // [chunk][16-row quarter], this level

    auto drain = [&](int n) {
#ifdef CPN_GBWD_NO_DRAIN                                       // timing-only ablation (tools/gbwd_bench.py): scan phase alone
        if (n >= 0) return;
#endif
        // raw halves are carried across the iteration and converted at use, so the loads of batch i+1 stay in
        // flight during the accumulation of batch i (a conversion at the load would wait for it there)
        __half cur[NB], nxt[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            nxt[u] = dcol[(size_t)queue[min(u, n - 1)].x * ldx];      // unconditional: no branch, no wait between loads
        }
        for (int i = 0; i < n; i += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
#pragma unroll
            for (int u = 0; u < NB; ++u)
                nxt[u] = dcol[(size_t)queue[min(i + NB + u, n - 1)].x * ldx];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (i + u >= n) break;
                const float du = __half2float(cur[u]);
                const uint4 q = queue[i + u];
                const int pk = __builtin_amdgcn_readfirstlane((int)q.y);
                const float hfx = __uint_as_float(q.z), hfy = __uint_as_float(q.w);
                const int hx = (pk & 255) - 1, hy = ((pk >> 8) & 255) - 1;
                float* t = tile + (hy * TP + hx) * TC + lane;
                if (pk & (1 << 16)) t[0] += du * ((1.0f - hfx) * (1.0f - hfy));
                if (pk & (2 << 16)) t[TC] += du * (hfx * (1.0f - hfy));
                if (pk & (4 << 16)) t[TP * TC] += du * ((1.0f - hfx) * hfy);
                if (pk & (8 << 16)) t[TP * TC + TC] += du * (hfx * hfy);
            }
        }
    };

    int qn = 0;
    // A1: lane = chunk; its four quarter boxes are one 64-byte read, and the next 64 chunks' boxes are in flight while
    // this batch is worked on (a tile scans every chunk of its image: the scan is a latency chain unless it is fed ahead)
    auto load_boxes = [&](int cb, int4 (&bx)[4]) {
        const int c = min(cb + lane, c_end - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) bx[k] = boxes[(size_t)c * 4 + k];
    };
    int4 nxt[4];
    if (g * cpg < c_end) load_boxes(g * cpg, nxt);
    for (int cb = g * cpg; cb < c_end; cb += 64) {
        int4 cur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        if (cb + 64 < c_end) load_boxes(cb + 64, nxt);
        unsigned long long qmask[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int4 bx = cur[k];
            const bool maybe = (cb + lane < c_end) && (bx.z >= tx0) && (bx.x < tx0 + TP) && (bx.w >= ty0) && (bx.y < ty0 + TPY);
            qmask[k] = __ballot(maybe);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
        unsigned long long cmask = qmask[k];
        while (cmask) {
            // A2: up to four hit quarters (quarter k of four chunks) at once: lane = (hit slot, row of its quarter)
            int hq = -1;
#pragma unroll
            for (int h = 0; h < 4; ++h)
                if (cmask) {                                   // wave-uniform
                    const int bit = (int)__builtin_ctzll(cmask);
                    cmask &= cmask - 1;
                    if ((lane >> 4) == h) hq = bit;
                }
            const int idx = hq < 0 ? total : (cb + hq) * 64 + k * 16 + (lane & 15);
            uint4 desc = make_uint4(0, 0, 0, 0);
            int flags = 0;
            if (idx < total) {
                const RowRef rf = row_of(idx, per, S, s_pow2, s_shift, rlo, b, vi, V, R, ray0, pixel_val, sec_grid);
                float x, y;
                level_xy(rf.g, rf.j, fW, fH, x, y);
                const float xf = floorf(x), yf = floorf(y);
                const int x0 = (int)xf, y0 = (int)yf;
                const float fx = x - xf, fy = y - yf;
                const int hx = x0 - tx0, hy = y0 - ty0;
                if (hx >= -1 && hx < TP && hy >= -1 && hy < TPY) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int xi = x0 + (k & 1), yi = y0 + (k >> 1);
                        const float wk = ((k & 1) ? fx : 1.0f - fx) * ((k >> 1) ? fy : 1.0f - fy);
                        const bool in_img = (xi >= 0) && (xi < Wl) && (yi >= 0) && (yi < Hl);
                        const bool in_tile = (xi >= tx0) && (xi < tx0 + TP) && (yi >= ty0) && (yi < ty0 + TPY);
                        if (in_img && in_tile && wk != 0.0f) flags |= 1 << k;
                    }
                    desc.x = rf.row;
                    desc.y = (unsigned)((hx + 1) | ((hy + 1) << 8) | (flags << 16));
                    desc.z = __float_as_uint(fx);
                    desc.w = __float_as_uint(fy);
                }
            }
            const unsigned long long mask = __ballot(flags != 0);
            if (mask) {
                const int pos =
                    __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                if (flags) queue[qn + pos] = desc;
                __builtin_amdgcn_wave_barrier();
                qn += __builtin_popcountll(mask);
                if (qn > QCAP - 64) { drain(qn); qn = 0; }
            }
        }
        }
    }
    if (qn) drain(qn);

    float* m = base + (size_t)img * Hl * Wl * C + slice * TC + lane;
#pragma unroll 4
    for (int pix = 0; pix < TP * TPY; ++pix) {
        const float v = tile[pix * TC + lane];
        const int gy = ty0 + (pix >> 3), gx = tx0 + (pix & 7);
        if (v != 0.0f && gy < Hl && gx < Wl) atomicAdd(m + ((size_t)gy * Wl + gx) * C, v);
    }
}

}  // namespace

extern "C" int cpn_attend_hidden_bwd(const uint16_t* qa, const uint16_t* qb, const uint16_t* hid, const float* at_wt,
                                     const float* dhbar, const float* dw_ext, int B, int V, int R, int S, int ray0,
                                     int nrays, uint16_t* dqa, uint16_t* dqb, uint16_t* dhid, const uint16_t* dqb_acc,
                                     void* stream) {
    CPN_REQUIRE(qa && qb && hid && at_wt && dhbar && dqa && dqb, CPN_E_ARG, "cpn_attend_hidden_bwd: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && V * S <= 2048, CPN_E_SHAPE, "cpn_attend_hidden_bwd: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_attend_hidden_bwd: ray range outside B*R");
    const size_t lds = (size_t)(HC + 2 * V * S + 4) * sizeof(float);
    hipLaunchKernelGGL(attend_hidden_bwd_kernel, dim3(nrays), dim3(256), lds, (hipStream_t)stream,
                       (const __half*)qa, (const __half*)qb, (const __half*)hid, at_wt, dhbar, dw_ext, V, R, S, ray0,
                       (__half*)dqa, (__half*)dqb, (__half*)dhid, (const __half*)dqb_acc);
    CPN_LAUNCH_CHECK("cpn_attend_hidden_bwd");
    return 0;
}

extern "C" int cpn_hid_grad_combine(const uint16_t* dkey, const uint16_t* hid, const float* w1, const float* dh1,
                                    const float* w2, const float* dh2, int B, int V, int R, int S, int ray0, int nrays,
                                    uint16_t* out, void* stream) {
    CPN_REQUIRE(hid && out && (!w1 || dh1) && (!w2 || dh2), CPN_E_ARG, "cpn_hid_grad_combine: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0, CPN_E_SHAPE, "cpn_hid_grad_combine: bad shape");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_hid_grad_combine: ray range outside B*R");
    const long long nchunks = (long long)nrays * V * S * 2 * 104;
    CPN_REQUIRE(nchunks / 256 < (1LL << 31), CPN_E_SHAPE, "cpn_hid_grad_combine: chunk too large");
    hipLaunchKernelGGL(hid_grad_combine_kernel, dim3((unsigned)cpn_cdiv(nchunks, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)dkey, (const __half*)hid, w1, dh1, w2, dh2, V, R, S, ray0, nchunks, (__half*)out);
    CPN_LAUNCH_CHECK("cpn_hid_grad_combine");
    return 0;
}

static int gather_rows_bwd_launch(const uint16_t* dxin, int ldx, int H, int W, const float* pixel_val, const float* sec_grid,
                                  int B, int V, int R, int S, int ray0, int nrays, float* dmap0, float* dmap1, float* dmap2,
                                  float* dmap3, int32_t* chunk_boxes, int first_level, int col3, void* stream) {
    const int maxchunks = (int)cpn_gather_bwd_chunks(R, S);
    const int nimg = B * V;
    const long long nwaves = (long long)nimg * maxchunks;
    const long long cand = 2LL * (long long)((nrays + B - 1) / B) * S;       // rows that read one image
    for (int l = first_level; l < 4; ++l) {
        GatherBwdPlan plan;
        const int shift = 4 - l - (l == 3);
        const int Hl = H >> shift, Wl = W >> shift;
        plan.lvl = l;
        plan.col0 = l == 3 ? col3 : l * 256;
        plan.maxchunks = maxchunks;
        plan.tiles_x = (Wl + TP - 1) / TP;
        plan.tiles = plan.tiles_x * ((Hl + TPY - 1) / TPY);
        plan.slices = (l == 3 ? 64 : 256) / TC;
        const long long hits = cand / plan.tiles;                            // expected rows landing on one tile
        plan.G = (int)std::min<long long>(64, std::max<long long>(1, (hits + 1023) / 2048));
        const int nroles = nimg * plan.G * plan.tiles * plan.slices;
        hipLaunchKernelGGL(gather_bbox_kernel, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, H, W,
                           pixel_val, sec_grid, V, R, S, ray0, nrays, maxchunks, nimg, l, (int4*)chunk_boxes);
        hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((nroles + WAVES - 1) / WAVES), dim3(64 * WAVES), 0,
                           (hipStream_t)stream, (const __half*)dxin, ldx, H, W, pixel_val, sec_grid, V, R, S, ray0,
                           nrays, dmap0, dmap1, dmap2, dmap3, plan, nroles, (const int4*)chunk_boxes);
    }
    return 0;
}

extern "C" int cpn_gather_rows_bwd_level3(const uint16_t* dxin, int ldx, int col0, int H, int W, const float* pixel_val,
                                          const float* sec_grid, int B, int V, int R, int S, int ray0, int nrays,
                                          float* dmap3, int32_t* chunk_boxes, void* stream) {
    CPN_REQUIRE(dxin && pixel_val && sec_grid && dmap3 && chunk_boxes, CPN_E_ARG, "cpn_gather_rows_bwd_level3: null pointer");
    CPN_REQUIRE(B > 0 && V == 2 && R > 0 && S > 0 && H >= 16 && (H % 16) == 0 && (W % 16) == 0 && col0 >= 0 && ldx >= col0 + 64,
                CPN_E_SHAPE, "cpn_gather_rows_bwd_level3: bad shape");
    CPN_REQUIRE(H <= 1024 && W <= 1024, CPN_E_SHAPE, "cpn_gather_rows_bwd_level3: maps larger than 1024 pixels a side");
    CPN_REQUIRE(ray0 >= 0 && nrays > 0 && (long long)ray0 + nrays <= (long long)B * R, CPN_E_ARG,
                "cpn_gather_rows_bwd_level3: ray range outside B*R");
    CPN_REQUIRE((long long)nrays * V * S * 2 < (1LL << 31), CPN_E_SHAPE, "cpn_gather_rows_bwd_level3: chunk too large");
    gather_rows_bwd_launch(dxin, ldx, H, W, pixel_val, sec_grid, B, V, R, S, ray0, nrays, dmap3, dmap3, dmap3, dmap3,
                           chunk_boxes, 3, col0, stream);
    CPN_LAUNCH_CHECK("cpn_gather_rows_bwd_level3");
    return 0;
}

extern "C" long long cpn_gather_bwd_chunks(int R, int S) { return (2LL * R * S + 63) / 64; }
