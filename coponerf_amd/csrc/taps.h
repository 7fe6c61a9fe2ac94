// Bilinear tap set of one feature level, exactly ATen's grid_sampler_2d with align_corners=False
// (/root/reference models/CoPoNeRF.py:312 'border', :370 'zeros').  Shared by the forward gathers (gather.hip,
// encode.hip) and the scatter of the training backward (backward.hip) so that all three agree bit for bit on the
// texel indices.
#pragma once
#include <hip/hip_runtime.h>

struct Taps {
    int off[4];      // texel offsets (in texels) of nw, ne, sw, se; clamped into the map
    float w[4];      // weights; 0 for out-of-map taps (zeros padding)
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int Wl, int Hl, bool border) {
    float x = ((gx + 1.0f) * (float)Wl - 1.0f) / 2.0f;
    float y = ((gy + 1.0f) * (float)Hl - 1.0f) / 2.0f;
    if (border) {
        x = fminf(fmaxf(x, 0.0f), (float)(Wl - 1));
        y = fminf(fmaxf(y, 0.0f), (float)(Hl - 1));
    } else {
        // |coordinate| can reach 1e10 (geometry.py:390-391): keep the int conversion defined; anything
        // beyond one texel outside the map has all four taps out of range anyway.
        x = fminf(fmaxf(x, -2.0f), (float)Wl + 1.0f);
        y = fminf(fmaxf(y, -2.0f), (float)Hl + 1.0f);
    }
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const float fx = x - xf, fy = y - yf;
    Taps t;
    const float wx[2] = {1.0f - fx, fx}, wy[2] = {1.0f - fy, fy};
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int xi = x0 + i, yi = y0 + j;
            const bool inside = (xi >= 0) && (xi <= Wl - 1) && (yi >= 0) && (yi <= Hl - 1);
            const int xc = min(max(xi, 0), Wl - 1), yc = min(max(yi, 0), Hl - 1);
            t.off[j * 2 + i] = yc * Wl + xc;
            t.w[j * 2 + i] = inside ? wx[i] * wy[j] : 0.0f;
        }
    return t;
}
