#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef short short4_t __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int stride_bytes) {
    __shared__ short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    __attribute__((address_space(3))) short4_t* p =
        (__attribute__((address_space(3))) short4_t*)((__attribute__((address_space(3))) char*)lds + lane * stride_bytes);
    short4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {8, 32, 64}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes\n", stride);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" %4d", h[l * 4 + e]); printf("\n"); }
    }
    return 0;
}
