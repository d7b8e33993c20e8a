// GPU box: how far into the 160 KB LDS can an LDS-DMA (buffer_load_dwordx4 ... lds, base in M0) write on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void __launch_bounds__(64) k(const float* src, int* out, int n_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) lds[i] = -1.0f;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
    for (int t = 0; t < n_off; ++t) {
        const int kb = 8 + 8 * t;                       // destination: kb KiB into the LDS
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + kb * 256), 16, lane * 16, t * 1024, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int t = 0; t < n_off; ++t) {
        const int kb = 8 + 8 * t;
        const float v = lds[kb * 256 + lane * 4];       // expect src[t * 256 + lane * 4]
        if (lane == 0) out[t] = (int)v;
    }
}
int main() {
    float* src; int* out;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, 256);
    float h[20 * 256];
    for (int i = 0; i < 20 * 256; ++i) h[i] = 1000.0f + i / 256;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, src, out, 19);
    int r[19];
    hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
    for (int t = 0; t < 19; ++t) printf("dst %3d KiB: read back %d (want %d)\n", 8 + 8 * t, r[t], 1000 + t);
    return 0;
}
