// GPU box micro-benchmark: how many bytes per shader clock ONE CU pulls from its XCD's L2 into VGPRs (buffer_load_dwordx4, 1 KiB per wavefront
// instruction, fully coalesced), one or two wavefronts per SIMD, region sizes that live in L1 (16 KiB per workgroup), L2 (2 MiB) or beyond (512 MiB).
// K12 / K13 stream their filter terms this way (96 KiB per 16-channel chunk and workgroup in K12): is that path 64 or 128 B/clk?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_to_vgpr_bw tools/l2_to_vgpr_bw.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ void __launch_bounds__(256) k(const unsigned* __restrict__ src, unsigned region_bytes, unsigned* __restrict__ sink, long long* __restrict__ cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, region_bytes, 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    // workgroup b walks the region in 1-KiB wavefront pieces starting at a workgroup-specific offset; the four wavefronts read different pieces
    unsigned off = ((blockIdx.x * 4 + wave) * 16 * 1024u + lane * 16u) % region_bytes;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
            off += 1024u;
            if (off >= region_bytes) off -= region_bytes;
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) acc ^= v[i];
    }
    const long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F>
static void run(const char* name, F kern, int blocks, unsigned region, const unsigned* src, unsigned* sink, long long* cyc, int inflight) {
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, region, sink, cyc, iters);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, region, sink, cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(blocks);
    (void)hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double bytes_wg = 4.0 * inflight * 1024.0 * iters;
    printf("%-44s %4d workgroups: %6.1f B/clk per workgroup (median)   %7.2f TB/s whole launch\n", name, blocks, bytes_wg / (double)c[blocks / 2], bytes_wg * blocks / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned *src, *sink; long long* cyc;
    (void)hipMalloc(&src, 512u << 20); (void)hipMemset(src, 1, 512u << 20); (void)hipMalloc(&sink, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
    for (unsigned region : {16u << 10, 2u << 20, 16u << 20, 512u << 20}) {
        char name[96];
        snprintf(name, sizeof name, "region %6u KiB, 8 loads in flight", region >> 10);
        run(name, k<8>, 256, region, src, sink, cyc, 8);
        snprintf(name, sizeof name, "region %6u KiB, 16 loads in flight", region >> 10);
        run(name, k<16>, 256, region, src, sink, cyc, 16);
        snprintf(name, sizeof name, "region %6u KiB, 16 in flight, 2 WG/CU", region >> 10);
        run(name, k<16>, 512, region, src, sink, cyc, 16);
    }
    return 0;
}
