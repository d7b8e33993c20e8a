// Micro-benchmark: what HBM bandwidth does this box give a 9-streams-in / 1-stream-out fp32 merge
// (the access pattern of K1's box role) and a plain float4 copy?  hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o /tmp/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <int NR>
__global__ void merge4(const float4* __restrict__ a, size_t run_stride, float4* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = a[i + r * run_stride];
    float4 acc = v[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
    out[i] = acc;
}
// two tensors x NR runs, both merged and stored (K1's class role: logits + log-variances)
template <int NR, bool NT>
__global__ void merge4x2(const float4* __restrict__ a, const float4* __restrict__ b, size_t run_stride, float4* __restrict__ oa,
                         float4* __restrict__ ob, size_t n) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f4 va[NR], vb[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        va[r] = NT ? __builtin_nontemporal_load((const f4*)(a + i + r * run_stride)) : *(const f4*)(a + i + r * run_stride);
        vb[r] = NT ? __builtin_nontemporal_load((const f4*)(b + i + r * run_stride)) : *(const f4*)(b + i + r * run_stride);
    }
    f4 x = va[0], y = vb[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) { x += va[r]; y += vb[r]; }
    *(f4*)(oa + i) = x;
    *(f4*)(ob + i) = y;
}
// the same 2 x NR-stream merge with LDS-DMA loads (global_load_lds_dwordx4: global -> LDS without staging VGPRs; the LDS
// destination of a wave's instruction is base + lane * 16), then one ds_read_b128 per stream
template <int NR, int AUX>
__global__ void __launch_bounds__(256) merge4x2_lds(const float4* __restrict__ a, const float4* __restrict__ b, size_t run_stride,
                                                    float4* __restrict__ oa, float4* __restrict__ ob, size_t n) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char lds[];     // [wave][2*NR streams][64 lanes] x 16 B
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* base = lds + (size_t)wave * (2 * NR) * 64 * 16;
    const size_t ii = i < n ? i : n - 1;                           // every lane issues (the LDS slot is positional)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a + ii + r * run_stride),
                                         (__attribute__((address_space(3))) void*)(base + (size_t)(2 * r) * 64 * 16), 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + ii + r * run_stride),
                                         (__attribute__((address_space(3))) void*)(base + (size_t)(2 * r + 1) * 64 * 16), 16, 0, AUX);
    }
    __builtin_amdgcn_s_waitcnt(0);       // vmcnt(0): the DMA writes have landed in LDS
    __builtin_amdgcn_wave_barrier();
    if (i >= n) return;
    f4 x = *(const f4*)(base + (size_t)lane * 16), y = *(const f4*)(base + (size_t)64 * 16 + (size_t)lane * 16);
#pragma unroll
    for (int r = 1; r < NR; ++r) {
        x += *(const f4*)(base + (size_t)(2 * r) * 64 * 16 + (size_t)lane * 16);
        y += *(const f4*)(base + (size_t)(2 * r + 1) * 64 * 16 + (size_t)lane * 16);
    }
    __builtin_nontemporal_store(x, (f4*)(oa + i));
    __builtin_nontemporal_store(y, (f4*)(ob + i));
}
template <int NR>
__global__ void merge4_gs(const float4* __restrict__ a, size_t run_stride, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = a[i + r * run_stride];
        float4 acc = v[0];
#pragma unroll
        for (int r = 1; r < NR; ++r) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
        out[i] = acc;
    }
}
int main() {
    const size_t run_elems = (size_t)193374 * 22 / 4;       // float4 per run (17 MB)
    const int NR = 9, SETS = 4, IT = 40;
    std::vector<float4*> in(SETS), out(SETS);
    for (int s = 0; s < SETS; ++s) { CK(hipMalloc(&in[s], run_elems * 16 * (NR + 1))); CK(hipMalloc(&out[s], run_elems * 16)); CK(hipMemset(in[s], 1, run_elems * 16 * (NR + 1))); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 4; ++i) launch(i % SETS);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < IT; ++i) launch(i % SETS);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.2f us/launch  %7.1f GB/s\n", name, 1e3 * ms / IT, bytes / (ms / IT * 1e-3) / 1e9);
        return 0;
    };
    const double mbytes = (double)run_elems * 16 * (NR + 1);
    for (int threads : {256, 512, 1024}) {
        char nm[64];
        snprintf(nm, 64, "merge 9 streams (1 f4/thr, %d thr)", threads);
        timeit(nm, [&](int s) { merge4<9><<<(run_elems + threads - 1) / threads, threads>>>(in[s], run_elems, out[s], run_elems); }, mbytes);
    }
    for (int blocks : {1024, 2048, 4096}) {
        char nm[64];
        snprintf(nm, 64, "merge 9 streams grid-stride %d blk", blocks);
        timeit(nm, [&](int s) { merge4_gs<9><<<blocks, 256>>>(in[s], run_elems, out[s], run_elems); }, mbytes);
    }
    // size dependence: the same merge on the 2K = 14 class channels only (108 MB, the product path's K1) and on 4x the data
    {
        const size_t re14 = (size_t)193374 * 14 / 4;
        timeit("merge 9 streams, 14 ch (108 MB)", [&](int s) { merge4<9><<<(re14 + 255) / 256, 256>>>(in[s], run_elems, out[s], re14); }, (double)re14 * 16 * 10);
        {   // 2 x 9 streams of 7 channels each, run stride = one 7-channel plane set (the real (N, A*K, H, W) layout)
            const size_t re7 = (size_t)193374 * 7 / 4;
            float4* b2 = in[0] + re7 * 10;   // second tensor right behind the first (10 runs each)
            timeit("merge 2x9 streams, 7+7 ch, nt", [&](int s) { merge4x2<9, true><<<(re7 + 255) / 256, 256>>>(in[s], in[s] + re7 * 10, re7, out[s], out[s] + re7, re7); }, (double)re7 * 16 * 20);
            timeit("merge 2x9 streams, 7+7 ch", [&](int s) { merge4x2<9, false><<<(re7 + 255) / 256, 256>>>(in[s], in[s] + re7 * 10, re7, out[s], out[s] + re7, re7); }, (double)re7 * 16 * 20);
            timeit("merge 2x9 streams, LDS-DMA", [&](int s) { merge4x2_lds<9, 0><<<(re7 + 127) / 128, 128, 2 * 18 * 64 * 16>>>(in[s], in[s] + re7 * 10, re7, out[s], out[s] + re7, re7); }, (double)re7 * 16 * 20);
            timeit("merge 2x9 streams, LDS-DMA nt", [&](int s) { merge4x2_lds<9, 2><<<(re7 + 127) / 128, 128, 2 * 18 * 64 * 16>>>(in[s], in[s] + re7 * 10, re7, out[s], out[s] + re7, re7); }, (double)re7 * 16 * 20);
            timeit("merge 9 streams, 14 ch, stride 14", [&](int s) { merge4<9><<<(re14 + 255) / 256, 256>>>(in[s], re14, out[s], re14); }, (double)re14 * 16 * 10);
            (void)b2;
        }
        float4 *big, *bigo;
        const size_t reb = run_elems * 4;
        CK(hipMalloc(&big, reb * 16 * 10)); CK(hipMalloc(&bigo, reb * 16)); CK(hipMemset(big, 1, reb * 16 * 10));
        timeit("merge 9 streams, 88 ch (681 MB)", [&](int) { merge4<9><<<(reb + 255) / 256, 256>>>(big, reb, bigo, reb); }, (double)reb * 16 * 10);
        timeit("empty-ish launch (1 block)", [&](int s) { copy4<<<1, 64>>>(in[s], out[s], 1); }, 0);
        hipFree(big); hipFree(bigo);
    }
    const size_t cn = run_elems * 5;    // 85 MB read + 85 MB write
    timeit("copy float4 (85 MB -> 85 MB)", [&](int s) { copy4<<<(cn + 255) / 256, 256>>>(in[s], in[s] + cn, cn); }, (double)cn * 32);
    timeit("merge 2 streams", [&](int s) { merge4<2><<<(run_elems * 4 + 255) / 256, 256>>>(in[s], run_elems * 4, out[s], run_elems); }, 0);
    return 0;
}
