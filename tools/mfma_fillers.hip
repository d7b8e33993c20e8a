// GPU box micro-benchmark: what does one NON-MFMA instruction cost beside fp32 MFMAs when a SIMD holds one 512-register
// wavefront (the regime pod_wino_conv3x3's K loop runs in)?   hipcc --offload-arch=gfx950 -O3 tools/mfma_fillers.hip -o /tmp/mf && /tmp/mf
//
// One "chunk" = 48 v_mfma_f32_32x32x2_f32 on 12 independent accumulators (192 AGPRs), exactly K11's chunk, with a chosen number
// of fillers pinned behind the MFMAs (sched_barrier after every slot): ds_read_b128 / b64, buffer_load_dwordx4 from an
// L2-resident slab, LDS-DMA pieces, packed or scalar fp32 VALU, a workgroup barrier.  Prints shader cycles per chunk (s_memtime of
// wavefront 0 of every workgroup, median over the 256 workgroups) and the wall time per chunk from HIP events.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Cfg {
    int nr;      // ds_read_b128 per chunk (behind MFMA 0 .. nr-1)
    int rw;      // bytes per lane of a read: 16 (b128) or 8 (two b64 reads stand for one b128 of the same bytes -> 2 nr reads)
    int conf;    // 1: K11's patch addresses (2-way bank conflict), 0: lane * 16 B
    int nf;      // buffer_load_dwordx4 per chunk (behind MFMA 12 ..)
    int nd;      // LDS-DMA pieces per chunk (behind MFMA 24 ..)
    int nv;      // packed VALU per chunk, 4 per MFMA slot from slot 28 on
    int vk;      // 0: v_pk_fma_f32, 1: two v_fma_f32 per packed op
    int bar;     // 1: __syncthreads() per chunk
    int waves;   // wavefronts per workgroup that run the loop (1, 2 or 4; the others exit)
    int pos;     // 0: filler right after the MFMA, 1: two fillers behind every second MFMA
};

template <int NR, int RW, int CONF, int NF, int ND, int NV, int VK, int BAR, int WAVES, int POS, int DPAT = 0, int NW = 0>
__global__ void __launch_bounds__(256, 1) k_fill(const float* __restrict__ gsrc, float* __restrict__ sink, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) lds[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    if (wave >= WAVES) return;
    const int i32 = lane & 31, h = lane >> 5;
    const int a_base = CONF ? (h * 360 + 2 * (i32 >> 2) * 20 + 2 * (i32 & 3)) * 4 : lane * 4 + wave * 1024;
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 0x7FFFFFF0, 0x00020000);
    const int u_off = (wave * 12 * 64 + lane) * 16;
    typedef __attribute__((address_space(3))) void lds_void;
    // DMA address pattern: 0 lane * 16 B (8 lines per instruction), 1 every lane its own 128-B line 1 KB apart (K11's patch: a pixel
    // per lane, C = 256), 2 eight lanes per line, lines 1 KB apart (a pixel's 32 channels per 8 lanes), 3 two lanes per line
    const int d_off = DPAT == 0 ? lane * 16 : DPAT == 1 ? lane * 1024 + (lane & 1) * 16 : DPAT == 2 ? (lane >> 3) * 1024 + (lane & 7) * 16 : (lane >> 1) * 1024 + (lane & 1) * 16;

    // 4: K11's real pattern from HBM -- a pixel (1 KB apart) per lane, 32 useful bytes of its line per chunk, the same lines again in
    // the next 3 chunks, every workgroup its own 6 MB; 5: the same bytes as full 128-B lines, 8 lanes per pixel, each line once
    const int cold_off = DPAT == 4 ? lane * 1024 + (lane & 1) * 16 : (lane >> 3) * 1024 + (lane & 7) * 16;
    f32x16 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 x[24], uA[12], uB[12], vA[6], vB[6];
#pragma unroll
    for (int i = 0; i < 24; ++i) x[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 12; ++i) uA[i] = uB[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane - i);
#pragma unroll
    for (int i = 0; i < 6; ++i) vA[i] = vB[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane + 3 * i);

    auto pk_fma = [](f32x2 k2, f32x2 q, f32x2 p) {
        f32x2 r;
        if (VK == 0) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(k2), "v"(q), "v"(p));
        } else {
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(k2.x), "v"(q.x), "v"(p.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(k2.y), "v"(q.y), "v"(p.y));
        }
        return r;
    };
    constexpr int NRI = RW == 16 ? NR : 2 * NR;     // read instructions per chunk
    auto chunk = [&](int ch, f32x4(&vC)[6], f32x4(&uC)[12], f32x4(&vN)[6], f32x4(&uN)[12]) {
        int vdone = 0;
#pragma unroll
        for (int j = 0; j < 48; ++j) {
            acc[j % 12] = __builtin_amdgcn_mfma_f32_32x32x2f32(vC[(j % 12) >> 1][j / 12], uC[j % 12][j / 12], acc[j % 12], 0, 0, 0);
            if (POS == 1 && (j & 1) == 0) {
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            const int lo = POS == 1 ? j - 1 : j, hi = j;
#pragma unroll
            for (int s = lo; s <= hi; ++s) {
                if (s < NRI) {
                    if (RW == 16) x[s] = *reinterpret_cast<const f32x4*>(lds + a_base + (s % 6) * 8 + (s / 6) * 80);
                    else {
                        const f32x2 t = *reinterpret_cast<const f32x2*>(lds + a_base + (s % 12) * 2 + (s / 12) * 80);
                        x[s >> 1][(s & 1) * 2] = t.x;
                        x[s >> 1][(s & 1) * 2 + 1] = t.y;
                    }
                } else if (s >= NRI && s - NRI < NF && NRI <= 24) {
                    const int i = s - NRI;
                    uN[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, (ch & 31) * 49152 + i * 1024, 0));
                } else if (s - NRI - NF >= 0 && s - NRI - NF < ND) {
                    const int i = s - NRI - NF;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, (lds_void*)(lds + 4096 + (wave * 3 + i) * 256), 16,
                                                             DPAT >= 4 ? cold_off + (i + wave * 3) * (DPAT == 4 ? 65536 : 8192) : d_off + (DPAT ? (i * 64 + wave * 192) * 1024 : i * 1024),
                                                             DPAT == 4 ? (int)blockIdx.x * (6 << 20) + ((ch >> 2) & 7) * (768 << 10) + (ch & 3) * 32
                                                             : DPAT == 5 ? (int)blockIdx.x * (6 << 20) + (ch & 63) * (96 << 10)
                                                                         : (DPAT ? (ch & 7) * 128 : (ch & 31) * 49152) + (3 << 20), 0, 0);
                } else if (s >= (POS == 2 ? 4 : 28) && vdone < NV) {
#pragma unroll
                    for (int q = 0; q < (POS == 2 ? 1 : POS == 3 ? 8 : 4); ++q)
                        if (vdone < NV) {
                            const int e = vdone % 12;
                            const f32x2 r = pk_fma(f32x2{0.5f, 0.25f}, f32x2{x[e].x, x[e].y}, f32x2{x[e + 12].z, x[e + 12].w});
                            vN[e % 6][(e / 6) * 2] = r.x;
                            vN[e % 6][(e / 6) * 2 + 1] = r.y;
                            ++vdone;
                        }
                }
            }
            if (j >= 40 && j - 40 < NW) *reinterpret_cast<f32x4*>(lds + 6144 + wave * 1024 + (j - 40) * 256 + lane * 4) = uC[j - 40];   // stage a loaded piece by hand
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
        else {
#pragma unroll
            for (int i = 0; i < 24; ++i) asm volatile("" ::"v"(x[i]));
#pragma unroll
            for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(uN[i]));
            if (ND) __builtin_amdgcn_s_waitcnt(0);
        }
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int ch = 0; ch < iters; ch += 2) {
        chunk(ch, vA, uA, vB, uB);
        chunk(ch + 1, vB, uB, vA, uA);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 24; ++i) s += x[i].x;
    sink[blockIdx.x * 256 + tid] = s + lds[4096 + lane];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// the same per-SIMD work cut over TWO wavefronts per SIMD (8 per workgroup, 96 accumulators each): is a filler issued by one
// wavefront hidden behind the other's MFMAs?
template <int NR, int NF, int NV>
__global__ void __launch_bounds__(512, 1) k_fill2(const float* __restrict__ gsrc, float* __restrict__ sink, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 512) lds[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    const int a_base = lane * 4 + (wave & 3) * 1024;
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 6 << 20, 0x00020000);
    const int u_off = (wave * 6 * 64 + lane) * 16;
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 x[12], uA[6], uB[6], vA[6], vB[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 6; ++i) uA[i] = uB[i] = vA[i] = vB[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane - i);
    auto chunk = [&](int ch, f32x4(&vC)[6], f32x4(&uC)[6], f32x4(&vN)[6], f32x4(&uN)[6]) {
        int vdone = 0;
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            acc[j % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(vC[j % 6][j / 6], uC[j % 6][j / 6], acc[j % 6], 0, 0, 0);
            if (j < NR) x[j] = *reinterpret_cast<const f32x4*>(lds + a_base + (j % 6) * 8 + (j / 6) * 80);
            else if (j - NR < NF) uN[j - NR] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, (ch & 31) * 49152 + (j - NR) * 1024, 0));
            else if (vdone < NV) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (vdone < NV) {
                        const int e = vdone % 6;
                        f32x2 r;
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(f32x2{0.5f, 0.25f}), "v"(f32x2{x[e].x, x[e].y}), "v"(f32x2{x[e + 6].z, x[e + 6].w}));
                        vN[e][(vdone / 6) & 1 ? 2 : 0] = r.x;
                        vN[e][(vdone / 6) & 1 ? 3 : 1] = r.y;
                        ++vdone;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(x[i]));
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" ::"v"(uN[i]));
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int ch = 0; ch < iters; ch += 2) {
        chunk(ch, vA, uA, vB, uB);
        chunk(ch + 1, vB, uB, vA, uA);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += x[i].x;
    sink[blockIdx.x * 512 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static float* g_src;
static float* g_sink;
static long long* g_cyc;

template <typename F>
static void run(const char* name, F kern, int threads, double mfma_per_chunk_simd) {
    const int iters = 2000, grid = 256, lds_bytes = 64 * 1024 + 72 * 1024;   // > 80 KB: one workgroup per CU
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, 0, g_src, g_sink, g_cyc, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds_bytes, 0, g_src, g_sink, g_cyc, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(grid);
    CK(hipMemcpy(c.data(), g_cyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double med = (double)c[grid / 2] / iters, mx = (double)c[grid - 1] / iters;
    printf("%-58s cycles/chunk med %7.1f max %7.1f | wall us/chunk %.4f | extra vs %4.0f: %6.1f\n", name, med, mx, ms * 1e3 / iters,
           mfma_per_chunk_simd * 64, med - mfma_per_chunk_simd * 64);
    fflush(stdout);
}

#define RUND(NR, RW, CONF, NF, ND, NV, VK, BAR, WAVES, POS, DPAT) \
    run("nr=" #NR " rw=" #RW " conf=" #CONF " nf=" #NF " nd=" #ND " nv=" #NV " vk=" #VK " bar=" #BAR " waves=" #WAVES " pos=" #POS " dpat=" #DPAT, k_fill<NR, RW, CONF, NF, ND, NV, VK, BAR, WAVES, POS, DPAT>, 256, 48)
#define RUN(NR, RW, CONF, NF, ND, NV, VK, BAR, WAVES, POS) \
    run("nr=" #NR " rw=" #RW " conf=" #CONF " nf=" #NF " nd=" #ND " nv=" #NV " vk=" #VK " bar=" #BAR " waves=" #WAVES " pos=" #POS, k_fill<NR, RW, CONF, NF, ND, NV, VK, BAR, WAVES, POS>, 256, 48)

int main() {
    CK(hipMalloc(&g_src, (size_t)1600 << 20));
    CK(hipMemset(g_src, 0, (size_t)1600 << 20));
    CK(hipMalloc(&g_sink, 256 * 512 * 4));
    CK(hipMalloc(&g_cyc, 256 * 8));
    // bare MFMAs
    RUN(0, 16, 0, 0, 0, 0, 0, 0, 4, 0);
    RUN(0, 16, 0, 0, 0, 0, 0, 0, 1, 0);
    RUN(0, 16, 0, 0, 0, 0, 0, 1, 4, 0);
    // LDS reads
    RUN(12, 16, 0, 0, 0, 0, 0, 0, 4, 0);
    RUN(12, 16, 0, 0, 0, 0, 0, 0, 1, 0);
    RUN(12, 16, 1, 0, 0, 0, 0, 0, 4, 0);
    RUN(24, 16, 0, 0, 0, 0, 0, 0, 4, 0);
    RUN(24, 16, 0, 0, 0, 0, 0, 0, 1, 0);
    RUN(12, 8, 0, 0, 0, 0, 0, 0, 4, 0);
    RUN(12, 16, 0, 0, 0, 0, 0, 0, 4, 1);
    // filter loads from L2
    RUN(0, 16, 0, 12, 0, 0, 0, 0, 4, 0);
    RUN(0, 16, 0, 12, 0, 0, 0, 0, 1, 0);
    RUN(0, 16, 0, 12, 0, 0, 0, 0, 4, 1);
    // LDS-DMA
    RUN(0, 16, 0, 0, 3, 0, 0, 0, 4, 0);
    RUN(0, 16, 0, 0, 3, 0, 0, 1, 4, 0);
    RUND(0, 16, 0, 0, 3, 0, 0, 1, 4, 0, 1);
    RUND(0, 16, 0, 0, 3, 0, 0, 1, 4, 0, 2);
    RUND(0, 16, 0, 0, 3, 0, 0, 1, 4, 0, 3);
    RUND(12, 16, 1, 12, 3, 40, 0, 1, 4, 0, 1);
    RUND(12, 16, 1, 12, 3, 40, 0, 1, 4, 0, 2);
    RUND(12, 16, 1, 12, 3, 40, 0, 1, 4, 0, 3);
    RUND(0, 16, 0, 0, 3, 0, 0, 1, 4, 0, 4);
    RUND(0, 16, 0, 0, 3, 0, 0, 1, 4, 0, 5);
    RUND(12, 16, 1, 12, 3, 40, 0, 1, 4, 0, 4);
    RUND(12, 16, 1, 12, 3, 40, 0, 1, 4, 0, 5);
    run("4 ds_write_b128 per chunk beside bare MFMAs", k_fill<0, 16, 0, 0, 0, 0, 0, 1, 4, 0, 0, 4>, 256, 48);
    run("4 extra loads + 4 ds_write_b128 beside bare MFMAs", k_fill<0, 16, 0, 4, 0, 0, 0, 1, 4, 0, 0, 4>, 256, 48);
    run("K11 mix, patch by 4 loads + 4 ds_write_b128 instead of DMA", k_fill<12, 16, 1, 16, 0, 40, 0, 1, 4, 0, 0, 4>, 256, 48);
    run("K11 mix, patch by 4 DMA pieces (hot)", k_fill<12, 16, 1, 12, 4, 40, 0, 1, 4, 0, 2, 0>, 256, 48);
    run("K11 mix, patch by 4 DMA pieces (cold, full lines)", k_fill<12, 16, 1, 12, 4, 40, 0, 1, 4, 0, 5, 0>, 256, 48);
    // VALU
    RUN(0, 16, 0, 0, 0, 40, 0, 0, 4, 0);
    RUN(0, 16, 0, 0, 0, 40, 1, 0, 4, 0);
    RUN(0, 16, 0, 0, 0, 40, 0, 0, 1, 0);
    RUN(0, 16, 0, 0, 0, 40, 0, 0, 4, 2);
    RUN(0, 16, 0, 0, 0, 40, 0, 0, 4, 3);
    RUN(0, 16, 0, 0, 0, 40, 1, 0, 4, 2);
    RUN(0, 16, 0, 0, 0, 20, 0, 0, 4, 0);
    RUN(0, 16, 0, 0, 0, 20, 0, 0, 4, 2);
    // K11's mix
    RUN(12, 16, 1, 12, 3, 40, 0, 1, 4, 0);
    RUN(12, 16, 0, 12, 3, 40, 0, 1, 4, 0);
    RUN(12, 16, 1, 12, 3, 40, 0, 0, 4, 0);
    RUN(12, 16, 1, 12, 0, 40, 0, 0, 4, 0);
    RUN(12, 16, 1, 12, 3, 40, 0, 1, 1, 0);
    RUN(12, 16, 1, 12, 3, 40, 1, 1, 4, 0);
    // two wavefronts per SIMD, half the work each
    run("2 waves/SIMD: bare", k_fill2<0, 0, 0>, 512, 48);
    run("2 waves/SIMD: 6 reads each", k_fill2<6, 0, 0>, 512, 48);
    run("2 waves/SIMD: 6 filter loads each", k_fill2<0, 6, 0>, 512, 48);
    run("2 waves/SIMD: 20 pk each", k_fill2<0, 0, 20>, 512, 48);
    run("2 waves/SIMD: 6 reads + 6 loads + 20 pk each", k_fill2<6, 6, 20>, 512, 48);
    return 0;
}
