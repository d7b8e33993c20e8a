// GPU box micro-benchmark (round 5): what ONE instruction of each kind costs in the shadow of v_mfma_f32_32x32x16_f16, one wavefront per SIMD --
// the instruction kinds of K12's split loop (k12_wino_conv_split.hip): plain fp32 VALU (v_fma_f32), the f16 "mix" family of the 2-way split
// (v_fma_mixlo_f16 / v_fma_mixhi_f16, v_fma_mix_f32), v_cvt_pk_f16_f32, and a packed fp32 op for reference.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_fillers tools/mfma_f16_fillers.hip && /tmp/mfma_f16_fillers
// A "chunk" = 36 MFMAs on 12 accumulators (K12's 6 positions x 2 channel blocks x 3 products) with N independent fillers spread evenly behind them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum Kind { FMA = 0, MIX_F32 = 1, MIXLO = 2, MIXHI = 3, CVT_PK = 4, PK_FMA = 5, MIXPAIR = 6, SUB = 7 };

template <int KIND>
__device__ __forceinline__ void filler(float (&r)[16], unsigned (&w)[8], int n, float sv) {
    const int a = n % 16, b = (n + 5) % 16, c = (n + 11) % 16, d = n % 8;
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[a]) : "v"(r[b]), "v"(sv), "v"(r[c]));
    else if constexpr (KIND == SUB) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r[a]) : "v"(r[b]), "v"(r[c]));
    else if constexpr (KIND == MIX_F32) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[a]) : "v"(r[b]), "v"(sv), "v"(w[d]));
    else if constexpr (KIND == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(w[d]) : "v"(r[b]), "v"(sv));
    else if constexpr (KIND == MIXHI) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(w[d]) : "v"(r[b]), "v"(sv));
    else if constexpr (KIND == MIXPAIR) {
        if (n & 1) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(w[d]) : "v"(r[b]), "v"(sv));
        else asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(w[d]) : "v"(r[b]), "v"(sv));
    } else if constexpr (KIND == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[d]) : "v"(r[b]), "v"(r[c]));
    else if constexpr (KIND == PK_FMA) {
        f32x2 o;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(f32x2{r[b], r[c]}), "v"(f32x2{sv, sv}), "v"(f32x2{r[(n + 3) % 16], r[(n + 7) % 16]}));
        r[a] = o.x;
    }
}

template <int KIND, int NFILL>
__global__ void __launch_bounds__(256, 1) k(float* __restrict__ sink, long long* __restrict__ cyc, int iters, float sv) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 U[4], V[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) U[i] = u32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane};
#pragma unroll
    for (int i = 0; i < 2; ++i) V[i] = u32x4{0x3c003c00u + i, 0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u};
    float r[16];
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 1e-3f * (lane + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = 0x3c003c00u + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int done = 0;
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            const int p = j / 6, m = j % 6, kb = m & 1, prod = m >> 1;
            acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, U[kb * 2 + (prod == 1)]), __builtin_bit_cast(f16x8, V[prod == 0]), acc[p * 2 + kb], 0, 0, 0);
            constexpr int u0 = 0;
            const int want = (j + 1) * NFILL / 36;
#pragma unroll
            for (int q = 0; q < (NFILL + 35) / 36 + 1; ++q)
                if (done < want) { filler<KIND>(r, w, done, sv); ++done; }
            (void)u0;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc[i][q];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)w[i];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F>
static double run(const char* name, int nfill, F kern, float* sink, long long* cyc, double base) {
    const int iters = 4000;
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, sink, cyc, iters, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, sink, cyc, iters, 0.5f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(256);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per = (double)c[128] / iters;
    printf("%-34s x%3d  cycles / chunk %7.1f  (+%6.1f = %5.2f per filler)   wall ns / chunk %7.1f\n", name, nfill, per, per - base, nfill ? (per - base) / nfill : 0.0, ms * 1e6 / iters);
    return per;
}
#define ROW(K, name) run(name, 72, k<K, 72>, sink, cyc, base); run(name, 144, k<K, 144>, sink, cyc, base); run(name, 288, k<K, 288>, sink, cyc, base);
int main() {
    float* sink; long long* cyc;
    hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const double base = run("36 f16 MFMAs alone", 0, k<FMA, 0>, sink, cyc, 0.0);
    ROW(FMA, "v_fma_f32")
    ROW(SUB, "v_sub_f32")
    ROW(MIX_F32, "v_fma_mix_f32")
    ROW(MIXLO, "v_fma_mixlo_f16")
    ROW(MIXPAIR, "v_fma_mixlo/hi_f16 alternating")
    ROW(CVT_PK, "v_cvt_pk_f16_f32")
    ROW(PK_FMA, "v_pk_fma_f32")
    return 0;
}
