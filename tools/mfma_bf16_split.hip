// GPU box micro-benchmark: the K loop of a 3-way-bf16-split Winograd chunk (16 input channels): 72 v_mfma_f32_32x32x16_bf16 on 12
// accumulators (6 positions x 2 channel blocks x 6 partial products) with the chunk's other work slotted behind them -- 24
// ds_read_b128 (raw patch), 36 buffer_load_dwordx4 (pre-split filters, L2), the fp32 input transform (80 packed ops) and the 3-way
// split of the 48 transformed values (truncation split: and / packed sub / perm).  How many cycles per chunk?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NVALU, int NREAD, int NLOAD>
__global__ void __launch_bounds__(256, 1) k(const float* __restrict__ gsrc, float* __restrict__ sink, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) lds[i] = 1.0f + 1e-3f * i;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 16 << 20, 0x00020000);
    const int u_off = (wave * 36 * 64 + lane) * 16;
    f32x16 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 x[8];
    u32x4 uP[2][6], vC[18], vN[18];       // filters of the current / next POSITION: [kb][split]; V: [p][split], current / next chunk
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 6; ++i) uP[0][i] = uP[1][i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane};
#pragma unroll
    for (int i = 0; i < 18; ++i) vC[i] = vN[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane};
    auto valu = [&](u32x4(&vn)[18], int n) {      // a 3-op unit: packed fma (transform), and + perm (split / pack) -- independent of the running MFMAs
        const int e = n % 8;
#ifdef DEPENDENT
        f32x2 r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(f32x2{0.5f, 0.25f}), "v"(f32x2{x[e].x, x[e].y}), "v"(f32x2{x[(e + 5) % 8].z, x[(e + 5) % 8].w}));
        unsigned m;
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m) : "v"(__builtin_bit_cast(unsigned, r.x)));
        unsigned p;
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p) : "v"(m), "v"(__builtin_bit_cast(unsigned, r.y)), "v"(0x07060302u));
        vn[(n / 3) % 18][(n / 3 / 18) % 4] = p;
#else
        // three INDEPENDENT ops (what a software-pipelined split looks like: every op's inputs were produced slots ago)
        unsigned m0, m1, p;
#if PK
        f32x2 r;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(f32x2{x[e].x, x[e].y}), "v"(f32x2{x[(e + 5) % 8].z, x[(e + 5) % 8].w}));
        m0 = __builtin_bit_cast(unsigned, r.x);
#else
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m0) : "v"(__builtin_bit_cast(unsigned, x[e].x)));
#endif
        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(m1) : "v"(__builtin_bit_cast(unsigned, x[(e + 3) % 8].y)));
        asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(p) : "v"(__builtin_bit_cast(unsigned, x[(e + 1) % 8].z)), "v"(__builtin_bit_cast(unsigned, x[(e + 2) % 8].w)), "v"(0x07060302u));
        vn[(n / 3) % 18][(n / 3 / 18) % 4] = p ^ m0 ^ m1;      // (two more independent xors: counted in the op total below as 5 per unit)
#endif
    };
    auto chunk = [&](int ch, u32x4(&vc)[18], u32x4(&vn)[18]) {
        int vdone = 0;
#pragma unroll
        for (int j = 0; j < 72; ++j) {
            const int p = j / 12, kb = (j / 6) & 1, prod = j % 6;
            const int sa = prod == 0 ? 0 : prod == 1 ? 0 : prod == 2 ? 1 : prod == 3 ? 0 : prod == 4 ? 2 : 1;
            const int sb = prod == 0 ? 0 : prod == 1 ? 1 : prod == 2 ? 0 : prod == 3 ? 2 : prod == 4 ? 0 : 1;
            acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, uP[p & 1][kb * 3 + sa]), __builtin_bit_cast(bf16x8, vc[p * 3 + sb]), acc[p * 2 + kb], 0, 0, 0);
            const int jj = j % 12;          // within the position: slots 0..5 load the next position's 6 filter pieces
            if (jj < 6 && p * 6 + jj < NLOAD)
                uP[(p + 1) & 1][jj] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, u_off, (ch & 15) * 147456 + (p * 6 + jj) * 1024, 0));
            else if (jj >= 6 && jj < 10 && p * 4 + jj - 6 < NREAD)
                x[(p * 4 + jj - 6) % 8] = *reinterpret_cast<const f32x4*>(lds + lane * 4 + wave * 1024 + ((p * 4 + jj - 6) % 16) * 256);
#pragma unroll
            for (int q = 0; q < (NVALU / 3 + 71) / 72; ++q)
                if (vdone < NVALU) { valu(vn, vdone); vdone += 3; }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    const long long t0 = __builtin_readcyclecounter();
    for (int ch = 0; ch < iters; ch += 2) {
        chunk(ch, vC, vN);
        chunk(ch + 1, vN, vC);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F>
static void run(const char* name, F kern, float* src, float* sink, long long* cyc) {
    const int iters = 2000, lds_bytes = 96 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds_bytes, 0, src, sink, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds_bytes, 0, src, sink, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(256);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("%-64s cycles / 16-channel chunk: %7.1f (72 MFMAs alone: 2304) | wall us %.4f\n", name, (double)c[128] / iters, ms * 1e3 / iters);
}
int main() {
    float *src, *sink; long long* cyc;
    hipMalloc(&src, 16 << 20); hipMemset(src, 0, 16 << 20); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run("72 bf16 MFMAs alone", k<0, 0, 0>, src, sink, cyc);
    run("+ 24 ds_read_b128", k<0, 24, 0>, src, sink, cyc);
    run("+ 24 reads + 36 filter loads", k<0, 24, 36>, src, sink, cyc);
    run("+ 144 VALU", k<144, 0, 0>, src, sink, cyc);
    run("+ 216 VALU", k<216, 0, 0>, src, sink, cyc);
    run("+ 288 VALU", k<288, 0, 0>, src, sink, cyc);
    run("+ 24 reads + 36 loads + 216 VALU", k<216, 24, 36>, src, sink, cyc);
    run("+ 24 reads + 36 loads + 288 VALU (the whole chunk)", k<288, 24, 36>, src, sink, cyc);
    return 0;
}
