// GPU box experiment (round 5): an fp32 product from 2-way FP16 splits on the f16 matrix cores (v_mfma_f32_32x32x16_f16, fp32
// accumulate) -- 3 partial products (a0b0, a0b1, a1b0) where the 3-way bf16 split needs 6.  a = a0 + a1 to <= 2^-23 |a| (11 + 1 + 11
// bits of 24), so the operand REPRESENTATION carries about one extra fp32 rounding; fp16's range needs a power-of-two operand scale.
// 64 random 32 x 32 output blocks per K; reference: fp64 on the host.     hipcc --offload-arch=gfx950 -O2 -o f16n f16_split_numerics.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// out[mode][block][32][32]; mode 0 fp32 MFMA, 1 bf16 x6, 2 f16 x3 (scaled), 3 f16 x4 (scaled, + a1 b1), 4 f16 x3 small-first order
__global__ void __launch_bounds__(64) k(const float* A0, const float* Bt0, float* out, int K, float sa, float sb, int nblk) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5, blk = blockIdx.x;
    const float* A = A0 + (size_t)blk * 32 * K;
    const float* Bt = Bt0 + (size_t)blk * 32 * K;
    f32x16 c32 = {0}, c6 = {0}, c3 = {0}, c4 = {0}, c3b = {0};
    for (int k0 = 0; k0 < K; k0 += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], Bt[r * K + k0 + h], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        f16x8 fa[2], fb[2];
        for (int j = 0; j < 8; ++j) {
            float x = A[r * K + k0 + 8 * h + j], y = Bt[r * K + k0 + 8 * h + j];
            float xs = x * sa, ys = y * sb;
            for (int s = 0; s < 3; ++s) {
                unsigned short xa = bf16_rn(x), yb = bf16_rn(y);
                a[s][j] = __builtin_bit_cast(__bf16, xa);
                b[s][j] = __builtin_bit_cast(__bf16, yb);
                x -= bf16_f(xa);
                y -= bf16_f(yb);
            }
            for (int s = 0; s < 2; ++s) {
                _Float16 xa = (_Float16)xs, yb = (_Float16)ys;      // v_cvt_f16_f32: round to nearest even
                fa[s][j] = xa; fb[s][j] = yb;
                xs -= (float)xa; ys -= (float)yb;
            }
        }
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c6, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c6, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[1], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[0], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[0], c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[1], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[1], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[0], c4, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[0], c4, 0, 0, 0);
    }
    // mode 4: the small partial products in their own accumulator over the whole K, added to the main one once at the end
    f32x16 lo = {0}, hi = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 fa[2], fb[2];
        for (int j = 0; j < 8; ++j) {
            float xs = A[r * K + k0 + 8 * h + j] * sa, ys = Bt[r * K + k0 + 8 * h + j] * sb;
            for (int s = 0; s < 2; ++s) {
                _Float16 xa = (_Float16)xs, yb = (_Float16)ys;
                fa[s][j] = xa; fb[s][j] = yb;
                xs -= (float)xa; ys -= (float)yb;
            }
        }
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[1], lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1], fb[0], lo, 0, 0, 0);
        hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0], fb[0], hi, 0, 0, 0);
    }
    c3b = hi + lo;
    const float inv = 1.0f / (sa * sb);
    float* o = out + (size_t)blk * 1024;
    const size_t ms = (size_t)nblk * 1024;
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        o[0 * ms + row * 32 + r] = c32[i];
        o[1 * ms + row * 32 + r] = c6[i];
        o[2 * ms + row * 32 + r] = c3[i] * inv;
        o[3 * ms + row * 32 + r] = c4[i] * inv;
        o[4 * ms + row * 32 + r] = c3b[i] * inv;
    }
}
int main() {
    const int NB = 64;
    for (int K : {256, 2304}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A((size_t)NB * 32 * K), Bt((size_t)NB * 32 * K);
            srand(1 + dist);
            auto rnd = []() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };
            float amax = 0, bmax = 0;
            for (auto& v : A) {
                v = dist == 0 ? fmaxf(rnd(), 0.0f) : dist == 1 ? rnd() * 7.0f : fmaxf(rnd(), 0.0f) * expf(3.0f * rnd());   // post-ReLU; transformed (signed); heavy-tailed
                amax = fmaxf(amax, fabsf(v));
            }
            for (auto& v : Bt) { v = rnd() * 0.03f; bmax = fmaxf(bmax, fabsf(v)); }
            const float sa = ldexpf(1.0f, 15 - (int)ceilf(log2f(amax))), sb = ldexpf(1.0f, 14 - (int)floorf(log2f(bmax)));
            float *dA, *dB, *dO;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dO, (size_t)5 * NB * 1024 * 4);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, dA, dB, dO, K, sa, sb, NB);
            std::vector<float> O((size_t)5 * NB * 1024);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            const char* names[5] = {"fp32 MFMA (32x32x2)", "bf16 x6 (3-way split)", "f16 x3 (2-way split)", "f16 x4 (2-way, all)", "f16 x3, small terms apart"};
            printf("K = %d, operand distribution %d (amax %.3g, scales 2^%d 2^%d)\n", K, dist, amax, (int)log2f(sa), (int)log2f(sb));
            std::vector<double> ref((size_t)NB * 1024), bound((size_t)NB * 1024);
            for (int b = 0; b < NB; ++b)
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double rr = 0, bb = 0;
                        const float* a = &A[((size_t)b * 32 + i) * K];
                        const float* w = &Bt[((size_t)b * 32 + j) * K];
                        for (int kk = 0; kk < K; ++kk) { rr += (double)a[kk] * w[kk]; bb += fabs((double)a[kk] * w[kk]); }
                        ref[(size_t)b * 1024 + i * 32 + j] = rr; bound[(size_t)b * 1024 + i * 32 + j] = bb;
                    }
            for (int m = 0; m < 5; ++m) {
                double worst = 0, rms = 0;
                for (size_t t = 0; t < (size_t)NB * 1024; ++t) {
                    const double e = fabs(O[m * (size_t)NB * 1024 + t] - ref[t]) / (ldexp(1.0, -24) * bound[t]);
                    worst = fmax(worst, e); rms += e * e;
                }
                printf("  %-28s c = max |err| / (2^-24 sum|a b|) = %8.3f   rms %8.3f\n", names[m], worst, sqrt(rms / (NB * 1024.0)));
            }
            hipFree(dA); hipFree(dB); hipFree(dO);
        }
    }
    return 0;
}
