// GPU box experiment: is an fp32 product emulated by 3-way bf16 splits on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
// accumulate) as accurate as the fp32 MFMA?  One 32 x 32 output block, K = 2304 (a 3x3 conv over 256 channels), random operands.
//   products kept: 6 (a0b0, a0b1, a1b0, a0b2, a2b0, a1b1) or 9 (all); reference: fp64 on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// A: [32][K] row-major fp32, B: [K][32] (as Bt[32][K] row-major), out[mode][32][32].  mode 0: fp32 MFMA, 1: bf16 x6, 2: bf16 x9, 3: bf16 x3 (2-way split)
__global__ void __launch_bounds__(64) k(const float* A, const float* Bt, float* out, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 c32 = {0}, c6 = {0}, c9 = {0}, c3 = {0};
    for (int k0 = 0; k0 < K; k0 += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], Bt[r * K + k0 + h], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            float x = A[r * K + k0 + 8 * h + j], y = Bt[r * K + k0 + 8 * h + j];
            for (int s = 0; s < 3; ++s) {
                unsigned short xa = bf16_rn(x), yb = bf16_rn(y);
                a[s][j] = __builtin_bit_cast(__bf16, xa);
                b[s][j] = __builtin_bit_cast(__bf16, yb);
                x -= bf16_f(xa);
                y -= bf16_f(yb);
            }
        }
        // small terms first, the big one last
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], c9, 0, 0, 0);
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], c9, 0, 0, 0);
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], c9, 0, 0, 0);
        for (f32x16* c : {&c6, &c9}) {
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], *c, 0, 0, 0);
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], *c, 0, 0, 0);
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], *c, 0, 0, 0);
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], *c, 0, 0, 0);
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], *c, 0, 0, 0);
            *c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], *c, 0, 0, 0);
        }
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c3, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c3, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;      // D[row][col = r]
        out[0 * 1024 + row * 32 + r] = c32[i];
        out[1 * 1024 + row * 32 + r] = c6[i];
        out[2 * 1024 + row * 32 + r] = c9[i];
        out[3 * 1024 + row * 32 + r] = c3[i];
    }
}
int main() {
    const int K = 2304;
    std::vector<float> A(32 * K), Bt(32 * K);
    srand(1);
    auto rnd = []() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };   // ~N(0,1)
    for (auto& v : A) v = fmaxf(rnd(), 0.0f);            // post-ReLU activations
    for (auto& v : Bt) v = rnd() * 0.03f;                // filters
    float *dA, *dB, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dO, 4 * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dO, K);
    std::vector<float> O(4 * 1024);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    const char* names[4] = {"fp32 MFMA (32x32x2)", "bf16 x6 (3-way split)", "bf16 x9 (3-way split)", "bf16 x3 (2-way split)"};
    for (int m = 0; m < 4; ++m) {
        double worst = 0, rms = 0, worst_rel = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, bound = 0;
                for (int kk = 0; kk < K; ++kk) { ref += (double)A[i * K + kk] * Bt[j * K + kk]; bound += fabs((double)A[i * K + kk] * Bt[j * K + kk]); }
                const double e = fabs(O[m * 1024 + i * 32 + j] - ref) / (ldexp(1.0, -24) * bound);
                worst = fmax(worst, e); rms += e * e;
            }
        printf("%-24s c = max |err| / (2^-24 sum|a b|) = %8.3f   rms %8.3f\n", names[m], worst, sqrt(rms / 1024));
    }
    return 0;
}
