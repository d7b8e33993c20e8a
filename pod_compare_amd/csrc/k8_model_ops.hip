// Fused element-wise ops for the conv-net side of the path (the convolutions themselves stay in MIOpen).
//
// pod_relu_dropout: y = dropout(relu(x), p) in place -- the `nn.ReLU(), nn.Dropout(p)` pair that follows every
// 3x3 conv of the probabilistic RetinaNet head's subnets (probabilistic_retinanet.py:403-424), evaluated N times
// per image in MC-dropout mode (PR:103-108).  torch runs it as two kernels (clamp: read+write, fused_dropout:
// read+write+mask); this is one pass, 16 B per lane, one Philox4x32-10 call per 4 elements (keep iff
// uniform >= p, scaled by 1/(1-p), torch.nn.functional.dropout's definition).  HBM-bound: 8 bytes per element.
#include "pod_device.h"

namespace pod {

constexpr uint32_t STREAM_DROPOUT = 0x64726f70u;

__global__ void __launch_bounds__(256) k_relu_dropout(float* __restrict__ x, int64_t n4, int64_t n, uint32_t thresh,
                                                      float scale, uint64_t seed, uint64_t offset) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint64_t ctr = offset + (uint64_t)i;
        const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, STREAM_DROPOUT}, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        v.x = (r.x >= thresh) ? fmaxf(v.x, 0.0f) * scale : 0.0f;
        v.y = (r.y >= thresh) ? fmaxf(v.y, 0.0f) * scale : 0.0f;
        v.z = (r.z >= thresh) ? fmaxf(v.z, 0.0f) * scale : 0.0f;
        v.w = (r.w >= thresh) ? fmaxf(v.w, 0.0f) * scale : 0.0f;
        *reinterpret_cast<float4*>(x + i * 4) = v;
    }
    // tail (n % 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t e = n4 * 4 + threadIdx.x;
        const uint64_t ctr = offset + (uint64_t)n4 + threadIdx.x;
        const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, STREAM_DROPOUT}, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        x[e] = (r.x >= thresh) ? fmaxf(x[e], 0.0f) * scale : 0.0f;
    }
}

}  // namespace pod

extern "C" int pod_relu_dropout(float* x, int64_t n, float p, uint64_t seed, uint64_t offset, pod_stream_t stream) {
    if (!x || n < 0 || !(p >= 0.0f && p < 1.0f) || (reinterpret_cast<uintptr_t>(x) & 15u) != 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    const int64_t n4 = n / 4;
    const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);   // keep iff u32 >= p * 2^32
    const float scale = 1.0f / (1.0f - p);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride: 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pod::k_relu_dropout, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, thresh, scale, seed, offset);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
