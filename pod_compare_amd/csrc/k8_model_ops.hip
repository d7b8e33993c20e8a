// Fused element-wise ops for the conv-net side of the path (the convolutions themselves stay in MIOpen).
//
// pod_relu_dropout: y = dropout(relu(x), p) in place -- the `nn.ReLU(), nn.Dropout(p)` pair that follows every
// 3x3 conv of the probabilistic RetinaNet head's subnets (probabilistic_retinanet.py:403-424), evaluated N times
// per image in MC-dropout mode (PR:103-108).  torch runs it as two kernels (clamp: read+write, fused_dropout:
// read+write+mask); this is one pass, 16 B per lane, 16 Philox bits per element (pod_device.h: dropout_words; keep iff
// uniform >= p, scaled by 1/(1-p), torch.nn.functional.dropout's definition).  HBM-bound: 8 bytes per element.
#include "pod_wino.h"

namespace pod {

constexpr uint32_t STREAM_DROPOUT = 0x64726f70u;

__global__ void __launch_bounds__(256) k_relu_dropout(float* __restrict__ x, int64_t n4, int64_t n, uint32_t thresh,
                                                      float scale, uint64_t seed, uint64_t offset) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        uint32_t w0, w1;
        dropout_words(offset, (uint64_t)i, 0u, STREAM_DROPOUT, seed, w0, w1);
        float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        v.x = ((w0 & 0xFFFFu) >= thresh) ? fmaxf(v.x, 0.0f) * scale : 0.0f;
        v.y = ((w0 >> 16) >= thresh) ? fmaxf(v.y, 0.0f) * scale : 0.0f;
        v.z = ((w1 & 0xFFFFu) >= thresh) ? fmaxf(v.z, 0.0f) * scale : 0.0f;
        v.w = ((w1 >> 16) >= thresh) ? fmaxf(v.w, 0.0f) * scale : 0.0f;
        *reinterpret_cast<float4*>(x + i * 4) = v;
    }
    // tail (n % 4 elements)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t e = n4 * 4 + threadIdx.x;
        const uint64_t ctr = offset + (uint64_t)n4 + threadIdx.x;
        const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, STREAM_DROPOUT}, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        x[e] = ((r.x & 0xFFFFu) >= thresh) ? fmaxf(x[e], 0.0f) * scale : 0.0f;
    }
}

// pod_bias_act: x[n, c, :] = dropout(relu((x + bias[c]) + (residual + res_bias[c])), p) in place, every stage optional.
// torch's conv on ROCm is MIOpen's kernel followed by a separate bias `add_`; with ReLU (+ dropout, + the residual
// add of a bottleneck) that is 2-4 element-wise passes over the activation.  The convolution is called without its
// bias and this ONE pass does the rest.  Same Philox counters as k_relu_dropout (counter = offset + float4 index).
struct BiasActParams {
    float* x;
    const float* bias;       // [C] or NULL
    const float* residual;   // same shape as x or NULL
    const float* res_bias;   // [C] or NULL (bias of the shortcut conv that produced `residual`)
    int64_t n4, n, HW;
    int32_t C, relu;
    uint32_t thresh;         // 0 = no dropout
    float scale;
    uint64_t seed, offset;
};

__device__ __forceinline__ float bias_act_one(float v, float b, float r, int relu, bool keep, float scale) {
    v = (v + b) + r;
    if (relu) v = fmaxf(v, 0.0f);
    return keep ? v * scale : 0.0f;
}

// LAYOUT 0: NCHW planes, H*W % 4 == 0 (4 elements share a channel); 1: NHWC, C % 4 == 0 (4 consecutive channels:
// one 16-byte bias load); 2: anything else (per-element channel).
template <int LAYOUT>
__global__ void __launch_bounds__(256) k_bias_act(const BiasActParams P) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P.n4; i += stride) {
        uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu;
        if (P.thresh) dropout_words(P.offset, (uint64_t)i, 0u, STREAM_DROPOUT, P.seed, w0, w1);
        float4 v = *reinterpret_cast<const float4*>(P.x + i * 4);
        float4 res = float4{0.f, 0.f, 0.f, 0.f};
        if (P.residual) res = *reinterpret_cast<const float4*>(P.residual + i * 4);
        float b[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
        if (LAYOUT == 0) {
            const int c = (int)(((i * 4) / P.HW) % P.C);
            if (P.bias) b[0] = b[1] = b[2] = b[3] = P.bias[c];
            if (P.res_bias) rb[0] = rb[1] = rb[2] = rb[3] = P.res_bias[c];
        } else if (LAYOUT == 1) {
            const int c = (int)((i * 4) % P.C);
            if (P.bias) {
                const float4 t = *reinterpret_cast<const float4*>(P.bias + c);
                b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
            }
            if (P.res_bias) {
                const float4 t = *reinterpret_cast<const float4*>(P.res_bias + c);
                rb[0] = t.x; rb[1] = t.y; rb[2] = t.z; rb[3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (int)(((i * 4 + j) / P.HW) % P.C);
                if (P.bias) b[j] = P.bias[c];
                if (P.res_bias) rb[j] = P.res_bias[c];
            }
        }
        v.x = bias_act_one(v.x, b[0], res.x + rb[0], P.relu, (w0 & 0xFFFFu) >= P.thresh, P.scale);
        v.y = bias_act_one(v.y, b[1], res.y + rb[1], P.relu, (w0 >> 16) >= P.thresh, P.scale);
        v.z = bias_act_one(v.z, b[2], res.z + rb[2], P.relu, (w1 & 0xFFFFu) >= P.thresh, P.scale);
        v.w = bias_act_one(v.w, b[3], res.w + rb[3], P.relu, (w1 >> 16) >= P.thresh, P.scale);
        *reinterpret_cast<float4*>(P.x + i * 4) = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (P.n & 3)) {   // tail (n % 4 elements)
        const int64_t e = P.n4 * 4 + threadIdx.x;
        bool keep = true;
        if (P.thresh) {
            const uint64_t ctr = P.offset + (uint64_t)P.n4 + threadIdx.x;
            const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 1u, STREAM_DROPOUT}, (uint32_t)P.seed,
                                          (uint32_t)(P.seed >> 32));
            keep = (r.x & 0xFFFFu) >= P.thresh;
        }
        const int c = (int)((e / P.HW) % P.C);
        const float res = (P.residual ? P.residual[e] : 0.0f) + (P.res_bias ? P.res_bias[c] : 0.0f);
        P.x[e] = bias_act_one(P.x[e], P.bias ? P.bias[c] : 0.0f, res, P.relu, keep, P.scale);
    }
}

// pod_expand_dropout: dst[c][i] = dropout(src[i], p) for c < copies, an independent mask per copy.  The first conv of a
// head subnet sees the same input in every MC run, so it is evaluated once; this writes the `copies` dropout-perturbed
// inputs of the second conv in one pass (torch: expand + fused_dropout, which also writes a mask tensor).  Flat arrays:
// any memory format, as long as src and every dst copy use the same one.
__global__ void __launch_bounds__(256) k_expand_dropout(const float* __restrict__ src, float* __restrict__ dst, int64_t n4, int32_t copies,
                                                        uint32_t thresh, float scale, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ epoch) {
    seed = dropout_key(seed, epoch);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
        for (int c = 0; c < copies; ++c) {
            uint32_t w0, w1;
            dropout_words(offset, (uint64_t)c * (uint64_t)n4 + (uint64_t)i, 2u, STREAM_DROPOUT, seed, w0, w1);
            float4 o;
            o.x = ((w0 & 0xFFFFu) >= thresh) ? v.x * scale : 0.0f;
            o.y = ((w0 >> 16) >= thresh) ? v.y * scale : 0.0f;
            o.z = ((w1 & 0xFFFFu) >= thresh) ? v.z * scale : 0.0f;
            o.w = ((w1 >> 16) >= thresh) ? v.w * scale : 0.0f;
            *reinterpret_cast<float4*>(dst + ((int64_t)c * n4 + i) * 4) = o;
        }
    }
}

// pod_bias_act_to_nchw: the same tail as pod_bias_act for a channels-last conv output, written as NCHW planes -- the
// layout change rides on the element-wise pass that exists anyway (a separate transposing copy of the 300 MB p3 trunk
// output costs 0.3 ms in torch).  Workgroup = one 64 (cells) x 64 (channels) tile through LDS: 16-byte loads along C,
// 16-byte stores along H*W.  Dropout fields are those of pod_bias_act on the NCHW result (float4 group = NCHW
// float4 index), so the output equals "transpose, then pod_bias_act" bit for bit.
__global__ void __launch_bounds__(256) k_bias_act_to_nchw(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ bias,
                                                          int32_t C, int64_t HW, int32_t relu, uint32_t thresh, float scale, uint64_t seed,
                                                          uint64_t offset, int32_t tiles_hw, int32_t tiles_c) {
    __shared__ float tile[64][65];   // [cell][channel], padded
    const int tid = threadIdx.x;
    int64_t t = blockIdx.x;
    const int tc = (int)(t % tiles_c);
    t /= tiles_c;
    const int th = (int)(t % tiles_hw);
    const int64_t n = t / tiles_hw;
    const int64_t hw0 = (int64_t)th * 64;
    const int c0 = tc * 64;
    // load: thread -> (cell row = tid / 16 + 16 * it, 4 channels at (tid % 16) * 4)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4;
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (hw0 + r < HW && c0 + c4 < C) {
            v = *reinterpret_cast<const float4*>(src + ((n * HW + hw0 + r) * C + c0 + c4));
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + c0 + c4);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
        }
        tile[r][c4 + 0] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
    __syncthreads();
    // store: thread -> (channel row = tid / 16 + 16 * it, 4 cells at (tid % 16) * 4)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (tid >> 4) + 16 * it, h4 = (tid & 15) * 4;
        if (c0 + c >= C || hw0 + h4 >= HW) continue;
        const int64_t e = (n * C + c0 + c) * HW + hw0 + h4;   // NCHW element index, a multiple of 4
        uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu;
        if (thresh) dropout_words(offset, (uint64_t)(e >> 2), 0u, STREAM_DROPOUT, seed, w0, w1);
        float4 v = float4{tile[h4 + 0][c], tile[h4 + 1][c], tile[h4 + 2][c], tile[h4 + 3][c]};
        v.x = bias_act_one(v.x, 0.0f, 0.0f, relu, (w0 & 0xFFFFu) >= thresh, scale);
        v.y = bias_act_one(v.y, 0.0f, 0.0f, relu, (w0 >> 16) >= thresh, scale);
        v.z = bias_act_one(v.z, 0.0f, 0.0f, relu, (w1 & 0xFFFFu) >= thresh, scale);
        v.w = bias_act_one(v.w, 0.0f, 0.0f, relu, (w1 >> 16) >= thresh, scale);
        *reinterpret_cast<float4*>(dst + e) = v;
    }
}

// pod_wino_reduce: finishes a convolution that pod_wino_conv3x3_split_partial cut over its input channels -- the n_splits channels-last
// partial sums (pixels, Kpad) are added in a FIXED order (split 0 first: the result does not depend on scheduling), bias and ReLU
// applied, and the K real channels written as NCHW planes of one image (HW = pixels): what the consumer of a backbone convolution
// reads.  64 (pixels) x 64 (channels) tiles through LDS: 16-byte loads along the channels, 16-byte stores along H*W.
__global__ void __launch_bounds__(256) k_wino_reduce(const float* __restrict__ partials, int32_t n_splits, int64_t split_stride, const float* __restrict__ bias,
                                                     float* __restrict__ planes, int64_t HW, int32_t Kpad, int32_t K, int32_t relu, int32_t tiles_c,
                                                     float* __restrict__ out_amax) {
    float lmax = 0.0f;
    __shared__ float tile[64][65];   // [pixel][channel], padded
    const int tid = threadIdx.x;
    const int tc = (int)(blockIdx.x % tiles_c);
    const int64_t hw0 = (int64_t)(blockIdx.x / tiles_c) * 64;
    const int c0 = tc * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4;
        float4 v = float4{0.f, 0.f, 0.f, 0.f};
        if (hw0 + r < HW && c0 + c4 < Kpad) {
            const float* src = partials + (hw0 + r) * Kpad + c0 + c4;
            v = *reinterpret_cast<const float4*>(src);
            for (int s = 1; s < n_splits; ++s) {
                const float4 w = *reinterpret_cast<const float4*>(src + (int64_t)s * split_stride);
                v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
            }
            if (bias) {                                   // (Kpad values: the caller's bias is padded with zeros)
                const float4 b = *reinterpret_cast<const float4*>(bias + c0 + c4);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
        }
        tile[r][c4 + 0] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
        lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));     // (padded channels: 0 + 0)
    }
    if (out_amax) wino_publish_amax_block(out_amax, lmax);
    __syncthreads();
    const bool vec = (HW & 3) == 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (tid >> 4) + 16 * it, h4 = (tid & 15) * 4;
        if (c0 + c >= K || hw0 + h4 >= HW) continue;
        float* dst = planes + (int64_t)(c0 + c) * HW + hw0 + h4;
        if (vec) {
            *reinterpret_cast<float4*>(dst) = float4{tile[h4 + 0][c], tile[h4 + 1][c], tile[h4 + 2][c], tile[h4 + 3][c]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (hw0 + h4 + j < HW) dst[j] = tile[h4 + j][c];
        }
    }
}

// pod_bias_act_to_nhwc: the reverse trip, for a conv whose CONSUMER is pod_wino_conv3x3 (channels-last input): the bias + ReLU pass
// that follows an NCHW (MIOpen) conv anyway writes [pixel][C] instead of planes.  64 (cells) x 64 (channels) tiles through LDS:
// 16-byte loads along H*W (scalar when H*W % 4 != 0), 16-byte stores along C.
__global__ void __launch_bounds__(256) k_bias_act_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ bias,
                                                          int32_t C, int64_t HW, int32_t relu, int32_t tiles_hw, int32_t tiles_c) {
    __shared__ float tile[64][65];   // [channel][cell], padded
    const int tid = threadIdx.x;
    int64_t t = blockIdx.x;
    const int tc = (int)(t % tiles_c);
    t /= tiles_c;
    const int th = (int)(t % tiles_hw);
    const int64_t n = t / tiles_hw;
    const int64_t hw0 = (int64_t)th * 64;
    const int c0 = tc * 64;
    const bool vec = (HW & 3) == 0;
    // load: thread -> (channel row = tid / 16 + 16 * it, 4 cells at (tid % 16) * 4)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = (tid >> 4) + 16 * it, h4 = (tid & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c0 + c < C) {
            const float* p = src + (n * C + c0 + c) * HW + hw0 + h4;
            if (vec && hw0 + h4 < HW) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (hw0 + h4 + j < HW) v[j] = p[j];
            }
            const float b = bias ? bias[c0 + c] : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] += b;
                if (relu) v[j] = fmaxf(v[j], 0.0f);
            }
        }
        tile[c][h4 + 0] = v[0]; tile[c][h4 + 1] = v[1]; tile[c][h4 + 2] = v[2]; tile[c][h4 + 3] = v[3];
    }
    __syncthreads();
    // store: thread -> (cell row = tid / 16 + 16 * it, 4 channels at (tid % 16) * 4)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (tid >> 4) + 16 * it, c4 = (tid & 15) * 4;
        if (hw0 + r >= HW || c0 + c4 >= C) continue;
        *reinterpret_cast<float4*>(dst + ((n * HW + hw0 + r) * C + c0 + c4)) = float4{tile[c4 + 0][r], tile[c4 + 1][r], tile[c4 + 2][r], tile[c4 + 3][r]};
    }
}

}  // namespace pod

extern "C" int pod_bias_act_to_nhwc(const float* src, float* dst, const float* bias, int64_t N, int32_t C, int64_t HW, int32_t relu,
                                    pod_stream_t stream) {
    if (!src || !dst || src == dst || N < 0 || C < 4 || (C & 3) != 0 || HW < 1) return POD_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15u) != 0) return POD_E_INVALID;
    if (N == 0) return POD_OK;
    const int64_t tiles_hw = (HW + 63) / 64, tiles_c = (C + 63) / 64;
    const int64_t blocks = N * tiles_hw * tiles_c;
    if (blocks > 0x7FFFFFFFLL || tiles_hw > 0x7FFFFFFFLL) return POD_E_INVALID;
    hipLaunchKernelGGL(pod::k_bias_act_to_nhwc, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, bias, C, HW, relu,
                       (int32_t)tiles_hw, (int32_t)tiles_c);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_bias_act_to_nchw(const float* src, float* dst, const float* bias, int64_t N, int32_t C, int64_t HW, int32_t relu,
                                    float p, uint64_t seed, uint64_t offset, pod_stream_t stream) {
    if (!src || !dst || src == dst || N < 0 || C < 4 || (C & 3) != 0 || HW < 4 || (HW & 3) != 0 || !(p >= 0.0f && p < 1.0f)) return POD_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15u) != 0 ||
        (reinterpret_cast<uintptr_t>(bias) & 15u) != 0)
        return POD_E_INVALID;
    if (N == 0) return POD_OK;
    const int64_t tiles_hw = (HW + 63) / 64, tiles_c = (C + 63) / 64;
    const int64_t blocks = N * tiles_hw * tiles_c;
    if (blocks > 0x7FFFFFFFLL || tiles_hw > 0x7FFFFFFFLL) return POD_E_INVALID;
    const uint32_t thresh = POD_DROPOUT_THRESH16(p);
    hipLaunchKernelGGL(pod::k_bias_act_to_nchw, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, bias, C, HW, relu, thresh,
                       1.0f / (1.0f - p), seed, offset, (int32_t)tiles_hw, (int32_t)tiles_c);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

// the abs-max of a tensor, max'ed into *amax (pod_mi355x.h: operand abs-max words)
namespace pod {
__global__ void __launch_bounds__(256) k_absmax(const float* __restrict__ x, int64_t n, float* __restrict__ amax) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    float m = 0.0f;
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i + 3 < n; i += stride) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (; i < n; ++i) m = fmaxf(m, fabsf(x[i]));        // (the tail of an n that is not a multiple of 4: one thread)
    wino_publish_amax_block(amax, m);
}
}  // namespace pod

extern "C" int pod_absmax(const float* x, int64_t n, float* amax, pod_stream_t stream) {
    if (!x || !amax || n < 0 || (reinterpret_cast<uintptr_t>(x) & 15u) != 0 || (reinterpret_cast<uintptr_t>(amax) & 3u) != 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
    hipLaunchKernelGGL(pod::k_absmax, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, amax);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_wino_reduce(const float* partials, int32_t n_splits, int64_t split_stride, const float* bias, float* planes, int64_t HW,
                               int32_t Kpad, int32_t K, int32_t relu, float* out_amax, pod_stream_t stream) {
    if (!partials || !planes || n_splits < 1 || n_splits > 16 || HW < 1 || Kpad < 4 || (Kpad & 3) != 0 || K < 1 || K > Kpad) return POD_E_INVALID;
    if (n_splits > 1 && (split_stride < HW * Kpad || (split_stride & 3) != 0)) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(partials) | reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(bias)) & 15u) != 0) return POD_E_INVALID;
    const int64_t tiles_hw = (HW + 63) / 64, tiles_c = (K + 63) / 64;
    hipLaunchKernelGGL(pod::k_wino_reduce, dim3((unsigned)(tiles_hw * tiles_c)), dim3(256), 0, (hipStream_t)stream, partials, n_splits, split_stride, bias,
                       planes, HW, Kpad, K, relu, (int32_t)tiles_c, out_amax);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_expand_dropout(const float* src, float* dst, int64_t n, int32_t copies, float p, uint64_t seed, uint64_t offset,
                                  const uint64_t* epoch, pod_stream_t stream) {
    if (!src || !dst || n < 0 || (n & 3) != 0 || copies < 1 || !(p >= 0.0f && p < 1.0f)) return POD_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) != 0 || (reinterpret_cast<uintptr_t>(dst) & 15u) != 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    const int64_t n4 = n / 4;
    const uint32_t thresh = POD_DROPOUT_THRESH16(p);
    const float scale = 1.0f / (1.0f - p);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pod::k_expand_dropout, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n4, copies, thresh, scale,
                       seed, offset, epoch);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_bias_act(float* x, const float* bias, const float* residual, const float* res_bias, int64_t n, int32_t C,
                            int64_t HW, int32_t relu, float p, uint64_t seed, uint64_t offset, pod_stream_t stream) {
    if (!x || n < 0 || C < 1 || HW < 1 || !(p >= 0.0f && p < 1.0f)) return POD_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0 || (reinterpret_cast<uintptr_t>(residual) & 15u) != 0) return POD_E_INVALID;
    if (res_bias && !residual) return POD_E_INVALID;
    if (n % ((int64_t)C * HW) != 0) return POD_E_INVALID;   // x is (N, C, H, W)
    if (n == 0) return POD_OK;
    pod::BiasActParams P;
    P.x = x; P.bias = bias; P.residual = residual; P.res_bias = res_bias;
    P.n4 = n / 4; P.n = n; P.HW = HW; P.C = C; P.relu = relu;
    P.thresh = POD_DROPOUT_THRESH16(p);   // keep iff 16-bit field >= p * 2^16
    P.scale = 1.0f / (1.0f - p);
    P.seed = seed; P.offset = offset;
    int64_t blocks = (P.n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride: 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    const bool bias16 = (reinterpret_cast<uintptr_t>(bias) & 15u) == 0 && (reinterpret_cast<uintptr_t>(res_bias) & 15u) == 0;
    if (HW % 4 == 0) hipLaunchKernelGGL(pod::k_bias_act<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    else if (HW == 1 && C % 4 == 0 && bias16) hipLaunchKernelGGL(pod::k_bias_act<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(pod::k_bias_act<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_relu_dropout(float* x, int64_t n, float p, uint64_t seed, uint64_t offset, pod_stream_t stream) {
    if (!x || n < 0 || !(p >= 0.0f && p < 1.0f) || (reinterpret_cast<uintptr_t>(x) & 15u) != 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    const int64_t n4 = n / 4;
    const uint32_t thresh = POD_DROPOUT_THRESH16(p);   // keep iff 16-bit field >= p * 2^16
    const float scale = 1.0f / (1.0f - p);
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // grid-stride: 16 workgroups per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(pod::k_relu_dropout, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, thresh, scale, seed, offset);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
