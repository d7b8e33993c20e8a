// Device-side helpers shared by the gfx950 kernels of the probabilistic-inference path.
// Built with -ffp-contract=off: every a*b+c below is two IEEE roundings unless written fmaf(),
// because the CPU reference (eager torch) rounds after every op and index parity (top-k order,
// NMS keep masks) depends on reproducing its fp32 values.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pod_mi355x.h"

#define POD_WAVE 64

#define POD_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return POD_E_LAUNCH; \
    } while (0)

namespace pod {

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter-based: the same (counter, key) gives the same
// normals in K1 (dense scoring) and K2b (candidate re-scoring), so nothing has to be stored.
// ---------------------------------------------------------------------------------------------
struct u32x4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c.x;
        const uint64_t p1 = (uint64_t)M1 * c.z;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

// Box-Muller on two 32-bit words -> two standard normals (native-RNG mode; statistical parity only).
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;   // (0,1]
    const float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * __logf(u1));
    float s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    n0 = rad * c;
    n1 = rad * s;
}

struct f32x4n {
    float v[4];
};

// stream ids for the counter's 4th word
constexpr uint32_t STREAM_CLS = 0x636c7300u;   // classification logit samples (PI:291-295)
constexpr uint32_t STREAM_BOX = 0x626f7800u;   // box-delta samples (PI:351-356)

__device__ __forceinline__ f32x4n philox_normals(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t stream) {
    const u32x4 r = philox4x32_10(u32x4{c0, c1, c2, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    f32x4n o;
    box_muller(r.x, r.y, o.v[0], o.v[1]);
    box_muller(r.z, r.w, o.v[2], o.v[3]);
    return o;
}

// ---------------------------------------------------------------------------------------------
// arithmetic restated from the reference, one rounding per op
// ---------------------------------------------------------------------------------------------

// torch.sigmoid on CPU: 1 / (1 + exp(-x)) with a correctly rounded divide.
__device__ __forceinline__ float sigmoid_ref(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }

// PI:216-222 merge of N runs, in the reference's association order.
//   quirk: acc = x0; acc += x0; acc += x1; ... acc += x_{N-2}; acc /= N
//   true : acc = x0; acc += x1; ... acc += x_{N-1};           acc /= N
// `term_run(t)` is the run whose value is term t (t = 0..N-1).
__device__ __forceinline__ int merge_term_run(int t, int quirk) { return (quirk && t > 0) ? t - 1 : t; }

// Class probability of one (anchor, class): PI:289-297.
//   no variance head : sigmoid(logit)
//   variance head    : mean_s sigmoid(logit + eps_s * sqrt(exp(var))), s = 0..S-1, summed in order, / S
template <class EpsFn>
__device__ __forceinline__ float class_prob(float logit, float logvar, bool has_var, int S, EpsFn eps) {
    if (!has_var) return sigmoid_ref(logit);
    const float sigma = sqrtf(expf(logvar));
    float acc = 0.0f;
    for (int s = 0; s < S; ++s) {
        const float x = logit + eps(s) * sigma;
        acc = acc + sigmoid_ref(x);
    }
    return __fdiv_rn(acc, (float)S);
}

// Eps source for one (level anchor r, class k): replay tensor (S, R_l, K) in reference layout, or Philox.
struct ClsEps {
    const float* replay;     // may be null
    int64_t stride_s;        // R_l * K
    int64_t offset;          // r * K + k
    uint64_t seed;
    uint32_t gid, k;         // global anchor id, class
    float c0, c1, c2, c3;
    int cached_call;
    __device__ __forceinline__ ClsEps(const float* rp, int64_t rl, int K, int r, int kk, uint64_t sd, uint32_t g)
        : replay(rp), stride_s(rl * K), offset((int64_t)r * K + kk), seed(sd), gid(g), k((uint32_t)kk), cached_call(-1) {}
    __device__ __forceinline__ float operator()(int s) {
        if (replay) return replay[(int64_t)s * stride_s + offset];
        const int call = s >> 2;
        if (call != cached_call) {
            const f32x4n z = philox_normals(seed, gid, k, (uint32_t)call, STREAM_CLS);
            c0 = z.v[0]; c1 = z.v[1]; c2 = z.v[2]; c3 = z.v[3];
            cached_call = call;
        }
        const int q = s & 3;
        return q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
    }
};

// detectron2 Box2BoxTransform.apply_deltas / IU:510-547 on one box; dw, dh clamped at log(1000/16).
struct Box {
    float x1, y1, x2, y2;
};
#define POD_SCALE_CLAMP 4.135166556742356f

__device__ __forceinline__ Box decode_box(float d0, float d1, float d2, float d3, const Box& a, const float* wts) {
    const float w = a.x2 - a.x1;
    const float h = a.y2 - a.y1;
    const float cx = a.x1 + 0.5f * w;
    const float cy = a.y1 + 0.5f * h;
    const float dx = __fdiv_rn(d0, wts[0]);
    const float dy = __fdiv_rn(d1, wts[1]);
    float dw = __fdiv_rn(d2, wts[2]);
    float dh = __fdiv_rn(d3, wts[3]);
    dw = fminf(dw, POD_SCALE_CLAMP);
    dh = fminf(dh, POD_SCALE_CLAMP);
    const float pcx = dx * w + cx;
    const float pcy = dy * h + cy;
    const float pw = expf(dw) * w;
    const float ph = expf(dh) * h;
    Box o;
    o.x1 = pcx - 0.5f * pw;
    o.y1 = pcy - 0.5f * ph;
    o.x2 = pcx + 0.5f * pw;
    o.y2 = pcy + 0.5f * ph;
    return o;
}

// detectron2 pairwise_iou on one pair (inter > 0 guard, areas without +1).
__device__ __forceinline__ float iou_pair(const Box& a, const Box& b) {
    float w = fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1);
    float h = fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1);
    w = fmaxf(w, 0.0f);
    h = fmaxf(h, 0.0f);
    const float inter = w * h;
    const float aa = (a.x2 - a.x1) * (a.y2 - a.y1);
    const float ab = (b.x2 - b.x1) * (b.y2 - b.y1);
    return inter > 0.0f ? __fdiv_rn(inter, (aa + ab) - inter) : 0.0f;
}

__device__ __forceinline__ Box load_box(const float* p, int i) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)i * 4);
    return Box{v.x, v.y, v.z, v.w};
}

// wavefront (64-lane) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 64-bit candidate key: descending key order = descending score, then ascending anchor index.
__device__ __forceinline__ uint64_t make_key(float score, int r) {
    return ((uint64_t)__float_as_uint(score) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)r);
}
__device__ __forceinline__ float key_score(uint64_t k) { return __uint_as_float((uint32_t)(k >> 32)); }
__device__ __forceinline__ int key_index(uint64_t k) { return (int)(0xFFFFFFFFu - (uint32_t)k); }

}  // namespace pod
