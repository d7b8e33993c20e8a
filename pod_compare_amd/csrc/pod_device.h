// Device-side helpers shared by the gfx950 kernels of the probabilistic-inference path.
// Built with -ffp-contract=off: every a*b+c below is two IEEE roundings unless written fmaf(),
// because the CPU reference (eager torch) rounds after every op and index parity (top-k order,
// NMS keep masks) depends on reproducing its fp32 values.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pod_mi355x.h"

#define POD_WAVE 64
#define POD_EPS_MAX 4.9f   // > sqrt(-2 ln 2^-17) = 4.855: largest |normal| box_muller16() can return

#define POD_CHECK_LAUNCH()                                  \
    do {                                                    \
        if (hipGetLastError() != hipSuccess) return POD_E_LAUNCH; \
    } while (0)

// -DPOD_TRACE (python -m pod_compare_amd.build with POD_TRACE=1; diagnostics only, never the shipped library): phase
// time stamps of the first workgroups of a kernel, constant 100 MHz clock, dumped by pod_trace_dump() of the same file.
#ifdef POD_TRACE
static __device__ long long g_pod_trace[4096 * 8];
#define POD_STAMP(wg, k)                                                                         \
    do {                                                                                         \
        if ((threadIdx.x & 63) == 0 && threadIdx.x < 64 && (wg) < 4096) g_pod_trace[(wg) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define POD_STAMP(wg, k)
#endif

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

// Dropout masks of the model-side kernels (k8_model_ops.hip, k11_wino_conv.hip): 16 random bits per element, 8 elements per
// Philox4x32-10 call.  Float4 group g (elements 4g .. 4g+3 of the flat tensor a kernel writes) owns words 2 (g & 1) and
// 2 (g & 1) + 1 of the call with counter base + (g >> 1); element j of the group keeps its value iff field j -- low / high half
// of the two words -- is >= thresh16 = (uint32_t)(p * 2^16) (0: no dropout), and is scaled by 1 / (1 - p).
// Philox key of a dropout launch: `seed`, or with a device-resident epoch word (launches replayed from a HIP graph) seed ^ mix(epoch)
__device__ __forceinline__ uint64_t dropout_key(uint64_t seed, const uint64_t* epoch) {
    return epoch ? seed ^ (*epoch * 0x9E3779B97F4A7C15ull) : seed;
}

__device__ __forceinline__ void dropout_words(uint64_t base, uint64_t g, uint32_t c2, uint32_t stream, uint64_t seed, uint32_t& w0, uint32_t& w1) {
    const uint64_t ctr = base + (g >> 1);
    const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), c2, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    w0 = (g & 1) ? r.z : r.x;
    w1 = (g & 1) ? r.w : r.y;
}
#define POD_DROPOUT_THRESH16(p) ((uint32_t)((double)(p) * 65536.0))

// Native-RNG normals: Box-Muller on two 16-bit uniforms (one 32-bit Philox word per pair of normals,
// 8 normals per Philox4x32-10 call).  u1 = (a + 0.5) / 2^16 >= 2^-17 bounds the radius, so
//     |z| <= sqrt(-2 ln 2^-17) = 4.855 < POD_EPS_MAX,
// which is what lets K1 prune anchors EXACTLY (see k1_mc_merge_score.hip).  Statistical parity only: the
// eps-replay mode never comes here.  v_log_f32 is log2; v_sin/v_cos take their argument in revolutions.
__device__ __forceinline__ void box_muller16(uint32_t w, float& n0, float& n1) {
    const float u1 = ((float)(w & 0xFFFFu) + 0.5f) * 1.52587890625e-05f;   // (0,1)
    const float u2 = ((float)(w >> 16) + 0.5f) * 1.52587890625e-05f;       // revolutions
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    n0 = rad * __builtin_amdgcn_cosf(u2);
    n1 = rad * __builtin_amdgcn_sinf(u2);
}

struct f32x8n {
    float v[8];
};

// stream ids for the counter's 4th word
constexpr uint32_t STREAM_CLS = 0x636c7300u;   // classification logit samples (PI:291-295)
constexpr uint32_t STREAM_BOX = 0x626f7800u;   // box-delta samples (PI:351-356)

__device__ __forceinline__ f32x8n philox_normals8(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t stream) {
    const u32x4 r = philox4x32_10(u32x4{c0, c1, c2, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    f32x8n o;
    box_muller16(r.x, o.v[0], o.v[1]);
    box_muller16(r.y, o.v[2], o.v[3]);
    box_muller16(r.z, o.v[4], o.v[5]);
    box_muller16(r.w, o.v[6], o.v[7]);
    return o;
}

// ---------------------------------------------------------------------------------------------
// arithmetic restated from the reference, one rounding per op
// ---------------------------------------------------------------------------------------------

// torch.sigmoid on CPU: 1 / (1 + exp(-x)) with a correctly rounded divide.
__device__ __forceinline__ float sigmoid_ref(float x) { return __fdiv_rn(1.0f, 1.0f + expf(-x)); }
// native-RNG mode only (no bit-level comparison with the CPU is possible there): v_exp_f32 + v_rcp_f32.
__device__ __forceinline__ float sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// PI:216-222 merge of N runs, in the reference's association order.
//   quirk: acc = x0; acc += x0; acc += x1; ... acc += x_{N-2}; acc /= N
//   true : acc = x0; acc += x1; ... acc += x_{N-1};           acc /= N
// `term_run(t)` is the run whose value is term t (t = 0..N-1).
__device__ __forceinline__ int merge_term_run(int t, int quirk) { return (quirk && t > 0) ? t - 1 : t; }

// Class probability of one (anchor, class): PI:289-297.
//   no variance head : sigmoid(logit)
//   variance head    : mean_s sigmoid(logit + eps_s * sqrt(exp(var))), s = 0..S-1, summed in order, / S
//
// eps source.  Replay (parity mode): tensor (S, R_l, K) in the reference layout, every op the reference's.
// Native: Philox draws organised per group of 4 consecutive cells (the 4 anchors one K1 lane owns); normal
// q = j*S + s (j = hw & 3, s = sample) is component q&7 of the call with counter
// (hw>>2, level<<16 | a<<8 | k, q>>3, STREAM_CLS).  K1, K1b and K2b all come through this one function, so
// they see bit-identical draws and sums and nothing has to be stored.  Transcendentals use the hardware
// approximations in native mode (the draws differ from torch's anyway).
__device__ __forceinline__ float class_prob_cell(float logit, float logvar, bool has_var, int S, const float* replay,
                                                int64_t level_anchors, int K, int A, int level, int hw, int a, int k, uint64_t seed) {
    if (!has_var) return sigmoid_ref(logit);
    float acc = 0.0f;
    if (replay) {
        const float sigma = sqrtf(expf(logvar));
        const float* e = replay + ((int64_t)hw * A + a) * K + k;
        const int64_t stride_s = level_anchors * K;
        for (int s = 0; s < S; ++s) {
            const float x = logit + e[(int64_t)s * stride_s] * sigma;
            acc = acc + sigmoid_ref(x);
        }
        return __fdiv_rn(acc, (float)S);
    }
    const float sigma = __builtin_amdgcn_exp2f(0.7213475204444817f * logvar);   // sqrt(exp(v)) = 2^(v / (2 ln 2))
    const int q0 = (hw & 3) * S, q1 = q0 + S;
    const uint32_t c0 = (uint32_t)(hw >> 2), c1 = ((uint32_t)level << 16) | ((uint32_t)a << 8) | (uint32_t)k;
    for (int call = q0 >> 3; call * 8 < q1; ++call) {
        const u32x4 r = philox4x32_10(u32x4{c0, c1, (uint32_t)call, STREAM_CLS}, (uint32_t)seed, (uint32_t)(seed >> 32));
        float z[8];
        box_muller16(r.x, z[0], z[1]);
        box_muller16(r.y, z[2], z[3]);
        box_muller16(r.z, z[4], z[5]);
        box_muller16(r.w, z[6], z[7]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int q = call * 8 + c;
            if (q >= q0 && q < q1) acc += sigmoid_fast(fmaf(z[c], sigma, logit));
        }
    }
    return acc * __builtin_amdgcn_rcpf((float)S);
}

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
    // x / 1.0f == x exactly: with the reference's weights (1,1,1,1) (PI:175-176) the four IEEE divides are skipped
    const bool unit = wts[0] == 1.0f && wts[1] == 1.0f && wts[2] == 1.0f && wts[3] == 1.0f;
    const float dx = unit ? d0 : __fdiv_rn(d0, wts[0]);
    const float dy = unit ? d1 : __fdiv_rn(d1, wts[1]);
    float dw = unit ? d2 : __fdiv_rn(d2, wts[2]);
    float dh = unit ? d3 : __fdiv_rn(d3, wts[3]);
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
