// pod_dump_cls_normals / pod_dump_box_normals -- test support: the in-kernel (native-RNG) draws, written out.
//
// The product path never stores a normal: K1b / K2b re-derive the classification draws of PI:291-294 from Philox counters
// (pod_device.h: class_prob_cell) and K3 the box-delta draws of PI:351-356 (pod_candidate.h: decode_candidate).  These two
// entry points evaluate the SAME counter -> normal maps and write the values in the reference's tensor layouts
//     eps_cls  : (cls_samples, H*W*A, K)   one tensor per level          (Normal(...).rsample((S,)),   PI:291-294)
//     eps_prop : (prop_samples, n, 4)      rows = the given anchors      (MVN.rsample((1000,)),        PI:351-356)
// so that a test can hand the oracle exactly the draws the product kernels used and compare the two end to end
// (tests/test_native_exact_gpu.py).  Not on the inference path; nothing else calls them.
#include "pod_device.h"
#include "../../include/pod_mi355x_test.h"
#include "pod_wino.h"

namespace pod {

// normal q = (hw & 3) * S + s of the group of 4 cells (hw >> 2): component q & 7 of Philox call q >> 3 -- class_prob_cell
__global__ void __launch_bounds__(256) k_dump_cls_normals(uint64_t seed, int level, int HW, int A, int K, int S, float* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (hw, a, k)
    const int64_t R = (int64_t)HW * A;
    if (t >= R * K) return;
    const int k = (int)(t % K);
    const int r = (int)(t / K);
    const int hw = r / A, a = r - hw * A;
    const uint32_t c0 = (uint32_t)(hw >> 2), c1 = ((uint32_t)level << 16) | ((uint32_t)a << 8) | (uint32_t)k;
    int have = -1;
    float z[8];
    for (int s = 0; s < S; ++s) {
        const int q = (hw & 3) * S + s;
        const int call = q >> 3;
        if (call != have) {
            const u32x4 w = philox4x32_10(u32x4{c0, c1, (uint32_t)call, STREAM_CLS}, (uint32_t)seed, (uint32_t)(seed >> 32));
            box_muller16(w.x, z[0], z[1]);
            box_muller16(w.y, z[2], z[3]);
            box_muller16(w.z, z[4], z[5]);
            box_muller16(w.w, z[6], z[7]);
            have = call;
        }
        float v = z[0];
#pragma unroll
        for (int c = 1; c < 8; ++c)
            if ((q & 7) == c) v = z[c];
        out[((int64_t)s * R + r) * K + k] = v;
    }
}

// sample s of anchor gid: components (s & 1) * 4 .. + 3 of Philox call (gid, s >> 1, 0, STREAM_BOX) -- decode_candidate
__global__ void __launch_bounds__(256) k_dump_box_normals(uint64_t seed, const int32_t* gids, int n, int S, float* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one thread per (s, i)
    if (t >= (int64_t)S * n) return;
    const int i = (int)(t % n), s = (int)(t / n);
    const f32x8n z = philox_normals8(seed, (uint32_t)gids[i], (uint32_t)(s >> 1), 0u, STREAM_BOX);
    float4 e;
    if (s & 1) e = float4{z.v[4], z.v[5], z.v[6], z.v[7]};
    else e = float4{z.v[0], z.v[1], z.v[2], z.v[3]};
    *reinterpret_cast<float4*>(out + ((int64_t)s * n + i) * 4) = e;
}

}  // namespace pod

extern "C" int pod_dump_cls_normals(const PodConfig* cfg, const PodLevel* levels, int32_t level, float* eps_cls, pod_stream_t stream) {
    if (!cfg || !levels || !eps_cls || level < 0 || level >= cfg->n_levels || cfg->n_levels > POD_MAX_LEVELS) return POD_E_INVALID;
    if (cfg->cls_samples < 1 || cfg->cls_samples > POD_MAX_CLS_SAMPLES || cfg->num_classes < 1 || cfg->num_classes > POD_MAX_CLASSES)
        return POD_E_INVALID;
    const int HW = levels[level].H * levels[level].W;
    const int64_t items = (int64_t)HW * cfg->num_anchors * cfg->num_classes;
    hipLaunchKernelGGL(pod::k_dump_cls_normals, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cfg->philox_seed,
                       (int)level, HW, (int)cfg->num_anchors, (int)cfg->num_classes, (int)cfg->cls_samples, eps_cls);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_dump_box_normals(const PodConfig* cfg, const int32_t* global_anchor_ids, int32_t n, float* eps_prop, pod_stream_t stream) {
    if (!cfg || !global_anchor_ids || !eps_prop || n < 0) return POD_E_INVALID;
    if (cfg->prop_samples < 2 || cfg->prop_samples > POD_MAX_PROP_SAMPLES) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    const int64_t items = (int64_t)cfg->prop_samples * n;
    hipLaunchKernelGGL(pod::k_dump_box_normals, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cfg->philox_seed,
                       global_anchor_ids, (int)n, (int)cfg->prop_samples, eps_prop);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

// pod_debug_f16_split2 -- test support: the 2-way f16 split of the power-of-two-scaled operands of the round-5 split kernels (pod_wino.h:
// wino_f16_split2, the functions the kernels' loops call): terms[2][n] f16 bit patterns of x[n] * scale (n even).
namespace pod {
__global__ void __launch_bounds__(256) k_debug_f16_split2(const float* __restrict__ x, float scale, uint16_t* __restrict__ terms, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t w[2];
    wino_f16_split2(x[i], x[i + 1], scale, w);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        terms[t * n + i] = (uint16_t)(w[t] & 0xFFFFu);
        terms[t * n + i + 1] = (uint16_t)(w[t] >> 16);
    }
}
}  // namespace pod

extern "C" int pod_debug_f16_split2(const float* x, float scale, void* terms, int64_t n, pod_stream_t stream) {
    if (!x || !terms || n < 0 || (n & 1) != 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    hipLaunchKernelGGL(pod::k_debug_f16_split2, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale,
                       reinterpret_cast<uint16_t*>(terms), n);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
