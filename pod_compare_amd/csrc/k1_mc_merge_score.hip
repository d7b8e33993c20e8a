// K1 mc_merge_score -- the HBM-bound kernel of the path (the roofline kernel).
//
// Replaces (reference, /root/reference/src/probabilistic_inference/probabilistic_inference.py):
//   :211-270  dense merge of the N MC-dropout runs / ensemble members of box_cls, box_cls_var,
//             box_delta, box_reg_var (incl. the quirk of :216-222);
//   :289-297  classification sampling  mean_s sigmoid(logit + eps_s * sqrt(exp(var)));
//   :301,:304 max over classes and the score-threshold test (top-k itself is K2).
//
// Mapping to CDNA4.  The head tensors stay in the conv's NCHW layout; one plane (a*C + c) is
// H*W contiguous floats, so a wavefront reads 64 x 16 B = 1 KiB of one plane per load
// instruction and no permute_to_N_HWA_K copy is ever made.  Workgroups are 64*K threads:
//   * "cls" workgroups: wave k owns class k of one anchor shape a for 256 consecutive cells;
//     it streams the N runs of logit and log-variance planes (2N independent 16-B loads per
//     lane), merges them in the reference's association order, writes the merged planes,
//     computes the 4 class probabilities per lane and parks them in LDS; after one barrier
//     256 threads take one anchor each, reduce max/argmax over the K classes from LDS and
//     append (score, r) keys of anchors above the threshold to the level's candidate list with
//     one wave-aggregated atomic per wave.
//   * "box" workgroups: flat element-wise merge of the delta / reg_var planes (pure streaming).
// ~1.6k workgroups, ~11k waves at BASELINE size (R = 193374, N = 10): every CU holds several
// waves with >= 16 loads in flight each.  No MFMA: element-wise + reductions.
#include "pod_device.h"

namespace pod {

struct K1Params {
    PodLevel lv[POD_MAX_LEVELS];
    int32_t seg_begin[3 * POD_MAX_LEVELS + 1];   // workgroup ranges: [cls l..][delta l..][reg l..]
    int32_t chunks[POD_MAX_LEVELS];              // cls role: 256-cell chunks per anchor shape
    uint8_t vec_cls[POD_MAX_LEVELS];             // 16-B vector path usable (alignment + H*W % 4 == 0)
    uint8_t vec_delta[POD_MAX_LEVELS];
    uint8_t vec_reg[POD_MAX_LEVELS];
    int32_t n_levels, n_runs, A, K, D, has_cls_var, quirk, cls_samples;
    float score_thresh;
    uint64_t seed;
    float* mean_cls;
    float* mean_cls_var;
    float* mean_delta;
    float* mean_reg_var;
    uint64_t* cand_keys;
    int32_t* cand_count;
};

__device__ __forceinline__ float4 ld4(const float* p, int64_t i, int64_t n, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(p + i);
    float4 v;
    v.x = (i + 0 < n) ? p[i + 0] : 0.0f;
    v.y = (i + 1 < n) ? p[i + 1] : 0.0f;
    v.z = (i + 2 < n) ? p[i + 2] : 0.0f;
    v.w = (i + 3 < n) ? p[i + 3] : 0.0f;
    return v;
}
__device__ __forceinline__ void st4(float* p, int64_t i, int64_t n, bool vec, float4 v) {
    if (vec) {
        *reinterpret_cast<float4*>(p + i) = v;
        return;
    }
    if (i + 0 < n) p[i + 0] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }

// Merge N runs of 4 consecutive floats starting at element `i` of each run (run stride `rs`).
// Loads are issued in batches of 8 independent 16-B loads; adds follow the reference order.
__device__ __forceinline__ float4 merge_runs4(const float* base, int64_t rs, int64_t i, int64_t n, bool vec, int n_runs,
                                             int quirk) {
    float4 acc = ld4(base, i, n, vec);   // term 0 = run 0
    if (n_runs == 1) return acc;
    // remaining terms t = 1..N-1 read run (quirk ? t-1 : t); in quirk mode term 1 re-uses run 0.
    int t = 1;
    if (quirk) {
        acc = add4(acc, acc);
        t = 2;
    }
    for (; t < n_runs; t += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tt = t + j;
            if (tt < n_runs) v[j] = ld4(base + (int64_t)merge_term_run(tt, quirk) * rs, i, n, vec);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (t + j < n_runs) acc = add4(acc, v[j]);
    }
    const float fn = (float)n_runs;
    return float4{__fdiv_rn(acc.x, fn), __fdiv_rn(acc.y, fn), __fdiv_rn(acc.z, fn), __fdiv_rn(acc.w, fn)};
}

__global__ void __launch_bounds__(1024) k1_mc_merge_score(const K1Params P) {
    extern __shared__ __attribute__((aligned(16))) float lds_probs[];   // [K][256]
    const int b = blockIdx.x;
    const int L = P.n_levels;
    // locate role + level (scalar search over <= 24 segment starts)
    int seg = 0;
#pragma unroll 1
    while (seg + 1 < 3 * L && b >= P.seg_begin[seg + 1]) ++seg;
    const int role = seg / L;
    const int l = seg - role * L;
    const PodLevel& lv = P.lv[l];
    const int local_b = b - P.seg_begin[seg];
    const int HW = lv.H * lv.W;
    const int tid = threadIdx.x;

    if (role != 0) {
        // ---- box role: element-wise merge of delta (role 1) or reg_var (role 2) ----------------
        const bool is_delta = role == 1;
        const int C = is_delta ? 4 : P.D;
        const float* src = is_delta ? lv.delta : lv.reg_var;
        float* dst = (is_delta ? P.mean_delta : P.mean_reg_var);
        if (dst == nullptr || P.n_runs == 1) return;
        dst += (int64_t)lv.anchor_base * C;
        const int64_t rs = is_delta ? lv.run_stride_delta : lv.run_stride_reg;
        const bool vec = is_delta ? P.vec_delta[l] : P.vec_reg[l];
        const int64_t n = (int64_t)P.A * C * HW;
        const int64_t i = ((int64_t)local_b * blockDim.x + tid) * 4;
        if (i >= n) return;
        st4(dst, i, n, vec, merge_runs4(src, rs, i, n, vec, P.n_runs, P.quirk));
        return;
    }

    // ---- cls role -------------------------------------------------------------------------------
    const int K = P.K;
    const int chunks = P.chunks[l];
    const int a = local_b / chunks;
    const int chunk = local_b - a * chunks;
    const int k = tid >> 6;             // wave id = class
    const int lane = tid & 63;
    const int hw0 = chunk * 256 + lane * 4;
    const bool vec = P.vec_cls[l];
    const bool has_var = P.has_cls_var != 0;
    float4 prob = float4{0.f, 0.f, 0.f, 0.f};
    if (hw0 < HW) {
        const int64_t plane = (int64_t)(a * K + k) * HW;
        const int64_t n = plane + HW;   // bound for the scalar tail path
        const int64_t i = plane + hw0;
        const float4 logit = merge_runs4(lv.cls, lv.run_stride_cls, i, n, vec, P.n_runs, P.quirk);
        float4 lvar = float4{0.f, 0.f, 0.f, 0.f};
        if (has_var) lvar = merge_runs4(lv.cls_var, lv.run_stride_cls, i, n, vec, P.n_runs, P.quirk);
        if (P.n_runs > 1) {
            const int64_t off = (int64_t)lv.anchor_base * K;
            if (P.mean_cls) st4(P.mean_cls + off, i, n, vec, logit);
            if (has_var && P.mean_cls_var) st4(P.mean_cls_var + off, i, n, vec, lvar);
        }
        const float lg[4] = {logit.x, logit.y, logit.z, logit.w};
        const float vr[4] = {lvar.x, lvar.y, lvar.z, lvar.w};
        float pr[4];
        const int64_t RL = (int64_t)HW * P.A;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (hw0 + j) * P.A + a;
            ClsEps eps(lv.eps_cls, RL, K, r, k, P.seed, (uint32_t)(lv.anchor_base + r));
            pr[j] = (hw0 + j < HW) ? class_prob(lg[j], vr[j], has_var, P.cls_samples, eps) : 0.0f;
        }
        prob = float4{pr[0], pr[1], pr[2], pr[3]};
    }
    *reinterpret_cast<float4*>(&lds_probs[k * 256 + lane * 4]) = prob;
    __syncthreads();

    // ---- 256 threads: one anchor each; max/argmax over classes; candidate emission ------------------
    if (tid < 256) {
        const int hw = chunk * 256 + tid;
        float best = lds_probs[tid];
#pragma unroll 1
        for (int kk = 1; kk < K; ++kk) {
            const float v = lds_probs[kk * 256 + tid];
            best = (v > best) ? v : best;   // torch.max keeps the first maximum; argmax is re-derived in K2b
        }
        const bool pass = (hw < HW) && (best > P.score_thresh);
        const unsigned long long m = __ballot(pass);
        if (m != 0ull) {
            const int total = __popcll(m);
            const int mylane = tid & 63;
            int base = 0;
            if (mylane == 0) base = atomicAdd(&P.cand_count[l], total);
            base = __shfl(base, 0, 64);
            if (pass) {
                const int off = __popcll(m & ((1ull << mylane) - 1ull));
                P.cand_keys[(int64_t)lv.anchor_base + base + off] = make_key(best, hw * P.A + a);
            }
        }
    }
}

}  // namespace pod

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int pod_abi_version(void) { return POD_ABI_VERSION; }

extern "C" int pod_reset_counters(int32_t* counters, int32_t n, pod_stream_t stream) {
    if (!counters || n <= 0) return POD_E_INVALID;
    if (hipMemsetAsync(counters, 0, sizeof(int32_t) * (size_t)n, (hipStream_t)stream) != hipSuccess) return POD_E_LAUNCH;
    return POD_OK;
}

extern "C" int pod_mc_merge_score(const PodConfig* cfg, const PodLevel* levels, float* mean_cls, float* mean_cls_var,
                                  float* mean_delta, float* mean_reg_var, uint64_t* cand_keys, int32_t* cand_count,
                                  pod_stream_t stream) {
    if (!cfg || !levels || !cand_keys || !cand_count) return POD_E_INVALID;
    const int L = cfg->n_levels, K = cfg->num_classes, A = cfg->num_anchors, N = cfg->n_runs, D = cfg->cov_dims;
    if (L < 1 || L > POD_MAX_LEVELS || K < 1 || K > POD_MAX_CLASSES || A < 1 || N < 1 || N > POD_MAX_RUNS) return POD_E_INVALID;
    if (!(D == 0 || D == 4 || D == 10)) return POD_E_INVALID;
    if (cfg->has_cls_var && (cfg->cls_samples < 1 || cfg->cls_samples > POD_MAX_CLS_SAMPLES)) return POD_E_INVALID;
    pod::K1Params P;
    const int threads = 64 * K;
    for (int l = 0; l < L; ++l) {
        const PodLevel& lv = levels[l];
        if (!lv.cls || !lv.delta || lv.H < 1 || lv.W < 1) return POD_E_INVALID;
        if (cfg->has_cls_var && !lv.cls_var) return POD_E_INVALID;
        if (D > 0 && !lv.reg_var) return POD_E_INVALID;
        P.lv[l] = lv;
        const int64_t HW = (int64_t)lv.H * lv.W;
        P.chunks[l] = (int32_t)((HW + 255) / 256);
        const bool hw4 = (HW % 4) == 0;
        P.vec_cls[l] = hw4 && aligned16(lv.cls) && (lv.run_stride_cls % 4 == 0) && (!cfg->has_cls_var || aligned16(lv.cls_var)) &&
                       ((int64_t)lv.anchor_base * K % 4 == 0) && aligned16(mean_cls) && aligned16(mean_cls_var);
        P.vec_delta[l] = aligned16(lv.delta) && (lv.run_stride_delta % 4 == 0) && aligned16(mean_delta);
        P.vec_reg[l] = D > 0 && aligned16(lv.reg_var) && (lv.run_stride_reg % 4 == 0) && ((int64_t)A * D * HW % 4 == 0) &&
                       ((int64_t)lv.anchor_base * D % 4 == 0) && aligned16(mean_reg_var);
    }
    int32_t nb = 0, s = 0;
    for (int l = 0; l < L; ++l) {   // cls role
        P.seg_begin[s++] = nb;
        nb += A * P.chunks[l];
    }
    for (int role = 1; role <= 2; ++role)
        for (int l = 0; l < L; ++l) {
            P.seg_begin[s++] = nb;
            const int C = role == 1 ? 4 : D;
            const bool active = N > 1 && C > 0 && (role == 1 ? mean_delta != nullptr : mean_reg_var != nullptr);
            if (active) {
                const int64_t items = ((int64_t)A * C * levels[l].H * levels[l].W + 3) / 4;
                nb += (int32_t)((items + threads - 1) / threads);
            }
        }
    P.seg_begin[s] = nb;
    P.n_levels = L; P.n_runs = N; P.A = A; P.K = K; P.D = D;
    P.has_cls_var = cfg->has_cls_var; P.quirk = cfg->merge_quirk; P.cls_samples = cfg->cls_samples;
    P.score_thresh = cfg->score_thresh; P.seed = cfg->philox_seed;
    P.mean_cls = mean_cls; P.mean_cls_var = mean_cls_var; P.mean_delta = mean_delta; P.mean_reg_var = mean_reg_var;
    P.cand_keys = cand_keys; P.cand_count = cand_count;
    hipLaunchKernelGGL(pod::k1_mc_merge_score, dim3(nb), dim3(threads), sizeof(float) * 256 * K, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
