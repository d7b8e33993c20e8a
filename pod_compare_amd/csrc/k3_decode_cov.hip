// K3 decode_cov -- per-candidate box decode with covariance propagation.
//
// Replaces:
//   modeling_utils.py:4-22          covariance_output_to_cholesky (diag sqrt(exp), optional strict-lower)
//   probabilistic_inference.py:344-368  MVN(delta, L).rsample((1000,)) -> decode every sample ->
//                                       sample mean + unbiased covariance
//   inference_utils.py:510-547      SampleBox2BoxTransform.apply_samples_deltas
//   inference_utils.py:337-371      compute_mean_covariance_torch (two-pass, / (S-1))
//   probabilistic_inference.py:323-331, :369-374  epistemic covariance over the N runs, added on top
//   probabilistic_inference.py:375-385  deterministic decode when there is no reg_var head
//
// The reference materialises ~0.7 GB of temporaries here (incl. (n,1000,4,4) outer products).
// On CDNA4: one wavefront per candidate, lane l owns the 16 consecutive samples [16l, 16l+16)
// (64 lanes x 16 = 1024 >= S), the 64 decoded coordinates stay in VGPRs between the two passes,
// and the partial sums are combined in the 16-row block cascade order of torch's CPU sum so the
// result tracks the reference to fp32 round-off.  Algorithmic HBM traffic is 12 + 4N floats in and
// 20 floats out per candidate: compute/latency-bound, not an HBM-roofline kernel.
#include "pod_candidate.h"

namespace pod {

// One 64-thread workgroup (= one wavefront) per candidate: barriers are wave-local and free.
__global__ void __launch_bounds__(64) k3_decode_cov(const K3Params P) {
    __shared__ float part[64 * 10];   // [lane][component]
    __shared__ float small[16 + 4 * POD_MAX_RUNS];
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= min(*P.n_total, P.n_capacity)) return;
    const float4 d4 = *reinterpret_cast<const float4*>(P.cand_delta + (size_t)i * 4);
    const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
    const Box anc = load_box(P.cand_anchor, i);
    float rv[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t gid = 0;
    if (P.D > 0) {
#pragma unroll
        for (int c = 0; c < 10; ++c)
            if (c < P.D) rv[c] = P.cand_reg_var[(size_t)i * P.D + c];
        gid = (uint32_t)(P.anchor_base[P.cand_level[i]] + P.cand_anchor_idx[i]);
    }
    decode_candidate(P, i, lane, dl, rv, anc, gid, P.n_runs > 1 ? P.cand_run_delta + (size_t)i * P.n_runs * 4 : nullptr, part, small);
}

// K2b + K3 fused (native draws): the wavefront that gathered a candidate decodes it.  The merged deltas and
// log-variances stay in registers (lane c of the gather owns channel c), the N runs' raw deltas go through LDS, so the
// candidate arrays in HBM are written for the later kernels but never read back here: one launch and two dependent
// global round trips less than K2b -> K3.
struct K23Params {
    K2bParams g;
    K3Params d;
};

__global__ void __launch_bounds__(64) k23_gather_decode(const K23Params P) {
    __shared__ float part[64 * 10];
    __shared__ float small[16 + 4 * POD_MAX_RUNS];
    __shared__ float run_delta[4 * POD_MAX_RUNS];
    const int lane = threadIdx.x;
    GatheredCandidate c;
    if (!gather_candidate(P.g, blockIdx.x, lane, run_delta, c)) return;   // wave-uniform
    const int K = P.g.K, D = P.g.D, nvar = P.g.has_cls_var ? K : 0;
    float dl[4], rv[10];
#pragma unroll
    for (int j = 0; j < 4; ++j) dl[j] = __shfl(c.merged, K + nvar + j, 64);
#pragma unroll
    for (int j = 0; j < 10; ++j) rv[j] = (j < D) ? __shfl(c.merged, K + nvar + 4 + (j < D ? j : 0), 64) : 0.0f;
    const Box anc = load_box(P.g.anchors, P.g.lv[c.level].anchor_base + c.r);
    const uint32_t gid = (uint32_t)(P.g.lv[c.level].anchor_base + c.r);
    __syncthreads();   // run_delta (LDS) written by the delta lanes
    decode_candidate(P.d, c.dst, lane, dl, rv, anc, gid, run_delta, part, small);
}

}  // namespace pod

extern "C" int pod_decode_cov(const PodConfig* cfg, const PodLevel* levels, const int32_t* n_total, int32_t n_capacity,
                              const float* cand_delta, const float* cand_reg_var, const float* cand_anchor,
                              const float* cand_run_delta, const int32_t* cand_anchor_idx, const int32_t* cand_level,
                              const float* eps_prop, int32_t n_replay, float* boxes, float* cov, pod_stream_t stream) {
    if (!cfg || !levels || !n_total || n_capacity < 1 || !cand_delta || !cand_anchor || !boxes || !cov) return POD_E_INVALID;
    if (cfg->cov_dims > 0 && (!cand_reg_var || !cand_anchor_idx || !cand_level)) return POD_E_INVALID;
    if (cfg->cov_dims > 0 && (cfg->prop_samples < 2 || cfg->prop_samples > POD_MAX_PROP_SAMPLES)) return POD_E_INVALID;
    if (cfg->n_runs > 1 && !cand_run_delta) return POD_E_INVALID;
    if (cfg->n_runs > POD_MAX_RUNS) return POD_E_INVALID;
    if (eps_prop && n_replay < 1) return POD_E_INVALID;
    pod::K3Params P;
    for (int l = 0; l < cfg->n_levels; ++l) P.anchor_base[l] = levels[l].anchor_base;
    P.n_runs = cfg->n_runs; P.D = cfg->cov_dims; P.S = cfg->prop_samples; P.n_capacity = n_capacity; P.n_replay = n_replay;
    for (int c = 0; c < 4; ++c) P.wts[c] = cfg->box_weights[c];
    P.seed = cfg->philox_seed; P.n_total = n_total; P.cand_delta = cand_delta; P.cand_reg_var = cand_reg_var;
    P.cand_anchor = cand_anchor; P.cand_run_delta = cand_run_delta; P.cand_anchor_idx = cand_anchor_idx;
    P.cand_level = cand_level; P.eps_prop = eps_prop; P.boxes = boxes; P.cov = cov;
    hipLaunchKernelGGL(pod::k3_decode_cov, dim3(n_capacity), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_gather_decode(const PodConfig* cfg, const PodLevel* levels, const float* anchors, const uint64_t* sel_keys,
                                 const int32_t* sel_count, int32_t* cand_anchor_idx, int32_t* cand_level, float* cand_score,
                                 int32_t* cand_class, float* cand_probs, float* cand_delta, float* cand_reg_var, float* cand_anchor,
                                 float* cand_run_delta, int32_t* n_total, float* boxes, float* cov, pod_stream_t stream) {
    if (!cfg || !levels || !anchors || !sel_keys || !sel_count || !cand_anchor_idx || !cand_level || !cand_score || !cand_class ||
        !cand_probs || !cand_delta || !cand_anchor || !n_total || !boxes || !cov)
        return POD_E_INVALID;
    if (cfg->cov_dims > 0 && !cand_reg_var) return POD_E_INVALID;
    if (cfg->cov_dims > 0 && (cfg->prop_samples < 2 || cfg->prop_samples > POD_MAX_PROP_SAMPLES)) return POD_E_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > POD_MAX_LEVELS || cfg->n_runs < 1 || cfg->n_runs > POD_MAX_RUNS) return POD_E_INVALID;
    if (cfg->n_levels * cfg->topk > POD_MAX_CANDIDATES * 4) return POD_E_INVALID;
    if (2 * cfg->num_classes + 4 + cfg->cov_dims > 64) return POD_E_INVALID;
    for (int l = 0; l < cfg->n_levels; ++l)
        if (levels[l].eps_cls) return POD_E_INVALID;   // native draws only: eps-replay uses pod_gather_candidates + pod_decode_cov
    pod::K23Params P;
    pod::K2bParams& G = P.g;
    for (int l = 0; l < cfg->n_levels; ++l) G.lv[l] = levels[l];
    G.n_levels = cfg->n_levels; G.n_runs = cfg->n_runs; G.A = cfg->num_anchors; G.K = cfg->num_classes; G.D = cfg->cov_dims;
    G.has_cls_var = cfg->has_cls_var; G.quirk = cfg->merge_quirk; G.cls_samples = cfg->cls_samples; G.topk = cfg->topk;
    G.seed = cfg->philox_seed; G.anchors = anchors; G.sel_keys = sel_keys; G.sel_count = sel_count;
    G.cand_anchor_idx = cand_anchor_idx; G.cand_level = cand_level; G.cand_score = cand_score; G.cand_class = cand_class;
    G.cand_probs = cand_probs; G.cand_delta = cand_delta; G.cand_reg_var = cand_reg_var; G.cand_anchor = cand_anchor;
    G.cand_run_delta = cfg->n_runs > 1 ? cand_run_delta : nullptr; G.n_total = n_total;
    pod::K3Params& Dp = P.d;
    for (int l = 0; l < cfg->n_levels; ++l) Dp.anchor_base[l] = levels[l].anchor_base;
    Dp.n_runs = cfg->n_runs; Dp.D = cfg->cov_dims; Dp.S = cfg->prop_samples; Dp.n_capacity = cfg->n_levels * cfg->topk; Dp.n_replay = 0;
    for (int c = 0; c < 4; ++c) Dp.wts[c] = cfg->box_weights[c];
    Dp.seed = cfg->philox_seed; Dp.n_total = n_total; Dp.cand_delta = cand_delta; Dp.cand_reg_var = cand_reg_var;
    Dp.cand_anchor = cand_anchor; Dp.cand_run_delta = cand_run_delta; Dp.cand_anchor_idx = cand_anchor_idx;
    Dp.cand_level = cand_level; Dp.eps_prop = nullptr; Dp.boxes = boxes; Dp.cov = cov;
    hipLaunchKernelGGL(pod::k23_gather_decode, dim3(cfg->n_levels * cfg->topk), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
