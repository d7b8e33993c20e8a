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

constexpr int K3_MAX_SAMPLES = POD_MAX_PROP_SAMPLES;
constexpr int WAVE_SCRATCH = 10 * 64 + (16 + 4 * POD_MAX_RUNS) + 4 * POD_MAX_RUNS;   // part + small + run_delta, floats per wavefront

// One 64-thread workgroup (= one wavefront) per candidate (eps-replay parity mode and the call-by-call path); the lane's
// 16 samples never leave its registers.
__global__ void __launch_bounds__(64) k3_decode_cov(const K3Params P) {
    __shared__ __attribute__((aligned(16))) float part[10 * 64];   // [component][lane]
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
    decode_candidate<false>(P, i, lane, dl, rv, anc, gid, P.n_runs > 1 ? P.cand_run_delta + (size_t)i * P.n_runs * 4 : nullptr, nullptr,
                            part, small);
}

// K2b + K3 fused (native draws): the workgroup that gathered a candidate decodes it; merged deltas / log-variances / per-run
// deltas never leave the workgroup (the candidate arrays in HBM are written for the later kernels, never read back here).
// 256-thread workgroups, two shapes chosen on the device by the candidate count n (uniform over the grid):
//   n <= K23_SMALL_MAX (every image of a typical detector: a few hundred candidates, fewer workgroups than CUs):
//     one candidate per workgroup; wavefront 0 gathers (lane = channel), all four wavefronts draw and decode the 1000
//     samples into LDS (Philox + Box-Muller + exp: ~6 000 VALU instructions when one wavefront did it alone, 14 of the
//     kernel's 22 us), wavefront 0 forms the moments in the reference's summation order;
//   n  > K23_SMALL_MAX: one candidate per WAVEFRONT (workgroup g takes rows 4g .. 4g+3), samples in registers: with
//     thousands of candidates every SIMD has work anyway, and three idle wavefronts per candidate would only cost occupancy.
struct K23Params {
    K2bParams g;
    K3Params d;
    int32_t small_max;   // candidate counts up to this use the 4-wavefronts-per-candidate shape
};

constexpr int K23_THREADS = 256;
constexpr int K23_SMALL_MAX = 1024;

__global__ void __launch_bounds__(K23_THREADS) k23_gather_decode(const K23Params P) {
    // small shape: xs | part | small | run_delta | hand;  big shape: 4 x WAVE_SCRATCH carved from the same array
    __shared__ __attribute__((aligned(16))) float lds[4 * K3_MAX_SAMPLES + WAVE_SCRATCH + 24];
    static_assert(4 * WAVE_SCRATCH <= 4 * K3_MAX_SAMPLES + WAVE_SCRATCH + 24 && WAVE_SCRATCH % 4 == 0, "per-wave scratch fits / stays aligned");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = P.g.K, D = P.g.D, nvar = P.g.has_cls_var ? K : 0;
    const int n = *P.g.n_total;
    GatheredCandidate c;
    POD_STAMP(blockIdx.x, 0);
    if (n > P.small_max) {
        float* part = lds + wave * WAVE_SCRATCH;
        float* small = part + 10 * 64;
        float* run_delta = small + 16 + 4 * POD_MAX_RUNS;
        if (!gather_candidate(P.g, (int)blockIdx.x * 4 + wave, lane, run_delta, c)) return;   // wave-uniform
        float dl[4], rv[10];
#pragma unroll
        for (int j = 0; j < 4; ++j) dl[j] = __shfl(c.merged, K + nvar + j, 64);
#pragma unroll
        for (int j = 0; j < 10; ++j) rv[j] = (j < D) ? __shfl(c.merged, K + nvar + 4 + (j < D ? j : 0), 64) : 0.0f;
        const Box anc{c.anchor.x, c.anchor.y, c.anchor.z, c.anchor.w};
        wave_sync();   // run_delta (LDS) written by the delta lanes of this wavefront
        decode_candidate<false>(P.d, c.dst, lane, dl, rv, anc, (uint32_t)(P.g.lv[c.level].anchor_base + c.r), run_delta, nullptr, part, small);
        return;
    }
    float4* xs = reinterpret_cast<float4*>(lds);
    float* part = lds + 4 * K3_MAX_SAMPLES;
    float* small = part + 10 * 64;
    float* run_delta = small + 16 + 4 * POD_MAX_RUNS;
    float* hand = run_delta + 4 * POD_MAX_RUNS;    // merged deltas, reg_var entries, anchor, gid, live flag
    if (tid < 64) {
        const bool live = gather_candidate(P.g, blockIdx.x, lane, run_delta, c);   // wave-uniform
        if (lane == 0) hand[19] = live ? 1.0f : 0.0f;
        if (live) {
            if (lane >= K + nvar && lane < K + nvar + 4 + D) hand[lane - K - nvar] = c.merged;
            if (lane == 0) {
                const float4 a = c.anchor;
                hand[14] = a.x; hand[15] = a.y; hand[16] = a.z; hand[17] = a.w;
                hand[18] = __uint_as_float((uint32_t)(P.g.lv[c.level].anchor_base + c.r));
            }
        }
    }
    __syncthreads();
    if (hand[19] == 0.0f) return;
    POD_STAMP(blockIdx.x, 4);
    const float dl[4] = {hand[0], hand[1], hand[2], hand[3]};
    const Box anc{hand[14], hand[15], hand[16], hand[17]};
    float rv[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) rv[j] = j < D ? hand[4 + j] : 0.0f;
    if (D > 0) {
        generate_samples(P.d, tid, K23_THREADS, dl, rv, anc, __float_as_uint(hand[18]), xs);
        __syncthreads();
    }
    POD_STAMP(blockIdx.x, 5);
    if (tid >= 64) return;
    decode_candidate<true>(P.d, c.dst, lane, dl, rv, anc, 0u, run_delta, xs, part, small);
    POD_STAMP(blockIdx.x, 6);
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

extern "C" int pod_gather_decode(const PodConfig* cfg, const PodLevel* levels, const float* anchors, const uint64_t* cat_keys,
                                 const int32_t* cat_level, const int32_t* n_total, int32_t* cand_count, const float* probs_dense,
                                 int32_t* cand_anchor_idx, int32_t* cand_level, float* cand_score,
                                 int32_t* cand_class, float* cand_probs, float* cand_delta, float* cand_reg_var, float* cand_anchor,
                                 float* cand_run_delta, float* boxes, float* cov, pod_stream_t stream) {
    if (!cfg || !levels || !anchors || !cat_keys || !cat_level || !n_total || !cand_count || !cand_anchor_idx || !cand_level ||
        !cand_score || !cand_class || !cand_probs || !cand_delta || !cand_anchor || !boxes || !cov)
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
    G.seed = cfg->philox_seed; G.anchors = anchors; G.cat_keys = cat_keys; G.cat_level = cat_level; G.n_total = n_total;
    G.cand_count = cand_count; G.probs_dense = probs_dense;
    G.cand_anchor_idx = cand_anchor_idx; G.cand_level = cand_level; G.cand_score = cand_score; G.cand_class = cand_class;
    G.cand_probs = cand_probs; G.cand_delta = cand_delta; G.cand_reg_var = cand_reg_var; G.cand_anchor = cand_anchor;
    G.cand_run_delta = cfg->n_runs > 1 ? cand_run_delta : nullptr;
    pod::K3Params& Dp = P.d;
    for (int l = 0; l < cfg->n_levels; ++l) Dp.anchor_base[l] = levels[l].anchor_base;
    Dp.n_runs = cfg->n_runs; Dp.D = cfg->cov_dims; Dp.S = cfg->prop_samples; Dp.n_capacity = cfg->n_levels * cfg->topk; Dp.n_replay = 0;
    for (int c = 0; c < 4; ++c) Dp.wts[c] = cfg->box_weights[c];
    Dp.seed = cfg->philox_seed; Dp.n_total = n_total; Dp.cand_delta = cand_delta; Dp.cand_reg_var = cand_reg_var;
    Dp.cand_anchor = cand_anchor; Dp.cand_run_delta = cand_run_delta; Dp.cand_anchor_idx = cand_anchor_idx;
    Dp.cand_level = cand_level; Dp.eps_prop = nullptr; Dp.boxes = boxes; Dp.cov = cov;
    P.small_max = pod::K23_SMALL_MAX;   // swept 1024 / 2048 / 3072 / 8192 over n = 490 .. 4594 (tools/sweep_candidates.py): flat up to
                                        // ~2000, above that the one-wavefront shape wins by up to 30 us
    hipLaunchKernelGGL(pod::k23_gather_decode, dim3(cfg->n_levels * cfg->topk), dim3(pod::K23_THREADS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

#ifdef POD_TRACE
#include <stdio.h>
extern "C" int pod_trace_dump(void) {   // diagnostics build only (not in include/pod_mi355x.h)
    static long long host[4096 * 8];
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pod_trace), sizeof(host)) != hipSuccess) return POD_E_LAUNCH;
    long long t0 = 0;
    int n = 0;
    for (int i = 0; i < 4096; ++i)
        if (host[i * 8 + 6]) {
            if (!t0 || host[i * 8] < t0) t0 = host[i * 8];
            ++n;
        }
    fprintf(stderr, "trace: %d live workgroups; 10 ns ticks from the earliest start: start | keys | merged | probs | handed | generated | end\n", n);
    for (int i = 0; i < 4096; ++i)
        if (host[i * 8 + 6] && (i < 6 || i % 50 == 0 || i >= n - 3)) {
            fprintf(stderr, " wg %4d", i);
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %5lld", host[i * 8 + k] - t0);
            fprintf(stderr, "  | tail: blocksums | mean | pass2 | cov | epi:");
            for (int k = 0; k < 5; ++k) fprintf(stderr, " %5lld", host[(i + 2048) * 8 + k] - t0);
            fprintf(stderr, "\n");
        }
    return POD_OK;
}
#endif
