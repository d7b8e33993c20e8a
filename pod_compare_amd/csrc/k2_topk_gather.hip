// K2 level_topk + K2b gather_candidates.
//
// Replaces (probabilistic_inference.py): `predicted_prob.topk(num_topk)` and the
// `> test_score_thresh` filter :300-308 (K1 already applied the threshold, which commutes with
// top-k), the per-level gathers :305-338 and the level concatenation :341-342 / :387-388.
//
// K2: one 1024-thread workgroup per FPN level.  Keys are 64-bit (score bits, ~anchor index), all
// distinct, so "top-k, ties to the lower index" is a plain descending sort.  Up to 2048
// candidates are sorted directly in LDS (bitonic network); a level with more candidates first runs
// an 8-pass MSB radix select (LDS histograms) to find the k-th largest key, then compacts.
// K2b: one thread per selected candidate re-derives the class probabilities with K1's own device
// function (bit-identical), and gathers merged deltas / log-variances, the anchor and every run's
// raw delta into the level-concatenated candidate arrays.
#include "pod_candidate.h"

namespace pod {

constexpr int TOPK_THREADS = 1024;
constexpr int SORT_CAP = POD_MAX_TOPK;   // 2048 keys = 16 KiB LDS

__device__ __forceinline__ void bitonic_sort_desc(uint64_t* s, int n_pow2, int tid, int nthreads) {
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n_pow2; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = s[i], b = s[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) {
                        s[i] = b;
                        s[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

struct K2Params {
    int32_t anchor_base[POD_MAX_LEVELS];
    int32_t topk;
    const uint64_t* cand_keys;
    int32_t* cand_count;
    uint64_t* sel_keys;
    int32_t* sel_count;
};

__global__ void __launch_bounds__(TOPK_THREADS) k2_level_topk(const K2Params P) {
    __shared__ uint64_t s_keys[SORT_CAP];
    __shared__ uint32_t s_hist[256];
    __shared__ uint64_t s_prefix;
    __shared__ int32_t s_remaining, s_fill, s_bucket;
    const int l = blockIdx.x;
    const int tid = threadIdx.x;
    const uint64_t* keys = P.cand_keys + P.anchor_base[l];
    const int C = P.cand_count[l];
    __syncthreads();
    if (tid == 0) P.cand_count[l] = 0;   // consumed: the next image's K1 appends from zero (no reset launch per image)
    const int k = min(P.topk, C);
    int n_sort;
    if (C <= SORT_CAP) {
        n_sort = 1;
        while (n_sort < C) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += TOPK_THREADS) s_keys[i] = (i < C) ? keys[i] : 0ull;
        __syncthreads();
    } else {
        // Radix select from the top byte down, until the keys that can still be in the top-k fit the LDS sort buffer:
        // after a pass the candidates are {keys above the chosen bucket} + {the bucket}; typically 2-4 passes.
        // 8 independent 8-byte loads per thread per step (a lone load per iteration made every pass latency-bound).
        if (tid == 0) {
            s_prefix = 0ull;
            s_remaining = k;
            s_fill = 0;
        }
        __syncthreads();
        uint64_t lower_bound = 0ull;     // every key >= lower_bound is still a top-k candidate; there are <= SORT_CAP of them at exit
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            if (tid < 256) s_hist[tid] = 0u;
            __syncthreads();
            const uint64_t prefix = s_prefix;
            const uint64_t himask = (pass == 7) ? 0ull : (~0ull << (shift + 8));
            for (int i0 = 0; i0 < C; i0 += TOPK_THREADS * 8) {
                uint64_t kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * TOPK_THREADS + tid;
                    kk[u] = (i < C) ? keys[i] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * TOPK_THREADS + tid;
                    const bool live = i < C && (kk[u] & himask) == prefix;
                    // top digits of score keys are heavily skewed: one LDS atomic when the whole wavefront agrees
                    const uint32_t digit = (uint32_t)(kk[u] >> shift) & 255u;
                    const unsigned long long lm = __ballot(live);
                    if (lm != 0ull) {
                        const int leader = __ffsll((long long)lm) - 1;
                        const uint32_t d0 = __shfl(digit, leader, 64);
                        if (__ballot(live && digit == d0) == lm) {
                            if ((threadIdx.x & 63) == leader) atomicAdd(&s_hist[d0], (uint32_t)__popcll(lm));
                        } else if (live) {
                            atomicAdd(&s_hist[digit], 1u);
                        }
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining;
                int d = 255;
                for (; d > 0; --d) {
                    const int c = (int)s_hist[d];
                    if (c >= rem) break;
                    rem -= c;
                }
                s_remaining = rem;                                  // still needed from bucket d
                s_prefix = prefix | ((uint64_t)d << shift);
                s_bucket = (int)s_hist[d];
            }
            __syncthreads();
            lower_bound = s_prefix;
            // candidates = (k - remaining) keys above the bucket + the bucket itself
            if ((k - s_remaining) + s_bucket <= SORT_CAP) break;
        }
        const int n_cand = (k - s_remaining) + s_bucket;            // exact count of keys >= lower_bound
        n_sort = 1;
        while (n_sort < n_cand) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += TOPK_THREADS) s_keys[i] = 0ull;
        __syncthreads();
        for (int i0 = 0; i0 < C; i0 += TOPK_THREADS * 8) {
            uint64_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                kk[u] = (i < C) ? keys[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                if (i < C && kk[u] >= lower_bound) s_keys[atomicAdd(&s_fill, 1)] = kk[u];
            }
        }
        __syncthreads();
    }
    bitonic_sort_desc(s_keys, n_sort, tid, TOPK_THREADS);
    uint64_t* out = P.sel_keys + (int64_t)l * P.topk;
    for (int i = tid; i < k; i += TOPK_THREADS) out[i] = s_keys[i];
    if (tid == 0) P.sel_count[l] = k;
}

__global__ void __launch_bounds__(64) k2b_gather(const K2bParams P) {
    GatheredCandidate g;
    gather_candidate(P, blockIdx.x, threadIdx.x, nullptr, g);
}

}  // namespace pod

extern "C" int pod_level_topk(const PodConfig* cfg, const PodLevel* levels, const uint64_t* cand_keys,
                              int32_t* cand_count, uint64_t* sel_keys, int32_t* sel_count, pod_stream_t stream) {
    if (!cfg || !levels || !cand_keys || !cand_count || !sel_keys || !sel_count) return POD_E_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > POD_MAX_LEVELS || cfg->topk < 1 || cfg->topk > POD_MAX_TOPK) return POD_E_INVALID;
    pod::K2Params P;
    for (int l = 0; l < cfg->n_levels; ++l) P.anchor_base[l] = levels[l].anchor_base;
    P.topk = cfg->topk; P.cand_keys = cand_keys; P.cand_count = cand_count; P.sel_keys = sel_keys; P.sel_count = sel_count;
    hipLaunchKernelGGL(pod::k2_level_topk, dim3(cfg->n_levels), dim3(pod::TOPK_THREADS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_gather_candidates(const PodConfig* cfg, const PodLevel* levels, const float* anchors,
                                     const uint64_t* sel_keys, const int32_t* sel_count, int32_t* cand_anchor_idx,
                                     int32_t* cand_level, float* cand_score, int32_t* cand_class, float* cand_probs,
                                     float* cand_delta, float* cand_reg_var, float* cand_anchor, float* cand_run_delta,
                                     int32_t* n_total, pod_stream_t stream) {
    if (!cfg || !levels || !anchors || !sel_keys || !sel_count || !cand_anchor_idx || !cand_level || !cand_score ||
        !cand_class || !cand_probs || !cand_delta || !cand_anchor || !n_total)
        return POD_E_INVALID;
    if (cfg->cov_dims > 0 && !cand_reg_var) return POD_E_INVALID;
    if (cfg->n_levels * cfg->topk > POD_MAX_CANDIDATES * 4) return POD_E_INVALID;
    if (2 * cfg->num_classes + 4 + cfg->cov_dims > 64) return POD_E_INVALID;
    pod::K2bParams P;
    for (int l = 0; l < cfg->n_levels; ++l) P.lv[l] = levels[l];
    P.n_levels = cfg->n_levels; P.n_runs = cfg->n_runs; P.A = cfg->num_anchors; P.K = cfg->num_classes; P.D = cfg->cov_dims;
    P.has_cls_var = cfg->has_cls_var; P.quirk = cfg->merge_quirk; P.cls_samples = cfg->cls_samples; P.topk = cfg->topk;
    P.seed = cfg->philox_seed; P.anchors = anchors; P.sel_keys = sel_keys; P.sel_count = sel_count;
    P.cand_anchor_idx = cand_anchor_idx; P.cand_level = cand_level; P.cand_score = cand_score; P.cand_class = cand_class;
    P.cand_probs = cand_probs; P.cand_delta = cand_delta; P.cand_reg_var = cand_reg_var; P.cand_anchor = cand_anchor;
    P.cand_run_delta = cfg->n_runs > 1 ? cand_run_delta : nullptr; P.n_total = n_total;
    const int slots = cfg->n_levels * cfg->topk;
    hipLaunchKernelGGL(pod::k2b_gather, dim3(slots), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
