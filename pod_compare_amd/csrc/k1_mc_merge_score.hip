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
    float skip_logit;                            // native mode: logit + EPS_MAX*sigma <= skip_logit can never pass the threshold
    uint64_t seed;
    float* mean_cls;
    float* mean_cls_var;
    float* mean_delta;
    float* mean_reg_var;
    uint64_t* cand_keys;
    int32_t* cand_count;
    uint64_t* maybe_bits;    // prune mode: bitmap of the anchors that MAY pass the threshold (exact superset); scored by K1b
    int32_t word_begin[POD_MAX_LEVELS + 1];   // bitmap: level l owns words [word_begin[l], word_begin[l+1]); PLANE (a, k) owns wpa[l] of them
    int32_t wpa[POD_MAX_LEVELS];              // words per plane = ceil(H*W / 64); bit (a, k, hw) = word (a*K + k)*wpa + hw/64, bit hw%64
    int32_t pseg_begin[3 * POD_MAX_LEVELS + 1];   // workgroup ranges of k1_prune_stream (256 threads): [cls pair l..][delta l..][reg l..]
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* p, int64_t i, int64_t n) {
    if (VEC) {
        // streamed once: non-temporal (global_load_dwordx4 ... nt), measured -0.5..-1 us per launch vs default policy
        const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + i));
        return float4{v.x, v.y, v.z, v.w};
    }
    float4 v;
    v.x = (i + 0 < n) ? p[i + 0] : 0.0f;
    v.y = (i + 1 < n) ? p[i + 1] : 0.0f;
    v.z = (i + 2 < n) ? p[i + 2] : 0.0f;
    v.w = (i + 3 < n) ? p[i + 3] : 0.0f;
    return v;
}
template <bool VEC, bool NT = false>
__device__ __forceinline__ void st4(float* p, int64_t i, int64_t n, float4 v) {
    if (VEC) {
        if (NT) {   // written once, re-read only sparsely (K1b): keep it out of the way of the streaming loads
            const f32x4_t w = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(w, reinterpret_cast<f32x4_t*>(p + i));
        } else {
            *reinterpret_cast<float4*>(p + i) = v;
        }
        return;
    }
    if (i + 0 < n) p[i + 0] = v.x;
    if (i + 1 < n) p[i + 1] = v.y;
    if (i + 2 < n) p[i + 2] = v.z;
    if (i + 3 < n) p[i + 3] = v.w;
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__device__ __forceinline__ float4 div4(float4 a, float d) {
    return float4{__fdiv_rn(a.x, d), __fdiv_rn(a.y, d), __fdiv_rn(a.z, d), __fdiv_rn(a.w, d)};
}

// CNT straight-line independent 16-B loads of runs run0..run0+CNT-1 of NT tensors (same index and run
// stride), then the adds in the reference's order.  No branch between the loads: they are all in flight
// together (CNT * NT * 16 B per lane), which is what keeps HBM busy with ~28 waves per CU.
template <bool VEC, int NT, int CNT>
__device__ __forceinline__ void merge_batch(float4* acc, const float* const* base, int64_t rs, int64_t i, int64_t n, int run0) {
    float4 v[NT][CNT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CNT; ++j) v[t][j] = ld4<VEC>(base[t] + (int64_t)(run0 + j) * rs, i, n);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CNT; ++j) acc[t] = add4(acc[t], v[t][j]);
}

// PI:216-222 merge of the N runs of NT tensors at elements [i, i+4).
//   quirk: acc = x0; acc += x0; acc += x1 .. x_{N-2}; acc /= N      true mean: acc = x0; acc += x1 .. x_{N-1}; acc /= N
template <bool VEC, int NT, int BATCH>
__device__ __forceinline__ void merge_runs4(float4* acc, const float* const* base, int64_t rs, int64_t i, int64_t n, int n_runs,
                                            int quirk) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = ld4<VEC>(base[t], i, n);
    if (n_runs == 1) return;
    int r = 1, last = n_runs;          // runs [r, last) are still to be added
    if (quirk) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = add4(acc[t], acc[t]);
        last = n_runs - 1;
    }
    while (r + BATCH <= last) {
        merge_batch<VEC, NT, BATCH>(acc, base, rs, i, n, r);
        r += BATCH;
    }
    if (BATCH > 4 && r + 4 <= last) {
        merge_batch<VEC, NT, 4>(acc, base, rs, i, n, r);
        r += 4;
    }
    if (BATCH > 2 && r + 2 <= last) {
        merge_batch<VEC, NT, 2>(acc, base, rs, i, n, r);
        r += 2;
    }
    if (r + 1 <= last) {
        merge_batch<VEC, NT, 1>(acc, base, rs, i, n, r);
        r += 1;
    }
    if (BATCH <= 2 && r < last) merge_batch<VEC, NT, 1>(acc, base, rs, i, n, r);
    const float fn = (float)n_runs;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = div4(acc[t], fn);
}

// ---- box role: element-wise merge of delta (role 1) or reg_var (role 2) ---------------------------------
template <bool VEC, int BATCH>
__device__ __forceinline__ void box_role(const K1Params& P, const PodLevel& lv, int l, int role, int local_b, int HW) {
    const bool is_delta = role == 1;
    const int C = is_delta ? 4 : P.D;
    const float* src = is_delta ? lv.delta : lv.reg_var;
    float* dst = (is_delta ? P.mean_delta : P.mean_reg_var);
    if (dst == nullptr || P.n_runs == 1) return;
    dst += (int64_t)lv.anchor_base * C;
    const int64_t rs = is_delta ? lv.run_stride_delta : lv.run_stride_reg;
    const int64_t n = (int64_t)P.A * C * HW;
    const int64_t i = ((int64_t)local_b * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 acc[1];
    const float* base[1] = {src};
    merge_runs4<VEC, 1, BATCH>(acc, base, rs, i, n, P.n_runs, P.quirk);
    st4<VEC>(dst, i, n, acc[0]);
}

// ---- cls role: merged logit / log-variance of class k (= wave id) for 4 cells per lane, then the 4 class
// probabilities ------------------------------------------------------------------------------------------
template <bool VEC, int BATCH>
__device__ __forceinline__ float4 cls_role(const K1Params& P, const PodLevel& lv, int l, int a, int k, int hw0, int HW) {
    const int K = P.K;
    const bool has_var = P.has_cls_var != 0;
    const int64_t plane = (int64_t)(a * K + k) * HW;
    const int64_t n = plane + HW;   // bound for the scalar tail path
    const int64_t i = plane + hw0;
    float4 m[2];
    m[1] = float4{0.f, 0.f, 0.f, 0.f};
    const float* base[2] = {lv.cls, lv.cls_var};
    if (has_var) merge_runs4<VEC, 2, BATCH>(m, base, lv.run_stride_cls, i, n, P.n_runs, P.quirk);
    else merge_runs4<VEC, 1, BATCH>(m, base, lv.run_stride_cls, i, n, P.n_runs, P.quirk);
    if (P.n_runs > 1) {
        const int64_t off = (int64_t)lv.anchor_base * K;
        if (P.mean_cls) st4<VEC>(P.mean_cls + off, i, n, m[0]);
        if (has_var && P.mean_cls_var) st4<VEC>(P.mean_cls_var + off, i, n, m[1]);
    }
    const float lg[4] = {m[0].x, m[0].y, m[0].z, m[0].w};
    const float vr[4] = {m[1].x, m[1].y, m[1].z, m[1].w};
    float pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool live = hw0 + j < HW;
        pr[j] = live ? class_prob_cell(lg[j], vr[j], has_var, P.cls_samples, lv.eps_cls, (int64_t)HW * P.A, K, P.A, l, hw0 + j, a, k, P.seed)
                     : 0.0f;
    }
    return float4{pr[0], pr[1], pr[2], pr[3]};
}

template <int BATCH>
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
        if (role == 1 ? P.vec_delta[l] : P.vec_reg[l]) box_role<true, BATCH>(P, lv, l, role, local_b, HW);
        else box_role<false, 1>(P, lv, l, role, local_b, HW);
        return;
    }

    const int K = P.K;
    const int chunks = P.chunks[l];
    const int a = local_b / chunks;
    const int chunk = local_b - a * chunks;
    const int k = tid >> 6;             // wave id = class
    const int lane = tid & 63;
    const int hw0 = chunk * 256 + lane * 4;
    float4 prob = float4{0.f, 0.f, 0.f, 0.f};
    if (hw0 < HW) prob = P.vec_cls[l] ? cls_role<true, BATCH>(P, lv, l, a, k, hw0, HW) : cls_role<false, 1>(P, lv, l, a, k, hw0, HW);
    *reinterpret_cast<float4*>(&lds_probs[k * 256 + lane * 4]) = prob;
    __syncthreads();

    // ---- 256 threads: one anchor each; max/argmax over classes; candidate emission ------------------
    // (blockDim = 64*K may be smaller than 256 when K < 4: whole wavefronts walk the 256 anchors)
    for (int p = tid; p < 256; p += blockDim.x) {
        const int hw = chunk * 256 + p;
        float best = lds_probs[p];
#pragma unroll 1
        for (int kk = 1; kk < K; ++kk) {
            const float v = lds_probs[kk * 256 + p];
            best = (v > best) ? v : best;   // torch.max keeps the first maximum; argmax is re-derived in K2b
        }
        {
            const bool pass = (hw < HW) && (best > P.score_thresh);
            const unsigned long long m = __ballot(pass);
            if (m != 0ull) {
                const int total = __popcll(m);
                const int mylane = tid & 63;
                int base = 0;
                if (mylane == 0) base = atomicAdd(&P.cand_count[l], total);
                base = __shfl(base, 0, 64);
                const int at = base + __popcll(m & ((1ull << mylane) - 1ull));
                // (at < level size always holds when cand_count was zero on entry; the bound keeps a stale counter from
                //  writing into the next level's slots)
                if (pass && at < HW * P.A) P.cand_keys[(int64_t)lv.anchor_base + at] = make_key(best, hw * P.A + a);
            }
        }
    }
}

// ---- K1 prune mode: one flat streaming kernel ------------------------------------------------------------------
// Native RNG + variance head.  box_muller16() bounds every draw by |eps| < POD_EPS_MAX, so
//     mean_s sigmoid(logit + eps_s*sigma) <= sigmoid(logit + POD_EPS_MAX*sigma):
// an (anchor, class) with logit + POD_EPS_MAX*sigma <= logit(score_thresh) can never become a candidate.  The dense
// pass therefore draws nothing: it merges, stores, and sets one bit per anchor that MAY pass (exact superset);
// K1b samples those.  With no max-over-classes left there is no LDS, no barrier and no class-per-wave shape: every
// tensor is walked as a flat array, 256 threads x 16 B = 4 KiB contiguous per run per workgroup (the access
// pattern of a plain streaming merge), logit and log-variance planes side by side.
// OR over the 16 lanes of a DPP row (row_ror:8,4,2,1): every lane ends up with the row's OR, no LDS crossbar
__device__ __forceinline__ uint32_t row_or16(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);
    return v;
}

template <bool VEC, int BATCH>
__device__ __forceinline__ void prune_cls(const K1Params& P, const PodLevel& lv, int l, int local_b, int HW) {
    const int K = P.K;
    const int64_t n = (int64_t)P.A * K * HW;
    const int64_t i = ((int64_t)local_b * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 m[2];
    const float* base[2] = {lv.cls, lv.cls_var};
    merge_runs4<VEC, 2, BATCH>(m, base, lv.run_stride_cls, i, n, P.n_runs, P.quirk);
    if (P.n_runs > 1) {
        const int64_t off = (int64_t)lv.anchor_base * K;
        st4<VEC, true>(P.mean_cls + off, i, n, m[0]);
        st4<VEC, true>(P.mean_cls_var + off, i, n, m[1]);
    }
    const float lg[4] = {m[0].x, m[0].y, m[0].z, m[0].w};
    const float vr[4] = {m[1].x, m[1].y, m[1].z, m[1].w};
    uint64_t* bits = P.maybe_bits + P.word_begin[l];
    if (VEC) {   // H*W % 4 == 0: the four elements are consecutive cells of one plane
        const uint32_t plane = (uint32_t)i / (uint32_t)HW;          // n < 2^31 (checked on the host)
        const int hw = (int)((uint32_t)i - plane * (uint32_t)HW);
        unsigned nib = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (fmaf(POD_EPS_MAX, __builtin_amdgcn_exp2f(0.7213475204444817f * vr[j]), lg[j]) > P.skip_logit) nib |= 1u << j;
        // one bitmap word per (plane, 64 cells): the classes of an anchor do NOT share a word.  (They did in round 1: with
        // every anchor flagged, 7 classes x 8 words hammered each 64-byte line with 56 same-line atomics, and K1 took 48 us
        // instead of 23 -- the stores, not their being atomic: tools/exp_k1_worst.py.)  K1b ORs the K words of an anchor.
        unsigned long long* word = reinterpret_cast<unsigned long long*>(bits + plane * P.wpa[l] + (hw >> 6));
        if ((HW & 63) == 0) {
            // planes are whole bitmap words and a wavefront starts on a 256-element boundary: the 16 lanes of a DPP row
            // own exactly one word -> OR-reduce their nibbles with row rotations (VALU only); nobody else writes this word:
            // a plain store of the non-zero ones (K1b leaves every word zero again)
            const unsigned long long w = (unsigned long long)nib << (hw & 63);
            const uint32_t lo = row_or16((uint32_t)w), hi = row_or16((uint32_t)(w >> 32));
            if ((threadIdx.x & 15) == 0 && (lo | hi) != 0u) *word = ((unsigned long long)hi << 32) | lo;
        } else if (nib) {
            atomicOr(word, (unsigned long long)nib << (hw & 63));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i + j >= n) break;
            if (!(fmaf(POD_EPS_MAX, __builtin_amdgcn_exp2f(0.7213475204444817f * vr[j]), lg[j]) > P.skip_logit)) continue;
            const int plane = (int)((i + j) / HW);
            const int hw = (int)(i + j - (int64_t)plane * HW);
            atomicOr(reinterpret_cast<unsigned long long*>(bits + plane * P.wpa[l] + (hw >> 6)), 1ull << (hw & 63));
        }
    }
}

template <int BATCH>
__global__ void __launch_bounds__(256) k1_prune_stream(const K1Params P) {
    const int b = blockIdx.x;
    const int L = P.n_levels;
    int seg = 0;
#pragma unroll 1
    while (seg + 1 < 3 * L && b >= P.pseg_begin[seg + 1]) ++seg;
    const int role = seg / L;
    const int l = L - 1 - (seg - role * L);   // last (smallest) level first: its ragged maps take the scalar path, whose
                                              // longer chain of dependent loads then overlaps the bulk instead of trailing it
    const PodLevel& lv = P.lv[l];
    const int local_b = b - P.pseg_begin[seg];
    const int HW = lv.H * lv.W;
    if (role == 0) {
        if (P.vec_cls[l]) prune_cls<true, BATCH>(P, lv, l, local_b, HW);
        else prune_cls<false, 4>(P, lv, l, local_b, HW);
    } else {
        if (role == 1 ? P.vec_delta[l] : P.vec_reg[l]) box_role<true, BATCH>(P, lv, l, role, local_b, HW);
        else box_role<false, 4>(P, lv, l, role, local_b, HW);
    }
}

// ---- K1b score_maybe ---------------------------------------------------------------------------------------
// Sparse companion of K1's prune mode: draws the cls_samples normals and evaluates
// mean_s sigmoid(logit + eps_s*sigma) (PI:289-295) only for the anchors K1 flagged, reading the merged
// logits / log-variances K1 just wrote (or the single run when N == 1).  Persistent grid, static work split
// (wavefront w owns bitmap words w, w + nwaves, ...: no work-list atomics); a wavefront scores 64/KP flagged
// anchors at a time (KP = 8 or 16 lanes per anchor, lane = class), reduces max over the class lanes by
// butterfly and appends the keys of anchors above the threshold with one aggregated atomic.
struct K1bParams {
    PodLevel lv[POD_MAX_LEVELS];
    int32_t word_begin[POD_MAX_LEVELS + 1];   // level l = bitmap words [word_begin[l], word_begin[l+1]), K per (anchor shape, 64 cells)
    int32_t unit_begin[POD_MAX_LEVELS + 1];   // work units (anchor shape a, 64-cell block): level l = [unit_begin[l], unit_begin[l+1])
    int32_t wpa[POD_MAX_LEVELS];              // words per plane: bit (a, k, hw) = word (a*K + k)*wpa + hw/64, bit hw%64
    int32_t n_levels, n_runs, A, K, cls_samples;
    float score_thresh;
    uint64_t seed;
    const float* mean_cls;
    const float* mean_cls_var;
    uint64_t* maybe_bits;
    uint64_t* cand_keys;
    int32_t* cand_count;
    float* probs_dense;      // (R, K), level-concatenated anchor order: the K probabilities of every anchor emitted here, or null
};

template <int KP>
__global__ void __launch_bounds__(256) k1b_score_maybe(const K1bParams P) {
    constexpr int G = 64 / KP;                       // anchors per wavefront per round
    __shared__ uint64_t s_park[4 * 64];
    const int lane = threadIdx.x & 63;
    const int sub = lane / KP, k = lane % KP;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int L = P.n_levels, K = P.K, A = P.A;
    const int total_units = P.unit_begin[L];
    for (int w = wave; w < total_units; w += nwaves) {
        int l = 0;
        while (l + 1 < L && w >= P.unit_begin[l + 1]) ++l;
        const PodLevel& lv = P.lv[l];
        const int wl = w - P.unit_begin[l];
        const int a = wl / P.wpa[l];
        const int blk = wl - a * P.wpa[l];
        const int hw_base = blk * 64;
        // the K words of this (anchor shape, 64 cells): lane k < K fetches (and clears) class k's word, OR over the lanes
        unsigned long long mine = 0ull;
        uint64_t* wk = P.maybe_bits + P.word_begin[l] + (int64_t)(a * K + (lane < K ? lane : 0)) * P.wpa[l] + blk;
        if (lane < K) mine = *wk;
        if (mine != 0ull) *wk = 0ull;                // leave the bitmap cleared for the next image
        uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
#pragma unroll
        for (int o = 1; o < POD_MAX_CLASSES; o <<= 1) {
            lo |= __shfl_xor(lo, o, 64);
            hi |= __shfl_xor(hi, o, 64);
        }
        unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) |
                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo);      // wave-uniform
        if (m == 0ull) continue;
        const int64_t HW = (int64_t)lv.H * lv.W;
        const float* src = P.n_runs > 1 ? P.mean_cls + (int64_t)lv.anchor_base * K : lv.cls;
        const float* srcv = P.n_runs > 1 ? P.mean_cls_var + (int64_t)lv.anchor_base * K : lv.cls_var;
        // keys of this word's anchors above the threshold are parked in LDS and appended with ONE atomic per word
        // (the per-round atomics of an all-candidates image serialise on 5 counters: 283 us -> see DESIGN.md)
        uint64_t* park = s_park + (threadIdx.x >> 6) * 64;
        int parked = 0;
        while (m != 0ull) {
            // hand the next G set bits to the G lane groups
            int bit = -1;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (m != 0ull) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    if (sub == g) bit = b;
                }
            }
            const bool valid = bit >= 0;
            const int hw = hw_base + bit;
            float p = 0.0f;
            if (valid && k < K) {
                const int64_t e = (int64_t)(a * K + k) * HW + hw;
                p = class_prob_cell(src[e], srcv[e], true, P.cls_samples, nullptr, HW * A, K, A, l, hw, a, k, P.seed);
            }
            float best = p;
#pragma unroll
            for (int o = KP >> 1; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
            if (P.probs_dense && valid && k < K && best > P.score_thresh)      // the gather kernel reuses them (same function, same inputs)
                P.probs_dense[((int64_t)lv.anchor_base + (int64_t)hw * A + a) * K + k] = p;
            const bool emit = valid && k == 0 && best > P.score_thresh;
            const unsigned long long em = __ballot(emit);
            if (emit) park[parked + __popcll(em & ((1ull << lane) - 1ull))] = make_key(best, hw * A + a);
            parked += __popcll(em);
        }
        if (parked > 0) {
            int pos = 0;
            if (lane == 0) pos = atomicAdd(&P.cand_count[l], parked);
            pos = __shfl(pos, 0, 64);
            if (lane < parked && (int64_t)pos + lane < HW * A) P.cand_keys[(int64_t)lv.anchor_base + pos + lane] = park[lane];   // same wave wrote park[]
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
                                  uint64_t* maybe_bits, pod_stream_t stream) {
    if (!cfg || !levels || !cand_keys || !cand_count) return POD_E_INVALID;
    if (maybe_bits) {   // prune mode: native RNG + variance head only, and K1b needs the merged planes
        if (!cfg->has_cls_var) return POD_E_INVALID;
        if (cfg->n_runs > 1 && (!mean_cls || !mean_cls_var)) return POD_E_INVALID;
        for (int l = 0; l < cfg->n_levels && l < POD_MAX_LEVELS; ++l)
            if (levels[l].eps_cls) return POD_E_INVALID;
    }
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
        if ((int64_t)A * (K > 4 + D ? K : 4 + D) * HW >= (int64_t)1 << 31) return POD_E_INVALID;   // 32-bit plane arithmetic
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
    {
        const double t = (double)cfg->score_thresh;
        P.skip_logit = (t > 0.0 && t < 1.0) ? (float)(log(t / (1.0 - t)) - 0.02) : -INFINITY;   // margin covers the fast-math error
    }
    P.mean_cls = mean_cls; P.mean_cls_var = mean_cls_var; P.mean_delta = mean_delta; P.mean_reg_var = mean_reg_var;
    P.cand_keys = cand_keys; P.cand_count = cand_count; P.maybe_bits = maybe_bits;
    if (maybe_bits) {
        // prune mode: flat streaming kernel, 256-thread workgroups
        int32_t wb = 0, pb = 0, q = 0;
        for (int l = 0; l < L; ++l) {
            P.word_begin[l] = wb;
            P.wpa[l] = (int32_t)(((int64_t)levels[l].H * levels[l].W + 63) / 64);
            wb += A * K * P.wpa[l];
        }
        P.word_begin[L] = wb;
        for (int role = 0; role <= 2; ++role)
            for (int l = L - 1; l >= 0; --l) {   // segment order inside a role: last level first (see k1_prune_stream)
                P.pseg_begin[q++] = pb;
                const int C = role == 0 ? K : (role == 1 ? 4 : D);
                const bool active = role == 0 || (N > 1 && C > 0 && (role == 1 ? mean_delta != nullptr : mean_reg_var != nullptr));
                if (active) pb += (int32_t)((((int64_t)A * C * levels[l].H * levels[l].W + 3) / 4 + 255) / 256);
            }
        P.pseg_begin[q] = pb;
        // 4 independent 16-B loads per tensor per lane: measured 34.2 us vs 34.1 (2) and 34.6 (8) per launch
        // (also measured this round, rejected: a persistent grid of 512 / 1024 workgroups looping over the chunks, 23.1 / 22.2 us
        //  against 21.9; write-through (sc0 sc1) stores of the merged planes instead of non-temporal ones, 22.5 us; LDS-DMA
        //  loads (global_load_lds_dwordx4 nt into a per-wavefront slab, then ds_read_b128): 22.9 us here although the bare
        //  2 x 9-stream merge of tools/membw.hip gains 1 us with them)
        hipLaunchKernelGGL(pod::k1_prune_stream<4>, dim3(pb), dim3(256), 0, (hipStream_t)stream, P);
        POD_CHECK_LAUNCH();
        return POD_OK;
    }
    const size_t lds = sizeof(float) * 256 * K;
    hipLaunchKernelGGL(pod::k1_mc_merge_score<4>, dim3(nb), dim3(threads), lds, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int64_t pod_maybe_words(const PodConfig* cfg, const PodLevel* levels) {
    if (!cfg || !levels || cfg->n_levels < 1 || cfg->n_levels > POD_MAX_LEVELS) return POD_E_INVALID;
    int64_t w = 0;
    for (int l = 0; l < cfg->n_levels; ++l)
        w += (int64_t)cfg->num_anchors * cfg->num_classes * (((int64_t)levels[l].H * levels[l].W + 63) / 64);
    return w;
}

extern "C" int pod_score_maybe(const PodConfig* cfg, const PodLevel* levels, const float* mean_cls, const float* mean_cls_var,
                               uint64_t* maybe_bits, uint64_t* cand_keys, int32_t* cand_count, float* probs_dense,
                               pod_stream_t stream) {
    if (!cfg || !levels || !maybe_bits || !cand_keys || !cand_count) return POD_E_INVALID;
    const int L = cfg->n_levels, K = cfg->num_classes;
    if (L < 1 || L > POD_MAX_LEVELS || K < 1 || K > POD_MAX_CLASSES || !cfg->has_cls_var) return POD_E_INVALID;
    if (cfg->n_runs > 1 && (!mean_cls || !mean_cls_var)) return POD_E_INVALID;
    if (cfg->cls_samples < 1 || cfg->cls_samples > POD_MAX_CLS_SAMPLES) return POD_E_INVALID;
    pod::K1bParams P;
    int32_t wb = 0, ub = 0;
    for (int l = 0; l < L; ++l) {
        if (!levels[l].cls || !levels[l].cls_var || levels[l].eps_cls) return POD_E_INVALID;
        P.lv[l] = levels[l];
        P.wpa[l] = (int32_t)(((int64_t)levels[l].H * levels[l].W + 63) / 64);
        P.word_begin[l] = wb;
        P.unit_begin[l] = ub;
        wb += cfg->num_anchors * K * P.wpa[l];
        ub += cfg->num_anchors * P.wpa[l];
    }
    P.word_begin[L] = wb;
    P.unit_begin[L] = ub;
    P.n_levels = L; P.n_runs = cfg->n_runs; P.A = cfg->num_anchors; P.K = K; P.cls_samples = cfg->cls_samples;
    P.score_thresh = cfg->score_thresh; P.seed = cfg->philox_seed; P.mean_cls = mean_cls; P.mean_cls_var = mean_cls_var;
    P.maybe_bits = maybe_bits; P.cand_keys = cand_keys; P.cand_count = cand_count; P.probs_dense = probs_dense;
    const int blocks = (ub + 3) / 4 < 1024 ? (ub + 3) / 4 : 1024;   // one wavefront per work unit up to a persistent 4096
    const dim3 grid(blocks), block(256);
    if (K <= 8) hipLaunchKernelGGL(pod::k1b_score_maybe<8>, grid, block, 0, (hipStream_t)stream, P);
    else hipLaunchKernelGGL(pod::k1b_score_maybe<16>, grid, block, 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
