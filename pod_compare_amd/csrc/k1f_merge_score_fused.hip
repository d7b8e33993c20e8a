// K1f merge_score_fused -- merge AND score in ONE streaming launch (the production form of SURVEY 8 rows a3 + a4 since round 4).
//
// Replaces (reference, /root/reference/src/probabilistic_inference/probabilistic_inference.py):
//   :211-270  merge of the N MC-dropout runs / ensemble members of box_cls and box_cls_var (incl. the quirk of :216-222)
//   :289-297  classification sampling  mean_s sigmoid(logit + eps_s * sqrt(exp(var)))
//   :301,:304 max over classes and the score-threshold test (top-k itself is K2)
// i.e. what pod_mc_merge_score (prune mode) + pod_score_maybe do in two launches with a bitmap between them.
//
// Why the first fusion (round 3) lost and what is different here.  There a lane owned ONE class of its cells: an anchor that might
// pass had to be claimed, its other classes re-loaded from HBM (18 scattered loads per class lane) and scored by the wavefront that
// found it -- and flagged anchors are spatially clustered, so a few wavefronts carried all the scoring after everybody else had
// finished.  Here a LANE OWNS ALL K CLASSES OF ITS CELL(S): 2K accumulators stay in registers through the run loop
// (the run loop is outside, every load instruction of a wavefront still reads 1 KiB of one plane of one run: the streaming pattern
// of the flat kernel), the prune test runs on the merged values where they are, and nothing is ever re-loaded.  The cells that may
// pass are parked in LDS with their 2K merged values, and the WHOLE WORKGROUP -- whose wavefronts stream chunks that lie far apart
// in the level, so that a cluster of objects is spread over many workgroups -- scores the parked cells 8 lanes per cell (lane =
// class), exactly as K1b does: same function (class_prob_cell), same butterfly, same keys, same stored probabilities.  No bitmap,
// no second launch, no claim atomics; one aggregated global atomic per level and workgroup.
//
// Geometry at BASELINE size (R = 193 374, A = 9, K = 7, N = 10): ONE cell per lane -- 3 060 wave-units of 64 cells x one anchor shape,
// every lane with 2 runs x 14 planes = 28 independent non-temporal loads in flight, 12 wavefronts per CU.  (Four cells per lane --
// 16-byte loads, 765 wavefronts, less than one per SIMD -- walk their 9 runs as a chain of dependent round trips: 36 us; what hides the
// HBM latency is wavefronts in flight.  Measured on one box, planted image: K1 + K1b 24.7 + 11.9 us; this kernel 25.9 us, of which the
// streaming part alone 21 us -- profiles/r04_experiments.md.)  No MFMA: element-wise + reductions.
#include <mutex>

#include "pod_device.h"
#include "pod_experiments.h"

// launch geometry: POD_K1F_WAVES / _WPE / _BATCH / _NT / _CELLS (pod_experiments.h; profiles/r05_k1f_variants.txt: the defaults are the fastest of twelve)

namespace pod {

struct K1fParams {
    PodLevel lv[POD_MAX_LEVELS];
    int32_t unit_begin[POD_MAX_LEVELS + 1];   // wave-units (anchor shape a, 256-cell chunk): level l = [unit_begin[l], unit_begin[l+1])
    int32_t chunks[POD_MAX_LEVELS];           // 256-cell chunks per anchor shape
    uint8_t vec[POD_MAX_LEVELS];              // 16-byte path usable (alignment, H*W % 4 == 0)
    int32_t n_levels, n_runs, A, K, has_cls_var, quirk, cls_samples;
    float score_thresh, skip_logit;
    uint64_t seed;
    float* mean_cls;         // merged planes, level-concatenated (level l at anchor_base_l * K), or null: not stored
    float* mean_cls_var;
    uint64_t* cand_keys;
    int32_t* cand_count;
    float* probs_dense;      // (R, K): the K probabilities of every anchor emitted, or null
};

// CPL consecutive cells of one plane per lane: 16-, 8- or 4-byte loads (a wavefront instruction reads 64 * CPL contiguous floats of one
// plane of one run).  Fewer cells per lane = more wavefronts with fewer registers each: what hides the HBM latency here is wavefronts
// in flight, as in the flat kernel -- with 4 cells per lane the launch is 765 wavefronts, less than one per SIMD, and every one of them
// walks its 9 runs as a chain of dependent round trips (measured: 36 us against 25 for K1 alone).
template <int CPL>
struct K1fVals {
    float v[CPL];
};
typedef float k1f_f32x4 __attribute__((ext_vector_type(4)));
typedef float k1f_f32x2 __attribute__((ext_vector_type(2)));

template <bool VEC, int CPL>
__device__ __forceinline__ K1fVals<CPL> k1f_ld(const float* p, int64_t i, int hw0, int HW) {
    K1fVals<CPL> r;
    if (VEC && CPL == 4) {
        const k1f_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const k1f_f32x4*>(p + i));
#pragma unroll
        for (int j = 0; j < CPL; ++j) r.v[j] = v[j];
    } else if (VEC && CPL == 2) {
        const k1f_f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const k1f_f32x2*>(p + i));
#pragma unroll
        for (int j = 0; j < CPL; ++j) r.v[j] = v[j];
    } else if (VEC) {
        r.v[0] = POD_K1F_NT ? __builtin_nontemporal_load(p + i) : p[i];
    } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j) r.v[j] = hw0 + j < HW ? p[i + j] : 0.0f;
    }
    return r;
}
template <bool VEC, int CPL>
__device__ __forceinline__ void k1f_st(float* p, int64_t i, int hw0, int HW, const K1fVals<CPL>& a) {
    if (VEC && CPL == 4) {
        const k1f_f32x4 w = {a.v[0], a.v[1 % CPL], a.v[2 % CPL], a.v[3 % CPL]};
        __builtin_nontemporal_store(w, reinterpret_cast<k1f_f32x4*>(p + i));
    } else if (VEC && CPL == 2) {
        const k1f_f32x2 w = {a.v[0], a.v[1 % CPL]};
        __builtin_nontemporal_store(w, reinterpret_cast<k1f_f32x2*>(p + i));
    } else if (VEC) {
        __builtin_nontemporal_store(a.v[0], p + i);
    } else {
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            if (hw0 + j < HW) p[i + j] = a.v[j];
    }
}
template <int CPL>
__device__ __forceinline__ void k1f_add(K1fVals<CPL>& a, const K1fVals<CPL>& b) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) a.v[j] = a.v[j] + b.v[j];
}

// CNT runs x (1 or 2) tensors x K planes of independent loads, then the adds in the reference's order (run after run)
template <bool VEC, bool VAR, int KP, int CPL, int CNT>
__device__ __forceinline__ void k1f_batch(K1fVals<CPL> (&mc)[KP], K1fVals<CPL> (&mv)[KP], const PodLevel& lv, int K, int64_t i0, int64_t plane_stride, int hw0, int HW,
                                          int run0) {
    K1fVals<CPL> c[CNT][KP], v[CNT][KP];
#pragma unroll
    for (int j = 0; j < CNT; ++j)
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) {
                c[j][k] = k1f_ld<VEC, CPL>(lv.cls + (int64_t)(run0 + j) * lv.run_stride_cls, i0 + k * plane_stride, hw0, HW);
                if (VAR) v[j][k] = k1f_ld<VEC, CPL>(lv.cls_var + (int64_t)(run0 + j) * lv.run_stride_cls, i0 + k * plane_stride, hw0, HW);
            }
#pragma unroll
    for (int j = 0; j < CNT; ++j)
#pragma unroll
        for (int k = 0; k < KP; ++k)
            if (k < K) {
                k1f_add(mc[k], c[j][k]);
                if (VAR) k1f_add(mv[k], v[j][k]);
            }
}

// PI:216-222 for the 2K planes of one (anchor shape, CPL cells):  quirk: acc = x0; acc += x0; acc += x1 .. x_{N-2}; acc /= N
//                                                                  true mean: acc = x0; acc += x1 .. x_{N-1}; acc /= N
template <bool VEC, bool VAR, int KP, int CPL>
__device__ __forceinline__ void k1f_merge(K1fVals<CPL> (&mc)[KP], K1fVals<CPL> (&mv)[KP], const K1fParams& P, const PodLevel& lv, int64_t i0, int64_t plane_stride,
                                          int hw0, int HW) {
    const int K = P.K;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) mc[k].v[j] = mv[k].v[j] = 0.0f;
        if (k < K) {
            mc[k] = k1f_ld<VEC, CPL>(lv.cls, i0 + k * plane_stride, hw0, HW);
            if (VAR) mv[k] = k1f_ld<VEC, CPL>(lv.cls_var, i0 + k * plane_stride, hw0, HW);
        }
    }
    if (P.n_runs == 1) return;
    int r = 1, last = P.n_runs;
    if (P.quirk) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            k1f_add(mc[k], mc[k]);
            k1f_add(mv[k], mv[k]);
        }
        last = P.n_runs - 1;
    }
    constexpr int B = CPL == 4 ? 2 : POD_K1F_BATCH;  // runs per batch: 2K * B * CPL registers of loads in flight per lane
    while (r + B <= last) {
        k1f_batch<VEC, VAR, KP, CPL, B>(mc, mv, lv, K, i0, plane_stride, hw0, HW, r);
        r += B;
    }
    if (B > 2 && r + 2 <= last) {
        k1f_batch<VEC, VAR, KP, CPL, 2>(mc, mv, lv, K, i0, plane_stride, hw0, HW, r);
        r += 2;
    }
    if (r < last) k1f_batch<VEC, VAR, KP, CPL, 1>(mc, mv, lv, K, i0, plane_stride, hw0, HW, r);
    const float fn = (float)P.n_runs;
#pragma unroll
    for (int k = 0; k < KP; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            mc[k].v[j] = __fdiv_rn(mc[k].v[j], fn);
            mv[k].v[j] = __fdiv_rn(mv[k].v[j], fn);
        }
}

template <int KP, int WAVES, int CPL>
struct K1fLds {
    static constexpr int CAP = 64 * CPL * WAVES;     // every cell of the workgroup may be parked
    float val[CAP][2 * KP];                          // merged logits, merged log-variances of a parked cell
    int32_t meta[CAP][2];                            // level << 8 | a, hw
    uint64_t key[CAP];                               // keys above the threshold ...
    int32_t key_info[CAP];                           // ... level << 16 | rank inside (workgroup, level)
    int32_t n_parked, n_keys;
    int32_t lvl_count[POD_MAX_LEVELS], lvl_base[POD_MAX_LEVELS];
};

// Occupancy target: POD_K1F_WPE wavefronts per SIMD for K <= 8 classes (2K accumulators + 2 runs x 2K loads in flight fit 170 registers);
// K > 8 (KP = 16) needs twice the registers per lane -- at 3 per SIMD it spilled 153 VGPRs to scratch (round 4) -- and runs 2 per SIMD.
template <int KP, int WAVES, int CPL, bool VAR>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(KP <= 8 ? POD_K1F_WPE : 2, KP <= 8 ? POD_K1F_WPE : 2))) k1f_merge_score(const K1fParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char k1f_lds_raw[];
    K1fLds<KP, WAVES, CPL>& S = *reinterpret_cast<K1fLds<KP, WAVES, CPL>*>(k1f_lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int L = P.n_levels, K = P.K, A = P.A;
    if (tid == 0) {
        S.n_parked = 0;
        S.n_keys = 0;
    }
    if (tid < POD_MAX_LEVELS) S.lvl_count[tid] = 0;
    __syncthreads();

    // ---- stream: wavefront w of workgroup b takes wave-unit b + w * gridDim.x (units of one workgroup lie far apart) -----------------
    // (rotating the quarters against each other so that the four units of a workgroup lie in different parts of the IMAGE as well:
    //  measured, no difference -- the 4 us this kernel takes beyond its streaming part are the barrier, one scoring round and the
    //  emission, not a cluster that piled up in one workgroup)
    const int u = POD_K1F_ADJ ? (int)blockIdx.x * WAVES + wave : (int)blockIdx.x + wave * (int)gridDim.x;
    if (u < P.unit_begin[L]) {
        int l = 0;
        while (l + 1 < L && u >= P.unit_begin[l + 1]) ++l;
        const PodLevel& lv = P.lv[l];
        const int local = u - P.unit_begin[l];
        const int a = local / P.chunks[l], chunk = local - a * P.chunks[l];          // chunk = 64 * CPL cells
        const int HW = lv.H * lv.W;
        const int hw0 = (chunk * 64 + lane) * CPL;
        if (hw0 < HW) {
            const int64_t i0 = (int64_t)a * K * HW + hw0;          // element of plane (a, k = 0); plane (a, k) is k * HW further
            K1fVals<CPL> mc[KP], mv[KP];
            if (P.vec[l]) k1f_merge<true, VAR, KP, CPL>(mc, mv, P, lv, i0, HW, hw0, HW);
            else k1f_merge<false, VAR, KP, CPL>(mc, mv, P, lv, i0, HW, hw0, HW);
            if (P.n_runs > 1 && P.mean_cls) {
                const int64_t off = (int64_t)lv.anchor_base * K + i0;
#pragma unroll
                for (int k = 0; k < KP; ++k)
                    if (k < K) {
                        if (P.vec[l]) {
                            k1f_st<true, CPL>(P.mean_cls, off + (int64_t)k * HW, hw0, HW, mc[k]);
                            if (VAR && P.mean_cls_var) k1f_st<true, CPL>(P.mean_cls_var, off + (int64_t)k * HW, hw0, HW, mv[k]);
                        } else {
                            k1f_st<false, CPL>(P.mean_cls, off + (int64_t)k * HW, hw0, HW, mc[k]);
                            if (VAR && P.mean_cls_var) k1f_st<false, CPL>(P.mean_cls_var, off + (int64_t)k * HW, hw0, HW, mv[k]);
                        }
                    }
            }
            // prune test (exact superset of the candidates: |eps| < POD_EPS_MAX, see k1_mc_merge_score.hip), cell by cell
            unsigned flags = 0;
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (k < K) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const float top = VAR ? fmaf(POD_EPS_MAX, __builtin_amdgcn_exp2f(0.7213475204444817f * mv[k].v[j]), mc[k].v[j]) : mc[k].v[j];
                        if (top > P.skip_logit) flags |= 1u << j;
                    }
                }
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                if (hw0 + j >= HW) flags &= ~(1u << j);
            if (POD_K1F_NOSCORE) flags = 0;                  // (experiment builds: the streaming part alone)
            // park the flagged cells with their 2K merged values
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const bool f = (flags >> j) & 1u;
                const unsigned long long m = __ballot(f);
                if (m == 0ull) continue;
                int base = 0;
                if (lane == (int)(__ffsll((long long)m) - 1)) base = atomicAdd(&S.n_parked, __popcll(m));
                base = __shfl(base, __ffsll((long long)m) - 1, 64);
                if (f) {
                    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                    S.meta[slot][0] = (l << 8) | a;
                    S.meta[slot][1] = hw0 + j;
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        S.val[slot][k] = mc[k].v[j];
                        S.val[slot][KP + k] = mv[k].v[j];
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- score the parked cells: KP lanes per cell (lane = class), exactly K1b's evaluation ------------------------------------------
    const int total = S.n_parked;
    if (total == 0) return;
    const int k = tid % KP;
    for (int s0 = 0; s0 < total; s0 += (64 * WAVES) / KP) {
        const int s = s0 + tid / KP;
        const bool valid = s < total;
        int l = 0, a = 0, hw = 0;
        float p = 0.0f;
        if (valid) {
            l = S.meta[s][0] >> 8;
            a = S.meta[s][0] & 0xFF;
            hw = S.meta[s][1];
            if (k < K) {
                const int64_t HW = (int64_t)P.lv[l].H * P.lv[l].W;
                p = class_prob_cell(S.val[s][k], S.val[s][KP + k], VAR, P.cls_samples, nullptr, HW * A, K, A, l, hw, a, k, P.seed);
            }
        }
        float best = p;
#pragma unroll
        for (int o = KP >> 1; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
        const bool pass = valid && best > P.score_thresh;
        if (P.probs_dense && pass && k < K) P.probs_dense[((int64_t)P.lv[l].anchor_base + (int64_t)hw * A + a) * K + k] = p;
        if (pass && k == 0) {
            const int rank = atomicAdd(&S.lvl_count[l], 1);
            const int at = atomicAdd(&S.n_keys, 1);
            S.key[at] = make_key(best, hw * A + a);
            S.key_info[at] = (l << 16) | rank;
        }
    }
    __syncthreads();
    if (tid < L && S.lvl_count[tid] > 0) S.lvl_base[tid] = atomicAdd(&P.cand_count[tid], S.lvl_count[tid]);
    __syncthreads();
    for (int i = tid; i < S.n_keys; i += 64 * WAVES) {
        const int l = S.key_info[i] >> 16, at = S.lvl_base[l] + (S.key_info[i] & 0xFFFF);
        // (at < level size always holds when cand_count was zero on entry; the bound keeps a stale counter from writing into the next level's slots)
        if (at < P.lv[l].H * P.lv[l].W * A) P.cand_keys[(int64_t)P.lv[l].anchor_base + at] = S.key[i];
    }
}

}  // namespace pod


template <int KP, int WAVES, int CPL, bool VAR>
static int k1f_launch(const pod::K1fParams& P, int units, hipStream_t stream) {
    constexpr size_t lds = sizeof(pod::K1fLds<KP, WAVES, CPL>);
    static std::once_flag once[64];
    static hipError_t attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return POD_E_LAUNCH;
    std::call_once(once[dev], [dev] {
        attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(pod::k1f_merge_score<KP, WAVES, CPL, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr[dev] != hipSuccess) return POD_E_LAUNCH;
    const int blocks = (units + WAVES - 1) / WAVES;
    hipLaunchKernelGGL((pod::k1f_merge_score<KP, WAVES, CPL, VAR>), dim3(blocks), dim3(64 * WAVES), lds, stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_merge_score_fused(const PodConfig* cfg, const PodLevel* levels, float* mean_cls, float* mean_cls_var, uint64_t* cand_keys,
                                     int32_t* cand_count, float* probs_dense, pod_stream_t stream) {
    if (!cfg || !levels || !cand_keys || !cand_count) return POD_E_INVALID;
    const int L = cfg->n_levels, K = cfg->num_classes, A = cfg->num_anchors, N = cfg->n_runs;
    if (L < 1 || L > POD_MAX_LEVELS || K < 1 || K > POD_MAX_CLASSES || A < 1 || A > 255 || N < 1 || N > POD_MAX_RUNS) return POD_E_INVALID;
    if (cfg->has_cls_var && (cfg->cls_samples < 1 || cfg->cls_samples > POD_MAX_CLS_SAMPLES)) return POD_E_INVALID;
    if ((mean_cls == nullptr) != (mean_cls_var == nullptr) && cfg->has_cls_var) return POD_E_INVALID;
    constexpr int CPL = POD_K1F_CELLS;
    pod::K1fParams P;
    int32_t ub = 0;
    for (int l = 0; l < L; ++l) {
        const PodLevel& lv = levels[l];
        if (!lv.cls || lv.H < 1 || lv.W < 1 || lv.eps_cls) return POD_E_INVALID;      // native draws only: the prune bound needs |eps| < POD_EPS_MAX
        if (cfg->has_cls_var && !lv.cls_var) return POD_E_INVALID;
        const int64_t HW = (int64_t)lv.H * lv.W;
        if ((int64_t)A * K * HW >= (int64_t)1 << 31) return POD_E_INVALID;
        P.lv[l] = lv;
        P.chunks[l] = (int32_t)((HW + 64 * CPL - 1) / (64 * CPL));
        P.unit_begin[l] = ub;
        ub += A * P.chunks[l];
        auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(4 * CPL - 1)) == 0; };
        P.vec[l] = (HW % CPL == 0) && al(lv.cls) && (lv.run_stride_cls % CPL == 0) && (!cfg->has_cls_var || al(lv.cls_var)) &&
                   ((int64_t)lv.anchor_base * K % CPL == 0) && al(mean_cls) && al(mean_cls_var);
    }
    P.unit_begin[L] = ub;
    P.n_levels = L; P.n_runs = N; P.A = A; P.K = K; P.has_cls_var = cfg->has_cls_var; P.quirk = cfg->merge_quirk; P.cls_samples = cfg->cls_samples;
    P.score_thresh = cfg->score_thresh; P.seed = cfg->philox_seed;
    {
        const double t = (double)cfg->score_thresh;
        P.skip_logit = (t > 0.0 && t < 1.0) ? (float)(log(t / (1.0 - t)) - 0.02) : -INFINITY;   // margin covers the fast-math error
    }
    P.mean_cls = mean_cls; P.mean_cls_var = mean_cls_var; P.cand_keys = cand_keys; P.cand_count = cand_count; P.probs_dense = probs_dense;
    const hipStream_t st = (hipStream_t)stream;
    if (cfg->has_cls_var) return K <= 8 ? k1f_launch<8, POD_K1F_WAVES, CPL, true>(P, ub, st) : k1f_launch<16, POD_K1F_WAVES, CPL, true>(P, ub, st);
    return K <= 8 ? k1f_launch<8, POD_K1F_WAVES, CPL, false>(P, ub, st) : k1f_launch<16, POD_K1F_WAVES, CPL, false>(P, ub, st);
}
