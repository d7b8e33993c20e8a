// K5 bayes_fuse, K6 anchor_stats_merge, K7 finalize, reg_nll.
//
// Replaces:
//   K5  probabilistic_inference.py:562-636 (post_processing_bayes_od cluster loop, one host
//       round-trip per cluster at :591-601) + inference_utils.py:292-334
//       (bounding_box_bayesian_inference: numpy fp32 LAPACK inv/det).
//   K6  inference_utils.py:91-154 (general_anchor_statistics_postprocessing cluster loop).
//   K7  inference_utils.py:42-53 (keep-gather of the standard-NMS path) and :374-425
//       (probabilistic_detector_postprocess), plus the XYWH records of :428-502.
//   NLL core/evaluation_tools/scoring_rules.py:68-74.
//
// One 256-thread workgroup per kept cluster centre.  Only the <= 100 needed rows of the IoU
// matrix are evaluated, members are streamed once per pass, every lane inverts its
// members' 4x4 covariances in fp64 registers, and the Gaussian moments / precisions are
// combined with wavefront butterflies plus one LDS hop across the 4 waves.
#include "pod_device.h"

namespace pod {

// ---- 4x4 helpers (fp64 registers) ---------------------------------------------------------------
struct M4 {
    double a[16];
};

__device__ __forceinline__ double inv4(const M4& m, M4& o) {
    const double* a = m.a;
    const double s0 = a[0] * a[5] - a[4] * a[1], s1 = a[0] * a[6] - a[4] * a[2], s2 = a[0] * a[7] - a[4] * a[3];
    const double s3 = a[1] * a[6] - a[5] * a[2], s4 = a[1] * a[7] - a[5] * a[3], s5 = a[2] * a[7] - a[6] * a[3];
    const double c5 = a[10] * a[15] - a[14] * a[11], c4 = a[9] * a[15] - a[13] * a[11], c3 = a[9] * a[14] - a[13] * a[10];
    const double c2 = a[8] * a[15] - a[12] * a[11], c1 = a[8] * a[14] - a[12] * a[10], c0 = a[8] * a[13] - a[12] * a[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    o.a[0] = (a[5] * c5 - a[6] * c4 + a[7] * c3) * id;
    o.a[1] = (-a[1] * c5 + a[2] * c4 - a[3] * c3) * id;
    o.a[2] = (a[13] * s5 - a[14] * s4 + a[15] * s3) * id;
    o.a[3] = (-a[9] * s5 + a[10] * s4 - a[11] * s3) * id;
    o.a[4] = (-a[4] * c5 + a[6] * c2 - a[7] * c1) * id;
    o.a[5] = (a[0] * c5 - a[2] * c2 + a[3] * c1) * id;
    o.a[6] = (-a[12] * s5 + a[14] * s2 - a[15] * s1) * id;
    o.a[7] = (a[8] * s5 - a[10] * s2 + a[11] * s1) * id;
    o.a[8] = (a[4] * c4 - a[5] * c2 + a[7] * c0) * id;
    o.a[9] = (-a[0] * c4 + a[1] * c2 - a[3] * c0) * id;
    o.a[10] = (a[12] * s4 - a[13] * s2 + a[15] * s0) * id;
    o.a[11] = (-a[8] * s4 + a[9] * s2 - a[11] * s0) * id;
    o.a[12] = (-a[4] * c3 + a[5] * c1 - a[6] * c0) * id;
    o.a[13] = (a[0] * c3 - a[1] * c1 + a[2] * c0) * id;
    o.a[14] = (-a[12] * s3 + a[13] * s1 - a[14] * s0) * id;
    o.a[15] = (a[8] * s3 - a[9] * s1 + a[10] * s0) * id;
    return det;
}

__device__ __forceinline__ double det4(const M4& m) {
    const double* a = m.a;
    const double s0 = a[0] * a[5] - a[4] * a[1], s1 = a[0] * a[6] - a[4] * a[2], s2 = a[0] * a[7] - a[4] * a[3];
    const double s3 = a[1] * a[6] - a[5] * a[2], s4 = a[1] * a[7] - a[5] * a[3], s5 = a[2] * a[7] - a[6] * a[3];
    const double c5 = a[10] * a[15] - a[14] * a[11], c4 = a[9] * a[15] - a[13] * a[11], c3 = a[9] * a[14] - a[13] * a[10];
    const double c2 = a[8] * a[15] - a[12] * a[11], c1 = a[8] * a[14] - a[12] * a[10], c0 = a[8] * a[13] - a[12] * a[9];
    return s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
}

__device__ __forceinline__ void load_m4(const float* p, M4& m) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + q * 4);
        m.a[q * 4 + 0] = v.x; m.a[q * 4 + 1] = v.y; m.a[q * 4 + 2] = v.z; m.a[q * 4 + 3] = v.w;
    }
}

// Block-wide sum of NV doubles per thread (256 threads = 4 waves); result valid in every thread.
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* lds /* [4][NV] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();   // protect lds from the previous use
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NV; ++q) lds[wave * NV + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q] = (lds[q] + lds[NV + q]) + (lds[2 * NV + q] + lds[3 * NV + q]);
}

__device__ __forceinline__ int argmax_probs(const float* p, int K) {
    float best = p[0];
    int bk = 0;
    for (int k = 1; k < K; ++k)
        if (p[k] > best) {
            best = p[k];
            bk = k;
        }
    return bk;
}

struct K7Params {
    const int32_t* keep;   // may be null (identity)
    const int32_t* n_rows;
    const float* boxes;
    const float* cov;      // may be null -> zeros (IU:52-53)
    const float* scores;
    const int32_t* classes;
    const float* probs;
    int32_t K, max_det;
    float sx, sy, out_h, out_w;
    float* det_boxes;
    float* det_cov;
    float* det_scores;
    int32_t* det_classes;
    float* det_probs;
    float* records;
    int32_t* n_det;
};

// K7 body: thread t < POD_MAX_DETECTIONS owns input row t; every thread of the workgroup must call it (one barrier).
// s_flag: POD_MAX_DETECTIONS ints of LDS.
__device__ __forceinline__ void finalize_rows(const K7Params& P, int t, int rows, int* s_flag) {
    const int K = P.K;
    const bool live = t < rows && t < POD_MAX_DETECTIONS;
    const int src = live ? (P.keep ? P.keep[t] : t) : 0;
    float x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    if (live) {
        const float* b = P.boxes + (size_t)src * 4;
        // Boxes.scale (IU:402) then Boxes.clip (IU:403)
        x1 = fminf(fmaxf(*(b + 0) * P.sx, 0.0f), P.out_w);
        y1 = fminf(fmaxf(*(b + 1) * P.sy, 0.0f), P.out_h);
        x2 = fminf(fmaxf(*(b + 2) * P.sx, 0.0f), P.out_w);
        y2 = fminf(fmaxf(*(b + 3) * P.sy, 0.0f), P.out_h);
    }
    const bool ok = live && ((x2 - x1) > 0.0f) && ((y2 - y1) > 0.0f);   // Boxes.nonempty (IU:404)
    // output row = number of non-empty rows before this one: ballots, not a walk over LDS flags (thread 127 paid 127
    // dependent LDS reads, ~5 us of this kernel's 7)
    static_assert(POD_MAX_DETECTIONS == 128, "two wavefronts of rows");
    const unsigned long long mask = __ballot(ok);
    if ((t & 63) == 0 && t < POD_MAX_DETECTIONS) s_flag[t >> 6] = __popcll(mask);
    __syncthreads();
    if (t >= POD_MAX_DETECTIONS) return;
    const int pos = __popcll(mask & ((1ull << (t & 63)) - 1ull)) + (t >= 64 ? s_flag[0] : 0);
    if (t == 0) *P.n_det = s_flag[0] + s_flag[1];
    if (!ok) return;
    *reinterpret_cast<float4*>(P.det_boxes + (size_t)pos * 4) = float4{x1, y1, x2, y2};
    const float score = *(P.scores + src);
    const int32_t cls = *(P.classes + src);
    P.det_scores[pos] = score;
    P.det_classes[pos] = cls;
    float pr[POD_MAX_CLASSES];
#pragma unroll
    for (int k = 0; k < POD_MAX_CLASSES; ++k)
        if (k < K) {
            pr[k] = *(P.probs + (size_t)src * K + k);
            P.det_probs[(size_t)pos * K + k] = pr[k];
        }
    const float s[4] = {P.sx, P.sy, P.sx, P.sy};
    float cv[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float v = P.cov ? *(P.cov + (size_t)src * 16 + a * 4 + b) : 0.0f;
            v = v + ((a == b) ? 1e-4f : 0.0f);                        // IU:409
            v = (s[a] * v) * s[b];                                    // IU:411-424  S cov S^T
            cv[a * 4 + b] = v;
            P.det_cov[(size_t)pos * 16 + a * 4 + b] = v;
        }
    if (P.records) {
        // instances_to_json IU:471-499: XYWH box, T cov T^T with T = [[1,0,0,0],[0,1,0,0],[-1,0,1,0],[0,-1,0,1]]
        float* r = P.records + (size_t)pos * (6 + K + 16);
        r[0] = x1; r[1] = y1; r[2] = x2 - x1; r[3] = y2 - y1;
        r[4] = score;
        r[5] = (float)cls;
#pragma unroll
        for (int k = 0; k < POD_MAX_CLASSES; ++k)
            if (k < K) r[6 + k] = pr[k];
        float tc[16];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            tc[0 * 4 + b] = cv[0 * 4 + b];
            tc[1 * 4 + b] = cv[1 * 4 + b];
            tc[2 * 4 + b] = cv[2 * 4 + b] - cv[0 * 4 + b];
            tc[3 * 4 + b] = cv[3 * 4 + b] - cv[1 * 4 + b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            r[6 + K + a * 4 + 0] = tc[a * 4 + 0];
            r[6 + K + a * 4 + 1] = tc[a * 4 + 1];
            r[6 + K + a * 4 + 2] = tc[a * 4 + 2] - tc[a * 4 + 0];
            r[6 + K + a * 4 + 3] = tc[a * 4 + 3] - tc[a * 4 + 1];
        }
    }
}

__global__ void __launch_bounds__(POD_MAX_DETECTIONS) k7_finalize(const K7Params P) {
    __shared__ int s_flag[POD_MAX_DETECTIONS];
    finalize_rows(P, threadIdx.x, min(*P.n_rows, P.max_det), s_flag);
}

struct K5Params {
    const int32_t* n_total;
    const int32_t* keep;
    const int32_t* n_keep;
    const float* boxes;
    const float* cov;
    const float* scores;
    const int32_t* classes;
    const float* probs;
    int32_t K, box_mode, cls_mode, n_capacity;
    int32_t n_alloc;       // rows the candidate arrays really hold (speculative first loads stay inside)
    float aff;
    float* out_boxes;
    float* out_cov;
    float* out_scores;
    int32_t* out_classes;
    float* out_probs;
};

__global__ void __launch_bounds__(256) k5_bayes_fuse(const K5Params P) {
    __shared__ double s_red[4 * 40];
    const int c = blockIdx.x;
    const int tid = threadIdx.x;
    // Round trip 1 -- everything that does not depend on the centre, issued together: the two counts, this cluster's
    // centre index (keep[] has a slot for every workgroup of the grid) and the first two candidate boxes of this thread.
    // The kernel is a chain of HBM / L2 round trips with a few hundred instructions in between: what counts is how many.
    const int n_live = *P.n_keep;
    const int n_raw = *P.n_total;
    const int ctr = P.keep[c];
    Box pre[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = tid + q * 256;
        pre[q] = j < P.n_alloc ? load_box(P.boxes, j) : Box{0.f, 0.f, 0.f, 0.f};
    }
    if (c >= n_live) return;
    const int n = min(n_raw, P.n_capacity);
    const int K = P.K;
    // round trip 2 -- the centre
    const Box bc = load_box(P.boxes, ctr);
    const int ccls = argmax_probs(P.probs + (size_t)ctr * K, K);     // PI:578-579

    // pass A: total precision, precision-weighted mean, member count, prob sums
    double acc[16 + 4 + 2 + POD_MAX_CLASSES];   // [0,16) sum of precisions, [16,20) sum P mu, [20] same-class members, [21] IoU members, [22,..) prob sums
#pragma unroll
    for (int q = 0; q < 22 + POD_MAX_CLASSES; ++q) acc[q] = 0.0;
    for (int j = tid, it = 0; j < n; j += 256, ++it) {
        const Box bj = it == 0 ? pre[0] : (it == 1 ? pre[1] : load_box(P.boxes, j));
        if (!(iou_pair(bc, bj) > P.aff)) continue;                    // PI:565-566
        // round trip 3 -- the members' probabilities and covariances (a handful of rows per cluster)
        if (P.cls_mode == 1) {                                        // PI:583-585: mean over ALL IoU members
#pragma unroll
            for (int k = 0; k < POD_MAX_CLASSES; ++k)             // static indices: the accumulators stay in registers
                if (k < K) acc[22 + k] += (double)P.probs[(size_t)j * K + k];
            acc[21] += 1.0;
        }
        M4 cv;
        load_m4(P.cov + (size_t)j * 16, cv);                          // issued with the probability loads of argmax_probs
        if (argmax_probs(P.probs + (size_t)j * K, K) != ccls) continue;   // PI:580-582
        M4 pr;
        inv4(cv, pr);                                                 // IU:306
        const double mu[4] = {bj.x1, bj.y1, bj.x2, bj.y2};
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += pr.a[q];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[16 + r] += pr.a[r * 4 + 0] * mu[0] + pr.a[r * 4 + 1] * mu[1] + pr.a[r * 4 + 2] * mu[2] + pr.a[r * 4 + 3] * mu[3];
        acc[20] += 1.0;
    }
    // the class-probability sums are only needed (and only accumulated) in cls_mode 1: 22 instead of 38 fp64 butterflies otherwise
    if (P.cls_mode == 1) block_sum<22 + POD_MAX_CLASSES>(acc, s_red);
    else block_sum<22>(acc, s_red);
    const double m_same = acc[20];
    M4 total;
#pragma unroll
    for (int q = 0; q < 16; ++q) total.a[q] = acc[q];

    double wsum[20];   // weighted precision sum + weighted P*mu (covariance intersection)
    if (P.box_mode == 1 && m_same > 0.0) {
        // IU:313-332: omega_i = (det(T) - det(T - P_i) + det(P_i)) / (m det(T) + sum_i (det(P_i) - det(T - P_i)))
        const double d_tot = det4(total);
        double dsum[1] = {0.0};
        for (int j = tid; j < n; j += 256) {
            const Box bj = load_box(P.boxes, j);
            if (!(iou_pair(bc, bj) > P.aff)) continue;
            if (argmax_probs(P.probs + (size_t)j * K, K) != ccls) continue;
            M4 cv, pr, rest;
            load_m4(P.cov + (size_t)j * 16, cv);
            inv4(cv, pr);
#pragma unroll
            for (int q = 0; q < 16; ++q) rest.a[q] = total.a[q] - pr.a[q];
            dsum[0] += det4(pr) - det4(rest);
        }
        block_sum<1>(dsum, s_red);
        const double denom = m_same * d_tot + dsum[0];
#pragma unroll
        for (int q = 0; q < 20; ++q) wsum[q] = 0.0;
        for (int j = tid; j < n; j += 256) {
            const Box bj = load_box(P.boxes, j);
            if (!(iou_pair(bc, bj) > P.aff)) continue;
            if (argmax_probs(P.probs + (size_t)j * K, K) != ccls) continue;
            M4 cv, pr, rest;
            load_m4(P.cov + (size_t)j * 16, cv);
            inv4(cv, pr);
#pragma unroll
            for (int q = 0; q < 16; ++q) rest.a[q] = total.a[q] - pr.a[q];
            const double omega = (d_tot - det4(rest) + det4(pr)) / denom;
            const double mu[4] = {bj.x1, bj.y1, bj.x2, bj.y2};
#pragma unroll
            for (int q = 0; q < 16; ++q) wsum[q] += omega * pr.a[q];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wsum[16 + r] += omega * (pr.a[r * 4 + 0] * mu[0] + pr.a[r * 4 + 1] * mu[1] + pr.a[r * 4 + 2] * mu[2] + pr.a[r * 4 + 3] * mu[3]);
        }
        block_sum<20>(wsum, s_red);
    } else {
#pragma unroll
        for (int q = 0; q < 20; ++q) wsum[q] = acc[q];
    }

    if (tid == 0) {
        float* ob = P.out_boxes + (size_t)c * 4;
        float* oc = P.out_cov + (size_t)c * 16;
        if (m_same > 0.0) {
            M4 psum, fused;
#pragma unroll
            for (int q = 0; q < 16; ++q) psum.a[q] = wsum[q];
            inv4(psum, fused);                                        // IU:308 / IU:326
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ob[r] = (float)(fused.a[r * 4 + 0] * wsum[16] + fused.a[r * 4 + 1] * wsum[17] + fused.a[r * 4 + 2] * wsum[18] + fused.a[r * 4 + 3] * wsum[19]);
#pragma unroll
            for (int q = 0; q < 16; ++q) oc[q] = (float)fused.a[q];
        } else {   // degenerate centre (self-IoU 0, Q12): keep the centre's own estimate
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = P.boxes[(size_t)ctr * 4 + r];
#pragma unroll
            for (int q = 0; q < 16; ++q) oc[q] = P.cov[(size_t)ctr * 16 + q];
        }
        float* op = P.out_probs + (size_t)c * K;
        if (P.cls_mode == 1 && acc[21] > 0.0) {                       // PI:583-585, :609-613
            float best = 0.0f;
            int bk = 0;
#pragma unroll
            for (int k = 0; k < POD_MAX_CLASSES; ++k) {
                if (k < K) {
                    const float p = (float)(acc[22 + k] / acc[21]);
                    op[k] = p;
                    if (k == 0 || p > best) {
                        best = p;
                        bk = k;
                    }
                }
            }
            P.out_scores[c] = best;
            P.out_classes[c] = bk;
        } else {                                                      // PI:614-617 max_score
            for (int k = 0; k < K; ++k) op[k] = P.probs[(size_t)ctr * K + k];
            P.out_scores[c] = P.scores[ctr];
            P.out_classes[c] = P.classes[ctr];
        }
    }
}

struct K6Params {
    const int32_t* n_total;
    const int32_t* keep;
    const int32_t* n_keep;
    const float* boxes;
    const float* cov;      // may be null
    const int32_t* classes;
    const float* probs;
    int32_t K, n_capacity;
    int32_t n_alloc;         // rows the candidate arrays really hold (speculative first loads stay inside)
    int32_t n_keep_alloc;    // entries of keep[]
    int32_t ensemble_rule;   // 0: anchor statistics (IoU > aff, IU:102 singleton rule); 1: black-box ensembles (IoU >= aff, IU:211-247)
    float aff;
    float* out_boxes;
    float* out_cov;
    float* out_scores;
    int32_t* out_classes;
    float* out_probs;
};

__global__ void __launch_bounds__(256) k6_anchor_stats(const K6Params P) {
    __shared__ double s_red[4 * 40];
    const int c = blockIdx.x;
    const int tid = threadIdx.x;
    // round trip 1: counts, centre index, this thread's first two candidate boxes and classes (see k5_bayes_fuse)
    const int n_live = *P.n_keep;
    const int n_raw = *P.n_total;
    const int ctr = c < P.n_keep_alloc ? P.keep[c] : 0;
    Box pre[2];
    int pre_cls[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = tid + q * 256;
        pre[q] = j < P.n_alloc ? load_box(P.boxes, j) : Box{0.f, 0.f, 0.f, 0.f};
        pre_cls[q] = j < P.n_alloc ? P.classes[j] : -1;
    }
    if (c >= n_live) return;
    const int n = min(n_raw, P.n_capacity);
    const int K = P.K;
    const Box bc = load_box(P.boxes, ctr);
    const int ccls = P.classes[ctr];                                 // IU:104
    const bool has_cov = P.cov != nullptr;

    // pass 1: member counts, box sum, prob sum, covariance sum of same-class members
    double acc[2 + 4 + POD_MAX_CLASSES + 16];
#pragma unroll
    for (int q = 0; q < 22 + POD_MAX_CLASSES; ++q) acc[q] = 0.0;
    const bool ens = P.ensemble_rule != 0;
    for (int j = tid, it = 0; j < n; j += 256, ++it) {
        const Box bj = it == 0 ? pre[0] : (it == 1 ? pre[1] : load_box(P.boxes, j));
        const float iou = iou_pair(bc, bj);
        if (!(ens ? iou >= P.aff : iou > P.aff)) continue;            // IU:91-92 (>) / IU:211-212 (>=)
        acc[0] += 1.0;                                                // IU:102 counts every IoU member
        const int cj = it == 0 ? pre_cls[0] : (it == 1 ? pre_cls[1] : P.classes[j]);
        if (cj != ccls) continue;                                     // IU:104-106
        acc[1] += 1.0;
        acc[2] += bj.x1; acc[3] += bj.y1; acc[4] += bj.x2; acc[5] += bj.y2;
#pragma unroll
        for (int k = 0; k < POD_MAX_CLASSES; ++k)                 // static indices: the accumulators stay in registers
            if (k < K) acc[6 + k] += (double)P.probs[(size_t)j * K + k];
        if (has_cov)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[6 + POD_MAX_CLASSES + q] += (double)P.cov[(size_t)j * 16 + q];
    }
    block_sum<22 + POD_MAX_CLASSES>(acc, s_red);
    const double m_all = acc[0], m = acc[1];
    const bool cluster = ens ? (m >= 1.0) : (m_all >= 2.0 && m >= 1.0);   // IU:226-247: a 1-member cluster is its own mean
    // the reference forms the mean in fp32 and subtracts it from fp32 boxes (IU:112-114)
    float mu[4] = {0, 0, 0, 0};
    if (cluster)
#pragma unroll
        for (int r = 0; r < 4; ++r) mu[r] = (float)(acc[2 + r] / m);
    double rr[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) rr[q] = 0.0;
    if (cluster) {
        for (int j = tid; j < n; j += 256) {
            const Box bj = load_box(P.boxes, j);
            const float iou = iou_pair(bc, bj);
            if (!(ens ? iou >= P.aff : iou > P.aff)) continue;
            if (P.classes[j] != ccls) continue;
            const float r[4] = {bj.x1 - mu[0], bj.y1 - mu[1], bj.x2 - mu[2], bj.y2 - mu[3]};
            int q = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b, ++q) rr[q] += (double)(r[a] * r[b]);
        }
    }
    block_sum<10>(rr, s_red);
    if (tid == 0) {
        float* ob = P.out_boxes + (size_t)c * 4;
        float* oc = P.out_cov + (size_t)c * 16;
        float* op = P.out_probs + (size_t)c * K;
        if (cluster) {
            const double denom = fmax(m - 1.0, 1.0);                  // IU:116
            int q = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b, ++q) {
                    float v = (float)(rr[q] / denom);
                    oc[a * 4 + b] = v;
                    oc[b * 4 + a] = v;
                }
            if (has_cov)                                              // IU:120-123
#pragma unroll
                for (int e = 0; e < 16; ++e) oc[e] = oc[e] + (float)(acc[6 + POD_MAX_CLASSES + e] / m);
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = mu[r];
#pragma unroll
            for (int k = 0; k < POD_MAX_CLASSES; ++k)
                if (k < K) op[k] = (float)(acc[6 + k] / m);           // IU:126
        } else {                                                      // IU:127-133
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r] = P.boxes[(size_t)ctr * 4 + r];
            for (int k = 0; k < K; ++k) op[k] = P.probs[(size_t)ctr * K + k];
#pragma unroll
            for (int e = 0; e < 16; ++e) oc[e] = has_cov ? P.cov[(size_t)ctr * 16 + e] : ((e % 5 == 0) ? 1e-4f : 0.0f);
        }
        const int bk = argmax_probs(op, K);                           // IU:146-152 (Q10)
        P.out_scores[c] = op[bk];
        P.out_classes[c] = bk;
    }
}

// ---- post-NMS ensemble merge: sequential same-class clustering IU:203-215 ------------------------
// Box i seeds a cluster unless an EARLIER seed's cluster already contains it (IoU >= aff and same class): a greedy sweep
// in index order.  One 1024-thread workgroup, LDS "covered" bitmap, thread 0 finds the next uncovered index.
struct KSeedParams {
    const int32_t* m_total;
    int32_t capacity;
    float aff;
    const float* boxes;
    const int32_t* classes;
    int32_t* seeds;
    int32_t* n_seeds;
};

__global__ void __launch_bounds__(1024) k_ensemble_seeds(const KSeedParams P) {
    __shared__ unsigned long long s_cov[POD_MAX_CANDIDATES / 64];
    __shared__ int s_cur, s_n;
    const int tid = threadIdx.x;
    const int M = min(*P.m_total, P.capacity);
    for (int i = tid; i < POD_MAX_CANDIDATES / 64; i += 1024) s_cov[i] = 0ull;
    if (tid == 0) {
        s_cur = -1;
        s_n = 0;
    }
    __syncthreads();
    const int nwords = (M + 63) >> 6;
    while (true) {
        if (tid == 0) {
            int next = -1;
            const int start = s_cur + 1;
            for (int w = start >> 6; w < nwords; ++w) {
                unsigned long long live = ~s_cov[w];
                if (w == (start >> 6)) live &= ~0ull << (start & 63);
                if (live) {
                    const int cand = (w << 6) + __ffsll((long long)live) - 1;
                    if (cand < M) next = cand;
                    break;
                }
            }
            if (next >= 0) {
                P.seeds[s_n] = next;
                s_n = s_n + 1;
            }
            s_cur = next;
        }
        __syncthreads();
        const int i = s_cur;
        if (i < 0) break;
        const Box bi = load_box(P.boxes, i);
        const int ci = P.classes[i];
        for (int j = i + 1 + tid; j < M; j += 1024)
            if (P.classes[j] == ci && iou_pair(bi, load_box(P.boxes, j)) >= P.aff) atomicOr(&s_cov[j >> 6], 1ull << (j & 63));
        __syncthreads();
    }
    if (tid == 0) *P.n_seeds = s_n;
}

// Appends the rows `keep[0:n_keep)` of one ensemble member's candidate arrays to the concatenated member-detection
// arrays (torch.cat of IU:191-196); `total` is the running row count on the device.
struct KAppendParams {
    const int32_t* keep;
    const int32_t* n_keep;
    const float* boxes;
    const float* cov;      // may be null -> zeros (IU:52-53)
    const int32_t* classes;
    const float* probs;
    int32_t K, capacity;
    float* dst_boxes;
    float* dst_cov;
    int32_t* dst_classes;
    float* dst_probs;
    int32_t* total;
};

__global__ void __launch_bounds__(POD_MAX_DETECTIONS) k_append_rows(const KAppendParams P) {
    const int t = threadIdx.x;
    const int n = *P.n_keep;
    const int base = *P.total;
    __syncthreads();
    if (t < n && base + t < P.capacity) {
        const int src = P.keep[t], dst = base + t;
        *reinterpret_cast<float4*>(P.dst_boxes + (size_t)dst * 4) = *reinterpret_cast<const float4*>(P.boxes + (size_t)src * 4);
        for (int e = 0; e < 16; ++e) P.dst_cov[(size_t)dst * 16 + e] = P.cov ? P.cov[(size_t)src * 16 + e] : 0.0f;
        P.dst_classes[dst] = P.classes[src];
        for (int k = 0; k < P.K; ++k) P.dst_probs[(size_t)dst * P.K + k] = P.probs[(size_t)src * P.K + k];
    }
    __syncthreads();
    if (t == 0) *P.total = min(base + n, P.capacity);
}

__global__ void __launch_bounds__(256) k_reg_nll(const float* means, const float* covs, const float* gt, int n, float* nll) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // -log N(gt; mean, cov + 1e-2 I) through a 4x4 Cholesky factor
    double a[4][4], L[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) a[r][c] = (double)(covs[(size_t)i * 16 + r * 4 + c] + ((r == c) ? 1e-2f : 0.0f));
    double logdet = 0.0;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c <= r; ++c) {
            double s = a[r][c];
            for (int k = 0; k < c; ++k) s -= L[r][k] * L[c][k];
            if (r == c) {
                L[r][r] = sqrt(s);
                logdet += log(L[r][r]);
            } else {
                L[r][c] = s / L[c][c];
            }
        }
    double y[4], maha = 0.0;
    for (int r = 0; r < 4; ++r) {
        double s = (double)gt[(size_t)i * 4 + r] - (double)means[(size_t)i * 4 + r];
        for (int k = 0; k < r; ++k) s -= L[r][k] * y[k];
        y[r] = s / L[r][r];
        maha += y[r] * y[r];
    }
    nll[i] = (float)(0.5 * maha + logdet + 2.0 * 1.8378770664093453);   // 0.5 * 4 * log(2 pi)
}

}  // namespace pod

extern "C" int pod_bayes_fuse(const PodConfig* cfg, const int32_t* n_total, const int32_t* keep, const int32_t* n_keep,
                              const float* boxes, const float* cov, const float* scores, const int32_t* classes,
                              const float* probs, int32_t box_mode, int32_t cls_mode, float* out_boxes, float* out_cov,
                              float* out_scores, int32_t* out_classes, float* out_probs, pod_stream_t stream) {
    if (!cfg || !n_total || !keep || !n_keep || !boxes || !cov || !scores || !classes || !probs || !out_boxes || !out_cov ||
        !out_scores || !out_classes || !out_probs)
        return POD_E_INVALID;
    if (box_mode < 0 || box_mode > 1 || cls_mode < 0 || cls_mode > 1) return POD_E_INVALID;
    if (cfg->num_classes < 1 || cfg->num_classes >= POD_MAX_CLASSES) return POD_E_INVALID;
    if (cfg->max_detections < 1 || cfg->max_detections > POD_MAX_DETECTIONS) return POD_E_INVALID;
    pod::K5Params P;
    P.n_total = n_total; P.keep = keep; P.n_keep = n_keep; P.boxes = boxes; P.cov = cov; P.scores = scores;
    P.classes = classes; P.probs = probs; P.K = cfg->num_classes; P.box_mode = box_mode; P.cls_mode = cls_mode;
    P.n_capacity = POD_MAX_CANDIDATES; P.aff = cfg->affinity_thresh; P.n_alloc = cfg->n_levels * cfg->topk;
    P.out_boxes = out_boxes; P.out_cov = out_cov; P.out_scores = out_scores; P.out_classes = out_classes; P.out_probs = out_probs;
    hipLaunchKernelGGL(pod::k5_bayes_fuse, dim3(cfg->max_detections), dim3(256), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_anchor_stats_merge(const PodConfig* cfg, const int32_t* n_total, const int32_t* keep,
                                      const int32_t* n_keep, const float* boxes, const float* cov, const int32_t* classes,
                                      const float* probs, float* out_boxes, float* out_cov, float* out_scores,
                                      int32_t* out_classes, float* out_probs, pod_stream_t stream) {
    if (!cfg || !n_total || !keep || !n_keep || !boxes || !classes || !probs || !out_boxes || !out_cov || !out_scores ||
        !out_classes || !out_probs)
        return POD_E_INVALID;
    if (cfg->num_classes < 1 || cfg->num_classes > POD_MAX_CLASSES) return POD_E_INVALID;
    if (cfg->max_detections < 1 || cfg->max_detections > POD_MAX_DETECTIONS) return POD_E_INVALID;
    pod::K6Params P;
    P.n_total = n_total; P.keep = keep; P.n_keep = n_keep; P.boxes = boxes; P.cov = cov; P.classes = classes; P.probs = probs;
    P.K = cfg->num_classes; P.n_capacity = POD_MAX_CANDIDATES; P.aff = cfg->affinity_thresh; P.ensemble_rule = 0;
    P.n_alloc = cfg->n_levels * cfg->topk; P.n_keep_alloc = cfg->max_detections;
    P.out_boxes = out_boxes; P.out_cov = out_cov; P.out_scores = out_scores; P.out_classes = out_classes; P.out_probs = out_probs;
    hipLaunchKernelGGL(pod::k6_anchor_stats, dim3(cfg->max_detections), dim3(256), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_ensemble_append(const PodConfig* cfg, const int32_t* keep, const int32_t* n_keep, const float* boxes,
                                   const float* cov, const int32_t* classes, const float* probs, int32_t capacity,
                                   float* dst_boxes, float* dst_cov, int32_t* dst_classes, float* dst_probs, int32_t* total,
                                   pod_stream_t stream) {
    if (!cfg || !keep || !n_keep || !boxes || !classes || !probs || !dst_boxes || !dst_cov || !dst_classes || !dst_probs || !total)
        return POD_E_INVALID;
    if (capacity < 1 || capacity > POD_MAX_CANDIDATES || cfg->num_classes < 1 || cfg->num_classes > POD_MAX_CLASSES) return POD_E_INVALID;
    pod::KAppendParams P;
    P.keep = keep; P.n_keep = n_keep; P.boxes = boxes; P.cov = cov; P.classes = classes; P.probs = probs; P.K = cfg->num_classes;
    P.capacity = capacity; P.dst_boxes = dst_boxes; P.dst_cov = dst_cov; P.dst_classes = dst_classes; P.dst_probs = dst_probs; P.total = total;
    hipLaunchKernelGGL(pod::k_append_rows, dim3(1), dim3(POD_MAX_DETECTIONS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_ensemble_merge(const PodConfig* cfg, const int32_t* m_total, int32_t capacity, const float* boxes,
                                  const float* cov, const int32_t* classes, const float* probs, int32_t* seeds, int32_t* n_seeds,
                                  float* out_boxes, float* out_cov, float* out_scores, int32_t* out_classes, float* out_probs,
                                  pod_stream_t stream) {
    if (!cfg || !m_total || !boxes || !cov || !classes || !probs || !seeds || !n_seeds || !out_boxes || !out_cov || !out_scores ||
        !out_classes || !out_probs)
        return POD_E_INVALID;
    if (capacity < 1 || capacity > POD_MAX_CANDIDATES || cfg->num_classes < 1 || cfg->num_classes > POD_MAX_CLASSES) return POD_E_INVALID;
    pod::KSeedParams S;
    S.m_total = m_total; S.capacity = capacity; S.aff = cfg->affinity_thresh; S.boxes = boxes; S.classes = classes; S.seeds = seeds;
    S.n_seeds = n_seeds;
    hipLaunchKernelGGL(pod::k_ensemble_seeds, dim3(1), dim3(1024), 0, (hipStream_t)stream, S);
    POD_CHECK_LAUNCH();
    pod::K6Params P;
    P.n_total = m_total; P.keep = seeds; P.n_keep = n_seeds; P.boxes = boxes; P.cov = cov; P.classes = classes; P.probs = probs;
    P.K = cfg->num_classes; P.n_capacity = capacity; P.aff = cfg->affinity_thresh; P.ensemble_rule = 1;
    P.n_alloc = capacity; P.n_keep_alloc = capacity;
    P.out_boxes = out_boxes; P.out_cov = out_cov; P.out_scores = out_scores; P.out_classes = out_classes; P.out_probs = out_probs;
    hipLaunchKernelGGL(pod::k6_anchor_stats, dim3(capacity), dim3(256), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_finalize(const PodConfig* cfg, const int32_t* keep, const int32_t* n_rows, const float* boxes,
                            const float* cov, const float* scores, const int32_t* classes, const float* probs,
                            float scale_x, float scale_y, float out_h, float out_w, float* det_boxes, float* det_cov,
                            float* det_scores, int32_t* det_classes, float* det_probs, float* records, int32_t* n_det,
                            pod_stream_t stream) {
    if (!cfg || !n_rows || !boxes || !scores || !classes || !probs || !det_boxes || !det_cov || !det_scores || !det_classes ||
        !det_probs || !n_det)
        return POD_E_INVALID;
    if (cfg->num_classes < 1 || cfg->num_classes > POD_MAX_CLASSES) return POD_E_INVALID;
    if (cfg->max_detections < 1 || cfg->max_detections > POD_MAX_DETECTIONS) return POD_E_INVALID;
    pod::K7Params P;
    P.keep = keep; P.n_rows = n_rows; P.boxes = boxes; P.cov = cov; P.scores = scores; P.classes = classes; P.probs = probs;
    P.K = cfg->num_classes; P.max_det = cfg->max_detections; P.sx = scale_x; P.sy = scale_y; P.out_h = out_h; P.out_w = out_w;
    P.det_boxes = det_boxes; P.det_cov = det_cov; P.det_scores = det_scores; P.det_classes = det_classes;
    P.det_probs = det_probs; P.records = records; P.n_det = n_det;
    hipLaunchKernelGGL(pod::k7_finalize, dim3(1), dim3(POD_MAX_DETECTIONS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_reg_nll(const float* means, const float* covs, const float* gt, int32_t n, float* nll, pod_stream_t stream) {
    if (!means || !covs || !gt || !nll || n < 0) return POD_E_INVALID;
    if (n == 0) return POD_OK;
    hipLaunchKernelGGL(pod::k_reg_nll, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, means, covs, gt, n, nll);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
