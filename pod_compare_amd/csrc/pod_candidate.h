// Per-candidate device code shared by K2b (gather), K3 (decode + covariance) and their fused form.
//
// gather_candidate  : probabilistic_inference.py:305-338, :341-342 (+ the N-run merge of PI:211-270 at this anchor)
// decode_candidate  : modeling_utils.py:4-22, probabilistic_inference.py:323-385, inference_utils.py:337-371, :510-547
// One wavefront per candidate in both; the fused kernel hands the merged deltas / log-variances over in registers and
// the per-run deltas in LDS instead of through the candidate arrays in HBM.
#pragma once
#include "pod_device.h"

namespace pod {

struct K2bParams {
    PodLevel lv[POD_MAX_LEVELS];
    int32_t n_levels, n_runs, A, K, D, has_cls_var, quirk, cls_samples, topk;
    uint64_t seed;
    const float* anchors;
    const uint64_t* cat_keys;      // level-concatenated selection written by K2 (row = candidate)
    const int32_t* cat_level;
    const int32_t* n_total;        // number of candidates (written by K2)
    int32_t* cand_count;           // the per-level counters K1 / K1b appended with: consumed (zeroed) here
    const float* probs_dense;      // (R, K) class probabilities K1b stored for the anchors it emitted, or null (recompute)
    int32_t* cand_anchor_idx;
    int32_t* cand_level;
    float* cand_score;
    int32_t* cand_class;
    float* cand_probs;
    float* cand_delta;
    float* cand_reg_var;
    float* cand_anchor;
    float* cand_run_delta;
};

// Merged value of one element (plane-layout offset `e` inside a run) in the reference order; all N
// loads of a batch are issued before the first add.  Optionally hands every run's raw value to `sink`.
template <class Sink>
__device__ __forceinline__ float merge_scalar(const float* base, int64_t rs, int64_t e, int n_runs, int quirk, Sink sink) {
    float acc = 0.0f;
    float x0 = 0.0f;
    for (int r0 = 0; r0 < n_runs; r0 += 16) {      // N = 10 (MC dropout) / 5 (ensembles): ONE round of independent loads
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r0 + j < n_runs) v[j] = base[(int64_t)(r0 + j) * rs + e];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int run = r0 + j;
            if (run < n_runs) {
                sink(run, v[j]);
                if (run == 0) {
                    x0 = v[j];
                    acc = (quirk && n_runs > 1) ? x0 + x0 : x0;   // term 0 (+ term 1 = run 0 again, PI:216-219)
                } else if (!quirk || run < n_runs - 1) {
                    acc = acc + v[j];                             // quirk: the last run is never added
                }
            }
        }
    }
    return n_runs == 1 ? acc : __fdiv_rn(acc, (float)n_runs);
}

struct GatheredCandidate {
    float4 anchor;       // its anchor box
    int dst, level, r;   // row in the level-concatenated candidate arrays; level; anchor index inside the level
    float merged;        // this lane's channel: [0,K) logits, [K,2K) log-variances, then 4 deltas, then D reg_var entries
};

// One wavefront per selected candidate.  Lane c < C = 2K+4+D owns channel c of the anchor
// ([0,K) logits, [K,2K) log-variances, then 4 deltas, then D reg_var entries): it loads that
// channel of all N runs (independent loads), merges them in the reference order, and the K logit
// lanes then re-derive the class probabilities with K1's device function (bit-identical).
// Returns false when there is no candidate row `dst`.  run_delta_lds (N*4 floats) also receives every run's raw delta.
__device__ __forceinline__ bool gather_candidate(const K2bParams& P, int dst, int lane, float* run_delta_lds, GatheredCandidate& out) {
    // workgroup `dst` = candidate row `dst` of the level-concatenated list K2 wrote (key + level per row): the count, the
    // key and the level are three independent loads, one round trip (walking per-level counts and then fetching the key
    // was two dependent ones, in a kernel whose gather phase is nothing but a chain of HBM round trips)
    const int n = *P.n_total;
    if (dst == 0 && lane < P.n_levels) P.cand_count[lane] = 0;   // consumed: the next image's K1 appends from zero
    const int row = dst < P.n_levels * P.topk ? dst : 0;      // the arrays hold n_levels * topk rows; n never exceeds that
    const uint64_t key = P.cat_keys[row];
    const int l = P.cat_level[row];
    if (dst >= n) return false;
    POD_STAMP(dst, 1);
    const PodLevel& lv = P.lv[l];
    const int r = key_index(key);
    const int A = P.A, K = P.K, D = P.D, N = P.n_runs;
    const int hw = r / A;
    const int a = r - hw * A;
    const int64_t HW = (int64_t)lv.H * lv.W;
    const bool has_var = P.has_cls_var != 0;
    const int nvar = has_var ? K : 0;
    const int C = K + nvar + 4 + D;
    // issued together with the run loads below (it used to trail them by a full round trip)
    const float4 anc = *reinterpret_cast<const float4*>(P.anchors + ((int64_t)lv.anchor_base + r) * 4);
    float merged = 0.0f;
    if (lane < K + nvar && P.probs_dense) {
        // class channels: only needed to re-derive the probabilities, which K1b already stored for this anchor
    } else if (lane < K) {
        merged = merge_scalar(lv.cls, lv.run_stride_cls, (int64_t)(a * K + lane) * HW + hw, N, P.quirk, [](int, float) {});
    } else if (lane < K + nvar) {
        merged = merge_scalar(lv.cls_var, lv.run_stride_cls, (int64_t)(a * K + lane - K) * HW + hw, N, P.quirk, [](int, float) {});
    } else if (lane < K + nvar + 4) {
        const int c = lane - K - nvar;
        float* rd = P.cand_run_delta;
        merged = merge_scalar(lv.delta, lv.run_stride_delta, (int64_t)(a * 4 + c) * HW + hw, N, P.quirk,
                              [=](int run, float v) {
                                  if (rd) rd[((int64_t)dst * N + run) * 4 + c] = v;
                                  if (run_delta_lds) run_delta_lds[run * 4 + c] = v;
                              });
        P.cand_delta[(int64_t)dst * 4 + c] = merged;
    } else if (lane < C) {
        const int c = lane - K - nvar - 4;
        merged = merge_scalar(lv.reg_var, lv.run_stride_reg, (int64_t)(a * D + c) * HW + hw, N, P.quirk, [](int, float) {});
        P.cand_reg_var[(int64_t)dst * D + c] = merged;
    }
    POD_STAMP(dst, 2);
    const float lvar = has_var ? __shfl(merged, (lane < K ? lane : 0) + K, 64) : 0.0f;
    float p = -1.0f;
    if (lane < K) {
        // K1b evaluated exactly this function on exactly these merged values when it emitted the anchor: reuse its result
        // (bit-identical; saves 3 Philox calls + 10 sigmoids on the critical path of every candidate)
        p = P.probs_dense ? P.probs_dense[((int64_t)lv.anchor_base + r) * K + lane]
                          : class_prob_cell(merged, lvar, has_var, P.cls_samples, lv.eps_cls, HW * A, K, A, l, hw, a, lane, P.seed);
        P.cand_probs[(int64_t)dst * K + lane] = p;
    }
    POD_STAMP(dst, 3);
    // max / first argmax over the K class lanes
    float best = __shfl(p, 0, 64);
    int best_k = 0;
    for (int k = 1; k < K; ++k) {
        const float v = __shfl(p, k, 64);
        if (v > best) {
            best = v;
            best_k = k;
        }
    }
    if (lane == 0) {
        P.cand_score[dst] = best;            // == key_score(key) by construction
        P.cand_class[dst] = best_k;
        P.cand_anchor_idx[dst] = r;
        P.cand_level[dst] = l;
        *reinterpret_cast<float4*>(P.cand_anchor + (int64_t)dst * 4) = anc;
    }
    out.anchor = anc;
    out.dst = dst;
    out.level = l;
    out.r = r;
    out.merged = merged;
    return true;
}

struct K3Params {
    int32_t anchor_base[POD_MAX_LEVELS];
    int32_t n_runs, D, S, n_capacity, n_replay;
    float wts[4];
    uint64_t seed;
    const int32_t* n_total;
    const float* cand_delta;
    const float* cand_reg_var;
    const float* cand_anchor;
    const float* cand_run_delta;
    const int32_t* cand_anchor_idx;
    const int32_t* cand_level;
    const float* eps_prop;
    float* boxes;
    float* cov;
};

// torch cascade_sum combination of per-block sums (blocks of 16 rows): block sums accumulate into
// acc1; every 16 blocks (256 rows) acc1 is flushed into acc2; the trailing partial block is acc0.
// `col[b]` (64 consecutive, 16-byte aligned floats of LDS) holds the sum of block b (b < S / 16) and, if S % 16 != 0,
// col[S / 16] the sum of the tail.  The whole column is fetched with 16 independent 16-byte LDS reads BEFORE the first
// add: written as a loop of scalar reads the compiler paired every add with its own LDS round trip (64 x ~100 cycles,
// 2.5-2.8 us per call, two calls per candidate -- measured with in-kernel time stamps, tools/exp_trace.py).
__device__ __forceinline__ float cascade_combine(const float* col, int S) {
    constexpr int MAXB = POD_MAX_PROP_SAMPLES / 16;     // 64 blocks
    const int nfull = S >> 4;
    float4 q[MAXB / 4];
#pragma unroll
    for (int i = 0; i < MAXB / 4; ++i) q[i] = reinterpret_cast<const float4*>(col)[i];
    __builtin_amdgcn_sched_barrier(0);                   // all reads issued; the adds below wait once
    float acc1 = 0.0f, acc2 = 0.0f, tail = 0.0f;
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
        const float4 w = q[b >> 2];
        const float v = (b & 3) == 0 ? w.x : ((b & 3) == 1 ? w.y : ((b & 3) == 2 ? w.z : w.w));
        if (b < nfull) {
            acc1 = acc1 + v;
            if (((b + 1) & 15) == 0) {
                acc2 = acc2 + acc1;
                acc1 = 0.0f;
            }
        } else if (b == nfull) {
            tail = v;
        }
    }
    float acc0 = (S & 15) ? tail : 0.0f;
    acc0 = acc0 + acc1;
    acc0 = acc0 + acc2;
    return acc0;
}

// ---- one sample of the propagation (PI:351-356 rsample + IU:510-547 decode) -------------------------------------------
struct CholeskyL {
    float m[4][4];   // row-major lower triangle, MU:4-22
};

__device__ __forceinline__ CholeskyL cholesky_from_head(const float (&rv)[10], int D) {
    CholeskyL L;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) L.m[r][c] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) L.m[c][c] = sqrtf(expf(rv[c]));
    if (D == 10) {   // torch.tril_indices(4,4,-1): (1,0),(2,0),(2,1),(3,0),(3,1),(3,2)
        L.m[1][0] = rv[4]; L.m[2][0] = rv[5]; L.m[2][1] = rv[6];
        L.m[3][0] = rv[7]; L.m[3][1] = rv[8]; L.m[3][2] = rv[9];
    }
    return L;
}

// delta + L eps, decoded against the anchor: the ONE definition of a sample's value, whoever computes it
__device__ __forceinline__ float4 decode_sample(const K3Params& P, const float (&dl)[4], const CholeskyL& L, const float (&e)[4],
                                                const Box& anc) {
    float d[4];
    if (P.D == 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = dl[c] + L.m[c][c] * e[c];   // L eps exact for diagonal L
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = L.m[c][0] * e[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) acc = fmaf(L.m[c][k], e[k], acc);
            d[c] = dl[c] + acc;
        }
    }
    const Box b = decode_box(d[0], d[1], d[2], d[3], anc, P.wts);
    return float4{b.x1, b.y1, b.x2, b.y2};
}

// Draw + decode the S samples of one candidate into LDS, by `nthreads` threads (thread u of them).  The expensive part of
// a candidate -- Philox, Box-Muller, exp -- is embarrassingly parallel over samples, so the fused kernel spreads it over
// 4 wavefronts when there are few candidates; the ORDER-sensitive part (torch's block sums) stays with one wavefront,
// which reads the decoded samples back from LDS (decode_candidate<true>).  Native draws only: one Philox call serves the
// two samples (2m, 2m+1); thread u takes calls m = u, u + nthreads, ...   xs: S float4 of LDS.
__device__ __forceinline__ void generate_samples(const K3Params& P, int u, int nthreads, const float (&dl)[4], const float (&rv)[10],
                                                 const Box& anc, uint32_t gid, float4* xs) {
    const int S = P.S;
    const CholeskyL L = cholesky_from_head(rv, P.D);
    for (int m = u; 2 * m < S; m += nthreads) {
        const f32x8n z = philox_normals8(P.seed, gid, (uint32_t)m, 0u, STREAM_BOX);
        const float e0[4] = {z.v[0], z.v[1], z.v[2], z.v[3]};
        xs[2 * m] = decode_sample(P, dl, L, e0, anc);
        if (2 * m + 1 < S) {
            const float e1[4] = {z.v[4], z.v[5], z.v[6], z.v[7]};
            xs[2 * m + 1] = decode_sample(P, dl, L, e1, anc);
        }
    }
}

// All LDS traffic of decode_candidate stays inside one wavefront: order it without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Candidate `i`, ONE wavefront (lane = threadIdx & 63 of the calling wave): sample moments in torch's summation order +
// epistemic covariance + stores.  Lane l owns samples [16l, 16l+16) = one 16-row block of torch's cascade sum.
//   FROM_LDS  : the samples were produced by generate_samples (xs_lds; the caller synchronised the producers);
//   otherwise : this wavefront draws / replays and decodes its own 16 samples per lane (rv, gid, P.eps_prop).
// dl its merged deltas, anc its anchor, run_delta = the N runs' raw deltas (N x 4 floats, HBM or LDS).
// part: 10*64 floats of LDS (16-byte aligned), small: 16 + 4*POD_MAX_RUNS floats of LDS -- private to this wavefront.
template <bool FROM_LDS>
__device__ __forceinline__ void decode_candidate(const K3Params& P, int i, int lane, const float (&dl)[4], const float (&rv)[10], const Box& anc,
                                                 uint32_t gid, const float* run_delta, const float4* xs_lds, float* part, float* small) {
    const int S = P.S, D = P.D, N = P.n_runs;
    float mean[4] = {0, 0, 0, 0};
    float cv[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) cv[c] = 0.0f;

    if (D > 0) {
        // ---- pass 1: the lane's 16 samples, block sums --------------------------------------------------------
        float xs[16][4];
        float bs[4] = {0, 0, 0, 0};
        CholeskyL L;
        if (!FROM_LDS) L = cholesky_from_head(rv, D);
        f32x8n z;   // native draws: one Philox call serves the two samples (2m, 2m+1)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int s = lane * 16 + t;
            const bool live = s < S;
            float4 v = float4{0.f, 0.f, 0.f, 0.f};
            if (FROM_LDS) {
                if (live) v = xs_lds[s];
            } else {
                float e[4] = {0, 0, 0, 0};
                if (live) {
                    if (P.eps_prop) {
                        const float4 e4 = *reinterpret_cast<const float4*>(P.eps_prop + ((size_t)s * P.n_replay + i) * 4);
                        e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
                    } else {
                        if ((t & 1) == 0) z = philox_normals8(P.seed, gid, (uint32_t)(s >> 1), 0u, STREAM_BOX);
                        e[0] = z.v[(t & 1) * 4 + 0]; e[1] = z.v[(t & 1) * 4 + 1]; e[2] = z.v[(t & 1) * 4 + 2]; e[3] = z.v[(t & 1) * 4 + 3];
                    }
                }
                v = decode_sample(P, dl, L, e, anc);
            }
            xs[t][0] = v.x; xs[t][1] = v.y; xs[t][2] = v.z; xs[t][3] = v.w;
#pragma unroll
            for (int c = 0; c < 4; ++c) bs[c] = live ? bs[c] + xs[t][c] : bs[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) part[c * 64 + lane] = bs[c];        // component-major: a column per component
        wave_sync();
        POD_STAMP(i + 2048, 0);
        if (lane < 4) small[lane] = __fdiv_rn(cascade_combine(part + lane * 64, S), (float)S);
        wave_sync();
        POD_STAMP(i + 2048, 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) mean[c] = small[c];
        wave_sync();
        // ---- pass 2: residual products, block sums, / (S-1) -------------------------------------------
        float ps[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) ps[c] = 0.0f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const bool live = lane * 16 + t < S;
            float r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = xs[t][c] - mean[c];
            int q = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a; b < 4; ++b, ++q) ps[q] = live ? ps[q] + r[a] * r[b] : ps[q];
        }
#pragma unroll
        for (int c = 0; c < 10; ++c) part[c * 64 + lane] = ps[c];
        wave_sync();
        POD_STAMP(i + 2048, 2);
        if (lane < 10) small[lane] = __fdiv_rn(cascade_combine(part + lane * 64, S), (float)(S - 1));
        wave_sync();
        POD_STAMP(i + 2048, 3);
#pragma unroll
        for (int c = 0; c < 10; ++c) cv[c] = small[c];
        wave_sync();
    } else {
        const Box b = decode_box(dl[0], dl[1], dl[2], dl[3], anc, P.wts);   // PI:384
        mean[0] = b.x1; mean[1] = b.y1; mean[2] = b.x2; mean[3] = b.y2;
    }

    // ---- epistemic covariance over the N runs (PI:323-331): lanes = runs ---------------------------------
    if (N > 1) {
        float e[4] = {0, 0, 0, 0};
        if (lane < N) {
            const float4 rd = *reinterpret_cast<const float4*>(run_delta + (size_t)lane * 4);
            const Box b = decode_box(rd.x, rd.y, rd.z, rd.w, anc, P.wts);
            e[0] = b.x1; e[1] = b.y1; e[2] = b.x2; e[3] = b.y2;
#pragma unroll
            for (int c = 0; c < 4; ++c) small[16 + lane * 4 + c] = e[c];
        }
        wave_sync();
        if (lane < 4) {
            float acc = 0.0f;
            for (int r = 0; r < N; ++r) acc = acc + small[16 + r * 4 + lane];
            small[lane] = __fdiv_rn(acc, (float)N);
        }
        wave_sync();
        float em[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) em[c] = small[c];
        wave_sync();
        if (lane < N)
#pragma unroll
            for (int c = 0; c < 4; ++c) small[16 + lane * 4 + c] = e[c] - em[c];
        wave_sync();
        if (lane < 10) {
            int a = 0, b = lane;   // unpack q -> (a,b), a <= b
            if (lane >= 4) { a = 1; b = lane - 3; }
            if (lane >= 7) { a = 2; b = lane - 5; }
            if (lane >= 9) { a = 3; b = 3; }
            float acc = 0.0f;
            for (int r = 0; r < N; ++r) acc = acc + small[16 + r * 4 + a] * small[16 + r * 4 + b];
            small[lane] = __fdiv_rn(acc, (float)(N - 1));
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < 10; ++c) cv[c] = cv[c] + small[c];   // PI:374 cov += epistemic
    }

    POD_STAMP(i + 2048, 4);
    if (lane == 0) {
        *reinterpret_cast<float4*>(P.boxes + (size_t)i * 4) = float4{mean[0], mean[1], mean[2], mean[3]};
        float* o = P.cov + (size_t)i * 16;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b, ++q) {
                o[a * 4 + b] = cv[q];
                o[b * 4 + a] = cv[q];
            }
    }
}

}  // namespace pod
