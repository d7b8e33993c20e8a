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
#include "pod_device.h"

namespace pod {

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
// `part[b]` holds the sum of block b (b < nblk_full) and, if S % 16 != 0, part[nblk_full] the tail.
__device__ __forceinline__ float cascade_combine(const float* part, int stride, int S) {
    const int nfull = S >> 4;
    float acc1 = 0.0f, acc2 = 0.0f;
    for (int b = 0; b < nfull; ++b) {
        acc1 = acc1 + part[b * stride];
        if (((b + 1) & 15) == 0) {
            acc2 = acc2 + acc1;
            acc1 = 0.0f;
        }
    }
    float acc0 = (S & 15) ? part[nfull * stride] : 0.0f;
    acc0 = acc0 + acc1;
    acc0 = acc0 + acc2;
    return acc0;
}

// One 64-thread workgroup (= one wavefront) per candidate: barriers are wave-local and free.
__global__ void __launch_bounds__(64) k3_decode_cov(const K3Params P) {
    __shared__ float part[64 * 10];   // [lane][component]
    __shared__ float small[16 + 4 * POD_MAX_RUNS];
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= min(*P.n_total, P.n_capacity)) return;
    const bool active = true;

    const int S = P.S, D = P.D, N = P.n_runs;
    float dl[4] = {0, 0, 0, 0};
    Box anc = {0, 0, 1, 1};
    if (active) {
        const float4 d4 = *reinterpret_cast<const float4*>(P.cand_delta + (size_t)i * 4);
        dl[0] = d4.x; dl[1] = d4.y; dl[2] = d4.z; dl[3] = d4.w;
        anc = load_box(P.cand_anchor, i);
    }
    float mean[4] = {0, 0, 0, 0};
    float cv[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) cv[c] = 0.0f;

    if (D > 0) {
        // ---- Cholesky factor (row-major lower triangle) --------------------------------------------
        float Lm[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) Lm[r][c] = 0.0f;
        if (active) {
            const float* rv = P.cand_reg_var + (size_t)i * D;
#pragma unroll
            for (int c = 0; c < 4; ++c) Lm[c][c] = sqrtf(expf(rv[c]));
            if (D == 10) {   // torch.tril_indices(4,4,-1): (1,0),(2,0),(2,1),(3,0),(3,1),(3,2)
                Lm[1][0] = rv[4]; Lm[2][0] = rv[5]; Lm[2][1] = rv[6];
                Lm[3][0] = rv[7]; Lm[3][1] = rv[8]; Lm[3][2] = rv[9];
            }
        }
        uint32_t gid = 0;
        if (active) gid = (uint32_t)(P.anchor_base[P.cand_level[i]] + P.cand_anchor_idx[i]);
        // ---- pass 1: draw, decode, block sums --------------------------------------------------------
        float xs[16][4];
        float bs[4] = {0, 0, 0, 0};
        f32x8n z;   // native mode: one Philox call serves the two samples (2m, 2m+1)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int s = lane * 16 + t;
            float e[4] = {0, 0, 0, 0};
            if (active && s < S) {
                if (P.eps_prop) {
                    const float4 e4 = *reinterpret_cast<const float4*>(P.eps_prop + ((size_t)s * P.n_replay + i) * 4);
                    e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
                } else {
                    if ((t & 1) == 0) z = philox_normals8(P.seed, gid, (uint32_t)(s >> 1), 0u, STREAM_BOX);
                    e[0] = z.v[(t & 1) * 4 + 0]; e[1] = z.v[(t & 1) * 4 + 1]; e[2] = z.v[(t & 1) * 4 + 2]; e[3] = z.v[(t & 1) * 4 + 3];
                }
            }
            float d[4];
            if (D == 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = dl[c] + Lm[c][c] * e[c];   // L eps exact for diagonal L
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float acc = Lm[c][0] * e[0];
#pragma unroll
                    for (int k = 1; k < 4; ++k) acc = fmaf(Lm[c][k], e[k], acc);
                    d[c] = dl[c] + acc;
                }
            }
            const Box b = decode_box(d[0], d[1], d[2], d[3], anc, P.wts);
            const bool live = s < S;
            xs[t][0] = b.x1; xs[t][1] = b.y1; xs[t][2] = b.x2; xs[t][3] = b.y2;
#pragma unroll
            for (int c = 0; c < 4; ++c) bs[c] = live ? bs[c] + xs[t][c] : bs[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) part[lane * 10 + c] = bs[c];
        __syncthreads();
        if (lane < 4) small[lane] = __fdiv_rn(cascade_combine(part + lane, 10, S), (float)S);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) mean[c] = small[c];
        __syncthreads();
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
        for (int c = 0; c < 10; ++c) part[lane * 10 + c] = ps[c];
        __syncthreads();
        if (lane < 10) small[lane] = __fdiv_rn(cascade_combine(part + lane, 10, S), (float)(S - 1));
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 10; ++c) cv[c] = small[c];
        __syncthreads();
    } else if (active) {
        const Box b = decode_box(dl[0], dl[1], dl[2], dl[3], anc, P.wts);   // PI:384
        mean[0] = b.x1; mean[1] = b.y1; mean[2] = b.x2; mean[3] = b.y2;
    }

    // ---- epistemic covariance over the N runs (PI:323-331): lanes = runs ---------------------------------
    if (N > 1) {
        float e[4] = {0, 0, 0, 0};
        if (active && lane < N) {
            const float4 rd = *reinterpret_cast<const float4*>(P.cand_run_delta + ((size_t)i * N + lane) * 4);
            const Box b = decode_box(rd.x, rd.y, rd.z, rd.w, anc, P.wts);
            e[0] = b.x1; e[1] = b.y1; e[2] = b.x2; e[3] = b.y2;
#pragma unroll
            for (int c = 0; c < 4; ++c) small[16 + lane * 4 + c] = e[c];
        }
        __syncthreads();
        if (lane < 4) {
            float acc = 0.0f;
            for (int r = 0; r < N; ++r) acc = acc + small[16 + r * 4 + lane];
            small[lane] = __fdiv_rn(acc, (float)N);
        }
        __syncthreads();
        float em[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) em[c] = small[c];
        __syncthreads();
        if (lane < N)
#pragma unroll
            for (int c = 0; c < 4; ++c) small[16 + lane * 4 + c] = e[c] - em[c];
        __syncthreads();
        if (lane < 10) {
            int a = 0, b = lane;   // unpack q -> (a,b), a <= b
            if (lane >= 4) { a = 1; b = lane - 3; }
            if (lane >= 7) { a = 2; b = lane - 5; }
            if (lane >= 9) { a = 3; b = 3; }
            float acc = 0.0f;
            for (int r = 0; r < N; ++r) acc = acc + small[16 + r * 4 + a] * small[16 + r * 4 + b];
            small[lane] = __fdiv_rn(acc, (float)(N - 1));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 10; ++c) cv[c] = cv[c] + small[c];   // PI:374 cov += epistemic
    }

    if (active && lane == 0) {
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
