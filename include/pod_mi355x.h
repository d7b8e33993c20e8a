/*
 * pod_mi355x.h -- C ABI of the MI355X-native probabilistic-inference hot path.
 *
 * Drop-in boundary for asharakeh/pod_compare's `probabilistic_inference` plugin
 * (reference files, relative to /root/reference/src):
 *   PI = probabilistic_inference/probabilistic_inference.py
 *   IU = probabilistic_inference/inference_utils.py
 *   MU = probabilistic_modeling/modeling_utils.py
 *   PR = probabilistic_modeling/probabilistic_retinanet.py
 *
 * Conventions
 *   - every pointer marked "dev" is a DEVICE pointer (HBM); everything else is host memory;
 *   - all floating point is fp32, anchor indices int32, class ids int32 (the Python
 *     boundary widens them to the reference's int64);
 *   - outputs and scratch are caller-allocated; the library never allocates device memory,
 *     never synchronises the device and holds no global state; distinct streams may be used
 *     concurrently from distinct threads;
 *   - every entry point takes the `hipStream_t` to launch on (passed as `void*` so that this
 *     header needs no HIP include), is asynchronous and hipGraph-capturable, and returns
 *     0 on success or a negative POD_E_* code (it never throws across the ABI);
 *   - counts that are only known on the device (number of candidates, detections ...) live in
 *     device int32 words; kernels are launched for the worst case and exit early.
 *
 * HBM data layout
 *   dense head tensors, per FPN level l (what the conv head writes, NCHW, PR:486-537):
 *       cls, cls_var : (n_runs, A*K, H_l, W_l)      delta : (n_runs, A*4, H_l, W_l)
 *       reg_var      : (n_runs, A*D, H_l, W_l), D = 4 (diagonal) or 10 (full), MU:4-22
 *     run r of a tensor starts `run_stride` elements after run r-1 (MC-dropout runs batched
 *     on the batch dim, or ensemble members stacked / gathered over RCCL).
 *   reference anchor index inside a level (permute_to_N_HWA_K, PR:343-349):
 *       r = (h*W + w)*A + a,   channel of class k = a*K + k.
 *   "plane layout" outputs keep the input's (A*C, H, W) order (coalesced, no transpose).
 */
#ifndef POD_MI355X_H
#define POD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POD_ABI_VERSION 14
#define POD_MAX_LEVELS 8
#define POD_MAX_CLASSES 16       /* K: BDD = 7 (Base-BDD-RetinaNet.yaml:11-12) */
#define POD_MAX_RUNS 64          /* MC-dropout runs / ensemble members */
#define POD_MAX_TOPK 2048        /* model.test_topk_candidates (detectron2 default 1000, PI:300) */
#define POD_MAX_PROP_SAMPLES 1024 /* PI:355 hard-codes 1000 */
#define POD_MAX_CLS_SAMPLES 64   /* CLS_VAR_LOSS.NUM_SAMPLES (10 in the reg_cls_var yamls) */
#define POD_MAX_CANDIDATES 8192  /* n = sum over levels of kept top-k; 5 x 1000 at BASELINE */
#define POD_MAX_DETECTIONS 128   /* model.max_detections_per_image (100) */

#define POD_OK 0
#define POD_E_INVALID (-1)       /* bad argument / unsupported size */
#define POD_E_LAUNCH (-2)        /* HIP launch error (see hipGetLastError) */

typedef void* pod_stream_t;      /* hipStream_t */

/* One FPN level of the dense head output (device pointers, NCHW). */
typedef struct PodLevel {
    const float* cls;            /* dev (n_runs, A*K, H, W) logits               PR:352-361 'box_cls'     */
    const float* cls_var;        /* dev (n_runs, A*K, H, W) log-variances or NULL            'box_cls_var' */
    const float* delta;          /* dev (n_runs, A*4, H, W)                                   'box_delta'   */
    const float* reg_var;        /* dev (n_runs, A*D, H, W) or NULL                           'box_reg_var' */
    const float* eps_cls;        /* dev (cls_samples, H*W*A, K) replayed normals in REFERENCE layout, or NULL
                                    (NULL = in-kernel Philox4x32-10; replay is the parity mode)            */
    int64_t run_stride_cls;      /* elements between consecutive runs of cls / cls_var */
    int64_t run_stride_delta;
    int64_t run_stride_reg;
    int32_t H, W;
    int32_t anchor_base;         /* index of this level's first anchor in the level-concatenated order */
    int32_t reserved;
} PodLevel;

/* Static description of the path (model attributes + config keys). */
typedef struct PodConfig {
    int32_t n_levels;            /* <= POD_MAX_LEVELS */
    int32_t n_runs;              /* N >= 1; > 1 merges runs (PI:211-270) and adds epistemic covariance */
    int32_t num_anchors;         /* A (9) */
    int32_t num_classes;         /* K (7) */
    int32_t cov_dims;            /* D: 0 (no reg_var head), 4 or 10 */
    int32_t has_cls_var;         /* 0/1 */
    int32_t merge_quirk;         /* 1 = reference behaviour (2*x0+x1+..+x_{n-2})/n, PI:216-222; 0 = true mean */
    int32_t cls_samples;         /* model.cls_var_num_samples, PI:294 */
    int32_t prop_samples;        /* 1000, PI:355 */
    int32_t topk;                /* model.test_topk_candidates, PI:300 */
    int32_t max_detections;      /* model.max_detections_per_image */
    float score_thresh;          /* model.test_score_thresh, PI:304 */
    float nms_thresh;            /* model.test_nms_thresh */
    float affinity_thresh;       /* PROBABILISTIC_INFERENCE.AFFINITY_THRESHOLD */
    float box_weights[4];        /* Box2BoxTransform weights (cfg.MODEL.RPN.BBOX_REG_WEIGHTS, PI:175-176) */
    uint64_t philox_seed;        /* native-RNG mode only */
} PodConfig;

int pod_abi_version(void);

/* ---- K1  mc_merge_score -------------------------------------------------------------------
 * Replaces: the dense MC-dropout / ensemble merge PI:211-270, the classification sampling
 * PI:289-297 and the max-over-classes + score-threshold test PI:301-304.
 * Streams every run of every level once (coalesced 16-byte loads along W), writes the merged
 * tensors in plane layout (skipped when n_runs == 1 or the pointer is NULL) and appends one
 * 64-bit key per anchor whose score exceeds `score_thresh` to its level's candidate list:
 *     key = (float_bits(score) << 32) | (0xFFFFFFFF - r)      (descending key = score desc, r asc)
 * mean_* : dev, level-concatenated plane layout: level l starts at anchor_base_l * C elements.
 * cand_keys : dev uint64[R_total], level l's list starts at anchor_base_l.
 * cand_count: dev int32[2 * n_levels] (counts, then pod_level_topk's tickets), MUST be zero on entry
 *             (pod_reset_counters once after allocation; pod_gather_candidates / pod_gather_decode leave it zeroed again).
 * HBM-bound; algorithmic bytes per image = 4 * R * (2K + 4 + D) * (N + 1)  (SURVEY 8d). */
int pod_mc_merge_score(const PodConfig* cfg, const PodLevel* levels,
                       float* mean_cls, float* mean_cls_var, float* mean_delta, float* mean_reg_var,
                       uint64_t* cand_keys, int32_t* cand_count,
                       uint64_t* maybe_bits, pod_stream_t stream);

/* ---- K1b score_maybe ------------------------------------------------------------------------
 * Sparse companion of K1's prune mode (native RNG + cls_var head): when `maybe_bits`
 * (dev uint64[pod_maybe_words()] = one word per (anchor shape, class, 64 cells), all zero on entry; pod_score_maybe leaves it
 * zeroed again) is passed to pod_mc_merge_score, the dense pass
 * draws no samples; it writes a bitmap of the anchors that can still reach `score_thresh` under the sampler's
 * hard bound |eps| < 4.9 (an exact superset of the candidates) and this kernel evaluates
 * mean_s sigmoid(logit + eps_s*sigma) (PI:289-295) for those only, appending the keys of the anchors
 * above the threshold to `cand_keys` / `cand_count` exactly as K1 does otherwise. */
int64_t pod_maybe_words(const PodConfig* cfg, const PodLevel* levels);
/* probs_dense : dev float (R_total, K) in level-concatenated anchor order, or NULL: the K class probabilities of every anchor
 *               emitted here are stored at its row, so that the gather kernel does not evaluate them a second time. */
int pod_score_maybe(const PodConfig* cfg, const PodLevel* levels, const float* mean_cls, const float* mean_cls_var,
                    uint64_t* maybe_bits, uint64_t* cand_keys, int32_t* cand_count, float* probs_dense, pod_stream_t stream);

/* ---- K1f merge_score_fused (round 4): pod_mc_merge_score's prune mode + pod_score_maybe in ONE streaming launch ---------------
 * Replaces: PI:211-270 for box_cls / box_cls_var, PI:289-297, PI:301, PI:304 -- the job SURVEY 8 a3 + a4 defines ("merge and score").
 * Native draws only (every levels[l].eps_cls NULL: the prune bound needs the sampler's |eps| < 4.9); with or without a variance head.
 * A lane keeps all K classes of its 4 cells in registers through the run loop; cells that may pass are parked in LDS and scored by the
 * whole workgroup exactly as pod_score_maybe scores them: identical candidate keys (any order inside a level), identical probs_dense rows.
 * mean_cls / mean_cls_var : merged class planes as pod_mc_merge_score writes them, or both NULL: not stored (nothing downstream of the
 *                           product path reads them; the gather kernel merges box_delta / box_reg_var at the candidates).
 * cand_keys / cand_count / probs_dense : as for pod_mc_merge_score / pod_score_maybe (cand_count zero on entry).
 * HBM-bound; algorithmic bytes per image = 4 * R * 2K * (N - 1) read (quirk on) [+ 4 * R * 2K written when the planes are asked for]. */
int pod_merge_score_fused(const PodConfig* cfg, const PodLevel* levels, float* mean_cls, float* mean_cls_var, uint64_t* cand_keys,
                          int32_t* cand_count, float* probs_dense, pod_stream_t stream);

/* Zeroes `n` int32 device words on the stream (graph-capturable memset node). */
int pod_reset_counters(int32_t* counters, int32_t n, pod_stream_t stream);

/* ---- K2  level_topk ------------------------------------------------------------------------
 * Replaces: `predicted_prob.topk(num_topk)` + `> test_score_thresh` filter PI:300-308, per level.
 * Exact top-`topk` of each level's candidate list, sorted by descending key (ties: lower anchor
 * index first).  A level with at most POD_MAX_TOPK candidates: one workgroup, LDS bitonic sort.
 * A bigger level: 16 workgroups select the top-k of one slice each (MSB radix select + sort), the
 * last one to finish selects the final top-k from their survivors (no spinning).
 * sel_keys : dev uint64[n_levels * topk];  sel_count : dev int32[n_levels] (written).
 * cat_keys / cat_level / n_total (all three or none): the same selection level-concatenated -- row offset_l + i of cat_keys
 *   = sel_keys[l][i], cat_level its level, n_total = number of rows -- dev uint64 / int32 [n_levels * topk] / int32.  This is
 *   what the gather kernels read: one workgroup per row finds its key, its level and the row count with three independent loads.
 * cand_keys is CONSUMED (big levels are compacted in place); cand_count is only READ here -- pod_gather_candidates /
 * pod_gather_decode consume it (leave it ZERO, ready for the next image's pod_mc_merge_score; no pod_reset_counters between
 * images).  cand_count : dev int32[2 * n_levels]: the counts, then n_levels ticket words (zero on entry, left zero). */
int pod_level_topk(const PodConfig* cfg, const PodLevel* levels, uint64_t* cand_keys,
                   int32_t* cand_count, uint64_t* sel_keys, int32_t* sel_count,
                   uint64_t* cat_keys, int32_t* cat_level, int32_t* n_total, pod_stream_t stream);

/* ---- K2b gather_candidates -----------------------------------------------------------------
 * Replaces: the index gathers PI:305-338 and the level concatenation PI:341-342, 387-388.
 * Candidate i of the level-concatenated list gets: anchor index inside its level, level id,
 * score, class, K probabilities (recomputed bit-identically to K1), merged delta, merged
 * reg_var (D values), its anchor box and every run's raw delta (for the epistemic term).
 * anchors : dev (R_total, 4) level-concatenated XYXY anchors (PR:101).
 * cat_keys / cat_level / n_total : pod_level_topk's level-concatenated selection (read).
 * cand_count : the counters pod_mc_merge_score / pod_score_maybe appended with; ZEROED here (consumed).
 * probs_dense : the array pod_score_maybe filled (native draws + variance head), or NULL = evaluate the probabilities here. */
int pod_gather_candidates(const PodConfig* cfg, const PodLevel* levels, const float* anchors,
                          const uint64_t* cat_keys, const int32_t* cat_level, const int32_t* n_total,
                          int32_t* cand_count, const float* probs_dense,
                          int32_t* cand_anchor_idx, int32_t* cand_level, float* cand_score, int32_t* cand_class,
                          float* cand_probs, float* cand_delta, float* cand_reg_var, float* cand_anchor,
                          float* cand_run_delta /* (n, n_runs, 4) or NULL when n_runs == 1 */,
                          pod_stream_t stream);

/* ---- K3  decode_cov ------------------------------------------------------------------------
 * Replaces: covariance_output_to_cholesky MU:4-22, the 1000-sample propagation PI:344-368
 * (MVN rsample, SampleBox2BoxTransform.apply_samples_deltas IU:510-547,
 * compute_mean_covariance_torch IU:337-371), the epistemic covariance PI:323-331 (+= PI:369-374)
 * and the deterministic decode PI:375-385.  One wavefront per candidate; samples live in
 * registers; two-pass moments with the CPU reference's 16-row block summation order.
 * eps_prop : dev (prop_samples, n_replay, 4) replayed normals (reference layout) or NULL (Philox).
 * boxes : dev (n,4)   cov : dev (n,4,4) (zeros when the model has neither reg_var nor runs). */
int pod_decode_cov(const PodConfig* cfg, const PodLevel* levels, const int32_t* n_total, int32_t n_capacity,
                   const float* cand_delta, const float* cand_reg_var, const float* cand_anchor,
                   const float* cand_run_delta, const int32_t* cand_anchor_idx, const int32_t* cand_level,
                   const float* eps_prop, int32_t n_replay,
                   float* boxes, float* cov, pod_stream_t stream);

/* ---- K2b + K3 in one launch (in-kernel draws only) ------------------------------------------------
 * pod_gather_candidates followed by pod_decode_cov, same arguments, same outputs (the candidate arrays are still written
 * for the later kernels), bit-identical results: one 256-thread workgroup per candidate -- wavefront 0 gathers, all four draw
 * and decode the 1000 samples into LDS, wavefront 0 forms the moments in the reference's summation order; merged deltas /
 * log-variances / per-run deltas never leave LDS.  n_capacity = n_levels * topk. */
int pod_gather_decode(const PodConfig* cfg, const PodLevel* levels, const float* anchors, const uint64_t* cat_keys,
                      const int32_t* cat_level, const int32_t* n_total, int32_t* cand_count, const float* probs_dense,
                      int32_t* cand_anchor_idx, int32_t* cand_level, float* cand_score,
                      int32_t* cand_class, float* cand_probs, float* cand_delta, float* cand_reg_var, float* cand_anchor,
                      float* cand_run_delta, float* boxes, float* cov, pod_stream_t stream);

/* ---- K4  nms_cluster -----------------------------------------------------------------------
 * Replaces: detectron2 batched_nms -> torchvision coordinate-trick NMS (call sites PI:554-560,
 * IU:31-36, IU:83-89): boxes + class*(max_coord+1), stable descending-score order, suppress
 * iff IoU > nms_thresh, first `max_detections` survivors.
 * classes : dev int32[n] in [0, cfg->num_classes) (other ids are honoured, on a slower single-workgroup route).
 * keep : dev int32[max_detections] candidate indices in keep order;  n_keep : dev int32 (written).
 * scratch : dev, 16-byte aligned, pod_nms_scratch_bytes(n_capacity) bytes, ZEROED ONCE after allocation and then owned
 *           by this entry point (per-class survivor lists between the two launches; a call-generation word and the
 *           survivor scores the class sweeps publish to each other so that a class stops as soon as max_detections
 *           higher-scoring survivors exist overall).  boxes must be 16-byte aligned. */
size_t pod_nms_scratch_bytes(int32_t n_capacity);
int pod_nms_cluster(const PodConfig* cfg, const int32_t* n_total, int32_t n_capacity,
                    const float* boxes, const float* scores, const int32_t* classes,
                    int32_t* keep, int32_t* n_keep, void* scratch, pod_stream_t stream);

/* ---- K5  bayes_fuse ------------------------------------------------------------------------
 * Replaces: post_processing_bayes_od PI:562-636 + bounding_box_bayesian_inference IU:292-334.
 * One workgroup per kept centre: members = {j : IoU(centre, j) > affinity (pairwise_iou, Q8:
 * only the <=100 needed rows) and argmax(probs_j) == argmax(probs_centre)}; 4x4 SPD inverses
 * in fp64 registers, wavefront reductions of the precisions.
 * box_mode: 0 = bayesian_inference, 1 = covariance_intersection;
 * cls_mode: 0 = max_score (centre's score/class/probs), 1 = bayesian_inference (mean of member probs).
 * Degenerate cluster (no member, Q12): falls back to the centre's box/covariance.
 * out_* : dev, max_detections rows.
 * CAPACITIES: boxes / cov / scores / classes / probs must hold cfg->n_levels * cfg->topk rows and keep cfg->max_detections
 * entries, whatever *n_total / *n_keep say: the first rows are loaded in the same round trip as the counts (rows past the
 * counts are read, never used). */
int pod_bayes_fuse(const PodConfig* cfg, const int32_t* n_total, const int32_t* keep, const int32_t* n_keep,
                   const float* boxes, const float* cov, const float* scores, const int32_t* classes,
                   const float* probs, int32_t box_mode, int32_t cls_mode,
                   float* out_boxes, float* out_cov, float* out_scores, int32_t* out_classes, float* out_probs,
                   pod_stream_t stream);

/* ---- K6  anchor_stats_merge ----------------------------------------------------------------
 * Replaces: general_anchor_statistics_postprocessing IU:91-154 (cluster mean, residual outer
 * products / max(m-1,1), + mean member covariance when the net provides one, mean prob vector,
 * singleton rule IU:127-133, score/class re-derived from the merged prob vector IU:146-152).
 * cov may be NULL (no network covariance).
 * CAPACITIES as for pod_bayes_fuse: candidate arrays of cfg->n_levels * cfg->topk rows, keep of cfg->max_detections entries
 * (speculative loads run ahead of the counts). */
int pod_anchor_stats_merge(const PodConfig* cfg, const int32_t* n_total, const int32_t* keep, const int32_t* n_keep,
                           const float* boxes, const float* cov, const int32_t* classes, const float* probs,
                           float* out_boxes, float* out_cov, float* out_scores, int32_t* out_classes, float* out_probs,
                           pod_stream_t stream);

/* ---- post-NMS ensemble merge ---------------------------------------------------
 * Replaces: general_black_box_ensembles_post_processing IU:165-289 (callers PI:444-481 MC-dropout post-NMS,
 * PI:506-534 ensembles post-NMS).
 * pod_ensemble_append: appends rows keep[0:n_keep) of one member's standard-NMS result (IU:42-53) to the
 *   concatenated member arrays (torch.cat IU:191-196); `total` (dev int32, zero before the first member) is the
 *   running row count; cov == NULL appends zeros.
 * pod_ensemble_merge: sequential same-class clustering IU:203-215 (greedy sweep in index order, IoU >= affinity),
 *   then per-cluster mean box, residual covariance / (m-1) + mean member covariance, mean prob vector IU:223-247
 *   and score/class = max of the merged prob vector IU:262-263.  seeds/n_seeds: dev int32[capacity] / int32.
 *   out_*: dev, `capacity` rows (one per cluster).  The second NMS (IU:269-274) is pod_nms_cluster on the output
 *   with n_total = n_seeds, then pod_finalize gathers through its keep list. */
int pod_ensemble_append(const PodConfig* cfg, const int32_t* keep, const int32_t* n_keep, const float* boxes, const float* cov,
                        const int32_t* classes, const float* probs, int32_t capacity,
                        float* dst_boxes, float* dst_cov, int32_t* dst_classes, float* dst_probs, int32_t* total,
                        pod_stream_t stream);
int pod_ensemble_merge(const PodConfig* cfg, const int32_t* m_total, int32_t capacity, const float* boxes, const float* cov,
                       const int32_t* classes, const float* probs, int32_t* seeds, int32_t* n_seeds,
                       float* out_boxes, float* out_cov, float* out_scores, int32_t* out_classes, float* out_probs,
                       pod_stream_t stream);

/* ---- K7  finalize --------------------------------------------------------------------------
 * Replaces: the keep-gather of general_standard_nms_postprocessing IU:42-53 (when `keep` is
 * non-NULL rows are gathered through it; cov == NULL gives the zeros of IU:52-53) and
 * probabilistic_detector_postprocess IU:374-425 (scale, clip, drop empty boxes, cov + 1e-4 I,
 * S cov S^T).  Also emits, per detection, the XYWH box and T cov T^T of instances_to_json /
 * covar_xyxy_to_xywh IU:428-502 as a fixed-stride record:
 *     records[i] = { x, y, w, h, score, class, probs[K], cov_xywh[16] }   (6 + K + 16 floats)
 * det_* : dev, max_detections rows;  n_det : dev int32 (written). */
int pod_finalize(const PodConfig* cfg, const int32_t* keep, const int32_t* n_rows,
                 const float* boxes, const float* cov, const float* scores, const int32_t* classes,
                 const float* probs, float scale_x, float scale_y, float out_h, float out_w,
                 float* det_boxes, float* det_cov, float* det_scores, int32_t* det_classes, float* det_probs,
                 float* records, int32_t* n_det, pod_stream_t stream);

typedef struct PodDetections {     /* pod_finalize's outputs */
    float* boxes;  float* cov;  float* scores;  int32_t* classes;  float* probs;  float* records;  int32_t* n_det;
} PodDetections;

/* ---- conv-net side: fused ReLU + dropout -------------------------------------------------------
 * Replaces: the `nn.ReLU(), nn.Dropout(p)` pair after every 3x3 conv of the head subnets (PR:403-424) in
 * MC-dropout mode (PR:103-108), in place, one pass.  x: dev fp32, 16-byte aligned, n elements.
 * Element e uses 16 bits of the Philox4x32-10 call with counter (offset + e/8): keep iff field >= p * 2^16; pass a different
 * `offset` (or seed) per call. */
int pod_relu_dropout(float* x, int64_t n, float p, uint64_t seed, uint64_t offset, pod_stream_t stream);

/* ---- conv-net side: bias + residual + ReLU + dropout in one pass ------------------------------------
 * Replaces: the element-wise tail of every convolution of the model (PR:403-424 head subnets; detectron2's
 * ResNet bottleneck `relu(conv3(x) + shortcut(x))` and FrozenBN-folded `relu(conv(x) + b)` of the backbone):
 * x (N, C, H, W) fp32, in place:  x = dropout(relu((x + bias[c]) + (residual + res_bias[c])), p).
 * torch runs the conv bias as a separate add, then clamp, then dropout (2-4 passes over the activation); the
 * conv is called without bias and this is the only pass.  bias / residual / res_bias may be NULL, relu 0/1,
 * p = 0 disables dropout; Philox counters as pod_relu_dropout.  n = N*C*H*W, HW = H*W.
 * A channels-last (NHWC) activation is the same call with HW = 1 (channel = element index mod C). */
int pod_bias_act(float* x, const float* bias, const float* residual, const float* res_bias, int64_t n, int32_t C,
                 int64_t HW, int32_t relu, float p, uint64_t seed, uint64_t offset, pod_stream_t stream);

/* Same tail for a channels-last (N, H*W, C) conv output, written as NCHW planes: dst = dropout(relu(src + bias[c]), p)
 * transposed through LDS tiles, so leaving the channels-last trunk costs no extra pass.  C % 4 == 0, HW % 4 == 0,
 * src != dst; dropout counters are those of pod_bias_act on the NCHW result. */
int pod_bias_act_to_nchw(const float* src, float* dst, const float* bias, int64_t N, int32_t C, int64_t HW, int32_t relu,
                         float p, uint64_t seed, uint64_t offset, pod_stream_t stream);
/* The reverse layout change, for an NCHW conv (MIOpen) whose consumer is pod_wino_conv3x3: dst[(n*HW + hw)*C + c] =
 * act(src[(n*C + c)*HW + hw] + bias[c]), the bias + ReLU pass of detectron2's bottleneck conv1 (the reference model's
 * backbone, probabilistic_retinanet.py:20-60 via build_retinanet_resnet_fpn_backbone) writing channels-last.  C % 4 == 0. */
int pod_bias_act_to_nhwc(const float* src, float* dst, const float* bias, int64_t N, int32_t C, int64_t HW, int32_t relu,
                         pod_stream_t stream);

/* ---- conv-net side: broadcast + dropout -------------------------------------------------------------
 * Replaces: feeding the SAME first-conv activation to every MC run's `nn.Dropout(p)` (PR:104-106 replicates the feature
 * lists N times; PR:403-424).  dst[c][i] = dropout(src[i], p), c < copies, independent masks; n % 4 == 0, flat arrays
 * (any memory format shared by src and each copy).  * epoch (pod_expand_dropout, pod_wino_conv3x3, pod_wino_conv3x3_split; ABI 7): NULL, or a device word: the Philox key of the masks is then
 * seed ^ (*epoch * 0x9E3779B97F4A7C15).  seed and offset are launch arguments -- constants of a launch captured in a HIP graph -- so a
 * replayed MC-dropout forward bumps this word (one captured add at the start of the forward) to draw fresh masks for every image. */
int pod_expand_dropout(const float* src, float* dst, int64_t n, int32_t copies, float p, uint64_t seed, uint64_t offset,
                       const uint64_t* epoch, pod_stream_t stream);

/* ---- conv-net side: the head subnets' 3x3 convolutions -------------------------------------------------
 * Replaces: the `nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(), nn.Dropout(p)` triples of cls_subnet / bbox_subnet
 * (probabilistic_retinanet.py:403-427) and their evaluation "for every MC run, for every FPN level" (PR:95-108, detectron2
 * RetinaNetHead.forward's loop over features): ONE launch covers all levels and all runs.  fp32 Winograd (F(2,3) down the rows, F(4,3) along
 * the columns: 24 multiply-adds per 2x4 outputs where the direct form has 72) on the fp32 matrix cores, bias + ReLU + dropout fused into the store.
 *
 * Activations are channels-last: in[pixel][C], out[pixel][K]; all images of the launch live in the two buffers.  `blocks`
 * (device, 16-byte aligned) holds n_blocks int32x4 records {first pixel of image 0 in `in`, first pixel of image 0 in `out`,
 * grid_cols << 24 | H << 12 | W, n_images << 24 | block_row << 12 | block_col}, one per 16x16-pixel block of a CANVAS of n_images
 * (1..127) consecutive H x W images (H, W < 4096) standing in a grid of grid_cols (1..255) columns: image i occupies canvas rows
 * (i / grid_cols) * (H + 1) .. + H - 1 and columns (i % grid_cols) * (W + 1) .. + W - 1 -- one zero row / column between neighbours,
 * the padding of both; block (r, c) covers canvas rows 16r.. and columns 16c.. and may lie across image borders (n_images = 1: the
 * plain ceil(H/16) x ceil(W/16) tiling of one image).  Blocks that touch no image need not be listed.
 * C % 8 == 0, K in {64, 128, 256, 512}.  U = pod_wino_filter_transform(weight): 24 * round_up(K, 64) * C floats; weight is
 * (K, C, 3, 3), output channels past K are zero (so a K = 63 predictor runs as K = 64; bias then has round_up(K, 64) entries).
 * k_planes == 0: out is channels-last.  k_planes > 0 (the predictor convs cls_score / bbox_pred / cls_var / bbox_cov,
 * PR:430-484): out is NCHW, image = k_planes planes of H*W starting at float `k_planes * first pixel`: the (N, A*K, H, W)
 * tensors pod_mc_merge_score streams; p must be 0.  Dropout: 16 Philox4x32-10 bits per element, keep iff field >= p * 2^16, counter
 * = offset + (flat index of the output element >> 3): the mask pod_bias_act draws on the same tensor. */
int pod_wino_filter_transform(const float* weight, float* U, int32_t K, int32_t C, pod_stream_t stream);
int pod_wino_conv3x3(const float* in, float* out, const float* U, const float* bias, const int32_t* blocks, int32_t n_blocks,
                     int32_t C, int32_t K, int32_t k_planes, int32_t relu, float p, uint64_t seed, uint64_t offset,
                     const uint64_t* epoch, pod_stream_t stream);

/* ---- operand abs-max records (round 5; ABI 12) ----------------------------------------------------------------------------------
 * The split convolutions below form every fp32 product from TWO F16 terms per operand; f16 has fp32's precision budget here (11 + 1 + 11
 * bits, pod_wino.h) but not its range, so every operand tensor is scaled by a power of two derived from its abs-max.  Filters: static,
 * inside the *_filter_split transforms.  Activations: a device RECORD `in_amax` of POD_AMAX_FLOATS floats whose largest element is
 * >= max |x| over what the launch reads (only elements k * POD_AMAX_STRIDE, k < POD_AMAX_SLOTS, are read or written: producers spread
 * their atomics over 16 cache lines -- thousands of same-address atomics serialise in the L2) -- an upper bound is
 * enough (a looser bound costs low-order bits of tiny values only; a bound BELOW the true maximum overflows f16: the outputs are inf /
 * nan, never silently wrong).  Producers publish it: every pod_* convolution takes `out_amax` (NULL, or a device record it max'es
 * atomically with |every value it stores| -- the caller zeroes the record before the launch); pod_absmax computes it for any other tensor
 * (the same max'ing: zero the record first; x may itself be a record: that is how two bounds are joined).  Non-finite values are ignored by the max (they make the consumer's products inf / nan as
 * they would in fp32). */
#define POD_AMAX_SLOTS 16
#define POD_AMAX_STRIDE 32
#define POD_AMAX_FLOATS (POD_AMAX_SLOTS * POD_AMAX_STRIDE)
int pod_absmax(const float* x, int64_t n, float* amax, pod_stream_t stream);

/* The same convolution with every fp32 product formed on the 16-BIT matrix cores (csrc/k12_wino_conv_split.hip).  Rounds 3-4: exact 3-way
 * bf16 splits, six partial products.  Round 5: 2-way F16 splits of the power-of-two-scaled operands, THREE partial products
 * (v_mfma_f32_32x32x16_f16, fp32 accumulate) -- half the matrix instructions, and measured against fp64 on the matrix cores a smaller
 * error than both the bf16 form and the fp32 MFMA (the fp32 accumulation chain is half as long; tools/f16_split_numerics.hip).
 * Us = pod_wino_filter_transform_split(weight): pod_wino_filter_split_bytes(K, C) bytes (2 * 24 * round_up(K, 64) * C f16 terms + a 16-byte
 * trailer holding the transformed filter's abs-max), 16-byte aligned.  C % 16 == 0.
 *
 * ONE entry for every form of the launch (rounds 3-4 exported pod_wino_conv3x3_split{,_replicas,_grouped,_partial}; they could not carry
 * the abs-max words and are gone -- ABI 12):
 *   - n_sets = 1..4 convolutions of ONE shape (C, K, relu, p, seed, epoch shared) in one grid: the cls- and the bbox-subnet layer l of the
 *     head (PR:403-427), their first layers with the replicas, the four predictors (PR:430-484).  The kernel runs one workgroup per CU,
 *     so a launch costs whole rounds of 256 workgroups: two launches of 1.5 rounds cost 4, one of 3.0 costs 3.  `blocks` = the sets' tables
 *     concatenated; set s owns blocks [sets[s].first_block, sets[s + 1].first_block) (first_block of set 0 = 0) and its records stay
 *     relative to ITS in / out.  Bit for bit the n_sets separate launches.
 *   - sets[s].replicas = r >= 1: the first conv of an MC-dropout subnet and the replication behind it (PR:403-427's first Conv2d + ReLU +
 *     Dropout under PR:95-108's run loop): every MC run sees the same features, so the conv is evaluated once and the store pass writes
 *     the runs' r (<= 127) dropout-masked copies -- replica i = image i of the record's output canvas (table: one input image per record,
 *     r output images).  Bit for bit p = 0 followed per level by pod_expand_dropout(copies = r, p, seed, offset + (index of the level's
 *     first output float) / 8, epoch).
 *   - sets[s].k_planes > 0: NCHW planes (the predictor convs); then p = 0 and no replicas.
 *   - n_splits > 1 (small maps; one set): a res5 convolution of the backbone is 48 workgroups of 32 chunks for 256 CUs.  The INPUT channels
 *     are cut into n_splits ranges of whole 32-channel super-chunks, one workgroup set each (grid.y): sets[0].out receives n_splits
 *     channels-last (out_pixels, K) arrays of partial sums (no bias / ReLU / dropout), split_stride floats apart; pod_wino_reduce /
 *     pod_reduce_partials add them in a fixed order. */
/* PodWinoConv.form: 0 or POD_WINO_FORM_4 = the shipped kernel (csrc/k12_wino_conv_split.hip: four wavefronts, one per SIMD, six Winograd
 * positions each).  POD_WINO_FORM_8 exists in round 6's EXPERIMENT builds only (tools/experiments/k16_wino_conv_split8.hip: eight wavefronts,
 * two per SIMD sharing a row of the position grid; bit-identical results, measured 5 - 8 % slower on every launch shape:
 * profiles/r06_k16_two_wavefronts_per_simd.md); the shipped library answers POD_E_INVALID for it. */
#define POD_WINO_FORM_4 4
#define POD_WINO_FORM_8 8
typedef struct PodConvSet {
    const float* in;        /* channels-last activations [pixel][C] */
    float* out;             /* channels-last [pixel][K], NCHW planes (k_planes > 0) or partial sums (n_splits > 1) */
    const void* Us;         /* pod_wino_filter_transform_split */
    const float* bias;      /* K floats (zero-padded) or NULL */
    const float* in_amax;   /* device record (POD_AMAX_FLOATS floats) bounding max |in| (see above); required */
    float* out_amax;        /* NULL, or a zeroed device record: max'ed with |every value stored| */
    uint64_t offset;        /* Philox offset of this set's dropout masks */
    int32_t first_block;    /* first record of the set in `blocks` */
    int32_t replicas;       /* 0: an ordinary launch; r >= 1: r masked replicas per input image */
    int32_t k_planes;       /* 0: channels-last out; > 0: NCHW planes of k_planes real channels */
    int32_t reserved;
} PodConvSet;
typedef struct PodWinoConv {
    const int32_t* blocks;  /* device, 16-byte aligned int32x4 records (pod_wino_conv3x3) */
    int32_t n_blocks, n_sets;
    int32_t C, K, relu;
    float p;                /* dropout rate of the store pass, [0, 1) */
    uint64_t seed;
    const uint64_t* epoch;  /* NULL or the device word folded into the Philox key (pod_expand_dropout) */
    int32_t n_splits;       /* <= 1: off */
    int32_t form;           /* 0 (or POD_WINO_FORM_4): the shipped kernel; POD_WINO_FORM_8: experiment builds only (see above) */
    int64_t split_stride;
    const int32_t* live_blocks; /* NULL, or pod_sparse_live_blocks' device list {count, ..., entries {record, need bits}}: only those records are
                                   computed, and the patch pixels whose need bit is clear are read as 0.0 */
    PodConvSet sets[4];
} PodWinoConv;
int64_t pod_wino_filter_split_bytes(int32_t K, int32_t C);
int pod_wino_filter_transform_split(const float* weight, void* Us, int32_t K, int32_t C, pod_stream_t stream);
int pod_wino_conv3x3_split(const PodWinoConv* conv, pod_stream_t stream);

/* ---- sparse bbox tower (round 5; csrc/k15_sparse_blocks.hip) ------------------------------------------------------------------------
 * probabilistic_inference.py:310-331 reads box_delta / box_reg_var only at the candidates of :300-308, while the head
 * (probabilistic_retinanet.py:518-537) evaluates bbox_subnet / bbox_pred / bbox_cov densely for every run.  With the cls tower evaluated
 * first: pod_sparse_reach turns pod_level_topk's selection into a per-cell map (one byte per cell of every level, level after level;
 * `scratch` as large) of how many convolution layers below the predictors a cell is still needed (0: a candidate's own cell .. 5; 255:
 * never -- Winograd tiles taken into account); pod_sparse_live_blocks lists the records of a pod_wino_conv3x3 table that hold a cell of
 * reach <= max_reach (rec_level[r] = FPN level of record r) for PodWinoConv.live_blocks: int32 word 0 = the count, then from word
 * POD_SPARSE_LIVE_HEAD one entry of POD_SPARSE_LIVE_STRIDE words per live record (order unspecified): {record index, 11 words of NEED
 * bits: bit 18 py + px <=> pixel (py, px) of the record's 18 x 18 input patch is an image cell of reach <= in_reach}.  The convolution
 * reads a patch pixel whose bit is clear as 0.0: with in_reach = max_reach + 1 (the reach the layer below was launched with; 255 for a
 * dense input) a launch never reads what an earlier image left in a buffer, every abs-max record is per image, and the tower's outputs
 * are a function of the image alone (ABI 14).  Needed cells depend on needed cells only: the detections equal the dense tower's (to the
 * last bits where an abs-max record differs in its binade). */
#define POD_SPARSE_LIVE_HEAD 4
#define POD_SPARSE_LIVE_STRIDE 12
int pod_sparse_reach(const PodConfig* cfg, const PodLevel* levels, const uint64_t* cat_keys, const int32_t* cat_level, const int32_t* n_total,
                     uint8_t* reach, uint8_t* scratch, pod_stream_t stream);
int pod_sparse_live_blocks(const PodConfig* cfg, const PodLevel* levels, const int32_t* records, const int32_t* rec_level, int32_t n_records,
                           const uint8_t* reach, int32_t max_reach, int32_t in_reach, int32_t* live, pod_stream_t stream);
/* pod_wino_reduce: the partial sums of an n_splits launch -> bias (K values, zero-padded) + ReLU -> the k_real planes of ONE NCHW image (HW =
 * out_pixels).  Replaces the same reference lines as pod_wino_conv3x3 (detectron2 BottleneckBlock.conv2, FPN.output_convs). */
int pod_wino_reduce(const float* partials, int32_t n_splits, int64_t split_stride, const float* bias, float* planes, int64_t HW,
                    int32_t K, int32_t k_real, int32_t relu, float* out_amax, pod_stream_t stream);

/* ---- the ResNet stem, channels-last (round 4; ABI 10; csrc/k14_stem_conv.hip) -------------------------------------------
 * Replaces detectron2's BasicStem as probabilistic_retinanet.py:96-100 runs it (`features = self.backbone(images.tensor)`):
 * conv1 (7x7, stride 2, padding 3, 3 -> 64 channels, FrozenBN folded into weight + bias) + ReLU, then max_pool2d(3, 2, 1).
 * pod_stem7x7_filter_split: weight (64, 3, 7, 7) fp32 -> Ws, 2 * 64 * 192 f16 values (the window padded to 8 x 8 with zeros, two f16
 * terms per scaled value) + a 16-byte trailer (the weight's abs-max): 64 * 192 * 4 + 16 bytes.  pod_stem7x7_split: x = the frame as (3, H_img, W_img) planes, fp32 or uint8 (x_is_u8); mean / stddev non-null:
 * the kernel normalises on load, (x - mean[c]) / stddev[c] in fp32 as PR:96's `self.normalizer` does (null: x is normalised already);
 * H x W >= H_img x W_img is the padded extent ImageList.from_tensors gives the frame (zeros outside the frame, as its padding and the
 * conv's own are) -> y ((H-1)/2+1) x ((W-1)/2+1) pixels x 64 channels, channels-last; every fp32 product from 2-way f16 splits of the
 * scaled operands (pod_conv1x1_split's arithmetic); in_amax: a device word >= max |normalised x| (operand abs-max words, above), out_amax:
 * NULL or the zeroed word that receives max |y|.  pod_maxpool3x3s2_cl: (H * W, C) channels-last -> ((H-1)/2+1) x ((W-1)/2+1) x C,
 * C % 4 == 0; taps outside the map do not take part (torch's -inf padding). */
int pod_stem7x7_filter_split(const float* weight, void* Ws, pod_stream_t stream);
int pod_stem7x7_split(const void* x, int32_t x_is_u8, int32_t H_img, int32_t W_img, const float* mean, const float* stddev, float* y, const void* Ws,
                      const float* bias, int32_t H, int32_t W, int32_t relu, const float* in_amax, float* out_amax, pod_stream_t stream);
int pod_maxpool3x3s2_cl(const float* x, float* y, int32_t H, int32_t W, int32_t C, pod_stream_t stream);
/* pod_im2col3x3s2_cl (ABI 13): the patch matrix of a 3x3 / stride 2 / padding 1 convolution on a channels-last map -- x (H * W, C) ->
 * y (((H-1)/2+1) * ((W-1)/2+1), 9 C), row (oy, ox) = the C-vectors of the nine taps in (ty, tx) order, zeros where a tap falls outside
 * the map; relu != 0: max(x, 0) on the way.  pod_conv1x1_split on y with the weight laid out (Cout, ty, tx, Cin) is then the convolution:
 * detectron2's LastLevelP6P7 (p6 = conv(res5), p7 = conv(relu(p6)); the last two convolutions of `self.backbone(images.tensor)`,
 * probabilistic_retinanet.py:96-100) without MIOpen, bit-reproducible run to run.  C % 4 == 0; an upper bound of |x| bounds |y|. */
int pod_im2col3x3s2_cl(const float* x, float* y, int32_t H, int32_t W, int32_t C, int32_t relu, pod_stream_t stream);

/* ---- ground-truth matching (offline metrics, SURVEY f-1) ------------------------------------------
 * Replaces: match_predictions_to_groundtruth core/evaluation_tools/evaluation_utils.py:191-367 for a whole data set
 * in one call.  Images are concatenated: image m owns detections [det_off[m], det_off[m+1]) and ground-truth boxes
 * [gt_off[m], gt_off[m+1]) (<= 128 detections per image); det_img / gt_img give the image slot of every row.
 * Outputs: gt_fn[g] = 1 iff every IoU <= iou_min (EU:245); gt_match_count[g] detections with IoU >= iou_correct, their
 * global indices / IoUs in gt_match_idx / gt_match_iou (n_gt x 128) ordered by descending max class probability
 * (entry 0 = the true positive, the rest duplicates, EU:282-299); det_fp[d] = 1 iff IoU <= iou_min with every
 * ground truth of its image (EU:255; an image without ground truth makes all its detections false positives). */
int pod_match_groundtruth(const float* det_boxes, const float* det_probs, const int32_t* det_off, const int32_t* det_img,
                          int32_t n_det, const float* gt_boxes, const int32_t* gt_off, const int32_t* gt_img, int32_t n_gt,
                          int32_t num_classes, float iou_min, float iou_correct, int32_t* gt_fn, int32_t* gt_match_count,
                          int32_t* gt_match_idx, float* gt_match_iou, int32_t* det_fp, pod_stream_t stream);

/* ---- NLL scoring rule ----------------------------------------------------------------------
 * Replaces: compute_reg_scores core/evaluation_tools/scoring_rules.py:68-74
 * (-MVN(mean, cov + 1e-2 I).log_prob(gt), the "NLL parity" half of the metric). */
int pod_reg_nll(const float* means, const float* covs, const float* gt, int32_t n, float* nll, pod_stream_t stream);

/* (test support -- the dumps of the in-kernel Philox draws and of the f16 split -- is declared in include/pod_mi355x_test.h: the library
 * exports those three entry points for tests/ and tools/, they are not part of the drop-in boundary.) */

/* ---- K13 conv1x1_split (round 4): the 1x1 convolutions of the backbone / FPN as a channels-last GEMM ------------------------
 * Replaces: detectron2 BottleneckBlock.conv1 / conv3 / shortcut (1x1, stride 1 or 2, FrozenBN folded, ReLU, residual add) and
 * FPN.lateral_convs as `self.backbone(images.tensor)` runs them (probabilistic_retinanet.py:96-100): 39 calls per image that PyTorch-ROCm
 * executes as fp32 GEMMs + one element-wise pass each.  Same arithmetic contract as pod_wino_conv3x3_split (round 5: 2-way f16 splits of
 * the power-of-two-scaled operands, 3 partial products, fp32 accumulate; in_amax / out_amax: operand abs-max words, above).
 *   y[p][k] = act(sum_c x[pin(p)][c] w[k][c] + bias[k] (+ residual[p][k])),  p = oy * W_out + ox,  pin = (stride oy) * W_in + stride ox
 * x dev (H_in * W_in, Cin), y / residual dev (H_out * W_out, Cout): channels-last.  Cin % 16 == 0, Cout % 64 == 0.
 * Ws = pod_conv1x1_filter_split(weight (Cout, Cin) fp32): pod_conv1x1_filter_split_bytes(Cout, Cin) bytes (2 * Cout * Cin f16 terms + a 16-byte trailer: the
 * weight's abs-max), 16-byte aligned.
 * n_splits > 1 (small maps): the input channels cut over workgroup sets, partial sums in `partials` (n_splits * H_out * W_out * Cout
 * floats), added in a fixed order with bias / residual / ReLU by a second launch.  waves (round 5): 1, 2 or 4 wavefronts of ONE workgroup
 * share a tile's K range and add their accumulators in LDS in a fixed order -- split-K without the partial sums' trip through HBM or a
 * second launch, to the bit the result of the same cut over workgroup sets; 0: chosen by the library (what the model uses).
 * flags: POD_C1_RELU (1) = ReLU; POD_C1_RESIDUAL_UP2 (2, n_splits == 1 only) = `residual` is the HALF-resolution map, ((H_out + 1) / 2) x
 * ((W_out + 1) / 2) pixels, and pixel (y, x) adds residual[(y >> 1) * ((W_out + 1) / 2) + (x >> 1)]: detectron2 FPN's top-down path,
 * `lateral + F.interpolate(top_down, scale_factor=2, mode="nearest")`, without materialising the upsampled map. */
#define POD_C1_RELU 1
#define POD_C1_RESIDUAL_UP2 2
int64_t pod_conv1x1_filter_split_bytes(int32_t Cout, int32_t Cin);
int pod_conv1x1_filter_split(const float* weight, void* Ws, int32_t Cout, int32_t Cin, pod_stream_t stream);
int pod_reduce_partials(const float* partials, int32_t n_splits, int64_t split_stride, const float* bias, const float* residual, float* y,
                        int64_t n, int32_t Cout, int32_t relu, float* out_amax, pod_stream_t stream);   /* y = act(sum_s partials[s] + bias + residual), channels-last */
int pod_conv1x1_split(const float* x, float* y, const void* Ws, const float* bias, const float* residual, int32_t H_out, int32_t W_out,
                      int32_t H_in, int32_t W_in, int32_t stride, int32_t Cin, int32_t Cout, int32_t flags, int32_t n_splits, float* partials,
                      int32_t waves, const float* in_amax, float* out_amax, pod_stream_t stream);

/* ---- one image, one call --------------------------------------------------------------------
 * Replaces: everything `RetinaNetProbabilisticPredictor.__call__` does after the conv net
 * (PI:86-111 -> PI:178-388 -> the mode's post-processing -> IU:374-425), i.e. the launch sequence
 *   pod_merge_score_fused (= pod_mc_merge_score + pod_score_maybe) -> pod_level_topk -> pod_gather_decode (= pod_gather_candidates + pod_decode_cov)
 *   -> pod_nms_cluster -> {pod_bayes_fuse | pod_anchor_stats_merge | -} -> pod_finalize
 * enqueued from C on `stream`, in-kernel Philox draws (levels[].eps_cls must be NULL: the eps-replay parity
 * mode needs the host between launches and uses the individual entry points).  Nothing here
 * synchronises or allocates; the workspace is caller-owned, every pointer is device memory sized as the
 * individual entry points document, n_capacity = n_levels * topk.  mean_delta / mean_reg_var may be NULL
 * (the merged box planes are not needed downstream), mean_cls / mean_cls_var only when n_runs == 1. */
typedef struct PodWorkspace {
    const float* anchors;          /* (R, 4) level-concatenated */
    float* mean_cls;  float* mean_cls_var;  float* mean_delta;  float* mean_reg_var;
    uint64_t* cand_keys;  int32_t* cand_count;  uint64_t* maybe_bits;
    uint64_t* sel_keys;   int32_t* sel_count;
    uint64_t* cat_keys;   int32_t* cat_level;          /* level-concatenated selection, n_capacity rows each */
    float* probs_dense;                                /* (R, K) or NULL when the model has no variance head */
    int32_t* n_total;     int32_t* cand_anchor_idx;  int32_t* cand_level;  int32_t* cand_class;
    float* cand_score;    float* cand_probs;  float* cand_delta;  float* cand_reg_var;  float* cand_anchor;
    float* cand_run_delta;
    float* boxes;  float* cov;
    int32_t* keep;  int32_t* n_keep;  void* nms_scratch;
    float* m_boxes;  float* m_cov;  float* m_scores;  int32_t* m_classes;  float* m_probs;
    int32_t n_capacity;  int32_t reserved;
} PodWorkspace;

#define POD_MODE_STANDARD_NMS 0    /* also the pre-NMS MC-dropout / ensemble modes (PI:402-442, PI:483-505) */
#define POD_MODE_BAYES_OD 1
#define POD_MODE_ANCHOR_STATISTICS 2

/* image_h/w: network input size (IU:39-41), out_h/w: output resolution (PI:106-107);
 * box_merge_mode / cls_merge_mode as in pod_bayes_fuse (ignored by the other modes). */
int pod_run_image(const PodConfig* cfg, const PodLevel* levels, const PodWorkspace* ws, int32_t mode,
                  int32_t box_merge_mode, int32_t cls_merge_mode, int32_t image_h, int32_t image_w,
                  int32_t out_h, int32_t out_w, const PodDetections* out, pod_stream_t stream);
/* The same in two parts (round 5): parts = 1 SELECT (merge + score + per-level top-k, PI:211-308: the candidates stand in ws->cat_keys /
 * cat_level / n_total afterwards; levels[l].delta / reg_var are not read and may be NULL, `out` too), 2 FINISH (gather + decode + NMS / fusion +
 * finalize, PI:310-636), 3 both.  The sparse bbox tower runs between 1 and 2. */
int pod_run_image_part(const PodConfig* cfg, const PodLevel* levels, const PodWorkspace* ws, int32_t mode,
                  int32_t box_merge_mode, int32_t cls_merge_mode, int32_t image_h, int32_t image_w,
                  int32_t out_h, int32_t out_w, const PodDetections* out, int32_t parts, pod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* POD_MI355X_H */
