// pod_run_image -- the whole post-conv-net path of one image enqueued from C.
//
// Replaces: RetinaNetProbabilisticPredictor.__call__ after the model forward (probabilistic_inference.py:86-111):
// retinanet_probabilistic_inference PI:178-388, then post_processing_standard_nms PI:390-400 /
// post_processing_bayes_od PI:536-636 / general_anchor_statistics_postprocessing IU:57-162 (the pre-NMS MC-dropout
// and ensemble modes PI:402-442, PI:483-505 are the standard-NMS branch with n_runs > 1), then
// probabilistic_detector_postprocess IU:374-425.
//
// Host code only: it calls the entry points of include/pod_mi355x.h in the reference's order on one stream.  Doing
// this in C instead of ten ctypes calls matters because the device finishes an image in ~100 us: the Python launch
// overhead (~5-8 us per call with 15-20 marshalled arguments) otherwise leaves gaps on the device timeline.
#include "pod_device.h"

#define POD_TRY(call)              \
    do {                           \
        const int rc_ = (call);    \
        if (rc_ != POD_OK) return rc_; \
    } while (0)

// parts: 1 = SELECT (merge + score + per-level top-k: PI:211-308 -- the candidates stand in ws->cat_keys / cat_level / n_total afterwards and
// levels[l].delta / reg_var are not read), 2 = FINISH (gather + decode + NMS / fusion + finalize: PI:310-636), 3 = both = pod_run_image.
// The sparse bbox tower (pod_sparse_reach / pod_sparse_live_blocks) runs between 1 and 2.
extern "C" int pod_run_image_part(const PodConfig* cfg, const PodLevel* levels, const PodWorkspace* ws, int32_t mode,
                                  int32_t box_merge_mode, int32_t cls_merge_mode, int32_t image_h, int32_t image_w,
                                  int32_t out_h, int32_t out_w, const PodDetections* out, int32_t parts, pod_stream_t stream) {
    if (!cfg || !levels || !ws || (!out && (parts & 2)) || parts < 1 || parts > 3) return POD_E_INVALID;
    if (mode != POD_MODE_STANDARD_NMS && mode != POD_MODE_BAYES_OD && mode != POD_MODE_ANCHOR_STATISTICS) return POD_E_INVALID;
    if (image_h < 1 || image_w < 1 || out_h < 1 || out_w < 1) return POD_E_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > POD_MAX_LEVELS) return POD_E_INVALID;
    if (ws->n_capacity != cfg->n_levels * cfg->topk) return POD_E_INVALID;
    for (int l = 0; l < cfg->n_levels; ++l)
        if (levels[l].eps_cls) return POD_E_INVALID;   // eps-replay needs the host between launches
    const bool merged = cfg->n_runs > 1;
    const bool prune = cfg->has_cls_var != 0;           // native draws + variance head: K1 flags, K1b samples
    const bool has_cov = cfg->cov_dims > 0 || merged;   // PI:381: otherwise the reference carries no covariance
    if (prune && !ws->probs_dense) return POD_E_INVALID;
    if (!ws->cat_keys || !ws->cat_level || !ws->n_total) return POD_E_INVALID;
    if (mode == POD_MODE_BAYES_OD && !has_cov) return POD_E_INVALID;

    // merge + score in one streaming launch (round 4; k1f_merge_score_fused.hip).  The merged class planes are not stored: nothing
    // downstream reads them (the gather kernel merges the box channels at the candidates and takes the class probabilities from
    // probs_dense, or evaluates them itself when there is no variance head).
    if (parts & 1) {
        POD_TRY(pod_merge_score_fused(cfg, levels, nullptr, nullptr, ws->cand_keys, ws->cand_count, prune ? ws->probs_dense : nullptr, stream));
        POD_TRY(pod_level_topk(cfg, levels, ws->cand_keys, ws->cand_count, ws->sel_keys, ws->sel_count, ws->cat_keys, ws->cat_level,
                               ws->n_total, stream));
    }
    if (!(parts & 2)) return POD_OK;
    POD_TRY(pod_gather_decode(cfg, levels, ws->anchors, ws->cat_keys, ws->cat_level, ws->n_total, ws->cand_count,
                              prune ? ws->probs_dense : nullptr, ws->cand_anchor_idx, ws->cand_level,
                              ws->cand_score, ws->cand_class, ws->cand_probs, ws->cand_delta,
                              cfg->cov_dims > 0 ? ws->cand_reg_var : nullptr, ws->cand_anchor, ws->cand_run_delta,
                              ws->boxes, ws->cov, stream));
    POD_TRY(pod_nms_cluster(cfg, ws->n_total, ws->n_capacity, ws->boxes, ws->cand_score, ws->cand_class, ws->keep, ws->n_keep,
                            ws->nms_scratch, stream));
    const float* cov_in = has_cov ? ws->cov : nullptr;
    // IU:394-396: scale factors are Python floats (doubles) rounded once to fp32
    const float sx = (float)((double)out_w / (double)image_w), sy = (float)((double)out_h / (double)image_h);
    const float oh = (float)out_h, ow = (float)out_w;
    // (K5 / K6 with K7's body run by the last cluster workgroup -- device-scope fence + ticket -- was built and measured:
    //  17.6 us against 9.7 + 5.9 us for the two launches; the fence costs more than the launch it saves.  See DESIGN.md.)
    if (mode == POD_MODE_BAYES_OD) {
        POD_TRY(pod_bayes_fuse(cfg, ws->n_total, ws->keep, ws->n_keep, ws->boxes, ws->cov, ws->cand_score, ws->cand_class,
                               ws->cand_probs, box_merge_mode, cls_merge_mode, ws->m_boxes, ws->m_cov, ws->m_scores,
                               ws->m_classes, ws->m_probs, stream));
    } else if (mode == POD_MODE_ANCHOR_STATISTICS) {
        POD_TRY(pod_anchor_stats_merge(cfg, ws->n_total, ws->keep, ws->n_keep, ws->boxes, cov_in, ws->cand_class, ws->cand_probs,
                                       ws->m_boxes, ws->m_cov, ws->m_scores, ws->m_classes, ws->m_probs, stream));
    }
    if (mode == POD_MODE_STANDARD_NMS)
        return pod_finalize(cfg, ws->keep, ws->n_keep, ws->boxes, cov_in, ws->cand_score, ws->cand_class, ws->cand_probs, sx, sy,
                            oh, ow, out->boxes, out->cov, out->scores, out->classes, out->probs, out->records, out->n_det, stream);
    return pod_finalize(cfg, nullptr, ws->n_keep, ws->m_boxes, ws->m_cov, ws->m_scores, ws->m_classes, ws->m_probs, sx, sy,
                        oh, ow, out->boxes, out->cov, out->scores, out->classes, out->probs, out->records, out->n_det, stream);
}

extern "C" int pod_run_image(const PodConfig* cfg, const PodLevel* levels, const PodWorkspace* ws, int32_t mode,
                             int32_t box_merge_mode, int32_t cls_merge_mode, int32_t image_h, int32_t image_w,
                             int32_t out_h, int32_t out_w, const PodDetections* out, pod_stream_t stream) {
    return pod_run_image_part(cfg, levels, ws, mode, box_merge_mode, cls_merge_mode, image_h, image_w, out_h, out_w, out, 3, stream);
}
