// Ground-truth matching for the offline probabilistic metrics.
//
// Replaces: match_predictions_to_groundtruth, core/evaluation_tools/evaluation_utils.py:191-367 -- a Python loop
// over images and ground-truth boxes that re-concatenates its result tensors on every hit.  Here the whole data set is
// one launch: images are concatenated, `*_off` give each image's slice.
//   k_match_gt : one wavefront per ground-truth box.  Lanes take the image's detections: IoU (detectron2
//                pairwise_iou), "fully missed" test (all IoU <= iou_min, EU:245), the detections with
//                IoU >= iou_correct (EU:269) ranked by descending max class probability (EU:286-288; ties: lower
//                detection index first) -- rank 0 is the true positive, the rest are duplicates.
//   k_match_det: one thread per detection: false positive iff IoU <= iou_min with every ground truth (EU:255).
// As in the reference (whose `gt_idxs_processed` is never filled, EU:270-281) a detection may match several boxes.
#include "pod_device.h"

namespace pod {

constexpr int MATCH_MAX_DET = 128;   // detections per image (max_detections_per_image = 100)

struct KMatchParams {
    const float* det_boxes;
    const float* det_probs;
    const int32_t* det_off;     // n_images + 1
    const float* gt_boxes;
    const int32_t* gt_off;      // n_images + 1
    const int32_t* gt_img;      // image slot of every ground-truth box
    const int32_t* det_img;     // image slot of every detection
    int32_t n_gt, n_det, K;
    float iou_min, iou_correct;
    int32_t* gt_fn;             // 1 = false negative
    int32_t* gt_match_count;
    int32_t* gt_match_idx;      // (n_gt, MATCH_MAX_DET) global detection indices, best first
    float* gt_match_iou;        // (n_gt, MATCH_MAX_DET)
    int32_t* det_fp;            // 1 = false positive
};

__global__ void __launch_bounds__(64) k_match_gt(const KMatchParams P) {
    __shared__ float s_score[MATCH_MAX_DET];
    __shared__ float s_iou[MATCH_MAX_DET];
    __shared__ int s_hit[MATCH_MAX_DET];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= P.n_gt) return;
    const int img = P.gt_img[g];
    const int d0 = P.det_off[img], nd = min(P.det_off[img + 1] - d0, MATCH_MAX_DET);
    const Box gb = load_box(P.gt_boxes, g);
    bool missed = true;
    for (int j = lane; j < MATCH_MAX_DET; j += 64) {
        float iou = 0.0f, sc = 0.0f;
        int hit = 0;
        if (j < nd) {
            iou = iou_pair(gb, load_box(P.det_boxes, d0 + j));
            if (!(iou <= P.iou_min)) missed = false;
            hit = iou >= P.iou_correct ? 1 : 0;
            const float* pr = P.det_probs + (size_t)(d0 + j) * P.K;
            sc = pr[0];
            for (int k = 1; k < P.K; ++k) sc = fmaxf(sc, pr[k]);
        }
        s_score[j] = sc;
        s_iou[j] = iou;
        s_hit[j] = hit;
    }
    __syncthreads();
    const unsigned long long any_seen = __ballot(!missed);
    int count = 0;
    for (int j = 0; j < nd; ++j) count += s_hit[j];
    for (int j = lane; j < nd; j += 64) {
        if (!s_hit[j]) continue;
        int rank = 0;
        const float mine = s_score[j];
        for (int q = 0; q < nd; ++q)
            if (s_hit[q] && (s_score[q] > mine || (s_score[q] == mine && q < j))) ++rank;
        P.gt_match_idx[(size_t)g * MATCH_MAX_DET + rank] = d0 + j;
        P.gt_match_iou[(size_t)g * MATCH_MAX_DET + rank] = s_iou[j];
    }
    if (lane == 0) {
        P.gt_fn[g] = any_seen == 0ull ? 1 : 0;
        P.gt_match_count[g] = count;
    }
}

__global__ void __launch_bounds__(256) k_match_det(const KMatchParams P) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= P.n_det) return;
    const int img = P.det_img[d];
    const Box db = load_box(P.det_boxes, d);
    bool fp = true;
    for (int g = P.gt_off[img]; g < P.gt_off[img + 1]; ++g)
        if (!(iou_pair(load_box(P.gt_boxes, g), db) <= P.iou_min)) fp = false;
    P.det_fp[d] = fp ? 1 : 0;
}

}  // namespace pod

extern "C" int pod_match_groundtruth(const float* det_boxes, const float* det_probs, const int32_t* det_off, const int32_t* det_img,
                                     int32_t n_det, const float* gt_boxes, const int32_t* gt_off, const int32_t* gt_img,
                                     int32_t n_gt, int32_t num_classes, float iou_min, float iou_correct, int32_t* gt_fn,
                                     int32_t* gt_match_count, int32_t* gt_match_idx, float* gt_match_iou, int32_t* det_fp,
                                     pod_stream_t stream) {
    if (n_det < 0 || n_gt < 0 || num_classes < 1) return POD_E_INVALID;
    if (n_det > 0 && (!det_boxes || !det_probs || !det_off || !det_img || !det_fp || !gt_off)) return POD_E_INVALID;
    if (n_gt > 0 && (!gt_boxes || !gt_off || !gt_img || !gt_fn || !gt_match_count || !gt_match_idx || !gt_match_iou || !det_off))
        return POD_E_INVALID;
    pod::KMatchParams P;
    P.det_boxes = det_boxes; P.det_probs = det_probs; P.det_off = det_off; P.gt_boxes = gt_boxes; P.gt_off = gt_off;
    P.gt_img = gt_img; P.det_img = det_img; P.n_gt = n_gt; P.n_det = n_det; P.K = num_classes; P.iou_min = iou_min;
    P.iou_correct = iou_correct; P.gt_fn = gt_fn; P.gt_match_count = gt_match_count; P.gt_match_idx = gt_match_idx;
    P.gt_match_iou = gt_match_iou; P.det_fp = det_fp;
    if (n_gt > 0) {
        hipLaunchKernelGGL(pod::k_match_gt, dim3(n_gt), dim3(64), 0, (hipStream_t)stream, P);
        POD_CHECK_LAUNCH();
    }
    if (n_det > 0) {
        hipLaunchKernelGGL(pod::k_match_det, dim3((n_det + 255) / 256), dim3(256), 0, (hipStream_t)stream, P);
        POD_CHECK_LAUNCH();
    }
    return POD_OK;
}
