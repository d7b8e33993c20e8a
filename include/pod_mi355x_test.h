/* Test support of libpod_mi355x.so: entry points that exist for tests/ and tools/ only (round 6: moved out of pod_mi355x.h, which declares
 * the drop-in boundary and nothing else).  They expose arithmetic the product kernels perform internally, so that a test can check it against
 * the oracle; nothing in pod_compare_amd/'s product path needs them to produce detections. */
#ifndef POD_MI355X_TEST_H
#define POD_MI355X_TEST_H

#include "pod_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the native-RNG draws, written out ---------------------------------------------------------
 * The in-kernel Philox draws that replace Normal(...).rsample((cls_samples,)) PI:291-294 and
 * MultivariateNormal(...).rsample((1000,)) PI:351-356 are never stored by the product path.  These two entry points
 * evaluate the same counter -> normal maps for the Philox key in cfg->philox_seed and write them in the reference's
 * tensor layouts, so a test can feed the CPU oracle exactly the draws pod_run_image used:
 *   pod_dump_cls_normals: eps_cls dev (cls_samples, H_l*W_l*A, K) of level `level`;
 *   pod_dump_box_normals: eps_prop dev (prop_samples, n, 4), row i = the draws of global anchor id
 *                         global_anchor_ids[i] (= anchor_base_l + index inside the level). */
int pod_dump_cls_normals(const PodConfig* cfg, const PodLevel* levels, int32_t level, float* eps_cls, pod_stream_t stream);
int pod_dump_box_normals(const PodConfig* cfg, const int32_t* global_anchor_ids, int32_t n, float* eps_prop, pod_stream_t stream);
/* test support for the split kernels' arithmetic contract (tests/test_wino_conv_gpu.py): terms[2][n] f16 bit patterns (dev uint16, n even): x[i] * scale = t0 + t1 to 2^-23 |x[i] scale| (scale a power of two; the round-5 split kernels' own code) */
int pod_debug_f16_split2(const float* x, float scale, void* terms, int64_t n, pod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
