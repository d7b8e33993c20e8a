// K15 sparse_blocks -- which 16x16-pixel blocks of the bbox tower have to be computed at all (round 5).
//
// Replaces nothing in the reference and changes no result: probabilistic_inference.py:310-331 reads box_delta / box_reg_var (and every
// run's deltas, :325-331) ONLY at the <= 1000-per-level anchors that survive the score threshold and the top-k (:300-308), yet
// ProbabilisticRetinaNetHead.forward (probabilistic_retinanet.py:518-537) evaluates bbox_subnet + bbox_pred + bbox_cov densely for every
// run -- half of the head's convolution work.  With the cls tower evaluated first, the candidates are known before the bbox tower runs;
// the tower is then launched over the blocks whose outputs can reach a candidate and nowhere else.
//
// Reach.  pod_wino_conv3x3_split computes 2 x 4 output tiles from 4 x 6 input patches, so a needed output at cell c needs the INPUT
// cells of rows [c.y - 2, c.y + 2] and columns [c.x - 4, c.x + 4] (tile alignment + halo), whatever the tile phase.  Layer by layer,
// from the predictors (needed: the candidate cells, reach 0) down to the first conv of the subnet (reach 4):
//     needed(reach j) = candidates (+) box(2 j rows, 4 j columns)
// and needed cells only ever depend on needed cells of the layer below.  What stands in the REST of a live block's patch influences only
// outputs nobody reads -- but it would influence the launch's abs-max record (the f16 split's operand scale), so since round 6 a launch
// reads every input cell the layer below did not have to compute for THIS image as 0.0 (the need bits of k_sparse_live): nothing an
// earlier image left in a buffer is ever read, the records are per image, and the tower's results are a function of the image alone.
// reach[cell] = the smallest j with cell in needed(j) (0..POD_SPARSE_MAX_REACH, 255: none), by a separable pass over the candidate mask;
// a table record (block) is LIVE for reach j iff one of its 256 canvas pixels is an image cell with reach <= j.
#include "pod_device.h"

namespace pod {

constexpr int SPARSE_MAX_REACH = 5;

struct SparseGeom {
    int32_t H[POD_MAX_LEVELS], W[POD_MAX_LEVELS], cell_base[POD_MAX_LEVELS + 1];
    int32_t n_levels, A;
};

__device__ __forceinline__ int sparse_level_of(const SparseGeom& G, int cell) {
    int l = 0;
    while (l + 1 < G.n_levels && cell >= G.cell_base[l + 1]) ++l;
    return l;
}

// candidate row i of the level-concatenated selection -> its cell := 0 (the map was filled with 255)
__global__ void __launch_bounds__(256) k_sparse_mark(const uint64_t* __restrict__ cat_keys, const int32_t* __restrict__ cat_level, const int32_t* __restrict__ n_total,
                                                     int32_t cap, SparseGeom G, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = *n_total;
    n = n < cap ? n : cap;
    if (i >= n) return;
    const int l = cat_level[i];
    mask[G.cell_base[l] + key_index(cat_keys[i]) / G.A] = 0;
}

// columns: h[y][x] = min over |dx| <= 4 R of ceil(|dx| / 4) where mask[y][x + dx] == 0
__global__ void __launch_bounds__(256) k_sparse_reach_x(const uint8_t* __restrict__ mask, uint8_t* __restrict__ hx, SparseGeom G) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= G.cell_base[G.n_levels]) return;
    const int l = sparse_level_of(G, cell), W = G.W[l], local = cell - G.cell_base[l], y = local / W, x = local - y * W;
    const uint8_t* row = mask + G.cell_base[l] + y * W;
    int best = 255;
    for (int dx = -4 * SPARSE_MAX_REACH; dx <= 4 * SPARSE_MAX_REACH; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W || row[xx] != 0) continue;
        const int j = ((dx < 0 ? -dx : dx) + 3) >> 2;
        best = j < best ? j : best;
    }
    hx[cell] = (uint8_t)best;
}
// rows: reach[y][x] = min over |dy| <= 2 R of max(ceil(|dy| / 2), h[y + dy][x])
__global__ void __launch_bounds__(256) k_sparse_reach_y(const uint8_t* __restrict__ hx, uint8_t* __restrict__ reach, SparseGeom G) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= G.cell_base[G.n_levels]) return;
    const int l = sparse_level_of(G, cell), W = G.W[l], H = G.H[l], local = cell - G.cell_base[l], y = local / W, x = local - y * W;
    int best = 255;
    for (int dy = -2 * SPARSE_MAX_REACH; dy <= 2 * SPARSE_MAX_REACH; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const int jy = ((dy < 0 ? -dy : dy) + 1) >> 1, jx = hx[G.cell_base[l] + yy * W + x];
        const int j = jy > jx ? jy : jx;
        best = j < best ? j : best;
    }
    reach[cell] = (uint8_t)best;
}

// one workgroup per table record: live iff a canvas pixel of the block is an image cell within `max_reach`.  live[0] counts; entry e (int32
// words POD_SPARSE_LIVE_HEAD + POD_SPARSE_LIVE_STRIDE e ..) = {record, 11 words of NEED bits}: bit (18 py + px) says that patch pixel (py, px)
// of the block's 18 x 18 INPUT patch is an image cell the layer below had to compute (reach <= in_reach; >= 255: every cell).  The convolution
// reads every other patch pixel as 0.0, so a launch never reads what an earlier image left in a block (round 6: the tower's results are
// a function of the image alone).  Entries are appended in the order the workgroups arrive: the ORDER of the list is unspecified (blocks
// are independent; nothing downstream depends on it).
__global__ void __launch_bounds__(256) k_sparse_live(const int4* __restrict__ records, const int32_t* __restrict__ rec_level, SparseGeom G,
                                                     const uint8_t* __restrict__ reach, int32_t max_reach, int32_t in_reach, int32_t* __restrict__ live) {
    __shared__ int32_t s_slot;
    const int r = blockIdx.x, t = threadIdx.x;
    const int4 d = records[r];
    const int l = rec_level[r];
    const int gcols = (d.z >> 24) & 0xFF, H = (d.z >> 12) & 0xFFF, W = d.z & 0xFFF, n_img = (d.w >> 24) & 0xFF;
    const int y0 = ((d.w >> 12) & 0xFFF) * 16, x0 = (d.w & 0xFFF) * 16;
    const uint8_t* const lr = reach + G.cell_base[l];
    auto cell_reach = [&](int vy, int vx) {                       // canvas pixel -> reach of its image cell (256: not a cell of any image)
        if (vy < 0 || vx < 0) return 256;
        const int m = vy / (H + 1), gy = vy - m * (H + 1), n = vx / (W + 1), gx = vx - n * (W + 1);
        return (gy < H && gx < W && n < gcols && m * gcols + n < n_img) ? (int)lr[gy * W + gx] : 256;
    };
    const bool hit = cell_reach(y0 + (t >> 4), x0 + (t & 15)) <= max_reach;
    if (!__syncthreads_or(hit)) return;
    if (t == 0) s_slot = atomicAdd(&live[0], 1);
    // need bits of the 18 x 18 patch (canvas rows y0 - 1 .. y0 + 16): thread t -> patch pixels t and t + 256
    const int all = in_reach >= 255;
    uint64_t b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = t + 256 * i, py = pp / 18, px = pp - py * 18;
        const int rc = pp < 324 ? cell_reach(y0 - 1 + py, x0 - 1 + px) : 256;
        b[i] = __ballot(rc < 256 && (all || rc <= in_reach));
    }
    __syncthreads();
    int32_t* e = live + POD_SPARSE_LIVE_HEAD + (int64_t)POD_SPARSE_LIVE_STRIDE * s_slot;
    const int wave = t >> 6;
    if ((t & 63) == 0) {
        e[1 + 2 * wave] = (int32_t)(uint32_t)b[0];
        e[2 + 2 * wave] = (int32_t)(uint32_t)(b[0] >> 32);
        if (wave == 0) { e[9] = (int32_t)(uint32_t)b[1]; e[10] = (int32_t)(uint32_t)(b[1] >> 32); }
        if (wave == 1) e[11] = (int32_t)(uint32_t)b[1];
    }
    if (t == 0) e[0] = r;
}

}  // namespace pod

// reach (device, one byte per cell of every level, level after level: sum_l H_l W_l bytes) and scratch (as large) from the candidates
// K2 selected (cat_keys / cat_level / n_total: pod_level_topk's level-concatenated output).
extern "C" int pod_sparse_reach(const PodConfig* cfg, const PodLevel* levels, const uint64_t* cat_keys, const int32_t* cat_level, const int32_t* n_total,
                                uint8_t* reach, uint8_t* scratch, pod_stream_t stream) {
    if (!cfg || !levels || !cat_keys || !cat_level || !n_total || !reach || !scratch || reach == scratch) return POD_E_INVALID;
    const int L = cfg->n_levels;
    if (L < 1 || L > POD_MAX_LEVELS || cfg->num_anchors < 1 || cfg->topk < 1) return POD_E_INVALID;
    pod::SparseGeom G;
    int32_t base = 0;
    for (int l = 0; l < L; ++l) {
        if (levels[l].H < 1 || levels[l].W < 1 || levels[l].anchor_base != base * cfg->num_anchors) return POD_E_INVALID;
        G.H[l] = levels[l].H; G.W[l] = levels[l].W; G.cell_base[l] = base;
        base += levels[l].H * levels[l].W;
    }
    G.cell_base[L] = base; G.n_levels = L; G.A = cfg->num_anchors;
    const hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(reach, 0xFF, (size_t)base, st) != hipSuccess) return POD_E_LAUNCH;
    const int cap = L * cfg->topk;
    hipLaunchKernelGGL(pod::k_sparse_mark, dim3((cap + 255) / 256), dim3(256), 0, st, cat_keys, cat_level, n_total, cap, G, reach);
    POD_CHECK_LAUNCH();
    // x pass: scratch (mask) -> reach (as hx); y pass would race in place, so the mask is first moved out of the way: mask lives in `reach`,
    // the x pass writes `scratch`, the y pass reads it and writes `reach`
    hipLaunchKernelGGL(pod::k_sparse_reach_x, dim3((base + 255) / 256), dim3(256), 0, st, reach, scratch, G);
    POD_CHECK_LAUNCH();
    hipLaunchKernelGGL(pod::k_sparse_reach_y, dim3((base + 255) / 256), dim3(256), 0, st, scratch, reach, G);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

// live (device, POD_SPARSE_LIVE_HEAD + POD_SPARSE_LIVE_STRIDE n_records int32): [0] <- number of live records, then one entry per live
// record (any order): {record index, 11 words of need bits of its 18 x 18 input patch} -- for the blocks of a pod_wino_conv3x3 table
// (`records`: n_records int32x4) whose record r belongs to level rec_level[r].  max_reach 0 .. 5: reach of the launch's OUTPUT (see above);
// in_reach: cells of the launch's INPUT with a larger reach are read as 0.0 (normally max_reach + 1: what the layer below computed for
// this image; 255: the input is dense, e.g. the FPN features in front of the subnet's first convolution).
extern "C" int pod_sparse_live_blocks(const PodConfig* cfg, const PodLevel* levels, const int32_t* records, const int32_t* rec_level, int32_t n_records,
                                      const uint8_t* reach, int32_t max_reach, int32_t in_reach, int32_t* live, pod_stream_t stream) {
    if (!cfg || !levels || !records || !rec_level || !reach || !live || n_records < 0 || max_reach < 0 || max_reach > pod::SPARSE_MAX_REACH ||
        in_reach < max_reach || in_reach > 255 || (reinterpret_cast<uintptr_t>(records) & 15u) != 0 || (reinterpret_cast<uintptr_t>(live) & 15u) != 0)
        return POD_E_INVALID;
    const int L = cfg->n_levels;
    if (L < 1 || L > POD_MAX_LEVELS) return POD_E_INVALID;
    pod::SparseGeom G;
    int32_t base = 0;
    for (int l = 0; l < L; ++l) {
        G.H[l] = levels[l].H; G.W[l] = levels[l].W; G.cell_base[l] = base;
        base += levels[l].H * levels[l].W;
    }
    G.cell_base[L] = base; G.n_levels = L; G.A = cfg->num_anchors;
    const hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(live, 0, 4 * POD_SPARSE_LIVE_HEAD, st) != hipSuccess) return POD_E_LAUNCH;
    if (n_records == 0) return POD_OK;
    hipLaunchKernelGGL(pod::k_sparse_live, dim3((unsigned)n_records), dim3(256), 0, st, reinterpret_cast<const int4*>(records), rec_level, G, reach, max_reach,
                       in_reach, live);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
