// K4 nms_cluster -- class-aware NMS of the candidate list.
//
// Replaces: detectron2.layers.batched_nms -> torchvision.ops.batched_nms (coordinate trick) ->
// torchvision nms, at the reference call sites probabilistic_inference.py:554-560,
// inference_utils.py:31-36, :83-89 (and :269-274 for the post-NMS ensemble merge):
//     boxes_for_nms = boxes + class * (boxes.max() + 1)        (fp32, rounds the shifted coordinates)
//     visit in stable descending-score order; area = (x2-x1)*(y2-y1);
//     suppress j iff inter / (area_i + area_j - inter) > thr   (strict)
//     keep[: max_detections]
//
// The greedy sweep is a serial chain (one LDS round trip + one barrier per survivor, ~0.65 us each), so the
// kernel is latency-bound, not bandwidth-bound.  Two launches:
//
//   k4_class_sweep, one 1024-thread workgroup PER CLASS (+ one that checks class independence, below).  The coordinate trick exists to make classes
//   independent; as long as the shifted boxes of different classes cannot overlap, each class can be swept on its
//   own CU and the chain is num_classes times shorter.  A workgroup (1) max/min-reduces the coordinates (offset
//   unit; independence test), (2) compacts the keys of its class into LDS, (3) orders them: rank sort straight out
//   of LDS for <= 1024 members (every thread counts the keys above its own, no barriers), LDS bitonic network
//   above that, (4) reuses the 128 KiB key pool for the sorted, class-shifted boxes (any n <= 8192 fits, no global
//   scratch) while every thread also keeps ITS boxes (sorted position = slot * 1024 + tid) in registers,
//   (5) sweeps with an LDS "removed" bitmap and ONE barrier per survivor: every wave finds the next survivor
//   itself (wave-uniform ffs scan), tests its register boxes against it (one broadcast LDS read) and publishes its
//   removed bits with one ballot store per slot -- no atomics.  A wave that runs ahead only adds bits above the
//   survivor the others are still looking for, so their scan result cannot change.  The sweep stops at
//   max_detections survivors (never the n x n matrix); each class's survivor list goes to scratch.
//
//   k4_merge, one workgroup: merges the per-class lists by (score, index) key rank and writes keep[:max_det].
//
// Exactness: the trick separates classes c < d only where  x2(A) + off[c] <= x1(B) + off[d]  (or the same in y)
// for A in c, B in d: then the clamped intersection is 0 in fp32 too (rounding is monotone) and A, B never
// interact.  Decoded boxes hang out of the frame, so a few pairs can violate both: A must reach within
// g = min(off[c+1] - off[c]) of the lowest x1 AND y1, B must start g below the highest x2 AND y2.  One extra
// workgroup of k4_class_sweep lists those boxes (normally none or a handful), evaluates the reference's shifted
// IoU for every cross-class A x B pair and raises a flag only if one exceeds the threshold (or a class id is
// outside [0, num_classes), or the lists are implausibly long).  With the flag up k4_merge runs the whole list
// through one workgroup with the reference's per-class offsets -- same code, bit-identical keep list, longer chain.
#include "pod_device.h"

namespace pod {

constexpr int NMS_THREADS = 1024;
constexpr int NMS_SLOTS = POD_MAX_CANDIDATES / NMS_THREADS;   // at most 8 boxes per thread
constexpr int NMS_LIST = POD_MAX_DETECTIONS;                   // per-class survivor list stride
constexpr int NMS_CHECK_EVERY = 8;                             // survivors between two early-stop checks of a class sweep

struct K4Params {
    const int32_t* n_total;
    int32_t n_capacity, max_det, num_classes;
    float thr;
    const float* boxes;
    const float* scores;
    const int32_t* classes;
    int32_t* keep;
    int32_t* n_keep;
    int32_t* flag;        // scratch[0]: 1 = classes may interact, run the single-workgroup sweep
    int32_t* cls_count;   // scratch: POD_MAX_CLASSES survivor counts
    uint64_t* cls_keys;   // scratch: POD_MAX_CLASSES x NMS_LIST (score, ~index) keys of a class's survivors, descending
    int32_t* gen;         // scratch[1]: call generation, bumped by k4_merge; tags the published survivor scores below
    uint64_t* pub;        // scratch: POD_MAX_CLASSES x NMS_LIST entries (gen << 32 | score bits) of the survivors found so far
};

struct NmsLds {
    unsigned char pool[POD_MAX_CANDIDATES * sizeof(float4)];   // 128 KiB: keys while sorting, sorted boxes after
    uint16_t order[POD_MAX_CANDIDATES];                        // sorted position -> candidate index
    unsigned long long removed[POD_MAX_CANDIDATES / 64];
    float red[5][NMS_THREADS / 64];
    int n_members;
    int ahead;            // early-stop check: survivors of the other classes that outrank this class's next one
};

struct NmsExtent {
    float max_all, min_x1, max_x2, min_y1, max_y2;
};

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
// readfirstlane returns int: go through uint32_t, or bit 31 of the low half sign-extends over the high half
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return ((unsigned long long)uniform_u32((uint32_t)(v >> 32)) << 32) | (unsigned long long)uniform_u32((uint32_t)v);
}

// What a thread fetched for candidate `tid` in the kernel's FIRST round trip, before the candidate count was known
// (row tid < n_capacity exists whatever n is): these kernels are chains of dependent HBM / L2 round trips with little
// arithmetic in between, so the loads that do not depend on n travel with the load of n.
struct NmsFirst {
    float4 box;
    float score;
    int cls;
    uint32_t gen;
};

__device__ __forceinline__ NmsFirst nms_first(const K4Params& P) {
    NmsFirst f;
    const int tid = threadIdx.x;
    const bool in = tid < P.n_capacity;
    f.box = in ? *reinterpret_cast<const float4*>(P.boxes + (size_t)tid * 4) : float4{0.f, 0.f, 0.f, 0.f};
    f.score = in ? P.scores[tid] : 0.0f;
    f.cls = in ? P.classes[tid] : -1;
    f.gen = P.gen ? (uint32_t)*P.gen : 0u;     // stable during k4_class_sweep: only k4_merge advances it
    return f;
}

// coordinate extremes of the n candidate boxes (every thread returns the same values)
__device__ NmsExtent nms_extent(NmsLds& S, const float* boxes, int n, const NmsFirst& first) {
    const int tid = threadIdx.x;
    float mx = -INFINITY, nx1 = INFINITY, xx2 = -INFINITY, ny1 = INFINITY, xy2 = -INFINITY;
    for (int i = tid; i < n; i += NMS_THREADS) {
        const float4 b = i == tid ? first.box : *reinterpret_cast<const float4*>(boxes + (size_t)i * 4);
        mx = fmaxf(mx, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
        nx1 = fminf(nx1, b.x);
        ny1 = fminf(ny1, b.y);
        xx2 = fmaxf(xx2, b.z);
        xy2 = fmaxf(xy2, b.w);
    }
    mx = wave_max(mx);
    nx1 = wave_min(nx1);
    xx2 = wave_max(xx2);
    ny1 = wave_min(ny1);
    xy2 = wave_max(xy2);
    if ((tid & 63) == 0) {
        const int w = tid >> 6;
        S.red[0][w] = mx;
        S.red[1][w] = nx1;
        S.red[2][w] = xx2;
        S.red[3][w] = ny1;
        S.red[4][w] = xy2;
    }
    __syncthreads();
    NmsExtent e{S.red[0][0], S.red[1][0], S.red[2][0], S.red[3][0], S.red[4][0]};
    for (int w = 1; w < NMS_THREADS / 64; ++w) {
        e.max_all = fmaxf(e.max_all, S.red[0][w]);
        e.min_x1 = fminf(e.min_x1, S.red[1][w]);
        e.max_x2 = fmaxf(e.max_x2, S.red[2][w]);
        e.min_y1 = fminf(e.min_y1, S.red[3][w]);
        e.max_y2 = fmaxf(e.max_y2, S.red[4][w]);
    }
    return e;
}

// Greedy NMS of the members of `only_class` (or of every candidate when only_class < 0) by one workgroup.
// Survivors, in descending (score, ~index) order, at most max_det: candidate indices to out_idx and / or their keys to
// out_keys (either may be null); the count to *out_count.
__device__ void nms_block(NmsLds& S, const K4Params& P, int n, float shift_unit, int only_class, const NmsFirst& first,
                          int32_t* out_idx, uint64_t* out_keys, int32_t* out_count) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t* const s_keys = reinterpret_cast<uint64_t*>(S.pool);                                            // first 64 KiB
    uint64_t* const s_sorted = reinterpret_cast<uint64_t*>(S.pool + POD_MAX_CANDIDATES * sizeof(uint64_t));  // rank-sort target
    float4* const s_box = reinterpret_cast<float4*>(S.pool);                                                 // after the sort
    // (2) keys of the members, compacted (arrival order does not matter: they are sorted next, and are distinct)
    if (tid == 0) S.n_members = 0;
    __syncthreads();
#pragma unroll 1
    for (int base = 0; base < n; base += NMS_THREADS) {
        const int i = base + tid;
        int cls = -1;
        if (i < n) cls = base == 0 ? first.cls : P.classes[i];
        const bool mine = i < n && (only_class < 0 || cls == only_class);
        const unsigned long long mask = __ballot(mine);
        int at = 0;
        if (lane == 0 && mask) at = atomicAdd(&S.n_members, __popcll(mask));
        at = (int)uniform_u32((uint32_t)at);
        if (mine) s_keys[at + __popcll(mask & ((1ull << lane) - 1ull))] = make_key(base == 0 ? first.score : P.scores[i], i);
    }
    __syncthreads();
    const int m = S.n_members;
    if (m == 0) {
        if (tid == 0) *out_count = 0;
        return;
    }
    // (3) order by (score desc, index asc)
    const uint64_t* sorted = s_keys;
    if (m <= 1024) {
        if (tid < m) {
            const uint64_t mine = s_keys[tid];
            int rank = 0;
#pragma unroll 16
            for (int i = 0; i < m; ++i) rank += (s_keys[i] > mine) ? 1 : 0;   // broadcast LDS reads
            s_sorted[rank] = mine;
        }
        __syncthreads();
        sorted = s_sorted;
    } else {
        int m_sort = 2048;
        while (m_sort < m) m_sort <<= 1;
        for (int i = m + tid; i < m_sort; i += NMS_THREADS) s_keys[i] = 0ull;
        __syncthreads();
        for (int k = 2; k <= m_sort; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < m_sort; i += NMS_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const uint64_t a = s_keys[i], b = s_keys[ixj];
                        const bool desc = (i & k) == 0;
                        if (desc ? (a < b) : (a > b)) {
                            s_keys[i] = b;
                            s_keys[ixj] = a;
                        }
                    }
                }
                __syncthreads();
            }
    }
    // (4) sorted position -> candidate index into registers, then the pool changes hands
    int ord[NMS_SLOTS];
#pragma unroll
    for (int sl = 0; sl < NMS_SLOTS; ++sl) {
        const int pos = sl * NMS_THREADS + tid;
        ord[sl] = pos < m ? key_index(sorted[pos]) : -1;
    }
    __syncthreads();
    float4 bx[NMS_SLOTS];
    float ar[NMS_SLOTS];
    unsigned dead = 0;   // bit sl: the box of slot sl is suppressed (or does not exist)
#pragma unroll
    for (int sl = 0; sl < NMS_SLOTS; ++sl) {
        const int pos = sl * NMS_THREADS + tid;
        bx[sl] = float4{0.f, 0.f, 0.f, 0.f};
        ar[sl] = 0.f;
        if (ord[sl] >= 0) {
            const int idx = ord[sl];
            const float4 b = *reinterpret_cast<const float4*>(P.boxes + (size_t)idx * 4);
            const float off = (float)P.classes[idx] * shift_unit;        // boxes + class * (max + 1)
            bx[sl] = float4{b.x + off, b.y + off, b.z + off, b.w + off};
            ar[sl] = (bx[sl].z - bx[sl].x) * (bx[sl].w - bx[sl].y);
            s_box[pos] = bx[sl];
            S.order[pos] = (uint16_t)idx;
        } else {
            dead |= 1u << sl;
        }
        const unsigned long long word = __ballot((dead >> sl) & 1u);
        if (lane == 0) S.removed[sl * 16 + wave] = word;   // pos >> 6 == sl * 16 + wave
    }
    __syncthreads();
    // (5) greedy sweep
    const int nwords = (m + 63) >> 6;
    int cur = -1, kept = 0;
    const bool publish = only_class >= 0 && P.pub != nullptr;
    const uint32_t gen = first.gen;
    while (true) {
        int next = -1;
        const int start = cur + 1;
        for (int w = start >> 6; w < nwords; ++w) {
            unsigned long long live = ~uniform_u64(S.removed[w]);
            if (w == (start >> 6)) live &= ~0ull << (start & 63);
            if (live) {
                next = (w << 6) + __ffsll((long long)live) - 1;   // < m: the bits at and above m are set
                break;
            }
        }
        if (next < 0) break;
        if (publish && kept > 0 && (kept % NMS_CHECK_EVERY) == 0) {
            // Only max_det detections survive OVERALL (keep[:max_det] of the score-ordered union).  The class lists are
            // swept in descending score, so once `ahead` survivors of the OTHER classes are known to outrank this class's
            // next survivor and kept + ahead >= max_det, neither it nor anything after it can reach keep[:max_det]: stop.
            // The other classes' survivor scores arrive through device-scope stores; a stale view only delays the stop.
            // (On inputs where every class fills up -- 650 members each -- this ends a class after ~16-24 survivors instead
            // of max_det: the sweep is a serial chain of ~0.65 us per survivor.)
            const float s_next = P.scores[S.order[next]];
            if (tid == 0) S.ahead = 0;
            __syncthreads();
            int mine = 0;
            for (int e = tid; e < POD_MAX_CLASSES * NMS_LIST; e += NMS_THREADS) {
                if (e / NMS_LIST == only_class) continue;
                const uint64_t v = __hip_atomic_load(P.pub + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(v >> 32) == gen && __uint_as_float((uint32_t)v) > s_next) ++mine;
            }
            mine = wave_sum(mine);
            if (lane == 0 && mine) atomicAdd(&S.ahead, mine);
            __syncthreads();
            if (kept + S.ahead >= P.max_det) break;
        }
        if (tid == 0) {
            const int idx = S.order[next];
            const float sc = P.scores[idx];
            if (out_idx) out_idx[kept] = idx;
            if (out_keys) out_keys[kept] = make_key(sc, idx);
            if (publish)
                __hip_atomic_store(P.pub + only_class * NMS_LIST + kept, ((uint64_t)gen << 32) | __float_as_uint(sc),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++kept;
        cur = next;
        if (kept >= P.max_det) break;   // survivor list full: the rest of the sweep cannot change keep[:max_det]
        const float4 bi = s_box[next];
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
#pragma unroll
        for (int sl = 0; sl < NMS_SLOTS; ++sl) {
            if ((sl + 1) * NMS_THREADS <= next + 1 || sl * NMS_THREADS >= m) continue;   // uniform: nothing later in this slot
            const int pos = sl * NMS_THREADS + tid;
            if (pos > next && !((dead >> sl) & 1u)) {
                const float w = fmaxf(0.0f, fminf(bi.z, bx[sl].z) - fmaxf(bi.x, bx[sl].x));
                const float h = fmaxf(0.0f, fminf(bi.w, bx[sl].w) - fmaxf(bi.y, bx[sl].y));
                const float inter = w * h;
                const float ovr = __fdiv_rn(inter, (ai + ar[sl]) - inter);
                if (ovr > P.thr) dead |= 1u << sl;
            }
            const unsigned long long word = __ballot((dead >> sl) & 1u);
            if (lane == 0) S.removed[sl * 16 + wave] = word;
        }
        __syncthreads();
    }
    if (tid == 0) *out_count = kept;
}

// Workgroup num_classes of k4_class_sweep: can boxes of different classes suppress each other?
__device__ void nms_independence_check(NmsLds& S, const K4Params& P, int n, const NmsExtent& e, float shift_unit) {
    constexpr int CAP = 2048;
    const int tid = threadIdx.x;
    uint16_t* const list_a = reinterpret_cast<uint16_t*>(S.pool);
    uint16_t* const list_b = list_a + CAP;
    int* const cnt = reinterpret_cast<int*>(S.removed);   // [0] |A|, [1] |B|, [2] flag
    if (tid < 3) cnt[tid] = 0;
    __syncthreads();
    double gap = INFINITY;
    for (int k = 0; k + 1 < P.num_classes; ++k)
        gap = fmin(gap, (double)((float)(k + 1) * shift_unit) - (double)((float)k * shift_unit));
    if (!(gap > 0.0)) cnt[2] = 1;   // also catches a NaN unit
    const double a_x = gap + (double)e.min_x1, a_y = gap + (double)e.min_y1;   // A: x2 > a_x and y2 > a_y
    const double b_x = (double)e.max_x2 - gap, b_y = (double)e.max_y2 - gap;   // B: x1 < b_x and y1 < b_y
    for (int i = tid; i < n; i += NMS_THREADS) {
        const float4 b = *reinterpret_cast<const float4*>(P.boxes + (size_t)i * 4);
        const int cls = P.classes[i];
        if (cls < 0 || cls >= P.num_classes) cnt[2] = 1;
        if ((double)b.z > a_x && (double)b.w > a_y) {
            const int at = atomicAdd(&cnt[0], 1);
            if (at < CAP) list_a[at] = (uint16_t)i;
        }
        if ((double)b.x < b_x && (double)b.y < b_y) {
            const int at = atomicAdd(&cnt[1], 1);
            if (at < CAP) list_b[at] = (uint16_t)i;
        }
    }
    __syncthreads();
    const int na = cnt[0], nb = cnt[1];
    if (na > CAP || nb > CAP || (long long)na * nb > (1 << 18)) {
        if (tid == 0) *P.flag = 1;   // not worth enumerating: take the single-workgroup sweep
        return;
    }
    for (int p = tid; p < na * nb; p += NMS_THREADS) {
        const int ia = list_a[p / nb], ib = list_b[p % nb];
        const int ca = P.classes[ia], cb = P.classes[ib];
        if (ca == cb) continue;
        const float4 ra = *reinterpret_cast<const float4*>(P.boxes + (size_t)ia * 4);
        const float4 rb = *reinterpret_cast<const float4*>(P.boxes + (size_t)ib * 4);
        const float oa = (float)ca * shift_unit, ob = (float)cb * shift_unit;
        const float4 A{ra.x + oa, ra.y + oa, ra.z + oa, ra.w + oa}, B{rb.x + ob, rb.y + ob, rb.z + ob, rb.w + ob};
        const float w = fmaxf(0.0f, fminf(A.z, B.z) - fmaxf(A.x, B.x));
        const float h = fmaxf(0.0f, fminf(A.w, B.w) - fmaxf(A.y, B.y));
        const float inter = w * h;
        const float ovr = __fdiv_rn(inter, ((A.z - A.x) * (A.w - A.y) + (B.z - B.x) * (B.w - B.y)) - inter);
        if (ovr > P.thr) cnt[2] = 1;
    }
    __syncthreads();
    if (tid == 0) *P.flag = cnt[2];
}

__global__ void __launch_bounds__(NMS_THREADS) k4_class_sweep(const K4Params P) {
    __shared__ NmsLds S;
    const int n_raw = *P.n_total;
    const NmsFirst first = nms_first(P);            // same round trip as the count
    const int n = min(n_raw, P.n_capacity);
    const int c = blockIdx.x;
    if (n <= 0) {
        if (threadIdx.x == 0) {
            if (c < P.num_classes) P.cls_count[c] = 0;
            else *P.flag = 0;
        }
        return;
    }
    const NmsExtent e = nms_extent(S, P.boxes, n, first);
    const float shift_unit = e.max_all + 1.0f;
    if (c < P.num_classes) nms_block(S, P, n, shift_unit, c, first, nullptr, P.cls_keys + c * NMS_LIST, P.cls_count + c);
    else nms_independence_check(S, P, n, e, shift_unit);
}

__global__ void __launch_bounds__(NMS_THREADS) k4_merge(const K4Params P) {
    __shared__ NmsLds S;
    const int tid = threadIdx.x;
    // one round trip: the count, the interaction flag, the generation, the per-class survivor counts
    const int n_raw = *P.n_total;
    const int flag = *P.flag;
    const int gen = P.gen ? *P.gen : 0;
    const int my_cnt = (tid < 64 && tid < P.num_classes) ? P.cls_count[tid] : 0;
    const int n = min(n_raw, P.n_capacity);
    if (n <= 0) {
        if (tid == 0) *P.n_keep = 0;
        return;
    }
    if (tid == 0 && P.gen) *P.gen = gen + 1;   // the next call's published scores carry a new tag (stale entries never match)
    if (flag) {   // classes may interact: the reference's sweep over the whole list, one workgroup
        const NmsFirst first = nms_first(P);
        const NmsExtent e = nms_extent(S, P.boxes, n, first);
        nms_block(S, P, n, e.max_all + 1.0f, -1, first, P.keep, nullptr, P.n_keep);
        return;
    }
    // merge the per-class survivor lists: position = number of survivors with a larger (score, ~index) key
    uint64_t* const s_keys = reinterpret_cast<uint64_t*>(S.pool);
    int* const s_begin = reinterpret_cast<int*>(S.removed);   // POD_MAX_CLASSES + 1 ints
    if (tid < 64) {   // prefix of the counts by a wavefront scan
        int incl = my_cnt;
#pragma unroll
        for (int o = 1; o < POD_MAX_CLASSES; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (tid >= o) incl += up;
        }
        if (tid <= P.num_classes) s_begin[tid] = incl - my_cnt;    // entry num_classes = total (its count is 0)
    }
    __syncthreads();
    const int total = s_begin[P.num_classes];
    for (int c = 0; c < P.num_classes; ++c) {
        const int cnt = s_begin[c + 1] - s_begin[c];
        if (tid < cnt) s_keys[s_begin[c] + tid] = P.cls_keys[c * NMS_LIST + tid];   // the sweeps left keys: no score gather
    }
    __syncthreads();
    for (int q = tid; q < total; q += NMS_THREADS) {
        const uint64_t mine = s_keys[q];
        int rank = 0;
        for (int i = 0; i < total; ++i) rank += (s_keys[i] > mine) ? 1 : 0;
        if (rank < P.max_det) P.keep[rank] = key_index(mine);
    }
    if (tid == 0) *P.n_keep = min(total, P.max_det);
}

}  // namespace pod

extern "C" size_t pod_nms_scratch_bytes(int32_t n_capacity) {
    if (n_capacity < 1) return 0;
    // flag + generation, counts, per-class survivor lists, published survivor scores
    return 256 + 2 * sizeof(uint64_t) * (size_t)POD_MAX_CLASSES * pod::NMS_LIST;
}

extern "C" int pod_nms_cluster(const PodConfig* cfg, const int32_t* n_total, int32_t n_capacity, const float* boxes,
                               const float* scores, const int32_t* classes, int32_t* keep, int32_t* n_keep, void* scratch,
                               pod_stream_t stream) {
    if (!cfg || !n_total || !boxes || !scores || !classes || !keep || !n_keep || !scratch) return POD_E_INVALID;
    if (n_capacity < 1 || n_capacity > POD_MAX_CANDIDATES) return POD_E_INVALID;
    if (cfg->max_detections < 1 || cfg->max_detections > POD_MAX_DETECTIONS) return POD_E_INVALID;
    if (cfg->num_classes < 1 || cfg->num_classes >= POD_MAX_CLASSES) return POD_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(scratch) & 15u) != 0 || (reinterpret_cast<uintptr_t>(boxes) & 15u) != 0) return POD_E_INVALID;
    pod::K4Params P;
    P.n_total = n_total; P.n_capacity = n_capacity; P.max_det = cfg->max_detections; P.num_classes = cfg->num_classes;
    P.thr = cfg->nms_thresh;
    P.boxes = boxes; P.scores = scores; P.classes = classes; P.keep = keep; P.n_keep = n_keep;
    int32_t* s = static_cast<int32_t*>(scratch);
    P.flag = s;
    P.gen = s + 1;
    P.cls_count = s + 16;
    P.cls_keys = reinterpret_cast<uint64_t*>(s + 64);
    P.pub = P.cls_keys + POD_MAX_CLASSES * pod::NMS_LIST;
    hipLaunchKernelGGL(pod::k4_class_sweep, dim3(cfg->num_classes + 1), dim3(pod::NMS_THREADS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    hipLaunchKernelGGL(pod::k4_merge, dim3(1), dim3(pod::NMS_THREADS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
