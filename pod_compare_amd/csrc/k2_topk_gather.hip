// K2 level_topk + K2b gather_candidates.
//
// Replaces (probabilistic_inference.py): `predicted_prob.topk(num_topk)` and the
// `> test_score_thresh` filter :300-308 (K1 already applied the threshold, which commutes with
// top-k), the per-level gathers :305-338 and the level concatenation :341-342 / :387-388.
//
// K2: one 1024-thread workgroup per FPN level.  Keys are 64-bit (score bits, ~anchor index), all
// distinct, so "top-k, ties to the lower index" is a plain descending sort.  Up to 2048
// candidates are sorted directly in LDS (bitonic network); a level with more candidates first runs
// an 8-pass MSB radix select (LDS histograms) to find the k-th largest key, then compacts.
// K2b: one thread per selected candidate re-derives the class probabilities with K1's own device
// function (bit-identical), and gathers merged deltas / log-variances, the anchor and every run's
// raw delta into the level-concatenated candidate arrays.
#include "pod_candidate.h"

namespace pod {

#ifdef POD_TRACE
__device__ long long g_k2_trace[128 * 16];          // [workgroup][stamp]: wall_clock64 (100 MHz) at the phases of k2_level_topk
#define K2_STAMP(i) do { if (threadIdx.x == 0) g_k2_trace[blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define K2_STAMP(i) do { } while (0)
#endif

constexpr int TOPK_THREADS = 1024;
constexpr int SORT_CAP = POD_MAX_TOPK;   // 2048 keys = 16 KiB LDS

// Descending bitonic sort of s[0 .. n_pow2), n_pow2 <= 2 * TOPK_THREADS, by TOPK_THREADS threads.  Thread t keeps elements t and
// t + 1024 in registers.  Of the (log n)(log n + 1)/2 compare-exchange steps only those with a partner distance of 64..512 go
// through LDS (two barriers each); distances below 64 are wavefront shuffles and distance 1024 is the thread's own pair, so a
// 2048-key sort pays 14 LDS rounds instead of 66 (27 us -> 9 us on one CU).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = __shfl_xor((uint32_t)v, mask, 64), hi = __shfl_xor((uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t bitonic_pick(uint64_t own, uint64_t other, int i, int j, int k) {
    const bool desc = (i & k) == 0, lower = (i & j) == 0;           // lower index of the pair takes the larger key in a desc block
    const uint64_t hi = own > other ? own : other, lo = own > other ? other : own;
    return (desc == lower) ? hi : lo;
}
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* s, int n_pow2, int tid, int nthreads) {
    const bool two = n_pow2 > TOPK_THREADS;
    uint64_t e0 = tid < n_pow2 ? s[tid] : 0ull, e1 = two ? s[tid + TOPK_THREADS] : 0ull;
    const int i0 = tid, i1 = tid + TOPK_THREADS;
    for (int k = 2; k <= n_pow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= TOPK_THREADS) {                 // k = 2048, j = 1024: the thread's own pair (desc block, i0 is the lower index)
                const uint64_t hi = e0 > e1 ? e0 : e1, lo = e0 > e1 ? e1 : e0;
                e0 = hi;
                e1 = lo;
            } else if (j >= 64) {
                __syncthreads();
                if (tid < n_pow2) s[i0] = e0;
                if (two) s[i1] = e1;
                __syncthreads();
                if (tid < n_pow2) e0 = bitonic_pick(e0, s[i0 ^ j], i0, j, k);
                if (two) e1 = bitonic_pick(e1, s[i1 ^ j], i1, j, k);
            } else {
                e0 = bitonic_pick(e0, shfl_xor_u64(e0, j), i0, j, k);
                if (two) e1 = bitonic_pick(e1, shfl_xor_u64(e1, j), i1, j, k);
            }
        }
    }
    __syncthreads();
    if (tid < n_pow2) s[i0] = e0;
    if (two) s[i1] = e1;
    __syncthreads();
}

struct K2Params {
    int32_t anchor_base[POD_MAX_LEVELS];
    int32_t topk;
    uint64_t* cand_keys;      // consumed: a big level's slices are compacted in place
    int32_t* cand_count;      // [n_levels] counts, then [n_levels] tickets (zero on entry, left zero)
    int32_t n_levels;
    uint64_t* sel_keys;
    int32_t* sel_count;
    uint64_t* cat_keys;       // level-concatenated selection: row offset_l + i = sel_keys[l][i]   (may be null)
    int32_t* cat_level;       // level of every row of cat_keys                                     (may be null)
    int32_t* n_total;         // sum over levels of min(count, topk)                                (may be null)
};

constexpr int TOPK_SLICES = 16;   // workgroups per level; only a level with more than SORT_CAP candidates uses more than one

struct TopkLds {
    uint64_t keys[SORT_CAP];
    uint32_t hist[256];
    uint64_t prefix;
    uint32_t wtot[4];
    int32_t remaining, fill, bucket, ticket, pick, pick_rem;
    uint64_t red_or[TOPK_THREADS / 64], red_and[TOPK_THREADS / 64];
    uint32_t red_n[TOPK_THREADS / 64];
};

// Where level l's rows start in the level-concatenated candidate list, and the list's length: the counts of ALL levels are
// final when this kernel starts (K1 / K1b are done) and nobody changes them before the gather kernel consumes them.
__device__ __forceinline__ int level_offset(const K2Params& P, int l, int& total) {
    int off = 0;
    total = 0;
#pragma unroll
    for (int i = 0; i < POD_MAX_LEVELS; ++i) {
        const int ci = i < P.n_levels ? min(P.cand_count[i], P.topk) : 0;
        if (i < l) off += ci;
        total += ci;
    }
    return off;
}

__device__ __forceinline__ void write_selection(const K2Params& P, const TopkLds& S, int l, int k, uint64_t* out, int off, int total) {
    for (int i = threadIdx.x; i < k; i += TOPK_THREADS) {
        const uint64_t key = S.keys[i];
        out[i] = key;
        if (P.cat_keys) {
            P.cat_keys[off + i] = key;
            P.cat_level[off + i] = l;
        }
    }
    if (threadIdx.x == 0) {
        P.sel_count[l] = k;
        if (P.cat_keys && l == 0 && P.n_total) *P.n_total = total;
    }
}

// The `want` largest of the `count` keys fetch(0..count-1) (all distinct, or 0 = absent), sorted descending in S.keys[0..want).
// count <= SORT_CAP: straight into the LDS bitonic network.  Otherwise an MSB radix select (LDS histograms, from the top byte
// down) runs until the keys that can still be in the top `want` fit the sort buffer: after a pass the candidates are {keys
// above the chosen bucket} + {the bucket}; typically 2-4 passes.  8 independent fetches per thread per step.
// SORT = false stops before the sort: S.keys[0 .. returned size) then holds an unordered SUPERSET of the top `want` (at most
// SORT_CAP keys, zero-padded), which is all a slice has to hand to the final selection.
template <bool SORT, class Fetch>
__device__ __forceinline__ int topk_into_lds(TopkLds& S, Fetch fetch, int count, int want) {
    const int tid = threadIdx.x;
    int n_sort;
    if (count <= SORT_CAP) {
        n_sort = 1;
        while (n_sort < count) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += TOPK_THREADS) S.keys[i] = (i < count) ? fetch(i) : 0ull;
        __syncthreads();
    } else {
        if (tid == 0) {
            S.prefix = 0ull;
            S.remaining = want;
            S.fill = 0;
        }
        __syncthreads();
        uint64_t lower_bound = 0ull;     // every key >= lower_bound is still a candidate; there are <= SORT_CAP of them at exit
        for (int pass = 7; pass >= 0; --pass) {
            const int shift = pass * 8;
            if (tid < 256) S.hist[tid] = 0u;
            __syncthreads();
            const uint64_t prefix = S.prefix;
            const uint64_t himask = (pass == 7) ? 0ull : (~0ull << (shift + 8));
            for (int i0 = 0; i0 < count; i0 += TOPK_THREADS * 8) {
                uint64_t kk[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * TOPK_THREADS + tid;
                    kk[u] = (i < count) ? fetch(i) : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * TOPK_THREADS + tid;
                    const bool live = i < count && (kk[u] & himask) == prefix;
                    // top digits of score keys are heavily skewed: one LDS atomic when the whole wavefront agrees
                    const uint32_t digit = (uint32_t)(kk[u] >> shift) & 255u;
                    const unsigned long long lm = __ballot(live);
                    if (lm != 0ull) {
                        const int leader = __ffsll((long long)lm) - 1;
                        const uint32_t d0 = __shfl(digit, leader, 64);
                        if (__ballot(live && digit == d0) == lm) {
                            if ((threadIdx.x & 63) == leader) atomicAdd(&S.hist[d0], (uint32_t)__popcll(lm));
                        } else if (live) {
                            atomicAdd(&S.hist[digit], 1u);
                        }
                    }
                }
            }
            __syncthreads();
            // bucket holding the `remaining`-th largest live key: the largest d with sum_{j >= d} hist[j] >= remaining.  Suffix sums by
            // wavefront scans (a single thread walking the 256 bins through LDS cost ~7 us per pass)
            uint32_t cnt = 0, incl = 0;
            if (tid < 256) {
                cnt = S.hist[tid];
                incl = cnt;
                const int ln = tid & 63;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_down(incl, o, 64);
                    if (ln + o < 64) incl += up;
                }
                if (ln == 0) S.wtot[tid >> 6] = incl;
            }
            __syncthreads();
            if (tid < 256) {
                for (int w = (tid >> 6) + 1; w < 4; ++w) incl += S.wtot[w];
                const uint32_t excl = incl - cnt;                   // live keys in the buckets above this one
                const uint32_t rem = (uint32_t)S.remaining;
                // exactly one bucket qualifies; bucket 0 takes what no higher bucket covers (as a walk from the top would)
                const bool chosen = tid == 0 ? excl < rem : (excl < rem && rem <= incl);
                if (chosen) {
                    S.pick = tid;
                    S.pick_rem = (int)(rem - excl);
                    S.bucket = (int)cnt;
                }
            }
            __syncthreads();
            if (tid == 0) {
                S.remaining = S.pick_rem;                            // still needed from bucket `pick`
                S.prefix = prefix | ((uint64_t)S.pick << shift);
            }
            __syncthreads();
            lower_bound = S.prefix;
            // candidates = (want - remaining) keys above the bucket + the bucket itself
            if ((want - S.remaining) + S.bucket <= SORT_CAP) break;
        }
        const int n_cand = (want - S.remaining) + S.bucket;         // exact count of keys >= lower_bound
        n_sort = 1;
        while (n_sort < n_cand) n_sort <<= 1;
        for (int i = tid; i < n_sort; i += TOPK_THREADS) S.keys[i] = 0ull;
        __syncthreads();
        for (int i0 = 0; i0 < count; i0 += TOPK_THREADS * 8) {
            uint64_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                kk[u] = (i < count) ? fetch(i) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_THREADS + tid;
                // one returning LDS atomic per wavefront and step, not per key (2048 same-address returning atomics: ~40 us)
                const bool sel = i < count && kk[u] >= lower_bound && kk[u] != 0ull;
                const unsigned long long sm = __ballot(sel);
                if (sm != 0ull) {
                    const int ln = threadIdx.x & 63, leader = __ffsll((long long)sm) - 1;
                    int at = 0;
                    if (ln == leader) at = atomicAdd(&S.fill, __popcll(sm));
                    at = __shfl(at, leader, 64);
                    if (sel) S.keys[at + __popcll(sm & ((1ull << ln) - 1ull))] = kk[u];
                }
            }
        }
        __syncthreads();
    }
    if (SORT) bitonic_sort_desc(S.keys, n_sort, tid, TOPK_THREADS);
    return n_sort;
}

// The same selection for count <= CACHE_CAP with every key fetched ONCE: a thread keeps its KPT keys in registers through all
// radix passes and the compaction (the loop version above re-fetches the list per pass: with 9 000 - 16 000 keys that was 3 - 4
// round trips to L2 / HBM per workgroup, twice on the critical path of a big level).  The digits start at the highest bit in
// which two present keys DIFFER (an OR / AND reduction over the workgroup): scores of candidates lie in (threshold, 1], so the
// top 6 bits of every key agree and a fixed byte grid would spend its first pass on them.  Narrows until at most `cap` keys
// (>= want) are left; SORT = false leaves them unordered, zero-padded to `cap`, and returns how many there are.
constexpr int KPT = 16;
constexpr int CACHE_CAP = KPT * TOPK_THREADS;

// How far the select narrows before the bitonic network takes over: the smallest power of two that holds the `want` keys (a
// 2048-key sort costs 20 us on one CU, a 1024-key sort 11, an extra radix pass over registers 2.5).
__device__ __forceinline__ int sort_cap(int want) {
    int c = 256;
    while (c < want) c <<= 1;
    return c;
}

template <bool SORT, class Fetch>
__device__ __forceinline__ int topk_cached(TopkLds& S, Fetch fetch, int count, int want, int cap, int tb = 0) {
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    uint64_t kk[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const int i = u * TOPK_THREADS + tid;
        kk[u] = (i < count) ? fetch(i) : 0ull;
    }
    uint64_t vor = 0ull, vand = ~0ull;
    uint32_t present = 0;
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if (kk[u] != 0ull) {
            vor |= kk[u];
            vand &= kk[u];
            ++present;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vor |= shfl_xor_u64(vor, o);
        vand &= shfl_xor_u64(vand, o);
        present += __shfl_xor(present, o, 64);
    }
    if (ln == 0) {
        S.red_or[wv] = vor;
        S.red_and[wv] = vand;
        S.red_n[wv] = present;
    }
    if (tid == 0) S.fill = 0;
    __syncthreads();
    vor = 0ull, vand = ~0ull, present = 0;
#pragma unroll
    for (int w = 0; w < TOPK_THREADS / 64; ++w) {
        vor |= S.red_or[w];
        vand &= S.red_and[w];
        present += S.red_n[w];
    }
    K2_STAMP(tb);
    const uint64_t diff = vor ^ vand;                    // bits in which two present keys differ (keys are distinct)
    int hi = diff ? 64 - __clzll((long long)diff) : 0;    // every present key agrees on bits >= hi
    uint64_t prefix = hi < 64 ? (vand >> hi) << hi : 0ull;
    int remaining = want, bucket = (int)present;         // candidates = (want - remaining) above the bucket + the bucket
    while ((want - remaining) + bucket > cap && hi > 0) {
        const int shift = hi > 8 ? hi - 8 : 0;
        const uint32_t dmask = (1u << (hi - shift)) - 1u;
        const uint64_t himask = hi < 64 ? (~0ull << hi) : 0ull;
        if (tid < 256) S.hist[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const bool live = kk[u] != 0ull && (kk[u] & himask) == prefix;
            const uint32_t digit = (uint32_t)(kk[u] >> shift) & dmask;
            const unsigned long long lm = __ballot(live);
            if (lm != 0ull) {
                const int leader = __ffsll((long long)lm) - 1;
                const uint32_t d0 = __shfl(digit, leader, 64);
                if (__ballot(live && digit == d0) == lm) {
                    if (ln == leader) atomicAdd(&S.hist[d0], (uint32_t)__popcll(lm));
                } else if (live) {
                    atomicAdd(&S.hist[digit], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t cnt = 0, incl = 0;
        if (tid < 256) {
            cnt = S.hist[tid];
            incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_down(incl, o, 64);
                if (ln + o < 64) incl += up;
            }
            if (ln == 0) S.wtot[wv] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            for (int w = wv + 1; w < 4; ++w) incl += S.wtot[w];
            const uint32_t excl = incl - cnt, rem = (uint32_t)remaining;
            const bool chosen = tid == 0 ? excl < rem : (excl < rem && rem <= incl);
            if (chosen) {
                S.pick = tid;
                S.pick_rem = (int)(rem - excl);
                S.bucket = (int)cnt;
            }
        }
        __syncthreads();
        remaining = S.pick_rem;
        bucket = S.bucket;
        prefix |= (uint64_t)S.pick << shift;
        hi = shift;
    }
    K2_STAMP(tb + 1);
    const int n_cand = (want - remaining) + bucket;            // exact count of present keys >= prefix
    int n_sort = 1;
    while (n_sort < n_cand) n_sort <<= 1;
    if (n_sort > SORT_CAP) n_sort = SORT_CAP;                 // only a caller's duplicate keys get here (distinct keys narrow to <= cap): the excess is dropped
    const int n_clear = SORT ? n_sort : (cap < SORT_CAP ? cap : SORT_CAP);
    for (int i = tid; i < n_clear; i += TOPK_THREADS) S.keys[i] = 0ull;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const bool sel = kk[u] != 0ull && kk[u] >= prefix;
        const unsigned long long sm = __ballot(sel);
        if (sm != 0ull) {
            const int leader = __ffsll((long long)sm) - 1;
            int at = 0;
            if (ln == leader) at = atomicAdd(&S.fill, __popcll(sm));
            at = __shfl(at, leader, 64);
            const int slot = at + __popcll(sm & ((1ull << ln) - 1ull));
            if (sel && slot < n_clear) S.keys[slot] = kk[u];      // (always true for distinct keys; a caller's duplicates must not write past the buffer)
        }
    }
    __syncthreads();
    K2_STAMP(tb + 2);
    if (SORT) {
        bitonic_sort_desc(S.keys, n_sort, tid, TOPK_THREADS);
        K2_STAMP(tb + 3);
        return n_sort;
    }
    return n_cand;
}

// Grid = n_levels x TOPK_SLICES workgroups.  A level with <= SORT_CAP candidates (every level of a typical image) is sorted by
// its slice-0 workgroup alone.  A bigger level is cut into TOPK_SLICES slices: the global top-k is contained in the union of the
// slices' top-k, so every workgroup narrows its slice down to <= 2048 candidates containing the slice's top-k (16x shorter scans,
// on 16 CUs), writes them back over the head of its own slice and takes a ticket; the workgroup that draws the last ticket selects
// and sorts the final top-k from the <= 32k survivors.
// No spinning: the others simply exit.
__global__ void __launch_bounds__(TOPK_THREADS) k2_level_topk(const K2Params P) {
    __shared__ TopkLds S;
    const int l = blockIdx.x / TOPK_SLICES, b = blockIdx.x % TOPK_SLICES;
    const int tid = threadIdx.x;
    uint64_t* keys = P.cand_keys + P.anchor_base[l];
    int32_t* ticket = P.cand_count + P.n_levels + l;
    K2_STAMP(0);
    const int C = P.cand_count[l];           // stays put: the gather kernel (K2b / K23) consumes (re-zeroes) the counts
    int total = 0;
    const int off = P.cat_keys ? level_offset(P, l, total) : 0;      // the other levels' counts: same round trip as C
    K2_STAMP(1);
    const int k = min(P.topk, C);
    uint64_t* out = P.sel_keys + (int64_t)l * P.topk;
    if (C <= SORT_CAP) {
        if (b != 0) return;
        topk_into_lds<true>(S, [=](int i) { return keys[i]; }, C, k);
        write_selection(P, S, l, k, out, off, total);
        return;
    }
    if (C <= CACHE_CAP) {                 // one workgroup holds the whole level in registers: no slices, no ticket
        if (b != 0) return;
        topk_cached<true>(S, [=](int i) { return keys[i]; }, C, k, sort_cap(k));
        write_selection(P, S, l, k, out, off, total);
        return;
    }
    const int slice = ((C + TOPK_SLICES - 1) / TOPK_SLICES + 7) & ~7;      // keys per slice
    const int begin = b * slice;
    const int len = max(0, min(slice, C - begin));
    const bool cached = slice <= CACHE_CAP;                                // else the loop version (levels beyond 262 144 candidates)
    // survivors of a slice = an unordered superset of its top-k, at most `cap` keys, written over the head of the slice and
    // zero-padded to min(len, cap); a slice that short is its own survivor list.  Sorting here would only be redone below.
    const int cap = (cached && k <= SORT_CAP / 2) ? SORT_CAP / 2 : SORT_CAP;
    if (len > cap) {
        const uint64_t* mine = keys + begin;
        const int n_out = cached ? topk_cached<false>(S, [=](int i) { return mine[i]; }, len, min(k, len), cap, 2)
                                 : topk_into_lds<false>(S, [=](int i) { return mine[i]; }, len, min(k, len));
        for (int i = tid; i < cap; i += TOPK_THREADS) keys[begin + i] = i < n_out ? S.keys[i] : 0ull;   // only this workgroup touches the slice
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) S.ticket = atomicAdd(ticket, 1);
    __syncthreads();
    K2_STAMP(6);
    if (S.ticket != TOPK_SLICES - 1) return;
    __threadfence();
    K2_STAMP(7);
    // last workgroup of the level: virtual list j -> entry (j % cap) of slice (j / cap)'s survivor list; 0 = no key
    auto survivors = [=](int j) -> uint64_t {
        const int sb = j / cap, r = j - sb * cap;
        const int sbegin = sb * slice;
        const int slen = max(0, min(slice, C - sbegin));
        // device-scope load: this CU's L1 may still hold the line from before the other workgroup compacted its slice
        return r < min(cap, slen) ? __hip_atomic_load(keys + sbegin + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    };
    if (TOPK_SLICES * cap <= CACHE_CAP)
        topk_cached<true>(S, survivors, TOPK_SLICES * cap, k, sort_cap(k), 8);
    else
        topk_into_lds<true>(S, survivors, TOPK_SLICES * cap, k);
    write_selection(P, S, l, k, out, off, total);
    if (tid == 0) *ticket = 0;
    K2_STAMP(12);
}

#ifdef POD_TRACE
}  // namespace pod
extern "C" int pod_k2_trace_dump(long long* host) {   // diagnostics build only (not in include/pod_mi355x.h): 128 x 16 stamps
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pod::g_k2_trace), sizeof(long long) * 128 * 16) == hipSuccess ? POD_OK : POD_E_LAUNCH;
}
namespace pod {
#endif

__global__ void __launch_bounds__(64) k2b_gather(const K2bParams P) {
    GatheredCandidate g;
    gather_candidate(P, blockIdx.x, threadIdx.x, nullptr, g);
}

}  // namespace pod

extern "C" int pod_level_topk(const PodConfig* cfg, const PodLevel* levels, uint64_t* cand_keys,
                              int32_t* cand_count, uint64_t* sel_keys, int32_t* sel_count, uint64_t* cat_keys,
                              int32_t* cat_level, int32_t* n_total, pod_stream_t stream) {
    if (!cfg || !levels || !cand_keys || !cand_count || !sel_keys || !sel_count) return POD_E_INVALID;
    if ((cat_keys != nullptr) != (cat_level != nullptr) || (cat_keys && !n_total)) return POD_E_INVALID;
    if (cfg->n_levels < 1 || cfg->n_levels > POD_MAX_LEVELS || cfg->topk < 1 || cfg->topk > POD_MAX_TOPK) return POD_E_INVALID;
    pod::K2Params P;
    for (int l = 0; l < cfg->n_levels; ++l) P.anchor_base[l] = levels[l].anchor_base;
    P.topk = cfg->topk; P.cand_keys = cand_keys; P.cand_count = cand_count; P.n_levels = cfg->n_levels; P.sel_keys = sel_keys;
    P.sel_count = sel_count; P.cat_keys = cat_keys; P.cat_level = cat_level; P.n_total = n_total;
    hipLaunchKernelGGL(pod::k2_level_topk, dim3(cfg->n_levels * pod::TOPK_SLICES), dim3(pod::TOPK_THREADS), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_gather_candidates(const PodConfig* cfg, const PodLevel* levels, const float* anchors,
                                     const uint64_t* cat_keys, const int32_t* cat_level, const int32_t* n_total,
                                     int32_t* cand_count, const float* probs_dense, int32_t* cand_anchor_idx,
                                     int32_t* cand_level, float* cand_score, int32_t* cand_class, float* cand_probs,
                                     float* cand_delta, float* cand_reg_var, float* cand_anchor, float* cand_run_delta,
                                     pod_stream_t stream) {
    if (!cfg || !levels || !anchors || !cat_keys || !cat_level || !n_total || !cand_count || !cand_anchor_idx || !cand_level ||
        !cand_score || !cand_class || !cand_probs || !cand_delta || !cand_anchor)
        return POD_E_INVALID;
    if (cfg->cov_dims > 0 && !cand_reg_var) return POD_E_INVALID;
    if (cfg->n_levels * cfg->topk > POD_MAX_CANDIDATES * 4) return POD_E_INVALID;
    if (2 * cfg->num_classes + 4 + cfg->cov_dims > 64) return POD_E_INVALID;
    pod::K2bParams P;
    for (int l = 0; l < cfg->n_levels; ++l) P.lv[l] = levels[l];
    P.n_levels = cfg->n_levels; P.n_runs = cfg->n_runs; P.A = cfg->num_anchors; P.K = cfg->num_classes; P.D = cfg->cov_dims;
    P.has_cls_var = cfg->has_cls_var; P.quirk = cfg->merge_quirk; P.cls_samples = cfg->cls_samples; P.topk = cfg->topk;
    P.seed = cfg->philox_seed; P.anchors = anchors; P.cat_keys = cat_keys; P.cat_level = cat_level; P.n_total = n_total;
    P.cand_count = cand_count; P.probs_dense = probs_dense;
    P.cand_anchor_idx = cand_anchor_idx; P.cand_level = cand_level; P.cand_score = cand_score; P.cand_class = cand_class;
    P.cand_probs = cand_probs; P.cand_delta = cand_delta; P.cand_reg_var = cand_reg_var; P.cand_anchor = cand_anchor;
    P.cand_run_delta = cfg->n_runs > 1 ? cand_run_delta : nullptr;
    const int slots = cfg->n_levels * cfg->topk;
    hipLaunchKernelGGL(pod::k2b_gather, dim3(slots), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
