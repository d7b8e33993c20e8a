// Shared by the two Winograd convolution kernels (k11_wino_conv.hip: fp32 MFMA; k12_wino_conv_split.hip: the same convolution with
// every fp32 product formed from 3-way bf16 splits on the bf16 matrix cores): parameters, LDS geometry, the slot table of the stage
// fills, wait immediates, diagnostics hooks.
#pragma once
#include <mutex>
#include <type_traits>
#include <utility>

#include "pod_device.h"
#include "pod_experiments.h"

namespace pod {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t STREAM_DROPOUT_CONV = 0x64726f70u;   // = STREAM_DROPOUT of k8_model_ops.hip: same mask as pod_bias_act

constexpr int WINO_U_FLOATS = 24 * 2 * 64 * 4;          // filter slab of a chunk: [24 positions][h][j][4 channels]  48 KB
constexpr int WINO_SB_FLOATS = 384 * 32;                 // raw patch stage of a SUPER-CHUNK (32 input channels): [pixel slot 360 (+24: 48 whole DMA instructions)][8 parts of 16 B], 48 KB
constexpr int WINO_LDS_BYTES = 4 * 32 * 4 * 65 * 4;      // 133 120 B of the CU's 160 KB: the output staging (the K loop needs 24 KB)

// -DPOD_TRACE (diagnostics build, tools/wino_trace.py): s_memtime stamps of every workgroup's phases
#ifdef POD_TRACE
static __device__ long long g_wino_trace[8192 * 16];
#define WINO_STAMP(k)                                                                                          \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_wino_trace[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
#define WINO_STAMP_WALL(k)                                                                                     \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_wino_trace[blockIdx.x * 16 + (k)] = wall_clock64();       \
    } while (0)
#else
#define WINO_STAMP(k)
#define WINO_STAMP_WALL(k)
#endif

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14): lgkmcnt(0) with vmcnt(4) / vmcnt(0)
constexpr int WINO_WAIT_VM4 = 0x0074, WINO_WAIT_VM0 = 0x0070, WINO_WAIT_VM16 = 0x4070, WINO_WAIT_LGKM0 = 0xC07F;

struct WinoParams {
    const float* in;
    float* out;
    const float* U;
    const float* bias;
    const int4* blocks;
    int32_t n_blocks, C, K, KS, in_stride, out_stride, relu, k_planes;   // k_planes > 0: NCHW planes of k_planes real channels
    uint32_t thresh;
    float scale;
    uint64_t seed, offset;
    // split over the input channels (pod_wino_conv3x3_split only; small maps: a res5 convolution is 48 workgroups of 32 chunks each):
    // grid.y = C / (16 c_split) workgroup sets, set z accumulates chunks [z c_split, (z + 1) c_split) and stores its partial sums at
    // out + z split_out_stride (channels-last, no bias / ReLU / dropout); pod_wino_reduce adds the partials in a fixed order.  0: no split.
    int32_t c_split;
    int64_t split_out_stride;
    // dropout masks of a REPLAYED launch (HIP graph): the Philox key is seed ^ mix(*epoch), epoch a device word the graph itself bumps
    // at the start of every replay -- seed and offset are launch arguments, i.e. constants of a captured launch.  null: key = seed.
    const uint64_t* epoch;
    // Sparse launch (round 5, k15_sparse_blocks.hip): null, or a device list {count, record indices ...} of the LIVE blocks: workgroup slot t
    // takes record live[1 + t] and slots >= live[0] exit at once (the grid is sized for the whole table: the count never visits the host).
    const int32_t* live;
    // the store pass writes `replicas` copies of the image (pod_wino_conv3x3_split_replicas: channels-last, one input image per record),
    // replica r as image r of the output canvas, each under its own dropout mask -- the mask pod_expand_dropout would draw for it.  1: off.
    int32_t replicas;
    // Grouped launch (pod_wino_conv3x3_split_grouped): up to four convolutions of the same shape -- the cls- and the bbox-subnet layer l,
    // the four predictors -- in ONE grid.  Blocks [set_first[s], set_first[s + 1]) of the concatenated table belong to set s (its records
    // are relative to ITS buffers); a set has its own input, output, filter, bias, Philox offset, replica count and plane count.
    // k_wino_conv3x3_split reads these, not the fields above (an ordinary launch is one set).
    struct Sets {
        int32_t first[4];         // first block of set s (first[0] = 0; unused sets: INT32_MAX)
        const float* in[4];
        float* out[4];
        const float* U[4];
        const float* bias[4];
        const float* in_amax[4];  // device word >= max |in| of the set (the f16 split's operand scale is derived from it)
        float* out_amax[4];       // null, or a device word the store pass max'es with |every value it stores|
        uint64_t offset[4];
        int32_t replicas[4];
        int32_t k_planes[4];
    } sets;
};


// What lane l3 = pixel slot, q = sub-slot of an LDS-DMA instruction fetches (see the layout in the kernel): per pixel slot 0..383
// (py 18 + px) | rot << 16 (pixel 324: the slot holds no pixel), computed at compile time.
struct WinoSlotTable {
    uint32_t v[384];
    constexpr WinoSlotTable() : v() {
        for (int p = 0; p < 384; ++p) {
            const int cls = p & 1, k = p >> 1, rr = k / 18, px = k - rr * 18;
            const int py = cls ? (rr < 4 ? rr + 4 : rr + 8) : (rr < 4 ? rr : rr < 8 ? rr + 4 : rr + 8);
            const bool ok = cls ? rr < 8 : rr < 10;
            v[p] = ok ? (uint32_t)((py * 18 + px) | ((((px >> 2) & 3) + 4 * ((py >> 1) & 1)) << 16)) : 324u;
        }
    }
};
static __device__ const WinoSlotTable g_wino_slots{};

// Workgroup -> (filter slice ks, block tb).  blockIdx & 7 is the XCD (round-robin dispatch).  An XCD serves TWO of the KS 64-channel
// filter slices (one when KS == 1) for its share of the blocks, and consecutive workgroups of an XCD take the SAME block with its two
// slices: the second one's patch comes out of that XCD's L2 instead of HBM (fp32 kernel, four slices: 2.39 -> 1.68 GB fetched per
// bench launch), while the two slices' filters (3 MB at C = 256) still stay resident in the 4 MB L2.
__device__ __forceinline__ void wino_schedule(int KS, int n_blocks, int& ks, int& tb) {
    const int xcd = blockIdx.x & 7, wi = (int)(blockIdx.x >> 3);
    const int SP = KS >= 2 ? 2 : 1, G = KS / SP;            // slices per XCD; XCD groups with different slice pairs
    ks = SP * (xcd % G) + (wi % SP);
    tb = (wi / SP) * (8 / G) + xcd / G;
}
inline int64_t wino_grid(int KS, int64_t n_blocks) {
    const int SP = KS >= 2 ? 2 : 1, G = KS / SP;
    return 8 * SP * ((n_blocks * G + 7) / 8);
}

// Ties values into the instruction order at this point (no instruction is emitted): what was computed before cannot sink below, what
// is computed from them cannot rise above.
template <typename T>
__device__ __forceinline__ void wino_pin_one(T& v) {
    asm volatile("" : "+v"(v));
}
template <typename... T>
__device__ __forceinline__ void wino_pin(T&... v) {
    (wino_pin_one(v), ...);
}

// ---- an fp32 value as the sum of two FP16 values (round 5: k12 / k13 / k14; pod_debug_f16_split2 exposes the same code to the tests)
// The f16 matrix cores run at the bf16 rate, and two f16 terms carry 11 + 1 (the sign of the residual) + 11 = 23 of an fp32's 24
// significand bits: x s = x0 + x1 + e, |e| <= 2^-23 |x s| in the worst case (exact whenever the residual has <= 11 significant bits), where
// x0 = f16(x s) (round to nearest even), r = x s - x0 EXACTLY (one fma), x1 = f16(r).  Three partial products (x0 u1, x1 u0, x0 u0) then form
// an fp32 product where the 3-way bf16 split needs six -- and with half as many roundings in the fp32 accumulation chain the result is
// CLOSER to the fp64 value than both the bf16 x 6 form and the fp32 MFMA (measured on the matrix cores: tools/f16_split_numerics.hip,
// profiles/r05_f16_split_numerics.txt).  What f16 lacks is range (2^-24 .. 65504): every operand tensor is therefore multiplied by a
// power of two s (exact) chosen from its abs-max, so that the largest scaled value lies in [2^14, 2^15) (filters; static) or below 2^15
// (activations: abs-max word of the producing launch x the largest gain of the transform); values more than ~2^29 below their tensor's
// abs-max fall into f16's denormals and keep an ABSOLUTE error of 2^-25 / s -- 2^-40 of the abs-max, against an fp32 rounding's 2^-24 |x|.
typedef _Float16 wino_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wino_f16x8 __attribute__((ext_vector_type(8)));
// 2^(top - floor(log2 amax)): the power of two that puts amax into [2^top, 2^(top+1)).  amax = 0 or absurdly small: the largest scale (the
// operand is zero / all products underflow anyway); inf / nan: the smallest (the products are inf / nan, as an fp32 product would be).
__device__ __forceinline__ float wino_pow2_scale(float amax, int top) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    int b = 254 + top - E;
    b = b < 1 ? 1 : b > 254 ? 254 : b;
    return __uint_as_float((uint32_t)b << 23);
}
__device__ __forceinline__ float wino_pow2_inverse(float s) {        // 1 / s for a power of two s = 2^k, |k| <= 126: exact
    return __uint_as_float((254u << 23) - __float_as_uint(s));
}
// (lo s, hi s) -> the f16 pair nearest to them (v_fma_mixlo_f16 / v_fma_mixhi_f16: the scaling rides on the conversion)
__device__ __forceinline__ uint32_t wino_f16_pair_scaled(float lo, float hi, float s) {
    uint32_t w;
    asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(w) : "v"(lo), "v"(hi), "s"(s));
    return w;
}
__device__ __forceinline__ uint32_t wino_f16_pair(float lo, float hi) {              // v_cvt_pk_f16_f32: nearest even, lo in bits 15:0
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, wino_f16x2));
}
// (lo, hi) <- (lo s, hi s) - the f16 pair w, exactly: one v_fma_mix_f32 per value (f32 x f32 - f16, a single rounding of a result that is
// representable: |x s - x0| <= 2^-11 |x s| and both are multiples of the last place of x s)
__device__ __forceinline__ void wino_f16_residual_scaled(uint32_t w, float& lo, float& hi, float s) {
    asm("v_fma_mix_f32 %0, %0, %1, -%2 op_sel_hi:[0,0,1]" : "+v"(lo) : "s"(s), "v"(w));
    asm("v_fma_mix_f32 %0, %0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(hi) : "s"(s), "v"(w));
}
__device__ __forceinline__ void wino_f16_split2(float lo, float hi, float s, uint32_t (&w)[2]) {
    w[0] = wino_f16_pair_scaled(lo, hi, s);
    wino_f16_residual_scaled(w[0], lo, hi, s);
    w[1] = wino_f16_pair(lo, hi);
}
// ---- operand abs-max RECORDS (include/pod_mi355x.h).  The abs-max of a tensor lives in POD_AMAX_SLOTS words POD_AMAX_STRIDE floats apart
// (one 128-byte line each); a producer max'es into slot (workgroup + wavefront) mod 16, a consumer takes the largest of the 16.  One word
// would do for the arithmetic -- but thousands of same-address atomics serialise in the L2 at ~10 ns each, and the wavefronts of a
// streaming launch all finish together (measured with one word: pod_absmax of 22 MB 109 us, a 5-us reduce launch 47 us).
// (POD_AMAX_SLOTS = 16, POD_AMAX_STRIDE = 32, POD_AMAX_FLOATS = 512: include/pod_mi355x.h)
// floats >= 0 order like their bit patterns: an integer atomic max.  One atomic per wavefront at most, skipped when the slot already holds more.
__device__ __forceinline__ void wino_publish_amax1(float* word, float lmax) {      // one wavefront's maximum into ONE word (a filter's trailer)
#pragma unroll
    for (int o = 32; o; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0 && lmax > 0.0f) {
        uint32_t* w = reinterpret_cast<uint32_t*>(word);
        const uint32_t bits = __float_as_uint(lmax);
        if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(w, bits);
    }
}
__device__ __forceinline__ void wino_publish_amax(float* rec, float lmax) {        // ... into its slot of a record
    const unsigned wg = blockIdx.x + blockIdx.y * gridDim.x;
    wino_publish_amax1(rec + ((wg * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (POD_AMAX_SLOTS - 1)) * POD_AMAX_STRIDE, lmax);
}
// the same for a whole workgroup (every thread calls it; <= 16 wavefronts): ONE atomic per workgroup
__device__ __forceinline__ void wino_publish_amax_block(float* rec, float lmax) {
    __shared__ float wave_max[16];
#pragma unroll
    for (int o = 32; o; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = lmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (unsigned w = 1; w < (blockDim.x >> 6); ++w) lmax = fmaxf(lmax, wave_max[w]);
        if (lmax > 0.0f) {
            uint32_t* word = reinterpret_cast<uint32_t*>(rec + ((blockIdx.x + blockIdx.y * gridDim.x) & (POD_AMAX_SLOTS - 1)) * POD_AMAX_STRIDE);
            const uint32_t bits = __float_as_uint(lmax);
            if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits) atomicMax(word, bits);
        }
    }
}
__device__ __forceinline__ float wino_load_amax(const float* rec) {                // this lane's slot of the record (ask early, reduce late)
    const int lane = threadIdx.x & 63;
    return lane < POD_AMAX_SLOTS ? rec[lane * POD_AMAX_STRIDE] : 0.0f;
}
__device__ __forceinline__ float wino_reduce_amax(float v) {                       // the record's value, wave-uniform (a scalar register)
#pragma unroll
    for (int o = POD_AMAX_SLOTS / 2; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float wino_read_amax(const float* rec) { return wino_reduce_amax(wino_load_amax(rec)); }

// the eight-wavefront form of pod_wino_conv3x3_split (tools/experiments/k16_wino_conv_split8.hip, -DPOD_WITH_K16 builds); P as k12's entry validated and filled it
int wino_split8_launch(const WinoParams& P, int64_t grid, unsigned grid_y, hipStream_t stream);

template <typename F, int... Js>
__device__ __forceinline__ void wino_static_for(F&& f, std::integer_sequence<int, Js...>) {
    (f(std::integral_constant<int, Js>{}), ...);
}


}  // namespace pod
