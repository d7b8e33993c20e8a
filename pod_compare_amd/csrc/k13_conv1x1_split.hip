// K13 conv1x1_split -- the 1x1 convolutions of the backbone and the FPN as a channels-last GEMM on the bf16 matrix cores.
//
// Replaces (reference: detectron2's ResNet / FPN as probabilistic_retinanet.py:96-100 runs them, `features = self.backbone(images.tensor)`):
// BottleneckBlock.conv1 / conv3 / shortcut (1x1, stride 1 or 2, FrozenBN folded into weight + bias, ReLU, residual add) and
// FPN.lateral_convs -- 39 calls per image, 85 GFLOP, which MIOpen runs as fp32 Tensile GEMMs at ~68 TFLOP/s plus one element-wise pass
// each for bias / residual / ReLU (1.5 ms of a 10.3-ms step once the 3x3 convolutions were off MIOpen).
//
//   y[p][k] = act( sum_c x[pin(p)][c] w[k][c] + bias[k] (+ residual[p][k]) ),   x, y, residual channels-last ([pixel][channel])
//
// Same arithmetic contract as pod_wino_conv3x3_split (k12).  Rounds 3-4: exact 3-way bf16 splits of both operands, 6 partial products.
// Round 5: 2-way F16 splits of the power-of-two-scaled operands (pod_wino.h: x s = x0 + x1 to 2^-23 |x s|), 3 partial products on
// v_mfma_f32_32x32x16_f16, fp32 accumulate -- half the matrix instructions and a shorter fp32 accumulation chain (closer to fp64 than
// both the bf16 form and the fp32 MFMA: tools/f16_split_numerics.hip).  The weights are split once (pod_conv1x1_filter_split, scale from
// their own abs-max), the activations in the loop with the very functions k12 uses (wino_f16_pair_scaled / wino_f16_residual_scaled),
// their scale from the launch's `in_amax` word; the store pass publishes the output's abs-max for the next convolution.
//
// Mapping.  Workgroup = ONE wavefront = 64 output pixels x (32 NCB) output channels (NCB = 2 as shipped): 2 NCB accumulator blocks
// of 32 x 32.  The filter is the ROW operand of the MFMAs (a lane's accumulator quad is 4 consecutive output
// channels of one pixel: 16-byte stores into the channels-last output).  No LDS: with both operands K-contiguous a lane's MFMA
// fragment IS a contiguous piece of memory -- 32 B of one pixel's channels, 16 B of one filter row's pre-split terms -- so fragments
// are loaded straight into registers, TWO k-steps (16 channels each) ahead through three rotating register buffers; wavefronts that
// share a CU and a channel tile meet in its L1 for the filter terms.  A pixel tile's wavefronts (one per channel tile) run on ONE XCD
// back to back, so the activations come out of that XCD's L2 after the first.  One-wavefront workgroups because nothing is shared
// through LDS and small maps need every tile to be its own schedulable unit (res5: 1008 pixels x 2048 channels = 256 tiles); where
// even that leaves the chip idle the input channels are cut over grid.y (partial sums, finished by pod_conv1x1_reduce in a fixed order).
#include "pod_wino.h"

#ifdef POD_C1_TRACE       // experiment builds: 10-ns time stamps of every wavefront (start | first k-step done | loop done | end), pod_c1_trace_dump()
static __device__ long long g_c1_trace[8192 * 4];
static __device__ long long g_c1_cycles[8192 * 4];      // s_memtime beside the constant 100-MHz clock: the shader clock the wavefront ran at
#define C1_STAMP(k)                                                                                   \
    do {                                                                                              \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                     \
        if (threadIdx.x == 0 && wg_ < 8192) {                                                         \
            g_c1_trace[wg_ * 4 + (k)] = wall_clock64();                                               \
            g_c1_cycles[wg_ * 4 + (k)] = clock64();                                                   \
        }                                                                                             \
    } while (0)
#else
#define C1_STAMP(k)
#endif

namespace pod {

typedef uint32_t c1_u32x4 __attribute__((ext_vector_type(4)));
constexpr int C1_KS_U16 = 2 * 2 * 256;       // u16 values of one (32-channel block, k-step): [term 2][h 2][i32 32][8 f16]
constexpr int C1_TOP = 14;                   // both operands: scaled abs-max in [2^14, 2^15)

struct C1Params {
    const float* x;
    float* y;                 // output, or the partial sums of split 0 (split z at + z * split_stride)
    const uint16_t* Ws;       // pre-split filter: [cout block 32][k-step 16][term 2][h 2][i32 32][8 f16], then the abs-max word (16-byte trailer)
    const float* in_amax;     // device word >= max |x| (the activation scale of the f16 split)
    float* out_amax;          // null, or a device word max'ed with |every value stored| (final pass only)
    const float* bias;
    const float* residual;
    int32_t P_out, W_out, W_in, stride, Cin, Cout, relu;
    int32_t res_up2, W_res;   // the residual is the half-resolution map (W_res columns): pixel (y, x) adds residual[(y >> 1) * W_res + (x >> 1)]
    int32_t n_pt, n_ct;       // pixel tiles (64), channel tiles (32 NCB)
    int32_t ks_per_split;     // k-steps (16 channels) a workgroup set accumulates; grid.y sets
    int64_t split_stride;     // floats between partial outputs; 0: no split (bias / residual / ReLU applied here)
};

// weight (Cout, Cin) fp32 -> Ws: two nearest-even f16 terms per scaled value (w s = w0 + w1 to 2^-23), in the order a lane loads them;
// s = the power of two that puts the weight's abs-max (first pass, -> the trailer word) into [2^14, 2^15)
__global__ void __launch_bounds__(256) k_conv1x1_filter_amax(const float* __restrict__ w, int64_t n, float* __restrict__ amax) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    wino_publish_amax1(amax, m);
}
__global__ void __launch_bounds__(256) k_conv1x1_filter_split(const float* __restrict__ w, uint16_t* __restrict__ Ws, const float* __restrict__ amax, int32_t Cout,
                                                              int32_t Cin) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (cout, pair of cins)
    if (t >= (int64_t)Cout * (Cin / 2)) return;
    const int k = (int)(t / (Cin / 2)), c = 2 * (int)(t % (Cin / 2));
    uint32_t terms[2];
    wino_f16_split2(w[(int64_t)k * Cin + c], w[(int64_t)k * Cin + c + 1], wino_pow2_scale(*amax, C1_TOP), terms);
    const int nks = Cin >> 4, cb = k >> 5, i32 = k & 31, ks = c >> 4, h = (c >> 3) & 1, e = c & 7;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        uint16_t* d = Ws + ((((int64_t)cb * nks + ks) * 2 + term) * 2 + h) * 256 + i32 * 8 + e;
        d[0] = (uint16_t)(terms[term] & 0xFFFFu);
        d[1] = (uint16_t)(terms[term] >> 16);
    }
}

template <int I>
using c1_ic = std::integral_constant<int, I>;

// RING: register buffers of one k-step each; loads run RING - 1 k-steps ahead.  3 leaves room for two wavefronts per SIMD.  Deeper rings
// were measured on the launches that have at most one wavefront per SIMD anyway (res4 / res5, the laterals: 1600 cycles per k-step
// against the 768 of its MFMAs) and change nothing (ring 4 / 5 / 6: 1.17 / 1.15 / 1.20 ms per image against 1.18): what those
// wavefronts wait for is not the distance of the loads but the L1's time for the activation fragments -- 32 cache lines per
// instruction (profiles/r04_experiments.md, K13).
template <int NCB, int RING>
__global__ void __launch_bounds__(64) k_conv1x1_split(const C1Params P) {
    const int lane = threadIdx.x & 63, i32 = lane & 31, h = lane >> 5;
    // blockIdx & 7 is the XCD (round-robin dispatch): an XCD takes pixel tiles xcd, xcd + 8, ... and runs all channel tiles of one back to back
    const int xcd = blockIdx.x & 7, wi = (int)(blockIdx.x >> 3);
    const int tp = (wi / P.n_ct) * 8 + xcd, tc = wi % P.n_ct;
    if (tp >= P.n_pt) return;
    C1_STAMP(0);
    const int nks_all = P.Cin >> 4, ks0 = (int)blockIdx.y * P.ks_per_split, nks = P.ks_per_split;
    // this lane's two pixels (pb = 0, 1) and where they live in the input (stride-2 convolutions read every second row / column)
    int pout[2], pin[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int p = tp * 64 + pb * 32 + i32;
        pout[pb] = p < P.P_out ? p : -1;
        const int q = p < P.P_out ? p : 0;
        if (P.stride == 1) {
            pin[pb] = q;
        } else {
            const int oy = q / P.W_out, ox = q - oy * P.W_out;
            pin[pb] = (P.stride * oy) * P.W_in + P.stride * ox;
        }
    }
    const float* __restrict__ xa[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) xa[pb] = P.x + (int64_t)pin[pb] * P.Cin + ks0 * 16 + 8 * h;
    const uint16_t* __restrict__ const wa = P.Ws + ((int64_t)(tc * NCB) * nks_all + ks0) * C1_KS_U16 + (h * 32 + i32) * 8;     // + cb * nks_all * 1024 + ks * 1024 + term * 512
    const int64_t w_cb = (int64_t)nks_all * C1_KS_U16;
    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wino_pow2_scale(wino_read_amax(P.in_amax), C1_TOP))));

    f32x16 acc[NCB][2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 araw[RING][2][2];           // [buffer][pb][4-channel half of the lane's 8]
    c1_u32x4 wf[RING][NCB][2];        // [buffer][cb][term]
    auto load = [&](auto buf_t, int ks) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_t)::value;
        if (!(POD_C1_ELIM & 1) || ks < RING) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
                araw[buf][pb][0] = *reinterpret_cast<const f32x4*>(xa[pb] + ks * 16);
                araw[buf][pb][1] = *reinterpret_cast<const f32x4*>(xa[pb] + ks * 16 + 4);
            }
        }
        if (!(POD_C1_ELIM & 2) || ks < RING) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int t = 0; t < 2; ++t) wf[buf][cb][t] = *reinterpret_cast<const c1_u32x4*>(wa + cb * w_cb + (int64_t)ks * C1_KS_U16 + t * 512);
        }
    };
    auto step = [&](auto buf_t, auto first_t) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_t)::value;
        constexpr bool first = decltype(first_t)::value;
        c1_u32x4 at[2][2];            // the lane's 8 channels of its two pixels as two f16 terms
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {       // pair i: channels 2 i, 2 i + 1
                float lo = araw[buf][pb][i >> 1][2 * (i & 1)], hi = araw[buf][pb][i >> 1][2 * (i & 1) + 1];
                if (POD_C1_ELIM & 4) {
                    at[pb][0][i] = at[pb][1][i] = __builtin_bit_cast(uint32_t, lo);
                    continue;
                }
                const uint32_t t0 = wino_f16_pair_scaled(lo, hi, sx);
                wino_f16_residual_scaled(t0, lo, hi, sx);
                at[pb][0][i] = t0;
                at[pb][1][i] = wino_f16_pair(lo, hi);
            }
        // the 3 partial products that matter, small ones first (as k12): w0 x1, w1 x0, w0 x0
#pragma unroll
        for (int prod = 0; prod < 3; ++prod) {
            const int sa = prod == 1 ? 1 : 0;
            const int sb = prod == 0 ? 1 : 0;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
                    if (first && prod == 0)
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, wf[buf][cb][sa]), __builtin_bit_cast(wino_f16x8, at[pb][sb]), zero16, 0, 0, 0);
                    else
                        acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, wf[buf][cb][sa]), __builtin_bit_cast(wino_f16x8, at[pb][sb]), acc[cb][pb], 0, 0, 0);
                }
        }
    };
    // k-step j computes from buffer j % RING and, before that, refills the buffer k-step j - 1 just left with k-step j + RING - 1;
    // the buffers rotate at compile time (RING k-steps per trip of the loop)
    auto prologue = [&](auto self, auto r_t) __attribute__((always_inline)) -> void {
        constexpr int R = decltype(r_t)::value;
        if constexpr (R < RING - 1) {
            if (R < nks) load(c1_ic<R>{}, R);
            self(self, c1_ic<R + 1>{});
        }
    };
    prologue(prologue, c1_ic<0>{});
    if (RING - 1 < nks) load(c1_ic<RING - 1>{}, RING - 1);
    step(c1_ic<0>{}, std::true_type{});
    C1_STAMP(1);
    auto trip = [&](auto self, auto r_t, int ks) __attribute__((always_inline)) -> void {        // k-steps ks + R, R = 0 .. RING - 1, ks % RING == 1
        constexpr int R = decltype(r_t)::value;
        if constexpr (R < RING) {
            if (ks + R < nks) {
                if (ks + R + RING - 1 < nks) load(c1_ic<R % RING>{}, ks + R + RING - 1);
                step(c1_ic<(R + 1) % RING>{}, std::false_type{});
                self(self, c1_ic<R + 1>{}, ks);
            }
        }
    };
    for (int ks = 1; ks < nks; ks += RING) trip(trip, c1_ic<0>{}, ks);

    C1_STAMP(2);
    // ---- epilogue: a lane's accumulator register r of block (cb, pb) is channel 32 cb + (r & 3) + 8 (r >> 2) + 4 h of pixel 32 pb + i32.
    // One wavefront per SIMD: nobody else hides this wavefront's latencies, so the residual quads of a pixel (4 NCB independent 16-byte
    // loads) are all requested before the first is used, and nothing may alias (`__restrict__`: a store to y would otherwise fence the
    // residual loads behind it and turn the epilogue into 32 serial round trips -- measured: 25 of res4-conv3's 37 us).
    float* __restrict__ const yo = P.y + (int64_t)blockIdx.y * P.split_stride;
    const float* __restrict__ const res = P.residual;
    const float* __restrict__ const bias = P.bias;
    const bool final_pass = P.split_stride == 0;
    const int k0 = tc * NCB * 32 + 4 * h;                       // + 32 cb + 8 q
    // the accumulators hold (s_w w) (s_x x) sums: the two powers of two come off here, exactly, inside the multiply-add that adds the bias
    const float inv1 = wino_pow2_inverse(sx) * wino_pow2_inverse(wino_pow2_scale(*reinterpret_cast<const float*>(P.Ws + (int64_t)P.Cout * P.Cin * 2), C1_TOP));
    const f32x4 inv = f32x4{inv1, inv1, inv1, inv1};
    float lmax = 0.0f;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        if (pout[pb] < 0) continue;
        const int64_t e0 = (int64_t)pout[pb] * P.Cout + k0;
        f32x4 r[NCB][4];
        if (final_pass && res) {
            int64_t re0 = e0;
            if (P.res_up2) {                                                     // FPN's top-down sum: the coarser map, nearest neighbour
                const int oy = pout[pb] / P.W_out, ox = pout[pb] - oy * P.W_out;
                re0 = ((int64_t)(oy >> 1) * P.W_res + (ox >> 1)) * P.Cout + k0;
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    r[cb][q] = (k0 + 32 * cb + 8 * q < P.Cout) ? *reinterpret_cast<const f32x4*>(res + re0 + 32 * cb + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + 32 * cb + 8 * q;
                if (k >= P.Cout) continue;
                f32x4 v = f32x4{acc[cb][pb][4 * q], acc[cb][pb][4 * q + 1], acc[cb][pb][4 * q + 2], acc[cb][pb][4 * q + 3]};
                v = __builtin_elementwise_fma(v, inv, final_pass && bias ? *reinterpret_cast<const f32x4*>(bias + k) : f32x4{0.f, 0.f, 0.f, 0.f});
                if (final_pass) {
                    if (res) v += r[cb][q];
                    if (P.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
                if (!(POD_C1_ELIM & 8) || v.x == 12345.678f) *reinterpret_cast<f32x4*>(yo + e0 + 32 * cb + 8 * q) = v;
            }
        }
    }
    if (final_pass && P.out_amax) wino_publish_amax(P.out_amax, lmax);
#ifdef POD_C1_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    C1_STAMP(3);
#endif
}

// The epilogue of the LDS kernels, in whole lines: the accumulators (a lane: 4 consecutive channels of ONE pixel per register quad -- 32
// pixels x 32 B per store instruction) go through 16 KB of LDS, [pixel 64][chunk position 16][16 B] with position = chunk ^ (pixel & 15),
// and come back as 4 pixels x 256 B per instruction: residual loads and output stores of 8 full lines each, all 16 residual loads
// requested before the first is used (one wavefront per SIMD: nobody else hides them).
__device__ __forceinline__ void c1_epilogue(const C1Params& P, float* const lds_o, f32x16 (&acc)[2][2], int tp, int tc, int lane, int i32, int h, float sx) {
    constexpr int NCB = 2;
    const float inv1 = wino_pow2_inverse(sx) * wino_pow2_inverse(wino_pow2_scale(*reinterpret_cast<const float*>(P.Ws + (int64_t)P.Cout * P.Cin * 2), C1_TOP));
    const f32x4 inv = f32x4{inv1, inv1, inv1, inv1};
    float lmax = 0.0f;
    float* __restrict__ const yo = P.y + (int64_t)blockIdx.y * P.split_stride;
    const float* __restrict__ const res = P.residual;
    const bool final_pass = P.split_stride == 0;
    const int oc = lane & 15, op = lane >> 4;                       // read side: chunk oc (channels 4 oc ..) of pixel 4 j + op
    const int gp0 = tp * 64 + op;
    const int64_t e0 = (int64_t)gp0 * P.Cout + tc * 64 + 4 * oc;    // + 4 j Cout
    f32x4 r[16];
    if (final_pass && res && P.res_up2) {                             // FPN's top-down sum: the coarser map, nearest neighbour -- pixel (y, x) adds (y >> 1, x >> 1)
        int oy = gp0 / P.W_out, ox = gp0 - oy * P.W_out;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            r[j] = (gp0 + 4 * j < P.P_out) ? *reinterpret_cast<const f32x4*>(res + ((int64_t)(oy >> 1) * P.W_res + (ox >> 1)) * P.Cout + tc * 64 + 4 * oc)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            ox += 4;
            while (ox >= P.W_out) {
                ox -= P.W_out;
                ++oy;
            }
        }
    } else if (final_pass && res) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = (gp0 + 4 * j < P.P_out) ? *reinterpret_cast<const f32x4*>(res + e0 + (int64_t)(4 * j) * P.Cout) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (final_pass && P.bias) b4 = *reinterpret_cast<const f32x4*>(P.bias + tc * 64 + 4 * oc);
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pix = 32 * pb + i32, c = 8 * cb + 2 * q + h;
                *reinterpret_cast<f32x4*>(lds_o + pix * 64 + 4 * (c ^ (i32 & 15))) =
                    f32x4{acc[cb][pb][4 * q], acc[cb][pb][4 * q + 1], acc[cb][pb][4 * q + 2], acc[cb][pb][4 * q + 3]};
            }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pix = 4 * j + op;
        f32x4 v = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(lds_o + pix * 64 + 4 * (oc ^ (pix & 15))), inv, b4);      // (b4 = 0 for partial sums)
        if (final_pass) {
            if (res) v += r[j];
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
        }
        if (gp0 + 4 * j < P.P_out) {
            *reinterpret_cast<f32x4*>(yo + e0 + (int64_t)(4 * j) * P.Cout) = v;
            lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    if (final_pass && P.out_amax) wino_publish_amax(P.out_amax, lmax);
}

// The same tile with the activations taken through LDS (the production form whenever a workgroup set accumulates an even number of
// k-steps).  k_conv1x1_split loads a lane's MFMA fragment straight from memory: 16 B of each of 32 pixels per instruction = 32 cache
// lines for 1 KB, and the CU's L1 takes a cycle per line -- measured (POD_C1_ELIM builds, res4 conv1): the activation loads are 6.3 of
// the launch's 31.8 us, the filter loads (8 lines per instruction) 1.7.  Here a PAIR of k-steps (32 channels = one 128-B line per
// pixel) is loaded in whole lines -- 8 instructions of 8 pixels x 128 B -- one pair ahead, parked in LDS, and the fragments come back
// by ds_read_b128.  LDS image of a pair: [pixel 64][chunk position 8][16 B]; position c of pixel p holds channels 4 (c ^ key(p)) ..
// + 3, key(p) = (p >> 1) & 7 (the swizzle is applied to the SOURCE address, the LDS write is lane-linear): the 16 lanes of a
// ds_read_b128 group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) then hit 16 different bank quads.
// One wavefront per workgroup still: no barrier anywhere, the LDS is a transposing buffer of this wavefront alone (two pairs, 16 KB).
// Arithmetic, channel order and accumulation order are those of k_conv1x1_split: the results are bit-identical (WAVES = 1).
//
// WAVES > 1 (round 5): SPLIT-K INSIDE THE WORKGROUP.  From res3 down a launch is 60 .. 500 tiles for 1024 SIMDs and every lone wavefront
// walks its whole K chain at 0.3-0.4 us per 16 channels (profiles/r05_conv_classes.md: 0.06-0.25 of the classes' own bounds); cutting K
// over workgroup SETS (grid.y) costs the partial sums' trip through HBM and a reduce launch.  Here the tile's K range is cut over the
// WAVES wavefronts of one workgroup -- each with its own LDS transposing buffer and filter ring, no barrier in the loop -- and their
// accumulators meet in LDS once at the end, added in a FIXED order (wavefront 0 + 1 + 2 + 3: the result does not depend on scheduling);
// wavefront 0 runs the epilogue.  The k-steps a wavefront accumulates stay in order, so WAVES = w equals a grid.y split of w to the bit.
template <int NCB, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, WAVES > 1 ? 1 : 2) k_conv1x1_split_lds(const C1Params P) {
    __shared__ __attribute__((aligned(16))) float lds_all[WAVES][2][64 * 32];
    const int wave = threadIdx.x >> 6;
    float (*const lds_a)[64 * 32] = lds_all[wave];
    const int lane = threadIdx.x & 63, i32 = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, wi = (int)(blockIdx.x >> 3);
    const int tp = (wi / P.n_ct) * 8 + xcd, tc = wi % P.n_ct;
    if (tp >= P.n_pt) return;
    C1_STAMP(0);
    const int nks_all = P.Cin >> 4, nks = P.ks_per_split / WAVES, ks0 = (int)blockIdx.y * P.ks_per_split + wave * nks;
    // staging: instruction j loads pixels 8 j .. 8 j + 7 of the tile, this lane chunk position lane & 7 of pixel 8 j + (lane >> 3)
    int32_t soff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pl = 8 * j + (lane >> 3), p = tp * 64 + pl;
        const int q = p < P.P_out ? p : 0;
        int pin = q;
        if (P.stride != 1) {
            const int oy = q / P.W_out, ox = q - oy * P.W_out;
            pin = (P.stride * oy) * P.W_in + P.stride * ox;
        }
        soff[j] = pin * P.Cin + ks0 * 16 + 4 * ((lane & 7) ^ ((pl >> 1) & 7));
    }
    const float* __restrict__ const xg = P.x;
    const int key = (i32 >> 1) & 7;
    const uint16_t* __restrict__ const wa = P.Ws + ((int64_t)(tc * NCB) * nks_all + ks0) * C1_KS_U16 + (h * 32 + i32) * 8;
    const int64_t w_cb = (int64_t)nks_all * C1_KS_U16;
    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wino_pow2_scale(wino_read_amax(P.in_amax), C1_TOP))));

    f32x16 acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;
    const int npairs = nks >> 1;
    f32x4 stg[8];                     // the pair in flight
    c1_u32x4 wf[3][NCB][2];           // [buffer][cb][term]
    auto stage_load = [&](int pair) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) stg[j] = *reinterpret_cast<const f32x4*>(xg + soff[j] + pair * 32);
    };
    auto stage_write = [&](int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(&lds_a[b][j * 256 + lane * 4]) = stg[j];
    };
    auto load_w = [&](auto buf_t, int ks) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_t)::value;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t) wf[buf][cb][t] = *reinterpret_cast<const c1_u32x4*>(wa + cb * w_cb + (int64_t)ks * C1_KS_U16 + t * 512);
    };
    // Built, measured and dropped (profiles/r04_experiments.md, K13): splitting k-step j + 1 beside the MFMAs of k-step j slot by slot as
    // k12 does -- first with units of 7 VALU instructions in every second MFMA gap (the loop kept its 0.73 us per k-step and the second
    // term buffer cost the second wavefront per SIMD that the large maps need: 1.02 ms per image against 0.975), then evenly (one part
    // of <= 3 instructions and <= 1 memory operation per slot, fragments read two k-steps ahead, one wavefront per SIMD, launches of
    // <= 1100 wavefronts only: 10 % fewer cycles per k-step at a 10 % lower clock; 5 - 10 % faster on the long-K shapes of one box,
    // nothing on another, and 1 % SLOWER end to end on cfg2, where a 456-register wavefront keeps other streams' work off its SIMD); a four-wavefront workgroup sharing
    // the channel tile's filter terms through LDS (half the L1 / L2 traffic per MFMA, one barrier per pair: 1.02 against 0.93, the
    // small maps 30-40 % slower).  The counters say where a lone wavefront's k-step goes: 768 cycles of MFMA + ~450 of VALU issue
    // (113 instructions) + ~250 of waits, one after the other -- the VALU block of a k-step does not overlap its own MFMAs.
    auto step = [&](auto buf_t, auto t_t, int b) __attribute__((always_inline)) {       // k-step t of the pair in LDS buffer b
        constexpr int buf = decltype(buf_t)::value;
        constexpr int t = decltype(t_t)::value;
        c1_u32x4 at[2][2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const float* row = &lds_a[b][(pb * 32 + i32) * 32];
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(row + 4 * ((4 * t + 2 * h) ^ key));
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(row + 4 * ((4 * t + 2 * h + 1) ^ key));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = i < 2 ? a0[2 * (i & 1)] : a1[2 * (i & 1)], hi = i < 2 ? a0[2 * (i & 1) + 1] : a1[2 * (i & 1) + 1];
                const uint32_t t0 = wino_f16_pair_scaled(lo, hi, sx);
                wino_f16_residual_scaled(t0, lo, hi, sx);
                at[pb][0][i] = t0;
                at[pb][1][i] = wino_f16_pair(lo, hi);
            }
        }
#pragma unroll
        for (int prod = 0; prod < 3; ++prod) {
            const int sa = prod == 1 ? 1 : 0;
            const int sb = prod == 0 ? 1 : 0;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, wf[buf][cb][sa]), __builtin_bit_cast(wino_f16x8, at[pb][sb]), acc[cb][pb], 0, 0, 0);
        }
    };
    // prologue: pair 0 into LDS buffer 0, pair 1 in flight, filter terms of k-steps 0 and 1
    stage_load(0);
    load_w(c1_ic<0>{}, 0);
    load_w(c1_ic<1>{}, 1);
    stage_write(0);
    if (npairs > 1) stage_load(1);
    C1_STAMP(1);
    // k-step j = 6 m + R: filter buffer R % 3 (refilled two k-steps ahead), pair j / 2 from LDS buffer (j / 2) & 1; after a pair's second
    // k-step the pair in flight is parked in the other buffer and the one after it requested
    auto trip = [&](auto self, auto r_t, int j0) __attribute__((always_inline)) -> void {
        constexpr int R = decltype(r_t)::value;
        if constexpr (R < 6) {
            const int j = j0 + R;
            if (j < nks) {
                if (j + 2 < nks) load_w(c1_ic<(R + 2) % 3>{}, j + 2);
                const int pair = j >> 1;
                step(c1_ic<R % 3>{}, c1_ic<R & 1>{}, pair & 1);
                if constexpr ((R & 1) == 1) {
                    if (pair + 1 < npairs) {
                        stage_write((pair + 1) & 1);
                        if (pair + 2 < npairs) stage_load(pair + 2);
                    }
                }
                self(self, c1_ic<R + 1>{}, j0);
            }
        }
    };
    for (int j0 = 0; j0 < nks; j0 += 6) trip(trip, c1_ic<0>{}, j0);
    C1_STAMP(2);

    if constexpr (WAVES > 1) {
        // the other wavefronts park their accumulators (lane-contiguous: [register][lane]) in their own transposing buffer -- nobody else
        // ever touched it --, wavefront 0 adds them in order
        if (wave > 0) {
            float* o = &lds_a[0][0];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[((cb * 2 + pb) * 16 + r) * 64 + lane] = acc[cb][pb][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) {
            const float* o = &lds_all[w][0][0];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][pb][r] += o[((cb * 2 + pb) * 16 + r) * 64 + lane];
        }
    }
    c1_epilogue(P, &lds_a[0][0], acc, tp, tc, lane, i32, h, sx);
#ifdef POD_C1_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    C1_STAMP(3);
#endif
}

// y = act(sum of the partial outputs in order + bias + residual), channels-last, 16 B per lane
__global__ void __launch_bounds__(256) k_conv1x1_reduce(const float* __restrict__ partials, int32_t n_splits, int64_t split_stride, const float* __restrict__ bias,
                                                        const float* __restrict__ residual, float* __restrict__ y, int64_t n4, int32_t Cout, int32_t relu,
                                                        float* __restrict__ out_amax) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float lmax = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = *reinterpret_cast<const f32x4*>(partials + 4 * i);
        for (int s = 1; s < n_splits; ++s) v += *reinterpret_cast<const f32x4*>(partials + (int64_t)s * split_stride + 4 * i);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + (int)((4 * i) % Cout));
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + 4 * i);
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4*>(y + 4 * i) = v;
        lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (out_amax) wino_publish_amax_block(out_amax, lmax);
}

}  // namespace pod

extern "C" int64_t pod_conv1x1_filter_split_bytes(int32_t Cout, int32_t Cin) {      // size of Ws: the terms + the 16-byte trailer (abs-max word)
    if (Cout < 32 || (Cout & 31) != 0 || Cin < 16 || (Cin & 15) != 0) return 0;
    return (int64_t)Cout * Cin * 4 + 16;
}

extern "C" int pod_conv1x1_filter_split(const float* weight, void* Ws, int32_t Cout, int32_t Cin, pod_stream_t stream) {
    if (!weight || !Ws || Cout < 32 || (Cout & 31) != 0 || Cin < 16 || (Cin & 15) != 0 || (reinterpret_cast<uintptr_t>(Ws) & 15u) != 0) return POD_E_INVALID;
    const int64_t n = (int64_t)Cout * (Cin / 2);
    float* amax = reinterpret_cast<float*>(reinterpret_cast<char*>(Ws) + (int64_t)Cout * Cin * 4);
    if (hipMemsetAsync(amax, 0, 16, (hipStream_t)stream) != hipSuccess) return POD_E_LAUNCH;
    hipLaunchKernelGGL(pod::k_conv1x1_filter_amax, dim3(256), dim3(256), 0, (hipStream_t)stream, weight, (int64_t)Cout * Cin, amax);
    POD_CHECK_LAUNCH();
    hipLaunchKernelGGL(pod::k_conv1x1_filter_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, reinterpret_cast<uint16_t*>(Ws), amax,
                       Cout, Cin);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_conv1x1_split(const float* x, float* y, const void* Ws, const float* bias, const float* residual, int32_t H_out, int32_t W_out, int32_t H_in,
                                 int32_t W_in, int32_t stride, int32_t Cin, int32_t Cout, int32_t flags, int32_t n_splits, float* partials, int32_t waves,
                                 const float* in_amax, float* out_amax, pod_stream_t stream) {
    if (!x || !y || !Ws || !in_amax || x == y || H_out < 1 || W_out < 1 || (stride != 1 && stride != 2) || Cin < 16 || (Cin & 15) != 0 || Cout < 64 || (Cout & 63) != 0)
        return POD_E_INVALID;
    if (H_in < (H_out - 1) * stride + 1 || W_in < (W_out - 1) * stride + 1 || (stride == 1 && (H_in != H_out || W_in != W_out))) return POD_E_INVALID;
    const int64_t P_out = (int64_t)H_out * W_out;
    if (P_out * (Cout > Cin ? Cout : Cin) >= ((int64_t)1 << 31) || (int64_t)H_in * W_in * Cin >= ((int64_t)1 << 31)) return POD_E_INVALID;
    const int nks = Cin / 16;
    if (n_splits < 1 || n_splits > 16 || nks % n_splits != 0 || (n_splits > 1 && !partials)) return POD_E_INVALID;
    const int res_up2 = (flags & POD_C1_RESIDUAL_UP2) ? 1 : 0, relu = flags & POD_C1_RELU;
    if (res_up2 && (!residual || n_splits > 1)) return POD_E_INVALID;     // (the second launch of a split adds full-resolution residuals only)
    // waves: wavefronts of a workgroup sharing a tile's K range (1, 2 or 4; each takes whole PAIRS of k-steps); 0 = choose here: as many as
    // keep the launch within ONE wavefront per SIMD and leave every wavefront at least 8 k-steps (measured per shape, splits x wavefronts: tools/conv1x1_splits.py, profiles/r05_conv1x1_splits.txt:
    // shorter chains or fuller launches lose more to the LDS meeting and the halved occupancy than the shorter chain wins)
    if (waves < 0 || waves > 4 || waves == 3) return POD_E_INVALID;
    {
        const int per_split = nks / n_splits;
        if (waves == 0) {
            const int64_t tiles = ((P_out + 63) / 64) * (Cout / 64) * n_splits;
            waves = 1;
            while (waves < 4 && per_split % (waves * 4) == 0 && per_split / (waves * 2) >= 8 && tiles * waves * 2 <= 1024) waves *= 2;
        }
        if (waves > 1 && (per_split % (2 * waves) != 0)) return POD_E_INVALID;
    }
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(Ws) | reinterpret_cast<uintptr_t>(bias) |
          reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(partials)) & 15u) != 0 ||
        ((reinterpret_cast<uintptr_t>(in_amax) | reinterpret_cast<uintptr_t>(out_amax)) & 3u) != 0)
        return POD_E_INVALID;
    pod::C1Params P;
    P.x = x; P.y = n_splits > 1 ? partials : y; P.Ws = reinterpret_cast<const uint16_t*>(Ws); P.bias = bias; P.residual = residual;
    P.in_amax = in_amax; P.out_amax = out_amax;
    P.P_out = (int32_t)P_out; P.W_out = W_out; P.W_in = W_in; P.stride = stride; P.Cin = Cin; P.Cout = Cout; P.relu = relu;
    P.res_up2 = res_up2; P.W_res = (W_out + 1) / 2;
    // 64 pixels x 64 channels per wavefront (two wavefronts per SIMD: one's epilogue under the other's MFMAs; 128-channel tiles measured
    // 1.29 ms per image against 1.17)
    P.n_pt = (int32_t)((P_out + 63) / 64); P.n_ct = Cout / 64;
    P.ks_per_split = nks / n_splits;
    P.split_stride = n_splits > 1 ? P_out * Cout : 0;
    const int64_t grid = 8LL * ((P.n_pt + 7) / 8) * P.n_ct;
    if (grid > 0x7FFFFFFFLL) return POD_E_INVALID;
    if (POD_C1_DIRECT)          // (experiment builds: the direct-fragment kernel everywhere, pod_experiments.h)
        hipLaunchKernelGGL((pod::k_conv1x1_split<2, POD_C1_RING>), dim3((unsigned)grid, (unsigned)n_splits), dim3(64), 0, (hipStream_t)stream, P);
    else if ((P.ks_per_split & 1) == 0 && waves == 4)
        hipLaunchKernelGGL((pod::k_conv1x1_split_lds<2, 4>), dim3((unsigned)grid, (unsigned)n_splits), dim3(256), 0, (hipStream_t)stream, P);
    else if ((P.ks_per_split & 1) == 0 && waves == 2)
        hipLaunchKernelGGL((pod::k_conv1x1_split_lds<2, 2>), dim3((unsigned)grid, (unsigned)n_splits), dim3(128), 0, (hipStream_t)stream, P);
    else if ((P.ks_per_split & 1) == 0)
        hipLaunchKernelGGL((pod::k_conv1x1_split_lds<2, 1>), dim3((unsigned)grid, (unsigned)n_splits), dim3(64), 0, (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL((pod::k_conv1x1_split<2, POD_C1_RING>), dim3((unsigned)grid, (unsigned)n_splits), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    if (n_splits > 1) {
        const int64_t n4 = P_out * Cout / 4;
        int64_t blocks = (n4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(pod::k_conv1x1_reduce, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, n_splits, P_out * Cout, bias, residual, y, n4,
                           Cout, relu, out_amax);
        POD_CHECK_LAUNCH();
    }
    return POD_OK;
}

// The fixed-order sum of channels-last partial outputs as its own entry point (pod_wino_conv3x3_split_partial's partials when the
// consumer wants channels-last, not planes): y = act(sum_s partials[s] + bias + residual), n = pixels * Cout floats.
extern "C" int pod_reduce_partials(const float* partials, int32_t n_splits, int64_t split_stride, const float* bias, const float* residual, float* y, int64_t n,
                                   int32_t Cout, int32_t relu, float* out_amax, pod_stream_t stream) {
    if (!partials || !y || n_splits < 1 || n_splits > 16 || n < 0 || (n & 3) != 0 || Cout < 4 || (Cout & 3) != 0 || n % Cout != 0) return POD_E_INVALID;
    if (n_splits > 1 && (split_stride < n || (split_stride & 3) != 0)) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(partials) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(residual)) & 15u) != 0)
        return POD_E_INVALID;
    if (n == 0) return POD_OK;
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pod::k_conv1x1_reduce, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, n_splits, split_stride, bias, residual, y, n / 4,
                       Cout, relu, out_amax);
    POD_CHECK_LAUNCH();
    return POD_OK;
}


#ifdef POD_C1_TRACE
#include <algorithm>
#include <stdio.h>
#include <vector>
extern "C" int pod_c1_trace_dump(void) {   // experiment builds only (not in include/pod_mi355x.h): the LAST launch's wavefronts
    static long long host[8192 * 4], cyc[8192 * 4];
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c1_trace), sizeof(host)) != hipSuccess) return POD_E_LAUNCH;
    if (hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_c1_cycles), sizeof(cyc)) != hipSuccess) return POD_E_LAUNCH;
    long long t0 = 0, t_end = 0;
    std::vector<long long> start, first, loop, epi, total, mhz;
    for (int i = 0; i < 8192; ++i)
        if (host[i * 4 + 3]) {
            if (!t0 || host[i * 4] < t0) t0 = host[i * 4];
            if (host[i * 4 + 3] > t_end) t_end = host[i * 4 + 3];
        }
    for (int i = 0; i < 8192; ++i)
        if (host[i * 4 + 3]) {
            start.push_back(host[i * 4] - t0);
            first.push_back(host[i * 4 + 1] - host[i * 4]);
            loop.push_back(host[i * 4 + 2] - host[i * 4 + 1]);
            epi.push_back(host[i * 4 + 3] - host[i * 4 + 2]);
            total.push_back(host[i * 4 + 3] - host[i * 4]);
            mhz.push_back((cyc[i * 4 + 2] - cyc[i * 4 + 1]) * 100 / std::max(1LL, host[i * 4 + 2] - host[i * 4 + 1]));
        }
    auto q = [](std::vector<long long>& v, double f) { std::sort(v.begin(), v.end()); return v.empty() ? 0LL : v[(size_t)(f * (v.size() - 1))]; };
    fprintf(stderr, "c1 trace: %zu wavefronts, first start to last end %lld0 ns; 10-ns ticks  min / median / 90%% / max\n", start.size(), t_end - t0);
    std::vector<long long>* vs[6] = {&start, &first, &loop, &epi, &total, &mhz};
    const char* names[6] = {"start after the first", "start -> first k-step done", "rest of the k loop", "epilogue (stores landed)", "whole wavefront", "s_memtime MHz in the k loop"};
    for (int k = 0; k < 6; ++k) fprintf(stderr, "  %-28s %5lld %5lld %5lld %5lld\n", names[k], q(*vs[k], 0.0), q(*vs[k], 0.5), q(*vs[k], 0.9), q(*vs[k], 1.0));
    static long long zero[8192 * 4];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_c1_trace), zero, sizeof(zero)) == hipSuccess ? POD_OK : POD_E_LAUNCH;
}
#endif
