// K14 stem7x7_split + maxpool3x3s2_cl -- the ResNet stem (7x7 / stride 2 / pad 3 convolution, 3 -> 64 channels, FrozenBN folded, ReLU) and
// its 3x3 / stride 2 max-pool, channels-last out: the last MIOpen call and the last layout / element-wise passes ahead of the trunk.
//
// Replaces (reference: detectron2's BasicStem as probabilistic_retinanet.py:96-100 runs it, `features = self.backbone(images.tensor)`):
// conv1 + FrozenBN + ReLU + max_pool2d(3, 2, 1).  Before: MIOpen's conv (106 us on the 768 x 1344 frame) + pod_bias_act (21) +
// torch's NCHW max-pool (38) + a transposing copy to channels-last (12).
//
// The convolution as a GEMM with k13's arithmetic (round 5: 2-way f16 splits of the power-of-two-scaled operands, 3 partial products on
// v_mfma_f32_32x32x16_f16, fp32 accumulate; `in_amax` bounds the NORMALISED input) and k13's filter layout: K = 3 channels x 4 row pairs x (2 rows x 8 columns) = 12 k-steps
// of 16 -- the 7 x 7 window padded to 8 x 8 with zero weights, ordered so that a lane's 8 k values are 8 CONSECUTIVE input columns of
// one row (h = the row of the pair): k = ((c 4 + d2) 2 + h) 8 + dx, dy = 2 d2 + h.  Workgroup = one wavefront = an 8 x 8 tile of output
// pixels x 64 channels; its 22 x 24 x 3 input patch is loaded once into LDS (zero outside the image) and every MFMA fragment is four
// ds_read_b64 of it; filter terms straight from L2 (3-deep register ring); epilogue through LDS in whole lines as k13's.
#include "pod_wino.h"

namespace pod {

constexpr int ST_KS_U16 = 2 * 2 * 256, ST_TOP = 14;     // u16 values of one (32-channel block, k-step); scaled abs-max of both operands in [2^14, 2^15)
typedef uint32_t st_u32x4 __attribute__((ext_vector_type(4)));
typedef float st_f32x2 __attribute__((ext_vector_type(2)));

constexpr int ST_KS = 12;                 // k-steps
constexpr int ST_PR = 22, ST_PC = 24;     // patch rows (2 * 7 + 7 + 1), columns (2 * 7 + 8, padded to 24)

struct StemParams {
    const void* x;            // (3, Hi, Wi) planes: fp32, or uint8 (x_u8) -- the frame as the data loader hands it over
    const float* mean;        // per-channel pixel mean / std (PR:96 `self.normalizer`): the patch load computes (x - mean) / std, in fp32 as torch
    const float* std_;        //   does; null: x is already normalised
    int32_t Hi, Wi, x_u8;     // extent of x; outside it -- the zero padding of the conv AND the frame's padding to a multiple of 32 -- the input is 0
    float* y;                 // (Ho * Wo, 64) channels-last
    const uint16_t* Ws;       // pre-split filter: [cout block 2][k-step 12][term 2][h 2][i32 32][8 f16], then the abs-max word (16-byte trailer)
    const float* in_amax;     // device word >= max |normalised input|
    float* out_amax;          // null, or a device word max'ed with |every value stored|
    const float* bias;
    int32_t H, W, Ho, Wo, tiles_x, relu;
};

// weight (64, 3, 7, 7) fp32 -> Ws: the (64 x 192) GEMM matrix in the k order above, two nearest-even f16 terms per scaled value
__global__ void __launch_bounds__(256) k_stem_filter_amax(const float* __restrict__ w, float* __restrict__ amax) {
    float m = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * 3 * 49; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    wino_publish_amax1(amax, m);
}
__global__ void __launch_bounds__(256) k_stem_filter_split(const float* __restrict__ w, uint16_t* __restrict__ Ws, const float* __restrict__ amax) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (cout, pair of k)
    if (t >= 64 * 96) return;
    const int co = t / 96, k = 2 * (t % 96);
    const int ks = k >> 4, h = (k >> 3) & 1, dx = k & 7, c = ks >> 2, dy = 2 * (ks & 3) + h;
    const float lo = (dy < 7 && dx < 7) ? w[((co * 3 + c) * 7 + dy) * 7 + dx] : 0.f;
    const float hi = (dy < 7 && dx + 1 < 7) ? w[((co * 3 + c) * 7 + dy) * 7 + dx + 1] : 0.f;
    uint32_t terms[2];
    wino_f16_split2(lo, hi, wino_pow2_scale(*amax, ST_TOP), terms);
    const int cb = co >> 5, i32 = co & 31;
#pragma unroll
    for (int term = 0; term < 2; ++term) {
        uint16_t* d = Ws + ((((int64_t)cb * ST_KS + ks) * 2 + term) * 2 + h) * 256 + i32 * 8 + dx;
        d[0] = (uint16_t)(terms[term] & 0xFFFFu);
        d[1] = (uint16_t)(terms[term] >> 16);
    }
}

template <int I>
using st_ic = std::integral_constant<int, I>;

__global__ void __launch_bounds__(64, 2) k_stem7x7_split(const StemParams P) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 64];                 // patch (3 x 22 x 24 floats), then the output tile (64 pixels x 64 channels)
    const int lane = threadIdx.x & 63, i32 = lane & 31, h = lane >> 5;
    const int tyb = blockIdx.x / P.tiles_x, txb = blockIdx.x - tyb * P.tiles_x;
    const int oy0 = 8 * tyb, ox0 = 8 * txb;
    // ---- the input patch: rows 2 oy0 - 3 .., columns 2 ox0 - 3 .., zero outside the image
    {
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
#pragma unroll
        for (int it = 0; it < (3 * ST_PR * ST_PC + 63) / 64; ++it) {            // 25 loads per lane, all in flight before the first is parked
            const int e = lane + 64 * it;
            if (e >= 3 * ST_PR * ST_PC) break;
            const int c = e / (ST_PR * ST_PC), r = (e - c * ST_PR * ST_PC) / ST_PC, col = e - c * ST_PR * ST_PC - r * ST_PC;
            const int iy = iy0 + r, ix = ix0 + col;
            float v = 0.f;
            if (iy >= 0 && iy < P.Hi && ix >= 0 && ix < P.Wi) {
                const int64_t at = ((int64_t)c * P.Hi + iy) * P.Wi + ix;
                v = P.x_u8 ? (float)static_cast<const uint8_t*>(P.x)[at] : static_cast<const float*>(P.x)[at];
                if (P.mean) v = (v - P.mean[c]) / P.std_[c];
            }
            lds[e] = v;
        }
    }
    const uint16_t* __restrict__ const wa = P.Ws + (h * 32 + i32) * 8;          // + cb * 12 * 1024 + ks * 1024 + term * 512
    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wino_pow2_scale(wino_read_amax(P.in_amax), ST_TOP))));
    // this lane's two pixels (pb = 0, 1): tile pixel q = 32 pb + i32 = (q >> 3, q & 7); patch offset of its window's row h, column 0
    int base[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        const int q = 32 * pb + i32;
        base[pb] = (2 * (q >> 3) + h) * ST_PC + 2 * (q & 7);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][pb][r] = 0.f;
    st_u32x4 wf[3][2][2];
    auto load_w = [&](auto buf_t, int ks) __attribute__((always_inline)) {
        constexpr int buf = decltype(buf_t)::value;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t) wf[buf][cb][t] = *reinterpret_cast<const st_u32x4*>(wa + (cb * ST_KS + ks) * ST_KS_U16 + t * 512);
    };
    auto step = [&](auto ks_t) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_t)::value, buf = ks % 3, c = ks >> 2, d2 = ks & 3;
        st_u32x4 at[2][2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const float* row = lds + c * ST_PR * ST_PC + 2 * d2 * ST_PC + base[pb];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const st_f32x2 v = *reinterpret_cast<const st_f32x2*>(row + 2 * i);
                float lo = v.x, hi = v.y;
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
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
                    acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, wf[buf][cb][sa]), __builtin_bit_cast(wino_f16x8, at[pb][sb]), acc[cb][pb], 0, 0, 0);
        }
    };
    load_w(st_ic<0>{}, 0);
    load_w(st_ic<1>{}, 1);
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the patch is written (one wavefront: no barrier)
    wino_static_for([&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        if constexpr (ks + 2 < ST_KS) load_w(st_ic<(ks + 2) % 3>{}, ks + 2);
        step(KS);
    }, std::make_integer_sequence<int, ST_KS>{});

    // ---- epilogue in whole lines through LDS (k13's): [pixel 64][chunk position 16][16 B], position = chunk ^ (pixel & 15)
    const int oc = lane & 15, op = lane >> 4;
    f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias) b4 = *reinterpret_cast<const f32x4*>(P.bias + 4 * oc);
    const float inv1 = wino_pow2_inverse(sx) * wino_pow2_inverse(wino_pow2_scale(*reinterpret_cast<const float*>(P.Ws + 64 * 192 * 2), ST_TOP));
    const f32x4 inv = f32x4{inv1, inv1, inv1, inv1};
    float lmax = 0.0f;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pix = 32 * pb + i32, c = 8 * cb + 2 * q + h;
                *reinterpret_cast<f32x4*>(lds + pix * 64 + 4 * (c ^ (i32 & 15))) =
                    f32x4{acc[cb][pb][4 * q], acc[cb][pb][4 * q + 1], acc[cb][pb][4 * q + 2], acc[cb][pb][4 * q + 3]};
            }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int pix = 4 * j + op, oy = oy0 + (pix >> 3), ox = ox0 + (pix & 7);
        f32x4 v = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(lds + pix * 64 + 4 * (oc ^ (pix & 15))), inv, b4);
        if (P.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (oy < P.Ho && ox < P.Wo) {
            *reinterpret_cast<f32x4*>(P.y + ((int64_t)oy * P.Wo + ox) * 64 + 4 * oc) = v;
            lmax = fmaxf(fmaxf(lmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    if (P.out_amax) wino_publish_amax(P.out_amax, lmax);
}

// max_pool2d(kernel 3, stride 2, padding 1) of a channels-last map: 16 B (4 channels) per lane, the window's taps that lie inside the map
__global__ void __launch_bounds__(256) k_maxpool3x3s2_cl(const float* __restrict__ x, float* __restrict__ y, int32_t Hi, int32_t Wi, int32_t Hp, int32_t Wp, int32_t C4) {
    const int64_t n = (int64_t)Hp * Wp * C4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c4 = (int)(i % C4);
        const int64_t p = i / C4;
        const int py = (int)(p / Wp), px = (int)(p - (int64_t)py * Wp);
        f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * py + dy;
            if (iy < 0 || iy >= Hi) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * px + dx;
                if (ix < 0 || ix >= Wi) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((int64_t)iy * Wi + ix) * C4 + c4) * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
    }
}

// The patch matrix of a 3x3 / stride 2 / padding 1 convolution on a channels-last map: row (oy, ox) = the nine taps' C-vectors in (ty, tx)
// order, zeros where a tap falls outside the map -- pod_conv1x1_split with Cin = 9 C then IS the convolution (FPN's LastLevelP6P7: 252 and
// 66 output pixels at the benchmark frame, where an implicit-GEMM kernel of its own would be a few workgroups).  relu: max(x, 0) on the
// way (p7 reads relu(p6)).  One thread per 16 bytes: whole 128-byte lines in, whole lines out.
__global__ void __launch_bounds__(256) k_im2col3x3s2_cl(const float* __restrict__ x, float* __restrict__ y, int Hi, int Wi, int Ho, int Wo, int C4, int relu) {
    const int64_t n = (int64_t)Ho * Wo * 9 * C4, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int tap = (int)(r % 9);
        r /= 9;
        const int ox = (int)(r % Wo), oy = (int)(r / Wo);
        const int iy = 2 * oy - 1 + tap / 3, ix = 2 * ox - 1 + tap % 3;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
            v = *reinterpret_cast<const f32x4*>(x + (((int64_t)iy * Wi + ix) * C4 + c4) * 4);
            if (relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

}  // namespace pod

extern "C" int pod_stem7x7_filter_split(const float* weight, void* Ws, pod_stream_t stream) {
    if (!weight || !Ws || (reinterpret_cast<uintptr_t>(Ws) & 15u) != 0) return POD_E_INVALID;
    float* amax = reinterpret_cast<float*>(reinterpret_cast<char*>(Ws) + 64 * 192 * 4);        // the 16-byte trailer behind the 2 x 64 x 192 f16 terms
    if (hipMemsetAsync(amax, 0, 16, (hipStream_t)stream) != hipSuccess) return POD_E_LAUNCH;
    hipLaunchKernelGGL(pod::k_stem_filter_amax, dim3(8), dim3(256), 0, (hipStream_t)stream, weight, amax);
    POD_CHECK_LAUNCH();
    hipLaunchKernelGGL(pod::k_stem_filter_split, dim3((64 * 96 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, reinterpret_cast<uint16_t*>(Ws), amax);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_stem7x7_split(const void* x, int32_t x_is_u8, int32_t H_img, int32_t W_img, const float* mean, const float* stddev, float* y, const void* Ws,
                                 const float* bias, int32_t H, int32_t W, int32_t relu, const float* in_amax, float* out_amax, pod_stream_t stream) {
    if (!x || !y || !Ws || !in_amax || ((reinterpret_cast<uintptr_t>(in_amax) | reinterpret_cast<uintptr_t>(out_amax)) & 3u) != 0 || x == static_cast<const void*>(y) || H < 1 || W < 1 || H > 16384 || W > 16384 || H_img < 1 || W_img < 1 || H_img > H || W_img > W)
        return POD_E_INVALID;
    if ((mean == nullptr) != (stddev == nullptr)) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(Ws) | reinterpret_cast<uintptr_t>(bias)) & 15u) != 0 ||
        (!x_is_u8 && (reinterpret_cast<uintptr_t>(x) & 3u) != 0))
        return POD_E_INVALID;
    pod::StemParams P;
    P.x = x; P.x_u8 = x_is_u8 ? 1 : 0; P.Hi = H_img; P.Wi = W_img; P.mean = mean; P.std_ = stddev;
    P.y = y; P.Ws = reinterpret_cast<const uint16_t*>(Ws); P.bias = bias; P.H = H; P.W = W; P.relu = relu; P.in_amax = in_amax; P.out_amax = out_amax;
    P.Ho = (H - 1) / 2 + 1; P.Wo = (W - 1) / 2 + 1;
    P.tiles_x = (P.Wo + 7) / 8;
    const int64_t grid = (int64_t)P.tiles_x * ((P.Ho + 7) / 8);
    if (grid > 0x7FFFFFFFLL) return POD_E_INVALID;
    hipLaunchKernelGGL(pod::k_stem7x7_split, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_maxpool3x3s2_cl(const float* x, float* y, int32_t H, int32_t W, int32_t C, pod_stream_t stream) {
    if (!x || !y || x == y || H < 1 || W < 1 || C < 4 || (C & 3) != 0) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) != 0) return POD_E_INVALID;
    const int32_t Hp = (H - 1) / 2 + 1, Wp = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)Hp * Wp * (C / 4);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pod::k_maxpool3x3s2_cl, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H, W, Hp, Wp, C / 4);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_im2col3x3s2_cl(const float* x, float* y, int32_t H, int32_t W, int32_t C, int32_t relu, pod_stream_t stream) {
    if (!x || !y || x == y || H < 1 || W < 1 || H > 16384 || W > 16384 || C < 4 || (C & 3) != 0) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) != 0) return POD_E_INVALID;
    const int32_t Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t n = (int64_t)Ho * Wo * 9 * (C / 4);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(pod::k_im2col3x3s2_cl, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, H, W, Ho, Wo, C / 4, relu ? 1 : 0);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
