// 1x1 fp32 convolution on channels-last activations, with the element-wise tail of detectron2's bottleneck fused into the store:
//     out[o][k] = act( sum_c in[src(o)][c] * w[k][c]  + bias[k]  + residual[o][k] )
// Replaces: BottleneckBlock.conv1 / conv3 / shortcut (1x1, stride 1 or 2 with STRIDE_IN_1X1) and FPN.lateral_convs of the
// backbone the reference builds (probabilistic_retinanet.py:60-66 -> detectron2 build_resnet_fpn_backbone), batch 1 -- together with
// the `out += shortcut; relu` of the block and the lateral + top-down sum of the FPN.  A 1x1 convolution on [pixel][C] data is the
// GEMM  [pixels x C] x [C x K]  with both operands contiguous along C: exactly the operand shape of v_mfma_f32_32x32x2_f32, so no
// layout change, no im2col and -- unlike the NCHW call into MIOpen it replaces -- no separate bias / residual / ReLU pass.
//
// Workgroup = 256 threads = 4 waves: 128 output pixels x 128 output channels; wave w owns pixels 32w .. 32w+31 and all 128
// channels (4 MFMA blocks of 32x32 = 64 accumulators).  Both operands go global -> LDS by LDS-DMA as full 128-byte lines (a row =
// one pixel's or one filter's 32 consecutive input channels = a SUPER-CHUNK), 256 rows per stage, two stages (64 KB: two
// workgroups per CU, the second one's MFMAs cover this one's barrier).  Rows are swizzled (16-byte part P of row r sits in sub-slot
// (P + r) & 7: the lane that fills sub-slot q asks for part (q - r) & 7), so the ds_read_b128 of 32 lanes with a 128-byte row
// stride is conflict-free.  Per super-chunk and wave: 64 MFMAs, 20 ds_read_b128, 8 DMA instructions, one barrier.
// Filters are used as they are stored: w is (K, C) row-major = conv.weight of a 1x1 convolution.  K and the pixel count need no
// padding: rows past the end are fetched with an out-of-range buffer offset (they read 0.0) and their stores are skipped.
#include "pod_wino.h"

namespace pod {

struct Conv1x1Params {
    const float* in;          // (n_in pixels, C)
    float* out;               // (n_out pixels, K)
    const float* w;           // (K, C)
    const float* bias;        // K or null
    const float* residual;    // (n_out pixels, K) or null
    int32_t n_in, n_out, Wi, Wo, stride, C, K, relu;
};

constexpr int C1_STAGE_BYTES = 256 * 128;      // 128 pixel rows + 128 filter rows of 128 B
constexpr int C1_LDS_BYTES = 2 * C1_STAGE_BYTES;

__global__ void __launch_bounds__(256, 2) k_conv1x1(const Conv1x1Params P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5, g = lane >> 3, q = lane & 7;
    const int pix0 = blockIdx.x * 128, k0 = blockIdx.y * 128;
    const int nsc = P.C >> 5;
    typedef __attribute__((address_space(3))) void lds_void;
    const auto in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, P.n_in * P.C * 4, 0x00020000);
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, P.K * P.C * 4, 0x00020000);
    // this lane's share of a stage fill: 4 pieces of pixel rows 32w + 8I + g and 4 of filter rows 32w + 8I + g, sub-slot q each
    int voff[4], uoff[4];
#pragma unroll
    for (int I = 0; I < 4; ++I) {
        const int r = 32 * w + 8 * I + g, part = (q - r) & 7;
        const int o = pix0 + r;
        int src = o;
        if (P.stride != 1) {
            const int oy = o / P.Wo;
            src = (oy * P.Wi + (o - oy * P.Wo)) * P.stride;
        }
        voff[I] = o < P.n_out ? (src * P.C + part * 4) * 4 : 0x7FFFFF00;
        const int k = k0 + r;
        uoff[I] = k < P.K ? (k * P.C + part * 4) * 4 : 0x7FFFFF00;
    }
    auto fill = [&](int stage, int sc, int I) {          // pieces I = 0..3: pixels, 4..7: filters
        float* dst = lds + stage * (C1_STAGE_BYTES / 4) + ((I < 4 ? 0 : 128) + 32 * w + 8 * (I & 3)) * 32;
        if (I < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, voff[I], sc * 128, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, uoff[I - 4], sc * 128, 0, 0);
    };
    // reads: lane (j, h) takes part 2c + h of row j (filters: 128 + 32 kb + j; pixels: 32 w + j) for chunk c of the super-chunk
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    uint32_t au[4], av[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        au[c] = lds_base + j * 128 + ((2 * c + h + j) & 7) * 16;
        av[c] = au[c] + 32 * w * 128;
    }
    f32x4 V[2], U[2][4];
#define C1_READ(buf, stage, c)                                                                                                   \
    do {                                                                                                                        \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(V[buf]) : "v"(av[c]), "i"((stage) * C1_STAGE_BYTES));                \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(U[buf][0]) : "v"(au[c]), "i"((stage) * C1_STAGE_BYTES + 16384));       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(U[buf][1]) : "v"(au[c]), "i"((stage) * C1_STAGE_BYTES + 16384 + 4096)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(U[buf][2]) : "v"(au[c]), "i"((stage) * C1_STAGE_BYTES + 16384 + 8192)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(U[buf][3]) : "v"(au[c]), "i"((stage) * C1_STAGE_BYTES + 16384 + 12288)); \
    } while (0)
    f32x16 acc[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kb][e] = 0.0f;

#pragma unroll
    for (int I = 0; I < 8; ++I) fill(0, 0, I);
    __builtin_amdgcn_s_waitcnt(WINO_WAIT_VM0);
    __builtin_amdgcn_s_barrier();
    C1_READ(0, 0, 0);
    // one super-chunk from stage `st` (compile-time: the read offsets are immediates) while the next one is fetched into the other
    auto super_chunk = [&](auto st_t, int sc) {
        constexpr int st = decltype(st_t)::value;
        const bool more = sc + 1 < nsc;
        wino_static_for([&](auto Cc) __attribute__((always_inline)) {
            constexpr int c = decltype(Cc)::value, buf = c & 1;
            __builtin_amdgcn_s_waitcnt(WINO_WAIT_LGKM0);                   // the operands of chunk c (read during chunk c - 1)
            if constexpr (c < 3) C1_READ(buf ^ 1, st, c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(U[buf][kb][s], V[buf][s], acc[kb], 0, 0, 0);
                    if (s == 1 && kb < 2 && more) fill(st ^ 1, sc + 1, 2 * c + kb);        // 2 of the next stage's 8 pieces per chunk
                    __builtin_amdgcn_sched_barrier(0);
                }
        }, std::make_integer_sequence<int, 4>{});
        __builtin_amdgcn_s_waitcnt(WINO_WAIT_VM0);                         // this wave's pieces of the next stage have landed
        __builtin_amdgcn_s_barrier();                                      // ... everyone's have, and everyone is done reading this stage
        if (more) C1_READ(0, st ^ 1, 0);
    };
    for (int sc = 0; sc < nsc; sc += 2) {
        super_chunk(std::integral_constant<int, 0>{}, sc);
        if (sc + 1 < nsc) super_chunk(std::integral_constant<int, 1>{}, sc + 1);
    }
#undef C1_READ
    // ---- store: accumulator register 4 g + e of block kb = channel k0 + 32 kb + 8 g + 4 h + e of pixel pix0 + 32 w + j
    const int o = pix0 + 32 * w + j;
    if (o >= P.n_out) return;
    float* orow = P.out + (int64_t)o * P.K;
    const float* rrow = P.residual ? P.residual + (int64_t)o * P.K : nullptr;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int k = k0 + 32 * kb + 8 * gg + 4 * h;
            if (k >= P.K) continue;
            f32x4 v = f32x4{acc[kb][4 * gg], acc[kb][4 * gg + 1], acc[kb][4 * gg + 2], acc[kb][4 * gg + 3]};
            if (P.bias) v += *reinterpret_cast<const f32x4*>(P.bias + k);
            if (rrow) v += *reinterpret_cast<const f32x4*>(rrow + k);
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<f32x4*>(orow + k) = v;
        }
}

}  // namespace pod

extern "C" int pod_conv1x1(const float* in, float* out, const float* w, const float* bias, const float* residual, int32_t Hi, int32_t Wi,
                           int32_t stride, int32_t C, int32_t K, int32_t relu, pod_stream_t stream) {
    if (!in || !out || !w || in == out || Hi < 1 || Wi < 1 || (stride != 1 && stride != 2) || C < 64 || (C & 31) != 0 || K < 4 || (K & 3) != 0)
        return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(bias) |
          reinterpret_cast<uintptr_t>(residual)) & 15u) != 0)
        return POD_E_INVALID;
    const int64_t Ho = (Hi + stride - 1) / stride, Wo = (Wi + stride - 1) / stride, n_in = (int64_t)Hi * Wi, n_out = Ho * Wo;
    if (n_in * C * 4 >= 0x7FFFFF00LL || (int64_t)K * C * 4 >= 0x7FFFFF00LL || n_out * K * 4 >= 0x7FFFFFFFLL * 4) return POD_E_INVALID;   // 32-bit buffer offsets
    static std::once_flag once[64];
    static hipError_t attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return POD_E_LAUNCH;
    std::call_once(once[dev], [dev] {
        attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(pod::k_conv1x1), hipFuncAttributeMaxDynamicSharedMemorySize, pod::C1_LDS_BYTES);
    });
    if (attr[dev] != hipSuccess) return POD_E_LAUNCH;
    pod::Conv1x1Params P;
    P.in = in; P.out = out; P.w = w; P.bias = bias; P.residual = residual;
    P.n_in = (int32_t)n_in; P.n_out = (int32_t)n_out; P.Wi = Wi; P.Wo = (int32_t)Wo; P.stride = stride; P.C = C; P.K = K; P.relu = relu;
    hipLaunchKernelGGL(pod::k_conv1x1, dim3((unsigned)((n_out + 127) / 128), (unsigned)((K + 127) / 128)), dim3(256), pod::C1_LDS_BYTES,
                       (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
