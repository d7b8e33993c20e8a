// 3x3 / stride 1 / pad 1 fp32 convolution of the probabilistic RetinaNet head's subnets (probabilistic_retinanet.py:403-427:
// four conv3x3(256 -> 256) + ReLU + Dropout per subnet, evaluated for every MC run on every FPN level) as ONE launch over
// all levels and all runs: Winograd F(2x2, 3x3) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), bias + ReLU + dropout
// fused into the store.
//
// Why Winograd: a direct fp32 convolution is bounded by the 157 TFLOP/s fp32 MFMA peak (MIOpen's implicit GEMM reaches
// 0.83 of it on the p3 maps and nothing can reach more than 1.0); F(2x2, 3x3) needs 16 multiplies per 2x2 outputs and
// (c, k) pair instead of 36, so the same matrix cores deliver up to 2.25x the direct-convolution rate, in fp32 throughout
// (the transforms only add and subtract; the filter transform has two halvings).  Error vs direct fp32: ~1e-6 relative.
//
//   Y = At [ (G g Gt) . (Bt d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs, "." summed over c
//
// Data layout (channels-last): activations are [pixel][C] fp32; every (level, run) image of a launch lives in the same
// buffer, a table of 16x16-pixel output blocks (int4 {first pixel of the image in `in`, in `out`, H << 16 | W, by << 16 | bx})
// says where.  Filters are transformed once (pod_wino_filter_transform) into the image the kernel's LDS stages want.
// The predictor convolutions (cls_score, bbox_pred, cls_var, bbox_cov: K = 63 / 36 / 90 real channels) write NCHW planes,
// the layout K1 streams, straight from the staging tile.
//
// Workgroup = 256 threads = 4 waves, one per SIMD: 64 tiles (8x8 tiles = 16x16 output pixels) x 64 output channels x the
// 16 Winograd positions.  Wave a owns ROW a of the 4x4 position grid for all 64 tiles and all 64 channels: 4 positions x
// (2 tile blocks x 2 channel blocks) of 32x32 = 16 MFMA blocks = 256 accumulator registers.  A row of Bt d is one sum or
// difference of two patch rows, so a lane transforms its two tiles with 16 packed adds per sub-step, every transformed value
// and every filter operand feeds two MFMAs (the split "32 tiles x 32 channels x all 16 positions per wave" has no operand
// reuse and twice the adds: 6 % slower).  Per chunk of 8 input channels the workgroup stages the raw 18x18-pixel input patch
// (no transformed copy exists anywhere) and the 16 x 8 x 64 filter slab (LDS-DMA) in LDS, triple-buffered against the 64 MFMAs
// of the chunk.  The output transform reduces each wave's row over its 4 positions in registers, parks the row sums in LDS and
// combines the four rows in the store pass.  Bank layout: see the address comments.
//
// Launch: blockIdx & 7 is the XCD (round-robin dispatch); an XCD always works on the same 64-channel filter slice (1 MB
// for C = 256), which therefore stays in that XCD's 4 MB L2 while the activations stream through.
#include <mutex>

#include "pod_device.h"

namespace pod {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t STREAM_DROPOUT_CONV = 0x64726f70u;   // = STREAM_DROPOUT of k8_model_ops.hip: same mask as pod_bias_act

constexpr int WINO_U_FLOATS = 16 * 2 * 2 * 64 * 2;      // filter slab of a chunk: [p][sp][h][j][2]            32 KB
constexpr int WINO_R_UNITS = 18 * 20;                    // 8-byte units of one (sp, h) plane of the raw patch
constexpr int WINO_R_FLOATS = 2 * 2 * WINO_R_UNITS * 2;  // [sp][h][row 18][parity 2][col/2: 9 (+1 pad)][2]    11.25 KB
constexpr int WINO_STAGE_FLOATS = WINO_U_FLOATS + WINO_R_FLOATS;
constexpr int WINO_LDS_BYTES = 8 * 64 * 65 * 4;   // 133 120 B of the CU's 160 KB: the output staging (three K-loop stages: 132 864 B)

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
};

// Filter transform U = G g Gt, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], written as the LDS image of the main kernel:
// U[ks][chunk][p][sp][h][j][s2] = U_p[c = 8 chunk + 4 h + 2 sp + s2][k = 64 ks + j]; channels >= K are zero.
__global__ void __launch_bounds__(256) k_wino_filter(const float* __restrict__ w, float* __restrict__ U, int32_t K, int32_t C, int32_t Kpad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Kpad * C) return;
    const int k = (int)(t / C), c = (int)(t % C);
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = k < K ? w[((int64_t)k * C + c) * 9 + i] : 0.0f;
    float t0[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        t0[0][j] = g[0][j];
        t0[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t0[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t0[3][j] = g[2][j];
    }
    const int nchunk = C / 8, ks = k >> 6, j64 = k & 63, ch = c >> 3, cc = c & 7;
    const int h = cc >> 2, sp = (cc >> 1) & 1, s2 = cc & 1;
    float* dst = U + ((int64_t)ks * nchunk + ch) * WINO_U_FLOATS + ((sp * 2 + h) * 64 + j64) * 2 + s2;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = t0[a][0], u1 = 0.5f * (t0[a][0] + t0[a][1] + t0[a][2]), u2 = 0.5f * (t0[a][0] - t0[a][1] + t0[a][2]),
                    u3 = t0[a][2];
        dst[(a * 4 + 0) * 512] = u0;
        dst[(a * 4 + 1) * 512] = u1;
        dst[(a * 4 + 2) * 512] = u2;
        dst[(a * 4 + 3) * 512] = u3;
    }
}

__global__ void __launch_bounds__(256, 1) k_wino_conv3x3(const WinoParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const int ks = xcd % P.KS;
    const int tb = (int)(blockIdx.x >> 3) * (8 / P.KS) + xcd / P.KS;
    if (tb >= P.n_blocks) return;
    const int4 desc = P.blocks[tb];
    const int64_t base_px = desc.x, out_px = desc.y;
    const int H = desc.z >> 16, W = desc.z & 0xFFFF, y0 = (desc.w >> 16) * 16, x0 = (desc.w & 0xFFFF) * 16;
    const int nchunk = P.C >> 3;

    // ---- global -> LDS plan of a chunk: 8 float4 of the filter slab (a straight copy) + up to 3 float4 of the raw patch, as
    // buffer loads: scalar base + 32-bit lane offset + scalar chunk offset (no 64-bit address arithmetic in the loop), and
    // an out-of-range offset reads as 0.0 -- that IS the zero padding of the convolution (and the "no item" case).
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.U + ((int64_t)ks * nchunk) * WINO_U_FLOATS), 0,
                                                          nchunk * WINO_U_FLOATS * 4, 0x00020000);
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in + base_px * P.in_stride), 0, H * W * P.in_stride * 4,
                                                          0x00020000);
    int roff[3], rdst[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int q = tid + 256 * r;              // (pixel of the 18x18 patch, channel half h)
        const int pix = q >> 1, h = q & 1, py = pix / 18, px = pix - py * 18;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = q < 648 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        roff[r] = ok ? ((gy * W + gx) * P.in_stride + 4 * h) * 4 : 0x7FFFFF00;
        // 8-byte unit of the pixel inside an (sp, h) plane: rows of 20 units, even columns first -- the 32 lanes of an MFMA
        // operand read (tile row stride 40 = 8 mod 32, tile column stride 1) hit 32 different units.  Unit 9 of a row is
        // padding nobody reads: threads without an item store there (no branch in the loop).
        rdst[r] = WINO_U_FLOATS + (q < 648 ? (h * WINO_R_UNITS + py * 20 + (px & 1) * 10 + (px >> 1)) : 9) * 2;
    }
    // A chunk's copy in 11 pieces (loads) / 14 pieces (stores): piece i rides behind the i-th MFMA of a phase.
    f32x4 gr[3];
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    typedef __attribute__((address_space(3))) void lds_void;
    auto fetch_piece = [&](float* stage, int ch, int i, f32x4(&sr)[3]) {
        // filter slab: LDS-DMA, 1 KB per wavefront instruction straight into the stage (the slab is stored as its LDS image)
        if (i < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(u_rsrc, (lds_void*)(stage + i * 1024 + wave_u * 256), 16, tid * 16,
                                                 ch * (WINO_U_FLOATS * 4) + i * 4096, 0, 0);
        else if (i < 11) sr[i - 8] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, roff[i - 8], ch * 32, 0));
    };
    auto stash_piece = [&](float* stage, int i, const f32x4(&sr)[3]) {
        if (i >= 8 && i < 14) {
            const int r = (i - 8) >> 1, sp = (i - 8) & 1;
            *reinterpret_cast<f32x2*>(stage + rdst[r] + sp * (2 * WINO_R_UNITS * 2)) = sp ? f32x2{sr[r].z, sr[r].w} : f32x2{sr[r].x, sr[r].y};
        }
    };

    // ---- operand addresses of this lane.  Wavefront `a` owns ROW a of the 4x4 Winograd position grid (positions 4a .. 4a+3) for
    // all 64 tiles (two 32-tile blocks, tb) and all 64 output channels (two 32-channel blocks, kb): 4 x 2 x 2 MFMA blocks = 256
    // accumulators.  Row a of Bt d is one sum or difference of two patch rows:
    //     a = 0: d0 - d2      a = 1: d1 + d2      a = 2: d2 - d1      a = 3: d1 - d3
    // = x0 + s x1 with wave-uniform row offsets and sign, so each lane transforms its two tiles with 2 x (4 + 4) packed adds per
    // sub-step (a quarter of Bt d B) and every transformed value feeds TWO MFMAs, every filter operand two as well.
    const int i32 = lane & 31, h = lane >> 5;
    const int a = __builtin_amdgcn_readfirstlane(wave);
    const int row0 = a == 0 ? 0 : a == 2 ? 2 : 1, row1 = a == 2 ? 1 : a == 3 ? 3 : 2;
    const float sgn = a == 1 ? 1.0f : -1.0f;
    const int a_base = WINO_U_FLOATS + (h * WINO_R_UNITS + 2 * (i32 >> 3) * 20 + (i32 & 7)) * 2;   // tile (i32>>3, i32&7) of block tb = 0
    const int a_r0 = a_base + row0 * 40, a_r1 = a_base + row1 * 40;     // + tb*320 + ((b&1)*10 + (b>>1))*2 + sp*1440
    const int b_base = (h * 64 + i32) * 2 + a * (4 * 512);              // + b*512 + kb*64 + sp*256

    f32x16 acc[16];                                                      // [b][tb][kb]
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.0f;

    f32x2 x[16], uA[8], uB[8], vA[8], vB[8], t[4];                       // x[tb][row][b], u[b][kb], v[tb][b]
    // operand reads of a sub-step in 12 pieces of two 8-byte reads (one ds_read2): the patch rows first (the transform needs them)
    auto read_piece = [&](const float* stage, int sp, f32x2(&u)[8], int i) {
        if (i < 8) {
            const int tb = i >> 2, row = (i >> 1) & 1, b0 = (i & 1) * 2;
#pragma unroll
            for (int b = b0; b < b0 + 2; ++b)
                x[(tb * 2 + row) * 4 + b] = *reinterpret_cast<const f32x2*>(stage + (row ? a_r1 : a_r0) + tb * 320 + ((b & 1) * 10 + (b >> 1)) * 2 +
                                                                             sp * (2 * WINO_R_UNITS * 2));
        } else if (i < 12) {
            const int b = i - 8;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) u[b * 2 + kb] = *reinterpret_cast<const f32x2*>(stage + b_base + b * 512 + kb * 64 + sp * 256);
        }
    };
    // row a of V = Bt d B for the lane's two tiles, two channels at once: 4 pieces (tile block x {t = x0 + s x1, V = t B})
    auto transform_piece = [&](f32x2(&v)[8], int i) {
        const int tb = i >> 1;
        if ((i & 1) == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x2 p = x[(tb * 2 + 0) * 4 + b], q = x[(tb * 2 + 1) * 4 + b];
                t[b] = f32x2{fmaf(sgn, q.x, p.x), fmaf(sgn, q.y, p.y)};
            }
        } else {
            v[tb * 4 + 0] = t[0] - t[2];
            v[tb * 4 + 1] = t[1] + t[2];
            v[tb * 4 + 2] = t[2] - t[1];
            v[tb * 4 + 3] = t[1] - t[3];
        }
    };

    // Three LDS stages: chunk ch is read from stage ch % 3 while chunk ch+2 is written to (ch+2) % 3 (last read during chunk
    // ch-1, i.e. before the barrier that ended it).  Inside the chunk every MFMA gets at most one memory instruction and a few
    // adds behind it (the order is pinned in the source, sched_barrier after every MFMA + its piece of the other work): the
    // four waves of the workgroup run in lock step, so memory instructions issued in a burst queue up behind each other at the
    // LDS / the texture path and stall the in-order instruction streams (measured: bursts cost their full LDS / TA throughput
    // time on top of the MFMA time), while one per 64-cycle MFMA mostly disappears behind it.  The loop is uniform: the last
    // chunks re-fetch / re-read harmlessly.
    // MFMA j of a sub-step: k-step j >> 4, accumulator m = j & 15 = (b, tb, kb)
#define WINO_MFMA(V, U, j)                                                                                                        \
    acc[(j) & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(                                                                         \
        (j) < 16 ? V[(((j) >> 1) & 1) * 4 + (((j) & 15) >> 2)].x : V[(((j) >> 1) & 1) * 4 + (((j) & 15) >> 2)].y,                 \
        (j) < 16 ? U[((((j) & 15) >> 2)) * 2 + ((j) & 1)].x : U[((((j) & 15) >> 2)) * 2 + ((j) & 1)].y, acc[(j) & 15], 0, 0, 0)
    float* cur = lds;
    float* nxt = lds + WINO_STAGE_FLOATS;
    float* nn = lds + 2 * WINO_STAGE_FLOATS;
    {   // prologue: chunks 0 and 1 in flight together
        f32x4 hr[3];
#pragma unroll
        for (int i = 0; i < 11; ++i) fetch_piece(cur, 0, i, gr);
#pragma unroll
        for (int i = 0; i < 11; ++i) fetch_piece(nxt, nchunk > 1 ? 1 : 0, i, hr);
#pragma unroll
        for (int i = 0; i < 14; ++i) stash_piece(cur, i, gr);
#pragma unroll
        for (int i = 0; i < 14; ++i) stash_piece(nxt, i, hr);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 12; ++i) read_piece(cur, 0, uA, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) transform_piece(vA, i);
    __builtin_amdgcn_s_waitcnt(0xC07F);                // lgkmcnt(0) here, or the loop header waits for its own new reads
    __builtin_amdgcn_sched_barrier(0);
    for (int ch = 0; ch < nchunk; ++ch) {              // branch-free body: 64 MFMAs, one barrier
        const int fch = ch + 2 < nchunk ? ch + 2 : nchunk - 1;
        // ---- sub-step 0: MFMAs of (ch, 0); read (ch, 1), transform it; fetch chunk ch+2
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            WINO_MFMA(vA, uA, j);
            if (j < 12) read_piece(cur, 1, uB, j);
            if (j >= 12 && j < 23) fetch_piece(nn, fch, j - 12, gr);
            if (j >= 24 && j < 28) transform_piece(vB, j - 24);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sub-step 1: MFMAs of (ch, 1); read (ch+1, 0), transform it; write the patch of chunk ch+2
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            WINO_MFMA(vB, uB, j);
            if (j < 12) read_piece(nxt, 0, uA, j);
            if (j >= 14 && j < 20) stash_piece(nn, j - 14 + 8, gr);
            if (j >= 24 && j < 28) transform_piece(vA, j - 24);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                // (waits for this wave's LDS traffic, so nothing is pending at the header)
        float* tmp = cur;
        cur = nxt;
        nxt = nn;
        nn = tmp;
    }
#undef WINO_MFMA
    __syncthreads();                                   // every wave is done reading the stages: they become the output staging

    // ---- output transform Y = At M A, At = [[1,1,1,0],[0,1,-1,-1]].  Lane: block row (tile) = (reg&3) + 8 (reg>>2) + 4 (lane>>5),
    // column (channel) = lane & 31.  Every wave reduces its row over b (R0 = m0 + m1 + m2, R1 = m1 - m2 - m3) and parks
    // R[a][x][tile][channel] in LDS (128 KB); the store pass combines the four rows in a fixed order:
    //     Y[0][x] = (R[0][x] + R[1][x]) + R[2][x]          Y[1][x] = (R[1][x] - R[2][x]) - R[3][x]
    const int LD = P.k_planes > 0 ? 65 : 64;   // tile row of the staging: 16-byte reads along k (64) or scalar reads, bank-spread (65)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float m0 = acc[0 * 4 + tb * 2 + kb][reg], m1 = acc[1 * 4 + tb * 2 + kb][reg], m2 = acc[2 * 4 + tb * 2 + kb][reg],
                            m3 = acc[3 * 4 + tb * 2 + kb][reg];
                const int tile = (4 * tb + (reg >> 2)) * 8 + (reg & 3) + 4 * h;
                float* o = lds + ((a * 2) * 64 + tile) * LD + kb * 32 + i32;
                o[0] = m0 + m1 + m2;
                o[64 * LD] = m1 - m2 - m3;
            }
    __syncthreads();
    if (P.k_planes > 0) {
        // NCHW planes: thread -> (channel, row of the block, 4 pixels along x); 64-byte runs per (channel, row)
        const int64_t HW = (int64_t)H * W;
        float* plane0 = P.out + out_px * P.k_planes;
        const bool vec = (W & 3) == 0 && ((out_px * P.k_planes) & 3) == 0;
#pragma unroll 2
        for (int it = 0; it < 16; ++it) {
            const int idx = it * 256 + tid, k = idx >> 6, oy = (idx >> 2) & 15, ox = (idx & 3) * 4;
            const int kg = ks * 64 + k, gy = y0 + oy, gx = x0 + ox;
            if (kg >= P.k_planes || gy >= H || gx >= W) continue;
            const float bias = P.bias ? P.bias[kg] : 0.0f;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tile = (oy >> 1) * 8 + ((ox + e) >> 1), xx = e & 1;
                const float* r = lds + (xx * 64 + tile) * 65 + k;              // R[a][xx][tile][k] at + a * 2 * 64 * 65
                y[e] = (oy & 1) == 0 ? (r[0] + r[2 * 64 * 65]) + r[4 * 64 * 65] : (r[2 * 64 * 65] - r[4 * 64 * 65]) - r[6 * 64 * 65];
            }
            f32x4 v = f32x4{y[0], y[1], y[2], y[3]} + bias;
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            float* dst = plane0 + kg * HW + (int64_t)gy * W + gx;
            if (vec) {
                *reinterpret_cast<f32x4*>(dst) = v;      // W % 4 == 0: gx + 3 < W and 16-byte aligned
            } else {
                dst[0] = v.x;
                if (gx + 1 < W) dst[1] = v.y;
                if (gx + 2 < W) dst[2] = v.z;
                if (gx + 3 < W) dst[3] = v.w;
            }
        }
    } else {
        const int k4 = (tid & 15) * 4, kg = ks * 64 + k4;
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (P.bias) bias = *reinterpret_cast<const f32x4*>(P.bias + kg);
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int pix = it * 16 + (tid >> 4), oy = pix >> 4, ox = pix & 15;
            const int gy = y0 + oy, gx = x0 + ox;
            if (gy >= H || gx >= W) continue;
            const float* r = lds + ((ox & 1) * 64 + (oy >> 1) * 8 + (ox >> 1)) * 64 + k4;      // R[a][ox & 1][tile][k4] at + a * 8192
            const f32x4 ra = *reinterpret_cast<const f32x4*>(r + ((oy & 1) ? 8192 : 0));
            const f32x4 rb = *reinterpret_cast<const f32x4*>(r + ((oy & 1) ? 16384 : 8192));
            const f32x4 rc = *reinterpret_cast<const f32x4*>(r + ((oy & 1) ? 24576 : 16384));
            f32x4 v = ((oy & 1) ? (ra - rb) - rc : (ra + rb) + rc) + bias;
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            const int64_t e = (out_px + (int64_t)gy * W + gx) * P.out_stride + kg;
            if (P.thresh) {
                const uint64_t ctr = P.offset + (uint64_t)(e >> 2);
                const u32x4 r4 = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, STREAM_DROPOUT_CONV}, (uint32_t)P.seed,
                                               (uint32_t)(P.seed >> 32));
                v.x = r4.x >= P.thresh ? v.x * P.scale : 0.f;
                v.y = r4.y >= P.thresh ? v.y * P.scale : 0.f;
                v.z = r4.z >= P.thresh ? v.z * P.scale : 0.f;
                v.w = r4.w >= P.thresh ? v.w * P.scale : 0.f;
            }
            *reinterpret_cast<f32x4*>(P.out + e) = v;
        }
    }
}

}  // namespace pod

extern "C" int pod_wino_filter_transform(const float* weight, float* U, int32_t K, int32_t C, pod_stream_t stream) {
    if (!weight || !U || K < 1 || C < 8 || (C & 7) != 0) return POD_E_INVALID;
    const int32_t Kpad = (K + 63) / 64 * 64;
    const int64_t n = (int64_t)Kpad * C;
    hipLaunchKernelGGL(pod::k_wino_filter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, U, K, C, Kpad);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

extern "C" int pod_wino_conv3x3(const float* in, float* out, const float* U, const float* bias, const int32_t* blocks, int32_t n_blocks,
                                int32_t C, int32_t K, int32_t k_planes, int32_t relu, float p, uint64_t seed, uint64_t offset,
                                pod_stream_t stream) {
    if (!in || !out || in == out || !U || !blocks || n_blocks < 0 || C < 8 || (C & 7) != 0 || K < 64 || (K & 63) != 0 ||
        !(p >= 0.0f && p < 1.0f) || k_planes < 0 || k_planes > K || (k_planes > 0 && p != 0.0f))
        return POD_E_INVALID;
    const int32_t KS = K / 64;
    if (KS != 1 && KS != 2 && KS != 4 && KS != 8) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(U) |
          reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(blocks)) & 15u) != 0)
        return POD_E_INVALID;
    if (n_blocks == 0) return POD_OK;
    static std::once_flag once;
    static hipError_t attr = hipSuccess;
    std::call_once(once, [] {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(pod::k_wino_conv3x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   pod::WINO_LDS_BYTES);
    });
    if (attr != hipSuccess) return POD_E_LAUNCH;
    pod::WinoParams P;
    P.in = in; P.out = out; P.U = U; P.bias = bias; P.blocks = reinterpret_cast<const int4*>(blocks);
    P.n_blocks = n_blocks; P.C = C; P.K = K; P.KS = KS; P.in_stride = C; P.out_stride = K; P.relu = relu; P.k_planes = k_planes;
    P.thresh = (uint32_t)((double)p * 4294967296.0);
    P.scale = 1.0f / (1.0f - p);
    P.seed = seed; P.offset = offset;
    const int per8 = 8 / KS;                                        // tile blocks per group of 8 consecutive workgroups
    const int64_t grid = ((int64_t)n_blocks + per8 - 1) / per8 * 8;
    if (grid > 0x7FFFFFFFLL) return POD_E_INVALID;
    hipLaunchKernelGGL(pod::k_wino_conv3x3, dim3((unsigned)grid), dim3(256), pod::WINO_LDS_BYTES, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
