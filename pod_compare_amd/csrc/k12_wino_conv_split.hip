// The head's 3x3 convolutions (probabilistic_retinanet.py:403-484) once more -- same Winograd F(2,3) x F(4,3) formulation, same data
// path (patch as full 128-byte lines by LDS-DMA, filters from L2, the same prologue and epilogue) and the same results to fp32
// rounding as pod_wino_conv3x3 (k11_wino_conv.hip) -- with every fp32 product formed on the 16-BIT matrix cores.
// Round 5: both operands are scaled by a power of two and split into TWO f16 terms (x s = x0 + x1 to 2^-23 |x s|, pod_wino.h) and the
// three partial products that matter (x0 u1, x1 u0, x0 u0) are accumulated in fp32 by v_mfma_f32_32x32x16_f16: half the matrix
// instructions of rounds 3-4's 3-way bf16 split (six products), two thirds of its filter bytes and 4/7 of its split arithmetic -- and,
// measured on the matrix cores against fp64 (tools/f16_split_numerics.hip), a SMALLER error than both the bf16 x 6 form and the fp32
// MFMA: the fp32 accumulation chain, not the products, is where these kernels lose bits, and it is half as long.
#include "pod_wino.h"

namespace pod {

typedef uint32_t vu32x4 __attribute__((ext_vector_type(4)));
constexpr int WINO_US_BYTES = 24 * 2 * 2 * 64 * 16;      // pre-split filter terms of a 16-channel chunk: [24 positions][kb][term][h][j][8 f16]  96 KB
constexpr int WINO_U_TOP = 14, WINO_V_TOP = 9;           // scaled filter abs-max in [2^14, 2^15); activations: 2^9 <= s amax < 2^10, x gain of Bt4 (x) Bt6 < 32
constexpr int WINO_WAIT_VM24 = 0x4078;                    // lgkmcnt(0) vmcnt(24)

// Filter transform U = G4 g G6t as in k_wino_filter.  Two passes: the abs-max of U (-> the trailer word of Us, behind the terms), then
// every value times the power of two that puts that abs-max into [2^14, 2^15) split into two f16 terms (round to nearest even) and
// written in the order the kernel's lanes load them:
// Us[ks][chunk16][q = 6 a + p][kb][term][h][j][e] = term(s U_q[c = 16 chunk16 + 8 h + e][k = 64 ks + 32 kb + j]); channels >= K are zero.
__device__ __forceinline__ void wino_filter_values(const float* __restrict__ w, int k, int c, int K, int C, float (&u)[4][6]) {
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
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float x0 = t0[a][0], x1 = t0[a][1], x2 = t0[a][2];
        u[a][0] = 0.25f * x0;
        u[a][1] = (-1.0f / 6.0f) * (x0 + x1 + x2);
        u[a][2] = (-1.0f / 6.0f) * (x0 - x1 + x2);
        u[a][3] = (1.0f / 24.0f) * x0 + (1.0f / 12.0f) * x1 + (1.0f / 6.0f) * x2;
        u[a][4] = (1.0f / 24.0f) * x0 - (1.0f / 12.0f) * x1 + (1.0f / 6.0f) * x2;
        u[a][5] = x2;
    }
}
__global__ void __launch_bounds__(256) k_wino_filter_amax(const float* __restrict__ w, float* __restrict__ amax, int32_t K, int32_t C, int32_t Kpad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (t < (int64_t)Kpad * C) {
        float u[4][6];
        wino_filter_values(w, (int)(t / C), (int)(t % C), K, C, u);
#pragma unroll
        for (int i = 0; i < 24; ++i) m = fmaxf(m, fabsf(u[i / 6][i % 6]));
    }
    wino_publish_amax1(amax, m);
}
__global__ void __launch_bounds__(256) k_wino_filter_split(const float* __restrict__ w, uint16_t* __restrict__ Us, const float* __restrict__ amax,
                                                           int32_t K, int32_t C, int32_t Kpad) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)Kpad * C) return;
    const int k = (int)(t / C), c = (int)(t % C);
    float u[4][6];
    wino_filter_values(w, k, c, K, C, u);
    const float sc = wino_pow2_scale(*amax, WINO_U_TOP);
    const int nchunk = C / 16, ks = k >> 6, kb = (k >> 5) & 1, j32 = k & 31, ch = c >> 4, hh = (c >> 3) & 1, e = c & 7;
    uint16_t* dst = Us + (((int64_t)ks * nchunk + ch) * (int64_t)WINO_US_BYTES) / 2 + (hh * 32 + j32) * 8 + e;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            const float x = u[a][p] * sc;                                        // exact
            const _Float16 h0 = (_Float16)x;                                     // v_cvt_f16_f32: round to nearest even
            const _Float16 h1 = (_Float16)(x - (float)h0);
            uint16_t* d = dst + (((a * 6 + p) * 2 + kb) * 2) * 512;            // 512 f16 = 64 lanes x 8 per (position, kb, term)
            d[0] = __builtin_bit_cast(uint16_t, h0);
            d[512] = __builtin_bit_cast(uint16_t, h1);
        }
}

__global__ void __launch_bounds__(256, 1) k_wino_conv3x3_split(const WinoParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ks, tb;
    wino_schedule(P.KS, P.n_blocks, ks, tb);
    if (tb >= P.n_blocks) return;
    uint32_t need[2] = {~0u, ~0u};                                        // need bits of this thread's patch pixels (tid, tid + 256): all, unless sparse
    if (P.live) {                                                         // sparse launch: slot -> live entry {record, need bits} (uniform: scalar loads)
        if (tb >= P.live[0]) return;
        const int32_t* const ent = P.live + POD_SPARSE_LIVE_HEAD + (int64_t)POD_SPARSE_LIVE_STRIDE * tb;
        tb = ent[0];
        // a patch pixel the layer below did not have to compute for this image reads as 0.0 (k15_sparse_blocks.hip): nothing an earlier
        // image left in the buffer is ever read, so the launch -- its abs-max record included -- is a function of this image alone
        need[0] = (uint32_t)ent[1 + (tid >> 5)];
        need[1] = (uint32_t)ent[9 + (tid >> 5 < 3 ? tid >> 5 : 2)];
    }
    // block record: the images of a (level, launch) stand in a GRID on a virtual canvas, image i at grid cell (i / gcols, i % gcols),
    // top-left canvas pixel (row (H + 1), col (W + 1)): one zero row / column between neighbours is the convolution's padding for
    // both (reads outside an image return 0.0), and 16x16 blocks are cut from the canvas without regard to image boundaries -- the
    // partial blocks at the right and bottom edges are paid once per level instead of once per image.
    WINO_STAMP(0);
    WINO_STAMP_WALL(12);
    uint32_t slot_e[12];                                                  // this lane's 12 pixel slots of a stage fill (constant table: asked for first, so
#pragma unroll                                                            // that nothing queues behind the patch loads that follow)
    for (int i = 0; i < 12; ++i) slot_e[i] = g_wino_slots.v[96 * (tid >> 6) + 8 * i + ((tid & 63) >> 3)];
    int mini_pidx[3];                                                     // ... and the patch pixel of its 3 slots of a mini-stage fill (324: none)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int pp = (((tid >> 6) * 3 + r) * 64 + (tid & 63)) >> 1, py = pp / 21, pi = pp - py * 21, px = 4 * (pi % 5) + pi / 5;
        mini_pidx[r] = py < 18 && pi < 20 && px < 18 ? py * 18 + px : 324;
    }
    // the filter operands of the first position of chunk 0 do not depend on the block record either: asked for now
    const int nchunk_all = P.C >> 4;                                      // chunks of 16 input channels (one bf16 MFMA k-step)
    const int nchunk = P.c_split ? P.c_split : nchunk_all;                // ... of which this workgroup set (blockIdx.y) accumulates its own range
    const int chunk0 = (int)blockIdx.y * nchunk;
    const int i32 = lane & 31, h = lane >> 5;
    const int a = __builtin_amdgcn_readfirstlane(wave);
    // which convolution of a grouped launch this block belongs to: read off the block index alone (no memory in the way of the first
    // filter loads below)
    const int set = (tb >= P.sets.first[1] ? 1 : 0) + (tb >= P.sets.first[2] ? 1 : 0) + (tb >= P.sets.first[3] ? 1 : 0);
    const float* const set_U = P.sets.U[set];
    const float* const set_in = P.sets.in[set];
    float* const set_out = P.sets.out[set];
    const float* const set_bias = P.sets.bias[set];
    // operand scales of the f16 split (pod_wino.h): two scalar loads, asked for first and needed only when the first patch is split
    const float in_amax_slot = wino_load_amax(P.sets.in_amax[set]);
    const float u_amax = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(set_U) + (int64_t)P.KS * (P.C >> 4) * WINO_US_BYTES);
    float* const set_out_amax = P.sets.out_amax[set];
    const uint64_t set_offset = P.sets.offset[set];
    const int set_replicas = P.sets.replicas[set], set_k_planes = P.sets.k_planes[set];
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(set_U) + ((int64_t)ks * nchunk_all + chunk0) * WINO_US_BYTES), 0,
                                                          nchunk * WINO_US_BYTES, 0x00020000);
    const int u_off = (a * 6 * 4 * 64 + h * 32 + i32) * 16;              // + ((p*2 + kb)*2 + term) KB, + chunk * 96 KB
    constexpr int WINO_U_LEAD = 3;   // positions the filter loads run ahead of their MFMAs (a position's 6 MFMAs last 192 cycles, an L2 round trip under load ~3x that)
    vu32x4 uP[6][4];                                                       // the filter terms of position p live in uP[p % 3]: [kb][term]; loaded TWO positions ahead (a position's
                                                                          // 6 MFMAs last 192 cycles, an L2 round trip under load longer)
    auto filter_piece = [&](int q16, int p, vu32x4(&u)[4], int i) {       // i = kb*2 + term: one buffer_load_dwordx4 (8 f16) each
        u[i] = __builtin_bit_cast(vu32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, q16 * WINO_US_BYTES + (p * 4 + i) * 1024, 0));
    };
#pragma unroll
    for (int pp = 0; pp < WINO_U_LEAD; ++pp)
#pragma unroll
        for (int i = 0; i < 4; ++i) filter_piece(0, pp, uP[pp], i);
    const int4 desc = P.blocks[tb];
    const int64_t base_px = desc.x, out_px = desc.y;                      // first pixel of image 0 in `in` / `out`
    const int gcols = (desc.z >> 24) & 0xFF, H = (desc.z >> 12) & 0xFFF, W = desc.z & 0xFFF, n_img = (desc.w >> 24) & 0xFF;
    const int y0 = ((desc.w >> 12) & 0xFFF) * 16, x0 = (desc.w & 0xFFF) * 16, Wv = W + 1, Hv = H + 1, HWi = H * W;
    const float rWv = 1.0f / (float)Wv, rHv = 1.0f / (float)Hv;
    // canvas coordinate v >= 0 -> (grid index, coordinate inside the cell); canvas extents < 2^16: exact after the fix-up
    auto cell = [](int v, int step, float rstep, int& idx) {
        int n = (int)((float)v * rstep);
        n -= n * step > v ? 1 : 0;
        n += (n + 1) * step <= v ? 1 : 0;
        idx = n;
        return v - n * step;
    };

    // ---- operands.  Tiles are 2 rows x 4 columns of outputs (F(2,3) down the rows: 4 patch rows; F(4,3) along the columns: 6
    // patch columns), 24 Winograd positions per tile and (c, k) pair where the direct convolution has 72 multiply-adds.  A
    // 16x16-pixel block is 8 x 4 = 32 tiles = one MFMA block of rows.  Wavefront `a` owns ROW a of the 4 x 6 position grid
    // (positions 6a .. 6a+5) for the 32 tiles and all 64 output channels (two 32-channel blocks, kb): 6 x 2 = 12 MFMA blocks =
    // 192 accumulators.  Row a of Bt4 d is one sum or difference of two patch rows:
    //     a = 0: d0 - d2      a = 1: d1 + d2      a = 2: d2 - d1      a = 3: d1 - d3
    // = x0 + s x1 with wave-uniform row offsets and sign (6 columns), followed by the 6-point column transform Bt6; every
    // transformed value feeds two MFMAs (kb).
    //   * filter operands never touch LDS: a lane needs U_q[its 4 channels][its output channel] for its row's 6 positions and
    //     both channel blocks = 12 x 16 bytes per chunk, loaded straight from L2 (the filter slice of this XCD) one chunk ahead;
    //     the four waves together read each slab byte exactly once;
    //   * the raw 18x18-pixel patch goes global -> LDS by LDS-DMA (buffer_load ... lds), 16-byte slots [h][row][col parity][col/2];
    //     out-of-range buffer offsets return 0.0 -- that IS the zero padding of the convolution; pad slots load nothing.
    const int row0 = a == 0 ? 0 : a == 2 ? 2 : 1, row1 = a == 2 ? 1 : a == 3 ? 3 : 2;
    const float sgn = a == 1 ? 1.0f : -1.0f;
    // Patch in LDS, one stage per SUPER-CHUNK of 32 input channels = the 128-byte line a pixel owns in the channels-last source:
    // [pixel slot][8 parts of 16 B], so that 8 consecutive lanes of an LDS-DMA instruction fetch ONE full line (measured,
    // tools/mfma_fillers.hip: a pixel per lane -- 64 lines per instruction, each line fetched again by the next three 8-channel
    // chunks -- stalls the in-order instruction streams by ~400 cycles per chunk once the lines come from HBM; full lines cost 55).
    // Pixel slot of patch pixel (py, px): 2 (rank(py) 18 + px) + ((py >> 2) & 1), rank = (py & 3) + 4 (py >> 3) (rows 0-3, 8-11, 16, 17
    // on the even slots, rows 4-7, 12-15 on the odd ones); part P of that pixel sits at sub-slot (P + rot) & 7,
    // rot = ((px >> 2) & 3) + 4 ((py >> 1) & 1): the 16 lanes a ds_read_b128 serves per LDS cycle (4 tile rows x 4 tile columns,
    // one part) then hit 16 different 16-byte bank groups -- conflict-free for every (row, column, chunk).
    const int ty = i32 >> 2, tx = i32 & 3;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    uint32_t areg[2][2][2][2];                                            // LDS byte address in stage 0: [row0 / row1][columns 0-3 / 4-5][16-channel half of the super-chunk][4-channel half of the lane's 8]
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
        const int py = 2 * ty + (rs ? row1 : row0);
        const int p0 = 2 * (((py & 3) + 4 * (py >> 3)) * 18 + 4 * tx) + ((py >> 2) & 1);   // slot of column 0 of the tile; column c: + 2 c
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            const int rot = ((tx + cl) & 3) + 4 * ((py >> 1) & 1);
#pragma unroll
            for (int c16 = 0; c16 < 2; ++c16)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) areg[rs][cl][c16][hf] = lds_base + p0 * 128 + ((4 * c16 + 2 * h + hf + rot) & 7) * 16;
        }
    }
    uint32_t amini[2];                                                    // LDS byte address in mini stage h (the lane's 8 channels of chunk 0): [row0 / row1]; + 16: second half
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) amini[rs] = lds_base + 2 * WINO_SB_FLOATS * 4 + h * 12288 + ((2 * ty + (rs ? row1 : row0)) * 21 + tx) * 32;
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(set_in + base_px * P.in_stride + chunk0 * 16), 0,
                                                          n_img * HWi * P.in_stride * 4 - chunk0 * 64, 0x00020000);
    // Where a patch pixel lives in the source: thread t works out pixel t (and t + 256) of the 18 x 18 patch ONCE -- canvas row ->
    // (grid row, row inside the image), canvas column -> (grid column, column) -- and parks its pixel index (-1: outside every image:
    // the loads then use a buffer offset that reads 0.0) in LDS; the lanes look their pieces up there: two divisions per thread
    // instead of two per lane and piece.
    int* pix_tab = reinterpret_cast<int*>(lds + 2 * WINO_SB_FLOATS + 2 * 3072);       // 324 ints behind the mini stages
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int t = tid + 256 * it;
        if (t >= 325) break;
        const int py = t / 18, px = t - py * 18, vy = y0 - 1 + py, vx = x0 - 1 + px;
        int m, n;
        const int gy = cell(vy < 0 ? 0 : vy, Hv, rHv, m), gx = cell(vx < 0 ? 0 : vx, Wv, rWv, n), img = m * gcols + n;
        const bool ok = (t < 324) & (vy >= 0) & (gy < H) & (vx >= 0) & (gx < W) & (n < gcols) & (img < n_img) & (((need[it] >> (t & 31)) & 1u) != 0);
        pix_tab[t] = ok ? img * HWi + gy * W + gx : -1;              // entry 324 = -1: the "no pixel" slots of the fills point here
    }
    __syncthreads();
    auto byte_offset = [&](int pix, int part4) {
        return pix >= 0 ? (pix * P.in_stride + part4) * 4 : 0x7FFFFF00;
    };
    // The first two chunks come from two MINI stages (8 channels each, 324 pixels x 32 B, 3 LDS-DMA instructions per wave each), so the
    // matrix cores start after 20 KB have landed instead of a 48 KB super-chunk; super-chunk 0 lands behind the first chunk's MFMAs.
    // Mini layout: 16-byte slot 2 (py 21 + (px & 3) 5 + (px >> 2)) + h: the 16 lanes of a ds_read_b128 group hit every bank group twice.
    WINO_STAMP(8);                                     // (the block record has arrived, the pixel table stands)
    int dmini[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) dmini[r] = pix_tab[mini_pidx[r]];
#pragma unroll
    for (int r = 0; r < 3; ++r) dmini[r] = byte_offset(dmini[r], 4 * (lane & 1));
    // LDS-DMA of a stage: 48 instructions of 8 pixel slots x 8 parts (the last 3 fetch nothing), wave a issues 12 a .. 12 a + 11.  Lane
    // (l3 = lane >> 3, q = lane & 7) of instruction I fills sub-slot q of pixel slot 8 I + l3 with part (q - rot) & 7 of its pixel.
    int doff[12];
    auto main_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 12; ++i) doff[i] = pix_tab[slot_e[i] & 0xFFFF];                  // 12 independent LDS reads, one round trip
#pragma unroll
        for (int i = 0; i < 12; ++i) doff[i] = byte_offset(doff[i], 4 * (((lane & 7) - (int)(slot_e[i] >> 16)) & 7));
    };
    typedef __attribute__((address_space(3))) void lds_void;
    auto mini_piece = [&](int which, int r) {                // 1 KB of the 8-channel patch of chunk `which` (0 / 1) into its mini stage
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(lds + 2 * WINO_SB_FLOATS + which * 3072 + (a * 3 + r) * 256), 16, dmini[r], which * 32, 0, 0);
    };
    auto patch_piece = [&](float* stage, int sc, int i) {    // 1 KB (8 pixels x 32 channels) of super-chunk sc, straight into LDS
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(stage + (a * 12 + i) * 256), 16, doff[i], sc * 128, 0, 0);
    };

    // Between the 32-cycle bf16 MFMAs an LDS-DMA piece costs its ~100 issue cycles in full (behind the 64-cycle fp32 MFMAs most of it
    // hides), so the K loop fills the stages through registers: every chunk loads 6 pieces (one buffer_load_dwordx4 each) and parks the
    // 6 it loaded a chunk earlier (ds_write_b128: free).
    f32x4 stg[6];
    const uint32_t stg_addr = lds_base + a * 12288 + lane * 16;          // + stage * 48 KB + piece * 1 KB
    auto stage_load = [&](int k, int sc, int i) {
        stg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, doff[i], sc * 128, 0));
    };
    auto stage_write = [&](int par, int k, int i) {
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((uintptr_t)(stg_addr + par * (WINO_SB_FLOATS * 4) + i * 1024)) = stg[k];
    };

    f32x16 acc[12];                                                      // [p][kb]; never cleared: chunk 0's first product multiplies into a zero C
    f32x4 x[12];                                                         // raw patch, one 4-channel half at a time: x[row][c]
    vu32x4 Vb[6][2];                                                      // the transformed patch as f16 operands: [position][term], 8 channels (regs 0-1: channels 0-3, 2-3: 4-7)
    float tN[6][4], vN[2][6][4];                                         // the NEXT chunk's transform in flight: row-combined columns; transformed values [half][position] (fp32, split later)
    if (POD_WINO_ELIM) {                                                  // (elimination builds: operands that were never loaded still need values)
#pragma unroll
    for (int i = 0; i < 12; ++i) Vb[i / 2][i % 2] = vu32x4{0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + lane};
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 48; ++i) vN[i / 24][(i / 4) % 6][i % 4] = x[i / 4][i % 4];
#pragma unroll
    for (int i = 0; i < 24; ++i) tN[i / 4][i % 4] = x[i / 4][i % 4];
    }
    // (reads as asm with hand-counted completion: see k11_wino_conv.hip)
#define WINO_READ(par, c16, hf, i)                                                                                                  \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(areg[(i) / 6][((i) % 6) >> 2][c16][hf]), "i"((par) * WINO_SB_FLOATS * 4 + ((i) % 6) * 256))
#define WINO_READ_MINI(hf, i)                                                                                                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(amini[(i) / 6]), "i"((hf) * 16 + ((((i) % 6) & 3) * 5 + (((i) % 6) >> 2)) * 32))
#define WINO_READ12(M, ...)                                                                                                          \
    M(__VA_ARGS__, 0); M(__VA_ARGS__, 1); M(__VA_ARGS__, 2); M(__VA_ARGS__, 3); M(__VA_ARGS__, 4); M(__VA_ARGS__, 5);                \
    M(__VA_ARGS__, 6); M(__VA_ARGS__, 7); M(__VA_ARGS__, 8); M(__VA_ARGS__, 9); M(__VA_ARGS__, 10); M(__VA_ARGS__, 11)
    // the reads have landed: the wait is tied to the twelve values, so that nothing computed from them can be scheduled above it
#define WINO_READS_LANDED()                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), \
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]))
    // Row a of V = Bt4 d Bt6^T for the lane's tile and 4 channels -- the SAME operations in the same order as the fp32 kernel's
    // packed transform (fused multiply-adds where it has them), but as scalar instructions (this file is compiled with
    // -fno-slp-vectorize): beside bf16 MFMAs a packed fp32 instruction costs ~40 cycles (tools/mfma_bf16_split.hip), an ordinary one
    // nothing --, then every value split into three bf16 terms by round-to-nearest (v = v0 + v1 + v2 to 2^-26 |v|; truncation would
    // make the dropped partial products one-signed: a bias the Winograd cancellation amplifies -- measured) and packed pairwise.
    //
    // The pieces of that work, so that they can be slotted behind MFMAs one small unit at a time (`unit` below) or run back to back
    // (`make_v`: the first two chunks).  The units keep their slots through the sched_barrier behind every MFMA (round 4 also pinned their
    // inputs and results with empty volatile asm: measured in round 5 to change nothing but +65 s_nop per chunk, deleted in round 6).
    auto rows_combine = [&](int c0, int c1) __attribute__((always_inline)) {              // tN[c] = x0[c] + s x1[c]
#pragma unroll
        for (int c = c0; c < c1; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tN[c][e] = __builtin_fmaf(sgn, x[6 + c][e], x[c][e]);
        }
    };
    float w0s[4], w1s[4], evs[4], ods[4], fs[4], gs[4];                                   // column transform, first level (per channel e)
    auto columns_level1 = [&](int e) __attribute__((always_inline)) {
        w0s[e] = __builtin_fmaf(-5.0f, tN[2][e], tN[4][e]);
        w1s[e] = __builtin_fmaf(-5.0f, tN[3][e], tN[5][e]);
        evs[e] = __builtin_fmaf(-4.0f, tN[2][e], tN[4][e]);
        ods[e] = __builtin_fmaf(-4.0f, tN[1][e], tN[3][e]);
        fs[e] = tN[4][e] - tN[2][e];
        gs[e] = tN[3][e] - tN[1][e];
    };
    auto columns_level2 = [&](int hf, int e) __attribute__((always_inline)) {
        vN[hf][0][e] = __builtin_fmaf(4.0f, tN[0][e], w0s[e]);
        vN[hf][5][e] = __builtin_fmaf(4.0f, tN[1][e], w1s[e]);
        vN[hf][1][e] = evs[e] + ods[e];
        vN[hf][2][e] = evs[e] - ods[e];
        vN[hf][3][e] = __builtin_fmaf(2.0f, gs[e], fs[e]);
        vN[hf][4][e] = __builtin_fmaf(-2.0f, gs[e], fs[e]);
    };
    // The two f16 terms of position p's 8 values (4 channel pairs: pair i = half i >> 1, channels 2 (i & 1) ..), in three steps whose
    // operations are independent of each other inside a step (a pair's own chain is convert -> residual -> convert):
    // w = nearest-even f16 pair of (v s) (v_fma_mixlo/hi_f16), residual r = v s - w exactly (v_fma_mix_f32, in place: vN is dead
    // afterwards), second term = nearest-even f16 pair of r (v_cvt_pk_f16_f32).
    const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wino_pow2_scale(wino_reduce_amax(in_amax_slot), WINO_V_TOP))));
    auto split_convert = [&](int p, int term) __attribute__((always_inline)) {
        vu32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                                      // i = 2 hf + pair: regs 0-1 channels 0-3, 2-3 channels 4-7
            w[i] = term == 0 ? wino_f16_pair_scaled(vN[i >> 1][p][2 * (i & 1)], vN[i >> 1][p][2 * (i & 1) + 1], sv)
                             : wino_f16_pair(vN[i >> 1][p][2 * (i & 1)], vN[i >> 1][p][2 * (i & 1) + 1]);
        }
        Vb[p][term] = w;
    };
    auto split_residual = [&](int p, int i0, int i1) __attribute__((always_inline)) {      // v <- v s - the first term, exactly
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            float& lo = vN[i >> 1][p][2 * (i & 1)];
            float& hi = vN[i >> 1][p][2 * (i & 1) + 1];
            wino_f16_residual_scaled(Vb[p][0][i], lo, hi, sv);
        }
    };
    auto split_position = [&](int p) __attribute__((always_inline)) {                      // all three steps back to back
        split_convert(p, 0); split_residual(p, 0, 4); split_convert(p, 1);
    };
    auto make_v = [&](int hf) __attribute__((always_inline)) {                             // serial form: x (12 reads of half hf) -> vN[hf]
        if (POD_WINO_ELIM & 8) return;
        rows_combine(0, 6);
#pragma unroll
        for (int e = 0; e < 4; ++e) columns_level1(e);
#pragma unroll
        for (int e = 0; e < 4; ++e) columns_level2(hf, e);
    };
    auto split_all = [&]() __attribute__((always_inline)) {
        if (POD_WINO_ELIM & 8) return;
#pragma unroll
        for (int p = 0; p < 6; ++p) split_position(p);
    };

    // One chunk = 16 input channels = one k-step of v_mfma_f32_32x32x16_f16.  Every fp32 product x * u is formed from the two f16 terms
    // of each (scaled) operand: the 3 partial products that matter (x1 u0, x0 u1, x0 u0: small ones first), fp32 accumulate.  36 MFMAs
    // per chunk: positions p = 0..5 of the wave's row, two channel blocks, 3 products; the filter terms of position p + 2 (4 x 16 B per
    // lane, pre-split, from L2) are loaded behind the first four MFMAs of position p.
    //
    // SOFTWARE PIPELINE (round 4, re-slotted for 36 MFMAs in round 5).  The VALU instructions that turn the next chunk's raw patch into f16
    // operands sit BEHIND the MFMAs of the running chunk, one unit of <= 6 independent instructions at a time:
    //     R0      12 LDS reads of the next chunk's first 4-channel half
    //     A       the two terms of the running chunk's own position 5 (its values were computed during the previous chunk; Vb[5] is read
    //             last, and could not be overwritten while the previous chunk's position 5 was still to come)
    //     T0      the row combination of the first half                                             (x -> tN)
    //     R1 V0   the second half's reads; the first half's column transform                       (tN -> vN[0])
    //     T1 V1   the same for the second half                                                      (-> vN[1])
    //     C0..C4  the two terms of the next chunk's positions 0..4 -- each after the running chunk's MFMAs of that position have issued
    // which needs the next chunk's patch to stand in a PUBLISHED stage when the chunk begins: super-chunk s + 1 is therefore complete one
    // chunk earlier than before -- chunk (s - 1, 1) parks its pieces 0..5, chunk (s, 0) pieces 6..11 -- in the stage whose last read (for
    // chunk (s - 1, 1), issued during (s - 1, 0)) lies a barrier behind.
    constexpr int N_UNITS = 58, N_SLOTS = 36;
    auto unit = [&](auto U, auto npar_t, auto n16_t, auto hasA_t) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value, npar = decltype(npar_t)::value, n16 = decltype(n16_t)::value;
        if constexpr (POD_WINO_ELIM & 8) { if constexpr (u >= 3 && !(u >= 13 && u < 16)) return; }
        if constexpr (u < 3) {                                                             // R0
            if constexpr (!(POD_WINO_ELIM & 1)) { WINO_READ(npar, n16, 0, 4 * u); WINO_READ(npar, n16, 0, 4 * u + 1); WINO_READ(npar, n16, 0, 4 * u + 2); WINO_READ(npar, n16, 0, 4 * u + 3); }
        } else if constexpr (u < 7) {                                                      // A (4 units)
            if constexpr (decltype(hasA_t)::value) {
                constexpr int k = u - 3;
                if constexpr (k == 0) split_convert(5, 0);
                else if constexpr (k == 1) split_residual(5, 0, 2);
                else if constexpr (k == 2) split_residual(5, 2, 4);
                else split_convert(5, 1);
            }
        } else if constexpr (u < 13) {                                                     // T0 (6 units: one column each)
            if constexpr (u == 7) WINO_READS_LANDED();
            rows_combine(u - 7, u - 6);
        } else if constexpr (u < 16) {                                                     // R1
            if constexpr (!(POD_WINO_ELIM & 1)) { WINO_READ(npar, n16, 1, 4 * (u - 13)); WINO_READ(npar, n16, 1, 4 * (u - 13) + 1); WINO_READ(npar, n16, 1, 4 * (u - 13) + 2); WINO_READ(npar, n16, 1, 4 * (u - 13) + 3); }
        } else if constexpr (u < 24) {                                                     // V0 (8 units: level 1 and level 2 of each channel, 6 ops each)
            constexpr int k = u - 16;
            if constexpr (k < 4) columns_level1(k);
            else columns_level2(0, k - 4);
        } else if constexpr (u < 30) {                                                     // T1
            if constexpr (u == 24) WINO_READS_LANDED();
            rows_combine(u - 24, u - 23);
        } else if constexpr (u < 38) {                                                     // V1
            constexpr int k = u - 30;
            if constexpr (k < 4) columns_level1(k);
            else columns_level2(1, k - 4);
        } else {                                                                           // C0..C4
            constexpr int p = (u - 38) / 4, k = (u - 38) % 4;
            if constexpr (k == 0) split_convert(p, 0);
            else if constexpr (k == 1) split_residual(p, 0, 2);
            else if constexpr (k == 2) split_residual(p, 2, 4);
            else split_convert(p, 1);
        }
    };
    // units of slot j: [j * 58 / 36, (j + 1) * 58 / 36) -- C(p) starts at unit 38 + 4 p = slot >= 23 + 2 p >= 6 (p + 1): behind position p's MFMAs
    const int last = nchunk - 1, last_s = last >> 1;
    const int sc1 = last_s < 1 ? last_s : 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) mini_piece(0, r);
#pragma unroll
    for (int r = 0; r < 3; ++r) mini_piece(1, r);
    WINO_STAMP(9);
    main_offsets();                                    // (behind the first loads: their latency hides it)
    WINO_STAMP(10);
#pragma unroll
    for (int i = 0; i < 12; ++i) patch_piece(lds, 0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) patch_piece(lds + WINO_SB_FLOATS, sc1, i);       // pieces 0..5 of super-chunk 1 straight into stage 1 ...
#pragma unroll
    for (int i = 0; i < 6; ++i) stage_load(i, sc1, 6 + i);                       // ... its pieces 6..11 through registers (chunk 0 parks them)
    WINO_STAMP(11);
    __builtin_amdgcn_s_waitcnt(WINO_WAIT_VM24);        // the mini stages and the first filter terms have landed; the 18 DMA pieces and the 6 register pieces fly on
    __builtin_amdgcn_s_barrier();
    WINO_STAMP(1);
    WINO_READ12(WINO_READ_MINI, 0);
    WINO_READS_LANDED();
    __builtin_amdgcn_sched_barrier(0);
    make_v(0);
    __builtin_amdgcn_sched_barrier(0);
    WINO_READ12(WINO_READ_MINI, 1);
    WINO_READS_LANDED();
    __builtin_amdgcn_sched_barrier(0);
    make_v(1);
    split_all();
    __builtin_amdgcn_sched_barrier(0);
    WINO_STAMP(2);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // chunk q = (super-chunk q >> 1, half c16 = q & 1), stage parity par = (q >> 1) & 1.
    //   mode 0: the first chunk (accumulators start from zero; nothing of the next chunk behind its MFMAs: stage 0 is still landing)
    //   mode 1: chunk 1 (operands made back to back after chunk 0; runs units R0 .. C4 for chunk 2)
    //   mode 2: steady state (units A .. C4)
    auto chunk = [&](auto mode_t, auto c16_t, auto par_t, int q) {
        constexpr int mode = decltype(mode_t)::value, c16 = decltype(c16_t)::value, par = decltype(par_t)::value;
        constexpr int n16 = c16 ^ 1, npar = c16 == 1 ? par ^ 1 : par;             // the NEXT chunk's half and stage
        const int qn = q + 1 <= last ? q + 1 : last;
        // stage traffic: chunk (s, 0) parks pieces 6..11 of super-chunk s + 1 (other stage) and loads 0..5 of s + 2; chunk (s, 1) parks those in
        // its OWN stage (nobody reads it any more: the next chunk's operands come from the other one) and loads 6..11 of s + 2
        const int ls0 = (q >> 1) + 2, ls = ls0 < last_s ? ls0 : last_s;
        wino_static_for([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value, p = j / 6, m = j % 6, kb = m & 1, prod = m >> 1;
            constexpr int sa = prod == 1 ? 1 : 0;      // filter term of the product
            constexpr int sb = prod == 0 ? 1 : 0;      // patch term:  x1 u0, x0 u1, x0 u0
            if constexpr (mode == 0 && prod == 0)
                acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, uP[p][kb * 2 + sa]), __builtin_bit_cast(wino_f16x8, Vb[p][sb]), zero16, 0, 0, 0);
            else
                acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, uP[p][kb * 2 + sa]), __builtin_bit_cast(wino_f16x8, Vb[p][sb]), acc[p * 2 + kb], 0, 0, 0);
            if constexpr (m < 4) { if (!(POD_WINO_ELIM & 2)) filter_piece(p + WINO_U_LEAD >= 6 ? qn : q, (p + WINO_U_LEAD) % 6, uP[(p + WINO_U_LEAD) % 6], m); }
            else if constexpr (m == 4) { if (!(POD_WINO_ELIM & 4)) stage_write(c16 == 0 ? par ^ 1 : par, p, c16 == 0 ? 6 + p : p); }
            if constexpr (p == 5 && m >= 4) {                    // the 6 stage loads (HBM) sit BEHIND the chunk's last filter loads: loads return in
                if (!(POD_WINO_ELIM & 4)) {                      // order, and every filter term ahead is needed soon
                    stage_load(3 * (m - 4), ls, 6 * c16 + 3 * (m - 4));
                    stage_load(3 * (m - 4) + 1, ls, 6 * c16 + 3 * (m - 4) + 1);
                    stage_load(3 * (m - 4) + 2, ls, 6 * c16 + 3 * (m - 4) + 2);
                }
            }
            if constexpr (mode != 0) {
                constexpr int u0 = j * N_UNITS / N_SLOTS, u1 = (j + 1) * N_UNITS / N_SLOTS;
                wino_static_for([&](auto K) __attribute__((always_inline)) {
                    unit(std::integral_constant<int, u0 + decltype(K)::value>{}, std::integral_constant<int, npar>{}, std::integral_constant<int, n16>{},
                         std::integral_constant<bool, mode == 2>{});
                }, std::make_integer_sequence<int, u1 - u0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, N_SLOTS>{});
        // No vmcnt wait: the pieces this chunk parked were loaded a chunk ago (hipcc waits for them where they are stored), the DMA pieces of
        // the prologue were issued before filter terms this chunk's MFMAs have consumed, and loads return in order.  The barrier publishes the
        // stage and retires this chunk's LDS reads.
        __builtin_amdgcn_s_waitcnt(WINO_WAIT_LGKM0);
        __builtin_amdgcn_s_barrier();
        if constexpr (mode == 0) {
            if (q >= last) return;
            // chunk 1's operands, back to back (stage 0 has landed and was published just now)
            if constexpr (!(POD_WINO_ELIM & 1)) WINO_READ12(WINO_READ, npar, n16, 0);
            WINO_READS_LANDED();
            __builtin_amdgcn_sched_barrier(0);
            make_v(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(POD_WINO_ELIM & 1)) WINO_READ12(WINO_READ, npar, n16, 1);
            WINO_READS_LANDED();
            __builtin_amdgcn_sched_barrier(0);
            make_v(1);
            split_all();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();               // chunk 1 parks pieces in stage 0: every wave must have read its operands out of it
        }
    };
    using std::integral_constant;
    chunk(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, 0);
    if (nchunk > 1) chunk(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, 1);
    for (int base = 0;; base += 4) {
#define WINO_CHUNK(t)                                                                                                              \
    if (base + (t) >= nchunk) break;                                                                                               \
    chunk(integral_constant<int, 2>{}, integral_constant<int, (t) & 1>{}, integral_constant<int, ((t) >> 1) & 1>{}, base + (t));
        WINO_CHUNK(2) WINO_CHUNK(3) WINO_CHUNK(4) WINO_CHUNK(5)
#undef WINO_CHUNK
    }
    __syncthreads();                                   // every wave is done reading the stages, no DMA in flight: they become the output staging
    WINO_STAMP(3);

    // ---- output transform Y = At2 M At4^T, At2 = [[1,1,1,0],[0,1,-1,-1]], At4 = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]].
    // Every wave applies At4 to its row of 6 positions in registers (4 output columns) and parks Z[a][tile][column][channel] in LDS
    // (130 KB); the store pass combines the four rows in a fixed order:  Y[0][x] = (Z[0][x] + Z[1][x]) + Z[2][x],
    // Y[1][x] = (Z[1][x] - Z[2][x]) - Z[3][x]
    if (POD_WINO_ELIM & 128) {
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(acc[i]));
        return;
    }
    // The MFMAs run with the FILTER as the row operand: a lane's accumulator register reg of block (p, kb) is channel
    // 32 kb + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of tile lane & 31 -- four consecutive channels per register quad, so the
    // transform runs on packed pairs and a 16-byte store parks 4 channels.  Staging: Z[a][tile][column e][64 channels], a tile's 4 x 64
    // floats + 4 pad (1040 B: the 8 tiles of a store's lane group hit 8 different 16-byte bank groups), 4 x 32 x 1040 B = 133 120 B.
    constexpr int TS = 260;                    // floats per (a, tile)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 m[6];
#pragma unroll
            for (int p6 = 0; p6 < 6; ++p6) m[p6] = f32x4{acc[p6 * 2 + kb][4 * g], acc[p6 * 2 + kb][4 * g + 1], acc[p6 * 2 + kb][4 * g + 2], acc[p6 * 2 + kb][4 * g + 3]};
            const f32x4 s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
            float* o = lds + (a * 32 + i32) * TS + kb * 32 + 8 * g + 4 * h;
            *reinterpret_cast<f32x4*>(o) = (m[0] + s1) + s2;
            *reinterpret_cast<f32x4*>(o + 64) = __builtin_elementwise_fma(f32x4{2.f, 2.f, 2.f, 2.f}, d2, d1);
            *reinterpret_cast<f32x4*>(o + 128) = __builtin_elementwise_fma(f32x4{4.f, 4.f, 4.f, 4.f}, s2, s1);
            *reinterpret_cast<f32x4*>(o + 192) = __builtin_elementwise_fma(f32x4{8.f, 8.f, 8.f, 8.f}, d2, d1) + m[5];
        }
    __syncthreads();
    WINO_STAMP(4);
    if (POD_WINO_ELIM & 32) return;
    constexpr int ZA = 32 * TS;                // floats per position row a
    float* const out_base = set_out + (int64_t)blockIdx.y * P.split_out_stride;
    // the accumulators hold (s_u U) (s_v V) sums: the two powers of two come off again in the store pass -- exactly, inside the
    // fused multiply-add that adds the bias
    const float inv1 = wino_pow2_inverse(sv) * wino_pow2_inverse(wino_pow2_scale(u_amax, WINO_U_TOP));
    const f32x4 inv = f32x4{inv1, inv1, inv1, inv1};
    float lmax = 0.0f;                         // abs-max of what this thread stores (-> out_amax: the next convolution's operand scale)
    if (set_k_planes > 0) {
        // NCHW planes: thread -> (channel, row of the block, 4 pixels along x = one tile's columns); 64-byte runs per (channel, row)
        const int oy = (tid >> 2) & 15, ox = (tid & 3) * 4;
        int m;
        const int gy = cell(y0 + oy, Hv, rHv, m);
        int64_t px0[4];                                   // output pixel (of plane 0) per column, -1: not a pixel of any image
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int n;
            const int gx = cell(x0 + ox + e, Wv, rWv, n), img = m * gcols + n;
            px0[e] = (n < gcols && img < n_img && gx < W && gy < H) ? (out_px + (int64_t)img * HWi) * set_k_planes + (int64_t)gy * W + gx : -1;
        }
        const bool vec = px0[0] >= 0 && px0[3] == px0[0] + 3 && (px0[0] & 3) == 0 && (HWi & 3) == 0;
        const int tile = (oy >> 1) * 4 + (tid & 3);
#pragma unroll 2
        for (int it = 0; it < 16; ++it) {
            const int k = it * 4 + (tid >> 6), kg = ks * 64 + k;
            if (kg >= set_k_planes) continue;
            const float bias = set_bias ? set_bias[kg] : 0.0f;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* r = lds + tile * TS + e * 64 + k;                 // Z[a][tile][e][k] at + a * ZA
                y[e] = (oy & 1) == 0 ? (r[0] + r[ZA]) + r[2 * ZA] : (r[ZA] - r[2 * ZA]) - r[3 * ZA];
            }
            f32x4 v = __builtin_elementwise_fma(f32x4{y[0], y[1], y[2], y[3]}, inv, f32x4{bias, bias, bias, bias});
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (set_out_amax)
                lmax = fmaxf(fmaxf(lmax, px0[0] >= 0 ? fabsf(v.x) : 0.f), fmaxf(fmaxf(px0[1] >= 0 ? fabsf(v.y) : 0.f, px0[2] >= 0 ? fabsf(v.z) : 0.f), px0[3] >= 0 ? fabsf(v.w) : 0.f));
            float* plane = out_base + (int64_t)kg * HWi;
            if (vec) {
                *reinterpret_cast<f32x4*>(plane + px0[0]) = v;
            } else {
                if (px0[0] >= 0) plane[px0[0]] = v.x;
                if (px0[1] >= 0) plane[px0[1]] = v.y;
                if (px0[2] >= 0) plane[px0[2]] = v.z;
                if (px0[3] >= 0) plane[px0[3]] = v.w;
            }
        }
    } else {
        // thread -> 8 consecutive channels (one Philox call: 16 mask bits per element) of one pixel column, rows of one parity
        const int k8 = (tid & 7) * 8, kg = ks * 64 + k8, ox = (tid >> 3) & 15, odd = tid >> 7;
        f32x4 bias0 = f32x4{0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
        if (set_bias) {
            bias0 = *reinterpret_cast<const f32x4*>(set_bias + kg);
            bias1 = *reinterpret_cast<const f32x4*>(set_bias + kg + 4);
        }
        const uint64_t drop_key = P.thresh ? dropout_key(P.seed, P.epoch) : 0ull;
        int n;
        const int gx = cell(x0 + ox, Wv, rWv, n);
        const bool col_ok = n < gcols && gx < W;
        int m, gy = cell(y0 + odd, Hv, rHv, m) - 2;                                   // canvas row y0 + 2 it + odd: grid row m, image row gy (H: the separator)
        const float* rbase = lds + (ox >> 2) * TS + (ox & 3) * 64 + k8 + (odd ? ZA : 0);      // Z[a][tile][ox & 3][k8] of row a = odd
        // Rows in BATCHES of four: the 24 LDS reads of a batch are issued together (one round trip, not four behind four branches), then
        // the four rows' arithmetic, Philox calls and stores run as independent chains (round 5: 12.6 k -> cycles of the workgroup's 81 k)
#pragma unroll 1
        for (int g = 0; g < 2; ++g) {
            f32x4 z[4][6];
            int gyi[4], imgi[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                gy += 2;
                if (gy >= Hv) {
                    gy -= Hv;
                    ++m;
                }
                gyi[it] = gy;
                imgi[it] = m * gcols + n;
                const float* r = rbase + (4 * g + it) * 4 * TS;                          // tile (4 g + it, ox >> 2)
                z[it][0] = *reinterpret_cast<const f32x4*>(r); z[it][1] = *reinterpret_cast<const f32x4*>(r + 4);
                z[it][2] = *reinterpret_cast<const f32x4*>(r + ZA); z[it][3] = *reinterpret_cast<const f32x4*>(r + ZA + 4);
                z[it][4] = *reinterpret_cast<const f32x4*>(r + 2 * ZA); z[it][5] = *reinterpret_cast<const f32x4*>(r + 2 * ZA + 4);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int gy_ = gyi[it], img = imgi[it];
                if (!col_ok || gy_ >= H || img >= n_img) continue;
                f32x4 v0 = __builtin_elementwise_fma(odd ? (z[it][0] - z[it][2]) - z[it][4] : (z[it][0] + z[it][2]) + z[it][4], inv, bias0);
                f32x4 v1 = __builtin_elementwise_fma(odd ? (z[it][1] - z[it][3]) - z[it][5] : (z[it][1] + z[it][3]) + z[it][5], inv, bias1);
                if (P.relu) {
                    v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                    v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
                }
                int64_t e = (out_px + (int64_t)img * HWi + (int64_t)gy_ * W + gx) * P.out_stride + kg;      // a multiple of 8
                if (set_out_amax) {                                      // (a masked value is 0 or v * scale: v * scale bounds both, whatever the masks)
                    const f32x4 a0v = __builtin_elementwise_abs(v0), a1v = __builtin_elementwise_abs(v1);
                    const float mx = fmaxf(fmaxf(fmaxf(a0v.x, a0v.y), fmaxf(a0v.z, a0v.w)), fmaxf(fmaxf(a1v.x, a1v.y), fmaxf(a1v.z, a1v.w)));
                    lmax = fmaxf(lmax, P.thresh ? mx * P.scale : mx);
                }
                if (set_replicas > 0) {                                  // (0: an ordinary launch; 1: one "replica" under the replicas' mask)
                    // The first conv of an MC-dropout subnet: its output is the same for every run, so the store pass writes the runs'
                    // masked replicas itself (replica r = image r of the output canvas) -- the separate expand pass read this tensor back
                    // and wrote them in a launch of its own.  Mask of replica r = pod_expand_dropout's: counter word 2, 16 bits per element.
                    for (int rep = 0; rep < set_replicas; ++rep, e += (int64_t)HWi * P.out_stride) {
                        f32x4 w0 = v0, w1 = v1;
                        if (P.thresh) {
                            const uint64_t ctr = set_offset + (uint64_t)(e >> 3);
                            const u32x4 r4 = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 2u, STREAM_DROPOUT_CONV}, (uint32_t)drop_key,
                                                           (uint32_t)(drop_key >> 32));
                            w0.x = (r4.x & 0xFFFFu) >= P.thresh ? v0.x * P.scale : 0.f;
                            w0.y = (r4.x >> 16) >= P.thresh ? v0.y * P.scale : 0.f;
                            w0.z = (r4.y & 0xFFFFu) >= P.thresh ? v0.z * P.scale : 0.f;
                            w0.w = (r4.y >> 16) >= P.thresh ? v0.w * P.scale : 0.f;
                            w1.x = (r4.z & 0xFFFFu) >= P.thresh ? v1.x * P.scale : 0.f;
                            w1.y = (r4.z >> 16) >= P.thresh ? v1.y * P.scale : 0.f;
                            w1.z = (r4.w & 0xFFFFu) >= P.thresh ? v1.z * P.scale : 0.f;
                            w1.w = (r4.w >> 16) >= P.thresh ? v1.w * P.scale : 0.f;
                        }
                        *reinterpret_cast<f32x4*>(out_base + e) = w0;
                        *reinterpret_cast<f32x4*>(out_base + e + 4) = w1;
                    }
                    continue;
                }
                if (P.thresh && !(POD_WINO_ELIM & 64)) {
                    const uint64_t ctr = set_offset + (uint64_t)(e >> 3);
                    const u32x4 r4 = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, STREAM_DROPOUT_CONV}, (uint32_t)drop_key,
                                                   (uint32_t)(drop_key >> 32));
                    v0.x = (r4.x & 0xFFFFu) >= P.thresh ? v0.x * P.scale : 0.f;
                    v0.y = (r4.x >> 16) >= P.thresh ? v0.y * P.scale : 0.f;
                    v0.z = (r4.y & 0xFFFFu) >= P.thresh ? v0.z * P.scale : 0.f;
                    v0.w = (r4.y >> 16) >= P.thresh ? v0.w * P.scale : 0.f;
                    v1.x = (r4.z & 0xFFFFu) >= P.thresh ? v1.x * P.scale : 0.f;
                    v1.y = (r4.z >> 16) >= P.thresh ? v1.y * P.scale : 0.f;
                    v1.z = (r4.w & 0xFFFFu) >= P.thresh ? v1.z * P.scale : 0.f;
                    v1.w = (r4.w >> 16) >= P.thresh ? v1.w * P.scale : 0.f;
                }
                *reinterpret_cast<f32x4*>(out_base + e) = v0;
                *reinterpret_cast<f32x4*>(out_base + e + 4) = v1;
            }
        }
    }
    if (set_out_amax) wino_publish_amax_block(set_out_amax, lmax);          // (set: uniform over the workgroup)
#ifdef POD_TRACE
    __builtin_amdgcn_s_waitcnt(0);                      // the stores have left
    WINO_STAMP(5);
    WINO_STAMP_WALL(13);
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wino_trace[blockIdx.x * 16 + 6] = ((long long)xcc << 32) | hw;
    }
#endif
}

}  // namespace pod

#ifdef POD_TRACE
extern "C" int pod_wino_trace_dump_split(long long* host, int32_t n_workgroups) {   // diagnostics build only
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    if (n_workgroups > 8192) n_workgroups = 8192;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pod::g_wino_trace), (size_t)n_workgroups * 16 * sizeof(long long)) != hipSuccess) return POD_E_LAUNCH;
    return POD_OK;
}
#endif

extern "C" int64_t pod_wino_filter_split_bytes(int32_t K, int32_t C) {           // size of Us: the terms + the 16-byte trailer (abs-max word)
    if (K < 1 || C < 16 || (C & 15) != 0) return 0;
    const int64_t Kpad = (K + 63) / 64 * 64;
    return Kpad / 64 * (C / 16) * (int64_t)pod::WINO_US_BYTES + 16;
}

extern "C" int pod_wino_filter_transform_split(const float* weight, void* Us, int32_t K, int32_t C, pod_stream_t stream) {
    if (!weight || !Us || K < 1 || C < 16 || (C & 15) != 0 || (reinterpret_cast<uintptr_t>(Us) & 15u) != 0) return POD_E_INVALID;
    const int32_t Kpad = (K + 63) / 64 * 64;
    const int64_t n = (int64_t)Kpad * C;
    float* amax = reinterpret_cast<float*>(reinterpret_cast<char*>(Us) + pod_wino_filter_split_bytes(K, C) - 16);
    if (hipMemsetAsync(amax, 0, 16, (hipStream_t)stream) != hipSuccess) return POD_E_LAUNCH;
    hipLaunchKernelGGL(pod::k_wino_filter_amax, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight, amax, K, C, Kpad);
    POD_CHECK_LAUNCH();
    hipLaunchKernelGGL(pod::k_wino_filter_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, weight,
                       reinterpret_cast<uint16_t*>(Us), amax, K, C, Kpad);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

static int wino_split_prepare() {        // the kernel's dynamic LDS size, once per device
    static std::once_flag once[64];
    static hipError_t attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return POD_E_LAUNCH;
    std::call_once(once[dev], [dev] {
        attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(pod::k_wino_conv3x3_split), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        pod::WINO_LDS_BYTES);
    });
    return attr[dev] == hipSuccess ? POD_OK : POD_E_LAUNCH;
}

// ONE entry for every form of the launch (round 5; rounds 3-4 had a symbol per form): 1..4 convolutions of one shape in one grid, each
// with its own buffers, filter, bias, operand abs-max words, Philox offset, replica count and plane count; or (n_splits > 1) one
// convolution cut over its input channels into partial sums.  See include/pod_mi355x.h: PodWinoConv.
extern "C" int pod_wino_conv3x3_split(const PodWinoConv* d, pod_stream_t stream) {
    if (!d || d->n_sets < 1 || d->n_sets > 4 || !d->blocks || d->n_blocks < 0 || d->C < 16 || (d->C & 15) != 0 || d->K < 64 || (d->K & 63) != 0 ||
        !(d->p >= 0.0f && d->p < 1.0f) || d->sets[0].first_block != 0 || (reinterpret_cast<uintptr_t>(d->blocks) & 15u) != 0 ||
        (d->form != 0 && d->form != POD_WINO_FORM_4 && d->form != POD_WINO_FORM_8))
        return POD_E_INVALID;
    const int32_t C = d->C, K = d->K, KS = K / 64;
    if (KS != 1 && KS != 2 && KS != 4 && KS != 8) return POD_E_INVALID;
    const bool partial = d->n_splits > 1;
    if (partial) {
        // whole 32-channel super-chunks (full 128-byte lines) per split; partial sums carry no bias / ReLU / dropout / replicas / planes
        if (d->n_sets != 1 || d->n_splits > 16 || (C / 16) % d->n_splits != 0 || ((C / 16 / d->n_splits) & 1) != 0 || d->split_stride < 0 || (d->split_stride & 3) != 0 ||
            d->p != 0.0f || d->relu || d->sets[0].bias || d->sets[0].replicas || d->sets[0].k_planes || d->sets[0].out_amax)
            return POD_E_INVALID;
    }
    pod::WinoParams P;
    for (int s = 0; s < 4; ++s) {
        const PodConvSet& q = d->sets[s < d->n_sets ? s : 0];
        if (s < d->n_sets) {
            if (!q.in || !q.out || q.in == q.out || !q.Us || !q.in_amax || q.replicas < 0 || q.replicas > 127 || q.k_planes < 0 || q.k_planes > K ||
                (q.k_planes > 0 && (d->p != 0.0f || q.replicas != 0)) || (s > 0 && q.first_block < d->sets[s - 1].first_block) || q.first_block > d->n_blocks)
                return POD_E_INVALID;
            if (((reinterpret_cast<uintptr_t>(q.in) | reinterpret_cast<uintptr_t>(q.out) | reinterpret_cast<uintptr_t>(q.Us) | reinterpret_cast<uintptr_t>(q.bias)) & 15u) != 0 ||
                ((reinterpret_cast<uintptr_t>(q.in_amax) | reinterpret_cast<uintptr_t>(q.out_amax)) & 3u) != 0)
                return POD_E_INVALID;
        }
        P.sets.first[s] = s < d->n_sets ? q.first_block : INT32_MAX;
        P.sets.in[s] = q.in; P.sets.out[s] = q.out; P.sets.U[s] = reinterpret_cast<const float*>(q.Us); P.sets.bias[s] = q.bias;
        P.sets.in_amax[s] = q.in_amax; P.sets.out_amax[s] = q.out_amax;
        P.sets.offset[s] = q.offset; P.sets.replicas[s] = q.replicas; P.sets.k_planes[s] = q.k_planes;
    }
    if (d->n_blocks == 0) return POD_OK;
    if (wino_split_prepare() != POD_OK) return POD_E_LAUNCH;
    const PodConvSet& q0 = d->sets[0];
    P.in = q0.in; P.out = q0.out; P.U = reinterpret_cast<const float*>(q0.Us); P.bias = q0.bias; P.blocks = reinterpret_cast<const int4*>(d->blocks);
    P.n_blocks = d->n_blocks; P.C = C; P.K = K; P.KS = KS; P.in_stride = C; P.out_stride = K; P.relu = d->relu; P.k_planes = q0.k_planes;
    P.thresh = POD_DROPOUT_THRESH16(d->p);
    P.scale = 1.0f / (1.0f - d->p);
    P.seed = d->seed; P.offset = q0.offset;
    P.c_split = partial ? C / 16 / d->n_splits : 0; P.split_out_stride = partial ? d->split_stride : 0; P.epoch = d->epoch; P.replicas = q0.replicas;
    P.live = d->live_blocks;
    const int64_t grid = pod::wino_grid(KS, d->n_blocks);
    if (grid > 0x7FFFFFFFLL) return POD_E_INVALID;
    if (d->form == POD_WINO_FORM_8) {                   // round 6's experiment build only (tools/experiments/k16_wino_conv_split8.hip: measured 5 - 8 % slower)
#ifdef POD_WITH_K16
        return pod::wino_split8_launch(P, grid, partial ? (unsigned)d->n_splits : 1u, (hipStream_t)stream);
#else
        return POD_E_INVALID;
#endif
    }
    hipLaunchKernelGGL(pod::k_wino_conv3x3_split, dim3((unsigned)grid, partial ? (unsigned)d->n_splits : 1u), dim3(256), pod::WINO_LDS_BYTES, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
