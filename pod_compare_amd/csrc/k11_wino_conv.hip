// 3x3 / stride 1 / pad 1 fp32 convolution of the probabilistic RetinaNet head's subnets (probabilistic_retinanet.py:403-427:
// four conv3x3(256 -> 256) + ReLU + Dropout per subnet, evaluated for every MC run on every FPN level) and predictors
// (PR:430-484) as ONE launch per conv layer over all levels and all runs: fp32 Winograd on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), bias + ReLU + dropout fused into the store.
//
// Why Winograd: a direct fp32 convolution is bounded by the 157 TFLOP/s fp32 MFMA peak (MIOpen's implicit GEMM reaches
// 0.83 of it on the p3 maps and nothing can reach more than 1.0).  F(2,3) down the rows x F(4,3) along the columns needs
// 4 x 6 = 24 multiply-adds per 2x4 outputs and (c, k) pair instead of 72, so the same matrix cores deliver up to 3x the
// direct-convolution rate, in fp32 throughout.  Error vs a direct fp32 convolution: ~4e-6 of the output scale at C = 256
// (F(2x2,3x3), the first version: 2e-6 and 16 / 36 of the multiply-adds).
//
//   Y = At2 [ (G4 g G6^T) . (Bt4 d Bt6^T) ] At4^T     d: 4x6 input patch, g: 3x3 filter, Y: 2x4 outputs, "." summed over c
//
// Data layout (channels-last): activations are [pixel][C] fp32; every (level, run) image of a launch lives in the same
// buffer, a table of 16x16-pixel output blocks (int4 {first pixel of image 0 in `in`, in `out`, H << 16 | W,
// n_images << 24 | by << 12 | bx}) says where.  Filters are transformed once (pod_wino_filter_transform) into the order the
// kernel's lanes load them in.  The predictor convolutions (cls_score, bbox_pred, cls_var, bbox_cov: K = 63 / 36 / 90 real
// channels) write NCHW planes, the layout K1 streams, straight from the staging tile.
//
// Workgroup = 256 threads = 4 waves, one per SIMD: 32 tiles (8 x 4 tiles of 2x4 = 16x16 output pixels) x 64 output
// channels x the 24 Winograd positions.  Wave a owns ROW a of the 4x6 position grid for the 32 tiles and all 64 channels:
// 6 positions x 2 channel blocks of 32x32 = 12 MFMA blocks = 192 accumulator registers.  A row of Bt4 d is one sum or
// difference of two patch rows, then the 6-point column transform: 20 four-channel operations per chunk, and every
// transformed value feeds two MFMAs.  Per chunk of 8 input channels (48 MFMAs per wave) the raw 18x18-pixel input patch is
// staged in LDS by LDS-DMA (no transformed copy exists anywhere; two 12 KB stages) and the filter operands go from L2
// straight into registers (each wave needs only its row's positions: the four waves read each slab byte once).
// With one wave per SIMD every non-MFMA instruction costs issue time on top of the MFMA time (fp32 MFMA and the other pipes
// do not overlap within a wave: measured), so the design minimises them: 27 memory instructions and 40 packed VALU per 48
// MFMAs.  The output transform applies At4 to each wave's row in registers, parks the result in LDS (128 KB) and combines the
// four rows (At2) in the store pass.
//
// Launch: blockIdx & 7 is the XCD (round-robin dispatch); an XCD always works on the same TWO 64-channel filter slices (3 MB for
// C = 256), which therefore stay in that XCD's 4 MB L2, and takes a block with both slices back to back, so that the second
// workgroup's patch comes out of the L2 as well (wino_schedule, pod_wino.h).
#include "pod_wino.h"

namespace pod {

// Filter transform U = G4 g G6t (4 x 6 positions: F(2,3) down the rows, F(4,3) along the columns),
//   G4 = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],  G6 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],
// written in the order the main kernel's lanes load it:
// U[ks][chunk][q = 6 a + p][h][j][s] = U_q[c = 8 chunk + 4 h + s][k = 64 ks + j] (16 bytes per lane and position); channels >= K are zero.
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
    const int h = cc >> 2, sc = cc & 3;
    float* dst = U + ((int64_t)ks * nchunk + ch) * WINO_U_FLOATS + (h * 64 + j64) * 4 + sc;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float x0 = t0[a][0], x1 = t0[a][1], x2 = t0[a][2];
        float u[6];
        u[0] = 0.25f * x0;
        u[1] = (-1.0f / 6.0f) * (x0 + x1 + x2);
        u[2] = (-1.0f / 6.0f) * (x0 - x1 + x2);
        u[3] = (1.0f / 24.0f) * x0 + (1.0f / 12.0f) * x1 + (1.0f / 6.0f) * x2;
        u[4] = (1.0f / 24.0f) * x0 - (1.0f / 12.0f) * x1 + (1.0f / 6.0f) * x2;
        u[5] = x2;
#pragma unroll
        for (int p = 0; p < 6; ++p) dst[(a * 6 + p) * 512] = u[p];
    }
}

__global__ void __launch_bounds__(256, 1) k_wino_conv3x3(const WinoParams P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ks, tb;
    wino_schedule(P.KS, P.n_blocks, ks, tb);
    if (tb >= P.n_blocks) return;
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
    // the filter operands of chunk 0 do not depend on the block record either: straight from L2 into registers, asked for now
    const int nchunk = P.C >> 3;
    const int i32 = lane & 31, h = lane >> 5;
    const int a = __builtin_amdgcn_readfirstlane(wave);
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.U + ((int64_t)ks * nchunk) * WINO_U_FLOATS), 0,
                                                          nchunk * WINO_U_FLOATS * 4, 0x00020000);
    const int u_off = ((a * 6 * 2 + h) * 64 + i32) * 16;                               // + (p*2*64 + kb*32)*16 bytes, + chunk*48 KB
    f32x4 uA[12];
    auto filter_piece = [&](int ch, f32x4(&u)[12], int i) {              // 12 pieces: one buffer_load_dwordx4 each
        u[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, ch * (WINO_U_FLOATS * 4) + ((i >> 1) * 128 + (i & 1) * 32) * 16, 0));
    };
#pragma unroll
    for (int i = 0; i < 12; ++i) filter_piece(0, uA, i);
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
    //     both channel blocks = 12 x 16 bytes per chunk, loaded straight from L2 (the filter slices of this XCD) one chunk ahead;
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
    uint32_t areg[2][2][4];                                               // LDS byte address in stage 0: [row0 / row1][columns 0-3 / 4-5][chunk of the super-chunk]
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
        const int py = 2 * ty + (rs ? row1 : row0);
        const int p0 = 2 * (((py & 3) + 4 * (py >> 3)) * 18 + 4 * tx) + ((py >> 2) & 1);   // slot of column 0 of the tile; column c: + 2 c
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            const int rot = ((tx + cl) & 3) + 4 * ((py >> 1) & 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) areg[rs][cl][c] = lds_base + p0 * 128 + ((2 * c + h + rot) & 7) * 16;
        }
    }
    uint32_t amini[2];                                                    // LDS byte address in mini stage 0: [row0 / row1]; column c: + ((c & 3) 5 + (c >> 2)) 32
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) amini[rs] = lds_base + 2 * WINO_SB_FLOATS * 4 + ((2 * ty + (rs ? row1 : row0)) * 21 + tx) * 32 + h * 16;
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in + base_px * P.in_stride), 0,
                                                          n_img * HWi * P.in_stride * 4, 0x00020000);
    // Where a patch pixel lives in the source: thread t works out pixel t (and t + 256) of the 18 x 18 patch ONCE -- canvas row ->
    // (grid row, row inside the image), canvas column -> (grid column, column) -- and parks its pixel index (-1: outside every image:
    // the loads then use a buffer offset that reads 0.0) in LDS; the lanes look their pieces up there: two divisions per thread
    // instead of two per lane and piece.
    int* pix_tab = reinterpret_cast<int*>(lds + 2 * WINO_SB_FLOATS + 2 * 3072);       // 324 ints behind the mini stages
#pragma unroll
    for (int t = tid; t < 325; t += 256) {
        const int py = t / 18, px = t - py * 18, vy = y0 - 1 + py, vx = x0 - 1 + px;
        int m, n;
        const int gy = cell(vy < 0 ? 0 : vy, Hv, rHv, m), gx = cell(vx < 0 ? 0 : vx, Wv, rWv, n), img = m * gcols + n;
        const bool ok = (t < 324) & (vy >= 0) & (gy < H) & (vx >= 0) & (gx < W) & (n < gcols) & (img < n_img);
        pix_tab[t] = ok ? img * HWi + gy * W + gx : -1;              // entry 324 = -1: the "no pixel" slots of the fills point here
    }
    __syncthreads();
    auto byte_offset = [&](int pix, int part4) {
        const int o = pix >= 0 ? (pix * P.in_stride + part4) * 4 : 0x7FFFFF00;
        return o;
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

    f32x16 acc[12];                                                      // [p][kb]; never cleared: chunk 0's first k-step multiplies into a zero C

    f32x4 x[12], uB[12], vA[6], vB[6], t[6], w6[4];                      // x[row][c], u[p][kb] (uA: above), v[p]: 4 channels each
    if (POD_WINO_ELIM) {                                                  // (elimination builds: operands that were never loaded still need values)
#pragma unroll
        for (int i = 0; i < 12; ++i) x[i] = uA[i] = uB[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane + i);
#pragma unroll
        for (int i = 0; i < 6; ++i) vA[i] = vB[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f} * (float)(lane - i);
    }
    // 12 pieces: one ds_read_b128 each, stage par, chunk c of its super-chunk.  Issued as asm: hipcc orders every LDS read it can
    // see behind ALL pending LDS-DMA (vmcnt(0): it cannot tell the two stages apart), which would drain the pieces flying into the
    // other stage; so the reads are hidden from it and their completion is counted by hand (WINO_WAIT_LGKM0 before the transform).
#define WINO_READ(par, c, i)                                                                                                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(areg[(i) / 6][((i) % 6) >> 2][c]), "i"((par) * WINO_SB_FLOATS * 4 + ((i) % 6) * 256))
#define WINO_READ_MINI(which, i)                                                                                                     \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(amini[(i) / 6]), "i"((which) * 12288 + ((((i) % 6) & 3) * 5 + (((i) % 6) >> 2)) * 32))
    // packed fp32 arithmetic on the halves of a 4-channel value: r = q * k + p
    auto pk_fma = [](f32x2 k2, f32x2 q, f32x2 p) {
        f32x2 r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(k2), "v"(q), "v"(p));
        return r;
    };
    auto pk_add = [](f32x2 p, f32x2 q) {
        f32x2 r;
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(p), "v"(q));
        return r;
    };
    auto pk_sub = [](f32x2 p, f32x2 q) {
        f32x2 r;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(p), "v"(q));
        return r;
    };
    auto fma4 = [&](float k, f32x4 q, f32x4 p) {   // q * k + p
        const f32x2 k2 = f32x2{k, k};
        const f32x2 l = pk_fma(k2, f32x2{q.x, q.y}, f32x2{p.x, p.y}), hq = pk_fma(k2, f32x2{q.z, q.w}, f32x2{p.z, p.w});
        return f32x4{l.x, l.y, hq.x, hq.y};
    };
    auto add4 = [&](f32x4 p, f32x4 q) { const f32x2 l = pk_add(f32x2{p.x, p.y}, f32x2{q.x, q.y}), hq = pk_add(f32x2{p.z, p.w}, f32x2{q.z, q.w}); return f32x4{l.x, l.y, hq.x, hq.y}; };
    auto sub4 = [&](f32x4 p, f32x4 q) { const f32x2 l = pk_sub(f32x2{p.x, p.y}, f32x2{q.x, q.y}), hq = pk_sub(f32x2{p.z, p.w}, f32x2{q.z, q.w}); return f32x4{l.x, l.y, hq.x, hq.y}; };
    // row a of V = Bt4 d Bt6^T for the lane's tile, 10 pieces of two 4-channel operations:
    //   t_c = x0_c + s x1_c (6);  V0 = 4 t0 - 5 t2 + t4;  V5 = 4 t1 - 5 t3 + t5;  e = t4 - 4 t2, o = t3 - 4 t1: V1 = e + o, V2 = e - o;
    //   f = t4 - t2, g = 2 (t3 - t1): V3 = f + g, V4 = f - g
    auto transform_piece = [&](f32x4(&v)[6], int i) {
        if (i < 3) {
#pragma unroll
            for (int c = 2 * i; c < 2 * i + 2; ++c) t[c] = fma4(sgn, x[6 + c], x[c]);
        } else if (i == 3) {
            w6[0] = fma4(-5.0f, t[2], t[4]);          // t4 - 5 t2
            w6[1] = fma4(-5.0f, t[3], t[5]);          // t5 - 5 t3
        } else if (i == 4) {
            v[0] = fma4(4.0f, t[0], w6[0]);
            v[5] = fma4(4.0f, t[1], w6[1]);
        } else if (i == 5) {
            w6[0] = fma4(-4.0f, t[2], t[4]);          // e
            w6[1] = fma4(-4.0f, t[1], t[3]);          // o
        } else if (i == 6) {
            v[1] = add4(w6[0], w6[1]);
            v[2] = sub4(w6[0], w6[1]);
        } else if (i == 7) {
            w6[2] = sub4(t[4], t[2]);                 // f
            w6[3] = sub4(t[3], t[1]);                 // g / 2
        } else if (i == 8) {
            v[3] = fma4(2.0f, w6[3], w6[2]);
        } else if (i == 9) {
            v[4] = fma4(-2.0f, w6[3], w6[2]);
        }
    };

    // One chunk = 8 input channels = 48 MFMAs (k-step j / 12 = channel of the lane's four, accumulator j % 12 = (p, kb)) with the next
    // chunk's work slotted behind them, at most one memory instruction per MFMA (the order is pinned in the source: the four waves
    // of the workgroup run in lock step, memory instructions issued in a burst queue behind each other and stall the in-order
    // instruction streams): patch reads of chunk ch+1 (LDS), filter loads of chunk ch+1 (L2), its transform, and a quarter of the
    // LDS-DMA of a later SUPER-CHUNK (4 chunks, two stages).  Chunk c of super-chunk s reads stage s & 1 (c = 3: the first chunk of
    // s + 1 from the other stage).  Super-chunk s + 1 is fetched into the stage s - 1 left behind, 4 instructions per wave during
    // each of the chunks (s-1, 3), (s, 0), (s, 1) -- every piece has a whole chunk to land before the barrier that publishes it.
#define WINO_MFMA(V, U, j)                                                                                                   \
    acc[(j) % 12] = __builtin_amdgcn_mfma_f32_32x32x2f32(U[(j) % 12][(j) / 12], V[((j) % 12) >> 1][(j) / 12], acc[(j) % 12], 0, 0, 0)
    const int last = nchunk - 1, last_s = last >> 2;
#pragma unroll
    for (int r = 0; r < 3; ++r) mini_piece(0, r);
#pragma unroll
    for (int r = 0; r < 3; ++r) mini_piece(last < 1 ? 0 : 1, r);
    WINO_STAMP(9);
    main_offsets();                                    // (behind the first loads: their latency hides it)
    WINO_STAMP(10);
#pragma unroll
    for (int i = 0; i < 12; ++i) patch_piece(lds, 0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) patch_piece(lds + WINO_SB_FLOATS, last_s < 1 ? last_s : 1, i);
    WINO_STAMP(11);
    __builtin_amdgcn_s_waitcnt(WINO_WAIT_VM16);        // the mini stages and the filters of chunk 0 have landed; the 16 pieces of the stages fly on
    __builtin_amdgcn_s_barrier();
    WINO_STAMP(1);
    WINO_READ_MINI(0, 0); WINO_READ_MINI(0, 1); WINO_READ_MINI(0, 2); WINO_READ_MINI(0, 3); WINO_READ_MINI(0, 4); WINO_READ_MINI(0, 5);
    WINO_READ_MINI(0, 6); WINO_READ_MINI(0, 7); WINO_READ_MINI(0, 8); WINO_READ_MINI(0, 9); WINO_READ_MINI(0, 10); WINO_READ_MINI(0, 11);
    __builtin_amdgcn_s_waitcnt(WINO_WAIT_LGKM0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) transform_piece(vA, i);
    // the transform's packed instructions are asm: hipcc neither pads the VALU-write -> MFMA-operand hazard behind them nor keeps the
    // first MFMA from being scheduled up among them (it reads a stale operand then: measured) -- fence and pad by hand
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");
    WINO_STAMP(2);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](auto first, auto c_t, auto par_t, int sc, f32x4(&vC)[6], f32x4(&uC)[12], f32x4(&vN)[6], f32x4(&uN)[12]) {
        constexpr int c = decltype(c_t)::value, par = decltype(par_t)::value;
        constexpr int rc = (c + 1) & 3, rpar = c == 3 ? par ^ 1 : par;            // what the reads fetch: the NEXT chunk's patch
        constexpr int ph = c == 3 ? 0 : c + 1, dpar = c == 3 ? par : par ^ 1;     // fill phase (c == 2: none) and the stage being filled
        const int ch = 4 * sc + c, c1 = ch + 1 < nchunk ? ch + 1 : last;
        const int fs0 = c == 3 ? sc + 2 : sc + 1, fs = fs0 < last_s ? fs0 : last_s;
        float* wr = lds + dpar * WINO_SB_FLOATS;
        wino_static_for([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            if constexpr (decltype(first)::value && j < 12) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uC[j][0], vC[j >> 1][0], zero16, 0, 0, 0);
            else WINO_MFMA(vC, uC, j);
            if constexpr (j < 12) {
                if constexpr (decltype(first)::value) WINO_READ_MINI(1, j);                  // chunk 1's patch: the second mini stage
                else if (!(POD_WINO_ELIM & 1)) WINO_READ(rpar, rc, j);
            }
            else if constexpr (j < 24) { if (!(POD_WINO_ELIM & 2)) filter_piece(c1, uN, j - 12); }
            else if constexpr (j == 24) __builtin_amdgcn_s_waitcnt(WINO_WAIT_LGKM0);         // the 12 reads (issued 12+ MFMAs ago)
            else if constexpr (j >= 25 && j < 35) { if (!(POD_WINO_ELIM & 8)) transform_piece(vN, j - 25); }
            else if constexpr (j >= 36 && j <= 45 && (j - 36) % 3 == 0) { if (c != 2 && !(POD_WINO_ELIM & 4)) patch_piece(wr, fs, 4 * ph + (j - 36) / 3); }
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 48>{});
        if (!(POD_WINO_ELIM & 16)) {
            // the filters of the next chunk and every patch piece issued before this chunk have landed; this chunk's 4 pieces may fly on
            __builtin_amdgcn_s_waitcnt(c != 2 ? WINO_WAIT_VM4 : WINO_WAIT_VM0);
            __builtin_amdgcn_s_barrier();
        }
    };
    using std::integral_constant;
    chunk(std::true_type{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, 0, vA, uA, vB, uB);
    for (int base = 0;; base += 8) {
#define WINO_CHUNK(t)                                                                                                              \
    if (base + (t) >= nchunk) break;                                                                                               \
    if ((t) & 1) chunk(std::false_type{}, integral_constant<int, (t) & 3>{}, integral_constant<int, ((t) >> 2) & 1>{}, (base + (t)) >> 2, vB, uB, vA, uA); \
    else chunk(std::false_type{}, integral_constant<int, (t) & 3>{}, integral_constant<int, ((t) >> 2) & 1>{}, (base + (t)) >> 2, vA, uA, vB, uB);
        WINO_CHUNK(1) WINO_CHUNK(2) WINO_CHUNK(3) WINO_CHUNK(4) WINO_CHUNK(5) WINO_CHUNK(6) WINO_CHUNK(7) WINO_CHUNK(8)
#undef WINO_CHUNK
    }
#undef WINO_MFMA
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
    if (P.k_planes > 0) {
        // NCHW planes: thread -> (channel, row of the block, 4 pixels along x = one tile's columns); 64-byte runs per (channel, row)
        const int oy = (tid >> 2) & 15, ox = (tid & 3) * 4;
        int m;
        const int gy = cell(y0 + oy, Hv, rHv, m);
        int64_t px0[4];                                   // output pixel (of plane 0) per column, -1: not a pixel of any image
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int n;
            const int gx = cell(x0 + ox + e, Wv, rWv, n), img = m * gcols + n;
            px0[e] = (n < gcols && img < n_img && gx < W && gy < H) ? (out_px + (int64_t)img * HWi) * P.k_planes + (int64_t)gy * W + gx : -1;
        }
        const bool vec = px0[0] >= 0 && px0[3] == px0[0] + 3 && (px0[0] & 3) == 0 && (HWi & 3) == 0;
        const int tile = (oy >> 1) * 4 + (tid & 3);
#pragma unroll 2
        for (int it = 0; it < 16; ++it) {
            const int k = it * 4 + (tid >> 6), kg = ks * 64 + k;
            if (kg >= P.k_planes) continue;
            const float bias = P.bias ? P.bias[kg] : 0.0f;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* r = lds + tile * TS + e * 64 + k;                 // Z[a][tile][e][k] at + a * ZA
                y[e] = (oy & 1) == 0 ? (r[0] + r[ZA]) + r[2 * ZA] : (r[ZA] - r[2 * ZA]) - r[3 * ZA];
            }
            f32x4 v = f32x4{y[0], y[1], y[2], y[3]} + bias;
            if (P.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            float* plane = P.out + (int64_t)kg * HWi;
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
        if (P.bias) {
            bias0 = *reinterpret_cast<const f32x4*>(P.bias + kg);
            bias1 = *reinterpret_cast<const f32x4*>(P.bias + kg + 4);
        }
        const uint64_t drop_key = P.thresh ? dropout_key(P.seed, P.epoch) : 0ull;
        int n;
        const int gx = cell(x0 + ox, Wv, rWv, n);
        const bool col_ok = n < gcols && gx < W;
        int m, gy = cell(y0 + odd, Hv, rHv, m) - 2;                                   // canvas row y0 + 2 it + odd: grid row m, image row gy (H: the separator)
        const float* rbase = lds + (ox >> 2) * TS + (ox & 3) * 64 + k8 + (odd ? ZA : 0);      // Z[a][tile][ox & 3][k8] of row a = odd
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            gy += 2;
            if (gy >= Hv) {
                gy -= Hv;
                ++m;
            }
            const int img = m * gcols + n;
            if (!col_ok || gy >= H || img >= n_img) continue;
            const float* r = rbase + it * 4 * TS;                                     // tile (it, ox >> 2)
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r), a1 = *reinterpret_cast<const f32x4*>(r + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(r + ZA), b1 = *reinterpret_cast<const f32x4*>(r + ZA + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(r + 2 * ZA), c1 = *reinterpret_cast<const f32x4*>(r + 2 * ZA + 4);
            f32x4 v0 = (odd ? (a0 - b0) - c0 : (a0 + b0) + c0) + bias0, v1 = (odd ? (a1 - b1) - c1 : (a1 + b1) + c1) + bias1;
            if (P.relu) {
                v0.x = fmaxf(v0.x, 0.f); v0.y = fmaxf(v0.y, 0.f); v0.z = fmaxf(v0.z, 0.f); v0.w = fmaxf(v0.w, 0.f);
                v1.x = fmaxf(v1.x, 0.f); v1.y = fmaxf(v1.y, 0.f); v1.z = fmaxf(v1.z, 0.f); v1.w = fmaxf(v1.w, 0.f);
            }
            const int64_t e = (out_px + (int64_t)img * HWi + (int64_t)gy * W + gx) * P.out_stride + kg;      // a multiple of 8
            if (P.thresh && !(POD_WINO_ELIM & 64)) {
                const uint64_t ctr = P.offset + (uint64_t)(e >> 3);
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
            *reinterpret_cast<f32x4*>(P.out + e) = v0;
            *reinterpret_cast<f32x4*>(P.out + e + 4) = v1;
        }
    }
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
extern "C" int pod_wino_trace_dump(long long* host, int32_t n_workgroups) {   // diagnostics build only (not in include/pod_mi355x.h)
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    if (n_workgroups > 8192) n_workgroups = 8192;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pod::g_wino_trace), (size_t)n_workgroups * 16 * sizeof(long long)) != hipSuccess) return POD_E_LAUNCH;
    return POD_OK;
}
#endif

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
                                const uint64_t* epoch, pod_stream_t stream) {
    if (!in || !out || in == out || !U || !blocks || n_blocks < 0 || C < 8 || (C & 7) != 0 || K < 64 || (K & 63) != 0 ||
        !(p >= 0.0f && p < 1.0f) || k_planes < 0 || k_planes > K || (k_planes > 0 && p != 0.0f))
        return POD_E_INVALID;
    const int32_t KS = K / 64;
    if (KS != 1 && KS != 2 && KS != 4 && KS != 8) return POD_E_INVALID;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(U) |
          reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(blocks)) & 15u) != 0)
        return POD_E_INVALID;
    if (n_blocks == 0) return POD_OK;
    // the 130 KB dynamic-LDS opt-in is a PER-DEVICE function attribute: once per device ordinal this process launches on
    static std::once_flag once[64];
    static hipError_t attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return POD_E_LAUNCH;
    std::call_once(once[dev], [dev] {
        attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(pod::k_wino_conv3x3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        pod::WINO_LDS_BYTES);
    });
    if (attr[dev] != hipSuccess) return POD_E_LAUNCH;
    pod::WinoParams P;
    P.in = in; P.out = out; P.U = U; P.bias = bias; P.blocks = reinterpret_cast<const int4*>(blocks);
    P.n_blocks = n_blocks; P.C = C; P.K = K; P.KS = KS; P.in_stride = C; P.out_stride = K; P.relu = relu; P.k_planes = k_planes;
    P.thresh = POD_DROPOUT_THRESH16(p);
    P.scale = 1.0f / (1.0f - p);
    P.seed = seed; P.offset = offset;
    P.c_split = 0; P.split_out_stride = 0; P.epoch = epoch; P.replicas = 1; P.live = nullptr;
    const int64_t grid = pod::wino_grid(KS, n_blocks);
    if (grid > 0x7FFFFFFFLL) return POD_E_INVALID;
    hipLaunchKernelGGL(pod::k_wino_conv3x3, dim3((unsigned)grid), dim3(256), pod::WINO_LDS_BYTES, (hipStream_t)stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}
