// K12's convolution (k12_wino_conv_split.hip: Winograd F(2,3) x F(4,3), fp32 products from two-term f16 splits on the 16-bit matrix cores;
// probabilistic_retinanet.py:403-484 looped over MC runs and FPN levels) as ONE WORKGROUP OF EIGHT WAVEFRONTS, two per SIMD (round 6).
//
// Why.  K12 holds 454 registers per lane: one wavefront per SIMD, and over its life that wavefront issues during 63 % of its cycles, waits
// during 23 % (vmcnt returns in order: a filter term from L2 cannot be seen before the older patch piece from HBM has landed) and does
// neither during 15 % (profiles/r05_k12_sq_counters.txt) -- the third that is not issue has nobody to go to.  Here the two wavefronts of a
// SIMD share ROW a of the 4 x 6 position grid: wavefront (a, hp) owns positions 3 hp .. 3 hp + 2 for the block's 32 tiles and all 64
// output channels -- 96 accumulators, 12 filter loads, half of the column transform and of the split per chunk; only the row combination
// (5 of 6 columns each) and the patch reads are done twice.  Everything a lane computes is the operation K12 computes for that value, in the
// same order: positions accumulate independently, the column transform's six outputs split 3 / 3 without a shared intermediate, the
// output transform's two halves meet in LDS -- the results are K12's BIT FOR BIT (tests/test_wino_conv_gpu.py).
//
//   per chunk and wavefront     K12 (4 wavefronts)      here (8 wavefronts)
//   MFMAs                       36                      18
//   filter loads (16 B / lane)  24                      12
//   patch reads (ds_read_b128)  24                      20
//   row combination             48                      40
//   column transform            96                      48
//   split                       120                     60
//   patch fill (load + park)    6 + 6                   3 + 3
#include "pod_wino.h"      // (pod_compare_amd/csrc: the experiment build adds it to the include path)

namespace pod {

typedef uint32_t vu32x4 __attribute__((ext_vector_type(4)));
constexpr int W8_US_BYTES = 24 * 2 * 2 * 64 * 16;          // = WINO_US_BYTES of k12: pre-split filter terms of a 16-channel chunk, 96 KB
constexpr int W8_U_TOP = 14, W8_V_TOP = 9;                 // (k12: WINO_U_TOP, WINO_V_TOP)
constexpr int W8_LDS_BYTES = 136 * 1024;                   // two patch stages (96 KB) + two mini stages (24 KB) + pixel table + the DMA offsets (12 KB); the output staging needs 130 KB
constexpr int W8_WAIT_VM12 = 0x007C;                        // lgkmcnt(0) vmcnt(12)
#ifndef W8_PARK_LATE
#define W8_PARK_LATE 0
#endif
#ifndef W8_ELIM
#define W8_ELIM 0      // tagged experiment builds only (tools/wino_elim16.sh): 1 patch reads, 2 filter loads, 4 patch fill, 8 transform + split compiled out
#endif
#ifndef W8_DMA_FILL
#define W8_DMA_FILL 0  // 1: the K loop fills the patch stages by LDS-DMA (all six pieces of a super-chunk during its predecessor's second chunk), no staging
#endif                 //    registers -- which pay for a filter ring of four positions, i.e. loads THREE positions ahead
constexpr bool W8_DMA = W8_DMA_FILL != 0;
constexpr int W8_LEAD = W8_DMA ? 3 : 2;                     // positions the filter loads run ahead of their MFMAs
constexpr int W8_RING = W8_LEAD + 1;                        // ... in a ring of LEAD + 1 positions: global position 3 q + p lives in uP[(3 q + p) % RING]

__global__ void __launch_bounds__(512) k_wino_conv3x3_split8(const WinoParams P) {
    extern __shared__ __attribute__((aligned(128))) float lds[];      // (128: the patch reads toggle address bits 4 and 6 by XOR)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave: a scalar register)
    int ks, tb;
    wino_schedule(P.KS, P.n_blocks, ks, tb);
    if (tb >= P.n_blocks) return;
    WINO_STAMP(0);
    WINO_STAMP_WALL(12);
    uint32_t need = ~0u;                                                  // need bits of this thread's patch pixel (tid): all, unless sparse
    if (P.live) {                                                         // sparse launch: slot -> live entry {record, need bits} (k12, k15)
        if (tb >= P.live[0]) return;
        const int32_t* const ent = P.live + POD_SPARSE_LIVE_HEAD + (int64_t)POD_SPARSE_LIVE_STRIDE * tb;
        tb = ent[0];
        need = (uint32_t)ent[1 + (tid >> 5 < 11 ? tid >> 5 : 10)];
    }
    // wavefront (a, hp): waves w and w + 4 sit on the same SIMD (round-robin placement; nothing depends on it but the overlap)
    const int a = wave & 3, hp = wave >> 2;
    uint32_t slot_e[6];                                                   // this lane's 6 pixel slots of a stage fill (48 instructions, 6 per wavefront)
#pragma unroll
    for (int i = 0; i < 6; ++i) slot_e[i] = g_wino_slots.v[48 * wave + 8 * i + (lane >> 3)];
    int mini_pidx[3];                                                     // the patch pixel of its 3 slots of a mini-stage fill (324: none)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int pp = ((a * 3 + r) * 64 + lane) >> 1, py = pp / 21, pi = pp - py * 21, px = 4 * (pi % 5) + pi / 5;
        mini_pidx[r] = py < 18 && pi < 20 && px < 18 ? py * 18 + px : 324;
    }
    const int nchunk_all = P.C >> 4;
    const int nchunk = P.c_split ? P.c_split : nchunk_all;
    const int chunk0 = (int)blockIdx.y * nchunk;
    const int i32 = lane & 31, h = lane >> 5;
    const int set = (tb >= P.sets.first[1] ? 1 : 0) + (tb >= P.sets.first[2] ? 1 : 0) + (tb >= P.sets.first[3] ? 1 : 0);
    const float* const set_U = P.sets.U[set];
    const float* const set_in = P.sets.in[set];
    float* const set_out = P.sets.out[set];
    const float* const set_bias = P.sets.bias[set];
    const float in_amax_slot = wino_load_amax(P.sets.in_amax[set]);
    const float u_amax = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(set_U) + (int64_t)P.KS * (P.C >> 4) * W8_US_BYTES);
    float* const set_out_amax = P.sets.out_amax[set];
    const uint64_t set_offset = P.sets.offset[set];
    const int set_replicas = P.sets.replicas[set], set_k_planes = P.sets.k_planes[set];
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(set_U) + ((int64_t)ks * nchunk_all + chunk0) * W8_US_BYTES), 0,
                                                          nchunk * W8_US_BYTES, 0x00020000);
    // Us[ks][chunk][q = 6 a + p][kb][term][h][j][8 f16]: this wavefront's positions are p = 3 hp + pp
    const int u_off = ((a * 6 + 3 * hp) * 4 * 64 + h * 32 + i32) * 16;    // + ((pp*2 + kb)*2 + term) KB, + chunk * 96 KB
    vu32x4 uP[W8_RING][4];                                                 // the filter terms of the positions in flight: [kb][term]
    auto filter_piece = [&](int q16, int pp, vu32x4(&u)[4], int i) {
        if (W8_ELIM & 2) return;
        u[i] = __builtin_bit_cast(vu32x4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, u_off, q16 * W8_US_BYTES + (pp * 4 + i) * 1024, 0));
    };
#pragma unroll
    for (int pp = 0; pp < W8_LEAD; ++pp)
#pragma unroll
        for (int i = 0; i < 4; ++i) filter_piece(0, pp, uP[pp], i);
    const int4 desc = P.blocks[tb];
    const int64_t base_px = desc.x, out_px = desc.y;
    const int gcols = (desc.z >> 24) & 0xFF, H = (desc.z >> 12) & 0xFFF, W = desc.z & 0xFFF, n_img = (desc.w >> 24) & 0xFF;
    const int y0 = ((desc.w >> 12) & 0xFFF) * 16, x0 = (desc.w & 0xFFF) * 16, Wv = W + 1, Hv = H + 1, HWi = H * W;
    const float rWv = 1.0f / (float)Wv, rHv = 1.0f / (float)Hv;
    auto cell = [](int v, int step, float rstep, int& idx) {              // canvas coordinate -> (grid index, coordinate inside the cell) (k12)
        int n = (int)((float)v * rstep);
        n -= n * step > v ? 1 : 0;
        n += (n + 1) * step <= v ? 1 : 0;
        idx = n;
        return v - n * step;
    };

    // ---- operands: k12's layouts (patch stages of 32-channel super-chunks, swizzled 128-byte pixel slots; two 8-channel mini stages for
    // chunk 0; filters from L2 straight into registers).  Row a of Bt4 d = x[row0] + sgn x[row1].
    const int row0 = a == 0 ? 0 : a == 2 ? 2 : 1, row1 = a == 2 ? 1 : a == 3 ? 3 : 2;
    const float sgn = a == 1 ? 1.0f : -1.0f;
    const int ty = i32 >> 2, tx = i32 & 3;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
    // Patch stage layout: k12's 128-byte pixel slots, with the eight 16-byte parts of a pixel XOR-swizzled: part P sits at sub-slot P ^ rot,
    // rot = ((px >> 2) & 3) + 4 ((py >> 1) & 1) -- as conflict-free as k12's rotation (a bijection of the sub-slots per pixel either way), and
    // the lane's part 4 c16 + 2 h + hf is then reached from ONE address per (row, column group) by flipping bits 6 and 4: four address
    // registers instead of k12's sixteen (this kernel has 256 registers per lane, not 512).
    uint32_t areg[2][2];                                                  // [row0 / row1][columns 0-3 / 4-5]: LDS byte address of part 2 h in stage 0
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
        const int py = 2 * ty + (rs ? row1 : row0);
        const int p0 = 2 * (((py & 3) + 4 * (py >> 3)) * 18 + 4 * tx) + ((py >> 2) & 1);   // slot of column 0 of the tile; column c: + 2 c
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            const int rot = ((tx + cl) & 3) + 4 * ((py >> 1) & 1);
            areg[rs][cl] = lds_base + p0 * 128 + (((2 * h) ^ rot) & 7) * 16;
        }
    }
    uint32_t amini[2];
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) amini[rs] = lds_base + 2 * WINO_SB_FLOATS * 4 + h * 12288 + ((2 * ty + (rs ? row1 : row0)) * 21 + tx) * 32;
    const auto r_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(set_in + base_px * P.in_stride + chunk0 * 16), 0,
                                                          n_img * HWi * P.in_stride * 4 - chunk0 * 64, 0x00020000);
    int* pix_tab = reinterpret_cast<int*>(lds + 2 * WINO_SB_FLOATS + 2 * 3072);       // 325 ints behind the mini stages
    if (tid < 325) {
        const int t = tid, py = t / 18, px = t - py * 18, vy = y0 - 1 + py, vx = x0 - 1 + px;
        int m, n;
        const int gy = cell(vy < 0 ? 0 : vy, Hv, rHv, m), gx = cell(vx < 0 ? 0 : vx, Wv, rWv, n), img = m * gcols + n;
        const bool ok = (t < 324) & (vy >= 0) & (gy < H) & (vx >= 0) & (gx < W) & (n < gcols) & (img < n_img) & (((need >> (t & 31)) & 1u) != 0);
        pix_tab[t] = ok ? img * HWi + gy * W + gx : -1;              // entry 324 = -1: the "no pixel" slots of the fills point here
    }
    __syncthreads();
    auto byte_offset = [&](int pix, int part4) { return pix >= 0 ? (pix * P.in_stride + part4) * 4 : 0x7FFFFF00; };
    int dmini[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) dmini[r] = pix_tab[mini_pidx[r]];
#pragma unroll
    for (int r = 0; r < 3; ++r) dmini[r] = byte_offset(dmini[r], 4 * (lane & 1));
    int doff[6];
    auto main_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) doff[i] = pix_tab[slot_e[i] & 0xFFFF];
#pragma unroll
        for (int i = 0; i < 6; ++i) doff[i] = byte_offset(doff[i], 4 * (((lane & 7) ^ (int)(slot_e[i] >> 16)) & 7));      // sub-slot q holds part q ^ rot
    };
    typedef __attribute__((address_space(3))) void lds_void;
    int* const doff_tab = pix_tab + 328 + wave * 384 + lane;          // (W8_DMA) the six source offsets of a stage fill, parked in LDS: see below
    // mini stage hp (channels 8 hp .. 8 hp + 7 of chunk 0) is filled by the four wavefronts of half hp: 3 instructions each
    auto mini_piece = [&](int r) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(lds + 2 * WINO_SB_FLOATS + hp * 3072 + (a * 3 + r) * 256), 16, dmini[r], hp * 32, 0, 0);
    };
    auto patch_piece = [&](float* stage, int sc, int i) {    // 1 KB (8 pixels x 32 channels) of super-chunk sc, straight into LDS: instruction 6 wave + i of 48
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(stage + (wave * 6 + i) * 256), 16, doff[i], sc * 128, 0, 0);
    };
    f32x4 stg[W8_DMA ? 1 : 3];                                            // register-staged fills of the K loop (k12): 3 pieces per chunk and wavefront
    const uint32_t stg_addr = lds_base + wave * 6144 + lane * 16;         // + stage * 48 KB + piece * 1 KB
    auto stage_load = [&](int k, int sc, int i) {
        if ((W8_ELIM & 4) || W8_DMA) return;
        stg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, doff[i], sc * 128, 0));
    };
    auto stage_write = [&](int par, int k, int i) {
        if ((W8_ELIM & 4) || W8_DMA) return;
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>((uintptr_t)(stg_addr + par * (WINO_SB_FLOATS * 4) + i * 1024)) = stg[k];
    };

    const float sv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wino_pow2_scale(wino_reduce_amax(in_amax_slot), W8_V_TOP))));
    const int last = nchunk - 1, last_s = last >> 1;
    const int sc1 = last_s < 1 ? last_s : 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) mini_piece(r);
    main_offsets();                                    // (behind the first loads: their latency hides it)
    if (W8_DMA) {                                      // the K loop reads them back one at a time in front of their DMA instruction: six registers less across it
#pragma unroll
        for (int i = 0; i < 6; ++i) doff_tab[64 * i] = doff[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) patch_piece(lds, 0, i);
#pragma unroll
    for (int i = 0; i < 3; ++i) patch_piece(lds + WINO_SB_FLOATS, sc1, i);       // pieces 0..2 of super-chunk 1 straight into stage 1 ...
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                                // ... its pieces 3..5 through registers (chunk 0 parks them), or by DMA as well
        if (W8_DMA) patch_piece(lds + WINO_SB_FLOATS, sc1, 3 + i);
        else stage_load(i, sc1, 3 + i);
    }
    __builtin_amdgcn_s_waitcnt(W8_WAIT_VM12);          // the mini stages and the first filter terms have landed; 9 DMA pieces and 3 register pieces fly on
    __builtin_amdgcn_s_barrier();
    WINO_STAMP(1);

    f32x16 acc[6];                                                       // [pp][kb]; never cleared: chunk 0's first product multiplies into a zero C
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    using std::integral_constant;

    // ---- the K loop of half HP: columns HP .. HP + 4 of the row, positions 3 HP .. 3 HP + 2
    auto kloop = [&](auto hp_t) __attribute__((always_inline)) {
        constexpr int HP = decltype(hp_t)::value;
        f32x4 x[4];                                                      // raw patch, TWO columns of one 4-channel half at a time: x[2 row + j], column c = C0 + j + HP
        vu32x4 Vb[3][2];                                                  // the transformed patch as f16 operands: [pp][term], 8 channels
        float tN[5][4], vN[2][3][4];                                     // the NEXT chunk's transform in flight (k12)
        // reads of columns c' = C0 + j (j = i & 1, row i >> 1) into x[i]; batches C0 = 0, 2 (two columns) and 4 (one column: i = 0, 2)
#define W8_READ(par, c16, hf, C0, i)                                                                                                  \
    if (!(W8_ELIM & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(areg[(i) >> 1][((C0) + ((i) & 1) + HP) >> 2] ^ (uint32_t)(64 * (c16) + 16 * (hf))), "i"((par) * WINO_SB_FLOATS * 4 + ((C0) + ((i) & 1) + HP) * 256))
#define W8_READ_MINI(hf, C0, i)                                                                                                       \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[i]) : "v"(amini[(i) >> 1]), "i"((hf) * 16 + (((((C0) + ((i) & 1) + HP)) & 3) * 5 + (((C0) + ((i) & 1) + HP) >> 2)) * 32))
#define W8_READ_2(M, C0, ...) M(__VA_ARGS__, C0, 0); M(__VA_ARGS__, C0, 2); M(__VA_ARGS__, C0, 1); M(__VA_ARGS__, C0, 3)
#define W8_READ_1(M, C0, ...) M(__VA_ARGS__, C0, 0); M(__VA_ARGS__, C0, 2)
#define W8_READS_LANDED() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
        // tN[c'] = x[row0][c] + sgn x[row1][c]  (k12's rows_combine: the same fused multiply-add per value)
        auto rows_combine = [&](int c0, int c1) __attribute__((always_inline)) {       // column c' sits in x[c' & 1] / x[2 + (c' & 1)] of its batch
#pragma unroll
            for (int c = c0; c < c1; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) tN[c][e] = __builtin_fmaf(sgn, x[2 + (c & 1)][e], x[c & 1][e]);
        };
        // column transform Bt6, k12's operations for this half's three positions (t_c = tN[c - HP]):
        //   HP 0: v0 = 4 t0 + (t4 - 5 t2), v1 = (t4 - 4 t2) + (t3 - 4 t1), v2 = (t4 - 4 t2) - (t3 - 4 t1)
        //   HP 1: v3 = (t4 - t2) + 2 (t3 - t1), v4 = (t4 - t2) - 2 (t3 - t1), v5 = 4 t1 + (t5 - 5 t3)
        float l1a[4], l1b[4], l1c[4];
        auto columns_level1 = [&](int e) __attribute__((always_inline)) {
            if constexpr (HP == 0) {
                l1a[e] = __builtin_fmaf(-5.0f, tN[2][e], tN[4][e]);                  // w0
                l1b[e] = __builtin_fmaf(-4.0f, tN[2][e], tN[4][e]);                  // ev
                l1c[e] = __builtin_fmaf(-4.0f, tN[1][e], tN[3][e]);                  // od
            } else {
                l1a[e] = __builtin_fmaf(-5.0f, tN[2][e], tN[4][e]);                  // w1 = t5 - 5 t3
                l1b[e] = tN[3][e] - tN[1][e];                                        // f = t4 - t2
                l1c[e] = tN[2][e] - tN[0][e];                                        // g = t3 - t1
            }
        };
        auto columns_level2 = [&](int hf, int e) __attribute__((always_inline)) {
            if constexpr (HP == 0) {
                vN[hf][0][e] = __builtin_fmaf(4.0f, tN[0][e], l1a[e]);
                vN[hf][1][e] = l1b[e] + l1c[e];
                vN[hf][2][e] = l1b[e] - l1c[e];
            } else {
                vN[hf][0][e] = __builtin_fmaf(2.0f, l1c[e], l1b[e]);
                vN[hf][1][e] = __builtin_fmaf(-2.0f, l1c[e], l1b[e]);
                vN[hf][2][e] = __builtin_fmaf(4.0f, tN[0][e], l1a[e]);               // 4 t1 + w1
            }
        };
        auto split_convert = [&](int p, int term) __attribute__((always_inline)) {
            vu32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                w[i] = term == 0 ? wino_f16_pair_scaled(vN[i >> 1][p][2 * (i & 1)], vN[i >> 1][p][2 * (i & 1) + 1], sv)
                                 : wino_f16_pair(vN[i >> 1][p][2 * (i & 1)], vN[i >> 1][p][2 * (i & 1) + 1]);
            Vb[p][term] = w;
        };
        auto split_residual = [&](int p, int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
            for (int i = i0; i < i1; ++i) wino_f16_residual_scaled(Vb[p][0][i], vN[i >> 1][p][2 * (i & 1)], vN[i >> 1][p][2 * (i & 1) + 1], sv);
        };
        auto split_position = [&](int p) __attribute__((always_inline)) {
            split_convert(p, 0); split_residual(p, 0, 4); split_convert(p, 1);
        };
        auto make_v = [&](int hf) __attribute__((always_inline)) {                      // (after rows_combine(0, 5) in two batches)
#pragma unroll
            for (int e = 0; e < 4; ++e) columns_level1(e);
#pragma unroll
            for (int e = 0; e < 4; ++e) columns_level2(hf, e);
        };
        // The next chunk's operands, slotted behind the running chunk's 18 MFMAs one unit of <= 6 independent instructions at a time (k12's
        // software pipeline with half the positions; the patch reads in batches of two columns: 16 registers instead of 40):
        //     A    0 ..  3   the two terms of the running chunk's own LAST position (values computed during the previous chunk)
        //     half 0:  16 units from 4: reads of columns 0, 1 | their row combination (2 units) | reads of 2, 3 | 2 units | reads of 4 | 1 unit |
        //              the column transform (8 units);   half 1: the same from 20
        //     C0 36 .. 39   C1 40 .. 43   the two terms of the next chunk's positions 0, 1 (their MFMAs of the running chunk have issued)
        constexpr int N_UNITS = 44, N_SLOTS = 18;
        auto half_unit = [&](auto K, auto npar_t, auto n16_t, auto hf_t) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value, npar = decltype(npar_t)::value, n16 = decltype(n16_t)::value, hf = decltype(hf_t)::value;
            if constexpr ((W8_ELIM & 8) != 0 && k != 0 && k != 3 && k != 6) return;
            if constexpr (k == 0) { W8_READ_2(W8_READ, 0, npar, n16, hf); }                // columns 0, 1
            else if constexpr (k == 1) { W8_READS_LANDED(); rows_combine(0, 1); }
            else if constexpr (k == 2) rows_combine(1, 2);
            else if constexpr (k == 3) { W8_READ_2(W8_READ, 2, npar, n16, hf); }           // columns 2, 3
            else if constexpr (k == 4) { W8_READS_LANDED(); rows_combine(2, 3); }
            else if constexpr (k == 5) rows_combine(3, 4);
            else if constexpr (k == 6) { W8_READ_1(W8_READ, 4, npar, n16, hf); }           // column 4
            else if constexpr (k == 7) { W8_READS_LANDED(); rows_combine(4, 5); }
            else if constexpr (k < 12) columns_level1(k - 8);
            else columns_level2(hf, k - 12);
        };
        auto unit = [&](auto U, auto npar_t, auto n16_t, auto hasA_t) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            if constexpr ((W8_ELIM & 8) != 0 && (u < 4 || u >= 36)) return;
            if constexpr (u < 4) {
                if constexpr (decltype(hasA_t)::value) {
                    if constexpr (u == 0) split_convert(2, 0);
                    else if constexpr (u == 1) split_residual(2, 0, 2);
                    else if constexpr (u == 2) split_residual(2, 2, 4);
                    else split_convert(2, 1);
                }
            } else if constexpr (u < 20) half_unit(integral_constant<int, u - 4>{}, npar_t, n16_t, integral_constant<int, 0>{});
            else if constexpr (u < 36) half_unit(integral_constant<int, u - 20>{}, npar_t, n16_t, integral_constant<int, 1>{});
            else {
                constexpr int p = (u - 36) / 4, k = (u - 36) % 4;
                if constexpr (k == 0) split_convert(p, 0);
                else if constexpr (k == 1) split_residual(p, 0, 2);
                else if constexpr (k == 2) split_residual(p, 2, 4);
                else split_convert(p, 1);
            }
        };
        // one half's operands back to back (chunks 0 and 1): M = W8_READ_MINI (hf) or W8_READ (par, c16, hf)
#define W8_HALF_SERIAL(hf, M, ...)                                                                                                    \
    W8_READ_2(M, 0, __VA_ARGS__); W8_READS_LANDED(); __builtin_amdgcn_sched_barrier(0); rows_combine(0, 2); __builtin_amdgcn_sched_barrier(0); \
    W8_READ_2(M, 2, __VA_ARGS__); W8_READS_LANDED(); __builtin_amdgcn_sched_barrier(0); rows_combine(2, 4); __builtin_amdgcn_sched_barrier(0); \
    W8_READ_1(M, 4, __VA_ARGS__); W8_READS_LANDED(); __builtin_amdgcn_sched_barrier(0); rows_combine(4, 5); make_v(hf); __builtin_amdgcn_sched_barrier(0)

        W8_HALF_SERIAL(0, W8_READ_MINI, 0);
        W8_HALF_SERIAL(1, W8_READ_MINI, 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) split_position(p);
        __builtin_amdgcn_sched_barrier(0);
        WINO_STAMP(2);

        // chunk q = (super-chunk q >> 1, half c16 = q & 1), stage parity par = (q >> 1) & 1; modes as in k12 (0: first chunk, 1: chunk 1, 2: steady)
        auto chunk = [&](auto mode_t, auto c16_t, auto par_t, auto q4_t, int q) __attribute__((always_inline)) {
            constexpr int mode = decltype(mode_t)::value, c16 = decltype(c16_t)::value, par = decltype(par_t)::value, Q4 = decltype(q4_t)::value;   // Q4 = q % 4
            constexpr int n16 = c16 ^ 1, npar = c16 == 1 ? par ^ 1 : par;
            const int qn = q + 1 <= last ? q + 1 : last;
            // stage traffic (k12, three pieces per wavefront): chunk (s, 0) parks pieces 3..5 of super-chunk s + 1 in the other stage and loads
            // 0..2 of s + 2; chunk (s, 1) parks those in its OWN stage and loads 3..5 of s + 2
            const int ls0 = (q >> 1) + 2, ls = ls0 < last_s ? ls0 : last_s;
            wino_static_for([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value, p = j / 6, m = j % 6, kb = m & 1, prod = m >> 1;
                constexpr int sa = prod == 1 ? 1 : 0;      // filter term of the product
                constexpr int sb = prod == 0 ? 1 : 0;      // patch term:  x1 u0, x0 u1, x0 u0
                if constexpr (mode == 0 && prod == 0)
                    acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, uP[(3 * Q4 + p) % W8_RING][kb * 2 + sa]), __builtin_bit_cast(wino_f16x8, Vb[p][sb]), zero16, 0, 0, 0);
                else
                    acc[p * 2 + kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wino_f16x8, uP[(3 * Q4 + p) % W8_RING][kb * 2 + sa]), __builtin_bit_cast(wino_f16x8, Vb[p][sb]), acc[p * 2 + kb], 0, 0, 0);
                if constexpr (m < 4) filter_piece(p + W8_LEAD >= 3 ? qn : q, (p + W8_LEAD) % 3, uP[(3 * Q4 + p + W8_LEAD) % W8_RING], m);
                if constexpr (W8_DMA && c16 == 1 && m >= 4) {       // super-chunk s + 2 into this chunk's OWN stage (nobody reads it any more), a chunk and a half
                    if (!(W8_ELIM & 4))                              // before the barrier that publishes it
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (lds_void*)(lds + par * WINO_SB_FLOATS + (wave * 6 + 2 * p + (m - 4)) * 256), 16,
                                                                 doff_tab[64 * (2 * p + (m - 4))], ls * 128, 0, 0);
                }
#if W8_PARK_LATE
                // (experiment: the three pieces are parked at the END of the chunk after the one that loaded them -- a whole chunk behind their loads)
                if constexpr (p == 2 && m == 3) { stage_write(c16 == 0 ? par ^ 1 : par, 0, c16 == 0 ? 3 : 0); stage_write(c16 == 0 ? par ^ 1 : par, 1, c16 == 0 ? 4 : 1);
                                                  stage_write(c16 == 0 ? par ^ 1 : par, 2, c16 == 0 ? 5 : 2); }
#else
                else if constexpr (m == 4) stage_write(c16 == 0 ? par ^ 1 : par, p, c16 == 0 ? 3 + p : p);
#endif
                if constexpr (p == 2 && m >= 4) {                    // the stage loads (HBM) sit BEHIND the chunk's last filter loads (loads return in order)
                    if constexpr (m == 4) { stage_load(0, ls, 3 * c16); stage_load(1, ls, 3 * c16 + 1); }
                    else stage_load(2, ls, 3 * c16 + 2);
                }
                if constexpr (mode != 0) {
                    constexpr int u0 = j * N_UNITS / N_SLOTS, u1 = (j + 1) * N_UNITS / N_SLOTS;
                    wino_static_for([&](auto K) __attribute__((always_inline)) {
                        unit(integral_constant<int, u0 + decltype(K)::value>{}, integral_constant<int, npar>{}, integral_constant<int, n16>{},
                             integral_constant<bool, mode == 2>{});
                    }, std::make_integer_sequence<int, u1 - u0>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, N_SLOTS>{});
            if constexpr (W8_DMA && c16 == 0) __builtin_amdgcn_s_waitcnt(W8_WAIT_VM12);   // the DMA pieces of the chunk before (12 filter loads are younger) have landed
            else __builtin_amdgcn_s_waitcnt(WINO_WAIT_LGKM0);
            __builtin_amdgcn_s_barrier();
            if constexpr (mode == 0) {
                if (q >= last) return;
                W8_HALF_SERIAL(0, W8_READ, npar, n16, 0);
                W8_HALF_SERIAL(1, W8_READ, npar, n16, 1);
#pragma unroll
                for (int p = 0; p < 3; ++p) split_position(p);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();               // chunk 1 parks pieces in stage 0: every wave must have read its operands out of it
            }
        };
        chunk(integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{}, 0);
        if (nchunk > 1) chunk(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{}, 1);
        for (int base = 0;; base += 4) {
#define W8_CHUNK(t)                                                                                                                \
    if (base + (t) >= nchunk) break;                                                                                               \
    chunk(integral_constant<int, 2>{}, integral_constant<int, (t) & 1>{}, integral_constant<int, ((t) >> 1) & 1>{}, integral_constant<int, (t) & 3>{}, base + (t));
            W8_CHUNK(2) W8_CHUNK(3) W8_CHUNK(4) W8_CHUNK(5)
#undef W8_CHUNK
        }
#undef W8_READ
#undef W8_READ_MINI
#undef W8_READ_2
#undef W8_READ_1
#undef W8_HALF_SERIAL
#undef W8_READS_LANDED
    };
    if (hp == 0) kloop(integral_constant<int, 0>{});
    else kloop(integral_constant<int, 1>{});
    __syncthreads();                                   // every wave is done reading the stages, no DMA in flight: they become the exchange + output staging
    WINO_STAMP(3);

    // ---- output transform Y = At2 M At4^T (k12).  At4 combines the row's SIX positions:  with s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4,
    // d2 = m3 - m4:  Z0 = (m0 + s1) + s2,  Z1 = 2 d2 + d1,  Z2 = 4 s2 + s1,  Z3 = (8 d2 + d1) + m5  -- the row's two wavefronts hold three
    // positions each.  Half 0 finishes channel block kb = 0, half 1 finishes kb = 1: each hands the OTHER block's three intermediates
    // (t = m0 + s1, s1, d1  |  s2, d2, m5: the same operations k12 performs) to its partner through LDS, reads the partner's, and parks
    // Z[a][tile][column][channel] exactly as k12 does.  Exchange X[wave][j = 3 g + v][lane] (f32x4): 8 x 12 KB.
    constexpr int TS = 260;                    // floats per (a, tile)
    f32x4 mine[4][3];                          // this half's intermediates of ITS channel block (kb = hp)
    auto dump = [&](auto hp_t) __attribute__((always_inline)) {
        constexpr int HP = decltype(hp_t)::value;
        float* const xw = lds + wave * 3072 + lane * 4;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 m[3];
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) m[pp] = f32x4{acc[pp * 2 + kb][4 * g], acc[pp * 2 + kb][4 * g + 1], acc[pp * 2 + kb][4 * g + 2], acc[pp * 2 + kb][4 * g + 3]};
                f32x4 v0, v1, v2;
                if constexpr (HP == 0) {
                    const f32x4 s1 = m[1] + m[2];
                    v0 = m[0] + s1; v1 = s1; v2 = m[1] - m[2];              // t, s1, d1
                } else {
                    v0 = m[0] + m[1]; v1 = m[0] - m[1]; v2 = m[2];          // s2, d2, m5
                }
                if (kb == HP) {
                    mine[g][0] = v0; mine[g][1] = v1; mine[g][2] = v2;
                } else {
                    *reinterpret_cast<f32x4*>(xw + (3 * g + 0) * 256) = v0;
                    *reinterpret_cast<f32x4*>(xw + (3 * g + 1) * 256) = v1;
                    *reinterpret_cast<f32x4*>(xw + (3 * g + 2) * 256) = v2;
                }
            }
    };
    if (hp == 0) dump(integral_constant<int, 0>{});
    else dump(integral_constant<int, 1>{});
    __syncthreads();
    f32x4 theirs[4][3];
    {
        const float* const xr = lds + (wave ^ 4) * 3072 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int v = 0; v < 3; ++v) theirs[g][v] = *reinterpret_cast<const f32x4*>(xr + (3 * g + v) * 256);
    }
    __syncthreads();                                   // the exchange area is the staging area
    auto park = [&](auto hp_t) __attribute__((always_inline)) {
        constexpr int HP = decltype(hp_t)::value;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (t, s1, d1) from half 0, (s2, d2, m5) from half 1
            const f32x4 t = HP == 0 ? mine[g][0] : theirs[g][0], s1 = HP == 0 ? mine[g][1] : theirs[g][1], d1 = HP == 0 ? mine[g][2] : theirs[g][2];
            const f32x4 s2 = HP == 0 ? theirs[g][0] : mine[g][0], d2 = HP == 0 ? theirs[g][1] : mine[g][1], m5 = HP == 0 ? theirs[g][2] : mine[g][2];
            float* o = lds + (a * 32 + i32) * TS + HP * 32 + 8 * g + 4 * h;
            *reinterpret_cast<f32x4*>(o) = t + s2;
            *reinterpret_cast<f32x4*>(o + 64) = __builtin_elementwise_fma(f32x4{2.f, 2.f, 2.f, 2.f}, d2, d1);
            *reinterpret_cast<f32x4*>(o + 128) = __builtin_elementwise_fma(f32x4{4.f, 4.f, 4.f, 4.f}, s2, s1);
            *reinterpret_cast<f32x4*>(o + 192) = __builtin_elementwise_fma(f32x4{8.f, 8.f, 8.f, 8.f}, d2, d1) + m5;
        }
    };
    if (hp == 0) park(integral_constant<int, 0>{});
    else park(integral_constant<int, 1>{});
    __syncthreads();
    WINO_STAMP(4);

    // ---- store pass: k12's, with twice the threads (every thread does half of k12's iterations; per-element arithmetic unchanged)
    constexpr int ZA = 32 * TS;                // floats per position row a
    float* const out_base = set_out + (int64_t)blockIdx.y * P.split_out_stride;
    const float inv1 = wino_pow2_inverse(sv) * wino_pow2_inverse(wino_pow2_scale(u_amax, W8_U_TOP));
    const f32x4 inv = f32x4{inv1, inv1, inv1, inv1};
    float lmax = 0.0f;
    if (set_k_planes > 0) {
        // NCHW planes: thread -> (channel, row of the block, 4 pixels along x = one tile's columns); 64-byte runs per (channel, row)
        const int oy = (tid >> 2) & 15, ox = (tid & 3) * 4;
        int m;
        const int gy = cell(y0 + oy, Hv, rHv, m);
        int64_t px0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int n;
            const int gx = cell(x0 + ox + e, Wv, rWv, n), img = m * gcols + n;
            px0[e] = (n < gcols && img < n_img && gx < W && gy < H) ? (out_px + (int64_t)img * HWi) * set_k_planes + (int64_t)gy * W + gx : -1;
        }
        const bool vec = px0[0] >= 0 && px0[3] == px0[0] + 3 && (px0[0] & 3) == 0 && (HWi & 3) == 0;
        const int tile = (oy >> 1) * 4 + (tid & 3);
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int k = it * 8 + (tid >> 6), kg = ks * 64 + k;
            if (kg >= set_k_planes) continue;
            const float bias = set_bias ? set_bias[kg] : 0.0f;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* r = lds + tile * TS + e * 64 + k;
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
        // thread -> 8 consecutive channels (one Philox call) of one pixel column, FOUR rows of one parity (k12: eight)
        const int k8 = (tid & 7) * 8, kg = ks * 64 + k8, ox = (tid >> 3) & 15, odd = (tid >> 7) & 1, grp = tid >> 8;
        f32x4 bias0 = f32x4{0.f, 0.f, 0.f, 0.f}, bias1 = bias0;
        if (set_bias) {
            bias0 = *reinterpret_cast<const f32x4*>(set_bias + kg);
            bias1 = *reinterpret_cast<const f32x4*>(set_bias + kg + 4);
        }
        const uint64_t drop_key = P.thresh ? dropout_key(P.seed, P.epoch) : 0ull;
        int n;
        const int gx = cell(x0 + ox, Wv, rWv, n);
        const bool col_ok = n < gcols && gx < W;
        int m, gy = cell(y0 + odd + 8 * grp, Hv, rHv, m) - 2;                          // canvas row y0 + 8 grp + 2 it + odd: grid row m, image row gy (H: the separator)
        const float* rbase = lds + (ox >> 2) * TS + (ox & 3) * 64 + k8 + (odd ? ZA : 0);
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
            const float* r = rbase + (4 * grp + it) * 4 * TS;                          // tile (4 grp + it, ox >> 2)
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
            if (set_out_amax) {
                const f32x4 a0v = __builtin_elementwise_abs(v0), a1v = __builtin_elementwise_abs(v1);
                const float mx = fmaxf(fmaxf(fmaxf(a0v.x, a0v.y), fmaxf(a0v.z, a0v.w)), fmaxf(fmaxf(a1v.x, a1v.y), fmaxf(a1v.z, a1v.w)));
                lmax = fmaxf(lmax, P.thresh ? mx * P.scale : mx);
            }
            if (set_replicas > 0) {                                  // the first conv of an MC-dropout subnet: the runs' masked replicas (k12)
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
            if (P.thresh) {
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
    if (set_out_amax) wino_publish_amax_block(set_out_amax, lmax);
#ifdef POD_TRACE
    __builtin_amdgcn_s_waitcnt(0);                      // the stores have left
    WINO_STAMP(5);
    WINO_STAMP_WALL(13);
#endif
}

static int wino_split8_prepare() {        // the kernel's dynamic LDS size, once per device
    static std::once_flag once[64];
    static hipError_t attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return POD_E_LAUNCH;
    std::call_once(once[dev], [dev] {
        attr[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wino_conv3x3_split8), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS_BYTES);
    });
    return attr[dev] == hipSuccess ? POD_OK : POD_E_LAUNCH;
}

// called by pod_wino_conv3x3_split (k12) with the parameter block it validated and filled
int wino_split8_launch(const WinoParams& P, int64_t grid, unsigned grid_y, hipStream_t stream) {
    if (wino_split8_prepare() != POD_OK) return POD_E_LAUNCH;
    hipLaunchKernelGGL(k_wino_conv3x3_split8, dim3((unsigned)grid, grid_y), dim3(512), W8_LDS_BYTES, stream, P);
    POD_CHECK_LAUNCH();
    return POD_OK;
}

}  // namespace pod

#ifdef POD_TRACE
extern "C" int pod_wino_trace_dump_split8(long long* host, int32_t n_workgroups) {   // diagnostics build only (tools/wino_trace.py)
    if (hipDeviceSynchronize() != hipSuccess) return POD_E_LAUNCH;
    if (n_workgroups > 8192) n_workgroups = 8192;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(pod::g_wino_trace), (size_t)n_workgroups * 16 * sizeof(long long)) != hipSuccess) return POD_E_LAUNCH;
    return POD_OK;
}
#endif
