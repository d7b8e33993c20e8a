// Experiment knobs of the kernels, in ONE place (round 6; they used to be #if blocks scattered through k11 / k12 / k13 / k1f).
// Every knob is a CONSTANT of the shipped library: pod_compare_amd/build.py refuses -D defines on untagged builds, so what is compiled here is
// always the default column below; tagged builds (POD_BUILD_TAG=<name> POD_EXTRA_DEFINES="-DPOD_WINO_ELIM=6 ...") live in lib/<name>/ beside it
// and exist to PRICE an ingredient -- their results are wrong by construction, only their time is read (tools/wino_elim12.sh,
// tools/conv1x1_elim.py, tools/k1f_variants.sh; the tables: profiles/r05_k12_elimination.txt, r05_conv1x1_trace.txt, r05_k1f_variants.txt).
// The kernels use the knobs as ordinary constant expressions (`if (POD_WINO_ELIM & 2) ...`): no code is hidden behind the preprocessor.
// Knobs whose verdict is final were deleted with their code (docs/KERNEL_NOTEBOOK.md has the measurements): POD_WINO_XFORM_PINS /
// POD_WINO_SPLIT_PINS (pinning the slotted arithmetic: no effect / +65 s_nop per chunk), POD_WINO_VAR (patch-source variants), POD_WINO_U_LEAD
// (filter loads 2 or 3 positions ahead: no difference; 3 shipped), POD_WINO_DEBUG_X (round 2's patch dump).
#pragma once

// k11 / k12 (pod_wino_conv3x3[_split]): bits compiled OUT -- 1 patch reads, 2 filter loads, 4 patch fill, 8 input transform (+ split), 16 chunk
// barrier (k11), 32 store pass, 64 dropout mask, 128 accumulator dump + store pass
#ifndef POD_WINO_ELIM
#define POD_WINO_ELIM 0
#endif
// k13 (pod_conv1x1_split): bits compiled out -- 1 activation loads, 2 filter loads, 4 split arithmetic, 8 stores
#ifndef POD_C1_ELIM
#define POD_C1_ELIM 0
#endif
#ifndef POD_C1_RING
#define POD_C1_RING 3        // register ring of the direct-fragment kernel (k-steps in flight + 1); 4 / 5 / 6 measured: no difference
#endif
#ifndef POD_C1_DIRECT
#define POD_C1_DIRECT 0      // 1: the direct-fragment kernel everywhere (tools/conv1x1_ab.py compares it with the LDS form: identical sha-256)
#endif
// k1f (pod_merge_score_fused): launch geometry (profiles/r05_k1f_variants.txt: the defaults are the fastest of the twelve measured)
#ifndef POD_K1F_WAVES
#define POD_K1F_WAVES 4      // wavefronts per workgroup (each streams its own, distant, chunk; all of them score the parked cells)
#endif
#ifndef POD_K1F_WPE
#define POD_K1F_WPE 3        // wavefronts per SIMD the register allocation aims at: 12 per CU = all 3 060 wavefronts of a BASELINE launch resident
#endif
#ifndef POD_K1F_BATCH
#define POD_K1F_BATCH 2      // runs whose loads are in flight together (CPL < 4): 2 x 2K loads per lane
#endif
#ifndef POD_K1F_NT
#define POD_K1F_NT 1         // non-temporal loads (the runs are read exactly once)
#endif
#ifndef POD_K1F_CELLS
#define POD_K1F_CELLS 1      // consecutive cells of a plane per lane (1, 2 or 4: 4-, 8- or 16-byte loads)
#endif
#ifndef POD_K1F_ADJ
#define POD_K1F_ADJ 0        // 1: the wavefronts of a workgroup stream ADJACENT chunks (measured: no difference)
#endif
#ifndef POD_K1F_NOSCORE
#define POD_K1F_NOSCORE 0    // 1: the streaming part alone (prices the scoring tail: 19.5 of 23 us)
#endif
