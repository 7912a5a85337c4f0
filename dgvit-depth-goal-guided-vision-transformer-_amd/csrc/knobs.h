// A/B and diagnostic knobs of the kernels and schedules.
//
// Product build (libdgvit_hip.so): every knob is a compile-time constant with its shipped default -- the library has NO mutable
// process-global state besides one-time initialisation (helper stream, per-device kernel attributes): nothing a second host
// thread (autograd's backward thread) could flip in the middle of a step, and the code of the switched-off alternatives
// (clock stamps, timing diagnostics, the pipelined persistent GEMM, the per-frame inference path, the non-default MFMA shape)
// is not in the binary.  The two schedule options a caller may legitimately want per call travel in dgvit_config.flags.
//
// Diagnostic build (-DDGVIT_DIAG -> libdgvit_hip_diag.so, for tools/ and the A/B equality tests): the same names are plain
// process-global variables, set through the entry points of include/dgvit_hip_diag.h.  Not thread-safe by design.
#pragma once

#ifdef DGVIT_DIAG
#define DGVIT_KNOB(type, name, def) extern type name;
#define DGVIT_DIAG_ONLY(...) __VA_ARGS__
#define KNOB_IF(k) if (k)
#else
#define DGVIT_KNOB(type, name, def) static constexpr type name = def;
#define DGVIT_DIAG_ONLY(...)
#define KNOB_IF(k) if constexpr ((k) != 0)   // inside templates the untaken branch is not even instantiated
#endif

DGVIT_KNOB(int, g_gemm_tile_hint, 0)            // fp32 GEMM workgroup tile (0 = automatic; BM*1000000 + BN*1000 + BK)
DGVIT_KNOB(int, g_gemm_split, 1)                // in-launch split-K of the forward / data-gradient GEMMs
DGVIT_KNOB(int, g_gemm_lds_pad, 0)              // extra dynamic LDS bytes per fp32 GEMM workgroup (occupancy probes)
DGVIT_KNOB(int, g_gemm_persist, 0)              // pipelined persistent fp32 GEMM: 0 never, 1 when a slot gets several tiles, 2 whenever eligible
DGVIT_KNOB(int, g_gemm_persist_grid, 0)         // its grid (0 = resident slots)
DGVIT_KNOB(int, g_gemm_diag, 0)                 // timing diagnostics of the per-tile fp32 GEMM (bits: dgvit_hip_diag.h)
DGVIT_KNOB(long long*, g_gemm_stamps, nullptr)  // per-workgroup clock stamps of the fp32 GEMM
DGVIT_KNOB(int, g_gemm_stamp_capacity, 0)
DGVIT_KNOB(int, g_group_reduce, 1)              // one grouped slab / partial reduction per layer
DGVIT_KNOB(int, g_ln_fusion, 1)                 // dim 64: LayerNorms inside the producing GEMM's epilogue
DGVIT_KNOB(int, g_conv_gather, 1)               // conv2 / conv3 forward as implicit GEMMs
DGVIT_KNOB(int, g_gelu_grad_store, 1)          // training forward stores gelu'(pre-activation) for the backward (0: the pre-activation, erf in the backward epilogue)
DGVIT_KNOB(int, g_block_path, 1)                // small no-grad batches: two launches per block with in-launch combines (block.hip)
DGVIT_KNOB(int, g_block_fuse, 2)                // ... bit 1 (on): the last MLP kernel applies the final RMSNorm; bit 0 (off: measured +14 us for one frame, -3 us for two): block 0 assembles its token rows
DGVIT_KNOB(int, g_block_path_max_rows, 4160)    // ... up to this many token rows (64 frames of 65 tokens)
DGVIT_KNOB(long long*, g_block_stamps, nullptr) // diagnostic: phase stamps of the two block kernels (32 int64)
DGVIT_KNOB(int, g_block_stamp_layer, -1)        // ... of this transformer block only (-1: every block writes, the last one wins)
DGVIT_KNOB(int, g_block_stamp_now, 1)           // (set by the schedule: is the block being launched the stamped one)
DGVIT_KNOB(int, g_small_path, 0)                // per-frame two-launch inference path (frame.hip; measured slower)
DGVIT_KNOB(int, g_small_path_max_rows, 4160)
DGVIT_KNOB(int, g_gemm_bf16_tile_hint, 0)       // bf16 GEMM tile (0 = automatic)
DGVIT_KNOB(int, g_gemm_bf16_m16, 1)             // ring GEMM on v_mfma_f32_16x16x32_bf16 (0: 32x32x16)
DGVIT_KNOB(int, g_gemm_bf16_group_m, 8)         // row panels per walk group of the persistent tile order
DGVIT_KNOB(int, g_gemm_bf16_l2_budget_kb, 2048)  // stream GEMM: L2 bytes a column block's B panels may take (0 = the round-3 group_m walk)
DGVIT_KNOB(long long*, g_gemm_bf16_stamps, nullptr)
DGVIT_KNOB(int, g_attn_bwd64, 1)                // single-pass fp32 attention backward for 32 < N <= 64
DGVIT_KNOB(int, g_gemm_zfold, 1)                // weight-gradient GEMMs: k-slices folded into blockIdx.x, k-slice major per XCD (0: grid z)
DGVIT_KNOB(int, g_attn_q1, 1)                   // one-query (token 0) fp32 attention forward / backward on plain FMAs for N <= 64
#ifdef DGVIT_DIAG
extern long long g_gemm_persist_launches;       // launches that took the pipelined kernel
#endif
