/* libdgvit_hip_diag.so -- the diagnostic build of the DGViT library (tools/, A/B equality tests).
 *
 * Built from the same sources as libdgvit_hip.so with -DDGVIT_DIAG.  It exports everything include/dgvit_hip.h declares plus the
 * knobs below, which are plain process-global variables (NOT thread-safe: set them from the thread that makes the calls, between
 * steps), and it contains the code those knobs switch to: clock stamps and timing diagnostics inside the GEMM kernels, the
 * pipelined persistent fp32 GEMM, the per-frame inference path, the 32x32x16 MFMA form of the bf16 ring GEMM.  The product
 * library has none of this: there every knob is a compile-time constant with the default named here.
 */
#ifndef DGVIT_HIP_DIAG_H
#define DGVIT_HIP_DIAG_H
#include "dgvit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* test/bench knob: force the GEMM workgroup tile (0 = automatic; BM*1000000 + BN*1000 + BK, e.g. 128128032) */
void dgvit_set_gemm_tile(int tile);
/* A/B knob (default 1): dgvit_got_backward sums a layer's split-K weight-gradient slabs and LayerNorm partials in ONE grouped
 * launch per layer; 0 = one reduction launch behind every producer (the round-1 schedule).  Results are bit-identical. */
void dgvit_set_grouped_reduce(int on);
/* A/B knob (default 1): forward / data-gradient GEMMs with far fewer output tiles than the chip has workgroup slots, or with a
 * nearly empty last round of tiles, are split over K inside the launch (partial tiles + last-arriver epilogue, deterministic).
 * 0 = one workgroup per output tile.  Results agree to fp32 rounding (the order of the k-sum changes). */
void dgvit_set_gemm_split(int on);
/* A/B knob: 1 (default) for dim == 64 the encoder forward runs its LayerNorms inside the epilogues of the GEMMs that produce their
 * inputs (to_out -> LN2, fc2 -> the next block's LN1); 0 = separate LayerNorm launches.  Bit-identical results. */
void dgvit_set_ln_fusion(int on);
/* A/B knob: 1 (default) dgvit_cnn_forward runs conv2 / conv3 as implicit GEMMs (5x5xC windows gathered by the GEMM's A-tile loader);
 * 0 = im2col + GEMM.  Bit-identical results. */
void dgvit_set_conv_gather(int on);
/* diagnostic: request `bytes` more dynamic LDS per fp32 GEMM workgroup than it uses (caps the workgroups per CU: occupancy probes) */
void dgvit_set_gemm_lds_pad(int bytes);
/* Diagnostics of the per-tile fp32 GEMM (tools only; default 0).  Bit 0: A/B knob, raise the wave priority (s_setprio 2) of the main
 * loop over the prologue / epilogue waves on the same SIMD (measured: no effect).  Bits 1 and 2 are TIMING diagnostics whose results are
 * garbage: bit 1 - the kernel returns after the main loop without writing C; bit 2 - every tile stores over tile 0 (the same epilogue
 * instructions and side reads, no write stream to HBM; LDS-image epilogue only).  Bit 3: A/B knob, use the LDS-image epilogue where the
 * direct (register) epilogue would be taken.  DESIGN.md 3.9 uses them to take the epilogue's cost apart. */
void dgvit_set_gemm_diagnostics(int bits);
/* The pipelined persistent fp32 GEMM (one k-tile stream per workgroup across its tiles, a tile's stores under the next tile's main
 * loop; NT / NN forms, 16-byte-aligned operands, K = 16 k-tiles of the chosen tile: 256 at 16-deep, 512 at 32-deep k-tiles).
 * mode 0 = never, 1 = when a resident workgroup slot gets at least two tiles and no tile is split, 2 = whenever the launch is
 * eligible.  workgroups > 0 overrides the grid (diagnostic; 0 = automatic).  Same results bit for bit as the per-tile kernel: the
 * k order of a tile does not change. */
void dgvit_set_gemm_persistent(int mode, int workgroups);
/* launches that took the pipelined kernel since the library was loaded (tests check that they exercise it) */
long long dgvit_gemm_persistent_launches(void);
/* diagnostic (tools/gemm_stamps.py): non-NULL = every fp32 GEMM launch writes 16 int64 per workgroup (< `workgroups`) into the
 * device buffer: [0..3] shader clock at kernel start / first k-tile in LDS / main loop done / stores issued, [7] stores drained (the
 * stamped run waits for them), [4] and [6] the 100 MHz counter at start and end, [5] HW_ID | XCC_ID << 32, [8 + 2c] / [9 + 2c] epilogue chunk c: C image in
 * LDS / stores issued.  NULL (default) = off; the product never sets it. */
void dgvit_set_gemm_stamps(long long* stamps, int workgroups);
/* Opt-in experiment (default OFF): dgvit_got_forward with save_for_backward == 0 and at most max_rows token rows (default
 * 4160 = 64 frames of 65 tokens) runs every transformer block as TWO launches (one workgroup per frame and head; one per
 * frame and 128-wide hidden chunk) instead of seven GEMM / LayerNorm / attention launches -- aimed at SAC.choose_action
 * (DRL.py:170-185).  Correct (parity-tested) but measured slower than the split-K GEMM schedule on MI355X: each workgroup
 * walks five dependent phases of L2 round trips, see DESIGN.md 3.7. */
void dgvit_set_small_batch_path(int on, int max_rows);
/* A/B knob: no-grad forwards of a few frames (at most max_rows token rows, default 4160, AND few enough workgroups to sit one per CU:
 * dgvit_api.hip, block_path_eligible) run every transformer block as TWO launches with the sums over heads / hidden chunks taken inside
 * them (block.hip; on = 1, the product's behaviour).  on = 0: the seven-launch GEMM schedule for every size; on = 2: the fused blocks for
 * every shape they support, whether or not they win there (tests).  Set it before the workspace
 * query of the call it should affect (the query sizes the combine scratch). */
void dgvit_set_block_path(int on, int max_rows);
/* diagnostic (tools/block_stamps.py): non-NULL = thread 0 of workgroup 0 of the two small-batch block kernels writes the 100 MHz wall clock
 * at its phase boundaries into this device buffer of 32 int64 (attention kernel at [0..7], MLP kernel at [16..22]); NULL (default) = off. */
void dgvit_set_block_stamps(long long* stamps);
/* ... of transformer block `layer` only (default -1: every block writes its stamps, the last block's survive) */
void dgvit_set_block_stamp_layer(int layer);
/* A/B knob: 1 (default, the product) the training forward's fc1 epilogue stores gelu'(pre-activation) in the slot the backward reads and the
 * data gradient of fc2 multiplies by it; 0: the round-3 form (pre-activation stored, erf and exp evaluated in the backward epilogue).
 * Same values bit for bit; a forward and its backward must run under the same setting. */
void dgvit_set_gelu_grad_store(int on);
/* A/B knob of the small-batch blocks: bit 0 (default OFF: measured slower for one frame) block 0's attention kernel assembles its token rows
 * (goal row, emb-dropout, counter zeroing: three launches fewer), bit 1 (default on) the last MLP kernel applies the final RMSNorm. */
void dgvit_set_block_fuse(int bits);
/* test/bench knob: force the bf16 GEMM workgroup tile (0 = automatic; 256256, 256128, 128128; 256254 = probe: per-tile kernel with
 * 4 waves of 128 x 128, profiles/r02_e_bf16_gemm_4wave_128x128_probe.txt) */
void dgvit_set_gemm_bf16_tile(int tile);
/* test/bench knob: row panels per walk group of the persistent bf16 GEMM's tile order (default 8) */
void dgvit_set_gemm_bf16_group_m(int rows);
/* A/B knob: L2 bytes (KB) that the B panels of one column block of the stream GEMM's tile walk may take (default 2048; the launcher
 * cuts the tile grid into as few column blocks as fit); 0 = the round-3 walk (groups of group_m row panels, column by column) */
void dgvit_set_gemm_bf16_l2_budget_kb(int kb);
/* A/B knob: 1 (default) the single-pass fp32 attention backward for 32 < N <= 64 (every tile pair computed once); 0 the two-phase
 * kernel for every shape.  Same results up to summation order. */
void dgvit_set_attention_bwd_single_pass(int on);
/* 0: a one-query attention (the last block's token 0) runs on the MFMA tile kernels as before round 4; 1 (default): attn_q1_*_kernel */
void dgvit_set_attention_single_query(int on);
/* A/B knob: 1 (default) the weight-gradient GEMMs deal (tile, k-slice) pairs to the XCDs k-slice major (an XCD reads its slices of dY and X
 * once); 0 the round 1-3 grid (tiles, 1, slices): an XCD owns a few tiles and all their slices.  Bit-identical results. */
void dgvit_set_gemm_wgrad_slice_major(int on);
/* dgvit_attention_forward / _backward (dgvit_hip.h) for the first nq query tokens only, as the encoder's last block calls them with nq = 1
 * (GoalFormer.py:167 reads x[:, 0]): rows >= nq of out / lse / dq are not written, dk and dv cover every key */
int dgvit_attention_forward_queries(const float* qkv, float* out, float* lse, int B, int N, int H, int dh, int nq, void* stream);
int dgvit_attention_backward_queries(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int B, int N, int H,
                                     int dh, int nq, void* stream);
/* A/B knob: which MFMA the ring GEMM issues: 1 (default) v_mfma_f32_16x16x32_bf16, 0 v_mfma_f32_32x32x16_bf16 (same cycles per
 * FLOP; the kernel runs under the chip's power limit and the 16x16 shape measured 2-3 % faster; same results up to summation order) */
void dgvit_set_gemm_bf16_mfma16(int on);
/* diagnostic (tools/bf16_stamps.py): non-NULL = the epilogue-0 ring GEMM runs its stamped build and writes, per workgroup,
 * 2 wave groups x 8 tiles x 4 int64 {s_memtime at tile start / after its main loop / after its epilogue, s_memrealtime}
 * to this device buffer (256 workgroups at most); NULL (default) = shipped kernels, no stamp executes. */
void dgvit_set_gemm_bf16_stamps(long long* stamps);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DGVIT_HIP_DIAG_H */
