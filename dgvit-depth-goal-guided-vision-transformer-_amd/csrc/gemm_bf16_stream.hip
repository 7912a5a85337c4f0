// bf16 GEMM for gfx950, "stream" kernel:  C[m][n] = sum_k A[m][k] * B[n][k]  (+ bias, erf-GELU), A / B bf16 with k contiguous,
// fp32 accumulate on v_mfma_f32_16x16x32_bf16.  The Linear layers of Attention and FeedForward (GoalFormer.py:42-50,64,66-69) in
// the bf16 configuration (BASELINE config 5), forward and data-gradient forms.
//
// What is different from gemm_bf16_ring_kernel (gemm_bf16.hip), and why (measured, tools/native/dma_probe.hip):
//  * FULL 128-BYTE LINES PER LDS-DMA INSTRUCTION.  A `buffer_load_dwordx4 ... lds` whose 64 lanes cover 16 rows x 64 bytes (a 32-deep
//    k-tile) tops out at 53-57 GB/s per CU whatever the tile, the ring depth or the cache state -- 1.75 PFLOP/s for a 256 x 256 tile
//    and the ring kernel already runs at 70 % of that.  8 rows x 128 bytes (a 64-deep k-tile) delivers 85-90 GB/s per CU.
//    A k-tile is therefore 64 deep: two PIECES of 256 rows x 128 bytes (32 KB), one of A and one of B, in a ring of five
//    (all 160 KB of LDS: the epilogue needs none).  Rows are swizzled by 16-byte chunk c -> c ^ ((row >> 1) & 7) on the SOURCE
//    address and again by the fragment reads (conflict-free for the 16x16x32 operand pattern of ds_read_b128).
//  * NO PING-PONG.  With 64 KB per k-tile the LDS holds 2.5 k-tiles; a ping-pong pair keeps a k-tile "being read" for two and a half
//    load / compute intervals and leaves a piece ~1 interval to arrive.  Here all eight waves run the same software pipeline: the
//    fragments of the NEXT 32-deep half are read into a second register set while the MFMAs of the current half run, so a k-tile
//    is read for exactly one k-tile period, ONE barrier per 64-deep k-tile hands its slots to the DMA stream, and a piece has
//    1 (B, the L2-resident weights) to 2.5 (A) periods of ~2 k cycles to land.
//  * THE DMA ISSUE IS SPREAD over the MFMAs (one instruction per eight MFMAs): issued as a burst of 4 x 8 waves the memory pipeline's
//    queue backs up and the issuing waves stall in front of it.
//  * ONE WAVE OF EVERY SIMD PAIR ISSUES THE DMAs OF BOTH (round 4).  Waves w and w + 4 share a SIMD; per-wave clock stamps showed the pair
//    out of balance: from the barrier on, waves 0-3 ran their second half in ~970 cycles and waves 4-7 in ~1540, so waves 0-3 sat ~870
//    cycles per k-tile at the next barrier while their partners worked alone, too slowly to keep the MFMA pipe busy (2.7 k cycles per k-tile
//    for 2.05 k of matrix work).  The matrix work cannot move between waves (the accumulators are theirs), the DMA issue can: waves 0-3
//    issue all eight instructions per piece (their own rows and their partner's, 32 rows / four 1 KB chunks away), waves 4-7 none.
//    Stamps after: 1040 + 1010 busy and 460 waiting (waves 0-3) against 1320 + 1060 + 145 (waves 4-7), 2.52 k cycles per k-tile; launches at
//    batch 440: fc2 332 -> 308-316 us, QKV 264-287 -> 254-262, fc1 397-402 -> 394-396 (bit-identical results; profiles/r04_d_stream_dma_sharing.txt).
//    The opposite assignment (waves 4-7 issue) loses; a priority window for waves 4-7 on top is worth at most another 1.5 % on fc1 only and
//    stays a timing variant.  Since a wave's epilogue staging chunks are now overwritten by ANOTHER wave's DMAs, one more barrier per output
//    tile separates the epilogue from the next first half (measured free).
//  * DIRECT EPILOGUE.  The MFMA operands are swapped (weights as the A operand), so a lane's four accumulator registers of a
//    16 x 16 block are four CONSECUTIVE output columns of one row: bias / GELU / bf16 conversion and one 8-byte (bf16) or
//    16-byte (fp32) buffer store straight from registers.  No LDS staging, no wave barriers, no VMEM loads: the bias comes in by
//    scalar loads (s_buffer_load, range-checked), because an ordinary load makes hipcc wait vmcnt(0) while LDS-DMAs are in
//    flight (cdna_hip_programming.md, "Pipelining across barriers") -- a whole ring of DMA latency per tile.
//  * Persistent as before: one workgroup per CU walks tiles id, id + grid, ...; the k-tile stream runs on across tile boundaries.
// vmcnt bookkeeping (loads, LDS-DMAs and stores retire in order on one counter): piece j is issued in the half-phase that
// precedes... see the loop; every wait of an issuing wave is `vmcnt(8)` (the piece issued during the current phase may stay in flight),
// so the stores of an epilogue simply have to be older than the next barrier's wait, which they are by a whole k-tile period; a wave
// that issues no DMAs waits for none (its partner's counted wait followed by the barrier orders its rows).
#include "bf16.h"
#include "kernels.h"

#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
constexpr unsigned OOB = 0x80000000u;
constexpr int PIECE = 256 * 128;   // bytes of one piece: 256 rows x 64 k
constexpr int NSLOT = 5;

__device__ __forceinline__ int xcd_chunk(int id, int n) {
  // blocks are dealt round-robin over the 8 XCDs: give each XCD one contiguous chunk of the tile grid (bijective)
  const int q = n >> 3, r = n & 7, xcd = id & 7, loc = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// s_waitcnt lgkmcnt(0) as the BUILTIN (simm16: vmcnt 63, expcnt 7, lgkmcnt 0): hipcc's own wait insertion sees it, so it does not put
// conservative `lgkmcnt(8)` waits in front of the next half's MFMAs for fragments this wait has already covered (an inline-asm wait is
// invisible to its scoreboard; with the LDS busy taking DMA data those spurious waits stalled the MFMA stream).
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Issue order of one half (one basic block: 12 fragment reads of the NEXT half, 32 MFMAs, 4 LDS-DMAs): the four B fragments first,
// then per A row block its 4 MFMAs followed by the read of the block's next fragment, and one DMA after every second block, so that
// the 32 DMA instructions a CU issues per half arrive at the memory pipeline evenly (~1 per 32 cycles; it takes ~1 per 24): issued
// as a burst at the top of the half they back up its queue and the waves stall in front of it before their first MFMA.
// (LLVM SchedGroupMask: MFMA 0x8, VMEM 0x10, DS read 0x100.)
template <int VAR, int PAR, int LEAD = 0>
__device__ __forceinline__ void sched_half() {
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  if ((VAR == 6 || VAR == 7) && LEAD > 0) {   // fragment reads LEAD row blocks ahead of the matrix work: the half's last read is issued with
                                             // 4 * LEAD MFMAs still to come instead of behind the last one (timing variants 1048576 / 2097152)
                                             // -- measured 2-4 % SLOWER on fc1 / fc2 (profiles/r04_d_stream_dma_sharing.txt, run 4): bunched reads
                                             // collide with the DMA writes in the LDS; the one-read-per-block spread stays
    __builtin_amdgcn_sched_group_barrier(0x100, LEAD, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      if (i < 8 - LEAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (VAR == 6) __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
    return;
  }
  if (VAR == 3) {          // leading wave of a SIMD pair: matrix work first (fragment reads between the blocks), the four DMAs at the end
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x10, 4, 0);
    return;
  }
  if (VAR == 5) {          // every fragment read of the half up front, the DMAs later, among matrix instructions only
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    return;
  }
  if (VAR == 6 || VAR == 7) {          // DMA sharing: the issuing wave one DMA per row block (eight per half), its partner none
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (VAR == 6) __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
    return;
  }
  if (VAR == 4) {          // its partner: DMAs and fragment reads first, matrix work behind them
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x8, 32, 0);
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (VAR == 1 && (i & 1) == PAR) {   // the DMA in the middle of the block's MFMAs, away from the fragment reads
      __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      if (VAR != 1 && (i & 1) == PAR) __builtin_amdgcn_sched_group_barrier(0x10, 1, 0);
    }
  }
}

// Timing diagnostics (diagnostic build only, dgvit_set_gemm_diagnostics; results are garbage): bit 0 (1) every piece re-fetches k-tile 0
// of its tile (cache-hot source), bit 1 (2) no LDS-DMA at all, bit 2 (4) no fragment reads inside the loop, bit 3 (8) no epilogue,
// bit 4 (16) no barrier, bit 5 (32) no MFMAs, bit 6 (64, with 8) the accumulators stay alive without an epilogue; 32768 every wave issues
// its own DMAs (the round-3 schedule), 65536 waves 4-7 issue them all, 131072 / 262144 priority window for waves 4-7 (exact results),
// 524288 no barrier after the epilogue (racy).
// (compile-time variants: a run-time test around every MFMA wrecks the very schedule being measured)
#define SDIAG(b) ((DIAG & (b)) != 0)
constexpr int dma_sharing(int DIAG) { return SDIAG(65536) ? 2 : (SDIAG(32768) || SDIAG(256) || SDIAG(128) || SDIAG(4096)) ? 0 : 1; }

template <int EPI, int DIAG, int ROLE>     // ROLE: 0 every wave the same schedule; 1 / 2 leading / trailing wave of a SIMD pair (timing variant 256)
__device__ __forceinline__ void stream_body(const GemmBf16Params& p, int ntiles) {
  // DMA sharing (see the file header): 1 waves 0-3 issue the LDS-DMAs of their SIMD partners too (the shipped schedule), 2 waves 4-7 do
  // (timing variant 65536), 0 every wave issues its own (round 3; timing variant 32768 and the variants that bring their own half schedule)
  constexpr int DSH = dma_sharing(DIAG);
  constexpr bool ISSUER = DSH != 0 && ROLE == DSH;
  constexpr int NDMA = DSH == 0 ? 4 : ISSUER ? 8 : 0;            // piece DMAs this wave issues per half
  // ... and the partner runs the first PK row blocks of every SECOND half (the one that starts at the barrier, where the older wave of the
  // pair otherwise wins every issue slot) at priority 1 (timing variants 131072 / 262144 on top: PK = 2 / 4 / 6)
  constexpr int PK = DSH != 0 && !ISSUER ? 2 * ((DIAG >> 17) & 3) : 0;
  constexpr int LEAD = SDIAG(1048576) ? 2 : SDIAG(2097152) ? 4 : 0;
  constexpr int HVAR = DSH ? (ISSUER ? 6 : 7) : ROLE == 1 ? 3 : ROLE == 2 ? 4 : SDIAG(4096) ? 5 : SDIAG(128) ? 1 : 0;
  constexpr bool OUT_F32 = EPI == BEPI_F32_PLAIN;
  // Output stores are non-temporal (aux bit 1): C is never read again by this launch and is larger than the L2s, so letting it allocate there
  // only evicts the A / B k-slices that the neighbouring tiles are about to re-read (measured at B = 440: QKV 313 -> 278 us, fc1 420 -> 394 us,
  // fc2 -- 3 column tiles, few stores -- unchanged; timing variant 2048 restores ordinary stores)
  constexpr int ST_AUX = SDIAG(2048) ? 0 : 2;
  constexpr int ES = OUT_F32 ? 4 : 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD
  const int wr = wave >> 2, wc = wave & 3;          // wave tile: rows 128 wr .. + 127, columns 64 wc .. + 63
  const int l15 = lane & 15, q = lane >> 4;
  const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + 255) / 256;
  // Tile order (the XCDs take contiguous chunks of it, xcd_chunk): COLUMN BLOCKS of the tile grid, each walked row panel by row panel
  // with the block's columns innermost.  The 32 workgroups of an XCD then work on (32 / width) row panels x the block's `width` column
  // tiles at a time: the block's B panels (width x 256 x K bf16, sized by the launcher to stay well inside the 4 MB L2) are re-used by
  // every row panel that follows, and an A panel is fetched once per column block and shared by the block's columns as it streams
  // through.  (Round 3 walked groups of 8 row panels column by column: 8 A panels + 4 B panels of a K = 768 GEMM are 4.7 MB, nothing
  // survived from one round of tiles to the next, 1.7 x the algorithmic bytes went past the L2s.)  p.col_blocks == 0 keeps that older
  // walk (groups of group_m row panels) for A/B in the diagnostic build.
  auto tile_mn = [&](int t, int& m0, int& n0) {
    if (p.col_blocks > 0) {
      const int nb = p.col_blocks, wq = tiles_n / nb, wrem = tiles_n - wq * nb;      // the first wrem blocks are wq + 1 columns wide
      const int big = tiles_m * (wq + 1);
      int w, c0, rem;
      if (t < wrem * big) {
        const int b = t / big;
        rem = t - b * big, w = wq + 1, c0 = b * (wq + 1);
      } else {
        const int t2 = t - wrem * big, small = tiles_m * wq, b = t2 / small;
        rem = t2 - b * small, w = wq, c0 = wrem * (wq + 1) + b * wq;
      }
      const int r = rem / w;
      m0 = r * 256;
      n0 = (c0 + rem - r * w) * 256;
      return;
    }
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * tiles_n, grp_i = t / per_group, within = t - grp_i * per_group;
    const int rows = tiles_m - grp_i * GROUP_M < GROUP_M ? tiles_m - grp_i * GROUP_M : GROUP_M;
    m0 = (grp_i * GROUP_M + within % rows) * 256;
    n0 = (within / rows) * 256;
  };
  const int nkt = (p.K + 63) / 64;                  // 64-deep k-tiles per output tile

  // ---- load side: the stream of pieces A(0) B(0) A(1) B(1) ... of all this workgroup's tiles, piece j -> slot j % 5 -------------
  // instruction i (0..3) of a wave fills rows 64 i + 8 wave + (lane >> 3) of the piece; the lane's physical 16-byte chunk lane & 7
  // holds logical chunk (lane & 7) ^ ((row >> 1) & 7), and (row >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7 for every i
  const int lrow = 8 * wave + (lane >> 3);
  const int lchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const unsigned offA = ((unsigned)lrow * (unsigned)p.lda + lchunk * 8u) * 2u, stepA = 64u * (unsigned)p.lda * 2u;
  const unsigned offB = ((unsigned)lrow * (unsigned)p.ldb + lchunk * 8u) * 2u, stepB = 64u * (unsigned)p.ldb * 2u;
  int ltile = blockIdx.x, lt = 0, lslot = 0, lop = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_load_tile = [&](int v) {
    int m0, n0;
    tile_mn(xcd_chunk(v, ntiles), m0, n0);
    long long abytes = ((long long)(p.M - 1 - m0) * p.lda + p.K) * 2, bbytes = ((long long)(p.N - 1 - n0) * p.ldb + p.K) * 2;
    if (abytes > 0x7FFFFFF0ll) abytes = 0x7FFFFFF0ll;
    if (bbytes > 0x7FFFFFF0ll) bbytes = 0x7FFFFFF0ll;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A + (long long)m0 * p.lda), 0, (int)abytes, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.B + (long long)n0 * p.ldb), 0, (int)bbytes, 0x00020000);
  };
  set_load_tile(ltile);   // blockIdx.x < ntiles by construction of the grid
  // Instruction i (0..3) of the A / B piece of the load side's current k-tile; an A piece is issued during the first half of a k-tile's
  // MFMAs, a B piece during the second (the pieces alternate A, B like the halves), then the state advances.  Straight-line code, so
  // that a whole half (fragment reads, 32 MFMAs, 4 DMAs) is ONE basic block whose issue order sched_group_barrier can pin.
  // Past the last tile the same instructions still issue (every lane out of range: zeros into a free slot): the vmcnt bookkeeping
  // never changes.
  unsigned ldead = lchunk * 8 < p.K ? 0u : OOB;
  // (`other`: the rows of the SIMD partner, wave ^ 4 -- 32 rows and four 1 KB chunks away, same swizzle since (wave & 1) is the same)
  const int pw = wave < 4 ? 4 : -4;
  auto dma_a = [&](int i, int other = 0) {
    if constexpr (!SDIAG(2))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(smem + lslot * PIECE + (wave + other * pw) * 1024 + i * 8192), 16,
                                               ((offA + (SDIAG(1) ? 0u : (unsigned)lt * 128u)) | ldead) + i * stepA + (unsigned)(other * pw * 16 * p.lda), 0, 0,
                                               SDIAG(8192) ? 2 : 0);
  };
  auto dma_b = [&](int i, int other = 0) {
    if constexpr (!SDIAG(2))
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)(smem + lslot * PIECE + (wave + other * pw) * 1024 + i * 8192), 16,
                                               ((offB + (SDIAG(1) ? 0u : (unsigned)lt * 128u)) | ldead) + i * stepB + (unsigned)(other * pw * 16 * p.ldb), 0, 0,
                                               SDIAG(16384) ? 2 : 0);
  };
  auto next_slot = [&]() { lslot = lslot + 1 == NSLOT ? 0 : lslot + 1; };
  auto next_ktile = [&]() {     // after the B piece
    if (++lt == nkt) {
      lt = 0;
      ltile += gridDim.x;
      if (ltile < ntiles) set_load_tile(ltile);
    }
    ldead = (ltile < ntiles && lt * 64 + lchunk * 8 < p.K) ? 0u : OOB;
  };

  // ---- compute side ------------------------------------------------------------------------------------------------------
  // fragment of 16 rows x 32 k: lane (l15, q) reads row l15, logical chunk q + 4 h of the 128-byte row (h = which 32-deep half):
  // byte  row * 128 + ((q + 4 h) ^ ((row >> 1) & 7)) * 16  =  lane_off ^ (h << 6)  (+ 2048 per 16-row block)
  const unsigned lane_off = (unsigned)l15 * 128u + (unsigned)((q ^ (l15 >> 1)) << 4);
  const unsigned a_off = (unsigned)wr * 16384u + lane_off, b_off = (unsigned)wc * 8192u + lane_off;
  f32x4 acc[8][4];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4;   // (and again in the epilogue, block by block as they are staged)
  bf16x8 fa0[8], fb0[4], fa1[8], fb1[4];
  int ctile = blockIdx.x, ct = 0, cslot = 0;
  bool after_epi = false;
  constexpr int NST = EPI == BEPI_GELU2_BF16 ? 64 : 32;   // global stores per wave and tile

  // De-phasing (timing knob of the diagnostic build; the launcher passes one phase): workgroups sharing an XCD (ids 8 apart) start
  // `stagger` phases apart, a fraction of a tile period each, so that the chip's whole output of a round -- 128 KB per CU, 32 MB in
  // all -- does not hit the L2s and HBM as one burst.  It paid with ordinary stores; with non-temporal ones lockstep is faster.
  if (p.stagger > 1) {
    const int phase = ((int)blockIdx.x >> 3) % p.stagger;
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long wait = (long long)phase * p.stagger_cycles;
    while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
  }
  if constexpr (SDIAG(1024)) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // timing variant: static priority for the second-dispatched half
  }
  // prologue: two k-tiles in flight, the first one landed, its first half in registers
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NDMA > 0) dma_a(i);
      if constexpr (ISSUER) dma_a(i, 1);
    }
    next_slot();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (NDMA > 0) dma_b(i);
      if constexpr (ISSUER) dma_b(i, 1);
    }
    next_slot();
    next_ktile();
  }
  wait_vmcnt<2 * NDMA>();
  __builtin_amdgcn_s_barrier();
  {
    const unsigned char* sA = smem + a_off, *sB = smem + PIECE + b_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) fb0[j] = *reinterpret_cast<const bf16x8*>(sB + j * 2048);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa0[i] = *reinterpret_cast<const bf16x8*>(sA + i * 2048);
  }
  wait_lgkm0();

#ifdef DGVIT_DIAG
  long long st_start = 0, st_wait = 0, st_h1 = 0, st_h2 = 0, st_t = 0;
  int st_tile = 0;
  if constexpr (SDIAG(512)) st_start = st_t = __builtin_amdgcn_s_memtime();
#define STAMP_ADD(acc_)                                          \
  if constexpr (SDIAG(512)) {                                    \
    const long long now_ = __builtin_amdgcn_s_memtime();         \
    acc_ += now_ - st_t;                                         \
    st_t = now_;                                                 \
  }
#else
#define STAMP_ADD(acc_)
#endif
  while (ctile < ntiles) {
    const int s1 = cslot + 1 >= NSLOT ? cslot + 1 - NSLOT : cslot + 1, s2 = cslot + 2 >= NSLOT ? cslot + 2 - NSLOT : cslot + 2,
              s3 = cslot + 3 >= NSLOT ? cslot + 3 - NSLOT : cslot + 3;
    // ---- first half of k-tile t: MFMAs on (fa0, fb0); fragments of its second half -> (fa1, fb1); piece 2 t + 4 is issued -------
    {
      const unsigned char* sA = smem + cslot * PIECE + (a_off ^ 64u), *sB = smem + s1 * PIECE + (b_off ^ 64u);
      auto first_half = [&](auto par) {
      constexpr int PAR = decltype(par)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) if constexpr (!SDIAG(4)) fb1[j] = *reinterpret_cast<const bf16x8*>(sB + j * 2048);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if constexpr (!SDIAG(32)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j], fa0[i], acc[i][j], 0, 0, 0);
        if constexpr (!SDIAG(4)) fa1[i] = *reinterpret_cast<const bf16x8*>(sA + i * 2048);
        if constexpr (DSH != 0) {
          if constexpr (ISSUER) dma_a(i >> 1, i & 1);
        } else if ((i & 1) == PAR) dma_a(i >> 1);
      }
      sched_half<HVAR, PAR, LEAD>();
      };
      first_half(std::integral_constant<int, 1>{});
    }
    next_slot();
    STAMP_ADD(st_h1)
    wait_lgkm0();   // this wave holds every fragment of k-tile t ...
    // ... and its shares of pieces 2 t + 2, 2 t + 3 (k-tile t + 1) have landed: everything but the piece issued during this half.
    // Right after an epilogue the tile's NST stores are younger than those pieces too and may stay in flight (every one of them is
    // issued unconditionally, invalid rows / columns go out of range, so the count is exact): waiting for their acknowledgement
    // here would stall the first k-tile of every tile behind the write stream.
    if constexpr (NDMA > 0) {     // (a wave that issues no DMAs has nothing to wait for: its partner's counted wait and the barrier cover its rows)
      if (after_epi) {
        wait_vmcnt<(NDMA + NST <= 63 ? NDMA + NST : 63)>();
        after_epi = false;
      } else {
        wait_vmcnt<NDMA>();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PK > 0) __builtin_amdgcn_s_setprio(1);
    if constexpr (!SDIAG(16)) __builtin_amdgcn_s_barrier();        // k-tile t + 1 is complete; the slots of k-tile t are free
    STAMP_ADD(st_wait)
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half: MFMAs on (fa1, fb1); first-half fragments of k-tile t + 1 -> (fa0, fb0); piece 2 t + 5 into a slot of k-tile t
    if (EPI != BEPI_F32_PLAIN && p.bias && ct == nkt - 1) {
      // last k-tile of an output tile: the wave's 64 bias values go by ONE LDS-DMA (4 bytes per lane, range-checked: columns past N
      // read 0) into the first 256 bytes of its staging area -- slot s1 is free from this barrier on.  (Scalar loads cost ~3 us
      // each under this load, ordinary loads make hipcc wait vmcnt(0) behind the whole DMA ring.)  It is older than the four piece
      // DMAs of this phase, so `vmcnt(4)` at the start of the epilogue covers it.
      int bm0, bn0;
      tile_mn(xcd_chunk(ctile, ntiles), bm0, bn0);
      const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr_t)(smem + s1 * PIECE + wave * 1024), 4, (unsigned)(bn0 + wc * 64 + lane) * 4u, 0, 0, 0);
    }
    {
      const unsigned char* nA = smem + s2 * PIECE + a_off, *nB = smem + s3 * PIECE + b_off;
      auto second_half = [&](auto par) {
      constexpr int PAR = decltype(par)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) if constexpr (!SDIAG(4)) fb0[j] = *reinterpret_cast<const bf16x8*>(nB + j * 2048);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) if constexpr (!SDIAG(32)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j], fa1[i], acc[i][j], 0, 0, 0);
        if constexpr (!SDIAG(4)) fa0[i] = *reinterpret_cast<const bf16x8*>(nA + i * 2048);
        if constexpr (PK > 0) {
          if (i == PK - 1) __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (DSH != 0) {
          if constexpr (ISSUER) dma_b(i >> 1, i & 1);
        } else if ((i & 1) == PAR) dma_b(i >> 1);
      }
      sched_half<HVAR, PAR, LEAD>();
      };
      second_half(std::integral_constant<int, 1>{});
    }
    next_slot();
    next_ktile();
    wait_lgkm0();
    STAMP_ADD(st_h2)
    __builtin_amdgcn_sched_barrier(0);
    cslot = s2;
    int cnkt = nkt;
    if (++ct < cnkt) continue;

    if constexpr (!SDIAG(8)) {
    // ---- epilogue of tile `ctile`, straight from the accumulators: lane (l15, q) holds, for 16 x 16 block (i, j), row
    // 128 wr + 16 i + l15 and the four consecutive columns 64 wc + 16 j + 4 q + (0..3) ------------------------------------------------
    {
      int m0, n0;
      tile_mn(xcd_chunk(ctile, ntiles), m0, n0);
      // C window of this tile: num_records ends with the tile's last valid row, so rows past M are dropped by the range check of the
      // buffer stores; columns past N get an out-of-range column offset (N % 4 == 0: a lane's four columns are valid together)
      const int vrows = p.M - m0 < 256 ? p.M - m0 : 256, vcols = p.N - n0 < 256 ? p.N - n0 : 256;
      const int cbytes = ((vrows - 1) * p.ldc + vcols) * ES;
      const __amdgpu_buffer_rsrc_t rsC =
          __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.C) + ((long long)m0 * p.ldc + n0) * ES, 0, cbytes, 0x00020000);
      __amdgpu_buffer_rsrc_t rsC2 = rsC;
      if (EPI == BEPI_GELU2_BF16)
        rsC2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(p.C2) + ((long long)m0 * p.ldc2 + n0) * 2, 0, cbytes, 0x00020000);
      // Staging: partial-line stores straight from the accumulator layout (8 bytes per lane, 32-byte runs) drained at 2 TB/s and
      // held up the DMA stream behind them (2.9 TB/s with 16-byte fp32 pieces).  The slot of piece B(t) is free between this
      // k-tile's barrier and the next phase's DMA issue, and a wave's own four 1 KB DMA chunks of it are written by nobody else:
      // they hold one 16-row x 64-column fp32 block (row r in chunk r >> 2, 16-byte pieces swizzled by ^ r), written as four
      // ds_write_b128 from the accumulator layout and read back as 16 lanes x 16 bytes per row, so that every global store
      // instruction covers four whole 128-byte (bf16) / 256-byte (fp32) rows.
      unsigned char* stg = smem + s1 * PIECE + wave * 1024;
      const int rr = lane >> 4, rc = lane & 15;                                               // read-back: row 4 u + rr, piece rc
      const unsigned rowpart = (unsigned)(wr * 128 + rr) * (unsigned)p.ldc * ES, rowstep = 4u * (unsigned)p.ldc * ES;
      const unsigned coloff = n0 + wc * 64 + 4 * rc < p.N ? (unsigned)(wc * 64 + 4 * rc) * ES : OOB;
      fx4 bv[4];
      if (EPI != BEPI_F32_PLAIN && p.bias) {
        wait_vmcnt<NDMA>();   // the bias DMA (issued before this phase's four piece DMAs) has landed; same wave: no barrier needed
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const fx4*>(stg + (16 * j + 4 * q) * 4);
        wait_lgkm0();
        __builtin_amdgcn_wave_barrier();
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = fx4{0.f, 0.f, 0.f, 0.f};
      }
      unsigned off = rowpart + coloff;
      if constexpr (OUT_F32) {
        // fp32 output: one 16 x 64 fp32 block fills the wave's four chunks (row r in chunk r >> 2, 16-byte pieces swizzled by ^ r)
        const unsigned w_off = (unsigned)(l15 >> 2) * 8192u + (unsigned)(l15 & 3) * 256u;      // + ((4 j + q) ^ l15) * 16
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<fx4*>(stg + w_off + (unsigned)(((4 * j + q) ^ l15) << 4)) = acc[i][j];
            acc[i][j] = zero4;      // the next tile accumulates from zero (a branch on "first k-tile" would split the half's basic block)
          }
          wait_lgkm0();     // (wave-private region: the wave's own writes are all it waits for)
          __builtin_amdgcn_wave_barrier();
          fx4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const fx4*>(stg + u * 8192 + rr * 256 + ((rc ^ (4 * u + rr)) << 4));
          wait_lgkm0();
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int u = 0; u < 4; ++u, off += rowstep) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v[u]), rsC, off, 0, ST_AUX);
        }
      } else {
        // bf16 outputs are staged as finished bf16 values: a 16 x 64 block is 2 KB = two of the wave's chunks (row r in chunk r >> 3,
        // 16-byte pieces swizzled by ^ ((r >> 1) & 7): conflict-free for the ds_write_b64 of the accumulator layout -- a lane's 8 bytes
        // are half of piece (4 j + q) >> 1 -- and for the ds_read_b128 of the row layout), read back as 8 lanes x 16 bytes per row so
        // that one global store instruction covers EIGHT whole 128-byte rows (half the store and LDS-read instructions of 8-byte
        // pieces).  Two blocks alternate and block i + 1 is written while block i's read is in flight: one LDS round trip per block on
        // the critical path instead of two.  (GELU with a pre-activation copy stages both outputs, one pair of chunks each, without
        // the overlap.)
        constexpr bool TWO = EPI == BEPI_GELU2_BF16;
        const unsigned wsw = (unsigned)((l15 >> 1) & 7);
        const unsigned w_off = (unsigned)(l15 >> 3) * 8192u + (unsigned)(l15 & 7) * 128u + (unsigned)(q & 1) * 8u;   // + (((4 j + q) >> 1) ^ wsw) * 16
        auto stage = [&](int i, int buf) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            fx4 v = acc[i][j] + bv[j];
            acc[i][j] = zero4;      // the next tile accumulates from zero
            const unsigned a = w_off + (((unsigned)(2 * j + (q >> 1)) ^ wsw) << 4);
            if constexpr (TWO) *reinterpret_cast<bf16x4*>(stg + 16384 + a) = __builtin_convertvector(v, bf16x4);   // pre-activation -> chunks 2, 3
            if constexpr (EPI == BEPI_GELU_BF16 || EPI == BEPI_GELU2_BF16) v = gelu_bf16x4(v);
            *reinterpret_cast<bf16x4*>(stg + (TWO ? 0 : buf * 16384) + a) = __builtin_convertvector(v, bf16x4);
          }
        };
        const int r8 = lane >> 3, k8 = lane & 7;                                      // read-back: row 8 u + r8 of the block, 16-byte piece k8
        const unsigned rd0 = (unsigned)r8 * 128u + (((unsigned)k8 ^ (unsigned)((r8 >> 1) & 3)) << 4);      // u = 0: swizzle (r8 >> 1) & 7
        const unsigned rd1 = 8192u + (unsigned)r8 * 128u + (((unsigned)k8 ^ (unsigned)(4 + ((r8 >> 1) & 3))) << 4);   // u = 1: rows 8..15
        unsigned off8 = (unsigned)(wr * 128 + r8) * (unsigned)p.ldc * 2u + (n0 + wc * 64 + 8 * k8 < p.N ? (unsigned)(wc * 64 + 8 * k8) * 2u : OOB);
        const unsigned rowstep8 = 8u * (unsigned)p.ldc * 2u;
        stage(0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          wait_lgkm0();     // block i is staged (wave-private region: its own writes are all it waits for)
          __builtin_amdgcn_wave_barrier();
          u32x4v v[2], v2[2];
          const unsigned char* src = stg + (TWO ? 0 : (i & 1) * 16384);
          v[0] = *reinterpret_cast<const u32x4v*>(src + rd0);
          v[1] = *reinterpret_cast<const u32x4v*>(src + rd1);
          if constexpr (TWO) {
            v2[0] = *reinterpret_cast<const u32x4v*>(stg + 16384 + rd0);
            v2[1] = *reinterpret_cast<const u32x4v*>(stg + 16384 + rd1);
          }
          if constexpr (!TWO) {
            if (i + 1 < 8) stage(i + 1, (i + 1) & 1);     // into the other pair of chunks, under this block's read latency
          }
          wait_lgkm0();
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int u = 0; u < 2; ++u, off8 += rowstep8) {
            if constexpr (TWO) __builtin_amdgcn_raw_buffer_store_b128(v2[u], rsC2, off8, 0, ST_AUX);   // ldc2 == ldc (checked at launch): same offset
            __builtin_amdgcn_raw_buffer_store_b128(v[u], rsC, off8, 0, ST_AUX);
          }
          if constexpr (TWO) {
            if (i + 1 < 8) stage(i + 1, 0);
          }
        }
      }
    }
    // DMA sharing: the partner's next DMAs go into THIS wave's staging chunks of slot s1 -- every wave must be out of its epilogue first
    if constexpr (DSH != 0 && !SDIAG(524288)) __builtin_amdgcn_s_barrier();     // (524288: timing only, racy)
    } else if constexpr (SDIAG(64)) {   // timing: main loop with its MFMAs, nothing stored (the accumulators are kept alive, then cleared)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          asm volatile("" ::"v"(acc[i][j]));
          acc[i][j] = zero4;
        }
    }   // (SDIAG(8): epilogue skipped)
#ifdef DGVIT_DIAG
    if constexpr (SDIAG(512)) {
      const long long now = __builtin_amdgcn_s_memtime();
      if (lane == 0 && st_tile < 8 && p.diag_stamps) {
        long long* o = p.diag_stamps + ((((long long)blockIdx.x * 8 + wave) * 8) + st_tile) * 8;
        o[0] = st_start; o[1] = st_t; o[2] = now; o[3] = st_wait; o[4] = st_h1; o[5] = st_h2; o[6] = __builtin_amdgcn_s_memrealtime(); o[7] = ctile;
      }
      ++st_tile;
      st_start = st_t = __builtin_amdgcn_s_memtime();
      st_wait = st_h1 = st_h2 = 0;
    }
#endif
    ct = 0;
    ctile += gridDim.x;
    after_epi = !SDIAG(8);
  }
  wait_vmcnt<0>();   // the trailing (all-zero) LDS-DMAs must land before the workgroup gives its LDS back
}

template <int EPI, int DIAG = 0>
__global__ void __launch_bounds__(512) gemm_bf16_stream_kernel(const GemmBf16Params p, int ntiles) {
  if constexpr (SDIAG(256) || dma_sharing(DIAG) != 0) {
    if (threadIdx.x < 256) stream_body<EPI, DIAG, 1>(p, ntiles);     // waves 0-3: one per SIMD
    else stream_body<EPI, DIAG, 2>(p, ntiles);                       // waves 4-7: their partners
  } else {
    stream_body<EPI, DIAG, 0>(p, ntiles);
  }
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n;
}

template <int EPI, int DIAG = 0>
int launch_stream(const GemmBf16Params& p_in, hipStream_t st) {
  GemmBf16Params p = p_in;
#ifdef DGVIT_DIAG
  p.diag_stamps = g_gemm_bf16_stamps;
#endif
  if (p.group_m <= 0) p.group_m = g_gemm_bf16_group_m > 0 ? g_gemm_bf16_group_m % 1000 : 8;
  if (p.group_m <= 0) p.group_m = 8;
  {   // column blocks of the tile walk (see tile_mn): as few as keep a block's B panels inside the L2 budget; when not even four
      // columns fit (K = 3072: 1.5 MB per panel) persistence across rounds of tiles is out of reach and ONE block keeps what the
      // workgroups can still share, the k-slices they read at the same time
    const int tiles_n = (p.N + 255) / 256;
    const long long panel = 256ll * p.K * 2;
    const long long budget = (long long)g_gemm_bf16_l2_budget_kb * 1024;
    int fit = (int)(budget / panel);
    p.col_blocks = budget <= 0 ? 0 : (fit >= tiles_n || fit < 4) ? 1 : (tiles_n + fit - 1) / fit;
  }
  const long long tiles = (long long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  {   // de-phasing of the workgroups (see the kernel): `phases` start times a fraction of a tile period apart (the period estimated
      // from the measured ~2300 cycles per k-tile).  With ordinary output stores 8 phases were worth +3..7 % at 12 and 16 tiles per CU
      // (K = 768); since the stores are non-temporal, lockstep wins on every shape (QKV 253-269 vs 261-290 us, to_out 89.5 vs 93.5,
      // fc1 385-390 vs 397-400, fc2 326 vs 360: workgroups that share an A panel or a B column fetch the same k-slices at the same
      // time, profiles/r03_b_stream_gemm_phase_sweep.txt), so the default is one phase; the diagnostic build keeps the knob.
    int phases = 1;
    KNOB_IF(g_gemm_bf16_group_m >= 1000) phases = g_gemm_bf16_group_m / 1000;
    p.stagger = phases;
    p.stagger_cycles = phases > 1 ? (long long)((p.K + 63) / 64) * 2300 / phases : 0;
  }
  DGVIT_CHECK_ARG(tiles < (1ll << 30), "gemm_bf16: too many tiles");
  DGVIT_CHECK_ARG((long long)258 * p.ldc * 4 < (1ll << 31) && (long long)258 * p.ldc2 * 2 < (1ll << 31), "gemm_bf16: output leading dimension too large");
  DGVIT_CHECK_ARG(EPI != BEPI_GELU2_BF16 || p.ldc2 == p.ldc, "gemm_bf16: the GELU epilogue with a pre-activation copy needs ldc2 == ldc");
  constexpr int LDS = NSLOT * PIECE;
  auto kern = gemm_bf16_stream_kernel<EPI, DIAG>;
  static DeviceOnce once;
  if (const unsigned long long bit = once.pending()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "gemm_bf16: cannot raise the dynamic LDS limit to %d bytes", LDS);
    once.mark(bit);
  }
  const int grid = (int)(tiles < num_cus() ? tiles : num_cus());   // one persistent workgroup per CU
  const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, st);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, st, p, (int)tiles);
  profile_end(slot, st);
  DGVIT_CHECK_LAUNCH("gemm_bf16_stream_kernel");
  return DGVIT_OK;
}

}  // namespace

// shapes / epilogues the stream kernel takes (the callers fall back to the ring / simple kernels otherwise)
bool gemm_bf16_stream_supports(int epi, const GemmBf16Params& p) {
  if (p.tn || p.ksplit > 1 || p.c_rgrp > 0 || p.res_mod > 0 || p.res) return false;
  if (!(epi == BEPI_BF16 || epi == BEPI_GELU_BF16 || epi == BEPI_GELU2_BF16 || epi == BEPI_F32_PLAIN)) return false;
  if (epi != BEPI_F32_PLAIN && (p.N % 8 || p.ldc % 8 || (epi == BEPI_GELU2_BF16 && p.ldc2 % 8))) return false;   // 16-byte bf16 store pieces
  return p.K % 8 == 0 && p.N % 4 == 0;
}

int gemm_bf16_stream(int epi, const GemmBf16Params& p, hipStream_t st) {
#ifdef DGVIT_DIAG   // timing variants (dgvit_set_gemm_diagnostics), epilogue 0 / 1 only
  if (g_gemm_diag && epi == BEPI_F32_PLAIN) {
    if (g_gemm_diag == 32) return launch_stream<BEPI_F32_PLAIN, 32>(p, st);
    if (g_gemm_diag == 40) return launch_stream<BEPI_F32_PLAIN, 40>(p, st);
    if (g_gemm_diag == 2) return launch_stream<BEPI_F32_PLAIN, 2>(p, st);
  }
  if (g_gemm_diag && (epi == BEPI_BF16 || epi == BEPI_GELU_BF16)) {
#define DGVIT_SD(D)                                                                        \
  if (g_gemm_diag == D) return epi == BEPI_BF16 ? launch_stream<BEPI_BF16, D>(p, st) : launch_stream<BEPI_GELU_BF16, D>(p, st);
    DGVIT_SD(1) DGVIT_SD(2) DGVIT_SD(4) DGVIT_SD(8) DGVIT_SD(10) DGVIT_SD(14) DGVIT_SD(18) DGVIT_SD(30) DGVIT_SD(32) DGVIT_SD(40) DGVIT_SD(41) DGVIT_SD(44) DGVIT_SD(16) DGVIT_SD(72) DGVIT_SD(73) DGVIT_SD(74) DGVIT_SD(76) DGVIT_SD(88) DGVIT_SD(128) DGVIT_SD(256) DGVIT_SD(512) DGVIT_SD(768) DGVIT_SD(1024) DGVIT_SD(1536) DGVIT_SD(1280) DGVIT_SD(2048) DGVIT_SD(2049) DGVIT_SD(3072) DGVIT_SD(4096) DGVIT_SD(4168) DGVIT_SD(8192) DGVIT_SD(16384) DGVIT_SD(24576) DGVIT_SD(32768) DGVIT_SD(65536) DGVIT_SD(33280) DGVIT_SD(33792) DGVIT_SD(131072) DGVIT_SD(262144) DGVIT_SD(393216) DGVIT_SD(524288) DGVIT_SD(1048576) DGVIT_SD(2097152) DGVIT_SD(1049088) DGVIT_SD(2097664)
#undef DGVIT_SD
    return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16_stream: no timing variant %d", g_gemm_diag);
  }
#endif
  switch (epi) {
    case BEPI_BF16: return launch_stream<BEPI_BF16>(p, st);
    case BEPI_GELU_BF16: return launch_stream<BEPI_GELU_BF16>(p, st);
    case BEPI_GELU2_BF16: return launch_stream<BEPI_GELU2_BF16>(p, st);
    case BEPI_F32_PLAIN: return launch_stream<BEPI_F32_PLAIN>(p, st);
    default: return dgvit_set_error(DGVIT_ERR_ARG, "gemm_bf16_stream: unsupported epilogue %d", epi);
  }
}
