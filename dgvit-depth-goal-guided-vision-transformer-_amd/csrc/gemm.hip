// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 in, fp32 accumulate).
//
// This is the dominant kernel of the DGViT hot path: every Linear of the encoder
// (GoalFormer.py:43,46,64,67,139 and the head Linears of got_sac_network.py) and both of its
// gradients run through it.  One kernel template covers the three operand layouts
//   NT  Y  = X W^T      (forward)         A k-contiguous, B k-contiguous
//   NN  dX = dY W       (data gradient)   A k-contiguous, B n-contiguous
//   TN  dW = dY^T X     (weight gradient) A m-contiguous, B n-contiguous, split over K = tokens
// without any transposed copy in HBM:
//   * a k-contiguous operand is staged as LDS[row][BK+4] and read back with one ds_read_b128 per
//     32-row MFMA tile and 8-deep k-group (lane (i, h) takes k = 8g + 4h .. +3; the +4 pad makes the
//     16-lane b128 groups hit 16 distinct 16-byte bank slots);
//   * an m/n-contiguous operand is staged as LDS[k][R+4]; a wave's 32-row MFMA tiles are interleaved over its rows (frag_mc) so
//     that one ds_read_b64 per k-row feeds two tiles (256 B/clk; ds_read_b32 runs at 128 B/clk and made the TN / NN forms
//     12 % / 6 % slower than NT on large problems);
//   both read paths feed MFMA step s of k-group g with k = 8g + 4h + s, so the contraction order is the
//   same permutation for A and B whatever their layouts.
// Global->LDS staging is register-staged and double-buffered in LDS: tile t+1 is fetched to VGPRs
// before the MFMAs of tile t issue and written to the other LDS buffer after them (one barrier per
// k-tile).  Four waves (2x2) per workgroup, each owning a (BM/2)x(BN/2) block of 32x32 accumulators.
// Epilogue: the accumulators (column on the lane, rows in registers) are transposed through the now idle
// staging LDS so that every global access of the epilogue -- C, residual, GELU input/output -- is a
// 16-byte-per-lane, 512-byte-per-row coalesced float4; bias / residual / GELU / GELU' / ReLU are applied
// on the way out.  The weight-gradient form also sums its A tiles over k (= the bias gradient) for free.
#include <algorithm>

#include "common.h"

#define TRY_RG(expr)      \
  do {                    \
    int rc_ = (expr);     \
    if (rc_) return rc_;  \
  } while (0)


namespace {

// BM x BN output tile, BK-deep k-tiles, WVM x WVN waves (each owning a (BM/WVM) x (BN/WVN) block of 32x32 accumulators).
// 2x2 waves everywhere: two-wave workgroups (1x2 / 2x1) measured 10-45 % slower and eight-wave ones (128x128 as 2x4,
// 256x128x16 as 4x2) hit the same 83 % in-CU ceiling as 2x2 (tools/gemm_fill_probe.py, round 1)
template <int BM_, int BN_, int BK_, int WVM_ = 2, int WVN_ = 2>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WVM = WVM_, WVN = WVN_, NT = 64 * WVM_ * WVN_;
  // workgroups per CU the two LDS stages allow (<= 32 KB each: 5, the occupancy the K = 256 shapes are tuned at); the register
  // allocator is held to it, so an epilogue variant cannot silently cost a resident workgroup
  static constexpr int LDS_BYTES = 2 * (BM_ + BN_) * (BK_ + 4) * 4;
  static constexpr int MINB = LDS_BYTES * 5 <= 160 * 1024 ? 5 : 2;
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
  // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous chunk of the tile grid so
  // that the column tiles sharing an A row-panel hit the same L2 (speed only, bijective for any n).
  const int q = n >> 3, r = n & 7, xcd = id & 7, loc = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// ---- global -> register tile fetch ---------------------------------------------------------------
// KC: tile is R rows x BK k (k contiguous in memory).  MC: tile is BK k-rows x R (row index contiguous).
// VEC == 4: 16-byte buffer loads through a per-workgroup resource descriptor; rows/columns/k outside the
// matrix are dropped by the hardware range check (offset >= num_records reads 0), so the fetch is
// branch-free and can be scheduled among the MFMAs.  VEC == 1: scalar loads with explicit predicates
// (odd leading dimensions / unaligned bases; small head and odd-patch GEMMs only).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define DGVIT_OOB 0xFFFFFFF0u

template <int R, int BK, bool KC, int VEC, int NT, bool GATHER = false>
struct Fetch {
  static_assert(!GATHER || (KC && VEC == 4), "the patch gather is a k-contiguous float4 fetch");
  static_assert(R * BK / 4 % NT == 0, "tile must split evenly over the workgroup's threads");
  static constexpr int NV = R * BK / 4 / NT;  // float4 slots per thread
  static constexpr int PER_ROW = KC ? BK / 4 : R / 4;

  // --- VEC == 4 ---------------------------------------------------------------------------------
  struct Plan {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned off[NV];   // byte offset of slot i at the block's first k-tile
    int kc[NV];         // KC: k offset of the slot inside a tile; MC: k row of the slot inside a tile
    unsigned bad[NV];   // MC: all ones when the slot's columns lie outside the matrix (OR-ed into the offset: no branch), else 0
    unsigned kstep;     // bytes to advance per k-tile
    int g_wi, g_pw, g_inv, g_shift;   // GATHER: image row pitch, window-row floats, 2^shift / pw + 1, shift
    int g_k0;                         // GATHER: first k of this workgroup's k-range (a k-slice of a split tile starts past 0)
  };

  // GATHER: A is never materialised.  Row m of the patch matrix starts at pixel (b, hy * ph, wx * xs) of the image; element k of
  // the row is p1 = k / pw image rows further down and p2 = k % pw floats to the right (k / pw by multiply-shift; dgvit_api checks
  // that it is exact for every k < K).  Non-overlapping patches (xs = pw) and the strided 5x5 windows of the NHWC convolutions
  // (xs = stride * C, pw = KW * C) are the same arithmetic.  The descriptor covers the whole image buffer.
  __device__ static __forceinline__ void plan_gather(Plan& pl, const GemmParams& p, int r0, int kbeg, int tid) {
    pl.g_k0 = kbeg;
    long long bytes = p.g_img_floats * 4;
    if (bytes > 0x7FFFFFFFll) bytes = 0x7FFFFFFFll;
    pl.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g_img), 0, (int)bytes, 0x00020000);
    pl.kstep = 0;
    pl.g_wi = p.g_wi; pl.g_pw = p.g_pw; pl.g_inv = p.g_inv; pl.g_shift = p.g_shift ? p.g_shift : 16;
    const int xs = p.g_xs ? p.g_xs : p.g_pw;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      const int m = r0 + a;
      const int b = m / p.g_P, pi = m - b * p.g_P, hy = pi / p.g_gw, wx = pi - hy * p.g_gw;
      pl.off[i] = ((unsigned)b * (unsigned)p.g_hw + (unsigned)(hy * p.g_ph) * (unsigned)p.g_wi + (unsigned)(wx * xs)) * 4u;
      pl.kc[i] = c;
      pl.bad[i] = m < p.M ? 0u : 0xFFFFFFFFu;
    }
  }

  __device__ static __forceinline__ void plan(Plan& pl, const float* base, int ld, int r0, int rmax, int kbeg, int ktotal,
                                              int tid) {
    // resource base = first element this workgroup can touch; num_records = bytes from there to the end of the matrix
    long long first, last;
    if (KC) {
      first = (long long)r0 * ld + kbeg;
      last = (long long)(rmax - 1) * ld + ktotal;       // one past the last valid element
    } else {
      first = (long long)kbeg * ld + r0;
      last = (long long)(ktotal - 1) * ld + rmax;
    }
    long long bytes = (last - first) * 4;
    if (bytes > 0x7FFFFFFFll) bytes = 0x7FFFFFFFll;
    if (bytes < 0) bytes = 0;
    pl.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + first), 0, (int)bytes, 0x00020000);
    pl.kstep = KC ? BK * 4u : (unsigned)BK * (unsigned)ld * 4u;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      if (KC) {
        pl.off[i] = ((unsigned)a * (unsigned)ld + (unsigned)c) * 4u;
        pl.kc[i] = c;
        pl.bad[i] = 0u;  // rows past rmax fall outside num_records
      } else {
        pl.off[i] = ((unsigned)a * (unsigned)ld + (unsigned)c) * 4u;
        pl.kc[i] = a;
        pl.bad[i] = r0 + c < rmax ? 0u : 0xFFFFFFFFu;
      }
    }
  }

  // fetch k-tile number `t` (k0 = kbeg + t*BK); klim = kend - kbeg
  __device__ static __forceinline__ void run4(float4 (&reg)[NV], const Plan& pl, int t, int klim) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      unsigned at;
      if constexpr (GATHER) {
        const unsigned k = (unsigned)(pl.g_k0 + t * BK + pl.kc[i]), p1 = (k * (unsigned)pl.g_inv) >> pl.g_shift, p2 = k - p1 * (unsigned)pl.g_pw;
        at = (pl.off[i] + (p1 * (unsigned)pl.g_wi + p2) * 4u) | pl.bad[i];
      } else {
        at = (pl.off[i] + (unsigned)t * pl.kstep) | pl.bad[i];   // num_records <= 0x7FFFFFFF: all ones is out of range
      }
      const unsigned o = t * BK + pl.kc[i] < klim ? at : DGVIT_OOB;
      reg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(pl.rsrc, o, 0, 0));
    }
  }

  // --- VEC == 1 ---------------------------------------------------------------------------------
  __device__ static __forceinline__ void run(float4 (&reg)[NV], const float* __restrict__ base, int ld, int r0,
                                             int rmax, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int row = r0 + a, k = k0 + c;
        if (row < rmax) {
          const float* src = base + (long long)row * ld + k;
          if (k + 0 < kend) v.x = src[0];
          if (k + 1 < kend) v.y = src[1];
          if (k + 2 < kend) v.z = src[2];
          if (k + 3 < kend) v.w = src[3];
        }
      } else {
        const int k = k0 + a, col = r0 + c;
        if (k < kend) {
          const float* src = base + (long long)k * ld + col;
          if (col + 0 < rmax) v.x = src[0];
          if (col + 1 < rmax) v.y = src[1];
          if (col + 2 < rmax) v.z = src[2];
          if (col + 3 < rmax) v.w = src[3];
        }
      }
      reg[i] = v;
    }
  }

  __device__ static __forceinline__ void stash(const float4 (&reg)[NV], float* lds, int tid) {
    constexpr int STRIDE = KC ? BK + 4 : R + 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      const int a = f / PER_ROW, c = (f % PER_ROW) * 4;
      *reinterpret_cast<float4*>(lds + a * STRIDE + c) = reg[i];
    }
  }
};

// ---- LDS -> MFMA operand fragments for one 8-deep k-group -----------------------------------------
template <int R, int BK, bool KC>
__device__ __forceinline__ void frag(float (&out)[4], const float* lds, int row, int g, int h) {
  if (KC) {
    const float4 v = *reinterpret_cast<const float4*>(lds + row * (BK + 4) + 8 * g + 4 * h);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
    const float* p = lds + (8 * g + 4 * h) * (R + 4) + row;
#pragma unroll
    for (int s = 0; s < 4; ++s) out[s] = p[s * (R + 4)];
  }
}

// m/n-contiguous operand, the wave's NTILE 32-row MFMA tiles INTERLEAVED: lane i of tile t owns row base + NTILE * i + t, so the
// NTILE values a lane needs from one k-row are adjacent in LDS and come in with one ds_read_b64 / b128 (256 B/clk) instead of NTILE
// ds_read_b32 (128 B/clk).  The permutation of the tile's rows is undone where the accumulators are written out.
template <int R, int NTILE>
__device__ __forceinline__ void frag_mc(float (&out)[NTILE][4], const float* lds, int base, int li, int g, int h) {
  typedef float vec_t __attribute__((ext_vector_type(NTILE == 1 ? 1 : NTILE == 2 ? 2 : 4)));
  static_assert(NTILE == 1 || NTILE == 2 || NTILE == 4, "frag_mc: tiles per wave");
  const float* p = lds + (8 * g + 4 * h) * (R + 4) + base + NTILE * li;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if constexpr (NTILE == 1) {
      out[0][s] = p[s * (R + 4)];
    } else {
      // volatile: keeps LLVM from pairing two of these into one ds_read2_b64, which runs at half the rate of two ds_read_b64
      typedef __attribute__((address_space(3))) const volatile vec_t lds_vec_t;
      const vec_t v = *(lds_vec_t*)(p + s * (R + 4));
#pragma unroll
      for (int t = 0; t < NTILE; ++t) out[t][s] = v[t];
    }
  }
}

// ---- instruction-order hints for the pipelined main loop -----------------------------------------------
// One k-tile = NG k-groups of MF MFMAs.  The LDS writes of the next tile (NW ds_write_b128) and the fetch of
// the tile after it (NW buffer loads) are spread one per MFMA over the first k-group; the fragments of
// k-group g+1 are read while k-group g's MFMAs run.  (LLVM SchedGroupMask: MFMA 0x8, VMEM read 0x20,
// DS read 0x100, DS write 0x200.)
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int cdiv_c(int a, int b) { return (a + b - 1) / b; }
// m-th of H MFMAs, each followed by its share of NI instructions of kind MASK
template <int m, int H, int NI, int MASK>
__device__ __forceinline__ void spread() {
  if constexpr (m < H) {
    SGB(0x8, 1);
    constexpr int c = cdiv_c((m + 1) * NI, H) - cdiv_c(m * NI, H);
    if constexpr (c > 0) SGB(MASK, c);
    spread<m + 1, H, NI, MASK>();
  }
}
template <int MF, int NW, int RPG, int NG>
__device__ __forceinline__ void sched_pattern() {
  static_assert(MF >= 2 && MF % 2 == 0, "sched_pattern: MFMAs per k-group");
  SGB(0x100, RPG);                       // fragments of k-group 0
  spread<0, MF / 2, NW, 0x200>();        // first half of k-group 0: LDS writes of the next tile
  if constexpr (NG > 1) SGB(0x100, RPG); // fragments of k-group 1
  spread<0, MF / 2, NW, 0x20>();         // second half: global fetch of the tile after next
  if constexpr (NG > 1) { if constexpr (NG > 2) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 2) { if constexpr (NG > 3) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 3) { if constexpr (NG > 4) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 4) { if constexpr (NG > 5) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 5) { if constexpr (NG > 6) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 6) { if constexpr (NG > 7) SGB(0x100, RPG); SGB(0x8, MF); }
  if constexpr (NG > 7) { SGB(0x8, MF); }
}
#undef SGB

// diagnostic stamps (dgvit_set_gemm_stamps): wave 0 of every workgroup records the shader clock at four points and where it ran
__device__ __forceinline__ void stamp(const GemmParams& p, int slot, int tid) {
#ifdef DGVIT_DIAG
  if (p.stamps && tid == 0 && (int)blockIdx.x < p.stamp_capacity) {
    long long* s = p.stamps + (long long)blockIdx.x * 16;
    s[slot] = __builtin_readcyclecounter();
    if (slot == 0) {
      s[4] = wall_clock64();
      s[5] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID, XCC_ID
    }
    if (slot == 3) s[6] = wall_clock64();
  }
#endif
}
// timing / A-B diagnostics of the per-tile kernel exist in the diagnostic build only (knobs.h)
#ifdef DGVIT_DIAG
#define DIAG_BIT(p, b) ((p).diag & (b))
#define DIAG_STAMPS(p) ((p).stamps != nullptr)
#else
#define DIAG_BIT(p, b) 0
#define DIAG_STAMPS(p) false
#endif

template <class T, int LAYOUT, int VEC, int EPI, bool GATHER = false>
__global__ void __launch_bounds__(T::NT, T::MINB) gemm_f32_kernel(const GemmParams p) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NT;
  constexpr bool AKC = LAYOUT != GEMM_TN;
  constexpr bool BKC = LAYOUT == GEMM_NT;
  // EPI_GELU2D / EPI_DMUL: the training forward stores gelu'(pre-activation) INSTEAD of the pre-activation (C) beside gelu (C2) -- the
  // erf form already holds exp(-x^2 / 2), the derivative costs two more FMAs there -- and the data-gradient GEMM multiplies by that
  // stored factor instead of evaluating erf and exp per element in its epilogue (same values, bit for bit)
  constexpr bool TWO_OUT = EPI == EPI_GELU2 || EPI == EPI_GELU2D;
  constexpr bool ACT_GRAD = EPI == EPI_DGELU || EPI == EPI_DMUL;
  // Direct-epilogue output stores of the two-output forward (fc1: gelu + gelu') and of the gelu'-scaled data gradient are NON-TEMPORAL: 10 MB
  // of output per round of tiles streamed through each 4 MB L2 and evicted the A panels the column tiles share -- fc1 forward read 121.6 MB per
  // launch for 28 MB of operands, the fc2 data gradient 346 for 238 (rocprofv3 FETCH_SIZE by launch, profiles/r04_g_gemm_traffic_by_launch.txt);
  // with aux 2 (no L2 allocation) 43.2 and 263.5 MB.  Step time unchanged (12.52 vs 12.53 ms interleaved with the bit on every direct epilogue).
  constexpr bool NT_STORES = TWO_OUT || ACT_GRAD;
  constexpr int WM = BM / T::WVM, WN = BN / T::WVN, TM = WM / 32, TN = WN / 32;
  static_assert(WM % 32 == 0 && WN % 32 == 0 && BM <= NT, "bad wave layout");
  constexpr int A_TILE = AKC ? BM * (BK + 4) : BK * (BM + 4);
  constexpr int B_TILE = BKC ? BN * (BK + 4) : BK * (BN + 4);
  constexpr int STAGE = A_TILE + B_TILE;
  using FA = Fetch<BM, BK, AKC, VEC, NT, GATHER>;
  using FB = Fetch<BN, BK, BKC, VEC, NT>;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / T::WVN, wn = wave % T::WVN;

  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  // In-launch split-K (forward / data-gradient forms): tiles >= p.split_from are cut into p.nsplit k-slices, one workgroup
  // each; the slices leave fp32 partial tiles in `p.slabs`, draw a ticket from the tile's counter, and the LAST arriver sums
  // the slices in slice order (deterministic) and runs the normal epilogue.  Two uses: GEMMs with far fewer tiles than the
  // chip has workgroup slots (small batches: T = 65 ... 2080 rows, K = 2048), and the last partial round of a large grid
  // (25600 x 256 outputs = 1600 tiles = 6.25 per CU: the 64 left-over tiles become 256 quarter-tiles).
  int tile, zs = 0, nz = 1;
  // Weight gradients (EPI_SPLITK): (tile, k-slice) pairs in ONE grid dimension, k-slice major, each XCD a contiguous chunk of that order.
  // An XCD then works on a few k-slices of EVERY output tile: the rows of dY and X in those slices are read once, by the one L2 that serves
  // all their tiles.  (Round 1-3 gave an XCD a few tiles and ALL their k-slices -- grid (tiles, 1, slices), XCD = blockIdx.x % 8: every XCD
  // streamed the whole of X, and the tiles sharing a dY block sat on one XCD but in different slices' dispatch order: 532 MB read per
  // fc1 / fc2 weight-gradient launch for 236 MB of operands, rocprofv3 FETCH_SIZE by launch, profiles/r04_g_gemm_traffic_by_launch.txt.)
  int zslice = blockIdx.z;
  if (EPI == EPI_SPLITK && p.zsplit > 1) {
    const int ntile = tiles_m * tiles_n, w = xcd_remap(blockIdx.x, ntile * p.zsplit);
    zslice = w / ntile;
    tile = w - zslice * ntile;
  } else if (EPI != EPI_SPLITK && p.nsplit > 1 && (int)blockIdx.x >= p.split_from) {
    const int r = (int)blockIdx.x - p.split_from;
    tile = p.split_from + r / p.nsplit;
    zs = r % p.nsplit;
    nz = p.nsplit;
  } else {
    tile = xcd_remap(blockIdx.x, EPI != EPI_SPLITK && p.nsplit > 1 ? p.split_from : tiles_m * tiles_n);
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = nz > 1 ? zs * p.kchunk_split : zslice * p.kchunk;
  const int kend = min(p.K, kbeg + (nz > 1 ? p.kchunk_split : p.kchunk));
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;  // TN only: column sum of this block's A rows (bias gradient), threads < BM of n-tile 0
  const bool do_colsum = EPI == EPI_SPLITK && p.colsum && n0 == 0 && tid < BM;

  float4 ra[FA::NV], rb[FB::NV];
  const int klim = kend - kbeg;
  stamp(p, 0, tid);

  // epilogue geometry: accumulator (col = lane&31, row = (r&3) + 8*(r>>2) + 4*h) -> LDS C image [rows][BN+4] -> float4 row pieces
  // Only ONE tile carries the rarely used epilogue forms, so that they do not inflate (and push into scratch memory) every other
  // instantiation: the element-wise path for outputs that cannot take float4 accesses (odd N / ldc, unaligned C: the host sends
  // those to 64 x 64 x 32, pick_tile) and the fused LayerNorm of a 64-wide output row.
  constexpr bool ELEMWISE = BM == 64 && BN == 64 && BK == 32;
  constexpr bool LN_OK = EPI == EPI_STORE && BN == 64 && VEC == 4 && !GATHER && LAYOUT != GEMM_TN;
  const bool evec = !ELEMWISE || p.evec;
  constexpr int CS = BN + 4;
  constexpr int NCHUNK = (BM * CS <= 2 * STAGE) ? 1 : T::WVM;   // whole tile at once, or one wave-row of the tile at a time
  constexpr int CROWS = BM / NCHUNK;
  static_assert(CROWS * CS <= 2 * STAGE && (NCHUNK == 1 || CROWS == WM), "epilogue C image does not fit the staging LDS");
  constexpr int C4 = BN / 4, RPP = NT / C4;
  const int cc = (tid % C4) * 4, rr0 = tid / C4;
  const int n = n0 + cc;
  // The bias row piece is fetched BEFORE the main loop (4 registers): at the epilogue it would be a dependent round trip of
  // several thousand cycles under load, paid by every tile.
  // which epilogue (uniform over the launch except for split tiles): see "direct epilogue" below
  constexpr bool DIRECT_OK = LAYOUT != GEMM_TN && VEC == 4 && !GATHER && EPI != EPI_SPLITK && (BKC || TN <= 2);
  const bool direct = DIRECT_OK && nz == 1 && evec && p.c_rgrp == 0 && p.res_mod == 0 && (!TWO_OUT || p.ldc2 == p.ldc) && !DIAG_BIT(p, 8) &&
                      !(LN_OK && p.ln_y);   // (the fused LayerNorm reduces over the 16 lanes that hold a row of the LDS image)
  int dcol[TN];     // direct epilogue: this lane's column(s) inside the tile
  float dbias[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    dcol[j] = wn * WN + (BKC ? j * 32 + li : li * TN + j);
    dbias[j] = 0.f;
  }
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if ((EPI == EPI_STORE || TWO_OUT || EPI == EPI_RELU || EPI == EPI_GELU) && p.bias) {
    if (direct) {
#pragma unroll
      for (int j = 0; j < TN; ++j) dbias[j] = p.bias[min(n0 + dcol[j], p.N - 1)];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) bias4[e] = p.bias[n + e];
    }
  }

  // LDS -> fragments of one 8-deep k-group; MFMAs of one k-group
  auto load_frags = [&](float (&fa)[TM][4], float (&fb)[TN][4], const float* la, const float* lb, int g) {
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < TM; ++i) frag<BM, BK, true>(fa[i], la, wm * WM + i * 32 + li, g, h);
    } else {
      frag_mc<BM, TM>(fa, la, wm * WM, li, g, h);
    }
    if constexpr (BKC) {
#pragma unroll
      for (int j = 0; j < TN; ++j) frag<BN, BK, true>(fb[j], lb, wn * WN + j * 32 + li, g, h);
    } else {
      frag_mc<BN, TN>(fb, lb, wn * WN, li, g, h);
    }
  };
  auto do_mfma = [&](const float (&fa)[TM][4], const float (&fb)[TN][4]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s4], fb[j][s4], acc[i][j], 0, 0, 0);
  };
  auto mma_group = [&](const float* la, const float* lb, int g) {
    float fa[TM][4], fb[TN][4];
    load_frags(fa, fb, la, lb, g);
    do_mfma(fa, fb);
  };

  if (VEC == 4) {
    // Software pipeline (2 LDS buffers, 1 register set, 1 barrier per k-tile):
    //   iteration t:  LDS[t+1] <- registers (tile t+1, fetched during iteration t-1)
    //                 registers <- global tile t+2         (in flight for a whole iteration)
    //                 MFMAs on LDS[t]
    // the fetch is branch-free (hardware range check), so loads, LDS writes and MFMAs share one basic block.
    typename FA::Plan pa;
    typename FB::Plan pb;
    if constexpr (GATHER) FA::plan_gather(pa, p, m0, kbeg, tid);
    else FA::plan(pa, p.A, p.lda, m0, p.M, kbeg, p.K, tid);
    FB::plan(pb, p.B, p.ldb, n0, p.N, kbeg, p.K, tid);
    // prologue: the fetches of k-tiles 0 AND 1 are in flight together (one exposed round trip per output tile instead of two;
    // with K = 256 a tile has only 8-16 k-tiles to amortise it over)
    {
      float4 ra0[FA::NV], rb0[FB::NV];
      FA::run4(ra0, pa, 0, klim);
      FB::run4(rb0, pb, 0, klim);
      FA::run4(ra, pa, 1, klim);
      FB::run4(rb, pb, 1, klim);
      FA::stash(ra0, smem, tid);
      FB::stash(rb0, smem + A_TILE, tid);
    }
    __syncthreads();
    stamp(p, 1, tid);
    if (DIAG_BIT(p, 1)) __builtin_amdgcn_s_setprio(2);   // A/B knob: main-loop waves ahead of the prologue / epilogue waves they share a SIMD with
    for (int kt = 0; kt < nk; ++kt) {
      const float* la = smem + (kt & 1) * STAGE;
      const float* lb = la + A_TILE;
      float* wa = smem + ((kt + 1) & 1) * STAGE;
      // program order = wanted issue order where LDS reads and writes may alias for the compiler:
      // fragments of k-group 0, then the next tile's LDS writes, then the k-groups (each prefetching the next)
      float fa[2][TM][4], fb[2][TN][4];
      load_frags(fa[0], fb[0], la, lb, 0);
      FA::stash(ra, wa, tid);
      FB::stash(rb, wa + A_TILE, tid);
      FA::run4(ra, pa, kt + 2, klim);
      FB::run4(rb, pb, kt + 2, klim);
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        if (g + 1 < BK / 8) load_frags(fa[(g + 1) & 1], fb[(g + 1) & 1], la, lb, g + 1);
        do_mfma(fa[g & 1], fb[g & 1]);
      }
      sched_pattern<4 * TM * TN, FA::NV + FB::NV, (AKC ? TM : (TM == 1 ? 2 : 4)) + (BKC ? TN : (TN == 1 ? 2 : 4)), BK / 8>();
      if (EPI == EPI_SPLITK && do_colsum) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += la[kk * (BM + 4) + tid];
      }
      __syncthreads();
    }
    if (DIAG_BIT(p, 1)) __builtin_amdgcn_s_setprio(0);
  } else {
    FA::run(ra, p.A, p.lda, m0, p.M, kbeg, kend, tid);
    FB::run(rb, p.B, p.ldb, n0, p.N, kbeg, kend, tid);
    FA::stash(ra, smem, tid);
    FB::stash(rb, smem + A_TILE, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const float* la = smem + (kt & 1) * STAGE;
      const float* lb = la + A_TILE;
      const bool more = kt + 1 < nk;
      if (more) {
        FA::run(ra, p.A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, tid);
        FB::run(rb, p.B, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, tid);
      }
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) mma_group(la, lb, g);
      if (EPI == EPI_SPLITK && do_colsum) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += la[kk * (BM + 4) + tid];
      }
      if (more) {
        float* wa = smem + ((kt + 1) & 1) * STAGE;
        FA::stash(ra, wa, tid);
        FB::stash(rb, wa + A_TILE, tid);
      }
      __syncthreads();
    }
  }

  stamp(p, 2, tid);
  if (DIAG_BIT(p, 2)) {   // diagnostic (dgvit_set_gemm_diagnostics(2)): main loop only - what would a free epilogue be worth?
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 1.2345678e33f) p.C[0] = sacc;
    return;
  }
  // ---- direct epilogue (NT / NN forms, whole-K tiles, vector-aligned outputs, plain row mapping) ---------------------------------
  // The accumulators go to global memory straight from registers: a lane's 4-byte (NT: one column per MFMA tile) or 8-byte (NN: its
  // two adjacent columns) pieces, 128 / 256 contiguous bytes per half-wave, through a buffer descriptor whose range check drops the
  // rows past M (columns past N get an out-of-range offset) - no LDS image, no barrier, no branch.  Side inputs (residual /
  // activation-gradient operand) of the whole wave tile are requested first and folded in before the first store.  Against the LDS
  // image path below (kept for split tiles, the row-remapped patch embedding and unaligned callers): QKV 113 -> 116-122 TFLOP/s.
  if constexpr (DIRECT_OK) {
    if (direct) {
      constexpr int CW = BKC ? 1 : TN;
      auto tile_rsrc = [&](const float* base, int ld) {
        long long bytes = ((long long)(p.M - 1 - m0) * ld + (p.N - n0)) * 4;   // tile origin .. end of the matrix
        if (bytes > 0x7FFFFFFFll) bytes = 0x7FFFFFFFll;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (long long)m0 * ld + n0), 0, (int)bytes, 0x00020000);
      };
      // byte offset of accumulator element r of MFMA tile (i, j) in that window; everything in the VGPR operand (the scalar offset of
      // a buffer instruction is not range-checked)
      auto eoff = [&](int i, int j, int r, int ld) -> unsigned {
        const unsigned rowpart = (unsigned)((wm * WM + i * 32 + (r & 3) + 8 * (r >> 2)) * ld * 4);   // uniform
        return n0 + dcol[j] < p.N ? rowpart + (unsigned)((4 * h * ld + dcol[j]) * 4) : DGVIT_OOB;
      };
      constexpr bool HAS_SIDE = EPI == EPI_STORE || ACT_GRAD || EPI == EPI_DRELU;
      const bool use_side = HAS_SIDE && (EPI == EPI_STORE ? p.res != nullptr : true);
      if (HAS_SIDE && use_side) {
        const float* sb = EPI == EPI_STORE ? p.res : p.aux;
        const int sld = EPI == EPI_STORE ? p.ldr : p.ldaux;
        const __amdgpu_buffer_rsrc_t sr = tile_rsrc(sb, sld);
        float side[TM][16][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (CW == 2) {
              // one 8-byte load for the lane's two adjacent columns.  The result is bit-cast as a WHOLE vector: casting the two
              // dwords to float one by one is what this toolchain narrows to a one-dword load (the equality tests
              // against the LDS-image epilogue and the pipelined kernel, which use 4-byte loads, would catch it)
              typedef float f32x2v __attribute__((ext_vector_type(2)));
              const f32x2v v2 = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(sr, eoff(i, 0, r, sld), 0, 0));
              side[i][r][0] = v2[0];
              side[i][r][1] = v2[1];
            } else {
#pragma unroll
              for (int j = 0; j < TN; ++j)
                side[i][r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, eoff(i, j, r, sld), 0, 0));
            }
          }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float sv = side[i][r][j];
              if (EPI == EPI_STORE) acc[i][j][r] += sv;                     // residual first, bias below: the order of the LDS path
              else if (EPI == EPI_DGELU) acc[i][j][r] *= gelu_erf_grad(sv);
              else if (EPI == EPI_DMUL) acc[i][j][r] *= sv;
              else acc[i][j][r] = sv > 0.f ? acc[i][j][r] : 0.f;
            }
      }
      // output stores: ordinary, or non-temporal (aux 2: no allocation in the L2) for NT_STORES epilogues -- and for every direct epilogue under
      // diagnostic bit 16 of dgvit_set_gemm_diagnostics
      auto st32 = [&](unsigned v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
        if (NT_STORES || DIAG_BIT(p, 16)) __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
      };
      auto st64 = [&](u32x2 v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
        if (NT_STORES || DIAG_BIT(p, 16)) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, 0);
      };
      const __amdgpu_buffer_rsrc_t cr = tile_rsrc(p.C, p.ldc);
      __amdgpu_buffer_rsrc_t c2r = cr;
      if (TWO_OUT) c2r = tile_rsrc(p.C2, p.ldc2);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            v[j] = acc[i][j][r];
            if (EPI == EPI_STORE || TWO_OUT) v[j] += dbias[j];
            else if (EPI == EPI_RELU) v[j] = fmaxf(v[j] + dbias[j], 0.f);
            else if (EPI == EPI_GELU) v[j] = gelu_erf(v[j] + dbias[j]);
          }
          if constexpr (CW == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (EPI == EPI_GELU2D) {
                float gl, gd;
                gelu_erf_both(v[j], gl, gd);
                st32(__builtin_bit_cast(unsigned, gd), cr, eoff(i, j, r, p.ldc));
                st32(__builtin_bit_cast(unsigned, gl), c2r, eoff(i, j, r, p.ldc));
              } else {
                st32(__builtin_bit_cast(unsigned, v[j]), cr, eoff(i, j, r, p.ldc));
                if (EPI == EPI_GELU2)
                  st32(__builtin_bit_cast(unsigned, gelu_erf(v[j])), c2r, eoff(i, j, r, p.ldc));
              }
            }
          } else {
            u32x2 w;
            if constexpr (EPI == EPI_GELU2D) {
              float gl0, gd0, gl1, gd1;
              gelu_erf_both(v[0], gl0, gd0);
              gelu_erf_both(v[1], gl1, gd1);
              w[0] = __builtin_bit_cast(unsigned, gd0);
              w[1] = __builtin_bit_cast(unsigned, gd1);
              st64(w, cr, eoff(i, 0, r, p.ldc));
              w[0] = __builtin_bit_cast(unsigned, gl0);
              w[1] = __builtin_bit_cast(unsigned, gl1);
              st64(w, c2r, eoff(i, 0, r, p.ldc));
              continue;
            }
            w[0] = __builtin_bit_cast(unsigned, v[0]);
            w[1] = __builtin_bit_cast(unsigned, v[1]);
            st64(w, cr, eoff(i, 0, r, p.ldc));
            if (EPI == EPI_GELU2) {
              w[0] = __builtin_bit_cast(unsigned, gelu_erf(v[0]));
              w[1] = __builtin_bit_cast(unsigned, gelu_erf(v[1]));
              st64(w, c2r, eoff(i, 0, r, p.ldc));
            }
          }
        }
      stamp(p, 3, tid);
      if (DIAG_STAMPS(p)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(p, 7, tid);
      }
      return;
    }
  }
  // ---- epilogue ------------------------------------------------------------------------------------
  float* Cz = p.C;
  if (EPI == EPI_SPLITK) {
    Cz += (long long)zslice * p.slab_stride;
    if (p.colsum && n0 == 0 && tid < BM && m0 + tid < p.M) Cz[(long long)p.M * p.N + m0 + tid] = bsum;
  }
  // Fused LayerNorm of a finished output row (p.ln_y; launch checks N == BN, so the BN / 4 lanes tid % C4 of a wave hold the row).
  // The arithmetic and its order are those of layernorm_fwd_kernel (float4 partial sums, xor butterfly; the butterfly steps that kernel
  // takes over lanes holding nothing add zeros), so the result is bit-identical to running that kernel on C afterwards.
  auto ln_piece = [&](int m, const float (&v)[4]) {
    if constexpr (LN_OK) {
      float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
      for (int o = C4 / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mu = s / (float)p.N;
      const float a = v[0] - mu, b = v[1] - mu, c = v[2] - mu, d = v[3] - mu;
      float q = (a * a + b * b) + (c * c + d * d);
#pragma unroll
      for (int o = C4 / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
      const float rsd = rsqrtf(q / (float)p.N + p.ln_eps);
      const float4 g = *reinterpret_cast<const float4*>(p.ln_g + n), be = *reinterpret_cast<const float4*>(p.ln_b + n);
      *reinterpret_cast<float4*>(p.ln_y + (long long)m * p.ln_ld + n) =
          make_float4((v[0] - mu) * rsd * g.x + be.x, (v[1] - mu) * rsd * g.y + be.y, (v[2] - mu) * rsd * g.z + be.z, (v[3] - mu) * rsd * g.w + be.w);
      if (cc == 0) {
        p.ln_mean[m] = mu;
        p.ln_rstd[m] = rsd;
      }
    }
  };
  // one output row piece (row m, columns n .. n+3, v = the k-complete sums): fused epilogue arithmetic and the global stores
  auto finish = [&](int m, float (&v)[4]) {
    float w2[4];
    long long crow = m;
    if (EPI == EPI_STORE && p.c_rgrp > 0) crow = (long long)m + m / p.c_rgrp + 1;
    float* cptr = Cz + crow * p.ldc + n;
    if (EPI == EPI_STORE) {
      if (p.res) {
        const long long rrow = p.res_mod > 0 ? (long long)(m % p.res_mod) + 1 : (long long)m;
        const float* rp = p.res + rrow * p.ldr + n;
        if (evec) {
          const float4 q = *reinterpret_cast<const float4*>(rp);
          v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.N) v[e] += rp[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += bias4[e];
    } else if (TWO_OUT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += bias4[e];
        if constexpr (EPI == EPI_GELU2D) gelu_erf_both(v[e], w2[e], v[e]);
        else w2[e] = gelu_erf(v[e]);
      }
    } else if (EPI == EPI_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias4[e], 0.f);
    } else if (EPI == EPI_GELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e] + bias4[e]);
    } else if (ACT_GRAD || EPI == EPI_DRELU) {
      const float* ap = p.aux + (long long)m * p.ldaux + n;
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      if (evec) {
        const float4 q = *reinterpret_cast<const float4*>(ap);
        a4[0] = q.x; a4[1] = q.y; a4[2] = q.z; a4[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < p.N) a4[e] = ap[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = EPI == EPI_DGELU ? v[e] * gelu_erf_grad(a4[e]) : (EPI == EPI_DMUL ? v[e] * a4[e] : (a4[e] > 0.f ? v[e] : 0.f));
    }
    if (evec) {
      *reinterpret_cast<float4*>(cptr) = make_float4(v[0], v[1], v[2], v[3]);
      if (TWO_OUT) *reinterpret_cast<float4*>(p.C2 + (long long)m * p.ldc2 + n) = make_float4(w2[0], w2[1], w2[2], w2[3]);
      if (LN_OK && p.ln_y) ln_piece(m, v);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) {
          cptr[e] = v[e];
          if (TWO_OUT) p.C2[(long long)m * p.ldc2 + n + e] = w2[e];
        }
    }
  };
  // k-slice of a split tile: the raw partial sums go to this slice's dense [BM][BN] slab
  float* slab = nullptr;
  if (EPI != EPI_SPLITK && nz > 1) slab = p.slabs + ((long long)(tile - p.split_from) * nz + zs) * (BM * BN);
  // ---- vector fast path (p.evec: every shape of the encoder).  Stamps showed the per-piece `finish` below spending 12-18 k
  // cycles per chunk: a conditional side-input load inside the piece loop makes the compiler wait vmcnt(0) at the merge point
  // of every piece, and on gfx9 that also waits for the PREVIOUS piece's store - a serial chain of loaded-memory round trips
  // (~4 k cycles each), a third of a K = 256 tile's life.  Here the side inputs (residual / activation-gradient operand) of
  // every piece are fetched first, with row indices clamped instead of branched on, so nothing is in flight behind a store;
  // then each chunk is LDS reads -> arithmetic -> stores back to back, no wait in between.
  constexpr int NP = CROWS / RPP;                       // row pieces per thread and chunk
  static_assert(CROWS % RPP == 0, "epilogue row pieces");
  constexpr bool HAS_SIDE = EPI == EPI_STORE || ACT_GRAD || EPI == EPI_DRELU;
  constexpr bool SIDE_ALL = NCHUNK == 1 && NP <= 4;     // one chunk whose side inputs fit in 16 registers: fetch them before the C image barrier
                                                        // (two-chunk tiles take the direct epilogue on the hot path; prefetching both chunks here spilled)
  const bool fast = evec && !slab;
  const bool nvalid = n < p.N;
  const int nc = nvalid ? n : 0;
  const bool use_side = HAS_SIDE && (EPI == EPI_STORE ? p.res != nullptr : true);
  float4 side[HAS_SIDE ? (SIDE_ALL ? NCHUNK * NP : NP) : 1];
#pragma unroll
  for (auto& sv : side) sv = make_float4(0.f, 0.f, 0.f, 0.f);   // (defined on every path: undefined values here were spilled to scratch)
  auto load_side = [&](int ch, float4* dst) {
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int m = min(m0 + ch * CROWS + rr0 + it * RPP, p.M - 1);
      const float* sp;
      if (EPI == EPI_STORE) {
        const long long rrow = p.res_mod > 0 ? (long long)(m % p.res_mod) + 1 : (long long)m;
        sp = p.res + rrow * p.ldr + nc;
      } else {
        sp = p.aux + (long long)m * p.ldaux + nc;
      }
      dst[it] = *reinterpret_cast<const float4*>(sp);
    }
  };
  if (HAS_SIDE && SIDE_ALL && fast && use_side) {
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) load_side(ch, side + ch * NP);
  }
#pragma unroll
  for (int ch = 0; ch < NCHUNK; ++ch) {
    if (NCHUNK == 1 || wm == ch) {
      const int rbase = NCHUNK == 1 ? wm * WM : 0;
      // (an m/n-contiguous operand's tiles are interleaved, see frag_mc: tile-row rt of tile i is row TM * rt + i, and likewise
      //  for columns, where a lane's TN values are adjacent and leave as one vector write)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rt = (r & 3) + 8 * (r >> 2) + 4 * h;
          float* crow = smem + (rbase + (AKC ? i * 32 + rt : rt * TM + i)) * CS + wn * WN;
          if constexpr (BKC || TN == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j) crow[j * 32 + li] = acc[i][j][r];
          } else {
            typedef float cvec_t __attribute__((ext_vector_type(TN)));
            cvec_t v;
#pragma unroll
            for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
            *reinterpret_cast<cvec_t*>(crow + li * TN) = v;
          }
        }
    }
    __syncthreads();
    stamp(p, 8 + 2 * ch, tid);       // chunk's C image in LDS
    if (slab) {
      // write-through (sc1) stores: the partial tile leaves this XCD's L2 at once, so no release fence (an L2 write-back of
      // every dirty line the other workgroups' C stores left there) is needed before the ticket
      const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, BM * BN * 4, 0x00020000);
      for (int rr = rr0; rr < CROWS; rr += RPP) {
        const float4 t = *reinterpret_cast<const float4*>(smem + rr * CS + cc);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), srs, (unsigned)(((ch * CROWS + rr) * BN + cc) * 4), 0, 16);
      }
    } else if (fast) {
      if (HAS_SIDE && !SIDE_ALL && use_side) load_side(ch, side);   // big tiles: per chunk (waits for the previous chunk's stores)
      float4* sd = side + (HAS_SIDE && SIDE_ALL ? ch * NP : 0);
      if (EPI == EPI_DGELU) {   // operand -> gelu'(operand) in place, piece by piece (keeps the erf temporaries of one piece live)
#pragma unroll
        for (int it = 0; it < NP; ++it)
          sd[it] = make_float4(gelu_erf_grad(sd[it].x), gelu_erf_grad(sd[it].y), gelu_erf_grad(sd[it].z), gelu_erf_grad(sd[it].w));
      }
      // one-chunk tiles read all their row pieces first; two-chunk tiles (whose hot path is the direct epilogue above) read them
      // one by one: the batch of NP = 4 pieces beside 4 side inputs did not fit the register budget of 5 workgroups per CU
      constexpr bool T_FIRST = NCHUNK == 1;
      float4 t[T_FIRST ? NP : 1];
      if constexpr (T_FIRST) {
#pragma unroll
        for (int it = 0; it < NP; ++it) t[it] = *reinterpret_cast<const float4*>(smem + (rr0 + it * RPP) * CS + cc);
      }
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int m = m0 + ch * CROWS + rr0 + it * RPP;
        if constexpr (!T_FIRST) t[0] = *reinterpret_cast<const float4*>(smem + (rr0 + it * RPP) * CS + cc);
        const float4 tv = t[T_FIRST ? it : 0];
        float v[4] = {tv.x, tv.y, tv.z, tv.w}, w2[4];
        if (EPI == EPI_STORE) {
          if (use_side) { v[0] += sd[it].x; v[1] += sd[it].y; v[2] += sd[it].z; v[3] += sd[it].w; }   // same order as `finish`: residual, then bias
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bias4[e];
        } else if (TWO_OUT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += bias4[e];
            if constexpr (EPI == EPI_GELU2D) gelu_erf_both(v[e], w2[e], v[e]);
            else w2[e] = gelu_erf(v[e]);
          }
        } else if (EPI == EPI_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias4[e], 0.f);
        } else if (EPI == EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e] + bias4[e]);
        } else if (ACT_GRAD || EPI == EPI_DRELU) {
          const float a4[4] = {sd[it].x, sd[it].y, sd[it].z, sd[it].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ACT_GRAD ? v[e] * a4[e] : (a4[e] > 0.f ? v[e] : 0.f);
        }
        if (m < p.M && nvalid) {
          long long crow = m;
          if (EPI == EPI_STORE && p.c_rgrp > 0) crow = (long long)m + m / p.c_rgrp + 1;
          int ncol = n;
          if (DIAG_BIT(p, 4)) {   // diagnostic: every tile stores over tile 0 (the same instructions, no HBM write stream; results garbage)
            crow = m - m0;
            ncol = cc;
          }
          *reinterpret_cast<float4*>(Cz + crow * p.ldc + ncol) = make_float4(v[0], v[1], v[2], v[3]);   // (non-temporal stores measured the same: +-1 %)
          if (TWO_OUT) *reinterpret_cast<float4*>(p.C2 + crow * p.ldc2 + ncol) = make_float4(w2[0], w2[1], w2[2], w2[3]);
          if (LN_OK && p.ln_y) ln_piece(m, v);
        }
      }
    } else if (ELEMWISE && n < p.N) {
      // element-wise path (unaligned / odd-N callers): one piece at a time through `finish`
      for (int it = 0; it < CROWS / RPP; ++it) {
        const int rr = rr0 + it * RPP;
        const int m = m0 + ch * CROWS + rr;
        if (m < p.M) {
          const float4 t = *reinterpret_cast<const float4*>(smem + rr * CS + cc);
          float v[4] = {t.x, t.y, t.z, t.w};
          finish(m, v);
        }
      }
    }
    stamp(p, 9 + 2 * ch, tid);       // chunk's stores issued
    if (ch + 1 < NCHUNK) __syncthreads();
  }
  stamp(p, 3, tid);                                                 // stores issued
  if (DIAG_STAMPS(p)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // diagnostic run only: the last stamp sees them drained
    stamp(p, 7, tid);
  }
  if (EPI != EPI_SPLITK && nz > 1) {
    // publish the slab, take a ticket (cdna_hip_programming.md Guideline 16, recipe R1): write-through payload, every storing
    // wave drains its stores, barrier, ONE lane adds to the tile's counter; the last arriver acquires (agent scope) before
    // anybody reads the other slices.  Placement independent: nothing assumes which CU / XCD ran which slice.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nz - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    if (n < p.N) {
      const float* s0 = p.slabs + (long long)(tile - p.split_from) * nz * (BM * BN);
      for (int rr = rr0; rr < BM; rr += RPP) {
        const int m = m0 + rr;
        if (m >= p.M) break;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const float* sp = s0 + rr * BN + cc;
        for (int z0 = 0; z0 < nz; z0 += 8) {   // 8 slice loads in flight, then added in slice order: bit-reproducible
          float4 t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            t[u] = z0 + u < nz ? *reinterpret_cast<const float4*>(sp + (long long)(z0 + u) * (BM * BN)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w;
          }
        }
        finish(m, v);
      }
    }
  }
}

// out1[0..n1) , out2[0..n-n1)  <-  sum over slabs of slab[z][0..n), for up to DGVIT_REDUCE_JOBS independent jobs in ONE launch
// (a transformer layer's four split-K weight gradients + its two LayerNorm parameter-gradient partials).
// 256 threads = CW float4 columns x GS slab groups (CW * GS = 256; jobs with few columns and many slabs -- LayerNorm partials --
// take CW = 16, GS = 16): group y sums slabs y, y+GS, y+2GS, ... with 4 loads in flight, then the GS partial sums are combined
// through LDS in a fixed (tree) order: deterministic for a given (nslab, GS).
__global__ void __launch_bounds__(256) reduce_group_kernel(const ReduceGroup g) {
  __shared__ float4 part[256];
  int j = 0;
#pragma unroll
  for (int t = 1; t < DGVIT_REDUCE_JOBS; ++t)
    if (t < g.njobs && (int)blockIdx.x >= g.first_block[t]) j = t;
  const ReduceJob job = g.job[j];
  const int cwl = job.cw_log, CW = 1 << cwl, GS = 256 >> cwl;
  const int tx = threadIdx.x & (CW - 1), ty = threadIdx.x >> cwl;
  const long long i = (long long)((int)blockIdx.x - g.first_block[j]) * CW + tx;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < job.n4) {
    const float4* src = reinterpret_cast<const float4*>(job.slabs) + i;
    const long long st4 = job.stride4;
    int z = ty;
    for (; z + 3 * GS < job.nslab; z += 4 * GS) {
      const float4 a = src[(z + 0 * GS) * st4], b = src[(z + 1 * GS) * st4], c = src[(z + 2 * GS) * st4], d = src[(z + 3 * GS) * st4];
      s.x += (a.x + b.x) + (c.x + d.x);
      s.y += (a.y + b.y) + (c.y + d.y);
      s.z += (a.z + b.z) + (c.z + d.z);
      s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; z < job.nslab; z += GS) {
      const float4 a = src[z * st4];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  part[ty * CW + tx] = s;
  __syncthreads();
  for (int half = GS >> 1; half >= 1; half >>= 1) {   // fixed pairing: (y, y + half)
    if (ty < half) {
      const float4 a = part[ty * CW + tx], b = part[(ty + half) * CW + tx];
      part[ty * CW + tx] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (ty == 0 && i < job.n4) {
    const float4 r = part[tx];
    if (i < job.n14) reinterpret_cast<float4*>(job.out1)[i] = r;
    else reinterpret_cast<float4*>(job.out2)[i - job.n14] = r;
  }
}

__global__ void __launch_bounds__(256) reduce_slabs_scalar_kernel(const float* __restrict__ slabs, float* __restrict__ out1,
                                                                  float* __restrict__ out2, long long n, long long n1, int nslab,
                                                                  long long stride) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int z = 0; z < nslab; ++z) s += slabs[z * stride + i];
  if (i < n1) out1[i] = s;
  else out2[i - n1] = s;
}

#ifdef DGVIT_DIAG   // measured 4-15 % slower than the per-tile kernel: kept for tools/ and its equality tests only
// ---- pipelined persistent variant: one continuous k-tile stream per workgroup, a tile's stores under the next tile's MFMAs -----
// What the per-tile kernel loses at the K = 256 shapes (45 % of the step's GEMM time) is its epilogue: with the stores skipped
// (diagnostic bit of dgvit_set_gemm_diagnostics) QKV / fc1 / fc2-dgrad run at 133-135 TFLOP/s instead of 101-107
// (profiles/r02_c_gemm_no_epilogue_bound.txt).  Shortening the epilogue did not help (the time reappears as waiting elsewhere), and
// a persistent tile loop that keeps epilogue and main loop as separate phases is slower still: all workgroups of the chip fall into
// step and store at the same moment (profiles/r02_c_gemm_persistent_kernel_negative_result.txt).  So here there are no phases:
//   * the workgroup walks over its tiles (tile id += gridDim.x, same XCD-contiguous order) with ONE k-tile pipeline: iteration kt
//     of a tile fetches k-tile kt + 2 - of the NEXT tile for the last two iterations - so there is no prologue after the first;
//   * the finished tile's accumulators are copied to a second register set and leave during the next tile's main loop, one
//     accumulator row piece (TN values per lane and MFMA row tile) per iteration: buffer stores straight from registers (a lane's
//     32-bit (NT) or 64-bit (NN) pieces, 128 / 256 contiguous bytes per half-wave; rows past M dropped by the descriptor's range
//     check, columns past N and the not-yet-existing previous tile of the first round by an out-of-range offset - no branch);
//   * side inputs (residual / activation-gradient operand) of piece r are requested two iterations before they are used, the
//     first two pieces during the tile's own last two iterations.
// The k-tile count is a compile-time constant (NK = 16: K = 256 with 16-deep and K = 512 with 32-deep k-tiles) and the main loop is
// fully unrolled, which is what gives every iteration ITS accumulator registers to drain.  Memory operations of one iteration, in
// program order: operand fetch, side-input request, stores - vmcnt retires in order on gfx9, so every wait the compiler needs is for
// something older than the stores around it.  64 accumulator registers: 4 workgroups per CU instead of 5.
template <class T, int LAYOUT, int EPI, int NK>
__global__ void __launch_bounds__(T::NT, T::LDS_BYTES * 4 <= 160 * 1024 ? 4 : 2) gemm_f32_pipe_kernel(const GemmParams p) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, NT = T::NT;
  static_assert(LAYOUT == GEMM_NT || LAYOUT == GEMM_NN, "pipelined GEMM: forward / data-gradient forms");
  static_assert(EPI != EPI_SPLITK && NK % 2 == 0 && NK >= 4, "pipelined GEMM: complete-K tiles, even k-tile count");
  constexpr bool BKC = LAYOUT == GEMM_NT;
  constexpr int WM = BM / T::WVM, WN = BN / T::WVN, TM = WM / 32, TN = WN / 32;
  static_assert(16 % NK == 0 || NK % 16 == 0, "pipelined GEMM: accumulator rows per iteration");
  constexpr int RPI = NK >= 16 ? 1 : 16 / NK;     // accumulator rows (r) drained per iteration
  constexpr int DRAIN_EVERY = NK >= 16 ? NK / 16 : 1;
  constexpr int A_TILE = BM * (BK + 4);
  constexpr int B_TILE = BKC ? BN * (BK + 4) : BK * (BN + 4);
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int CW = BKC ? 1 : TN;            // floats per store: NT one column per MFMA tile, NN the lane's TN adjacent columns
  static_assert(CW == 1 || CW == 2, "pipelined GEMM: one or two adjacent columns per lane");
  constexpr bool HAS_SIDE = EPI == EPI_STORE || EPI == EPI_DGELU || EPI == EPI_DRELU;
  using FA = Fetch<BM, BK, true, 4, NT>;
  using FB = Fetch<BN, BK, BKC, 4, NT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave / T::WVN, wn = wave % T::WVN;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
  constexpr int klim = NK * BK;               // == p.K (checked at launch)

  auto load_frags = [&](float (&fa)[TM][4], float (&fb)[TN][4], const float* la, const float* lb, int g) {
#pragma unroll
    for (int i = 0; i < TM; ++i) frag<BM, BK, true>(fa[i], la, wm * WM + i * 32 + li, g, h);
    if constexpr (BKC) {
#pragma unroll
      for (int j = 0; j < TN; ++j) frag<BN, BK, true>(fb[j], lb, wn * WN + j * 32 + li, g, h);
    } else {
      frag_mc<BN, TN>(fb, lb, wn * WN, li, g, h);
    }
  };
  int coln[TN];   // this lane's column(s) inside a tile
#pragma unroll
  for (int j = 0; j < TN; ++j) coln[j] = wn * WN + (BKC ? j * 32 + li : li * TN + j);
  const bool use_side = HAS_SIDE && (EPI == EPI_STORE ? p.res != nullptr : true);
  const float* side_base = EPI == EPI_STORE ? p.res : p.aux;
  const int sld = EPI == EPI_STORE ? p.ldr : p.ldaux;

  // window of one tile in a row-major matrix: tile origin .. end of the matrix (rows past M fall outside, columns are predicated)
  auto tile_rsrc = [&](const float* base, int ld, int m0, int n0) {
    long long bytes = ((long long)(p.M - 1 - m0) * ld + (p.N - n0)) * 4;
    if (bytes > 0x7FFFFFFFll) bytes = 0x7FFFFFFFll;
    if (bytes < 0) bytes = 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (long long)m0 * ld + n0), 0, (int)bytes, 0x00020000);
  };
  // byte offset of accumulator element r of MFMA tile (i, j) inside that window (all of it in the VGPR operand: the scalar offset of
  // a buffer instruction is not range-checked); `lanepart` is DGVIT_OOB for a lane that must not touch memory
  auto elem_off = [&](unsigned lanepart, int i, int r, int ld) -> unsigned {
    const unsigned rowpart = (unsigned)((wm * WM + i * 32 + (r & 3) + 8 * (r >> 2)) * ld * 4);   // uniform
    return lanepart == DGVIT_OOB ? DGVIT_OOB : rowpart + lanepart;
  };

  int id = blockIdx.x;
  int tile = xcd_remap(id, ntiles);
  int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  typename FA::Plan pa, pan;
  typename FB::Plan pb, pbn;
  float4 ra[FA::NV], rb[FB::NV];
  FA::plan(pa, p.A, p.lda, m0, p.M, 0, p.K, tid);
  FB::plan(pb, p.B, p.ldb, n0, p.N, 0, p.K, tid);
  {   // the only prologue: k-tiles 0 and 1 of the first tile
    float4 ra0[FA::NV], rb0[FB::NV];
    FA::run4(ra0, pa, 0, klim);
    FB::run4(rb0, pb, 0, klim);
    FA::run4(ra, pa, 1, klim);
    FB::run4(rb, pb, 1, klim);
    FA::stash(ra0, smem, tid);
    FB::stash(rb0, smem + A_TILE, tid);
  }
  __syncthreads();
  stamp(p, 0, tid);

  // the tile that is leaving: accumulators, bias, lane offsets (DGVIT_OOB until a first tile has finished), descriptors
  f32x16 prev[TM][TN];
  float bias_p[TN];
  unsigned lane_c[TN], lane_s[TN];   // lane part of the offsets into C (and C2) / the side input, or DGVIT_OOB
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    bias_p[j] = 0.f;
    lane_c[j] = DGVIT_OOB;
    lane_s[j] = DGVIT_OOB;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[i][j][r] = 0.f;
  }
  __amdgpu_buffer_rsrc_t c_rs = tile_rsrc(p.C, p.ldc, m0, n0), c2_rs = c_rs, s_rs = c_rs;
  float sd[16][TM][TN];   // side inputs of the leaving tile, by accumulator row (live from request to use only)
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sd[r][i][j] = 0.f;

  // request the side inputs of accumulator row r of the tile at (sm0, sn0): descriptor `rs`, lane parts `ls`
  auto side_request = [&](int r, const __amdgpu_buffer_rsrc_t& rs, const unsigned (&ls)[TN]) {
    if constexpr (HAS_SIDE) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          sd[r][i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, elem_off(ls[j], i, r, sld), 0, 0));
    }
  };
  // store accumulator row r of the leaving tile
  auto drain_row = [&](int r) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float v[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        v[j] = prev[i][j][r];
        if (EPI == EPI_STORE) {
          if (use_side) v[j] += sd[r][i][j];          // residual first, then bias: the order of gemm_f32_kernel
          v[j] += bias_p[j];
        } else if (EPI == EPI_GELU2) {
          v[j] += bias_p[j];
        } else if (EPI == EPI_RELU) {
          v[j] = fmaxf(v[j] + bias_p[j], 0.f);
        } else if (EPI == EPI_DGELU) {
          v[j] *= gelu_erf_grad(sd[r][i][j]);
        } else if (EPI == EPI_DRELU) {
          v[j] = sd[r][i][j] > 0.f ? v[j] : 0.f;
        }
      }
      if constexpr (CW == 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[j]), c_rs, elem_off(lane_c[j], i, r, p.ldc), 0, 0);
          if (EPI == EPI_GELU2)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gelu_erf(v[j])), c2_rs, elem_off(lane_c[j], i, r, p.ldc), 0, 0);
        }
      } else {
        u32x2 w;
        w[0] = __builtin_bit_cast(unsigned, v[0]);
        w[1] = __builtin_bit_cast(unsigned, v[1]);
        __builtin_amdgcn_raw_buffer_store_b64(w, c_rs, elem_off(lane_c[0], i, r, p.ldc), 0, 0);
        if (EPI == EPI_GELU2) {
          w[0] = __builtin_bit_cast(unsigned, gelu_erf(v[0]));
          w[1] = __builtin_bit_cast(unsigned, gelu_erf(v[1]));
          __builtin_amdgcn_raw_buffer_store_b64(w, c2_rs, elem_off(lane_c[0], i, r, p.ldc), 0, 0);
        }
      }
    }
  };

  while (true) {
    // ---- tile (m0, n0): its plans are pa / pb and its first two k-tiles are on their way (stage 0 in LDS, k-tile 1 in ra / rb)
    const int nid = id + (int)gridDim.x;
    const bool more = nid < ntiles;
    const int ntile = more ? xcd_remap(nid, ntiles) : 0;
    const int nm0 = (ntile / tiles_n) * BM, nn0 = (ntile % tiles_n) * BN;
    FA::plan(pan, p.A, p.lda, nm0, more ? p.M : 0, 0, p.K, tid);      // no next tile: an empty window, the fetches read 0
    FB::plan(pbn, p.B, p.ldb, nn0, more ? p.N : 0, 0, p.K, tid);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bias_c[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bias_c[j] = 0.f;
    if ((EPI == EPI_STORE || EPI == EPI_GELU2 || EPI == EPI_RELU) && p.bias) {
#pragma unroll
      for (int j = 0; j < TN; ++j) bias_c[j] = p.bias[min(n0 + coln[j], p.N - 1)];
    }
    // this tile's own output window and lane offsets (used for its first side requests now, for its stores during the next tile)
    const __amdgpu_buffer_rsrc_t cur_c = tile_rsrc(p.C, p.ldc, m0, n0);
    __amdgpu_buffer_rsrc_t cur_s = cur_c;
    if (HAS_SIDE && use_side) cur_s = tile_rsrc(side_base, sld, m0, n0);
    unsigned cur_lane_c[TN], cur_lane_s[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool ok = n0 + coln[j] < p.N;
      cur_lane_c[j] = ok ? (unsigned)((4 * h * p.ldc + coln[j]) * 4) : DGVIT_OOB;
      cur_lane_s[j] = (ok && use_side) ? (unsigned)((4 * h * sld + coln[j]) * 4) : DGVIT_OOB;
    }
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      const float* la = smem + (kt & 1) * STAGE;
      const float* lb = la + A_TILE;
      float* wa = smem + ((kt + 1) & 1) * STAGE;
      float fa[2][TM][4], fb[2][TN][4];
      load_frags(fa[0], fb[0], la, lb, 0);
      FA::stash(ra, wa, tid);
      FB::stash(rb, wa + A_TILE, tid);
      if (kt + 2 < NK) {
        FA::run4(ra, pa, kt + 2, klim);
        FB::run4(rb, pb, kt + 2, klim);
      } else {   // the stream runs on into the next tile
        FA::run4(ra, pan, kt + 2 - NK, klim);
        FB::run4(rb, pbn, kt + 2 - NK, klim);
      }
      // side inputs: rows of the leaving tile two drain steps ahead; its first two rows were requested by its own last iterations
      if constexpr (HAS_SIDE) {
        if (kt % DRAIN_EVERY == 0) {
#pragma unroll
          for (int q = 0; q < RPI; ++q) {
            const int r = (kt / DRAIN_EVERY) * RPI + q + 2 * RPI;
            if (r < 16) side_request(r, s_rs, lane_s);
          }
        }
      }
      if (kt % DRAIN_EVERY == 0) {
#pragma unroll
        for (int q = 0; q < RPI; ++q) drain_row((kt / DRAIN_EVERY) * RPI + q);
      }
      if constexpr (HAS_SIDE) {
        if (kt >= NK - 2 * DRAIN_EVERY && kt % DRAIN_EVERY == 0) {   // ... of THIS tile, for its first two drain steps in the next one
#pragma unroll
          for (int q = 0; q < RPI; ++q) side_request(((kt - (NK - 2 * DRAIN_EVERY)) / DRAIN_EVERY) * RPI + q, cur_s, cur_lane_s);
        }
      }
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        if (g + 1 < BK / 8) load_frags(fa[(g + 1) & 1], fb[(g + 1) & 1], la, lb, g + 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][i][s4], fb[g & 1][j][s4], acc[i][j], 0, 0, 0);
      }
      // (no sched_group_barrier pattern here: with it the side-input variants spill 230-270 registers and every variant measured
      //  slower; and without a sched_barrier per iteration the group solver does not finish on the unrolled body)
      __syncthreads();
    }
    // ---- the tile becomes the leaving one
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) prev[i][j] = acc[i][j];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bias_p[j] = bias_c[j];
      lane_c[j] = cur_lane_c[j];
      lane_s[j] = cur_lane_s[j];
    }
    c_rs = cur_c;
    s_rs = cur_s;
    if (EPI == EPI_GELU2) c2_rs = tile_rsrc(p.C2, p.ldc2, m0, n0);
    if (!more) break;
    id = nid;
    m0 = nm0;
    n0 = nn0;
    pa = pan;
    pb = pbn;
  }
  stamp(p, 2, tid);
  // ---- the last tile leaves without a main loop to hide under
  if constexpr (HAS_SIDE) {
#pragma unroll
    for (int r = 2 * RPI; r < 16; ++r) side_request(r, s_rs, lane_s);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) drain_row(r);
  stamp(p, 3, tid);
  if (DIAG_STAMPS(p)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(p, 7, tid);
  }
}

#endif   // DGVIT_DIAG
// ---- in-launch split-K policy -------------------------------------------------------------------------------------------
// (a) few tiles, long K (small batches: T = 65 ... 2080 rows against K = 2048): every tile is cut so that the grid fills the chip;
// (b) a big grid whose last partial round would leave most CUs idle (tiles mod 256 <= 128: measured at 25600 x 256 x 2048,
//     1600 tiles = 6.25 per CU run 12 % longer than 1536 tiles): only the left-over tiles are cut, into 256 / r slices.
// A slice keeps at least 4 k-tiles.  Returns nsplit <= 1 for "do not split".
inline GemmSplitPlan split_plan(int M, int N, int K, int BM, int BN, int BK, int wg_per_cu) {
  GemmSplitPlan pl = {};
  const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  pl.tiles = (int)tiles;
  pl.nsplit = 1;
  pl.split_from = (int)tiles;
  const int KT = (K + BK - 1) / BK, cus = 256, slots = cus * wg_per_cu;
  if (KT < 8 || tiles <= 0) return pl;
  int S = 1;
  long long from = tiles;
  if (tiles * 2 <= slots) {
    S = (int)std::min<long long>(KT / 4, slots / tiles);
    from = 0;
  } else if (tiles > slots) {
    const long long r = tiles % cus;
    if (r > 0 && r <= cus / 2) {
      S = (int)std::min<long long>(KT / 4, cus / r);
      from = tiles - r;
    }
  }
  if (S < 2) return pl;
  const int kt_per = (KT + S - 1) / S;
  S = (KT + kt_per - 1) / kt_per;          // no empty slices
  if (S < 2) return pl;
  pl.nsplit = S;
  pl.split_from = (int)from;
  pl.kchunk = kt_per * BK;
  pl.slab_floats = (tiles - from) * S * (long long)BM * BN;
  return pl;
}

#ifdef DGVIT_DIAG
template <class T, int LAYOUT, int EPI>
int launch_persistent(const GemmParams& p, hipStream_t stream, bool* taken) {
  *taken = false;
  constexpr int NK = 16;   // k-tiles per output tile the pipelined kernel is built for: K = 256 at BK = 16, K = 512 at BK = 32
  // built for the two tiles the automatic choice uses in these forms (every instantiation is a fully unrolled 16-iteration loop)
  constexpr bool TILE_OK = (T::BM == 64 && T::BN == 128 && T::BK == 16) || (T::BM == 64 && T::BN == 64 && T::BK == 32);
  if constexpr ((LAYOUT == GEMM_NT || LAYOUT == GEMM_NN) && EPI != EPI_SPLITK && TILE_OK) {
    constexpr int BM = T::BM, BN = T::BN, BK = T::BK;
    if (p.K != NK * BK) return DGVIT_OK;
    constexpr size_t lds = T::LDS_BYTES;
    auto kern = gemm_f32_pipe_kernel<T, LAYOUT, EPI, NK>;
    static int slots = 0;
    if (!slots) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
      int per_cu = 0;
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T::NT, lds);
      if (e != hipSuccess || per_cu < 1) return dgvit_set_error(DGVIT_ERR_HIP, "gemm: occupancy query: %s", hipGetErrorString(e));
      slots = 256 * per_cu;
    }
    const long long tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    if (g_gemm_persist == 1 && tiles < 2ll * slots) return DGVIT_OK;     // few tiles per slot: the per-tile kernel (and its tail split)
    if (tiles >= (1ll << 31)) return DGVIT_OK;
    GemmParams q = p;
    q.stamps = g_gemm_stamps;
    q.stamp_capacity = g_gemm_stamp_capacity;
    // equal shares: rounds = ceil(tiles / slots) tiles per workgroup, as few workgroups as that needs (a multiple of 8 for the XCD order)
    long long grid = slots;
    if (g_gemm_persist_grid > 0) {
      grid = g_gemm_persist_grid;
    } else if (tiles > slots) {
      const long long rounds = (tiles + slots - 1) / slots;
      grid = std::min<long long>(slots, ((tiles + rounds - 1) / rounds + 7) / 8 * 8);
    }
    grid = std::min<long long>(grid, tiles);
    const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(T::NT), lds, stream, q);
    profile_end(slot, stream);
    DGVIT_CHECK_LAUNCH("gemm_f32_pipe_kernel");
    ++g_gemm_persist_launches;
    *taken = true;
  }
  return DGVIT_OK;
}
#endif   // DGVIT_DIAG

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <class T, int LAYOUT, int VEC, int EPI, bool GATHER = false>
int launch(const GemmParams& p0, int nsplit, hipStream_t stream) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK;
  constexpr bool AKC = LAYOUT != GEMM_TN;
  constexpr bool BKC = LAYOUT == GEMM_NT;
  constexpr int A_TILE = AKC ? BM * (BK + 4) : BK * (BM + 4);
  constexpr int B_TILE = BKC ? BN * (BK + 4) : BK * (BN + 4);
  constexpr size_t lds = 2 * (A_TILE + B_TILE) * sizeof(float);
  static DeviceOnce once;
  auto kern = gemm_f32_kernel<T, LAYOUT, VEC, EPI, GATHER>;
  if (const unsigned long long bit = once.pending()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    once.mark(bit);
  }
  GemmParams p = p0;
  if (p.ln_y) {   // fused LayerNorm: the whole output row must sit in this tile's LDS image, vector path
    DGVIT_CHECK_ARG(EPI == EPI_STORE && VEC == 4 && !GATHER && LAYOUT != GEMM_TN && p.evec && p.N == BN && BN == 64 && p.c_rgrp == 0,
                    "gemm: fused LayerNorm needs the vector EPI_STORE path, a 64-wide tile and N == 64, got tile width %d, N = %d", BN, p.N);
    DGVIT_CHECK_ARG(p.ln_g && p.ln_b && p.ln_mean && p.ln_rstd && p.ln_ld % 4 == 0 && al16(p.ln_g) && al16(p.ln_b) && al16(p.ln_y),
                    "gemm: fused LayerNorm operands must be present and 16-byte aligned");
  }
  p.stamps = g_gemm_stamps;
  p.stamp_capacity = g_gemm_stamp_capacity;
  p.diag = g_gemm_diag;
  const long long tiles = (long long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  DGVIT_CHECK_ARG(tiles > 0 && tiles < (1ll << 31), "gemm: bad tile count %lld", tiles);
  long long blocks = tiles;
  p.nsplit = 1;
  p.split_from = (int)tiles;
  if (EPI != EPI_SPLITK && VEC == 4 && p.counters && p.slabs && p.evec && g_gemm_split) {   // (the gather loader takes k-slices too)
    const int occ = (int)std::min<size_t>(8, (160 * 1024) / lds);
    const GemmSplitPlan pl = split_plan(p.M, p.N, p.K, BM, BN, BK, occ);
    if (pl.nsplit > 1 && pl.slab_floats <= p.slab_capacity && tiles <= p.counter_capacity) {
      p.nsplit = pl.nsplit; p.split_from = pl.split_from; p.kchunk_split = pl.kchunk;
      blocks = pl.split_from + (tiles - pl.split_from) * pl.nsplit;
    }
  }
#ifdef DGVIT_DIAG
  if constexpr (VEC == 4 && !GATHER && EPI != EPI_SPLITK && LAYOUT != GEMM_TN) {
    // whole tiles only, vector epilogue, plain row mapping: the persistent kernel (tile loop in the workgroup, next tile's fetch
    // under the epilogue) when a resident slot gets several tiles
    if (g_gemm_persist && (p.nsplit == 1 || g_gemm_persist == 2) && nsplit == 1 && p.evec && p.c_rgrp == 0 && p.res_mod == 0 &&
        (EPI != EPI_GELU2 || p.ldc2 == p.ldc) && g_gemm_lds_pad == 0 && !p.ln_y) {
      bool taken = false;
      const int rc = launch_persistent<T, LAYOUT, EPI>(p0, stream, &taken);
      if (rc != DGVIT_OK || taken) return rc;
    }
  }
#endif
  dim3 grid((unsigned)blocks, 1, (unsigned)nsplit);
  p.zsplit = 0;
  if (EPI == EPI_SPLITK && nsplit > 1 && g_gemm_zfold && blocks * nsplit < (1ll << 31)) {   // k-slices folded into blockIdx.x (see the kernel)
    p.zsplit = nsplit;
    grid = dim3((unsigned)(blocks * nsplit), 1, 1);
  }
  const int slot = profile_begin(PROF_GEMM, 2.0 * p.M * p.N * p.K, stream);
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), lds + (size_t)g_gemm_lds_pad, stream, p);
  profile_end(slot, stream);
  DGVIT_CHECK_LAUNCH("gemm_f32_kernel");
  return DGVIT_OK;
}

// tile_hint = BM*1000000 + BN*1000 + BK (e.g. 128128032), 0 = automatic
#define DGVIT_TILES(X) X(128, 128, 32) X(128, 128, 16) X(64, 64, 32) X(64, 64, 64) X(128, 64, 32) X(64, 128, 32) X(64, 128, 16) X(64, 64, 16)

// measured on MI355X at T = 25600 token rows (tools/gemm_shapes_bench.py, profiles/r01_b_gemm_tiles.txt):
// weight gradients (long K, split over tokens) like the 128x128 tile; forward / data-gradient GEMMs are
// tile-quantisation and prologue bound, so they take small tiles: wide outputs 64x128x16, narrow 64x64x32
inline int auto_tile(int layout, int M, int N) {
  if (layout == GEMM_TN) return (M >= 128 && N >= 128) ? 128128032 : 64064032;
  // (round 4 re-run of the table, profiles/r04_c_gemm_tiles.txt: the choices stand, except the data gradient of to_out -- NN, N = 512,
  //  K = 256 -- which gains 5 % on the wide tile)
  if (layout == GEMM_NN && N >= 512) return 64128016;
  return N >= 1024 ? 64128016 : 64064032;
}

template <int LAYOUT, int EPI>
int pick_tile(const GemmParams& p, int nsplit, bool vec4, int tile_hint, hipStream_t stream) {
  if constexpr (LAYOUT == GEMM_NT && (EPI == EPI_STORE || EPI == EPI_RELU)) {
    if (p.g_img) return launch<TileCfg<64, 64, 32>, LAYOUT, 4, EPI, true>(p, nsplit, stream);   // patch / window gather in the A loader
  }
  if (!vec4) return launch<TileCfg<64, 64, 32>, LAYOUT, 1, EPI>(p, nsplit, stream);
  int choice = tile_hint;
  if (choice == 64) choice = 64064032;
  if (choice == 128) choice = 128128032;
  if (choice == 0) choice = auto_tile(LAYOUT, p.M, p.N);
  if (!p.evec) choice = 64064032;   // the one tile that carries the element-wise epilogue (odd N / ldc, unaligned outputs)
#define X(BM_, BN_, BK_) \
  if (choice == BM_ * 1000000 + BN_ * 1000 + BK_) return launch<TileCfg<BM_, BN_, BK_>, LAYOUT, 4, EPI>(p, nsplit, stream);
  DGVIT_TILES(X)
#undef X
  return dgvit_set_error(DGVIT_ERR_ARG, "gemm: unknown tile %d", choice);
}


}  // namespace


// the same for a GEMM whose A operand is gathered from an image (patch embedding, implicit-GEMM convolutions): always the 64 x 64 x 32 tile
GemmSplitPlan gemm_split_plan_gather(int M, int N, int K) {
  GemmSplitPlan none = {};
  none.nsplit = 1;
  if (M <= 0 || N <= 0 || K <= 0) return none;
  const size_t lds = 2 * (size_t)(64 * (32 + 4) + 64 * (32 + 4)) * sizeof(float);
  return split_plan(M, N, K, 64, 64, 32, (int)std::min<size_t>(8, (160 * 1024) / lds));
}

GemmSplitPlan gemm_split_plan(int layout, int M, int N, int K) {
  GemmSplitPlan none = {};
  none.nsplit = 1;
  if (layout == GEMM_TN || M <= 0 || N <= 0 || K <= 0) return none;
  const int c = auto_tile(layout, M, N);
  const int BM = c / 1000000, BN = (c / 1000) % 1000, BK = c % 1000;
  const bool akc = true, bkc = layout == GEMM_NT;
  const size_t lds = 2 * (size_t)((akc ? BM * (BK + 4) : BK * (BM + 4)) + (bkc ? BN * (BK + 4) : BK * (BN + 4))) * sizeof(float);
  return split_plan(M, N, K, BM, BN, BK, (int)std::min<size_t>(8, (160 * 1024) / lds));
}

int gemm_f32(int layout, int epi, const GemmParams& p, int nsplit, hipStream_t stream) {
  DGVIT_CHECK_ARG((p.A || p.g_img) && p.B && p.C, "gemm: null operand");
  DGVIT_CHECK_ARG(!p.g_img || (layout == GEMM_NT && (epi == EPI_STORE || epi == EPI_RELU) && p.g_pw % 4 == 0 && p.g_wi % 4 == 0 && p.g_xs % 4 == 0 &&
                               al16(p.g_img) && al16(p.B) && p.ldb % 4 == 0 && p.K % 4 == 0 && p.K == (p.g_kh ? p.g_kh : p.g_ph) * p.g_pw &&
                               (long long)p.K * p.g_inv < (1ll << 32)),
                  "gemm: patch gather needs the NT / store form, patch and image widths that are multiples of 4 and 16-byte aligned operands");
  DGVIT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  DGVIT_CHECK_ARG(nsplit >= 1 && p.kchunk > 0 && p.kchunk % 32 == 0, "gemm: kchunk must be a positive multiple of 32");
  DGVIT_CHECK_ARG((long long)p.kchunk * nsplit >= p.K, "gemm: split does not cover K");
  // float4 staging needs 16-byte aligned bases, leading dims that are multiples of 4 and a
  // contiguous extent that is a multiple of 4; anything else takes the scalar-load 64x64 variant.
  bool vec4 = al16(p.A) && al16(p.B) && p.lda % 4 == 0 && p.ldb % 4 == 0;
  if (layout == GEMM_NT) vec4 = vec4 && p.K % 4 == 0;
  if (layout == GEMM_NN) vec4 = vec4 && p.K % 4 == 0 && p.N % 4 == 0;
  if (layout == GEMM_TN) vec4 = vec4 && p.M % 4 == 0 && p.N % 4 == 0;
  DGVIT_CHECK_ARG(p.a_kgrp == 0, "gemm: a_kgrp is no longer supported (gather the rows first)");
  DGVIT_CHECK_ARG((long long)p.lda * 512 < 0x7FFFFFFFll && (long long)p.ldb * 512 < 0x7FFFFFFFll, "gemm: leading dimension too large");
  DGVIT_CHECK_ARG(layout == GEMM_NT || (long long)p.kchunk * p.ldb * 4 < 0x7FFFFFFFll, "gemm: k-chunk x ldb exceeds the 2 GiB descriptor window");
  DGVIT_CHECK_ARG(layout != GEMM_TN || (long long)p.kchunk * p.lda * 4 < 0x7FFFFFFFll, "gemm: k-chunk x lda exceeds the 2 GiB descriptor window");
  GemmParams q = p;
  q.evec = p.N % 4 == 0 && p.ldc % 4 == 0 && al16(p.C) && (!p.res || (p.ldr % 4 == 0 && al16(p.res))) &&
           (!p.C2 || (p.ldc2 % 4 == 0 && al16(p.C2))) && (!p.aux || (p.ldaux % 4 == 0 && al16(p.aux))) &&
           (epi != EPI_SPLITK || p.slab_stride % 4 == 0);
  const int th = g_gemm_tile_hint;
#define CASE(L, E) \
  if (layout == L && epi == E) return pick_tile<L, E>(q, nsplit, vec4, th, stream);
  CASE(GEMM_NT, EPI_STORE)
  CASE(GEMM_NT, EPI_GELU2)
  CASE(GEMM_NT, EPI_GELU2D)
  CASE(GEMM_NT, EPI_GELU)
  CASE(GEMM_NT, EPI_RELU)
  CASE(GEMM_NN, EPI_STORE)
  CASE(GEMM_NN, EPI_DGELU)
  CASE(GEMM_NN, EPI_DMUL)
  CASE(GEMM_NN, EPI_DRELU)
  CASE(GEMM_TN, EPI_SPLITK)
#undef CASE
  return dgvit_set_error(DGVIT_ERR_ARG, "gemm: unsupported layout/epilogue %d/%d", layout, epi);
}

// ---- grouped deterministic reductions ----------------------------------------------------------------------------------
void reduce_group_init(ReduceGroup& g) { g.njobs = 0; g.first_block[0] = 0; }

// launch every queued job as ONE kernel (no-op when empty)
int reduce_group_flush(ReduceGroup& g, hipStream_t stream) {
  if (g.njobs == 0) return DGVIT_OK;
  const int slot = profile_begin(PROF_OTHER, 0.0, stream);
  hipLaunchKernelGGL(reduce_group_kernel, dim3((unsigned)g.first_block[g.njobs]), dim3(256), 0, stream, g);
  profile_end(slot, stream);
  g.njobs = 0;
  DGVIT_CHECK_LAUNCH("reduce_group");
  return DGVIT_OK;
}

// queue: out1 gets the first n1 sums, out2 (may be null when n1 == n) the remaining n - n1, of nslab slabs slab_stride floats apart.
// Jobs that cannot take the float4 path (odd sizes / alignment: tiny head Linears) run at once on the scalar kernel.
int reduce_group_add(ReduceGroup& g, const float* slabs, float* out1, long long n1, float* out2, long long n, int nslab,
                     long long slab_stride, hipStream_t stream) {
  DGVIT_CHECK_ARG(slabs && out1 && n > 0 && n1 > 0 && n1 <= n && nslab >= 1 && (n1 == n || out2), "reduce_slabs: bad arguments");
  if (!(n % 4 == 0 && n1 % 4 == 0 && slab_stride % 4 == 0 && al16(slabs) && al16(out1) && (n1 == n || al16(out2)))) {
    const int slot = profile_begin(PROF_OTHER, 0.0, stream);
    hipLaunchKernelGGL(reduce_slabs_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, slabs, out1, out2, n, n1,
                       nslab, slab_stride);
    profile_end(slot, stream);
    DGVIT_CHECK_LAUNCH("reduce_slabs");
    return DGVIT_OK;
  }
  if (g.njobs == DGVIT_REDUCE_JOBS) TRY_RG(reduce_group_flush(g, stream));
  ReduceJob& job = g.job[g.njobs];
  job.slabs = slabs; job.out1 = out1; job.out2 = out2;
  job.n4 = n / 4; job.n14 = n1 / 4; job.nslab = nslab; job.stride4 = slab_stride / 4;
  job.cw_log = (job.n4 <= 1024 && nslab >= 64) ? 4 : 6;   // few columns, many slabs: 16 slab groups per block
  const long long blocks = (job.n4 + (1 << job.cw_log) - 1) >> job.cw_log;
  DGVIT_CHECK_ARG(blocks + g.first_block[g.njobs] < (1ll << 30), "reduce_slabs: too many blocks");
  g.first_block[g.njobs + 1] = g.first_block[g.njobs] + (int)blocks;
  ++g.njobs;
  return DGVIT_OK;
}

// out1 gets the first n1 sums, out2 (may be null when n1 == n) the remaining n - n1
int reduce_slabs2(const float* slabs, float* out1, long long n1, float* out2, long long n, int nslab, long long slab_stride,
                  hipStream_t stream) {
  ReduceGroup g;
  reduce_group_init(g);
  TRY_RG(reduce_group_add(g, slabs, out1, n1, out2, n, nslab, slab_stride, stream));
  return reduce_group_flush(g, stream);
}

int reduce_slabs(const float* slabs, float* out, long long n, int nslab, long long slab_stride, hipStream_t stream) {
  return reduce_slabs2(slabs, out, n, nullptr, n, nslab, slab_stride, stream);
}
