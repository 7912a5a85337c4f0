// extern "C" surface of libdgvit_hip.so (include/dgvit_hip.h) and the encoder forward/backward schedules.
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>
#include <mutex>

#include "../../include/dgvit_hip.h"
#include "common.h"
#include "kernels.h"

// Schedule options travel with every call in dgvit_config.flags (the forward and its backward see the same value, whatever thread
// runs them); A/B and diagnostic knobs are compile-time constants in this build unless DGVIT_DIAG is defined (knobs.h).
#ifdef DGVIT_DIAG
#include "../../include/dgvit_hip_diag.h"
int g_gemm_tile_hint = 0, g_gemm_split = 1, g_gemm_lds_pad = 0, g_gemm_persist = 0, g_gemm_persist_grid = 0, g_gemm_diag = 0;
long long* g_gemm_stamps = nullptr;
int g_gemm_stamp_capacity = 0;
long long g_gemm_persist_launches = 0;
int g_group_reduce = 1, g_ln_fusion = 1, g_conv_gather = 1, g_small_path = 0, g_small_path_max_rows = 4160, g_block_path = 1, g_block_path_max_rows = 4160, g_gelu_grad_store = 1, g_block_fuse = 2;
int g_gemm_bf16_tile_hint = 0, g_gemm_bf16_m16 = 1, g_gemm_bf16_group_m = 8, g_attn_bwd64 = 1, g_attn_q1 = 1, g_gemm_zfold = 1, g_gemm_bf16_l2_budget_kb = 2048;
long long* g_gemm_bf16_stamps = nullptr;
long long* g_block_stamps = nullptr;
int g_block_stamp_layer = -1, g_block_stamp_now = 1;
#endif
static inline bool dense_last_block(const dgvit_config* c) { return (c->flags & DGVIT_FLAG_DENSE_LAST_BLOCK) != 0; }
static inline bool wgrad_overlap(const dgvit_config* c) { return (c->flags & DGVIT_FLAG_WGRAD_OVERLAP) != 0; }

// ---------------------------------------------------------------------------------------------- helper stream
// dgvit_got_backward forks every weight-gradient GEMM (+ its slab reduction) onto one internal non-blocking
// stream and joins it back with events, so the wgrad workgroups fill the tail / prologue bubbles of the
// data-gradient kernels on the caller's stream.  All ordering is event based (capturable into a hipGraph);
// the helper stream and a ring of events are created on first use and live for the process.
namespace {
struct Side {
  hipStream_t stream = nullptr;
  hipEvent_t ring[32];
  unsigned next = 0;
  bool ready = false;
};
Side g_side;
std::mutex g_side_mu;

int side_init() {
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side.ready) return DGVIT_OK;
  if (hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking) != hipSuccess)
    return dgvit_set_error(DGVIT_ERR_HIP, "cannot create the helper stream");
  for (auto& e : g_side.ring)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
      return dgvit_set_error(DGVIT_ERR_HIP, "cannot create helper events");
  g_side.ready = true;
  return DGVIT_OK;
}
// `to` waits for everything enqueued on `from` so far
int chain(hipStream_t from, hipStream_t to) {
  hipEvent_t e = g_side.ring[g_side.next++ % 32];
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess)
    return dgvit_set_error(DGVIT_ERR_HIP, "event fork/join failed");
  return DGVIT_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

int dgvit_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define TRY(expr)          \
  do {                     \
    int rc_ = (expr);      \
    if (rc_) return rc_;   \
  } while (0)

#define HIP_TRY(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return dgvit_set_error(DGVIT_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

static inline long long al4(long long n) { return (n + 3) & ~3ll; }

// ---------------------------------------------------------------------------------------------- GEMM helpers
namespace {

GemmParams gp(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K) {
  GemmParams p = {};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.kchunk = (K + 31) / 32 * 32;
  return p;
}

// split-K plan for weight gradients: tiles x splits ~ 2 workgroups per CU (all co-resident, one balanced wave)
int wgrad_splits(int M, int N, int K) {
  const int bt = (M >= 128 && N >= 128) ? 128 : 64;  // must mirror pick_tile's automatic TN choice
  const long long tiles = (long long)((M + bt - 1) / bt) * ((N + bt - 1) / bt);
  long long s = 512 / tiles;
  const long long maxs = (K + 255) / 256;  // at least 8 k-tiles per split
  if (s > maxs) s = maxs;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

long long wgrad_slab(int M, int N) { return al4((long long)M * N + M); }
long long wgrad_scratch(int M, int N, int K) { return (long long)wgrad_splits(M, N, K) * wgrad_slab(M, N); }

// dW (M x N) = A^T B with A (K x M, lda), B (K x N, ldb); optional db (M) = column sums of A (fused in the kernel);
// dW == nullptr: the weight is frozen (its requires_grad is off): only the bias gradient, if wanted, is computed.
// grp != null: the slab reduction is queued there and `scratch` must stay untouched until the caller has flushed the group.
int wgrad(const float* A, int lda, const float* B, int ldb, float* dW, float* db, int M, int N, int K, float* scratch,
          long long scratch_floats, hipStream_t st, ReduceGroup* grp = nullptr) {
  if (!dW) {
    if (!db) return DGVIT_OK;
    if (scratch_floats < (long long)colsum_blocks(K) * M) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "wgrad: scratch too small for the bias gradient");
    return colsum(A, lda, db, scratch, K, M, 0, st);
  }
  const int ns = wgrad_splits(M, N, K);
  const long long slab = wgrad_slab(M, N);
  if (scratch_floats < ns * slab) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "wgrad: scratch %lld < %lld floats", scratch_floats, ns * slab);
  GemmParams p = gp(A, lda, B, ldb, scratch, N, M, N, K);
  const int kt = (K + 31) / 32;
  p.kchunk = ((kt + ns - 1) / ns) * 32;
  p.slab_stride = slab;
  p.colsum = db ? 1 : 0;
  const int ns_eff = (K + p.kchunk - 1) / p.kchunk;
  TRY(gemm_f32(GEMM_TN, EPI_SPLITK, p, ns_eff, st));
  const long long mn = (long long)M * N;
  ReduceGroup local;
  if (!grp) reduce_group_init(local);
  ReduceGroup& g = grp ? *grp : local;
  if (db && mn % 4 == 0) {
    TRY(reduce_group_add(g, scratch, dW, mn, db, mn + M, ns_eff, slab, st));
  } else {
    TRY(reduce_group_add(g, scratch, dW, mn, nullptr, mn, ns_eff, slab, st));
    if (db) TRY(reduce_group_add(g, scratch + mn, db, M, nullptr, M, ns_eff, slab, st));
  }
  return grp ? DGVIT_OK : reduce_group_flush(local, st);
}

struct Dims {
  int B, P, N, D, I, M, L, H, dh, pd, pool_mean;
  int proj;       // 0: heads == 1 and dim_head == dim -- the reference's Attention has no output projection (to_out = nn.Identity(), GoalFormer.py:56,66-69)
  long long T;
};

// scratch of the in-launch split-K GEMMs (gemm.hip): fp32 partial tiles + one arrival counter per output tile
struct SplitNeed {
  long long slab = 0;
  long long tiles = 0;
  void take(const GemmSplitPlan& pl) {
    if (pl.nsplit > 1) {
      slab = std::max(slab, pl.slab_floats);
      tiles = std::max<long long>(tiles, pl.tiles);
    }
  }
  void add(int layout, long long M, int N, int K) {
    if (M > 0) take(gemm_split_plan(layout, (int)M, N, K));
  }
  void add_gather(long long M, int N, int K) {     // A gathered from an image (fixed 64 x 64 x 32 tile)
    if (M > 0) take(gemm_split_plan_gather((int)M, N, K));
  }
};
struct SplitBuf {
  int* counters = nullptr; float* slabs = nullptr; long long slab_cap = 0; int ncounters = 0;
  void attach(GemmParams& p) const {
    p.counters = counters; p.slabs = slabs; p.slab_capacity = slab_cap; p.counter_capacity = ncounters;
  }
};

int make_dims(const dgvit_config* c, int batch, Dims& d) {
  DGVIT_CHECK_ARG(c, "null config");
  DGVIT_CHECK_ARG(batch > 0, "batch must be positive");
  DGVIT_CHECK_ARG(c->patch_h > 0 && c->patch_w > 0 && c->image_h > 0 && c->image_w > 0 && c->image_h % c->patch_h == 0 &&
                      c->image_w % c->patch_w == 0,
                  "Image dimensions must be divisible by the patch size.");
  DGVIT_CHECK_ARG(c->dim > 0 && c->dim % 4 == 0 && c->dim <= 1024, "dim=%d must be a multiple of 4 and <= 1024", c->dim);
  DGVIT_CHECK_ARG(c->depth > 0 && c->heads > 0 && c->mlp_dim > 0 && c->mlp_dim % 4 == 0, "bad depth/heads/mlp_dim");
  DGVIT_CHECK_ARG(c->dim_head == 64 || c->dim_head == 32, "dim_head=%d unsupported (64 or 32)", c->dim_head);
  d.B = batch;
  d.P = (c->image_h / c->patch_h) * (c->image_w / c->patch_w);
  d.N = d.P + 1;
  d.D = c->dim; d.H = c->heads; d.dh = c->dim_head; d.I = d.H * d.dh; d.M = c->mlp_dim; d.L = c->depth;
  d.pd = c->patch_h * c->patch_w;
  d.pool_mean = c->pool_mean ? 1 : 0;
  d.proj = !(d.H == 1 && d.dh == d.D);
  d.T = (long long)batch * d.N;
  DGVIT_CHECK_ARG(d.N <= 288, "tokens N=%d exceeds the fused-attention limit (288)", d.N);
  DGVIT_CHECK_ARG(d.T < (1ll << 31) && d.T * (long long)(3 * d.I > d.M ? 3 * d.I : d.M) < (1ll << 40), "batch too large");
  return DGVIT_OK;
}

SplitNeed forward_split_need(const Dims& d) {
  SplitNeed n;
  const int T = (int)d.T;
  n.add(GEMM_NT, (long long)d.B * d.P, d.D, d.pd);
  n.add_gather((long long)d.B * d.P, d.D, d.pd);
  n.add(GEMM_NT, T, 3 * d.I, d.D); n.add(GEMM_NT, T, 2 * d.I, d.D); n.add(GEMM_NT, d.B, d.I, d.D);
  for (int tok : {T, d.B}) {
    n.add(GEMM_NT, tok, d.D, d.I); n.add(GEMM_NT, tok, d.M, d.D); n.add(GEMM_NT, tok, d.D, d.M);
  }
  return n;
}
SplitNeed backward_split_need(const Dims& d) {
  SplitNeed n;
  const int T = (int)d.T;
  for (int tok : {T, d.B}) {
    n.add(GEMM_NN, tok, d.M, d.D); n.add(GEMM_NN, tok, d.D, d.M); n.add(GEMM_NN, tok, d.I, d.D); n.add(GEMM_NN, tok, d.D, d.I);
  }
  n.add(GEMM_NN, T, d.D, 3 * d.I); n.add(GEMM_NN, T, d.D, 2 * d.I);
  return n;
}

// activation workspace carve-up (floats); `save` keeps per-layer buffers distinct
struct Ws {
  long long patches, x0, pooled, layer0, layer_stride, layer_floats, total;
  long long sk_counters, sk_slabs, sk_ncounters, sk_slab_floats;   // in-launch split-K scratch (0 floats when no GEMM splits)
  long long bp_counters, bp_slabs, bp_ncounters;                    // combine scratch of the two-launch small-batch blocks (block.hip; inference only)
  // per-layer offsets relative to the layer base
  long long mean1, rstd1, ln1, qkv, ao, lse, xmid, mean2, rstd2, ln2, h1, a1, xout;
};

// no-grad forwards of a few frames take the two-launch blocks of block.hip.  Where they win, measured against the seven-launch GEMM
// schedule as captured single-launch graphs of policy.sample() (tools/small_batch_ab.py, profiles/r04_*_small_batch_ab.txt): while the
// attention kernel's workgroups (frames x heads x query tiles) fit the chip one per CU -- 0.68-0.76 of the GEMM schedule's time at D = 64
// for 1-16 frames, 0.75-0.89 at D = 128 for 1-32 frames; parity at 32 frames of the shipped model, slower beyond -- and, at D = 256, where
// a workgroup's projections (contraction over 256 on ONE CU) outweigh the saved launches, for a single frame only (0.95; 1.1-1.5 beyond).
bool block_path_eligible(const Dims& d) {
  if (!d.proj || d.T > g_block_path_max_rows || !block_path_supports(d.B, d.N, d.D, d.H, d.dh, d.M)) return false;
  KNOB_IF(g_block_path == 2) return true;      // diagnostic build: every supported shape (tests of ragged row tiles, many frames)
  const long long items = (long long)d.B * d.H * ((d.N + 31) / 32);
  return d.D <= 128 ? items <= 256 : items <= 16;
}

Ws make_ws(const Dims& d, int save) {
  Ws w;
  long long o = 0;
  w.patches = o; o += al4((long long)d.B * d.P * d.pd);
  w.x0 = o; o += al4(d.T * d.D);
  w.pooled = o; o += al4((long long)d.B * d.D);   // token mean (pool='mean' only)
  const SplitNeed sn = forward_split_need(d);
  w.sk_ncounters = sn.tiles; w.sk_slab_floats = sn.slab;
  w.sk_counters = o; o += al4(sn.tiles);
  // (the small-batch blocks' arrival counters sit right behind the split-K ones: ONE memset zeroes both)
  w.bp_counters = o; w.bp_ncounters = 0;
  const bool blocks = !save && block_path_eligible(d);
  if (blocks) {
    w.bp_ncounters = block_path_counters(d.B, d.N);
    o += al4(w.bp_ncounters);
  }
  w.sk_slabs = o; o += al4(sn.slab);
  w.bp_slabs = o;
  if (blocks) o += al4(block_path_slab_floats(d.B, d.N, d.D, d.H, d.M));
  long long l = 0;
  w.mean1 = l; l += al4(d.T);
  w.rstd1 = l; l += al4(d.T);
  w.ln1 = l; l += al4(d.T * d.D);
  w.qkv = l; l += al4(d.T * 3 * d.I);
  w.ao = l; l += al4(d.T * d.I);
  w.lse = l; l += al4((long long)d.B * d.H * d.N);   // base-2 log-sum-exp of every attention row
  w.xmid = l; l += al4(d.T * d.D);
  w.mean2 = l; l += al4(d.T);
  w.rstd2 = l; l += al4(d.T);
  w.ln2 = l; l += al4(d.T * d.D);
  w.h1 = l; l += al4(d.T * d.M);
  w.a1 = l; l += al4(d.T * d.M);
  w.xout = l; l += al4(d.T * d.D);
  w.layer0 = o;
  w.layer_floats = l;
  if (save) {
    w.layer_stride = l;
    o += l * d.L;
  } else {
    // inference: one shared set of temporaries; odd layers write their output into one extra
    // residual-stream buffer placed right behind it, even layers into the shared `xout`
    w.layer_stride = 0;
    long long region = l + al4(d.T * d.D);
#ifdef DGVIT_DIAG
    if (frame_path_supports(d.B, d.N, d.D, d.H, d.dh, d.M)) region = std::max(region, al4(frame_path_scratch_floats(d.B, d.N, d.D, d.H, d.M)));
#endif
    o += region;
  }
  w.total = o;
  return w;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- misc exports
extern "C" int dgvit_abi_version(void) { return DGVIT_ABI_VERSION; }
extern "C" int dgvit_config_size(void) { return (int)sizeof(dgvit_config); }
extern "C" const char* dgvit_last_error(void) { return g_err; }
extern "C" int dgvit_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}
#ifdef DGVIT_DIAG   // include/dgvit_hip_diag.h: libdgvit_hip_diag.so only
extern "C" void dgvit_set_gemm_tile(int tile) { g_gemm_tile_hint = tile; }
extern "C" void dgvit_set_grouped_reduce(int on) { g_group_reduce = on ? 1 : 0; }
extern "C" void dgvit_set_conv_gather(int on) { g_conv_gather = on ? 1 : 0; }
extern "C" void dgvit_set_ln_fusion(int on) { g_ln_fusion = on ? 1 : 0; }
extern "C" void dgvit_set_gemm_split(int on) { g_gemm_split = on ? 1 : 0; }
extern "C" void dgvit_set_gemm_stamps(long long* stamps, int workgroups) {
  g_gemm_stamps = stamps;
  g_gemm_stamp_capacity = stamps ? workgroups : 0;
}
extern "C" long long dgvit_gemm_persistent_launches(void) { return g_gemm_persist_launches; }
extern "C" void dgvit_set_gemm_persistent(int mode, int workgroups) {
  g_gemm_persist = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  g_gemm_persist_grid = workgroups > 0 ? workgroups : 0;
}
extern "C" void dgvit_set_gemm_diagnostics(int on) { g_gemm_diag = on & 0x3FFFFF; }
extern "C" void dgvit_set_gemm_lds_pad(int bytes) { g_gemm_lds_pad = bytes > 0 ? bytes : 0; }
extern "C" void dgvit_set_small_batch_path(int on, int max_rows) {
  g_small_path = on ? 1 : 0;
  if (max_rows > 0) g_small_path_max_rows = max_rows;
}
extern "C" void dgvit_set_block_path(int on, int max_rows) {
  g_block_path = on < 0 ? 0 : (on > 2 ? 2 : on);
  g_block_path_max_rows = max_rows > 0 ? max_rows : 4160;
}
extern "C" void dgvit_set_block_stamps(long long* stamps) { g_block_stamps = stamps; }
extern "C" void dgvit_set_block_stamp_layer(int layer) { g_block_stamp_layer = layer; }
extern "C" void dgvit_set_gelu_grad_store(int on) { g_gelu_grad_store = on ? 1 : 0; }
extern "C" void dgvit_set_block_fuse(int bits) { g_block_fuse = bits & 3; }
extern "C" void dgvit_set_gemm_bf16_tile(int tile) { g_gemm_bf16_tile_hint = tile; }
extern "C" void dgvit_set_gemm_bf16_mfma16(int on) { g_gemm_bf16_m16 = on ? 1 : 0; }
extern "C" void dgvit_set_attention_bwd_single_pass(int on) { g_attn_bwd64 = on ? 1 : 0; }
extern "C" void dgvit_set_attention_single_query(int on) { g_attn_q1 = on ? 1 : 0; }
extern "C" void dgvit_set_gemm_wgrad_slice_major(int on) { g_gemm_zfold = on ? 1 : 0; }
extern "C" int dgvit_attention_forward_queries(const float* qkv, float* out, float* lse, int B, int N, int H, int dh, int nq, void* stream) {
  return attention_fwd(qkv, out, lse, B, N, H, dh, nq, (hipStream_t)stream);
}
extern "C" int dgvit_attention_backward_queries(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                                int B, int N, int H, int dh, int nq, void* stream) {
  return attention_bwd(qkv, out, dout, lse, dqkv, B, N, H, dh, nq, (hipStream_t)stream);
}
extern "C" void dgvit_set_gemm_bf16_group_m(int rows) { g_gemm_bf16_group_m = rows > 0 ? rows : 8; }
extern "C" void dgvit_set_gemm_bf16_stamps(long long* stamps) { g_gemm_bf16_stamps = stamps; }
extern "C" void dgvit_set_gemm_bf16_l2_budget_kb(int kb) { g_gemm_bf16_l2_budget_kb = kb > 0 ? kb : 0; }
#endif   // DGVIT_DIAG

// ---------------------------------------------------------------------------------------------- encoder
extern "C" long long dgvit_got_workspace_floats(const dgvit_config* cfg, int batch, int save) {
  Dims d;
  if (make_dims(cfg, batch, d)) return -1;
  return make_ws(d, save).total;
}

namespace {
struct Bs {  // backward scratch carve-up
  long long dxa, dxb, dln, dqkv, dao, dh1, part, part_ln2, part_ln1, slabs, total, slabs_floats;
  long long sl_fc2, sl_fc1, sl_out, sl_qkv, n_fc2, n_fc1, n_out, n_qkv;   // a layer's four weight gradients keep separate slab regions
  long long sk_counters, sk_slabs, sk_ncounters, sk_slab_floats;          // in-launch split-K scratch of the data-gradient GEMMs
};
Bs make_bs(const Dims& d) {
  Bs s;
  long long o = 0;
  s.dxa = o; o += al4(d.T * d.D);
  s.dxb = o; o += al4(d.T * d.D);
  s.dln = o; o += al4(d.T * d.D);
  s.dqkv = o; o += al4(d.T * 3 * d.I);
  s.dao = o; o += al4(d.T * d.I);
  s.dh1 = o; o += al4(d.T * d.M);
  // reduction partials: LN (blocks*2*D), colsum (blocks*max width), rms, dpos (blocks * N*D)
  long long part = (long long)layernorm_bwd_blocks((int)d.T) * 2 * d.D;
  const long long widest = (long long)(d.M > 3 * d.I ? d.M : 3 * d.I);
  const long long cs = (long long)colsum_blocks((int)d.T) * widest;
  if (cs > part) part = cs;
  const long long dp = (long long)colsum_blocks(d.B) * d.N * d.D;
  if (dp > part) part = dp;
  const long long rp = (long long)rmsnorm_bwd_blocks(d.B) * d.D;
  if (rp > part) part = rp;
  s.part = o; o += al4(part);
  // the two LayerNorm backward passes of a layer keep their dgamma / dbeta partials until the layer's ONE grouped reduction
  const long long lnp = al4((long long)layernorm_bwd_blocks((int)d.T) * 2 * d.D);
  s.part_ln2 = o; o += lnp;
  s.part_ln1 = o; o += lnp;
  // ... and so do its four split-K weight gradients (the last block's to_qkv gradient is two GEMMs: Q rows, K/V rows)
  s.n_fc2 = wgrad_scratch(d.D, d.M, (int)d.T);
  s.n_fc1 = wgrad_scratch(d.M, d.D, (int)d.T);
  s.n_out = wgrad_scratch(d.D, d.I, (int)d.T);
  s.n_qkv = std::max(wgrad_scratch(3 * d.I, d.D, (int)d.T), wgrad_scratch(d.I, d.D, d.B) + wgrad_scratch(2 * d.I, d.D, (int)d.T));
  s.sl_fc2 = 0; s.sl_fc1 = s.n_fc2; s.sl_out = s.sl_fc1 + s.n_fc1; s.sl_qkv = s.sl_out + s.n_out;
  long long sl = s.sl_qkv + s.n_qkv;
  sl = std::max(sl, wgrad_scratch(d.D, d.pd, d.B * d.P));
  s.slabs = o; s.slabs_floats = sl; o += sl;
  const SplitNeed sn = backward_split_need(d);
  s.sk_ncounters = sn.tiles; s.sk_slab_floats = sn.slab;
  s.sk_counters = o; o += al4(sn.tiles);
  s.sk_slabs = o; o += al4(sn.slab);
  s.total = o;
  return s;
}
}  // namespace

extern "C" long long dgvit_got_backward_scratch_floats(const dgvit_config* cfg, int batch) {
  Dims d;
  if (make_dims(cfg, batch, d)) return -1;
  return make_bs(d).total;
}

enum { P_POS = 0, P_PW = 1, P_PB = 2, P_RMS = 3, P_L0 = 4 };
enum { L_LN1W = 0, L_LN1B, L_QKV, L_OUTW, L_OUTB, L_LN2W, L_LN2B, L_FC1W, L_FC1B, L_FC2W, L_FC2B };
// the to_out slots of the parameter table are unused (may be NULL) when the attention has no output projection
static inline bool no_projection_slot(const Dims& d, int i) {
  if (d.proj || i < P_L0) return false;
  const int j = (i - P_L0) % DGVIT_PARAMS_PER_LAYER;
  return j == L_OUTW || j == L_OUTB;
}

extern "C" int dgvit_got_forward(const dgvit_config* cfg, const float* const* params, const float* img, const float* goal,
                                 float* feat, float* ws, long long ws_floats, int batch, int save, float keep,
                                 unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  TRY(make_dims(cfg, batch, d));
  DGVIT_CHECK_ARG(params && img && goal && feat && ws, "dgvit_got_forward: null pointer");
  DGVIT_CHECK_ARG(keep > 0.f && keep <= 1.f, "dropout_keep must be in (0, 1]");
  const Ws w = make_ws(d, save);
  if (ws_floats < w.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "forward workspace %lld < %lld floats", ws_floats, w.total);
  for (int i = 0; i < P_L0 + DGVIT_PARAMS_PER_LAYER * d.L; ++i) DGVIT_CHECK_ARG(params[i] || no_projection_slot(d, i), "parameter %d is null", i);
  const int T = (int)d.T;

  SplitBuf sk;
  if (w.sk_slab_floats > 0) {
    sk.counters = reinterpret_cast<int*>(ws + w.sk_counters); sk.ncounters = (int)w.sk_ncounters;
    sk.slabs = ws + w.sk_slabs; sk.slab_cap = w.sk_slab_floats;
  }
  // Small no-grad batches (SAC.choose_action on one frame, the target passes of learn() on a few frames): two launches per block, the
  // sums over heads / hidden chunks taken inside the launches (block.hip), the LayerNorms in their combine steps
  const bool use_blocks = !save && g_block_path && !g_small_path && w.bp_ncounters > 0;
  // patch embedding (GoalFormer.py:137-139,157) + goal token, positional embedding, dropout (:160-163)
  float* patches = ws + w.patches;
  float* x = ws + w.x0;
  // Inference: the patch rearrangement (GoalFormer.py:138) happens inside the GEMM's A-tile loader -- depth patches go from the
  // image straight into the LDS tiles, no (B * P, pd) copy in HBM.  Training keeps the copy: the weight gradient reads it.
  const int inv = 65536 / cfg->patch_w + 1;
  bool gather = !save && cfg->patch_w % 4 == 0 && cfg->image_w % 4 == 0 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)params[P_PW] & 15) == 0 &&
                (long long)d.pd * inv < (1ll << 31) && (long long)d.B * cfg->image_h * cfg->image_w < (1ll << 29);
  for (int k = 0; gather && k < d.pd; ++k) gather = (int)(((unsigned)k * (unsigned)inv) >> 16) == k / cfg->patch_w;   // exact k / pw
  // ... with the loader gather no GEMM of this call splits, so nothing needs the counters before the first block's attention kernel,
  // which can then zero them itself AND assemble the token rows (goal row, emb-dropout): three launches fewer.  Built, parity-tested
  // and NOT the default (g_block_fuse bit 0, diagnostic build): in one process, graphed sample() of the shipped actor, it measures
  // +14 us for one frame, -3 us for two, +4 us for eight (profiles/r04_c_block_fuse_ab.txt) -- the three launches it removes were not
  // on the critical path the way the in-kernel work that replaces them is.  The RMSNorm fusion (bit 1) is worth 2 us everywhere and stays.
  const bool fused_first = use_blocks && gather && (g_block_fuse & 1);
  // arrival counters of the split GEMMs and of the small-batch blocks (adjacent): every user leaves them zero again
  if (!fused_first && (w.sk_slab_floats > 0 || w.bp_ncounters > 0))
    HIP_TRY(hipMemsetAsync(ws + w.sk_counters, 0, sizeof(int) * (w.bp_counters - w.sk_counters + w.bp_ncounters), st));
  if (!gather) TRY(patchify(img, patches, d.B, cfg->image_h, cfg->image_w, cfg->patch_h, cfg->patch_w, st));
  {
    GemmParams p = gp(patches, d.pd, params[P_PW], d.pd, x, d.D, d.B * d.P, d.D, d.pd);
    p.bias = params[P_PB];
    p.res = params[P_POS]; p.ldr = d.D; p.res_mod = d.P;  // + pos_embedding[1 + patch]
    p.c_rgrp = d.P;                                       // row (b, patch) -> token row b*N + 1 + patch
    if (gather) {
      p.g_img = img; p.g_img_floats = (long long)d.B * cfg->image_h * cfg->image_w;
      p.g_wi = cfg->image_w; p.g_hw = cfg->image_h * cfg->image_w; p.g_ph = cfg->patch_h; p.g_pw = cfg->patch_w;
      p.g_gw = cfg->image_w / cfg->patch_w; p.g_P = d.P; p.g_inv = inv;
    }
    sk.attach(p);
    TRY(gemm_f32(GEMM_NT, EPI_STORE, p, 1, st));
  }
  if (!fused_first) {
    TRY(goal_row(goal, params[P_POS], x, d.B, d.N, d.D, st));
    if (keep < 1.f) TRY(dropout_inplace(x, d.T * d.D, seed, seed_dev, keep, st));
  }

  // inference on a handful of frames (SAC.choose_action, the no-grad passes of learn() at batch 32): two launches per block
#ifdef DGVIT_DIAG   // (measured slower than the schedule below, DESIGN 3.7: not in the product library)
  if (!save && g_small_path && !d.pool_mean && d.proj && d.T <= g_small_path_max_rows && frame_path_supports(d.B, d.N, d.D, d.H, d.dh, d.M))
    return frame_path_forward(x, params, d.L, ws + w.layer0, feat, d.B, d.N, d.D, d.H, d.dh, d.M, st);
#endif

  bool feat_done = false;
  if (use_blocks) {
    int* counters = reinterpret_cast<int*>(ws + w.bp_counters);       // (zeroed above, or by the first attention kernel)
    float* lb = ws + w.layer0;
    BlockFirst first = {};
    first.goal = goal; first.pos0 = params[P_POS]; first.xres = lb + w.xmid;   // (xmid: free in this path) the assembled, dropped-out token rows
    first.keep = keep; first.seed = seed; first.seed_dev = seed_dev;
    for (int i = 0; i < d.L; ++i) {
      const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
      float* xo = !(i & 1) ? lb + w.xout : ws + w.layer0 + w.layer_floats;
      const bool last = !dense_last_block(cfg) && !d.pool_mean && i == d.L - 1;
      const float* next_ln[2] = {nullptr, nullptr};
      if (i + 1 < d.L) {
        next_ln[0] = params[P_L0 + DGVIT_PARAMS_PER_LAYER * (i + 1) + L_LN1W];
        next_ln[1] = params[P_L0 + DGVIT_PARAMS_PER_LAYER * (i + 1) + L_LN1B];
      }
      DGVIT_DIAG_ONLY(g_block_stamp_now = g_block_stamp_layer < 0 || g_block_stamp_layer == i;)
      // (block 0 normalises its input inside the attention kernel; later blocks read the rows the previous MLP kernel normalised;
      //  the pruned last block's MLP kernel also applies the final RMSNorm to the pooled rows: GoalFormer.py:167-170)
      TRY(block_path_layer(x, i == 0 ? nullptr : lb + w.ln1, xo, lb + w.ln1, lp, i + 1 < d.L ? next_ln : nullptr, last ? 1 : 0, ws + w.bp_slabs,
                           counters, i == 0 && fused_first ? &first : nullptr, last && (g_block_fuse & 2) ? params[P_RMS] : nullptr, last ? feat : nullptr, d.B, d.N,
                           d.D, d.H, d.dh, d.M, st));
      feat_done = last && (g_block_fuse & 2);
      x = xo;
    }
  }
  if (feat_done) return DGVIT_OK;
  const bool ln_fused = g_ln_fusion && d.D == 64 && g_gemm_tile_hint == 0;   // (the automatic tile for N = 64 is 64 wide)
  for (int i = 0; i < d.L && !use_blocks; ++i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    float* lb = ws + w.layer0 + w.layer_stride * i;
    float* xo = (save || !(i & 1)) ? lb + w.xout : ws + w.layer0 + w.layer_floats;
    // The output only reads token 0 of the last block (GoalFormer.py:167): there, K and V are needed for every
    // token but Q, the attention output, to_out and the whole feed-forward only for row b*N of each frame.
    // `tok` = rows processed, `rs` = row step (in token rows) of those rows inside the (T, .) buffers.
    const bool last = !dense_last_block(cfg) && !d.pool_mean && i == d.L - 1;
    const int tok = last ? d.B : T, rs = last ? d.N : 1;
    // x = attn(LN(x)) + x   (GoalFormer.py:103, 36-37, 71-82)
    // D <= 64 (the shipped model): a 64-wide GEMM tile holds whole rows of the residual stream, so each LayerNorm runs inside the
    // epilogue of the GEMM that produces its input (to_out -> LN2, fc2 -> the next block's LN1; bit-identical to the LayerNorm
    // kernel).  Only the first block's LN1 is a launch of its own: 8 -> 1 LayerNorm launches in the shipped 4-block encoder.
    if (!(ln_fused && i > 0))
      TRY(layernorm_fwd(x, lp[L_LN1W], lp[L_LN1B], lb + w.ln1, lb + w.mean1, lb + w.rstd1, T, d.D, 1e-5f, 1, st));
    if (!last) {
      GemmParams p = gp(lb + w.ln1, d.D, lp[L_QKV], d.D, lb + w.qkv, 3 * d.I, T, 3 * d.I, d.D);
      sk.attach(p);
      TRY(gemm_f32(GEMM_NT, EPI_STORE, p, 1, st));
    } else {
      GemmParams kv = gp(lb + w.ln1, d.D, lp[L_QKV] + (long long)d.I * d.D, d.D, lb + w.qkv + d.I, 3 * d.I, T, 2 * d.I, d.D);
      sk.attach(kv);
      TRY(gemm_f32(GEMM_NT, EPI_STORE, kv, 1, st));
      GemmParams q = gp(lb + w.ln1, rs * d.D, lp[L_QKV], d.D, lb + w.qkv, rs * 3 * d.I, tok, d.I, d.D);
      sk.attach(q);
      TRY(gemm_f32(GEMM_NT, EPI_STORE, q, 1, st));
    }
    TRY(attention_fwd(lb + w.qkv, lb + w.ao, save ? lb + w.lse : nullptr, d.B, d.N, d.H, d.dh, last ? 1 : d.N, st));
    if (!d.proj) {
      // to_out = nn.Identity() (GoalFormer.py:56,66-69): the head's output IS the branch output (I == D): xmid = attn + x (:103)
      TRY(add_rows(lb + w.ao, (long long)rs * d.I, x, (long long)rs * d.D, lb + w.xmid, (long long)rs * d.D, tok, d.D, st));
    } else {
      GemmParams p = gp(lb + w.ao, rs * d.I, lp[L_OUTW], d.I, lb + w.xmid, rs * d.D, tok, d.D, d.I);
      p.bias = lp[L_OUTB]; p.res = x; p.ldr = rs * d.D;
      if (ln_fused) {
        p.ln_g = lp[L_LN2W]; p.ln_b = lp[L_LN2B]; p.ln_y = lb + w.ln2; p.ln_ld = (long long)rs * d.D;
        p.ln_mean = lb + w.mean2; p.ln_rstd = lb + w.rstd2; p.ln_eps = 1e-5f;
      }
      sk.attach(p);
      TRY(gemm_f32(GEMM_NT, EPI_STORE, p, 1, st));
    }
    // x = ff(LN(x)) + x     (GoalFormer.py:104, 42-50)
    if (!ln_fused || !d.proj) TRY(layernorm_fwd(lb + w.xmid, lp[L_LN2W], lp[L_LN2B], lb + w.ln2, lb + w.mean2, lb + w.rstd2, tok, d.D, 1e-5f, rs, st));
    {
      // training: a1 = gelu(t) for fc2 and the weight gradient, and -- in the h1 slot -- gelu'(t), the factor the data gradient of fc2
      // multiplies by (the erf form already holds its exponential; the backward epilogue then evaluates nothing).  No-grad passes
      // store a1 only: the pre-activation (210 MB per layer at C3) is never written.
      GemmParams p = gp(lb + w.ln2, rs * d.D, lp[L_FC1W], d.D, save ? lb + w.h1 : lb + w.a1, d.M, tok, d.M, d.D);   // h1 / a1 are dense (tok, M)
      p.bias = lp[L_FC1B];
      if (save) { p.C2 = lb + w.a1; p.ldc2 = d.M; }
      sk.attach(p);
      TRY(gemm_f32(GEMM_NT, save ? (g_gelu_grad_store ? EPI_GELU2D : EPI_GELU2) : EPI_GELU, p, 1, st));
    }
    {
      GemmParams p = gp(lb + w.a1, d.M, lp[L_FC2W], d.M, xo, rs * d.D, tok, d.D, d.M);
      p.bias = lp[L_FC2B]; p.res = lb + w.xmid; p.ldr = rs * d.D;
      if (ln_fused && i + 1 < d.L) {   // the next block's LN1 (this block is never the pruned last one: all T rows)
        const float* const* ln = params + P_L0 + DGVIT_PARAMS_PER_LAYER * (i + 1);
        float* nb = ws + w.layer0 + w.layer_stride * (i + 1);
        p.ln_g = ln[L_LN1W]; p.ln_b = ln[L_LN1B]; p.ln_y = nb + w.ln1; p.ln_ld = d.D;
        p.ln_mean = nb + w.mean1; p.ln_rstd = nb + w.rstd1; p.ln_eps = 1e-5f;
      }
      sk.attach(p);
      TRY(gemm_f32(GEMM_NT, EPI_STORE, p, 1, st));
    }
    x = xo;
  }
  // pool: x[:, 0] (cls slot = goal token) or the token mean (GoalFormer.py:167), then RMSNorm (:170)
  if (d.pool_mean) {
    float* pooled = ws + w.pooled;
    TRY(avgpool(x, pooled, d.B, d.N, d.D, st));
    return rmsnorm_fwd(pooled, d.D, params[P_RMS], feat, d.B, d.D, st);
  }
  return rmsnorm_fwd(x, (long long)d.N * d.D, params[P_RMS], feat, d.B, d.D, st);
}

// gradient-ready events (dgvit_grad_events): recorded on the caller's stream where a group of parameter gradients is final
static int check_events(const dgvit_grad_events* ev, int depth) {
  if (!ev) return DGVIT_OK;
  DGVIT_CHECK_ARG(ev->n_layers == depth, "dgvit_grad_events: n_layers %d != depth %d", ev->n_layers, depth);
  DGVIT_CHECK_ARG(ev->layer, "dgvit_grad_events: layer table is null");
  return DGVIT_OK;
}
static int mark_ready(void* event, hipStream_t st) {
  if (event) HIP_TRY(hipEventRecord((hipEvent_t)event, st));
  return DGVIT_OK;
}

extern "C" int dgvit_event_create(void** event) {
  DGVIT_CHECK_ARG(event, "dgvit_event_create: null pointer");
  hipEvent_t e;
  HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = (void*)e;
  return DGVIT_OK;
}
extern "C" int dgvit_event_destroy(void* event) {
  if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return DGVIT_OK;
}
extern "C" int dgvit_stream_wait_event(void* stream, void* event) {
  DGVIT_CHECK_ARG(event, "dgvit_stream_wait_event: null event");
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return DGVIT_OK;
}

extern "C" int dgvit_got_backward(const dgvit_config* cfg, const float* const* params, float* const* grads, const float* dfeat,
                                  float* dgoal, const float* ws, long long ws_floats, float* scratch, long long scratch_floats,
                                  int batch, float keep, unsigned long long seed, const unsigned long long* seed_dev,
                                  void* stream) {
  return dgvit_got_backward_ev(cfg, params, grads, dfeat, dgoal, ws, ws_floats, scratch, scratch_floats, batch, keep, seed, seed_dev, stream,
                               nullptr);
}

extern "C" int dgvit_got_backward_ev(const dgvit_config* cfg, const float* const* params, float* const* grads, const float* dfeat,
                                     float* dgoal, const float* ws, long long ws_floats, float* scratch, long long scratch_floats,
                                     int batch, float keep, unsigned long long seed, const unsigned long long* seed_dev,
                                     void* stream, const dgvit_grad_events* events) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  TRY(make_dims(cfg, batch, d));
  DGVIT_CHECK_ARG(params && grads && dfeat && ws && scratch, "dgvit_got_backward: null pointer");
  TRY(check_events(events, cfg->depth));
  const Ws w = make_ws(d, 1);
  const Bs s = make_bs(d);
  if (ws_floats < w.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "backward workspace %lld < %lld floats", ws_floats, w.total);
  if (scratch_floats < s.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "backward scratch %lld < %lld floats", scratch_floats, s.total);
  const int np = P_L0 + DGVIT_PARAMS_PER_LAYER * d.L;
  for (int i = 0; i < np; ++i) DGVIT_CHECK_ARG(params[i] || no_projection_slot(d, i), "parameter %d is null", i);   // grads[i] == NULL: frozen parameter, its gradient is skipped
  const int T = (int)d.T;
  float* dx = scratch + s.dxa;    // gradient of the residual stream entering the current op
  float* dx2 = scratch + s.dxb;
  float* dln = scratch + s.dln;
  float* dqkv = scratch + s.dqkv;
  float* dao = scratch + s.dao;
  float* dh1 = scratch + s.dh1;
  float* part = scratch + s.part;
  float* slabs = scratch + s.slabs;

  SplitBuf sk;   // the data-gradient GEMMs all run on the caller's stream, one after the other: one counter / slab region serves them
  if (s.sk_slab_floats > 0) {
    sk.counters = reinterpret_cast<int*>(scratch + s.sk_counters); sk.ncounters = (int)s.sk_ncounters;
    sk.slabs = scratch + s.sk_slabs; sk.slab_cap = s.sk_slab_floats;
    HIP_TRY(hipMemsetAsync(sk.counters, 0, sizeof(int) * s.sk_ncounters, st));
  }
  // weight gradients run on the helper stream `sw`; `done[j]` = main must wait for the previous layer's j-th
  // wgrad before overwriting the buffer it reads (dx, dh1, dx2, dqkv)
  hipStream_t sw = st;
  if (wgrad_overlap(cfg)) {
    TRY(side_init());
    sw = g_side.stream;
    TRY(chain(st, sw));   // helper starts after everything already queued by the caller
  }
  auto fork = [&]() -> int { return sw == st ? DGVIT_OK : chain(st, sw); };
  auto join = [&]() -> int { return sw == st ? DGVIT_OK : chain(sw, st); };

  // RMSNorm on token 0 of the last layer's output; every other token row gets zero gradient
  const float* xl = ws + w.layer0 + w.layer_stride * (d.L - 1) + w.xout;
  if (d.pool_mean) {
    // feat = RMSNorm(mean_tokens(x)): gradient of the pooled vector (into dln as scratch), then broadcast / N
    TRY(rmsnorm_bwd(dfeat, ws + w.pooled, d.D, params[P_RMS], dln, d.D, grads[P_RMS], part, d.B, d.D, st));
    TRY(mean_bwd(dln, dx, d.B, d.N, d.D, st));
  } else {
    HIP_TRY(hipMemsetAsync(dx, 0, sizeof(float) * d.T * d.D, st));
    TRY(rmsnorm_bwd(dfeat, xl, (long long)d.N * d.D, params[P_RMS], dx, (long long)d.N * d.D, grads[P_RMS], part, d.B, d.D, st));
  }
  if (events) TRY(mark_ready(events->head, st));

  for (int i = d.L - 1; i >= 0; --i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    float* const* lg = grads + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    const float* lb = ws + w.layer0 + w.layer_stride * i;
    const float* xin = i == 0 ? ws + w.x0 : ws + w.layer0 + w.layer_stride * (i - 1) + w.xout;
    const bool last = !dense_last_block(cfg) && !d.pool_mean && i == d.L - 1;   // see dgvit_got_forward: only rows b*N carry gradient here
    const int tok = last ? d.B : T, rs = last ? d.N : 1;
    // ---- feed-forward branch: xout = fc2(gelu(fc1(ln2))) + xmid
    // (helper-stream kernels are ordered among themselves, so the slab scratch is reused safely; a `join` before
    //  a main-stream kernel that overwrites a buffer makes sure the wgrads that read it have finished)
    // (every weight gradient of the layer writes its split-K slabs into a region of its own; their fixed-order sums and the
    //  two LayerNorm parameter-gradient sums are ONE grouped launch at the end of the layer instead of six)
    ReduceGroup grp;
    reduce_group_init(grp);
    ReduceGroup* gq = g_group_reduce ? &grp : nullptr;   // null: every reduction is launched where it is produced
    TRY(fork());
    TRY(wgrad(dx, rs * d.D, lb + w.a1, d.M, lg[L_FC2W], lg[L_FC2B], d.D, d.M, tok, slabs + s.sl_fc2, s.n_fc2, sw, gq));
    {
      GemmParams p = gp(dx, rs * d.D, lp[L_FC2W], d.M, dh1, d.M, tok, d.M, d.D);
      p.aux = lb + w.h1; p.ldaux = d.M;             // (the h1 slot holds gelu'(pre-activation), written by the forward)
      sk.attach(p);
      TRY(gemm_f32(GEMM_NN, g_gelu_grad_store ? EPI_DMUL : EPI_DGELU, p, 1, st));   // dh1 = (dx W2) * gelu'(h1)   [previous layer's wgrads joined below]
    }
    TRY(fork());
    TRY(wgrad(dh1, d.M, lb + w.ln2, rs * d.D, lg[L_FC1W], lg[L_FC1B], d.M, d.D, tok, slabs + s.sl_fc1, s.n_fc1, sw, gq));
    {
      GemmParams p = gp(dh1, d.M, lp[L_FC1W], d.D, dln, rs * d.D, tok, d.D, d.M);
      sk.attach(p);
      TRY(gemm_f32(GEMM_NN, EPI_STORE, p, 1, st));  // dln2 = dh1 W1
    }
    if (last) HIP_TRY(hipMemsetAsync(dx2, 0, sizeof(float) * d.T * d.D, st));   // rows other than b*N get no gradient
    TRY(layernorm_bwd(dln, lb + w.xmid, lb + w.mean2, lb + w.rstd2, lp[L_LN2W], dx, dx2, lg[L_LN2W], lg[L_LN2B], scratch + s.part_ln2, tok, d.D,
                      rs, st, gq));
    // ---- attention branch: xmid = to_out(attn(to_qkv(ln1))) + xin       (dx2 = d xmid)
    if (d.proj) {
      TRY(fork());
      TRY(wgrad(dx2, rs * d.D, lb + w.ao, rs * d.I, lg[L_OUTW], lg[L_OUTB], d.D, d.I, tok, slabs + s.sl_out, s.n_out, sw, gq));
      GemmParams p = gp(dx2, rs * d.D, lp[L_OUTW], d.I, dao, rs * d.I, tok, d.I, d.D);
      sk.attach(p);
      TRY(gemm_f32(GEMM_NN, EPI_STORE, p, 1, st));  // dao = dxmid Wo
    }
    // (no output projection: the gradient of the attention output is the residual-stream gradient itself, I == D)
    TRY(attention_bwd(lb + w.qkv, lb + w.ao, d.proj ? dao : dx2, lb + w.lse, dqkv, d.B, d.N, d.H, d.dh, last ? 1 : d.N, st));
    TRY(fork());
    if (!last) {
      TRY(wgrad(dqkv, 3 * d.I, lb + w.ln1, d.D, lg[L_QKV], nullptr, 3 * d.I, d.D, T, slabs + s.sl_qkv, s.n_qkv, sw, gq));
      GemmParams p = gp(dqkv, 3 * d.I, lp[L_QKV], d.D, dln, d.D, T, d.D, 3 * d.I);
      sk.attach(p);
      TRY(gemm_f32(GEMM_NN, EPI_STORE, p, 1, st));  // dln1 = dqkv Wqkv
    } else {
      // dWq from the token-0 rows, dWk/dWv from all rows; dln1 = dkv Wkv (+ dq Wq on the token-0 rows)
      const long long nq_slabs = wgrad_scratch(d.I, d.D, tok);
      TRY(wgrad(dqkv, rs * 3 * d.I, lb + w.ln1, rs * d.D, lg[L_QKV], nullptr, d.I, d.D, tok, slabs + s.sl_qkv, nq_slabs, sw, gq));
      TRY(wgrad(dqkv + d.I, 3 * d.I, lb + w.ln1, d.D, lg[L_QKV] ? lg[L_QKV] + (long long)d.I * d.D : nullptr, nullptr, 2 * d.I, d.D, T,
                slabs + s.sl_qkv + nq_slabs, s.n_qkv - nq_slabs, sw, gq));
      GemmParams kv = gp(dqkv + d.I, 3 * d.I, lp[L_QKV] + (long long)d.I * d.D, d.D, dln, d.D, T, d.D, 2 * d.I);
      sk.attach(kv);
      TRY(gemm_f32(GEMM_NN, EPI_STORE, kv, 1, st));
      GemmParams q = gp(dqkv, rs * 3 * d.I, lp[L_QKV], d.D, dln, rs * d.D, tok, d.D, d.I);
      q.res = dln; q.ldr = rs * d.D;
      sk.attach(q);
      TRY(gemm_f32(GEMM_NN, EPI_STORE, q, 1, st));
    }
    // dx, dh1, dx2 and dqkv are overwritten from here on (this LayerNorm backward and the next layer): wait for the
    // helper stream.  Only this layer's last wgrad (qkv) can still be running; it overlapped the dln1 GEMM above.
    TRY(join());
    TRY(layernorm_bwd(dln, xin, lb + w.mean1, lb + w.rstd1, lp[L_LN1W], dx2, dx, lg[L_LN1W], lg[L_LN1B], scratch + s.part_ln1, T, d.D, 1, st, gq));
    // the layer's grouped reduction: behind the weight gradients on the helper stream (it also reads this stream's LayerNorm
    // partials, hence the fork), and the caller's stream waits for it before the slab / partial regions are written again
    TRY(fork());
    TRY(reduce_group_flush(grp, sw));
    TRY(join());
    if (events) TRY(mark_ready(events->layer[i], st));    // every gradient of block i is final in stream order
  }
  // ---- token assembly: x0 = dropout(cat(goal, patches W^T + b) + pos)
  if (keep < 1.f) TRY(dropout_inplace(dx, d.T * d.D, seed, seed_dev, keep, st));
  if (dgoal)
    HIP_TRY(hipMemcpy2DAsync(dgoal, sizeof(float) * d.D, dx, sizeof(float) * d.N * d.D, sizeof(float) * d.D, d.B,
                             hipMemcpyDeviceToDevice, st));
  if (grads[P_POS]) TRY(colsum(dx, (long long)d.N * d.D, grads[P_POS], part, d.B, d.N * d.D, 0, st));  // dpos = sum over frames
  if (!grads[P_PW] && !grads[P_PB]) return DGVIT_OK;
  // patch rows of dx0 (token rows 1..P of every frame) packed densely, then dW_pe = dx_patch^T patches, db_pe = column sums
  HIP_TRY(hipMemcpy2DAsync(dln, sizeof(float) * d.P * d.D, dx + d.D, sizeof(float) * d.N * d.D, sizeof(float) * d.P * d.D, d.B,
                           hipMemcpyDeviceToDevice, st));
  TRY(wgrad(dln, d.D, ws + w.patches, d.pd, grads[P_PW], grads[P_PB], d.D, d.pd, d.B * d.P, slabs, s.slabs_floats, st));
  return DGVIT_OK;
}

// ---------------------------------------------------------------------------------------------- head Linears
extern "C" int dgvit_linear_forward(const float* x, const float* wt, const float* b, float* y, int M, int N, int K, int act,
                                    void* stream) {
  DGVIT_CHECK_ARG(act == 0 || act == 1, "linear: act must be 0 (identity) or 1 (relu)");
  DGVIT_CHECK_ARG(x && wt && y && M > 0 && N > 0 && K > 0, "linear: bad arguments");
  GemmParams p = gp(x, K, wt, K, y, N, M, N, K);
  p.bias = b;
  return gemm_f32(GEMM_NT, act ? EPI_RELU : EPI_STORE, p, 1, (hipStream_t)stream);
}

extern "C" long long dgvit_linear_backward_scratch_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  return al4((long long)M * N) + al4((long long)colsum_blocks(M) * N) + wgrad_scratch(N, K, M);
}

extern "C" int dgvit_linear_backward(const float* dy, const float* x, const float* wt, const float* y, float* dx, float* dw,
                                     float* db, float* scratch, long long scratch_floats, int M, int N, int K, int act,
                                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DGVIT_CHECK_ARG(act == 0 || act == 1, "linear: act must be 0 (identity) or 1 (relu)");
  DGVIT_CHECK_ARG(dy && x && wt && dw && scratch && M > 0 && N > 0 && K > 0, "linear_backward: bad arguments");
  DGVIT_CHECK_ARG(act == 0 || y, "linear_backward: relu needs the forward output");
  const long long need = dgvit_linear_backward_scratch_floats(M, N, K);
  if (scratch_floats < need) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "linear_backward scratch %lld < %lld floats", scratch_floats, need);
  float* dpre = scratch;
  float* part = scratch + al4((long long)M * N);
  float* slabs = part + al4((long long)colsum_blocks(M) * N);
  const float* g = dy;
  if (act == 1) {
    TRY(relu_bwd(dy, y, dpre, (long long)M * N, st));
    g = dpre;
  }
  TRY(wgrad(g, N, x, K, dw, db, N, K, M, slabs, wgrad_scratch(N, K, M), st));
  if (dx) {
    GemmParams p = gp(g, N, wt, K, dx, K, M, K, N);
    TRY(gemm_f32(GEMM_NN, EPI_STORE, p, 1, st));
  }
  return DGVIT_OK;
}

// ---------------------------------------------------------------------------------------------- fused MLP heads
extern "C" int dgvit_mlp_head_forward(const dgvit_mlp_desc* desc, const float* const* in, const float* const* params, float* h1,
                                      float* h2, float* y, void* stream) {
  return mlp_head_forward(desc, in, params, h1, h2, y, (hipStream_t)stream);
}
extern "C" long long dgvit_mlp_head_backward_scratch_floats(const dgvit_mlp_desc* desc) { return mlp_head_backward_scratch(desc); }
extern "C" int dgvit_mlp_head_backward(const dgvit_mlp_desc* desc, const float* const* in, const float* const* params, const float* h1,
                                       const float* h2, const float* const* dy, float* const* din, float* const* dparams, float* scratch,
                                       long long scratch_floats, void* stream) {
  return mlp_head_backward(desc, in, params, h1, h2, dy, din, dparams, scratch, scratch_floats, (hipStream_t)stream);
}

extern "C" int dgvit_tanh_gaussian_forward(const float* mean, const float* log_std_raw, const float* eps, const float* scale,
                                           const float* bias, int scale_n, float ls_min, float ls_max, float* action, float* log_prob,
                                           float* tanh_mean, int B, int A, void* stream) {
  return tanh_gaussian_forward(mean, log_std_raw, eps, scale, bias, scale_n, ls_min, ls_max, action, log_prob, tanh_mean, B, A, (hipStream_t)stream);
}
extern "C" int dgvit_tanh_gaussian_backward(const float* mean, const float* log_std_raw, const float* eps, const float* scale, int scale_n,
                                            float ls_min, float ls_max, const float* d_action, const float* d_log_prob,
                                            const float* d_tanh_mean, float* dmean, float* dlog_std_raw, int B, int A, void* stream) {
  return tanh_gaussian_backward(mean, log_std_raw, eps, scale, scale_n, ls_min, ls_max, d_action, d_log_prob, d_tanh_mean, dmean, dlog_std_raw, B, A,
                                (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- operator exports
extern "C" long long dgvit_gemm_scratch_floats(int layout, int M, int N, int K) {
  if (layout != GEMM_TN) {   // in-launch split-K: arrival counters (one per tile) + partial tiles; 0 when the shape is not split
    const GemmSplitPlan pl = gemm_split_plan(layout, M, N, K);
    return pl.nsplit > 1 ? al4(pl.tiles) + al4(pl.slab_floats) : 0;
  }
  return wgrad_scratch(M, N, K);
}

extern "C" int dgvit_gemm(int layout, int epilogue, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                          int N, int K, const float* bias, const float* res, int ldr, float* C2, int ldc2, const float* aux,
                          int ldaux, float* scratch, long long scratch_floats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (layout == GEMM_TN) {
    DGVIT_CHECK_ARG(epilogue == EPI_STORE && !bias && !res, "gemm: layout TN supports the plain epilogue only");
    DGVIT_CHECK_ARG(ldc == N, "gemm: layout TN writes a dense C (ldc == N)");
    DGVIT_CHECK_ARG(scratch, "gemm: layout TN needs scratch");
    return wgrad(A, lda, B, ldb, C, nullptr, M, N, K, scratch, scratch_floats, st);
  }
  DGVIT_CHECK_ARG(layout == GEMM_NT || layout == GEMM_NN, "gemm: bad layout %d", layout);
  DGVIT_CHECK_ARG(epilogue >= EPI_STORE && epilogue <= EPI_DRELU, "gemm: bad epilogue %d", epilogue);
  DGVIT_CHECK_ARG(epilogue != EPI_GELU2 || C2, "gemm: epilogue 1 needs C2");
  DGVIT_CHECK_ARG((epilogue != EPI_DGELU && epilogue != EPI_DRELU) || aux, "gemm: epilogue needs aux");
  GemmParams p = gp(A, lda, B, ldb, C, ldc, M, N, K);
  p.bias = bias; p.res = res; p.ldr = ldr; p.C2 = C2; p.ldc2 = ldc2; p.aux = aux; p.ldaux = ldaux;
  const GemmSplitPlan pl = gemm_split_plan(layout, M, N, K);
  if (pl.nsplit > 1 && scratch && scratch_floats >= al4(pl.tiles) + al4(pl.slab_floats)) {
    p.counters = reinterpret_cast<int*>(scratch); p.counter_capacity = pl.tiles;
    p.slabs = scratch + al4(pl.tiles); p.slab_capacity = scratch_floats - al4(pl.tiles);
    HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(int) * pl.tiles, st));
  }
  return gemm_f32(layout, epilogue, p, 1, st);
}

extern "C" int dgvit_layernorm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                       int rows, int D, void* stream) {
  return layernorm_fwd(x, gamma, beta, y, mean, rstd, rows, D, 1e-5f, 1, (hipStream_t)stream);
}
extern "C" long long dgvit_layernorm_backward_scratch_floats(int rows, int D) {
  if (rows <= 0 || D <= 0) return -1;
  return (long long)layernorm_bwd_blocks(rows) * 2 * D;
}
extern "C" int dgvit_layernorm_backward(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                        const float* dres, float* dx, float* dgamma, float* dbeta, float* scratch,
                                        long long scratch_floats, int rows, int D, void* stream) {
  if (scratch_floats < dgvit_layernorm_backward_scratch_floats(rows, D))
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "layernorm_backward: scratch too small");
  return layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dgamma, dbeta, scratch, rows, D, 1, (hipStream_t)stream);
}
extern "C" int dgvit_rmsnorm_forward(const float* x, long long ldx, const float* g, float* y, int rows, int D, void* stream) {
  return rmsnorm_fwd(x, ldx, g, y, rows, D, (hipStream_t)stream);
}
extern "C" long long dgvit_rmsnorm_backward_scratch_floats(int rows, int D) {
  if (rows <= 0 || D <= 0) return -1;
  return (long long)rmsnorm_bwd_blocks(rows) * D;
}
extern "C" int dgvit_rmsnorm_backward(const float* dy, const float* x, long long ldx, const float* g, float* dx, long long lddx,
                                      float* dg, float* scratch, long long scratch_floats, int rows, int D, void* stream) {
  if (scratch_floats < dgvit_rmsnorm_backward_scratch_floats(rows, D))
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "rmsnorm_backward: scratch too small");
  return rmsnorm_bwd(dy, x, ldx, g, dx, lddx, dg, scratch, rows, D, (hipStream_t)stream);
}
extern "C" int dgvit_attention_forward(const float* qkv, float* out, float* lse, int B, int N, int H, int dh, void* stream) {
  return attention_fwd(qkv, out, lse, B, N, H, dh, N, (hipStream_t)stream);
}
extern "C" int dgvit_attention_backward(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                        int B, int N, int H, int dh, void* stream) {
  return attention_bwd(qkv, out, dout, lse, dqkv, B, N, H, dh, N, (hipStream_t)stream);
}
extern "C" int dgvit_patchify(const float* img, float* patches, int B, int ih, int iw, int ph, int pw, void* stream) {
  return patchify(img, patches, B, ih, iw, ph, pw, (hipStream_t)stream);
}
extern "C" int dgvit_dropout(float* x, long long n, unsigned long long seed, float keep, void* stream) {
  return dropout_inplace(x, n, seed, nullptr, keep, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- optimiser step
extern "C" int dgvit_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                               float eps, float weight_decay, long long step, const long long* step_dev, void* stream) {
  return adam_step(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, step_dev, (hipStream_t)stream);
}
extern "C" int dgvit_soft_update(float* target, const float* source, long long n, float tau, void* stream) {
  return soft_update(target, source, n, tau, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- CNN feature stack
// (the shipped critic QNetwork and the CNN actor; got_sac_network.py:129-133,151-155)
namespace {
struct ConvDims {
  int B, H[4], W[4], C[4], KP[3];   // layer l maps (H[l], W[l], C[l]) -> (H[l+1], W[l+1], C[l+1])
  long long M[4];                   // rows of the NHWC activation l (M[0] unused)
};
int make_conv_dims(int B, int H, int W, ConvDims& d) {
  DGVIT_CHECK_ARG(B > 0 && H >= 29 && W >= 29, "cnn: need batch > 0 and frames of at least 29x29");
  d.B = B; d.H[0] = H; d.W[0] = W; d.C[0] = 1; d.C[1] = 16; d.C[2] = 64; d.C[3] = 256;
  for (int l = 0; l < 3; ++l) {
    d.H[l + 1] = (d.H[l] - 5) / 2 + 1;
    d.W[l + 1] = (d.W[l] - 5) / 2 + 1;
    d.KP[l] = l == 0 ? 28 : 25 * d.C[l];
    d.M[l + 1] = (long long)B * d.H[l + 1] * d.W[l + 1];
    DGVIT_CHECK_ARG(d.H[l + 1] > 0 && d.W[l + 1] > 0 && d.M[l + 1] < (1ll << 31), "cnn: frame too small or batch too large");
  }
  return DGVIT_OK;
}
long long conv_cols_floats(const ConvDims& d) {
  long long m = 0;
  for (int l = 0; l < 3; ++l) m = std::max(m, d.M[l + 1] * d.KP[l]);
  return al4(m);
}
long long conv_wp_floats(const ConvDims& d) { return al4(16 * 28) + al4(64 * 400) + al4(256 * 1600); }
}  // namespace

extern "C" long long dgvit_cnn_workspace_floats(int B, int H, int W) {
  ConvDims d;
  if (make_conv_dims(B, H, W, d)) return -1;
  return al4(d.M[1] * 16) + al4(d.M[2] * 64) + al4(d.M[3] * 256);
}
// split-K scratch of the forward's implicit-GEMM convolutions (conv2 / conv3; conv1 goes through im2col and is not split)
static SplitNeed conv_split_need(const ConvDims& d) {
  SplitNeed n;
  for (int l = 1; l < 3; ++l) n.add_gather(d.M[l + 1], d.C[l + 1], d.KP[l]);
  return n;
}
static long long conv_split_floats(const ConvDims& d) {
  const SplitNeed n = conv_split_need(d);
  return n.tiles > 0 ? al4(n.tiles) + n.slab : 0;
}
extern "C" long long dgvit_cnn_forward_scratch_floats(int B, int H, int W) {
  ConvDims d;
  if (make_conv_dims(B, H, W, d)) return -1;
  return conv_cols_floats(d) + conv_wp_floats(d) + conv_split_floats(d);
}
extern "C" long long dgvit_cnn_backward_scratch_floats(int B, int H, int W) {
  ConvDims d;
  if (make_conv_dims(B, H, W, d)) return -1;
  long long dy = std::max(std::max(d.M[1] * 16, d.M[2] * 64), d.M[3] * 256);
  long long slabs = std::max(std::max(wgrad_scratch(16, 28, (int)d.M[1]), wgrad_scratch(64, 400, (int)d.M[2])), wgrad_scratch(256, 1600, (int)d.M[3]));
  return conv_cols_floats(d) + 2 * conv_wp_floats(d) + 2 * al4(dy) + slabs;
}

// params: conv1.weight (16,1,5,5), conv1.bias, conv2.weight (64,16,5,5), conv2.bias, conv3.weight (256,64,5,5), conv3.bias
extern "C" int dgvit_cnn_forward(const float* img, const float* const* params, float* feat, float* ws, long long ws_floats,
                                 float* scratch, long long scratch_floats, int B, int H, int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ConvDims d;
  TRY(make_conv_dims(B, H, W, d));
  DGVIT_CHECK_ARG(img && params && feat && ws && scratch, "dgvit_cnn_forward: null pointer");
  for (int i = 0; i < 6; ++i) DGVIT_CHECK_ARG(params[i], "cnn parameter %d is null", i);
  if (ws_floats < dgvit_cnn_workspace_floats(B, H, W) || scratch_floats < dgvit_cnn_forward_scratch_floats(B, H, W))
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "dgvit_cnn_forward: workspace or scratch too small");
  float* act[4] = {nullptr, ws, ws + al4(d.M[1] * 16), ws + al4(d.M[1] * 16) + al4(d.M[2] * 64)};
  float* cols = scratch;
  float* wp[3] = {scratch + conv_cols_floats(d), nullptr, nullptr};
  wp[1] = wp[0] + al4(16 * 28);
  wp[2] = wp[1] + al4(64 * 400);
  // in-launch split-K of the implicit-GEMM convolutions at small batches (conv3 at B = 32: 72 tiles of a 1600-deep GEMM on 256 CUs ran
  // 60 us; cut into 12 k-slices 3-4x less): arrival counters (left clean by the kernel: one memset per forward) + slabs behind the weights
  const SplitNeed csn = conv_split_need(d);
  SplitBuf csk;
  if (csn.tiles > 0) {
    csk.counters = reinterpret_cast<int*>(wp[0] + conv_wp_floats(d));
    csk.ncounters = (int)csn.tiles;
    csk.slabs = wp[0] + conv_wp_floats(d) + al4(csn.tiles);
    csk.slab_cap = csn.slab;
    HIP_TRY(hipMemsetAsync(csk.counters, 0, sizeof(int) * csn.tiles, st));
  }
  const float* in = img;
  for (int l = 0; l < 3; ++l) {
    TRY(weight_pack(params[2 * l], wp[l], d.C[l + 1], d.C[l], d.KP[l], 0, st));
    // conv2 / conv3 (NHWC input with 16 / 64 channels): implicit GEMM - the 5x5xC windows go from the activation straight into
    // the GEMM's A tiles (the patch-gather loader with a window step of 2 C floats), no column matrix.  conv1 (one channel,
    // 25 -> 28 padded taps) keeps im2col; the backward builds the columns it needs for the weight gradients itself.
    const int C = d.C[l], pw = 5 * C;
    const int shift = 24, inv = (1 << shift) / (pw > 0 ? pw : 1) + 1;
    bool gather = g_conv_gather && C % 4 == 0 && d.KP[l] == 25 * C && ((uintptr_t)in & 15) == 0 && (long long)d.KP[l] * inv < (1ll << 32) &&
                  (long long)B * d.H[l] * d.W[l] * C < (1ll << 29);
    for (int k = 0; gather && k < d.KP[l]; ++k) gather = (int)(((unsigned long long)(unsigned)k * (unsigned)inv) >> shift) == k / pw;   // exact k / pw
    if (!gather) TRY(im2col(in, cols, B, d.H[l], d.W[l], d.C[l], d.H[l + 1], d.W[l + 1], d.KP[l], st));
    GemmParams p = gp(cols, d.KP[l], wp[l], d.KP[l], act[l + 1], d.C[l + 1], (int)d.M[l + 1], d.C[l + 1], d.KP[l]);
    p.bias = params[2 * l + 1];
    if (gather) {
      p.g_img = in; p.g_img_floats = (long long)B * d.H[l] * d.W[l] * C;
      p.g_wi = d.W[l] * C; p.g_hw = d.H[l] * d.W[l] * C; p.g_ph = 2; p.g_kh = 5; p.g_pw = pw; p.g_xs = 2 * C;
      p.g_gw = d.W[l + 1]; p.g_P = d.H[l + 1] * d.W[l + 1]; p.g_inv = inv; p.g_shift = shift;
    }
    if (gather) csk.attach(p);
    TRY(gemm_f32(GEMM_NT, EPI_RELU, p, 1, st));   // relu(conv + bias), rows = next layer's NHWC input
    in = act[l + 1];
  }
  return avgpool(act[3], feat, B, d.H[3] * d.W[3], 256, st);
}

extern "C" int dgvit_cnn_backward(const float* img, const float* const* params, float* const* grads, const float* dfeat,
                                  const float* ws, long long ws_floats, float* scratch, long long scratch_floats, int B, int H,
                                  int W, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  ConvDims d;
  TRY(make_conv_dims(B, H, W, d));
  DGVIT_CHECK_ARG(img && params && grads && dfeat && ws && scratch, "dgvit_cnn_backward: null pointer");
  for (int i = 0; i < 6; ++i) DGVIT_CHECK_ARG(params[i] && grads[i], "cnn parameter/gradient %d is null", i);
  if (ws_floats < dgvit_cnn_workspace_floats(B, H, W) || scratch_floats < dgvit_cnn_backward_scratch_floats(B, H, W))
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "dgvit_cnn_backward: workspace or scratch too small");
  const float* act[4] = {img, ws, ws + al4(d.M[1] * 16), ws + al4(d.M[1] * 16) + al4(d.M[2] * 64)};
  const long long dyf = al4(std::max(std::max(d.M[1] * 16, d.M[2] * 64), d.M[3] * 256));
  float* cols = scratch;
  float* wp = cols + conv_cols_floats(d);          // packed weight of the current layer (largest first)
  float* dwp = wp + conv_wp_floats(d);             // packed weight gradient
  float* dya = dwp + conv_wp_floats(d);
  float* dyb = dya + dyf;
  float* slabs = dyb + dyf;
  const long long slab_floats = scratch_floats - (slabs - scratch);
  // d(avgpool) and the ReLU of conv3
  TRY(avgpool_bwd_relu(dfeat, act[3], dya, B, d.H[3] * d.W[3], 256, st));
  float* dy = dya;
  float* dnext = dyb;
  for (int l = 2; l >= 0; --l) {
    const int cout = d.C[l + 1], KP = d.KP[l];
    const int M = (int)d.M[l + 1];
    TRY(im2col(act[l], cols, B, d.H[l], d.W[l], d.C[l], d.H[l + 1], d.W[l + 1], KP, st));
    TRY(wgrad(dy, cout, cols, KP, dwp, grads[2 * l + 1], cout, KP, M, slabs, slab_floats, st));
    TRY(weight_pack(dwp, grads[2 * l], cout, d.C[l], KP, 1, st));
    if (l > 0) {
      TRY(weight_pack(params[2 * l], wp, cout, d.C[l], KP, 0, st));
      GemmParams p = gp(dy, cout, wp, KP, cols, KP, M, KP, cout);   // dcols = dy W  (cols buffer reused)
      TRY(gemm_f32(GEMM_NN, EPI_STORE, p, 1, st));
      TRY(col2im_relu(cols, act[l], dnext, B, d.H[l], d.W[l], d.C[l], d.H[l + 1], d.W[l + 1], st));
      float* t = dy; dy = dnext; dnext = t;
    }
  }
  return DGVIT_OK;
}

// ---------------------------------------------------------------------------------------------- replay staging
extern "C" int dgvit_gather_rows(const float* src, const long long* idx, float* out, long long nsel, long long row_floats,
                                 long long nrows, void* stream) {
  return gather_rows(src, idx, out, nsel, row_floats, nrows, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- SURVEY 8(f4)
extern "C" long long dgvit_depth_preprocess_scratch_floats(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return -1;
  return 2 * al4((long long)B * H * W) + al4(depth_normalize_scratch_floats(B));
}
extern "C" int dgvit_depth_normalize_u8(const float* depth, float* out, float* scratch, long long scratch_floats, int B, int H, int W,
                                        void* stream) {
  DGVIT_CHECK_ARG(scratch && scratch_floats >= depth_normalize_scratch_floats(B), "dgvit_depth_normalize_u8: scratch too small");
  return depth_normalize_u8(depth, out, scratch, B, H, W, (hipStream_t)stream);
}
extern "C" int dgvit_noise_clip(const float* img, const float* noise, float* out, long long n, float noise_level, unsigned long long seed,
                                void* stream) {
  return noise_clip(img, noise, out, n, noise_level, seed, (hipStream_t)stream);
}
extern "C" int dgvit_gaussian_blur(const float* img, float* out, float* tmp, int B, int H, int W, int ksize, int row0, int row1,
                                   void* stream) {
  return gaussian_blur_band(img, out, tmp, B, H, W, ksize, row0, row1, (hipStream_t)stream);
}
extern "C" int dgvit_resize_bilinear(const float* img, float* out, int B, int Hs, int Ws, int Hd, int Wd, float scale, void* stream) {
  return resize_bilinear(img, out, B, Hs, Ws, Hd, Wd, scale, (hipStream_t)stream);
}
// listener_callback (env_lab.py:420-434) + the resize of step() / reset() (:295-299): depth (B, H, W) -> state (B, out_h, out_w) in [0, 1]
extern "C" int dgvit_depth_to_state(const float* depth, const float* noise, float noise_level, unsigned long long seed, float* state,
                                    float* scratch, long long scratch_floats, int B, int H, int W, int out_h, int out_w, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DGVIT_CHECK_ARG(depth && state && scratch && B > 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0, "dgvit_depth_to_state: bad arguments");
  DGVIT_CHECK_ARG(((long long)H * W) % 4 == 0, "dgvit_depth_to_state: H * W must be a multiple of 4");
  const long long n = (long long)B * H * W;
  if (scratch_floats < dgvit_depth_preprocess_scratch_floats(B, H, W))
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "dgvit_depth_to_state: scratch %lld < %lld floats", scratch_floats,
                           dgvit_depth_preprocess_scratch_floats(B, H, W));
  float* a = scratch;
  float* b = scratch + al4(n);
  float* part = b + al4(n);
  TRY(depth_normalize_u8(depth, a, part, B, H, W, st));              // :424-426
  TRY(noise_clip(a, noise, a, n, noise_level, seed, st));             // add_nose :86-88
  TRY(gaussian_blur_band(a, a, b, B, H, W, 5, 0, H, st));             // add_nose :89   (b = horizontal pass, a = result)
  const int bh = H / 5, y1 = H / 2 - bh / 2;                          // get_center_band :33-39
  TRY(gaussian_blur_band(a, a, b, B, H, W, 11, y1, y1 + bh, st));     // blurring :69-76
  return resize_bilinear(a, state, B, H, W, out_h, out_w, 1.0f / 255.0f, st);   // :295, :299
}

// ---------------------------------------------------------------------------------------------- bf16 configuration
// BASELINE config 5 (224x224, ViT-Base variant, bf16): bf16 storage for GEMM operands (LayerNorm output, qkv,
// attention output, MLP hidden, branch outputs, weights, and in backward their gradients), fp32 residual stream and its
// gradient / LayerNorm statistics / biases / softmax / parameter gradients, fp32 accumulation on
// v_mfma_f32_32x32x16_bf16.  Same schedule as dgvit_got_forward / dgvit_got_backward (GoalFormer.py:156-171).
#include "bf16.h"

namespace {

inline long long al128(long long bytes) { return (bytes + 255) & ~255ll; }
inline int up8(long long n) { return (int)((n + 7) & ~7ll); }

// bf16 weight arena (elements): patch weight, then per layer to_qkv, to_out, fc1, fc2 in the reference's (out, in)
// layouts, followed by their transposes (in, out) -- the B operands of the data-gradient GEMMs dX = dY W
struct Wp {
  long long patch, layer0, qkv, out, fc1, fc2, qkvT, outT, fc1T, fc2T, layer_elems, total;
};
Wp make_wp(const Dims& d) {
  Wp w;
  long long o = 0;
  w.patch = o; o += al4((long long)d.D * d.pd);
  long long l = 0;
  w.qkv = l; l += (long long)3 * d.I * d.D;
  w.out = l; l += (long long)d.D * d.I;
  w.fc1 = l; l += (long long)d.M * d.D;
  w.fc2 = l; l += (long long)d.D * d.M;
  w.qkvT = l; l += (long long)3 * d.I * d.D;
  w.outT = l; l += (long long)d.D * d.I;
  w.fc1T = l; l += (long long)d.M * d.D;
  w.fc2T = l; l += (long long)d.D * d.M;
  w.layer0 = o; w.layer_elems = al4(l);
  o += w.layer_elems * d.L;
  w.total = o;
  return w;
}

// activation workspace in BYTES.  save: every layer keeps what backward needs; else the layers share one block.
struct Wsb {
  long long patches, xa, xb, pooled, delta, layer0, layer_stride, total;
  long long ln, qkv, ao, lse, xmid, ln2, h1, a1, xout, mean1, rstd1, mean2, rstd2, layer_bytes;   // relative to the layer base
};
Wsb make_wsb(const Dims& d, int save) {
  Wsb w;
  long long o = 0;
  w.patches = o; o += al128((long long)d.B * d.P * d.pd * 2);
  w.xa = o; o += al128(d.T * d.D * 4);
  w.xb = o; o += al128(d.T * d.D * 4);
  w.pooled = o; o += al128((long long)d.B * d.D * 4);
  w.delta = o; o += al128(d.T * d.D * 2);   // bf16 branch output (attention / feed-forward) before it joins the fp32 residual stream
  long long l = 0;
  w.ln = l; l += al128(d.T * d.D * 2);
  w.qkv = l; l += al128(d.T * 3 * d.I * 2);
  w.ao = l; l += al128(d.T * d.I * 2);
  w.lse = l; l += al128((long long)d.B * d.H * d.N * 4);
  w.xmid = l; l += al128(d.T * d.D * 4);
  w.a1 = l; l += al128(d.T * d.M * 2);
  w.ln2 = w.ln; w.h1 = w.xout = w.mean1 = w.rstd1 = w.mean2 = w.rstd2 = -1;
  if (save) {
    w.ln2 = l; l += al128(d.T * d.D * 2);
    w.h1 = l; l += al128(d.T * d.M * 2);      // pre-GELU hidden (GELU' in backward)
    w.xout = l; l += al128(d.T * d.D * 4);    // the layer's output = the next layer's residual input
    w.mean1 = l; l += al128(d.T * 4);
    w.rstd1 = l; l += al128(d.T * 4);
    w.mean2 = l; l += al128(d.T * 4);
    w.rstd2 = l; l += al128(d.T * 4);
  }
  w.layer0 = o; w.layer_bytes = l;
  w.layer_stride = save ? l : 0;
  o += save ? l * d.L : l;
  w.total = o;
  return w;
}

int check_bf16_dims(const Dims& d) {
  DGVIT_CHECK_ARG(d.dh == 64, "bf16 path: dim_head=%d unsupported (64)", d.dh);
  DGVIT_CHECK_ARG(d.proj, "bf16 path: heads == 1 with dim_head == dim (attention without output projection) runs on the fp32 path only");
  DGVIT_CHECK_ARG(d.D % 8 == 0 && d.M % 8 == 0 && d.pd % 8 == 0, "bf16 path: dim, mlp_dim and patch pixels must be multiples of 8");
  return DGVIT_OK;
}

GemmBf16Params gpb(const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc, int M, int N, int K) {
  GemmBf16Params p = {};
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  return p;
}

// split-K plan of a weight gradient dW (Mo x Ko) = sum over Tp token columns: about one virtual tile per CU
struct SplitPlan { int splits, kchunk; long long slab; };
SplitPlan wgrad_bf16_plan(int Mo, int Ko, int Tp) {
  const long long tiles = (long long)((Mo + 255) / 256) * ((Ko + 255) / 256);
  long long s = 256 / tiles;
  if (s < 1) s = 1;
  const long long maxs = (Tp + 511) / 512;   // at least 16 k-tiles per slice
  if (s > maxs) s = maxs;
  SplitPlan pl;
  pl.kchunk = (int)((((Tp + s - 1) / s) + 31) / 32 * 32);
  pl.splits = (Tp + pl.kchunk - 1) / pl.kchunk;
  pl.slab = al4((long long)Mo * Ko);
  return pl;
}
long long wgrad_bf16_scratch(int Mo, int Ko, int Tp) {
  const SplitPlan pl = wgrad_bf16_plan(Mo, Ko, Tp);
  return pl.splits > 1 ? pl.splits * pl.slab : 0;
}
// dW (Mo x Ko, fp32) = dY^T X straight from the token-major bf16 activations dY (T x Mo, row stride ldy), X (T x Ko, ldx):
// the TN layout of the ring GEMM (transposed LDS reads), split over tokens
int wgrad_bf16_tn(const bf16_t* dY, int ldy, const bf16_t* X, int ldx, float* dW, int Mo, int Ko, int T, float* slabs, long long slab_floats,
                  hipStream_t st) {
  const SplitPlan pl = wgrad_bf16_plan(Mo, Ko, T);
  GemmBf16Params p = gpb(dY, ldy, X, ldx, dW, Ko, Mo, Ko, T);
  p.tn = 1;
  if (pl.splits == 1) return gemm_bf16(BEPI_F32_PLAIN, p, st);
  if (slab_floats < pl.splits * pl.slab)
    return dgvit_set_error(DGVIT_ERR_WORKSPACE, "wgrad_bf16: slabs %lld < %lld floats", slab_floats, pl.splits * pl.slab);
  p.C = slabs;
  p.ksplit = pl.splits; p.kchunk = pl.kchunk; p.slab_stride = pl.slab;
  TRY(gemm_bf16(BEPI_F32_PLAIN, p, st));
  return reduce_slabs(slabs, dW, (long long)Mo * Ko, pl.splits, pl.slab, st);
}

// backward scratch in BYTES
struct Bsb {
  long long dxa, dxb, dxh, dln, dqkv, dao, dh1, slabs, part, delta, patches32, total;
  long long slab_floats;
};
Bsb make_bsb(const Dims& d) {
  Bsb s;
  long long o = 0;
  s.dxa = o; o += al128(d.T * d.D * 4);
  s.dxb = o; o += al128(d.T * d.D * 4);
  s.dxh = o; o += al128(d.T * d.D * 2);
  s.dln = o; o += al128(d.T * d.D * 2);
  s.dqkv = o; o += al128(d.T * 3 * d.I * 2);
  s.dao = o; o += al128(d.T * d.I * 2);
  s.dh1 = o; o += al128(d.T * d.M * 2);
  const long long widest = std::max<long long>(std::max(3 * d.I, d.M), d.D);
  const int Ti = (int)d.T;
  long long sl = wgrad_bf16_scratch(3 * d.I, d.D, Ti);
  sl = std::max(sl, wgrad_bf16_scratch(d.D, d.I, Ti));
  sl = std::max(sl, wgrad_bf16_scratch(d.M, d.D, Ti));
  sl = std::max(sl, wgrad_bf16_scratch(d.D, d.M, Ti));
  sl = std::max(sl, wgrad_scratch(d.D, d.pd, d.B * d.P));   // fp32 patch-embedding weight gradient
  s.slab_floats = sl;
  s.slabs = o; o += al128(sl * 4);
  long long part = (long long)layernorm_bwd_blocks((int)d.T) * 2 * d.D;
  part = std::max(part, (long long)colsum_blocks(d.B) * d.N * d.D);
  part = std::max(part, (long long)rmsnorm_bwd_blocks(d.B) * d.D);
  part = std::max(part, (long long)colsum_bf16_blocks((int)d.T) * widest);   // bias-gradient partials
  s.part = o; o += al128(part * 4);
  s.delta = o; o += al128((long long)d.B * d.H * d.N * 4);   // rowsum(dO o O) of the attention backward
  s.patches32 = o; o += al128((long long)d.B * d.P * d.pd * 4);
  s.total = o;
  return s;
}

}  // namespace

extern "C" long long dgvit_got_bf16_weight_elems(const dgvit_config* cfg) {
  Dims d;
  if (make_dims(cfg, 1, d) || check_bf16_dims(d)) return -1;
  return make_wp(d).total;
}

extern "C" long long dgvit_got_bf16_workspace_bytes(const dgvit_config* cfg, int batch, int save) {
  Dims d;
  if (make_dims(cfg, batch, d) || check_bf16_dims(d)) return -1;
  return make_wsb(d, save).total;
}

extern "C" long long dgvit_got_bf16_backward_scratch_bytes(const dgvit_config* cfg, int batch) {
  Dims d;
  if (make_dims(cfg, batch, d) || check_bf16_dims(d)) return -1;
  return make_bsb(d).total;
}

extern "C" int dgvit_got_pack_weights_bf16(const dgvit_config* cfg, const float* const* params, unsigned short* wpack,
                                           long long wpack_elems, int with_transposes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  TRY(make_dims(cfg, 1, d));
  TRY(check_bf16_dims(d));
  DGVIT_CHECK_ARG(params && wpack, "dgvit_got_pack_weights_bf16: null pointer");
  const Wp w = make_wp(d);
  if (wpack_elems < w.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "bf16 weight arena %lld < %lld elements", wpack_elems, w.total);
  CastBatch cb;     // the straight copies of all layers go out as one launch (49 segments at depth 12)
  cast_batch_init(cb);
  TRY(cast_batch_add(cb, params[P_PW], wpack + w.patch, (long long)d.D * d.pd, st));
  for (int i = 0; i < d.L; ++i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    bf16_t* lw = wpack + w.layer0 + w.layer_elems * i;
    TRY(cast_batch_add(cb, lp[L_QKV], lw + w.qkv, (long long)3 * d.I * d.D, st));
    TRY(cast_batch_add(cb, lp[L_OUTW], lw + w.out, (long long)d.D * d.I, st));
    TRY(cast_batch_add(cb, lp[L_FC1W], lw + w.fc1, (long long)d.M * d.D, st));
    TRY(cast_batch_add(cb, lp[L_FC2W], lw + w.fc2, (long long)d.D * d.M, st));
  }
  TRY(cast_batch_flush(cb, st));
  for (int i = 0; i < d.L; ++i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    bf16_t* lw = wpack + w.layer0 + w.layer_elems * i;
    if (!with_transposes) continue;
    TRY(transpose_cast_f32_bf16(lp[L_QKV], lw + w.qkvT, 3 * d.I, d.D, st));   // (3I, D) -> (D, 3I)
    TRY(transpose_cast_f32_bf16(lp[L_OUTW], lw + w.outT, d.D, d.I, st));      // (D, I)  -> (I, D)
    TRY(transpose_cast_f32_bf16(lp[L_FC1W], lw + w.fc1T, d.M, d.D, st));      // (M, D)  -> (D, M)
    TRY(transpose_cast_f32_bf16(lp[L_FC2W], lw + w.fc2T, d.D, d.M, st));      // (D, M)  -> (M, D)
  }
  return DGVIT_OK;
}

extern "C" int dgvit_got_forward_bf16(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack,
                                      const float* img, const float* goal, float* feat, void* workspace, long long ws_bytes,
                                      int batch, int save, float keep, unsigned long long seed,
                                      const unsigned long long* seed_dev, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  TRY(make_dims(cfg, batch, d));
  TRY(check_bf16_dims(d));
  DGVIT_CHECK_ARG(params && wpack && img && goal && feat && workspace, "dgvit_got_forward_bf16: null pointer");
  DGVIT_CHECK_ARG(keep > 0.f && keep <= 1.f, "dropout_keep must be in (0, 1]");
  const Wsb w = make_wsb(d, save);
  const Wp wp = make_wp(d);
  if (ws_bytes < w.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "bf16 forward workspace %lld < %lld bytes", ws_bytes, w.total);
  DGVIT_CHECK_ARG((uintptr_t)workspace % 256 == 0 && (uintptr_t)wpack % 16 == 0, "bf16 path: workspace must be 256-byte aligned");
  for (int i = 0; i < P_L0 + DGVIT_PARAMS_PER_LAYER * d.L; ++i) DGVIT_CHECK_ARG(params[i], "parameter %d is null", i);
  unsigned char* ws = (unsigned char*)workspace;
  const int T = (int)d.T;
  auto f32 = [&](unsigned char* base, long long off) { return save ? (float*)(base + off) : (float*)nullptr; };

  bf16_t* patches = (bf16_t*)(ws + w.patches);
  bf16_t* delta = (bf16_t*)(ws + w.delta);
  float* x = (float*)(ws + w.xa);
  TRY(patchify_bf16(img, patches, d.B, cfg->image_h, cfg->image_w, cfg->patch_h, cfg->patch_w, st));
  {
    GemmBf16Params p = gpb(patches, d.pd, wpack + wp.patch, d.pd, x, d.D, d.B * d.P, d.D, d.pd);
    p.bias = params[P_PB];
    p.res = params[P_POS]; p.ldr = d.D; p.res_mod = d.P;
    p.c_rgrp = d.P;
    TRY(gemm_bf16(BEPI_F32, p, st));
  }
  TRY(goal_row(goal, params[P_POS], x, d.B, d.N, d.D, st));
  if (keep < 1.f) TRY(dropout_inplace(x, d.T * d.D, seed, seed_dev, keep, st));

  // The branch outputs (to_out, fc2: GoalFormer.py:82,49) are stored bf16 like every other GEMM output and join the fp32
  // residual stream inside the LayerNorm kernel of the next sub-block (x = attn(..) + x; x = ff(..) + x, :103-104): the GEMM
  // epilogues then have no fp32 residual read on their critical path.
  {
    unsigned char* lb0 = ws + w.layer0;
    TRY(layernorm_fwd_bf16(x, params[P_L0 + L_LN1W], params[P_L0 + L_LN1B], (bf16_t*)(lb0 + w.ln), f32(lb0, w.mean1), f32(lb0, w.rstd1),
                           T, d.D, 1e-5f, 1, st));
  }
  for (int i = 0; i < d.L; ++i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    const bf16_t* lw = wpack + wp.layer0 + wp.layer_elems * i;
    unsigned char* lb = ws + w.layer0 + w.layer_stride * i;
    bf16_t* ln = (bf16_t*)(lb + w.ln);
    bf16_t* ln2 = (bf16_t*)(lb + w.ln2);
    bf16_t* qkv = (bf16_t*)(lb + w.qkv);
    bf16_t* ao = (bf16_t*)(lb + w.ao);
    float* xmid = (float*)(lb + w.xmid);
    bf16_t* a1 = (bf16_t*)(lb + w.a1);
    // inference: the last block only needs token 0 downstream of K/V (see dgvit_got_forward); training keeps it dense
    const bool last = !dense_last_block(cfg) && !save && !d.pool_mean && i == d.L - 1;
    const int tok = last ? d.B : T, rs = last ? d.N : 1;
    float* xo = save ? (float*)(lb + w.xout) : (x == (float*)(ws + w.xa) ? (float*)(ws + w.xb) : (float*)(ws + w.xa));
    if (!last) {
      GemmBf16Params p = gpb(ln, d.D, lw + wp.qkv, d.D, qkv, 3 * d.I, T, 3 * d.I, d.D);
      TRY(gemm_bf16(BEPI_BF16, p, st));
    } else {
      GemmBf16Params kv = gpb(ln, d.D, lw + wp.qkv + (long long)d.I * d.D, d.D, qkv + d.I, 3 * d.I, T, 2 * d.I, d.D);
      TRY(gemm_bf16(BEPI_BF16, kv, st));
      GemmBf16Params q = gpb(ln, rs * d.D, lw + wp.qkv, d.D, qkv, rs * 3 * d.I, tok, d.I, d.D);
      TRY(gemm_bf16(BEPI_BF16, q, st));
    }
    TRY(attention_fwd_bf16(qkv, ao, f32(lb, w.lse), d.B, d.N, d.H, d.dh, last ? 1 : d.N, st));
    {
      GemmBf16Params p = gpb(ao, rs * d.I, lw + wp.out, d.I, delta, rs * d.D, tok, d.D, d.I);
      p.bias = lp[L_OUTB];
      TRY(gemm_bf16(BEPI_BF16, p, st));
    }
    // xmid = x + to_out(..);  ln2 = LN2(xmid).  No-grad passes do not store xmid in the blocks that have a successor: the feed-forward
    // output goes to the (free again) attention-output buffer and both branch outputs join the stream in ONE pass below,
    // (x + d_attn) + d_ff in the order of the two-step schedule: identical results, 22 instead of 24 bytes per element and block
    const bool joint = !save && i + 1 < d.L && d.I >= d.D;     // (the attention-output buffer holds T x I elements)
    if (joint)
      TRY(add2_layernorm_fwd_bf16(x, delta, nullptr, nullptr, lp[L_LN2W], lp[L_LN2B], ln2, T, d.D, 1e-5f, st));
    else
      TRY(add_layernorm_fwd_bf16(x, delta, xmid, lp[L_LN2W], lp[L_LN2B], ln2, f32(lb, w.mean2), f32(lb, w.rstd2), tok, d.D, 1e-5f, rs, st));
    {
      GemmBf16Params p = gpb(ln2, rs * d.D, lw + wp.fc1, d.D, a1, d.M, tok, d.M, d.D);
      p.bias = lp[L_FC1B];
      if (save) {
        p.C2 = (bf16_t*)(lb + w.h1); p.ldc2 = d.M;
        TRY(gemm_bf16(BEPI_GELU2_BF16, p, st));
      } else {
        TRY(gemm_bf16(BEPI_GELU_BF16, p, st));
      }
    }
    {
      GemmBf16Params p = gpb(a1, d.M, lw + wp.fc2, d.M, joint ? ao : delta, rs * d.D, tok, d.D, d.M);
      p.bias = lp[L_FC2B];
      TRY(gemm_bf16(BEPI_BF16, p, st));
    }
    // xo = xmid + ff(..), and the next block's LN1 of it
    if (joint) {
      unsigned char* nb = ws + w.layer0 + w.layer_stride * (i + 1);
      TRY(add2_layernorm_fwd_bf16(x, delta, ao, xo, lp[DGVIT_PARAMS_PER_LAYER + L_LN1W], lp[DGVIT_PARAMS_PER_LAYER + L_LN1B],
                                  (bf16_t*)(nb + w.ln), T, d.D, 1e-5f, st));
    } else if (i + 1 < d.L) {
      unsigned char* nb = ws + w.layer0 + w.layer_stride * (i + 1);
      TRY(add_layernorm_fwd_bf16(xmid, delta, xo, lp[DGVIT_PARAMS_PER_LAYER + L_LN1W], lp[DGVIT_PARAMS_PER_LAYER + L_LN1B],
                                 (bf16_t*)(nb + w.ln), f32(nb, w.mean1), f32(nb, w.rstd1), T, d.D, 1e-5f, 1, st));
    } else {
      TRY(residual_add_bf16(xmid, delta, xo, tok, d.D, rs, st));
    }
    x = xo;
  }
  if (d.pool_mean) {
    float* pooled = (float*)(ws + w.pooled);
    TRY(avgpool(x, pooled, d.B, d.N, d.D, st));
    return rmsnorm_fwd(pooled, d.D, params[P_RMS], feat, d.B, d.D, st);
  }
  return rmsnorm_fwd(x, (long long)d.N * d.D, params[P_RMS], feat, d.B, d.D, st);
}

// Gradient of dgvit_got_forward_bf16 (save_for_backward = 1).  Data-gradient GEMMs take the transposed weight copies of
// the arena as B operand; weight-gradient GEMMs contract over tokens, so both operands are first transposed to
// token-contiguous bf16 copies (zero padded to a multiple of 8 tokens) and the product is split over tokens into fp32
// slabs that a fixed-order reduction sums (deterministic).  Bias gradients are the row sums of the transposed dY.
extern "C" int dgvit_got_backward_bf16(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack,
                                       float* const* grads, const float* dfeat, float* dgoal, const float* img,
                                       const void* workspace, long long ws_bytes, void* scratch, long long scratch_bytes, int batch,
                                       float keep, unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
  return dgvit_got_backward_bf16_ev(cfg, params, wpack, grads, dfeat, dgoal, img, workspace, ws_bytes, scratch, scratch_bytes, batch, keep, seed,
                                    seed_dev, stream, nullptr);
}

extern "C" int dgvit_got_backward_bf16_ev(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack,
                                          float* const* grads, const float* dfeat, float* dgoal, const float* img,
                                          const void* workspace, long long ws_bytes, void* scratch, long long scratch_bytes, int batch,
                                          float keep, unsigned long long seed, const unsigned long long* seed_dev, void* stream,
                                          const dgvit_grad_events* events) {
  hipStream_t st = (hipStream_t)stream;
  Dims d;
  TRY(make_dims(cfg, batch, d));
  TRY(check_bf16_dims(d));
  DGVIT_CHECK_ARG(params && wpack && grads && dfeat && img && workspace && scratch, "dgvit_got_backward_bf16: null pointer");
  TRY(check_events(events, cfg->depth));
  const Wsb w = make_wsb(d, 1);
  const Wp wp = make_wp(d);
  const Bsb s = make_bsb(d);
  if (ws_bytes < w.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "bf16 backward workspace %lld < %lld bytes", ws_bytes, w.total);
  if (scratch_bytes < s.total) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "bf16 backward scratch %lld < %lld bytes", scratch_bytes, s.total);
  DGVIT_CHECK_ARG((uintptr_t)workspace % 256 == 0 && (uintptr_t)scratch % 256 == 0, "bf16 path: workspace / scratch must be 256-byte aligned");
  const int np = P_L0 + DGVIT_PARAMS_PER_LAYER * d.L;
  for (int i = 0; i < np; ++i) DGVIT_CHECK_ARG(params[i], "parameter %d is null", i);   // grads[i] == NULL: frozen parameter
  const unsigned char* ws = (const unsigned char*)workspace;
  unsigned char* sc = (unsigned char*)scratch;
  const int T = (int)d.T;
  float* dx = (float*)(sc + s.dxa);      // gradient of the residual stream entering the current op (fp32)
  float* dx2 = (float*)(sc + s.dxb);
  bf16_t* dxh = (bf16_t*)(sc + s.dxh);   // its bf16 copy: A operand of the data-gradient GEMMs, source of the transposed dY
  bf16_t* dln = (bf16_t*)(sc + s.dln);
  bf16_t* dqkv = (bf16_t*)(sc + s.dqkv);
  bf16_t* dao = (bf16_t*)(sc + s.dao);
  bf16_t* dh1 = (bf16_t*)(sc + s.dh1);
  float* slabs = (float*)(sc + s.slabs);
  float* part = (float*)(sc + s.part);

  // dW (no x ni) and optionally db (no) from dY (T x no, row stride ldy) and X (T x ni, row stride ldx)
  auto wgrad = [&](const bf16_t* dY, int ldy, const bf16_t* X, int ldx, float* dW, float* db, int no, int ni) -> int {
    if (db) TRY(colsum_bf16(dY, ldy, db, part, T, no, st));
    if (!dW) return DGVIT_OK;
    return wgrad_bf16_tn(dY, ldy, X, ldx, dW, no, ni, T, slabs, s.slab_floats, st);
  };

  // RMSNorm on token 0 (or the token mean) of the last layer's output
  const float* xl = (const float*)(ws + w.layer0 + w.layer_stride * (d.L - 1) + w.xout);
  if (d.pool_mean) {
    TRY(rmsnorm_bwd(dfeat, (const float*)(ws + w.pooled), d.D, params[P_RMS], dx2, d.D, grads[P_RMS], part, d.B, d.D, st));
    TRY(mean_bwd(dx2, dx, d.B, d.N, d.D, st));
  } else {
    HIP_TRY(hipMemsetAsync(dx, 0, sizeof(float) * d.T * d.D, st));
    TRY(rmsnorm_bwd(dfeat, xl, (long long)d.N * d.D, params[P_RMS], dx, (long long)d.N * d.D, grads[P_RMS], part, d.B, d.D, st));
  }
  if (events) TRY(mark_ready(events->head, st));
  TRY(cast_f32_bf16(dx, dxh, d.T * d.D, st));

  for (int i = d.L - 1; i >= 0; --i) {
    const float* const* lp = params + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    float* const* lg = grads + P_L0 + DGVIT_PARAMS_PER_LAYER * i;
    const bf16_t* lw = wpack + wp.layer0 + wp.layer_elems * i;
    const unsigned char* lb = ws + w.layer0 + w.layer_stride * i;
    const float* xin = i == 0 ? (const float*)(ws + w.xa) : (const float*)(ws + w.layer0 + w.layer_stride * (i - 1) + w.xout);
    const bf16_t* ln1 = (const bf16_t*)(lb + w.ln);
    const bf16_t* ln2 = (const bf16_t*)(lb + w.ln2);
    const bf16_t* qkv = (const bf16_t*)(lb + w.qkv);
    const bf16_t* ao = (const bf16_t*)(lb + w.ao);
    const bf16_t* h1 = (const bf16_t*)(lb + w.h1);
    const bf16_t* a1 = (const bf16_t*)(lb + w.a1);
    const float* xmid = (const float*)(lb + w.xmid);
    // ---- feed-forward branch: xout = xmid + fc2(gelu(fc1(ln2)))        (dx / dxh = d xout)
    TRY(wgrad(dxh, d.D, a1, d.M, lg[L_FC2W], lg[L_FC2B], d.D, d.M));
    {
      GemmBf16Params p = gpb(dxh, d.D, lw + wp.fc2T, d.D, dh1, d.M, T, d.M, d.D);   // dh1 = (dx W2) * gelu'(h1)
      p.aux = h1; p.ldaux = d.M;
      TRY(gemm_bf16(BEPI_DGELU_BF16, p, st));
    }
    TRY(wgrad(dh1, d.M, ln2, d.D, lg[L_FC1W], lg[L_FC1B], d.M, d.D));
    {
      GemmBf16Params p = gpb(dh1, d.M, lw + wp.fc1T, d.M, dln, d.D, T, d.D, d.M);     // dln2 = dh1 W1
      TRY(gemm_bf16(BEPI_BF16, p, st));
    }
    TRY(layernorm_bwd_bf16(dln, xmid, (const float*)(lb + w.mean2), (const float*)(lb + w.rstd2), lp[L_LN2W], dx, dx2, dxh, lg[L_LN2W],
                           lg[L_LN2B], part, T, d.D, 1, st));
    // ---- attention branch: xmid = xin + to_out(attn(to_qkv(ln1)))      (dx2 / dxh = d xmid)
    TRY(wgrad(dxh, d.D, ao, d.I, lg[L_OUTW], lg[L_OUTB], d.D, d.I));
    {
      GemmBf16Params p = gpb(dxh, d.D, lw + wp.outT, d.D, dao, d.I, T, d.I, d.D);     // dao = dxmid Wo
      TRY(gemm_bf16(BEPI_BF16, p, st));
    }
    TRY(attention_bwd_bf16(qkv, ao, dao, (const float*)(lb + w.lse), dqkv, (float*)(sc + s.delta), d.B, d.N, d.H, d.dh, st));
    TRY(wgrad(dqkv, 3 * d.I, ln1, d.D, lg[L_QKV], nullptr, 3 * d.I, d.D));
    {
      GemmBf16Params p = gpb(dqkv, 3 * d.I, lw + wp.qkvT, 3 * d.I, dln, d.D, T, d.D, 3 * d.I);   // dln1 = dqkv Wqkv
      TRY(gemm_bf16(BEPI_BF16, p, st));
    }
    TRY(layernorm_bwd_bf16(dln, xin, (const float*)(lb + w.mean1), (const float*)(lb + w.rstd1), lp[L_LN1W], dx2, dx, dxh, lg[L_LN1W],
                           lg[L_LN1B], part, T, d.D, 1, st));
    if (events) TRY(mark_ready(events->layer[i], st));
  }
  // ---- token assembly: x0 = dropout(cat(goal, patches W^T + b) + pos): fp32, as dgvit_got_backward
  if (keep < 1.f) TRY(dropout_inplace(dx, d.T * d.D, seed, seed_dev, keep, st));
  if (dgoal)
    HIP_TRY(hipMemcpy2DAsync(dgoal, sizeof(float) * d.D, dx, sizeof(float) * d.N * d.D, sizeof(float) * d.D, d.B,
                             hipMemcpyDeviceToDevice, st));
  if (grads[P_POS]) TRY(colsum(dx, (long long)d.N * d.D, grads[P_POS], part, d.B, d.N * d.D, 0, st));
  if (!grads[P_PW] && !grads[P_PB]) return DGVIT_OK;
  HIP_TRY(hipMemcpy2DAsync(dx2, sizeof(float) * d.P * d.D, dx + d.D, sizeof(float) * d.N * d.D, sizeof(float) * d.P * d.D, d.B,
                           hipMemcpyDeviceToDevice, st));
  float* patches32 = (float*)(sc + s.patches32);
  TRY(patchify(img, patches32, d.B, cfg->image_h, cfg->image_w, cfg->patch_h, cfg->patch_w, st));
  return ::wgrad(dx2, d.D, patches32, d.pd, grads[P_PW], grads[P_PB], d.D, d.pd, d.B * d.P, slabs, s.slab_floats, st);
}

// operator-level exports of the bf16 kernels (parity tests, benches)
extern "C" int dgvit_cast_f32_bf16(const float* src, unsigned short* dst, long long n, void* stream) {
  return cast_f32_bf16(src, dst, n, (hipStream_t)stream);
}
extern "C" int dgvit_gemm_bf16(int epilogue, const unsigned short* A, int lda, const unsigned short* B, int ldb, void* C, int ldc,
                               int M, int N, int K, const float* bias, const float* res, int ldr, unsigned short* C2, int ldc2,
                               const unsigned short* aux, int ldaux, void* stream) {
  DGVIT_CHECK_ARG(A && B && C, "dgvit_gemm_bf16: null pointer");
  GemmBf16Params p = gpb(A, lda, B, ldb, C, ldc, M, N, K);
  p.bias = bias; p.res = res; p.ldr = ldr; p.C2 = C2; p.ldc2 = ldc2; p.aux = aux; p.ldaux = ldaux;
  DGVIT_CHECK_ARG(epilogue != BEPI_DGELU_BF16 || aux, "dgvit_gemm_bf16: epilogue 3 needs aux");
  return gemm_bf16(epilogue, p, (hipStream_t)stream);
}
extern "C" long long dgvit_wgrad_bf16_scratch_floats(int Mo, int Ko, int T) {
  return wgrad_bf16_scratch(Mo, Ko, T) + (long long)colsum_bf16_blocks(T) * Mo;
}
extern "C" int dgvit_wgrad_bf16(const unsigned short* dY, const unsigned short* X, float* dW, float* db, float* scratch,
                                long long scratch_floats, int T, int Mo, int Ko, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DGVIT_CHECK_ARG(dY && X && dW && T > 0 && Mo > 0 && Ko > 0 && Mo % 8 == 0 && Ko % 8 == 0, "dgvit_wgrad_bf16: bad arguments");
  // scratch layout: [split-K slabs | bias-gradient partials]
  const long long nsl = wgrad_bf16_scratch(Mo, Ko, T), npart = db ? (long long)colsum_bf16_blocks(T) * Mo : 0;
  if (scratch_floats < nsl + npart) return dgvit_set_error(DGVIT_ERR_WORKSPACE, "dgvit_wgrad_bf16: scratch %lld < %lld floats", scratch_floats, nsl + npart);
  DGVIT_CHECK_ARG(scratch || nsl + npart == 0, "dgvit_wgrad_bf16: null scratch");
  if (db) TRY(colsum_bf16(dY, Mo, db, scratch + nsl, T, Mo, st));
  return wgrad_bf16_tn(dY, Mo, X, Ko, dW, Mo, Ko, T, scratch, nsl, st);
}
extern "C" int dgvit_layernorm_forward_bf16(const float* x, const float* gamma, const float* beta, unsigned short* y, float* mean,
                                            float* rstd, int rows, int D, void* stream) {
  return layernorm_fwd_bf16(x, gamma, beta, y, mean, rstd, rows, D, 1e-5f, 1, (hipStream_t)stream);
}
extern "C" int dgvit_attention_backward_bf16(const unsigned short* qkv, const unsigned short* out, const unsigned short* dout,
                                             const float* lse, unsigned short* dqkv, float* delta, int B, int N, int H, int dh,
                                             void* stream) {
  return attention_bwd_bf16(qkv, out, dout, lse, dqkv, delta, B, N, H, dh, (hipStream_t)stream);
}
extern "C" int dgvit_attention_forward_bf16(const unsigned short* qkv, unsigned short* out, float* lse, int B, int N, int H, int dh,
                                            void* stream) {
  return attention_fwd_bf16(qkv, out, lse, B, N, H, dh, N, (hipStream_t)stream);
}
