/* libdgvit_hip.so -- C ABI of the MI355X (gfx950) DGViT encoder hot path.
 *
 * The reference (REGRAGUIahmed/DGViT) has no FFI: its boundary is the Python nn.Module surface of
 *   src/vis_nav/vis_nav/GoalFormer.py        (GoT, Transformer, Attention, FeedForward, PreNorm, RMSNorm)
 *   src/vis_nav/vis_nav/got_sac_network.py   (GoTPolicy, GoTQNetwork, DeterministicGoTPolicy)
 * This header is the lower boundary a Python (ctypes) or C++ host binds instead of the aten ops those
 * modules call.  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a caller-allocated, contiguous fp32 buffer (row-major);
 *   - nothing allocates, frees or synchronises; all work is enqueued on `stream` (a hipStream_t passed
 *     as void*, 0 = the null stream); callable from any host thread (autograd's backward thread too);
 *   - no mutable global state besides one-time initialisation (a helper stream, per-device kernel attributes): there are no
 *     setter functions; the two schedule options are fields of dgvit_config (`flags`) and travel with every call.  The A/B and
 *     diagnostic knobs of tools/ exist only in the separately built libdgvit_hip_diag.so (include/dgvit_hip_diag.h);
 *   - return value 0 = success, negative = error; dgvit_last_error() gives the thread-local message;
 *   - sizes are element counts unless a name says bytes.
 */
#ifndef DGVIT_HIP_H
#define DGVIT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define DGVIT_ABI_VERSION 7

/* error codes */
#define DGVIT_OK 0
#define DGVIT_ERR_ARG (-1)
#define DGVIT_ERR_HIP (-2)
#define DGVIT_ERR_ALIGN (-3)
#define DGVIT_ERR_WORKSPACE (-4)

int dgvit_abi_version(void);
/* sizeof(dgvit_config) as this library was compiled: a host binding checks its own struct against it before the first call
 * (a struct one field short makes the library read past its end) */
int dgvit_config_size(void);
const char* dgvit_last_error(void);
/* number of visible HIP devices (<0 on error); does not create a context */
int dgvit_device_count(void);

/* ----------------------------------------------------------------------------------------------
 * Encoder shape = the GoT constructor arguments (GoalFormer.py:124-154).  patch_h/patch_w are honoured
 * (the reference hard-wires 16x20, GoalFormer.py:137-139).  dim_head must be 64 or 32; tokens
 * N = (image_h/patch_h)*(image_w/patch_w) + 1 <= 288.
 * -------------------------------------------------------------------------------------------- */
typedef struct dgvit_config {
  int image_h, image_w;
  int patch_h, patch_w;
  int dim;       /* D */
  int depth;     /* L */
  int heads;     /* H */
  int dim_head;  /* dh, inner width I = H*dh */
  int mlp_dim;   /* M */
  int pool_mean; /* 0: pool='cls' (token 0, what the reference's networks use), 1: pool='mean' (GoalFormer.py:167) */
  int flags;     /* schedule options, DGVIT_FLAG_* below (0 = defaults); a forward and its backward must be given the same value */
} dgvit_config;
/* The output reads only token 0 of the last block (GoalFormer.py:167), so by default that block computes Q, the attention output,
 * to_out and the feed-forward for one row per frame (K / V for all) -- identical results, fewer FLOPs.  This flag runs the dense
 * last block instead (A/B measurements; the bf16 training path is always dense). */
#define DGVIT_FLAG_DENSE_LAST_BLOCK 1
/* dgvit_got_backward runs the weight-gradient GEMMs on one internal helper stream (created on first use, ordered against the
 * caller's stream with events only, so it is capturable) beside the data-gradient chain: +3..5 % frames/s at BASELINE config 3 on
 * MI355X, but concurrent kernels stretch each other's durations, so per-kernel timings (dgvit_profile_*, rocprof) stop being
 * interpretable.  Off: everything stays on the caller's stream. */
#define DGVIT_FLAG_WGRAD_OVERLAP 2

/* Parameter / gradient tables: arrays of DGVIT_NUM_GLOBAL_PARAMS + DGVIT_PARAMS_PER_LAYER*depth device
 * pointers in this order (reference state_dict key in brackets, prefix "trans."):
 *   0 pos_embedding (1,N,D)                         [pos_embedding]
 *   1 patch weight (D, patch_h*patch_w)             [to_patch_embedding.1.weight]
 *   2 patch bias (D)                                [to_patch_embedding.1.bias]
 *   3 final RMSNorm gain (D)                        [layer_norm.g]
 *   then for layer i (prefix transformer.layers.i.):
 *   +0 LN1 weight (D) [0.norm.weight]   +1 LN1 bias (D) [0.norm.bias]
 *   +2 to_qkv weight (3I, D) [0.fn.to_qkv.weight]
 *   +3 to_out weight (D, I) [0.fn.to_out.0.weight]  +4 to_out bias (D) [0.fn.to_out.0.bias]
 *   +5 LN2 weight (D) [1.norm.weight]   +6 LN2 bias (D) [1.norm.bias]
 *   +7 MLP fc1 weight (M, D) [1.fn.net.0.weight]    +8 fc1 bias (M) [1.fn.net.0.bias]
 *   +9 MLP fc2 weight (D, M) [1.fn.net.3.weight]    +10 fc2 bias (D) [1.fn.net.3.bias]
 * cls_token and mlp_head.* never take part in the forward (GoalFormer.py:143,151-154) and are not passed.
 * heads == 1 and dim_head == dim: the reference's Attention has NO output projection (to_out = nn.Identity(), GoalFormer.py:56,66-69);
 * slots +3 / +4 are then ignored in both tables (pass NULL) and the attention output joins the residual stream directly (fp32 path;
 * the bf16 entry points refuse that shape). */
#define DGVIT_NUM_GLOBAL_PARAMS 4
#define DGVIT_PARAMS_PER_LAYER 11

/* floats of activation workspace the forward needs for `batch` frames.
 * save_for_backward != 0: every layer keeps its activations (input of dgvit_got_backward);
 * == 0: layers reuse one set of buffers (inference). */
long long dgvit_got_workspace_floats(const dgvit_config* cfg, int batch, int save_for_backward);
/* floats of scratch dgvit_got_backward needs (gradient temporaries, split-K slabs, reduction partials) */
long long dgvit_got_backward_scratch_floats(const dgvit_config* cfg, int batch);

/* GoT.forward (GoalFormer.py:156-171) incl. Transformer/PreNorm/Attention/FeedForward/RMSNorm
 * (GoalFormer.py:31-122).
 *   img  (B, image_h, image_w)   goal (B, D)   ->   feat (B, D)
 * dropout_keep < 1 applies train-mode nn.Dropout(emb_dropout) (GoalFormer.py:163) with a Philox mask
 * derived from dropout_seed; pass 1.0f for eval mode.  dropout_seed_dev (may be NULL): a DEVICE pointer to the
 * seed, read by the kernel at run time and overriding dropout_seed -- the form to use inside a captured HIP
 * graph, where a by-value seed would be frozen into every replay. */
int dgvit_got_forward(const dgvit_config* cfg, const float* const* params, const float* img, const float* goal,
                      float* feat, float* workspace, long long workspace_floats, int batch, int save_for_backward,
                      float dropout_keep, unsigned long long dropout_seed, const unsigned long long* dropout_seed_dev,
                      void* stream);

/* Gradient of dgvit_got_forward (what autograd derives for GoalFormer.py:156-171).
 *   dfeat (B, D) -> grads[] (same table order as params, each written, not accumulated), dgoal (B, D).
 * A NULL entry of grads[] marks a frozen parameter (requires_grad off, e.g. the heads-only optimiser of DRL.py:145-148 with the
 * encoder frozen): its weight-gradient GEMM / reduction is skipped.
 * `workspace` is the buffer the matching forward (save_for_backward=1) filled; same dropout_keep/seed. */
int dgvit_got_backward(const dgvit_config* cfg, const float* const* params, float* const* grads, const float* dfeat,
                       float* dgoal, const float* workspace, long long workspace_floats, float* scratch,
                       long long scratch_floats, int batch, float dropout_keep, unsigned long long dropout_seed,
                       const unsigned long long* dropout_seed_dev, void* stream);

/* Gradient-ready events: the data-parallel gradient exchange (RCCL all-reduce, one bucket per transformer block) can start while the
 * backward is still running on earlier blocks.  dgvit_got_backward_ev / dgvit_got_backward_bf16_ev are dgvit_got_backward /
 * dgvit_got_backward_bf16 with one more argument: events the call RECORDS ON `stream` at the point where a group of parameter
 * gradients is final (all split-K slab sums and LayerNorm partial sums included):
 *   head      after the final-norm gradient (grads[3]), before the last block's backward;
 *   layer[i]  after every gradient of transformer block i (grads[4 + 11 i .. 4 + 11 i + 10]); blocks finish in the order L-1 .. 0.
 * The embedding gradients (grads[0..2]) are final when the call's work on `stream` is.  NULL entries are skipped; events == NULL is
 * the plain call.  Events are hipEvent_t handles: the caller's own, or made by dgvit_event_create (timing disabled).  A consumer on
 * another stream orders itself with dgvit_stream_wait_event (= hipStreamWaitEvent) -- host code never blocks.
 * (The reference has no counterpart: torch DDP's bucket hooks on autograd, which one fused backward call bypasses.) */
typedef struct dgvit_grad_events {
  int n_layers;        /* must equal cfg->depth */
  void* const* layer;  /* [n_layers] hipEvent_t or NULL */
  void* head;          /* hipEvent_t or NULL */
} dgvit_grad_events;
int dgvit_event_create(void** event);
int dgvit_event_destroy(void* event);
int dgvit_stream_wait_event(void* stream, void* event);
int dgvit_got_backward_ev(const dgvit_config* cfg, const float* const* params, float* const* grads, const float* dfeat,
                          float* dgoal, const float* workspace, long long workspace_floats, float* scratch,
                          long long scratch_floats, int batch, float dropout_keep, unsigned long long dropout_seed,
                          const unsigned long long* dropout_seed_dev, void* stream, const dgvit_grad_events* events);

/* ----------------------------------------------------------------------------------------------
 * Head Linears (got_sac_network.py:111,115-121,226,230-234,429,433-435):  y = act(x W^T + b)
 *   x (M, K), w (N, K), b (N) or NULL, y (M, N); act: 0 = identity, 1 = ReLU.
 * -------------------------------------------------------------------------------------------- */
int dgvit_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int N, int K, int act,
                         void* stream);
long long dgvit_linear_backward_scratch_floats(int M, int N, int K);
/* dy (M,N) is the gradient of y; with act = 1 it is masked by (y > 0) first (y = the forward output).
 * Writes dx (M,K) (NULL to skip), dw (N,K), db (N) (NULL to skip). */
int dgvit_linear_backward(const float* dy, const float* x, const float* w, const float* y, float* dx, float* dw,
                          float* db, float* scratch, long long scratch_floats, int M, int N, int K, int act,
                          void* stream);

/* ----------------------------------------------------------------------------------------------
 * Fused MLP heads (got_sac_network.py:114-121 twin Q, :230-235 policy, :433-435 deterministic policy, CNN twins :157-166, :303-307):
 *     y_j = W3_j relu(W2 relu(W1 cat(x_0 .. x_{nseg-1}) + b1) + b2) + b3_j ,  j < heads3,  for `towers` independent towers
 * one launch forward, one backward (+ one grouped reduction when batch > 32), instead of one GEMM / mask / split-K / reduction
 * launch per Linear.  A policy head = 1 tower, 2 third layers (mean_linear, log_std_linear); a twin-Q head = 2 towers
 * (fc1,fc2,fc3 | fc11,fc21,fc31) over the same concatenated input.  n1, n2: multiples of 32 up to 128; n3 <= 4; sum kx <= 512.
 *   in[s]     : (batch, kx[s]) with row stride ldx[s]          (the torch.cat of the reference is never materialised)
 *   params    : per tower  W1 (n1, K0), b1, W2 (n2, n1), b2, then per third layer W3 (n3, n2), b3
 *   h1, h2    : (towers, batch, n1 / n2) post-ReLU activations, kept for the backward
 *   y         : (towers, heads3, batch, n3)
 *   dy        : towers * heads3 pointers to (batch, n3) gradients of the single outputs; NULL = that output is unused: it
 *               contributes nothing and its third layer gets no gradient (like an nn.Linear outside the autograd graph)
 *   din[s]    : (batch, kx[s]) dense, NULL to skip;  dparams: like params, NULL entries skipped (written, not accumulated)
 * -------------------------------------------------------------------------------------------- */
typedef struct dgvit_mlp_desc {
  int batch, nseg, kx[3], ldx[3];
  int n1, n2, n3, towers, heads3;
} dgvit_mlp_desc;
int dgvit_mlp_head_forward(const dgvit_mlp_desc* desc, const float* const* in, const float* const* params, float* h1, float* h2,
                           float* y, void* stream);
long long dgvit_mlp_head_backward_scratch_floats(const dgvit_mlp_desc* desc);
int dgvit_mlp_head_backward(const dgvit_mlp_desc* desc, const float* const* in, const float* const* params, const float* h1,
                            const float* h2, const float* const* dy, float* const* din, float* const* dparams, float* scratch,
                            long long scratch_floats, void* stream);

/* tanh-Gaussian action sampling of GoTPolicy.sample / GaussianPolicy.sample (got_sac_network.py:238-251, 310-321) in one launch:
 *   ls = clamp(log_std_raw, ls_min, ls_max); x = mean + exp(ls) * eps; y = tanh(x); action = y * scale + bias;
 *   log_prob (B) = sum_a [Normal(mean, exp(ls)).log_prob(x) - log(scale * (1 - y^2) + 1e-6)];  tanh_mean = tanh(mean) * scale + bias.
 * mean, log_std_raw, eps, action, tanh_mean: (B, A); scale, bias: `scale_n` = 1 (scalar) or A values; eps ~ N(0, 1) from the caller.
 * backward: gradients of action / log_prob / tanh_mean (NULL = none) -> dmean, dlog_std_raw (the clamp's mask included). */
int dgvit_tanh_gaussian_forward(const float* mean, const float* log_std_raw, const float* eps, const float* scale, const float* bias,
                                int scale_n, float ls_min, float ls_max, float* action, float* log_prob, float* tanh_mean, int B,
                                int A, void* stream);
int dgvit_tanh_gaussian_backward(const float* mean, const float* log_std_raw, const float* eps, const float* scale, int scale_n,
                                 float ls_min, float ls_max, const float* d_action, const float* d_log_prob, const float* d_tanh_mean,
                                 float* dmean, float* dlog_std_raw, int B, int A, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Per-operator entry points (used by the encoder above; exported for operator-level parity tests).
 * -------------------------------------------------------------------------------------------- */
/* generic GEMM C = op(A) op(B) (+ epilogue); layout 0 NT (A MxK, B NxK), 1 NN (A MxK, B KxN), 2 TN (A KxM, B KxN).
 * epilogue 0: C = acc + bias + res;  1: C = acc + bias, C2 = gelu(C);  2: C = acc * gelu'(aux);
 *          3: C = relu(acc + bias);  4: C = aux > 0 ? acc : 0.
 * layout 2 runs split over K with `scratch` slabs (dgvit_gemm_scratch_floats) and then reduces into C. */
long long dgvit_gemm_scratch_floats(int layout, int M, int N, int K);
int dgvit_gemm(int layout, int epilogue, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
               int N, int K, const float* bias, const float* res, int ldr, float* C2, int ldc2, const float* aux,
               int ldaux, float* scratch, long long scratch_floats, void* stream);

/* nn.LayerNorm(D), eps 1e-5 (GoalFormer.py:34,37): y, and the per-row mean / rstd saved for backward */
int dgvit_layernorm_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            int rows, int D, void* stream);
long long dgvit_layernorm_backward_scratch_floats(int rows, int D);
/* dx = dres + dLN(dy) (dres may be NULL), dgamma, dbeta */
int dgvit_layernorm_backward(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* dres, float* dx, float* dgamma, float* dbeta, float* scratch,
                             long long scratch_floats, int rows, int D, void* stream);

/* RMSNorm (GoalFormer.py:120-122) on `rows` vectors x[r*ldx .. +D) */
int dgvit_rmsnorm_forward(const float* x, long long ldx, const float* g, float* y, int rows, int D, void* stream);
long long dgvit_rmsnorm_backward_scratch_floats(int rows, int D);
int dgvit_rmsnorm_backward(const float* dy, const float* x, long long ldx, const float* g, float* dx, long long lddx,
                           float* dg, float* scratch, long long scratch_floats, int rows, int D, void* stream);

/* Attention core (GoalFormer.py:73-81): qkv (B, N, 3*H*dh) -> out (B, N, H*dh); scale dh^-1/2; N <= 288, dh 64 or 32.
 * lse (B, H, N): base-2 log-sum-exp of every scaled score row, written by forward (NULL = not kept) and needed by
 * backward, which recomputes the probabilities from it tile by tile (nothing of size N x N is ever stored). */
int dgvit_attention_forward(const float* qkv, float* out, float* lse, int B, int N, int H, int dh, void* stream);
int dgvit_attention_backward(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                             int B, int N, int H, int dh, void* stream);

/* 'b (h p1) (w p2) -> b (h w) (p1 p2)' (GoalFormer.py:138) */
int dgvit_patchify(const float* img, float* patches, int B, int image_h, int image_w, int patch_h, int patch_w,
                   void* stream);
/* in-place dropout with the encoder's Philox stream (n multiple of 4) */
int dgvit_dropout(float* x, long long n, unsigned long long seed, float keep, void* stream);

/* ----------------------------------------------------------------------------------------------
 * CNN feature stack of the shipped critic QNetwork and of GaussianPolicy
 * (got_sac_network.py:129-133,151-155 / 263-266,292-296):
 *   Conv2d(1,16,5,s2) ReLU Conv2d(16,64,5,s2) ReLU Conv2d(64,256,5,s2) ReLU AdaptiveAvgPool2d(1)
 * img (B, H, W) single channel -> feat (B, 256).  params / grads: conv1.weight (16,1,5,5), conv1.bias,
 * conv2.weight (64,16,5,5), conv2.bias, conv3.weight (256,64,5,5), conv3.bias in the reference's layouts.
 * `ws` keeps the three NHWC activations for backward; scratch holds im2col rows, packed weights, split-K slabs.
 * -------------------------------------------------------------------------------------------- */
long long dgvit_cnn_workspace_floats(int B, int H, int W);
long long dgvit_cnn_forward_scratch_floats(int B, int H, int W);
long long dgvit_cnn_backward_scratch_floats(int B, int H, int W);
int dgvit_cnn_forward(const float* img, const float* const* params, float* feat, float* ws, long long ws_floats,
                      float* scratch, long long scratch_floats, int B, int H, int W, void* stream);
int dgvit_cnn_backward(const float* img, const float* const* params, float* const* grads, const float* dfeat,
                       const float* ws, long long ws_floats, float* scratch, long long scratch_floats, int B, int H, int W,
                       void* stream);

/* ----------------------------------------------------------------------------------------------
 * the step BEFORE the path.  The reference samples numpy batches from cpprb and copies
 * (B,128,160) fp32 obs / next_obs to the device every step (DRL.py:375-386).  With the transitions resident in HBM
 * (dgvit_amd.replay.DeviceReplayBuffer) a sample is an index draw plus this gather:
 *   out[i][0..row_floats) = src[idx[i]][0..row_floats)      idx: int64 device array, row_floats % 4 == 0
 * -------------------------------------------------------------------------------------------- */
int dgvit_gather_rows(const float* src, const long long* idx, float* out, long long nsel, long long row_floats,
                      long long nrows, void* stream);

/* ----------------------------------------------------------------------------------------------
 * the depth-frame preprocessing in front of the path -- what env_lab.py does with OpenCV on the
 * host for every camera message (listener_callback :420-434: cv2.normalize MINMAX -> uint8, add_nose :78-89 (N(0, 50) noise,
 * clip, 5x5 Gaussian blur), blurring :69-76 (11x11 blur of the centre band)) and for every step (:295-299: cv2.resize to
 * 160x128, / 255).  Frames are fp32 (B, H, W) on the device; noise == NULL draws N(0, noise_level) on the device (Philox).
 * Parity against OpenCV is unpinned (cv2 is not installed in the build image; the oracle restates its published formulas).
 * -------------------------------------------------------------------------------------------- */
long long dgvit_depth_preprocess_scratch_floats(int B, int H, int W);
int dgvit_depth_to_state(const float* depth, const float* noise, float noise_level, unsigned long long seed, float* state,
                         float* scratch, long long scratch_floats, int B, int H, int W, int out_h, int out_w, void* stream);
/* the stages (operator-level tests): scratch of dgvit_depth_normalize_u8 = 128 * B floats; tmp of dgvit_gaussian_blur = B*H*W floats,
 * rows [row0, row1) are blurred (ksize 5 or 11, reflection inside the band), img may equal out */
int dgvit_depth_normalize_u8(const float* depth, float* out, float* scratch, long long scratch_floats, int B, int H, int W, void* stream);
int dgvit_noise_clip(const float* img, const float* noise, float* out, long long n, float noise_level, unsigned long long seed, void* stream);
int dgvit_gaussian_blur(const float* img, float* out, float* tmp, int B, int H, int W, int ksize, int row0, int row1, void* stream);
int dgvit_resize_bilinear(const float* img, float* out, int B, int Hs, int Ws, int Hd, int Wd, float scale, void* stream);

/* ----------------------------------------------------------------------------------------------
 * The step after the path: torch.optim.Adam.step over all tensors of a network
 * (DRL.py:401-403,412-414) and the Polyak target update target = target*(1-tau) + source*tau (utils.py:31-33),
 * each as ONE pass over flat fp32 buffers (n multiple of 4, 16-byte aligned; see dgvit_amd.optim).
 * Adam follows torch.optim.Adam: m,v updates, bias corrections with `step` (1-based), eps added to sqrt(v_hat),
 * weight_decay as L2 term added to the gradient.  step_dev (may be NULL): DEVICE pointer to the 1-based step
 * counter, overriding `step` (graph-capturable form: bias corrections are then computed in the kernel).
 * -------------------------------------------------------------------------------------------- */
int dgvit_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, long long step, const long long* step_dev, void* stream);
int dgvit_soft_update(float* target, const float* source, long long n, float tau, void* stream);

/* ----------------------------------------------------------------------------------------------
 * bf16 configuration (BASELINE.json config 5: 224x224 depth frames, 12-layer ViT-Base variant with goal token, bf16).
 * Same GoT.forward (GoalFormer.py:156-171) with bf16 STORAGE for every GEMM operand (LayerNorm outputs, qkv, attention
 * output, MLP hidden, the four weight matrices of each block and the patch weight) and fp32 everywhere else (residual
 * stream, LayerNorm statistics, biases, softmax, accumulation on v_mfma_f32_32x32x16_bf16, RMSNorm, output).
 * bf16 values are raw 16-bit patterns (unsigned short).  Needs dim_head 64 and dim, mlp_dim, patch pixels % 8 == 0.
 *   wpack: bf16 copies of the GEMM weights in one arena of dgvit_got_bf16_weight_elems elements
 *          [patch weight | per layer: to_qkv, to_out, fc1, fc2 and their transposes], refreshed with dgvit_got_pack_weights_bf16 whenever the
 *          fp32 master parameters change; `params` is the fp32 table of dgvit_got_forward (biases, norms, pos_embedding).
 *   workspace: dgvit_got_bf16_workspace_bytes BYTES, 256-byte aligned.
 * -------------------------------------------------------------------------------------------- */
long long dgvit_got_bf16_weight_elems(const dgvit_config* cfg);
/* with_transposes == 0 fills only the (out, in) copies the forward reads (inference); != 0 also the transposes the backward's
 * data-gradient GEMMs read.  The library keeps no state: call it (on the stream of the forward) whenever the fp32 masters may
 * have changed -- the in-tree host re-packs before every forward. */
int dgvit_got_pack_weights_bf16(const dgvit_config* cfg, const float* const* params, unsigned short* wpack,
                                long long wpack_elems, int with_transposes, void* stream);
long long dgvit_got_bf16_workspace_bytes(const dgvit_config* cfg, int batch, int save_for_backward);
int dgvit_got_forward_bf16(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack, const float* img,
                           const float* goal, float* feat, void* workspace, long long workspace_bytes, int batch,
                           int save_for_backward, float dropout_keep, unsigned long long dropout_seed,
                           const unsigned long long* dropout_seed_dev, void* stream);
/* Training in the bf16 configuration: forward with save_for_backward = 1 (dense last block), then
 *   dfeat (B, D) -> grads[] (fp32, table order of params, each written), dgoal (B, D) (may be NULL).
 * Gradients of GEMM operands travel bf16 (dY of every Linear), the residual-stream gradient, LayerNorm / bias / weight
 * gradients are fp32; weight gradients are fp32 split-K slabs summed in a fixed order (deterministic).
 * `img` is the forward's input (the patch-embedding weight gradient re-gathers the patches in fp32);
 * scratch: dgvit_got_bf16_backward_scratch_bytes BYTES, 256-byte aligned. */
long long dgvit_got_bf16_backward_scratch_bytes(const dgvit_config* cfg, int batch);
int dgvit_got_backward_bf16(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack, float* const* grads,
                            const float* dfeat, float* dgoal, const float* img, const void* workspace, long long workspace_bytes,
                            void* scratch, long long scratch_bytes, int batch, float dropout_keep, unsigned long long dropout_seed,
                            const unsigned long long* dropout_seed_dev, void* stream);
/* ... with gradient-ready events (see dgvit_grad_events) */
int dgvit_got_backward_bf16_ev(const dgvit_config* cfg, const float* const* params, const unsigned short* wpack, float* const* grads,
                               const float* dfeat, float* dgoal, const float* img, const void* workspace, long long workspace_bytes,
                               void* scratch, long long scratch_bytes, int batch, float dropout_keep, unsigned long long dropout_seed,
                               const unsigned long long* dropout_seed_dev, void* stream, const dgvit_grad_events* events);
/* operator-level entry points of the bf16 kernels (parity tests, benches) */
/* dW (Mo, Ko) fp32 = dY^T X and db (Mo, may be NULL) = column sums of dY, for dY (T, Mo) and X (T, Ko) bf16, token-major
 * (Mo, Ko % 8 == 0): the TN layout of the ring GEMM (transposed LDS reads), split over tokens into fp32 slabs that are summed in
 * a fixed order.  scratch: dgvit_wgrad_bf16_scratch_floats floats. */
long long dgvit_wgrad_bf16_scratch_floats(int Mo, int Ko, int T);
int dgvit_wgrad_bf16(const unsigned short* dY, const unsigned short* X, float* dW, float* db, float* scratch,
                     long long scratch_floats, int T, int Mo, int Ko, void* stream);
int dgvit_cast_f32_bf16(const float* src, unsigned short* dst, long long n, void* stream);
/* C = A B^T (+ epilogue), A (M,K) / B (N,K) bf16 with k contiguous, K, lda, ldb % 8 == 0, N, ldc % 4 == 0.
 * epilogue 0: C bf16 = acc + bias;  1: C bf16 = gelu(acc + bias);  2: C fp32 = acc + bias + res (fp32);
 *          3: C bf16 = acc * gelu'(aux bf16);  4: C fp32 = acc;  5: as 1 and C2 bf16 = acc + bias (pre-activation). */
int dgvit_gemm_bf16(int epilogue, const unsigned short* A, int lda, const unsigned short* B, int ldb, void* C, int ldc, int M,
                    int N, int K, const float* bias, const float* res, int ldr, unsigned short* C2, int ldc2,
                    const unsigned short* aux, int ldaux, void* stream);
/* LayerNorm with fp32 input and bf16 output (mean / rstd may be NULL) */
int dgvit_layernorm_forward_bf16(const float* x, const float* gamma, const float* beta, unsigned short* y, float* mean,
                                 float* rstd, int rows, int D, void* stream);
/* attention core on bf16 qkv (B, N, 3*H*64) -> bf16 out (B, N, H*64); lse fp32 (B, H, N) or NULL */
int dgvit_attention_forward_bf16(const unsigned short* qkv, unsigned short* out, float* lse, int B, int N, int H, int dh,
                                 void* stream);
/* gradient of the attention core: dqkv (B, N, 3*H*64) bf16 from qkv, the forward's out and lse, and dout; delta: B*H*N floats
 * of scratch (rowsum(dout o out), handed from the dQ kernel to the dK/dV kernel) */
int dgvit_attention_backward_bf16(const unsigned short* qkv, const unsigned short* out, const unsigned short* dout, const float* lse,
                                  unsigned short* dqkv, float* delta, int B, int N, int H, int dh, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Optional live kernel timing (HIP events on the launch stream around kernel launches).
 * kinds: 0 GEMM (work = 2*M*N*K FLOPs), 1 attention fwd, 2 attention bwd (work = algorithmic FLOPs),
 *        3 normalisation / reductions / elementwise (work = 0).
 * -------------------------------------------------------------------------------------------- */
#define DGVIT_PROFILE_KINDS 4
int dgvit_profile_start(int max_records);
int dgvit_profile_stop(double* ms, double* work, long long* launches);
/* Sampling (ABI 4).  An event pair around a launch keeps it from overlapping the tail of its predecessor and the ramp of its
 * successor; around EVERY launch that costs the C3 training step about 7 % (13.7 -> 14.8 ms).  dgvit_profile_sampling(s) times
 * every s-th launch of each kind only (s = 1: all; the setting persists); the sums of dgvit_profile_stop then cover the sampled
 * launches, and dgvit_profile_totals gives work and launch counts of ALL launches seen between start and stop. */
int dgvit_profile_sampling(int stride);
int dgvit_profile_totals(double* work_all, long long* launches_all);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* DGVIT_HIP_H */
