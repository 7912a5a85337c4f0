// Internal prototypes of the launch functions implemented in the kernel translation units
// (norm.hip, attention.hip, embed.hip, conv.hip, optim.hip, gemm.hip); the schedules in dgvit_api.hip call these.
#pragma once
#include "common.h"

int layernorm_fwd(const float*, const float*, const float*, float*, float*, float*, int, int, float, int, hipStream_t);
int layernorm_bwd_blocks(int T);
int layernorm_bwd(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*,
                  float*, int, int, int, hipStream_t, ReduceGroup* grp = nullptr);
int rmsnorm_fwd(const float*, long long, const float*, float*, int, int, hipStream_t);
int rmsnorm_bwd_blocks(int B);
int rmsnorm_bwd(const float*, const float*, long long, const float*, float*, long long, float*, float*, int, int, hipStream_t);
int colsum_blocks(int T);
int colsum(const float*, long long, float*, float*, int, int, int, hipStream_t);
int attention_fwd(const float*, float*, float*, int, int, int, int, int, hipStream_t);
int attention_bwd(const float*, const float*, const float*, const float*, float*, int, int, int, int, int, hipStream_t);
int patchify(const float*, float*, int, int, int, int, int, hipStream_t);
int add_rows(const float*, long long, const float*, long long, float*, long long, long long, int, hipStream_t);
int goal_row(const float*, const float*, float*, int, int, int, hipStream_t);
int dropout_inplace(float*, long long, unsigned long long, const unsigned long long*, float, hipStream_t);
int relu_bwd(const float*, const float*, float*, long long, hipStream_t);
int adam_step(float*, const float*, float*, float*, long long, float, float, float, float, float, long long, const long long*,
              hipStream_t);
int soft_update(float*, const float*, long long, float, hipStream_t);
int im2col(const float*, float*, int, int, int, int, int, int, int, hipStream_t);
int col2im_relu(const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);
int weight_pack(const float*, float*, int, int, int, int, hipStream_t);
int avgpool(const float*, float*, int, int, int, hipStream_t);
int avgpool_bwd_relu(const float*, const float*, float*, int, int, int, hipStream_t);
int gather_rows(const float*, const long long*, float*, long long, long long, long long, hipStream_t);
int mean_bwd(const float*, float*, int, int, int, hipStream_t);
int depth_normalize_u8(const float*, float*, float*, int, int, int, hipStream_t);
long long depth_normalize_scratch_floats(int);
int noise_clip(const float*, const float*, float*, long long, float, unsigned long long, hipStream_t);
int gaussian_blur_band(const float*, float*, float*, int, int, int, int, int, int, hipStream_t);
int resize_bilinear(const float*, float*, int, int, int, int, int, float, hipStream_t);
int mlp_head_forward(const dgvit_mlp_desc*, const float* const*, const float* const*, float*, float*, float*, hipStream_t);
long long mlp_head_backward_scratch(const dgvit_mlp_desc*);
int mlp_head_backward(const dgvit_mlp_desc*, const float* const*, const float* const*, const float*, const float*, const float* const*,
                      float* const*, float* const*, float*, long long, hipStream_t);
int tanh_gaussian_forward(const float*, const float*, const float*, const float*, const float*, int, float, float, float*, float*, float*,
                          int, int, hipStream_t);
int tanh_gaussian_backward(const float*, const float*, const float*, const float*, int, float, float, const float*, const float*,
                           const float*, float*, float*, int, int, hipStream_t);
// block.hip: two launches per transformer block for small batches (no-grad forward), cross-workgroup sums inside the launches
bool block_path_supports(int B, int N, int D, int H, int dh, int M);
long long block_path_slab_floats(int B, int N, int D, int H, int M);
long long block_path_counters(int B, int N);
struct BlockFirst {     // block 0 assembling its own token rows (block.hip)
  const float* goal; const float* pos0; float* xres;
  float keep; unsigned long long seed; const unsigned long long* seed_dev;
};
int block_path_layer(const float* x, const float* ln1, float* xout, float* ln1_out, const float* const* lp, const float* const* next_ln,
                     int token0_only, float* slabs, int* counters, const BlockFirst* first, const float* rms_g, float* feat, int B, int N, int D,
                     int H, int dh, int M, hipStream_t st);
// (frame.hip: diagnostic build only)
bool frame_path_supports(int B, int N, int D, int H, int dh, int M);
long long frame_path_scratch_floats(int B, int N, int D, int H, int M);
int frame_path_forward(const float* x0, const float* const* params, int L, float* scratch, float* feat, int B, int N, int D, int H, int dh,
                       int M, hipStream_t st);
