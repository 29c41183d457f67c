// rd_plan.h -- the "token plan": which (sample, time step) pairs of a batch are live, and where they are stored.
//
// The reference pads every sample to max_len steps and masks the padded steps out again: as attention KEYS
// (src_key_padding_mask, code/models_rd.py:298-299,358) and in the masked mean (code/models_rd.py:366-367,379).  A padded
// step therefore never reaches the logits -- nor, by the same two masks, any gradient: its row of every encoder activation
// gradient is exactly zero.  With a plan registered (rd_set_token_plan) the step kernels store and process only the live
// rows: sample b's steps t < len_b sit at rows off[rank_b] + t of every [tokens, *] tensor, samples ordered by descending
// length ("rank"; ties by sample index, so the layout and every reduction order are deterministic).  Logits, loss and all
// parameter gradients are the same function of the inputs as on the padded layout; what differs is that the padded rows,
// which nothing ever reads, do not exist.
//
// Layout of the plan (int32, device memory; written by rd_token_plan, read by every kernel that takes it):
//   [0] M_live = sum_b len_b          [1] ceil(M_live / 32)      [2] B      [3] T
//   [4] input slack: max over samples of (1 + last step with a non-zero observation) - len_b, clamped at 0; rd_token_plan
//       zeroes it and the fused message-passing forward raises it (integer atomicMax: order-independent)
//   [8 ..]            off[r], r = 0..B      first row of rank r (off[B] = M_live)
//   [8 + (B+1) ..]    rank[b]               rank of sample b
//   [.. + B]          order[r]              sample at rank r
//   [.. + B]          len[r]                clamp(lengths[order[r]], 0, T)
//   [.. + B]          cnt[t], t = 0..T      number of samples with len > t
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rd {
namespace plan {

constexpr int HDR = 8;
constexpr int I_MLIVE = 0, I_S32 = 1, I_B = 2, I_T = 3, I_SLACK = 4;

__host__ __device__ inline int off_base() { return HDR; }
__host__ __device__ inline int rank_base(int B) { return HDR + B + 1; }
__host__ __device__ inline int order_base(int B) { return HDR + 2 * B + 1; }
__host__ __device__ inline int len_base(int B) { return HDR + 3 * B + 1; }
__host__ __device__ inline int cnt_base(int B) { return HDR + 4 * B + 1; }
__host__ __device__ inline size_t ints(int B, int T) { return (size_t)HDR + 4 * (size_t)B + 1 + T + 1; }

}  // namespace plan

// plan registered by rd_set_token_plan on this host thread (nullptr: padded layout); read when a call is ENQUEUED
const int32_t* token_plan();

}  // namespace rd
