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
//   [.. + T + 1]      coff[r], r = 0..B     first 16-ROW GROUP of rank r in the PER-SAMPLE group space: rank r owns ceil(len_r / 16)
//                                           groups, two groups make a 32-row chunk of the weight-gradient stream (group g = rows
//                                           16 (g & 1) .. of chunk g >> 1); [5] = ceil(coff[B] / 2) = chunks.  This is the row
//                                           order in which the fused attention kernels (rd_attnfuse.hip) export x and dqkv as
//                                           row tiles -- a sample's rows start a group there, so the workgroup that owns one
//                                           sample writes whole 16-byte tile slots (rows past its length: zeros; the unowned
//                                           second half of the last chunk is zeroed by the owner of the last group)
//   [.. + B + 1]      brow[b], b = 0..B-1   first row of SAMPLE b (= off[rank[b]]) and
//   [.. + B]          blen[b]               its clamped length: one independent look-up each for kernels that walk the caller's
//                                           sample order (the unfused message passing's scatter / gather, rd_pe_mask)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace rd {
namespace plan {

constexpr int HDR = 8;
constexpr int I_MLIVE = 0, I_S32 = 1, I_B = 2, I_T = 3, I_SLACK = 4, I_SCHUNK = 5;

__host__ __device__ inline int off_base() { return HDR; }
__host__ __device__ inline int rank_base(int B) { return HDR + B + 1; }
__host__ __device__ inline int order_base(int B) { return HDR + 2 * B + 1; }
__host__ __device__ inline int len_base(int B) { return HDR + 3 * B + 1; }
__host__ __device__ inline int cnt_base(int B) { return HDR + 4 * B + 1; }
__host__ __device__ inline int coff_base(int B, int T) { return HDR + 4 * B + 1 + T + 1; }
__host__ __device__ inline int brow_base(int B, int T) { return HDR + 5 * B + 2 + T + 1; }
__host__ __device__ inline int blen_base(int B, int T) { return HDR + 6 * B + 2 + T + 1; }
__host__ __device__ inline size_t ints(int B, int T) { return (size_t)HDR + 7 * (size_t)B + 2 + T + 1; }


#if defined(__HIPCC__)
// LDS ints the plan needs: len [B], cnt [T + 2], within [B], cc [ceil(B / 64)][T + 1]
__host__ __device__ inline size_t lds_bytes(int B, int T) { return ((size_t)2 * B + T + 2 + (size_t)((B + 63) / 64) * (T + 1)) * sizeof(int); }

// The whole plan by ONE workgroup of any size (a multiple of 64 threads).  rank of b = (samples strictly longer) + (equally long
// samples with a smaller index): a stable counting sort made of ballots and prefix sums only, so nothing depends on an execution
// order.  Wave w takes the 64-sample chunks w, w + nw, ..: for every length value v one ballot gives the chunk's count of v and,
// by the lower-lane bits, every member's position among the chunk's v's; the chunk counts are prefix-summed over the chunks per v
// (thread per v).  Also bumps the dropout seed cell (one launch fewer per step).
__device__ inline void token_plan_body(const int64_t* __restrict__ lengths, int32_t* __restrict__ p, int B, int T, uint64_t* seed_cell,
                                       uint64_t delta, int* psm) {
  const int nthr = blockDim.x, nw = nthr >> 6;
  const int nchunk = (B + 63) >> 6, T1 = T + 1;
  int* len = psm;                 // [B]
  int* cnt = psm + B;             // [T + 2]: histogram, then cnt[t] = #(len > t)
  int* within = cnt + T + 2;      // [B]: equally long samples with a smaller index inside the same chunk
  int* cc = within + B;           // [nchunk][T + 1]: count of v in chunk, then in the chunks before it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = tid; b < B; b += nthr) {
    const int64_t l = lengths[b];
    len[b] = (int)(l < 0 ? 0 : (l > T ? T : l));
  }
  __syncthreads();
  for (int ch = wave; ch < nchunk; ch += nw) {
    const int b = ch * 64 + lane;
    const int l = b < B ? len[b] : -1;
    int sl = 0;
    for (int v = 0; v <= T; ++v) {
      const unsigned long long m = __ballot(l == v);
      if (l == v) sl = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) cc[ch * T1 + v] = __popcll(m);
    }
    if (b < B) within[b] = sl;
  }
  __syncthreads();
  for (int v = tid; v <= T; v += nthr) {
    int run = 0;
    for (int ch = 0; ch < nchunk; ++ch) { const int t = cc[ch * T1 + v]; cc[ch * T1 + v] = run; run += t; }
    cnt[v] = run;
  }
  __syncthreads();
  // cnt[t] = number of samples with len > t  (T + 1 entries, cnt[T] = 0): exclusive suffix sum of the histogram
  if (wave == 0) {
    if (T1 <= 64) {
      const int h = lane <= T ? cnt[lane] : 0;
      int sfx = h;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_down(sfx, o); if (lane + o < 64) sfx += v; }
      if (lane <= T) cnt[lane] = sfx - h;
    } else if (lane == 0) {
      int above = 0;
      for (int t = T; t >= 0; --t) { const int h = cnt[t]; cnt[t] = above; above += h; }
    }
    if (lane == 0 && seed_cell) *seed_cell += delta;
  }
  __syncthreads();
  int* off = p + plan::off_base();
  int* rank = p + plan::rank_base(B);
  int* order = p + plan::order_base(B);
  int* lenr = p + plan::len_base(B);
  int* cntg = p + plan::cnt_base(B);
  for (int b = tid; b < B; b += nthr) {
    const int l = len[b];
    const int r = cnt[l] + cc[(b >> 6) * T1 + l] + within[b];
    rank[b] = r; order[r] = b; lenr[r] = l;
    int s = 0;                                             // off[r], per sample (the off[] loop below recomputes it per rank)
    for (int t = 0; t < T; ++t) s += min(r, cnt[t]);
    p[plan::brow_base(B, T) + b] = s; p[plan::blen_base(B, T) + b] = l;
  }
  // off[r] = sum of the r longest lengths = sum_t min(r, cnt[t])   (the samples with len > t are the first cnt[t] ranks)
  for (int r = tid; r <= B; r += nthr) {
    int s = 0;
    for (int t = 0; t < T; ++t) s += min(r, cnt[t]);
    off[r] = s;
    if (r == B) { p[plan::I_MLIVE] = s; p[plan::I_S32] = (s + 31) >> 5; }
  }
  for (int t = tid; t <= T; t += nthr) cntg[t] = cnt[t];
  // coff[r] = sum over the r longest samples of ceil(len / 16) = sum_g min(r, cnt[16 g])   (a sample has a group g iff len > 16 g)
  int* coff = p + plan::coff_base(B, T);
  for (int r = tid; r <= B; r += nthr) {
    int s = 0;
    for (int t = 0; t < T; t += 16) s += min(r, cnt[t]);
    coff[r] = s;
    if (r == B) p[plan::I_SCHUNK] = (s + 1) >> 1;
  }
  if (tid == 0) { p[plan::I_B] = B; p[plan::I_T] = T; p[plan::I_SLACK] = 0; p[6] = 0; p[7] = 0; }
}
#endif

}  // namespace plan

// plan registered by rd_set_token_plan on this host thread (nullptr: padded layout); read when a call is ENQUEUED
const int32_t* token_plan();

}  // namespace rd
