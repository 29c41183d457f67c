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
// LDS ints a plan workgroup needs: the clamped lengths [B], padded to a multiple of 4
__host__ __device__ inline size_t lds_bytes(int B, int T) { (void)T; return (size_t)((B + 3) / 4 * 4) * sizeof(int); }

// integer sum over the 64 lanes of a wave on the DPP path (rd_common.h's wave_sum64_dpp for ints), result uniform
__device__ __forceinline__ int wave_isum64(int v) {
#define RD_DPP_IADD(ctrl, rmask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)
  RD_DPP_IADD(0x111, 0xf); RD_DPP_IADD(0x112, 0xf); RD_DPP_IADD(0x114, 0xf); RD_DPP_IADD(0x118, 0xf);
  RD_DPP_IADD(0x142, 0xa); RD_DPP_IADD(0x143, 0xc);
#undef RD_DPP_IADD
  return __builtin_amdgcn_readlane(v, 63);
}

// The plan, by `nparts` workgroups of any size (a multiple of 64 threads) that do not talk to each other: every output is a
// function of the length vector alone.  rank of b = (samples strictly longer) + (equally long samples with a smaller index);
// a WAVE counts that for one sample -- lanes stride over the other samples, three DPP sums -- and the same pass sums the lengths
// and the 16-row groups of the samples in front of it: its first row and first group.  Part `part` takes the samples (and, for
// cnt[], the steps) part * waves + wave, + nparts * waves, ...  Nothing depends on an execution order.
// (Round 4.  Rounds 3's form -- one workgroup, ballots over the T + 1 length values, prefix and suffix sums behind four barriers,
// three 60-step dependent loops -- took ~10 us and was the critical path of the step's first launch; the first rewrite, one THREAD
// per sample scanning all B lengths, was 14 instructions x B per thread and no faster.)
// Part 0 also writes the header and bumps the dropout seed cell (one launch fewer per step).
__device__ inline void token_plan_part(const int64_t* __restrict__ lengths, int32_t* __restrict__ p, int B, int T, uint64_t* seed_cell,
                                       uint64_t delta, int* psm, int part, int nparts) {
  const int nthr = blockDim.x, tid = threadIdx.x, lane = tid & 63, nw = nthr >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B4 = (B + 3) >> 2;
  int* len = psm;                 // [4 B4], entries >= B hold -1 (shorter than everything, counted by nobody)
  for (int b = tid; b < 4 * B4; b += nthr) {
    int l = -1;
    if (b < B) { const int64_t v = lengths[b]; l = (int)(v < 0 ? 0 : (v > T ? T : v)); }
    len[b] = l;
  }
  if (part == 0 && tid == 0) {
    p[plan::I_B] = B; p[plan::I_T] = T; p[plan::I_SLACK] = 0; p[6] = 0; p[7] = 0;
    if (seed_cell) *seed_cell += delta;
  }
  __syncthreads();
  int* off = p + plan::off_base();
  int* rank = p + plan::rank_base(B);
  int* order = p + plan::order_base(B);
  int* lenr = p + plan::len_base(B);
  int* cntg = p + plan::cnt_base(B);
  int* coff = p + plan::coff_base(B, T);
  for (int b = part * nw + wave; b < B; b += nparts * nw) {       // wave-uniform
    const int l = len[b];
    int r = 0, row = 0, grp = 0;
    for (int o = lane; o < 4 * B4; o += 64) {
      const int lv = len[o];
      const bool before = lv > l || (lv == l && o < b);
      r += before ? 1 : 0;
      row += before ? lv : 0;
      grp += before ? (lv + 15) >> 4 : 0;
    }
    r = wave_isum64(r); row = wave_isum64(row); grp = wave_isum64(grp);
    if (lane == 0) {
      rank[b] = r; order[r] = b; lenr[r] = l;
      off[r] = row; coff[r] = grp;               // first row / first 16-row group of rank r = sums over the r samples in front
      p[plan::brow_base(B, T) + b] = row; p[plan::blen_base(B, T) + b] = l;
      if (r == B - 1) {                          // the last rank closes both prefix arrays
        const int m = row + l, g = grp + ((l + 15) >> 4);
        off[B] = m; coff[B] = g;
        p[plan::I_MLIVE] = m; p[plan::I_S32] = (m + 31) >> 5; p[plan::I_SCHUNK] = (g + 1) >> 1;
      }
    }
  }
  // cnt[t] = number of samples with len > t  (T + 1 entries, cnt[T] = 0)
  for (int t = part * nw + wave; t <= T; t += nparts * nw) {
    int c = 0;
    for (int o = lane; o < 4 * B4; o += 64) c += len[o] > t ? 1 : 0;
    c = wave_isum64(c);
    if (lane == 0) cntg[t] = c;
  }
}
#endif

}  // namespace plan

// plan registered by rd_set_token_plan on this host thread (nullptr: padded layout); read when a call is ENQUEUED
const int32_t* token_plan();

}  // namespace rd
