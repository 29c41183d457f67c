// rd_plan.hip -- rd_token_plan: lengths [B] -> the token plan (rd_plan.h), one small launch per step.
//
// Integer work only: a counting sort of the samples by length (descending, ties by sample index) and the prefix sums of
// the sorted lengths.  Everything downstream that walks tokens reads this instead of the [B,T] padding mask of
// code/models_rd.py:298-299 -- the plan carries the same information (mask[b,t] == (t >= len_b)).
#include "rd_common.h"
#include "rd_plan.h"

namespace rd {

static thread_local const int32_t* g_token_plan = nullptr;
const int32_t* token_plan() { return g_token_plan; }

namespace {

constexpr int PL_THR = 1024;

// One workgroup.  rank of b = (samples strictly longer) + (equally long samples with a smaller index): a stable counting sort
// made of ballots and prefix sums only, so nothing depends on an execution order.  Wave w takes the 64-sample chunks w, w+16, ..:
// for every length value v one ballot gives the chunk's count of v and, by the lower-lane bits, every member's position among
// the chunk's v's; the chunk counts are prefix-summed over the chunks per v (thread per v).
__global__ __launch_bounds__(PL_THR) void k_token_plan(const int64_t* __restrict__ lengths, int32_t* __restrict__ p, int B, int T,
                                                      uint64_t* seed_cell, uint64_t delta) {
  extern __shared__ int psm[];
  const int nchunk = (B + 63) >> 6, T1 = T + 1;
  int* len = psm;                 // [B]
  int* cnt = psm + B;             // [T + 2]: histogram, then cnt[t] = #(len > t)
  int* within = cnt + T + 2;      // [B]: equally long samples with a smaller index inside the same chunk
  int* cc = within + B;           // [nchunk][T + 1]: count of v in chunk, then in the chunks before it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = tid; b < B; b += PL_THR) {
    const int64_t l = lengths[b];
    len[b] = (int)(l < 0 ? 0 : (l > T ? T : l));
  }
  __syncthreads();
  for (int ch = wave; ch < nchunk; ch += PL_THR / 64) {
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
  for (int v = tid; v <= T; v += PL_THR) {
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
  for (int b = tid; b < B; b += PL_THR) {
    const int l = len[b];
    const int r = cnt[l] + cc[(b >> 6) * T1 + l] + within[b];
    rank[b] = r; order[r] = b; lenr[r] = l;
  }
  // off[r] = sum of the r longest lengths = sum_t min(r, cnt[t])   (the samples with len > t are the first cnt[t] ranks)
  for (int r = tid; r <= B; r += PL_THR) {
    int s = 0;
    for (int t = 0; t < T; ++t) s += min(r, cnt[t]);
    off[r] = s;
    if (r == B) { p[plan::I_MLIVE] = s; p[plan::I_S32] = (s + 31) >> 5; }
  }
  for (int t = tid; t <= T; t += PL_THR) cntg[t] = cnt[t];
  if (tid == 0) { p[plan::I_B] = B; p[plan::I_T] = T; p[plan::I_SLACK] = 0; p[5] = 0; p[6] = 0; p[7] = 0; }
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" size_t rd_token_plan_bytes(const rd_shape* s) {
  if (!s || s->B < 0 || s->T <= 0) return 0;
  return align_up(plan::ints(s->B, s->T) * sizeof(int32_t), 256);
}

extern "C" int rd_token_plan(const rd_shape* s, const int64_t* lengths, int32_t* plan_out, uint64_t* seed_cell,
                             uint64_t delta, void* stream) {
  RD_REQUIRE(s && s->B > 0 && s->T > 0, "bad rd_shape");
  RD_REQUIRE(lengths && plan_out, "NULL tensor");
  const size_t lds = ((size_t)2 * s->B + s->T + 2 + (size_t)((s->B + 63) / 64) * (s->T + 1)) * sizeof(int);
  RD_REQUIRE(lds <= 64 * 1024, "rd_token_plan: B x T too large for one workgroup (%d, %d)", s->B, s->T);
  hipLaunchKernelGGL(k_token_plan, dim3(1), dim3(PL_THR), lds, (hipStream_t)stream, lengths, plan_out, s->B, s->T, seed_cell, delta);
  return check_launch("k_token_plan");
}

extern "C" int rd_set_token_plan(const int32_t* device_plan) {
  g_token_plan = device_plan;
  return RD_OK;
}
