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

// PL_PARTS independent workgroups; the body is rd_plan.h's token_plan_part (shared with the step's combined first launch, rd_step_begin).
constexpr int PL_PARTS = 16;
__global__ __launch_bounds__(PL_THR) void k_token_plan(const int64_t* __restrict__ lengths, int32_t* __restrict__ p, int B, int T,
                                                      uint64_t* seed_cell, uint64_t delta) {
  extern __shared__ __attribute__((aligned(16))) int psm[];
  plan::token_plan_part(lengths, p, B, T, seed_cell, delta, psm, blockIdx.x, gridDim.x);
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
  const size_t lds = plan::lds_bytes(s->B, s->T);
  RD_REQUIRE(lds <= 64 * 1024, "rd_token_plan: B x T too large for one workgroup (%d, %d)", s->B, s->T);
  hipLaunchKernelGGL(k_token_plan, dim3(PL_PARTS), dim3(PL_THR), lds, (hipStream_t)stream, lengths, plan_out, s->B, s->T, seed_cell, delta);
  return check_launch("k_token_plan");
}

extern "C" int rd_set_token_plan(const int32_t* device_plan) {
  g_token_plan = device_plan;
  return RD_OK;
}
