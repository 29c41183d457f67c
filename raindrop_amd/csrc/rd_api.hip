// rd_api.hip -- library identity, error reporting and the generic dense entry points.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "rd_common.h"
#include "rd_rng.h"
#include "rd_trailing.h"

namespace rd {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

// Registered per HOST THREAD and read when a call is ENQUEUED (the pointer becomes a kernel argument): a captured hipGraph
// keeps the pointer it was captured with, so callers register the cell around the enqueue / capture only.
static thread_local const uint64_t* g_seed_cell = nullptr;
const uint64_t* seed_cell() { return g_seed_cell; }
static thread_local AdamCellReg g_adam_cell = {nullptr, 0.f, 0.f};
AdamCellReg adam_cell() { return g_adam_cell; }

static int g_precision = -1;
// per-call override on THIS host thread (rd_linear_fwd_fp32): the mode is read when a launch is chosen, at enqueue
static thread_local int g_precision_override = -1;
int precision() {
  if (g_precision_override >= 0) return g_precision_override;
  if (g_precision < 0) {
    const char* e = getenv("RD_PRECISION");
    g_precision = (e && strcmp(e, "fp32") == 0) ? RD_PREC_FP32 : ((e && strcmp(e, "bf16") == 0) ? RD_PREC_BF16 : RD_PREC_BF16X3);
  }
  return g_precision;
}

static thread_local hipStream_t g_side = nullptr;
static thread_local bool g_side_busy = false;          // something was forked since the last join
static hipEvent_t side_event(int which) {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  if (!ev[which] && hipEventCreateWithFlags(&ev[which], hipEventDisableTiming) != hipSuccess) ev[which] = nullptr;
  return ev[which];
}
hipStream_t side_fork(hipStream_t main_stream) {
  if (!g_side || g_side == main_stream) return main_stream;
  hipEvent_t e = side_event(0);
  if (!e || hipEventRecord(e, main_stream) != hipSuccess || hipStreamWaitEvent(g_side, e, 0) != hipSuccess) return main_stream;
  g_side_busy = true;
  return g_side;
}
int side_join(hipStream_t main_stream) {
  if (!g_side || !g_side_busy || g_side == main_stream) return RD_OK;
  hipEvent_t e = side_event(1);
  if (!e) return fail(RD_EINVAL, "side_join: no event");
  RD_HIP(hipEventRecord(e, g_side));
  RD_HIP(hipStreamWaitEvent(main_stream, e, 0));
  g_side_busy = false;
  return RD_OK;
}

// ---- trailing launches parked for the next backward chain launch (rd_trailing.h) ----
int launch_twg_reduce_standalone(const TwArgs& a, int nblocks, hipStream_t st);
int launch_head_wgrad_standalone(const HwArgs& h, hipStream_t st);
static thread_local bool g_defer = false;
static thread_local RiderArgs g_parked{};
bool trailing_deferred() { return g_defer; }
int trailing_launch(const RiderArgs& r, hipStream_t st) {
  if (r.kind == RIDER_TWG) return launch_twg_reduce_standalone(r.tw, r.nblocks, st);
  if (r.kind == RIDER_HEAD) return launch_head_wgrad_standalone(r.hw, st);
  return RD_OK;
}
int trailing_park(const RiderArgs& r, hipStream_t st) {
  if (g_parked.kind != RIDER_NONE) {                       // nobody picked the previous one up: it runs now, on its own
    const RiderArgs old = g_parked;
    g_parked.kind = RIDER_NONE;
    const int rc = trailing_launch(old, st);
    if (rc) return rc;
  }
  g_parked = r;
  return RD_OK;
}
RiderArgs trailing_take() {
  RiderArgs r = g_parked;
  g_parked.kind = RIDER_NONE;
  if (!g_defer) r.kind = RIDER_NONE;                       // a slot left behind by a caller that has switched the mode off is dropped, never launched
  return r;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace rd

using namespace rd;

extern "C" int rd_set_precision(int32_t mode) {
  RD_REQUIRE(mode == RD_PREC_FP32 || mode == RD_PREC_BF16X3 || mode == RD_PREC_BF16, "unknown precision mode %d", mode);
  g_precision = mode;
  return RD_OK;
}
extern "C" int rd_get_precision(void) { return precision(); }

namespace {
__global__ void k_seed_advance(uint64_t* cell, uint64_t delta) { *cell += delta; }
}
extern "C" int rd_set_seed_cell(const uint64_t* device_cell) { g_seed_cell = device_cell; return RD_OK; }
extern "C" int rd_set_adam_state(void* state, float beta1, float beta2) {
  RD_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15) == 0, "misaligned optimizer state");
  g_adam_cell = AdamCellReg{(double*)state, beta1, beta2};
  return RD_OK;
}
// Switching the mode OFF discards whatever is still parked: the slot holds raw device pointers of the step that parked it, and a
// caller that leaves the mode through an error path (a capture that raised) must not have them enqueued -- or baked into a later
// graph -- by the next chain launch of this host thread.  A normal caller has flushed (rd_flush_trailing) before, so nothing is lost.
extern "C" int rd_set_defer_trailing(int32_t on) {
  g_defer = on != 0;
  if (!g_defer) g_parked.kind = RIDER_NONE;
  return RD_OK;
}
extern "C" int rd_drop_trailing(void) { g_parked.kind = RIDER_NONE; return RD_OK; }
extern "C" int rd_flush_trailing(void* stream) {
  if (g_parked.kind == RIDER_NONE) return RD_OK;
  const RiderArgs r = trailing_take();
  return trailing_launch(r, (hipStream_t)stream);
}
extern "C" int rd_set_side_stream(void* stream) {
  g_side = (hipStream_t)stream; g_side_busy = false;
  if (stream == nullptr) g_parked.kind = RIDER_NONE;       // same rule as rd_set_defer_trailing(0)
  return RD_OK;
}
extern "C" int rd_side_join(void* main_stream) { return side_join((hipStream_t)main_stream); }
extern "C" int rd_seed_cell_advance(uint64_t* device_cell, uint64_t delta, void* stream) {
  RD_REQUIRE(device_cell != nullptr, "NULL cell");
  hipLaunchKernelGGL(k_seed_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, device_cell, delta);
  return check_launch("k_seed_advance");
}

namespace {
// out[i] = x[i] * scale * (keep ? 1/(1-p) : 0): nn.Dropout (after a scalar scale) as a pure function of (seed, site, element), four
// elements per thread; p == 0 is the plain scale
__global__ __launch_bounds__(256) void k_dropout(const float* __restrict__ x, float* __restrict__ out, long n, float scale, float p,
                                                 uint64_t seed, uint32_t site, const uint64_t* cell) {
  seed = eff_seed(seed, cell);
  const float inv_keep = scale / (1.0f - p);
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; 4 * q < n; q += (long)gridDim.x * blockDim.x) {
    float uu[4] = {1.f, 1.f, 1.f, 1.f};
    if (p > 0.f) { const float4 u = uniform4(seed, site, (uint64_t)q); uu[0] = u.x; uu[1] = u.y; uu[2] = u.z; uu[3] = u.w; }
    for (int c = 0; c < 4 && 4 * q + c < n; ++c) out[4 * q + c] = uu[c] >= p ? x[4 * q + c] * inv_keep : 0.f;
  }
}
}  // namespace
// The backward of dropout is the same call on the gradient (same seed, same site): the mask is regenerated, not stored.
extern "C" int rd_scale_dropout(int64_t n, const float* x, float scale, float p_drop, uint64_t seed, uint32_t site, float* out,
                                void* stream) {
  RD_REQUIRE(n >= 0 && p_drop >= 0.f && p_drop < 1.f, "bad n=%ld / p_drop", (long)n);
  if (n == 0) return RD_OK;
  RD_REQUIRE(x && out, "NULL tensor");
  long blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_dropout, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)n, scale, p_drop, seed, site, seed_cell());
  return check_launch("k_dropout");
}

extern "C" int rd_version(void) { return RD_ABI_VERSION; }
extern "C" const char* rd_arch(void) { return "gfx950"; }
extern "C" const char* rd_last_error(void) { return err_buf(); }

extern "C" int rd_linear_fwd(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx,
                             const float* W, const float* b, float* y, int32_t ldy, int32_t act,
                             void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  if (M == 0) return RD_OK;
  RD_REQUIRE(x && W && y, "NULL tensor");
  RD_REQUIRE(ldx >= K && ldy >= N, "leading dimension too small");
  RD_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (relu)");
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.nsplit = 1;
  g.A = x; g.sa_m = ldx; g.sa_k = 1;
  g.B = W; g.sb_n = K; g.sb_k = 1;
  g.C = y; g.sc_m = ldy;
  g.bias = b; g.relu = act;
  return launch_gemm(g, (hipStream_t)stream);
}

// rd_linear_fwd on the exact-fp32 matrix instruction whatever the process's arithmetic mode: for values that feed INDEX work (the
// use_beta branch's edge scores -> top-K pruning; index work is bit-exact by contract).  The override is thread-local and lasts for
// this enqueue only: no other host thread ever sees a changed mode (the Python side used to toggle rd_set_precision).
extern "C" int rd_linear_fwd_fp32(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx,
                                  const float* W, const float* b, float* y, int32_t ldy, int32_t act, void* stream) {
  struct Scope { Scope() { g_precision_override = RD_PREC_FP32; } ~Scope() { g_precision_override = -1; } } scope;
  return rd_linear_fwd(M, N, K, x, ldx, W, b, y, ldy, act, stream);
}

extern "C" int rd_linear_bwd_input(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                                   const float* W, float* dx, int32_t lddx, void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  if (M == 0) return RD_OK;
  RD_REQUIRE(dy && W && dx, "NULL tensor");
  RD_REQUIRE(lddy >= N && lddx >= K, "leading dimension too small");
  GemmArgs g{};
  g.M = M; g.N = K; g.K = N; g.nsplit = 1;
  g.A = dy; g.sa_m = lddy; g.sa_k = 1;
  g.B = W; g.sb_n = 1; g.sb_k = K;   // B(n=k_out, k=n_red) = W[n_red*K + k_out]
  g.C = dx; g.sc_m = lddx;
  return launch_gemm(g, (hipStream_t)stream);
}

// dx = (dy W) gated by gate > 0: the ReLU between two Linear layers folded into the dgrad product
extern "C" int rd_linear_bwd_input_gated(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                                         const float* W, const float* gate, int32_t ldgate, float* dx,
                                         int32_t lddx, void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  if (M == 0) return RD_OK;
  RD_REQUIRE(dy && W && gate && dx, "NULL tensor");
  RD_REQUIRE(lddy >= N && lddx >= K && ldgate >= K, "leading dimension too small");
  GemmArgs g{};
  g.M = M; g.N = K; g.K = N; g.nsplit = 1;
  g.A = dy; g.sa_m = lddy; g.sa_k = 1;
  g.B = W; g.sb_n = 1; g.sb_k = K;
  g.C = dx; g.sc_m = lddx;
  g.posmask = gate; g.pm_m = ldgate;
  return launch_gemm(g, (hipStream_t)stream);
}

namespace {
// mean cross entropy over B rows of C logits and its gradient (softmax - onehot) / B, one workgroup,
// fixed-order tree reduction (code/Raindrop.py:255,322: CrossEntropyLoss().forward + backward)
__global__ __launch_bounds__(256) void k_softmax_xent(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                      float* __restrict__ loss, float* __restrict__ dlogits, int B, int C) {
  __shared__ float red[256];
  float acc = 0.f;
  const float invB = 1.0f / (float)B;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* row = logits + (long)b * C;
    float m = row[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(row[c] - m);
    const float lse = m + logf(se);
    const int t = (int)y[b];
    acc += lse - row[t];
    for (int c = 0; c < C; ++c) dlogits[(long)b * C + c] = (expf(row[c] - lse) - (c == t ? 1.f : 0.f)) * invB;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] * invB;
}
}  // namespace

extern "C" int rd_softmax_xent(int32_t B, int32_t C, const float* logits, const int64_t* y, float* loss,
                               float* dlogits, void* stream) {
  RD_REQUIRE(B > 0 && C > 0, "bad dims B=%d C=%d", B, C);
  RD_REQUIRE(logits && y && loss && dlogits, "NULL tensor");
  hipLaunchKernelGGL(k_softmax_xent, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, y, loss, dlogits, B, C);
  return check_launch("k_softmax_xent");
}

extern "C" size_t rd_linear_bwd_weight_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  return align_up((size_t)wgrad_ws_floats(M, N, K) * sizeof(float), 256);
}

extern "C" int rd_linear_bwd_weight(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                                    const float* x, int32_t ldx, float* dW, float* db,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    RD_REQUIRE(dW != nullptr, "NULL tensor");
    RD_HIP(hipMemsetAsync(dW, 0, sizeof(float) * N * K, st));
    if (db) RD_HIP(hipMemsetAsync(db, 0, sizeof(float) * N, st));
    return RD_OK;
  }
  RD_REQUIRE(dy && x && dW, "NULL tensor");
  RD_REQUIRE(lddy >= N && ldx >= K, "leading dimension too small");
  RD_REQUIRE(workspace && workspace_bytes >= rd_linear_bwd_weight_workspace_bytes(M, N, K),
             "workspace too small");
  return launch_wgrad(M, N, K, dy, lddy, x, ldx, dW, db, (float*)workspace, st);
}
