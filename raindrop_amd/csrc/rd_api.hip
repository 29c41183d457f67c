// rd_api.hip -- library identity, error reporting and the generic dense entry points.
#include <stdarg.h>
#include <string.h>

#include "rd_common.h"

namespace rd {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
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

extern "C" int rd_version(void) { return RD_ABI_VERSION; }
extern "C" const char* rd_arch(void) { return "gfx950"; }
extern "C" const char* rd_last_error(void) { return err_buf(); }

extern "C" int rd_linear_fwd(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx,
                             const float* W, const float* b, float* y, int32_t ldy, int32_t act,
                             void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
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

extern "C" int rd_linear_bwd_input(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                                   const float* W, float* dx, int32_t lddx, void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  RD_REQUIRE(dy && W && dx, "NULL tensor");
  RD_REQUIRE(lddy >= N && lddx >= K, "leading dimension too small");
  GemmArgs g{};
  g.M = M; g.N = K; g.K = N; g.nsplit = 1;
  g.A = dy; g.sa_m = lddy; g.sa_k = 1;
  g.B = W; g.sb_n = 1; g.sb_k = K;   // B(n=k_out, k=n_red) = W[n_red*K + k_out]
  g.C = dx; g.sc_m = lddx;
  return launch_gemm(g, (hipStream_t)stream);
}

namespace {
void bwd_weight_plan(int M, int N, int K, int* nsplit, int* kps) {
  const int tiles = cdiv(N, 64) * cdiv(K, 64);
  int ns = cdiv(512, tiles);
  int per = (int)align_up((size_t)cdiv(M > 0 ? M : 1, ns), 32);
  *kps = per;
  *nsplit = cdiv(M > 0 ? M : 1, per);
}
}  // namespace

extern "C" size_t rd_linear_bwd_weight_workspace_bytes(int32_t M, int32_t N, int32_t K) {
  if (M < 0 || N <= 0 || K <= 0) return 0;
  int ns, kps;
  bwd_weight_plan(M, N, K, &ns, &kps);
  return align_up((size_t)ns * N * K * sizeof(float), 256) +
         align_up((size_t)colsum_ws_floats(M, N) * sizeof(float), 256) + 256;
}

extern "C" int rd_linear_bwd_weight(int32_t M, int32_t N, int32_t K, const float* dy, int32_t lddy,
                                    const float* x, int32_t ldx, float* dW, float* db,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  RD_REQUIRE(M >= 0 && N > 0 && K > 0, "bad dims M=%d N=%d K=%d", M, N, K);
  RD_REQUIRE(dy && x && dW, "NULL tensor");
  RD_REQUIRE(lddy >= N && ldx >= K, "leading dimension too small");
  RD_REQUIRE(workspace && workspace_bytes >= rd_linear_bwd_weight_workspace_bytes(M, N, K),
             "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    hipMemsetAsync(dW, 0, sizeof(float) * N * K, st);
    if (db) hipMemsetAsync(db, 0, sizeof(float) * N, st);
    return RD_OK;
  }
  int ns, kps;
  bwd_weight_plan(M, N, K, &ns, &kps);
  float* part = (float*)workspace;
  float* csws = (float*)((char*)workspace + align_up((size_t)ns * N * K * sizeof(float), 256));
  GemmArgs t{};
  t.M = N; t.N = K; t.K = M;
  t.A = dy; t.sa_m = 1; t.sa_k = lddy;
  t.B = x; t.sb_n = 1; t.sb_k = ldx;
  t.nsplit = ns; t.k_per_split = kps;
  int rc;
  if (ns > 1) {
    t.C = part; t.sc_m = K; t.sc_split = (long)N * K;
    if ((rc = launch_gemm(t, st))) return rc;
    if ((rc = launch_splitk_reduce(part, ns, (long)N * K, dW, st))) return rc;
  } else {
    t.C = dW; t.sc_m = K;
    if ((rc = launch_gemm(t, st))) return rc;
  }
  if (db) return launch_colsum(dy, M, N, lddy, db, csws, st);
  return RD_OK;
}
