// rd_feed.hip -- batch feed from a device-resident dataset.
//
// Replaces the per-step host fancy-index + H2D of code/Raindrop.py:310-315
//   P, Ptime, Pstatic, y = Ptrain_tensor[:, idx, :].cuda(), Ptrain_time_tensor[:, idx].cuda(),
//                          Ptrain_static_tensor[idx].cuda(), ytrain_tensor[idx].cuda()
// and `lengths = torch.sum(Ptime > 0, dim=0)` (:317) with ONE launch over tensors that already live in
// HBM: pure byte movement (bit-exact), bound by HBM: 2 * B * T * (W + 1) * 4 bytes per batch
// (P19, B=256: 8.5 MB -> 1.1 us at 8 TB/s).
#include "rd_common.h"

namespace rd {
namespace {

constexpr int FD_RPB = 16;        // batch rows per copy workgroup: 16 rows x 16 lanes x 16 bytes per pass

struct FeedArgs {
  const float *P_all, *time_all, *static_all;
  const int64_t *y_all, *idx;
  float *src, *times, *static_out;
  int64_t *y_out, *lengths;
  int32_t* bad;
  long N;
  int T, B, W, ds, ncopy, nbg, vec;
};

// an out-of-range index is counted (integer atomic) and clamped; the caller decides what to do with the count
__device__ __forceinline__ long checked_index(const FeedArgs& a, int b, bool count) {
  long n = a.idx[b];
  if (n < 0 || n >= a.N) {
    if (count && a.bad) atomicAdd(a.bad, 1);
    n = n < 0 ? 0 : a.N - 1;
  }
  return n;
}

__global__ __launch_bounds__(256) void k_batch_gather(FeedArgs a) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.ncopy) {
    // ---- [T, N, W] -> [T, B, W] rows and [T, N] -> [T, B] time stamps -----------------------------------
    const int t = blockIdx.x / a.nbg, bg = blockIdx.x - t * a.nbg;
    const int r = tid >> 4, l = tid & 15;
    const int b = bg * FD_RPB + r;
    if (b >= a.B) return;
    const long n = checked_index(a, b, false);
    const float* s = a.P_all + ((long)t * a.N + n) * a.W;
    float* d = a.src + ((long)t * a.B + b) * a.W;
    if (a.vec) {
      for (int c = 4 * l; c < a.W; c += 64) *reinterpret_cast<float4*>(d + c) = *reinterpret_cast<const float4*>(s + c);
    } else {
      for (int c = l; c < a.W; c += 16) d[c] = s[c];
    }
    if (l == 0) a.times[(long)t * a.B + b] = a.time_all[(long)t * a.N + n];
    return;
  }
  // ---- per-sample tail, one wavefront per sample: lengths (code/Raindrop.py:317) as a wave-wide count over
  // the time axis (one round trip instead of a T-long dependent loop), labels, static features -------------
  const int lane = tid & 63;
  const int b = (blockIdx.x - a.ncopy) * 4 + (tid >> 6);
  if (b >= a.B) return;
  const long n = checked_index(a, b, lane == 0);
  int cnt = 0;
  for (int t = lane; t < a.T; t += 64) cnt += a.time_all[(long)t * a.N + n] > 0.f ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) {
    a.lengths[b] = (int64_t)cnt;
    if (a.y_all) a.y_out[b] = a.y_all[n];
  }
  if (a.static_all)
    for (int j = lane; j < a.ds; j += 64) a.static_out[(long)b * a.ds + j] = a.static_all[n * a.ds + j];
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" int rd_batch_gather(int32_t T, int32_t B, int32_t W, int32_t d_static, int64_t N, const float* P_all,
                               const float* time_all, const float* static_all, const int64_t* y_all,
                               const int64_t* idx, float* src, float* times, float* static_out, int64_t* y_out,
                               int64_t* lengths, int32_t* bad_index_count, void* stream) {
  RD_REQUIRE(T >= 0 && B >= 0 && W > 0 && d_static >= 0 && N > 0, "bad dims T=%d B=%d W=%d d_static=%d N=%ld", T, B, W,
             d_static, (long)N);
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return RD_OK;
  RD_REQUIRE(P_all && time_all && idx && src && times && lengths, "NULL tensor");
  RD_REQUIRE(d_static == 0 || static_all == nullptr || static_out != nullptr, "static_all given without static_out");
  RD_REQUIRE((y_all == nullptr) == (y_out == nullptr), "y_all and y_out must be given together");
  FeedArgs a{};
  a.P_all = P_all; a.time_all = time_all; a.static_all = d_static > 0 ? static_all : nullptr; a.y_all = y_all; a.idx = idx;
  a.src = src; a.times = times; a.static_out = static_out; a.y_out = y_out; a.lengths = lengths; a.bad = bad_index_count;
  a.N = N; a.T = T; a.B = B; a.W = W; a.ds = d_static;
  a.nbg = cdiv(B, FD_RPB);
  a.ncopy = T * a.nbg;
  a.vec = ((W & 3) == 0 && ((reinterpret_cast<uintptr_t>(P_all) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) ? 1 : 0;
  const int ntail = cdiv(B, 4);
  hipLaunchKernelGGL(k_batch_gather, dim3(a.ncopy + ntail), dim3(256), 0, st, a);
  return check_launch("k_batch_gather");
}
