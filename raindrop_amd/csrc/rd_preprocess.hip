// rd_preprocess.hip -- the reference's host preprocessing on the device:
//   code/utils_rd.py:149-161  getStats             per-sensor mean / population std of the OBSERVED (> 0) values
//   code/utils_rd.py:164-175  mask_normalize       (x - mean) / (std + 1e-18) * mask  ++  mask
//   code/utils_rd.py:206-219  mask_normalize_static
//   code/utils_rd.py:221-257  tensorize_normalize / _other (float32 casts, minutes -> hours)
//   code/Raindrop.py:215-231  Setting-2/3 feature removal on the validation / test tensors
// minutes of numpy per split at P19 scale in the reference; here a handful of launches over data resident in HBM.
//
// Bit-exactness.  The reference computes in float64 numpy and casts to float32 at the end.  Every elementwise step
// is done here in float64 with the same operations in the same order (IEEE add / mul / div / sqrt are correctly
// rounded on gfx950; contraction to fma is switched off for this file), so those results are bit-identical.  The
// statistics are sums over up to millions of values: numpy's float64 add-reduction is a PAIRWISE sum with a fixed
// shape (8192-element buffers added in order; inside a buffer blocks of <= 128 elements summed with 8 interleaved
// accumulators, halves split at multiples of 8) and the
// result depends on that shape, so k_pairwise_* below reproduce it exactly: the leaves (independent, <= 128 elements)
// in parallel, 8 lanes per leaf = the 8 accumulators, and the (tiny) recursion tree above them by one thread per
// sensor in the same order numpy's recursion combines it.  Pinned against numpy itself on the GPU box (tests).
#pragma clang fp contract(off)
#include "rd_common.h"

namespace rd {
namespace {

constexpr int PW_BLOCK = 128;          // numpy PW_BLOCKSIZE
constexpr long NP_BUF = 8192;          // numpy's reduction runs through its buffered iterator: the inner (pairwise) loop sees
                                       // 8192 elements at a time and the chunk sums are accumulated in order (measured against
                                       // numpy 2.2: n = 8193 already differs from one pairwise tree over the whole array)
constexpr int CHUNK = 2048;            // rows per compaction workgroup

// ---- ordered compaction of the observed values of every sensor: vals[f][0..n_f) = P[rows with P > 0, f] in row order ----
__global__ __launch_bounds__(256) void k_prep_count(const double* __restrict__ P, long NT, int F, int* __restrict__ cnt) {
  extern __shared__ int sc[];                      // [F]
  for (int f = threadIdx.x; f < F; f += 256) sc[f] = 0;
  __syncthreads();
  const long r0 = (long)blockIdx.x * CHUNK, r1 = min(NT, r0 + CHUNK);
  for (long e = r0 * F + threadIdx.x; e < r1 * F; e += 256)
    if (P[e] > 0.0) atomicAdd(&sc[(int)(e % F)], 1);           // integer LDS atomics: order-independent
  __syncthreads();
  for (int f = threadIdx.x; f < F; f += 256) cnt[(long)blockIdx.x * F + f] = sc[f];
}

// exclusive scan over the chunks, per sensor; total[f] = n_f
__global__ void k_prep_scan(int* __restrict__ cnt, long nchunk, int F, long* __restrict__ total) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  long acc = 0;
  for (long c = 0; c < nchunk; ++c) {
    const int v = cnt[c * F + f];
    cnt[c * F + f] = (int)acc;                     // chunk-local counts are < 2^31; offsets too (NT < 2^31 checked by the host)
    acc += v;
  }
  total[f] = acc;
}

__global__ __launch_bounds__(256) void k_prep_compact(const double* __restrict__ P, long NT, int F, const int* __restrict__ off,
                                                      double* __restrict__ vals) {
  __shared__ int pre[256];
  const long r0 = (long)blockIdx.x * CHUNK;
  const int tid = threadIdx.x;
  constexpr int RPT = CHUNK / 256;                 // 8 consecutive rows per thread: order inside a thread, then across threads
  for (int f = 0; f < F; ++f) {
    double v[RPT]; int c = 0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const long r = r0 + (long)tid * RPT + i;
      v[i] = r < NT ? P[r * F + f] : 0.0;
      c += v[i] > 0.0;
    }
    pre[tid] = c;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {            // inclusive Hillis-Steele scan of the 256 counts
      const int add = tid >= o ? pre[tid - o] : 0;
      __syncthreads();
      pre[tid] += add;
      __syncthreads();
    }
    long w = (long)off[(long)blockIdx.x * F + f] + pre[tid] - c;
    double* dst = vals + (long)f * NT;
#pragma unroll
    for (int i = 0; i < RPT; ++i)
      if (v[i] > 0.0) dst[w++] = v[i];
    __syncthreads();
  }
}

// ---- numpy's pairwise sum -------------------------------------------------------------------------------------------
// tree walk shared by the three passes: calls leaf(lo, n) for every leaf in order / combines in numpy's order
struct Frame { long lo, n; int state; double left; };

// one thread per sensor: leaves in order -> leaf table (lo, n); nleaf[f]
__global__ void k_pw_leaves(const long* __restrict__ total, int F, long leaf_cap, long* __restrict__ leaf_lo,
                            int* __restrict__ leaf_n, long* __restrict__ nleaf) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const long n = total[f];
  long* lo_out = leaf_lo + (long)f * leaf_cap; int* n_out = leaf_n + (long)f * leaf_cap;
  long cnt = 0;
  for (long c0 = 0; c0 < n; c0 += NP_BUF) {
    long slo[64], sn[64]; int sp = 0;
    slo[0] = c0; sn[0] = min(NP_BUF, n - c0); sp = 1;
    while (sp > 0) {
      --sp;
      const long lo = slo[sp], m = sn[sp];
      if (m <= PW_BLOCK) { lo_out[cnt] = lo; n_out[cnt] = (int)m; ++cnt; continue; }
      long n2 = m / 2; n2 -= n2 % 8;
      slo[sp] = lo + n2; sn[sp] = m - n2; ++sp;      // right pushed first: left is popped (visited) first
      slo[sp] = lo; sn[sp] = n2; ++sp;
    }
  }
  nleaf[f] = cnt;
}

// leaf sums: 8 lanes per leaf = numpy's 8 interleaved accumulators.  mode 0: sum of a[i]; mode 1: sum of (a[i] - mean)^2
// with the square formed exactly as numpy does (x = a - mean; x = x * x; then summed)
__global__ __launch_bounds__(256) void k_pw_leafsum(const double* __restrict__ vals, long NT, int F, long leaf_cap,
                                                    const long* __restrict__ leaf_lo, const int* __restrict__ leaf_n,
                                                    const long* __restrict__ nleaf, const double* __restrict__ mean, int mode,
                                                    double* __restrict__ leaf_sum) {
  const int f = blockIdx.y;
  const long leaf = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int j = threadIdx.x & 7;
  const bool live = leaf < nleaf[f];
  const long li = live ? leaf : 0;
  const double* a = vals + (long)f * NT + leaf_lo[(long)f * leaf_cap + li];
  const int n = live ? leaf_n[(long)f * leaf_cap + li] : 0;
  const double mu = mode ? mean[f] : 0.0;
  auto val = [&](int i) { double x = a[i]; if (mode) { x = x - mu; x = x * x; } return x; };
  double res;
  if (n < 8) {                                     // plain loop from 0.0 (lane 0 only matters)
    res = 0.0;
    for (int i = 0; i < n; ++i) res += val(i);
  } else {
    double r = val(j);
    const int nb = n - (n % 8);
    for (int i = 8; i < nb; i += 8) r += val(i + j);
    // res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
    double s = r + __shfl_down(r, 1, 8);           // lanes 0,2,4,6: r_j + r_{j+1}
    s = s + __shfl_down(s, 2, 8);                  // lanes 0,4: (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
    res = s + __shfl_down(s, 4, 8);                // lane 0
    for (int i = nb; i < n; ++i) res += val(i);    // the tail is added by lane 0 (all lanes compute, lane 0 stores)
  }
  if (live && j == 0) leaf_sum[(long)f * leaf_cap + leaf] = res;
}

// one thread per sensor: combine the leaf sums in numpy's recursion order (post-order: left + right)
__global__ void k_pw_combine(const long* __restrict__ total, int F, long leaf_cap, const double* __restrict__ leaf_sum,
                             int mode, double* __restrict__ mean, double* __restrict__ stdv) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const long n = total[f];
  const double* ls = leaf_sum + (long)f * leaf_cap;
  long next = 0;
  double sum = 0.0;                                // np.add.reduce starts from the identity and adds the chunk sums in order
  for (long c0 = 0; c0 < n; c0 += NP_BUF) {
    Frame st[64]; int sp = 0;
    st[0] = Frame{c0, min(NP_BUF, n - c0), 0, 0.0}; sp = 1;
    double ret = 0.0;
    while (sp > 0) {
      Frame& fr = st[sp - 1];
      if (fr.n <= PW_BLOCK) { ret = ls[next++]; --sp; continue; }
      long n2 = fr.n / 2; n2 -= n2 % 8;
      if (fr.state == 0) { fr.state = 1; st[sp++] = Frame{fr.lo, n2, 0, 0.0}; }
      else if (fr.state == 1) { fr.left = ret; fr.state = 2; st[sp++] = Frame{fr.lo + n2, fr.n - n2, 0, 0.0}; }
      else { ret = fr.left + ret; --sp; }
    }
    sum = sum + ret;
  }
  if (mode == 0) mean[f] = sum / (double)n;        // np.mean: sum / count (nan for an empty sensor, like numpy)
  else {
    const double sd = sqrt(sum / (double)n);       // np.std, ddof = 0
    stdv[f] = (1e-7 > sd) ? 1e-7 : sd;             // max(std, eps) with Python's max semantics (nan stays nan)
  }
}

// ---- elementwise passes ---------------------------------------------------------------------------------------------
// P [N,T,F] f64 -> out f32: values (x - mf)/(stdf + 1e-18) * mask in channels [0,F), mask in [F,2F).
// layout 0: out [N,T,2F] (utils_rd.tensorize_normalize); layout 1: out [T,N,2F] (the permute of code/Raindrop.py:232-238 fused)
__global__ __launch_bounds__(256) void k_prep_normalize(const double* __restrict__ P, long N, int T, int F,
                                                        const double* __restrict__ mf, const double* __restrict__ stdf,
                                                        float* __restrict__ out, int layout) {
  const long total = N * T * (long)F;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int f = (int)(e % F);
    const long nt = e / F;
    const double x = P[e];
    const double m = x > 0.0 ? 1.0 : 0.0;
    const double v = (x - mf[f]) / (stdf[f] + 1e-18) * m;
    long row = nt;
    if (layout == 1) { const long n = nt / T; const int t = (int)(nt - n * T); row = (long)t * N + n; }
    out[row * (2 * F) + f] = (float)v;
    out[row * (2 * F) + F + f] = (float)m;
  }
}

__global__ __launch_bounds__(256) void k_prep_static(const double* __restrict__ S, long total, int D, const double* __restrict__ ms,
                                                     const double* __restrict__ ss, float* __restrict__ out) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int d = (int)(e % D);
    double v = (S[e] - ms[d]) / (ss[d] + 1e-18);
    if (v <= 0.0) v = 0.0;
    out[e] = (float)v;
  }
}

// minutes (f64) -> hours (f32): torch.Tensor(time) / 60.0  (float32 division).  layout 1: [N,T] -> [T,N]
__global__ __launch_bounds__(256) void k_prep_time(const double* __restrict__ minutes, long N, int T, float* __restrict__ out, int layout) {
  const long total = N * T;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const float h = (float)minutes[e] / 60.0f;
    long o = e;
    if (layout == 1) { const long n = e / T; const int t = (int)(e - n * T); o = (long)t * N + n; }
    out[o] = h;
  }
}

// Setting 2 / 3: zero `k` value channels of every sample.  P f32 [N,T,2F] (layout 0) or [T,N,2F] (layout 1);
// idx int32 [N,k] (per_sample) or [k] (the same set for everybody).  The mask half is left untouched, like the reference.
__global__ __launch_bounds__(256) void k_prep_remove(float* __restrict__ P, long N, int T, int F, const int* __restrict__ idx, int k,
                                                     int per_sample, int layout) {
  const long total = N * T * (long)k;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int j = (int)(e % k);
    const long nt = e / k;
    const long n = nt / T; const int t = (int)(nt - n * T);
    const int f = idx[per_sample ? n * k + j : j];
    const long row = layout == 1 ? (long)t * N + n : nt;
    if (f >= 0 && f < F) P[row * (2 * F) + f] = 0.f;
  }
}

struct PrepWs { int* cnt; long* total; double* vals; long* leaf_lo; int* leaf_n; long* nleaf; double* leaf_sum; long leaf_cap, nchunk; size_t bytes; };
PrepWs carve_prep(long NT, int F, void* base) {
  PrepWs w; size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? (void*)((char*)base + off) : nullptr; off += align_up(bytes, 256); return p; };
  w.nchunk = (NT + CHUNK - 1) / CHUNK;
  w.leaf_cap = NT / 60 + 8;                         // leaves of the recursion hold > 60 elements each once n > 128
  w.cnt = (int*)take((size_t)w.nchunk * F * sizeof(int));
  w.total = (long*)take((size_t)F * sizeof(long));
  w.vals = (double*)take((size_t)NT * F * sizeof(double));
  w.leaf_lo = (long*)take((size_t)w.leaf_cap * F * sizeof(long));
  w.leaf_n = (int*)take((size_t)w.leaf_cap * F * sizeof(int));
  w.nleaf = (long*)take((size_t)F * sizeof(long));
  w.leaf_sum = (double*)take((size_t)w.leaf_cap * F * sizeof(double));
  w.bytes = off;
  return w;
}

int grid_for(long total) { long g = (total + 255) / 256; return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g)); }

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" size_t rd_prep_stats_workspace_bytes(int64_t NT, int32_t F) {
  if (NT <= 0 || F <= 0) return 0;
  return carve_prep(NT, F, nullptr).bytes;
}

extern "C" int rd_prep_stats(int64_t NT, int32_t F, const double* P, double* mf, double* stdf, void* workspace,
                             size_t workspace_bytes, void* stream) {
  RD_REQUIRE(NT > 0 && F > 0 && NT < (1L << 31), "bad dims NT=%ld F=%d", (long)NT, F);
  RD_REQUIRE(P && mf && stdf && workspace, "NULL tensor");
  PrepWs w = carve_prep(NT, F, workspace);
  RD_REQUIRE(workspace_bytes >= w.bytes, "workspace too small: %zu < %zu", workspace_bytes, w.bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_prep_count, dim3((unsigned)w.nchunk), dim3(256), F * sizeof(int), st, P, (long)NT, F, w.cnt);
  hipLaunchKernelGGL(k_prep_scan, dim3(cdiv(F, 64)), dim3(64), 0, st, w.cnt, w.nchunk, F, w.total);
  hipLaunchKernelGGL(k_prep_compact, dim3((unsigned)w.nchunk), dim3(256), 0, st, P, (long)NT, F, w.cnt, w.vals);
  hipLaunchKernelGGL(k_pw_leaves, dim3(cdiv(F, 64)), dim3(64), 0, st, w.total, F, w.leaf_cap, w.leaf_lo, w.leaf_n, w.nleaf);
  const dim3 lgrid((unsigned)((w.leaf_cap * 8 + 255) / 256), (unsigned)F);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k_pw_leafsum, lgrid, dim3(256), 0, st, w.vals, (long)NT, F, w.leaf_cap, w.leaf_lo, w.leaf_n, w.nleaf, mf, mode,
                       w.leaf_sum);
    hipLaunchKernelGGL(k_pw_combine, dim3(cdiv(F, 64)), dim3(64), 0, st, w.total, F, w.leaf_cap, w.leaf_sum, mode, mf, stdf);
  }
  return check_launch("rd_prep_stats");
}

extern "C" int rd_prep_mask_normalize(int64_t N, int32_t T, int32_t F, const double* P, const double* mf, const double* stdf,
                                      float* out, int32_t layout, void* stream) {
  RD_REQUIRE(N >= 0 && T > 0 && F > 0 && (layout == 0 || layout == 1), "bad arguments");
  if (N == 0) return RD_OK;
  RD_REQUIRE(P && mf && stdf && out, "NULL tensor");
  hipLaunchKernelGGL(k_prep_normalize, dim3(grid_for(N * T * (long)F)), dim3(256), 0, (hipStream_t)stream, P, (long)N, T, F, mf, stdf,
                     out, layout);
  return check_launch("k_prep_normalize");
}

extern "C" int rd_prep_static(int64_t N, int32_t D, const double* S, const double* ms, const double* ss, float* out, void* stream) {
  RD_REQUIRE(N >= 0 && D > 0, "bad arguments");
  if (N == 0) return RD_OK;
  RD_REQUIRE(S && ms && ss && out, "NULL tensor");
  hipLaunchKernelGGL(k_prep_static, dim3(grid_for(N * (long)D)), dim3(256), 0, (hipStream_t)stream, S, N * (long)D, D, ms, ss, out);
  return check_launch("k_prep_static");
}

extern "C" int rd_prep_time(int64_t N, int32_t T, const double* minutes, float* hours, int32_t layout, void* stream) {
  RD_REQUIRE(N >= 0 && T > 0 && (layout == 0 || layout == 1), "bad arguments");
  if (N == 0) return RD_OK;
  RD_REQUIRE(minutes && hours, "NULL tensor");
  hipLaunchKernelGGL(k_prep_time, dim3(grid_for(N * (long)T)), dim3(256), 0, (hipStream_t)stream, minutes, (long)N, T, hours, layout);
  return check_launch("k_prep_time");
}

extern "C" int rd_prep_remove_features(int64_t N, int32_t T, int32_t F, float* P, const int32_t* idx, int32_t k, int32_t per_sample,
                                       int32_t layout, void* stream) {
  RD_REQUIRE(N >= 0 && T > 0 && F > 0 && k >= 0 && (layout == 0 || layout == 1), "bad arguments");
  if (N == 0 || k == 0) return RD_OK;
  RD_REQUIRE(P && idx, "NULL tensor");
  hipLaunchKernelGGL(k_prep_remove, dim3(grid_for(N * T * (long)k)), dim3(256), 0, (hipStream_t)stream, P, (long)N, T, F, idx, k,
                     per_sample, layout);
  return check_launch("k_prep_remove");
}
