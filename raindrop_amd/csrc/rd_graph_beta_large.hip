// rd_graph_beta_large.hip -- the use_beta graph operator of Observation_progation (code/Ob_propagation.py:161-185,190-191,195,
// 200,207-208,227) for graphs that do not fit one workgroup's LDS: any N <= 1024 nodes and any number of edges
// (SYN256: 256 sensors, 65 536 edges, T = 512 steps -- the per-step scores of ONE sample are 512 KB, its sort keys another 512 KB).
//
// Same operator, same order of every floating-point sum as the LDS-staged kernels of rd_graph_beta.hip (edges of a source in
// pruning order, time steps in order), as a sequence of element-parallel launches over (sample, node, step) or (sample, edge)
// with the per-sample state in a caller-provided workspace:
//   scores beta [N,T] -> sort keys [P2] -> bitonic sort (4096-key chunks in LDS, strides >= 4096 in global memory) -> kept edges
//   in pruning order -> per-source lists (stable: pruning order) -> softmax statistics per (source, step) -> aggregation;
//   backward: lists by source AND by target, statistics, S = sum_e weight dweight per (source, step), dV, dbeta -> dH, d map_weights,
//   d edge weight.
// The pruning sort's keys are unique (score bits | edge id), so any correct sort yields the same order as the small kernel's:
// descending score, ties by edge id.
#include "rd_graph_beta.h"

namespace rd {
namespace {

constexpr int BL_THR = 256;
constexpr int BL_CH = 4096;              // keys per LDS-sorted chunk (32 KB)
constexpr int BL_LCH = 8192;             // list building: keys staged per pass (32 KB)
constexpr int BL_LTHR = 1024;            // list building: one thread per node

// byte offsets of one sample's state inside its workspace slice
struct WsLayout { size_t keys, ksrc, ktgt, kw, soff, slist, toff, tlist, mx, inv, S, db, stride; };
__host__ __device__ inline WsLayout ws_layout(int N, int T, int P2, int Kk) {
  WsLayout l; size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t kk = (size_t)(Kk > 0 ? Kk : 1), nt = (size_t)N * T;
  l.keys = take((size_t)P2 * 8);
  l.ksrc = take(kk * 4); l.ktgt = take(kk * 4); l.kw = take(kk * 4);
  l.soff = take((size_t)(N + 1) * 4); l.slist = take(kk * 4);
  l.toff = take((size_t)(N + 1) * 4); l.tlist = take(kk * 4);
  l.mx = take(nt * 4); l.inv = take(nt * 4); l.S = take(nt * 4); l.db = take(nt * 4);
  l.stride = off;
  return l;
}

struct LArgs { BetaArgs a; unsigned char* ws; WsLayout L; int P2; };

template <typename Tp>
__device__ __forceinline__ Tp* wsp(const LArgs& g, int b, size_t off) { return reinterpret_cast<Tp*>(g.ws + (size_t)b * g.L.stride + off); }

// ---- beta[i][t] = mean over 32 channels of increase_dim(x_i)[t] * cat(map_weights[i], p_t[t]) -> beta_save -------------------
__global__ __launch_bounds__(BL_THR) void k_bl_beta(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T;
  const long i = (long)blockIdx.x * BL_THR + threadIdx.x;
  if (i >= (long)N * T) return;
  const int n = (int)(i / T), t = (int)(i - (long)n * T);
  const float* h = a.H + ((size_t)b * N * T + i) * 32;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += h[c] * a.map_w[n * 16 + c];
#pragma unroll
  for (int c = 0; c < 16; ++c) s += h[16 + c] * pt[t * 16 + c];
  a.beta_save[(size_t)b * N * T + i] = s * (1.0f / 32.0f);
}

// mean score of edge e of sample b: sum over the steps in order, then / T (the small kernel's expression)
__device__ __forceinline__ float edge_mean(const float* beta, int tg, int T, float w) {
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += beta[(size_t)tg * T + t] * w;
  return s / (float)T;
}

__global__ __launch_bounds__(BL_THR) void k_bl_keys(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y;
  const int e = blockIdx.x * BL_THR + threadIdx.x;
  if (e >= g.P2) return;
  unsigned long long key = ~0ull;                                     // padding sorts last
  if (e < a.E) {
    const int tg = node_of(a.ei[a.ei_stride + e], a.N);
    const float s = edge_mean(a.beta_save + (size_t)b * a.N * a.T, tg, a.T, a.w[(size_t)b * a.w_bstride + e]);
    key = ((unsigned long long)sortable_desc(s) << 32) | (unsigned)e;
  }
  wsp<unsigned long long>(g, b, g.L.keys)[e] = key;
}

// ---- bitonic sort, ascending keys.  Compare-exchange (i, i ^ j) of stage k goes up where (i & k) == 0, i the GLOBAL index -----
// all stages k <= chunk size of one chunk, in LDS
__global__ __launch_bounds__(1024) void k_bl_sort_local(LArgs g, int chunk) {
  __shared__ unsigned long long sk[BL_CH];
  unsigned long long* keys = wsp<unsigned long long>(g, blockIdx.y, g.L.keys) + (size_t)blockIdx.x * chunk;
  const int base = blockIdx.x * chunk;
  for (int i = threadIdx.x; i < chunk; i += 1024) sk[i] = keys[i];
  __syncthreads();
  for (int k = 2; k <= chunk; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < chunk; i += 1024) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = sk[i], y = sk[p];
          const bool up = ((base + i) & k) == 0;
          if ((x > y) == up) { sk[i] = y; sk[p] = x; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < chunk; i += 1024) keys[i] = sk[i];
}
// one step (k, j) with j >= chunk size: partners live in different chunks -> global memory
__global__ __launch_bounds__(BL_THR) void k_bl_merge_global(LArgs g, int k, int j) {
  unsigned long long* keys = wsp<unsigned long long>(g, blockIdx.y, g.L.keys);
  const int i = blockIdx.x * BL_THR + threadIdx.x;
  if (i >= g.P2) return;
  const int p = i ^ j;
  if (p > i) {
    const unsigned long long x = keys[i], y = keys[p];
    const bool up = (i & k) == 0;
    if ((x > y) == up) { keys[i] = y; keys[p] = x; }
  }
}
// the steps j < chunk size of stage k (> chunk size), per chunk in LDS
__global__ __launch_bounds__(1024) void k_bl_merge_local(LArgs g, int k) {
  __shared__ unsigned long long sk[BL_CH];
  unsigned long long* keys = wsp<unsigned long long>(g, blockIdx.y, g.L.keys) + (size_t)blockIdx.x * BL_CH;
  const int base = blockIdx.x * BL_CH;
  for (int i = threadIdx.x; i < BL_CH; i += 1024) sk[i] = keys[i];
  __syncthreads();
  for (int j = BL_CH >> 1; j > 0; j >>= 1) {
    for (int i = threadIdx.x; i < BL_CH; i += 1024) {
      const int p = i ^ j;
      if (p > i) {
        const unsigned long long x = sk[i], y = sk[p];
        const bool up = ((base + i) & k) == 0;
        if ((x > y) == up) { sk[i] = y; sk[p] = x; }
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < BL_CH; i += 1024) keys[i] = sk[i];
}

// ---- kept edges in pruning order (forward: from the sorted keys; backward: from the saved edge ids) ---------------------------
__global__ __launch_bounds__(BL_THR) void k_bl_kept(LArgs g, int bwd) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, Kk = a.Kk;
  const int q = blockIdx.x * BL_THR + threadIdx.x;
  if (q >= Kk) return;
  const int e = bwd ? a.kept[(size_t)b * Kk + q] : (int)(wsp<unsigned long long>(g, b, g.L.keys)[q] & 0xFFFFFFFFu);
  const int sr = node_of(a.ei[e], a.N), tg = node_of(a.ei[a.ei_stride + e], a.N);
  const float w = a.w[(size_t)b * a.w_bstride + e];
  wsp<int>(g, b, g.L.ksrc)[q] = sr; wsp<int>(g, b, g.L.ktgt)[q] = tg; wsp<float>(g, b, g.L.kw)[q] = w;
  if (!bwd) {
    a.kept[(size_t)b * Kk + q] = e;
    a.ei_out[(size_t)b * 2 * Kk + q] = sr; a.ei_out[(size_t)b * 2 * Kk + Kk + q] = tg;
    a.alpha_out[(size_t)b * Kk + q] = edge_mean(a.beta_save + (size_t)b * a.N * a.T, tg, a.T, w);   // self._alpha = mean(gamma[top], -1)  (:191)
  }
}

// ---- per-node lists (by source: which = 0, by target: which = 1) of kept positions, in pruning order -------------------------
// One workgroup per sample, one thread per node; the key array passes through LDS in chunks that every thread scans.
__global__ __launch_bounds__(BL_LTHR) void k_bl_lists(LArgs g, int which) {
  __shared__ __attribute__((aligned(16))) int sk[BL_LCH];
  __shared__ int cnt[BL_LTHR + 1];
  const BetaArgs& a = g.a;
  const int b = blockIdx.x, N = a.N, Kk = a.Kk, n = threadIdx.x;
  const int* key = wsp<int>(g, b, which ? g.L.ktgt : g.L.ksrc);
  int* off = wsp<int>(g, b, which ? g.L.toff : g.L.soff);
  int* list = wsp<int>(g, b, which ? g.L.tlist : g.L.slist);
  int c = 0;
  for (int c0 = 0; c0 < Kk; c0 += BL_LCH) {
    const int m = min(BL_LCH, Kk - c0);
    for (int i = threadIdx.x; i < BL_LCH; i += BL_LTHR) sk[i] = i < m ? key[c0 + i] : -1;
    __syncthreads();
    if (n < N) {
      const int4* s4 = reinterpret_cast<const int4*>(sk);
      for (int i = 0; i < (m + 3) >> 2; ++i) {
        const int4 v = s4[i];
        c += (v.x == n) + (v.y == n) + (v.z == n) + (v.w == n);
      }
    }
    __syncthreads();
  }
  cnt[n + 1] = n < N ? c : 0;
  if (n == 0) cnt[0] = 0;
  __syncthreads();
  if (n == 0) for (int i = 0; i < N; ++i) cnt[i + 1] += cnt[i];
  __syncthreads();
  if (n <= N) off[n] = cnt[n];
  if (n == 0) off[N] = cnt[N];                           // N == 1024: no thread has n == N
  int w = n < N ? cnt[n] : 0;
  for (int c0 = 0; c0 < Kk; c0 += BL_LCH) {
    const int m = min(BL_LCH, Kk - c0);
    for (int i = threadIdx.x; i < BL_LCH; i += BL_LTHR) sk[i] = i < m ? key[c0 + i] : -1;
    __syncthreads();
    if (n < N)
      for (int i = 0; i < m; ++i)
        if (sk[i] == n) list[w++] = c0 + i;
    __syncthreads();
  }
}

// ---- softmax statistics per (source n, step t) over n's kept edges: mx, inv = 1 / (sum exp(g - mx) + 1e-16) --------------------
__global__ __launch_bounds__(BL_THR) void k_bl_stats(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T;
  const long i = (long)blockIdx.x * BL_THR + threadIdx.x;
  if (i >= (long)N * T) return;
  const int n = (int)(i / T), t = (int)(i - (long)n * T);
  const float* beta = a.beta_save + (size_t)b * N * T;
  const int* soff = wsp<int>(g, b, g.L.soff); const int* slist = wsp<int>(g, b, g.L.slist);
  const int* ktgt = wsp<int>(g, b, g.L.ktgt); const float* kw = wsp<float>(g, b, g.L.kw);
  const int q0 = soff[n], q1 = soff[n + 1];
  float m = -INFINITY;
  for (int q = q0; q < q1; ++q) { const int e = slist[q]; m = fmaxf(m, beta[(size_t)ktgt[e] * T + t] * kw[e]); }
  float z = 0.f;
  for (int q = q0; q < q1; ++q) { const int e = slist[q]; z += expf(beta[(size_t)ktgt[e] * T + t] * kw[e] - m); }
  wsp<float>(g, b, g.L.mx)[i] = m; wsp<float>(g, b, g.L.inv)[i] = 1.0f / (z + 1e-16f);
}

// per-sample views used by the remaining kernels
struct View {
  const float *beta, *mx, *inv, *kw; const int *ksrc, *ktgt;
  int T;
  __device__ __forceinline__ float weight(int e, int t) const {
    const int n = ksrc[e];
    return expf(beta[(size_t)ktgt[e] * T + t] * kw[e] - mx[(size_t)n * T + t]) * inv[(size_t)n * T + t];
  }
};
__device__ __forceinline__ View view_of(const LArgs& g, int b) {
  View v;
  v.beta = g.a.beta_save + (size_t)b * g.a.N * g.a.T; v.mx = wsp<float>(g, b, g.L.mx); v.inv = wsp<float>(g, b, g.L.inv);
  v.kw = wsp<float>(g, b, g.L.kw); v.ksrc = wsp<int>(g, b, g.L.ksrc); v.ktgt = wsp<int>(g, b, g.L.ktgt); v.T = g.a.T;
  return v;
}

// ---- out[n][t*4 + c] = sum over n's kept out-edges of softmax weight[e][t] * V[tgt(e)][t*4 + c]  (d_ob == 4) --------------------
__global__ __launch_bounds__(BL_THR) void k_bl_out(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T, K = a.K;
  const long i = (long)blockIdx.x * BL_THR + threadIdx.x;
  if (i >= (long)N * T) return;
  const int n = (int)(i / T), t = (int)(i - (long)n * T);
  const View v = view_of(g, b);
  const float* V = a.V + (size_t)b * N * K;
  const int* soff = wsp<int>(g, b, g.L.soff); const int* slist = wsp<int>(g, b, g.L.slist);
  const float m = v.mx[i], iv = v.inv[i];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int q = soff[n]; q < soff[n + 1]; ++q) {
    const int e = slist[q], tg = v.ktgt[e];
    const float wg = expf(v.beta[(size_t)tg * T + t] * v.kw[e] - m) * iv;
    const float4 x = *reinterpret_cast<const float4*>(V + (size_t)tg * K + 4 * t);
    acc[0] += wg * x.x; acc[1] += wg * x.y; acc[2] += wg * x.z; acc[3] += wg * x.w;
  }
  *reinterpret_cast<float4*>(a.out + (size_t)b * N * K + (size_t)n * K + 4 * t) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ---- backward ------------------------------------------------------------------------------------------------------------------
// d loss / d weight[e][t] = sum_c dout[src][4t + c] * V[tgt][4t + c]
__device__ __forceinline__ float dwgt_of(const float* dout, const float* V, int K, int sr, int tg, int t) {
  const float4 o = *reinterpret_cast<const float4*>(dout + (size_t)sr * K + 4 * t);
  const float4 x = *reinterpret_cast<const float4*>(V + (size_t)tg * K + 4 * t);
  float s = 0.f;
  s += o.x * x.x; s += o.y * x.y; s += o.z * x.z; s += o.w * x.w;
  return s;
}
// S[n][t] = sum over n's out-edges of weight * dweight
__global__ __launch_bounds__(BL_THR) void k_bl_S(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T, K = a.K;
  const long i = (long)blockIdx.x * BL_THR + threadIdx.x;
  if (i >= (long)N * T) return;
  const int n = (int)(i / T), t = (int)(i - (long)n * T);
  const View v = view_of(g, b);
  const float* V = a.V + (size_t)b * N * K; const float* dout = a.dout + (size_t)b * N * K;
  const int* soff = wsp<int>(g, b, g.L.soff); const int* slist = wsp<int>(g, b, g.L.slist);
  float s = 0.f;
  for (int q = soff[n]; q < soff[n + 1]; ++q) { const int e = slist[q]; s += v.weight(e, t) * dwgt_of(dout, V, K, n, v.ktgt[e], t); }
  wsp<float>(g, b, g.L.S)[i] = s;
}
// dV[i][4t + c] = sum over kept edges INTO i of weight[e][t] * dout[src(e)][4t + c];
// dbeta[i][t] = sum over kept edges into i of w[e] * weight * (dweight - S[src]);  dH[i][t][c] = dbeta / 32 * cat(map_w[i], p_t[t])[c]
__global__ __launch_bounds__(BL_THR) void k_bl_dv_dbeta(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T, K = a.K;
  const long i = (long)blockIdx.x * BL_THR + threadIdx.x;
  if (i >= (long)N * T) return;
  const int n = (int)(i / T), t = (int)(i - (long)n * T);
  const View v = view_of(g, b);
  const float* V = a.V + (size_t)b * N * K; const float* dout = a.dout + (size_t)b * N * K;
  const int* toff = wsp<int>(g, b, g.L.toff); const int* tlist = wsp<int>(g, b, g.L.tlist);
  const float* S = wsp<float>(g, b, g.L.S);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float s = 0.f;
  for (int q = toff[n]; q < toff[n + 1]; ++q) {
    const int e = tlist[q], sr = v.ksrc[e];
    const float wg = v.weight(e, t);
    const float4 o = *reinterpret_cast<const float4*>(dout + (size_t)sr * K + 4 * t);
    acc[0] += wg * o.x; acc[1] += wg * o.y; acc[2] += wg * o.z; acc[3] += wg * o.w;
    s += v.kw[e] * (wg * (dwgt_of(dout, V, K, sr, n, t) - S[(size_t)sr * T + t]));
  }
  *reinterpret_cast<float4*>(a.dV + (size_t)b * N * K + (size_t)n * K + 4 * t) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  const float db = s * (1.0f / 32.0f);
  float* ph = a.dH + ((size_t)b * N * T + i) * 32;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
#pragma unroll
  for (int c = 0; c < 16; ++c) ph[c] = db * a.map_w[n * 16 + c];
#pragma unroll
  for (int c = 0; c < 16; ++c) ph[16 + c] = db * pt[t * 16 + c];
  wsp<float>(g, b, g.L.db)[i] = db;
}
// dmap[i][c < 16] = sum_t dbeta[i][t] / 32 * H[i][t][c], steps in order
__global__ __launch_bounds__(BL_THR) void k_bl_dmap(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T;
  const int i = blockIdx.x * BL_THR + threadIdx.x;
  if (i >= N * 16) return;
  const int n = i >> 4, c = i & 15;
  const float* db = wsp<float>(g, b, g.L.db);
  const float* H = a.H + (size_t)b * N * T * 32;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += db[(size_t)n * T + t] * H[((size_t)n * T + t) * 32 + c];
  a.dmap_part[(size_t)b * N * 16 + i] = s;
}
__global__ __launch_bounds__(BL_THR) void k_bl_dw_zero(LArgs g) {
  const int e = blockIdx.x * BL_THR + threadIdx.x;
  if (e < g.a.E) g.a.dw[(size_t)blockIdx.y * g.a.E + e] = 0.f;
}
// d loss / d w[e] = sum_t weight (dweight - S[src]) beta[tgt][t] for kept edges (0 for pruned ones: k_bl_dw_zero)
__global__ __launch_bounds__(BL_THR) void k_bl_dw(LArgs g) {
  const BetaArgs& a = g.a;
  const int b = blockIdx.y, N = a.N, T = a.T, K = a.K, Kk = a.Kk;
  const int q = blockIdx.x * BL_THR + threadIdx.x;
  if (q >= Kk) return;
  const View v = view_of(g, b);
  const float* V = a.V + (size_t)b * N * K; const float* dout = a.dout + (size_t)b * N * K;
  const float* S = wsp<float>(g, b, g.L.S);
  const int sr = v.ksrc[q], tg = v.ktgt[q];
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += v.weight(q, t) * (dwgt_of(dout, V, K, sr, tg, t) - S[(size_t)sr * T + t]) * v.beta[(size_t)tg * T + t];
  a.dw[(size_t)b * a.E + a.kept[(size_t)b * Kk + q]] = s;
}

int check_large(const BetaArgs& a, void* ws, size_t ws_bytes) {
  if (a.N > BL_LTHR) return fail(RD_EUNSUPPORTED, "rd_graph_beta: N = %d nodes (the workspace form builds its per-node lists with one thread per node: N <= %d)", a.N, BL_LTHR);
  if (a.B > 65535) return fail(RD_EINVAL, "rd_graph_beta: B = %d exceeds the grid's y extent", a.B);
  if ((long)a.N * a.T > (1L << 31) || a.E > (1 << 28)) return fail(RD_EINVAL, "rd_graph_beta: graph too large (N=%d, T=%d, E=%d)", a.N, a.T, a.E);
  const size_t need = beta_large_ws_bytes(a.B, a.N, a.T, a.E);
  if (!ws || ws_bytes < need)
    return fail(RD_EINVAL, "rd_graph_beta: this graph (N=%d, T=%d, E=%d) does not fit one workgroup's LDS and needs a workspace of %zu bytes "
                "(rd_graph_beta_workspace_bytes); got %zu", a.N, a.T, a.E, need, ws_bytes);
  if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(RD_EINVAL, "rd_graph_beta: workspace must be 256-byte aligned");
  return RD_OK;
}

}  // namespace

size_t beta_large_ws_bytes(int B, int N, int T, int E) {
  const int Kk = (int)((double)E * 0.5);
  return (size_t)B * ws_layout(N, T, beta_next_pow2(E > 1 ? E : 2), Kk).stride;
}

#define BL_LAUNCH(kernel, gx, thr, ...)                                                                    \
  do {                                                                                                     \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(gx), (unsigned)a.B), dim3(thr), 0, st, __VA_ARGS__);        \
    const int rc_ = check_launch(#kernel);                                                                 \
    if (rc_) return rc_;                                                                                   \
  } while (0)

int beta_large_fwd(const BetaArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
  int rc = check_large(a, ws, ws_bytes);
  if (rc) return rc;
  LArgs g{};
  g.a = a; g.ws = (unsigned char*)ws; g.P2 = beta_next_pow2(a.E > 1 ? a.E : 2); g.L = ws_layout(a.N, a.T, g.P2, a.Kk);
  const long nt = (long)a.N * a.T;
  BL_LAUNCH(k_bl_beta, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  BL_LAUNCH(k_bl_keys, cdiv(g.P2, BL_THR), BL_THR, g);
  const int chunk = g.P2 < BL_CH ? g.P2 : BL_CH;
  BL_LAUNCH(k_bl_sort_local, g.P2 / chunk, 1024, g, chunk);
  for (int k = 2 * BL_CH; k <= g.P2; k <<= 1) {
    for (int j = k >> 1; j >= BL_CH; j >>= 1) BL_LAUNCH(k_bl_merge_global, cdiv(g.P2, BL_THR), BL_THR, g, k, j);
    BL_LAUNCH(k_bl_merge_local, g.P2 / BL_CH, 1024, g, k);
  }
  if (a.Kk > 0) BL_LAUNCH(k_bl_kept, cdiv(a.Kk, BL_THR), BL_THR, g, 0);
  hipLaunchKernelGGL(k_bl_lists, dim3(a.B), dim3(BL_LTHR), 0, st, g, 0);
  if ((rc = check_launch("k_bl_lists"))) return rc;
  BL_LAUNCH(k_bl_stats, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  BL_LAUNCH(k_bl_out, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  return RD_OK;
}

int beta_large_bwd(const BetaArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
  int rc = check_large(a, ws, ws_bytes);
  if (rc) return rc;
  LArgs g{};
  g.a = a; g.ws = (unsigned char*)ws; g.P2 = beta_next_pow2(a.E > 1 ? a.E : 2); g.L = ws_layout(a.N, a.T, g.P2, a.Kk);
  const long nt = (long)a.N * a.T;
  if (a.Kk > 0) BL_LAUNCH(k_bl_kept, cdiv(a.Kk, BL_THR), BL_THR, g, 1);
  hipLaunchKernelGGL(k_bl_lists, dim3(a.B), dim3(BL_LTHR), 0, st, g, 0);
  if ((rc = check_launch("k_bl_lists"))) return rc;
  hipLaunchKernelGGL(k_bl_lists, dim3(a.B), dim3(BL_LTHR), 0, st, g, 1);
  if ((rc = check_launch("k_bl_lists"))) return rc;
  BL_LAUNCH(k_bl_stats, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  BL_LAUNCH(k_bl_S, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  BL_LAUNCH(k_bl_dv_dbeta, (nt + BL_THR - 1) / BL_THR, BL_THR, g);
  BL_LAUNCH(k_bl_dmap, cdiv(a.N * 16, BL_THR), BL_THR, g);
  if (a.dw) {
    if (a.E > 0) BL_LAUNCH(k_bl_dw_zero, cdiv(a.E, BL_THR), BL_THR, g);
    if (a.Kk > 0) BL_LAUNCH(k_bl_dw, cdiv(a.Kk, BL_THR), BL_THR, g);
  }
  return RD_OK;
}

}  // namespace rd
