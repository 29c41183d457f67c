// rd_graph_beta.hip -- the paper-faithful graph operator of Observation_progation as ONE batched
// kernel, one workgroup per sample graph with the graph staged in LDS:
//   code/Ob_propagation.py:161-185  per-edge, per-time-step scores  gamma[e, t] = beta[tgt(e), t] * w[e]
//                                   beta[i, t] = mean_c( increase_dim(x_i)[t, c] * cat(map_weights[i], p_t[t])[c] )   (32 channels)
//   :179-185                        prune to the top half of the edges by mean score (descending), REORDER them that way
//   :184,195                        softmax of the kept scores per channel, normalised over the edges of one SOURCE node
//   :200,208,227                    out[n] = sum_{kept e: src(e) = n} softmax(gamma)[e] (.) relu(lin_value(x_tgt(e)))
// Every per-edge quantity of the reference depends on the edge only through (target, weight), so lin_value and
// increase_dim run once per NODE (rd_linear_fwd) and this kernel does the genuinely graph-shaped part: scores, an LDS
// bitonic sort for the pruning, per-source per-time-step softmax, neighbour aggregation; backward likewise.
// Also here: the structure-distance regulariser of code/models_rd.py:345-346, mean pairwise L2 distance between the
// samples' edge-score vectors (identically 0 on the shipped path, non-trivial as soon as the scores differ per sample).
// Ties in the pruning sort are broken by edge order (lower edge id first).
#include <stdlib.h>

#include "rd_common.h"
#include "rd_graph_beta.h"

namespace rd {
namespace {

constexpr int GB_THR = 256;
constexpr int GB_MAXE = 4096;            // edges per graph held in LDS (N <= 64 nodes)
constexpr int GB_MAXN = 64;

// shared LDS layout helpers
struct Lds {
  float* beta;              // [N][T]
  unsigned long long* keys; // [P2]
  int* ksrc; int* ktgt; float* kw;   // kept edges, pruning order [Kk]
  int* soff; int* slist;    // per-source lists of kept positions: soff [N+1], slist [Kk]
  int* toff; int* tlist;    // per-target lists
  float* mx; float* inv;    // [N][T] softmax max / 1/(Z + 1e-16) per source and time step
  float* S; float* db;      // backward: [N][Tc] sum_e weight * dweight per source; dbeta / 32 per target
  float* dmacc; float* dwacc;   // backward: sums over the time chunks: d map_weights [N][16], d edge weight [Kk]
};

// Forward (bwd == false): beta, mx, inv for ALL T steps (the pruning sort needs every step's score) + the sort keys.
// Backward: no sort; the five [N][Tc] arrays hold one CHUNK of Tc time steps at a time (everything but the two sums over time --
// d map_weights and d edge weight, accumulated in dmacc / dwacc -- is independent per step), so T is not bounded by LDS.
__device__ Lds carve(unsigned char* base, int N, int T, int P2, int Kk, bool bwd) {
  Lds l; size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base + off; off += (bytes + 15) & ~(size_t)15; return p; };
  l.beta = (float*)take((size_t)N * T * 4);
  l.keys = bwd ? nullptr : (unsigned long long*)take((size_t)P2 * 8);
  l.ksrc = (int*)take((size_t)Kk * 4); l.ktgt = (int*)take((size_t)Kk * 4); l.kw = (float*)take((size_t)Kk * 4);
  l.soff = (int*)take((size_t)(N + 1) * 4); l.slist = (int*)take((size_t)Kk * 4);
  l.toff = (int*)take((size_t)(N + 1) * 4); l.tlist = (int*)take((size_t)Kk * 4);
  l.mx = (float*)take((size_t)N * T * 4); l.inv = (float*)take((size_t)N * T * 4);
  if (bwd) {
    l.S = (float*)take((size_t)N * T * 4); l.db = (float*)take((size_t)N * T * 4);
    l.dmacc = (float*)take((size_t)N * 16 * 4); l.dwacc = (float*)take((size_t)Kk * 4);
  } else { l.S = nullptr; l.db = nullptr; l.dmacc = nullptr; l.dwacc = nullptr; }
  return l;
}
// T here is the number of time steps RESIDENT in LDS (all of them forward, one chunk backward)
size_t lds_bytes(int N, int T, int P2, int Kk, bool bwd) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  const size_t lists = r((size_t)Kk * 4) * 5 + r((size_t)(N + 1) * 4) * 2;
  if (!bwd) return r((size_t)N * T * 4) * 3 + r((size_t)P2 * 8) + lists;
  return r((size_t)N * T * 4) * 5 + lists + r((size_t)N * 16 * 4) + r((size_t)Kk * 4);
}
// largest chunk of time steps the backward kernel can keep resident (>= 1 whenever the lists fit)
int bwd_chunk(int N, int T, int Kk) {
  int tc = T;
  while (tc > 1 && lds_bytes(N, tc, 0, Kk, true) > 160 * 1024) tc = (tc + 1) / 2;
  return tc;
}

// per-node lists (by source or by target) of kept positions, in pruning order: thread n scans the kept edges
__device__ void build_lists(const int* key, int Kk, int N, int* off, int* list, int tid) {
  // counts
  for (int n = tid; n < N; n += GB_THR) {
    int c = 0;
    for (int e = 0; e < Kk; ++e) c += key[e] == n;
    off[n + 1] = c;
  }
  __syncthreads();
  if (tid == 0) { off[0] = 0; for (int n = 0; n < N; ++n) off[n + 1] += off[n]; }
  __syncthreads();
  for (int n = tid; n < N; n += GB_THR) {
    int w = off[n];
    for (int e = 0; e < Kk; ++e) if (key[e] == n) list[w++] = e;
  }
  __syncthreads();
}

// softmax statistics per (source n, time step t) over n's kept edges: mx, inv = 1 / (sum exp(g - mx) + 1e-16)
// (tc resident steps, rows LT apart)
__device__ void softmax_stats(const Lds& l, int N, int tc, int LT, int tid) {
  for (int i = tid; i < N * tc; i += GB_THR) {
    const int n = i / tc, t = i - n * tc;
    float m = -INFINITY;
    for (int q = l.soff[n]; q < l.soff[n + 1]; ++q) { const int e = l.slist[q]; m = fmaxf(m, l.beta[l.ktgt[e] * LT + t] * l.kw[e]); }
    float z = 0.f;
    for (int q = l.soff[n]; q < l.soff[n + 1]; ++q) { const int e = l.slist[q]; z += expf(l.beta[l.ktgt[e] * LT + t] * l.kw[e] - m); }
    l.mx[n * LT + t] = m; l.inv[n * LT + t] = 1.0f / (z + 1e-16f);
  }
  __syncthreads();
}

__global__ __launch_bounds__(GB_THR) void k_graph_beta_fwd(BetaArgs a, int P2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = a.N, T = a.T, K = a.K, d = a.d, E = a.E, Kk = a.Kk;
  Lds l = carve(gsm, N, T, P2, Kk, false);
  const float* H = a.H + (size_t)b * N * T * 32;
  const float* V = a.V + (size_t)b * N * K;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
  const float* w = a.w + (size_t)b * a.w_bstride;
  // ---- beta[i][t] = mean over 32 channels of increase_dim(x_i)[t] * cat(map_weights[i], p_t[t]) ----
  for (int i = tid; i < N * T; i += GB_THR) {
    const int n = i / T, t = i - n * T;
    const float* h = H + ((size_t)n * T + t) * 32;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += h[c] * a.map_w[n * 16 + c];
#pragma unroll
    for (int c = 0; c < 16; ++c) s += h[16 + c] * pt[t * 16 + c];
    const float bt = s * (1.0f / 32.0f);
    l.beta[i] = bt;
    a.beta_save[(size_t)b * N * T + i] = bt;
  }
  __syncthreads();
  // ---- mean score of every edge, descending bitonic sort (ties: lower edge id first) ----
  for (int e = tid; e < P2; e += GB_THR) {
    unsigned long long key = ~0ull;                                   // padding sorts last
    if (e < E) {
      const int tg = node_of(a.ei[a.ei_stride + e], N);
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += l.beta[tg * T + t] * w[e];
      s = s / (float)T;                                               // == mean over the K = T*d repeated channels
      key = ((unsigned long long)sortable_desc(s) << 32) | (unsigned)e;
    }
    l.keys[e] = key;
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += GB_THR) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = l.keys[i], y = l.keys[p];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { l.keys[i] = y; l.keys[p] = x; }
        }
      }
      __syncthreads();
    }
  // ---- kept edges in pruning order ----
  for (int q = tid; q < Kk; q += GB_THR) {
    const int e = (int)(l.keys[q] & 0xFFFFFFFFu);
    const int sr = node_of(a.ei[e], N), tg = node_of(a.ei[a.ei_stride + e], N);
    l.ksrc[q] = sr; l.ktgt[q] = tg; l.kw[q] = w[e];
    a.kept[(size_t)b * Kk + q] = e;
    a.ei_out[(size_t)b * 2 * Kk + q] = sr; a.ei_out[(size_t)b * 2 * Kk + Kk + q] = tg;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += l.beta[tg * T + t] * w[e];
    a.alpha_out[(size_t)b * Kk + q] = s / (float)T;                  // self._alpha = mean(gamma[top], -1)  (:191)
  }
  __syncthreads();
  build_lists(l.ksrc, Kk, N, l.soff, l.slist, tid);
  softmax_stats(l, N, T, T, tid);
  // ---- out[n][k] = sum over n's kept out-edges of softmax weight[e][t(k)] * V[tgt(e)][k] ----
  float* out = a.out + (size_t)b * N * K;
  for (int i = tid; i < N * K; i += GB_THR) {
    const int n = i / K, k = i - n * K, t = k / d;
    const float m = l.mx[n * T + t], iv = l.inv[n * T + t];
    float acc = 0.f;
    for (int q = l.soff[n]; q < l.soff[n + 1]; ++q) {
      const int e = l.slist[q], tg = l.ktgt[e];
      acc += expf(l.beta[tg * T + t] * l.kw[e] - m) * iv * V[(size_t)tg * K + k];
    }
    out[i] = acc;
  }
}

__global__ __launch_bounds__(GB_THR) void k_graph_beta_bwd(BetaArgs a, int Tc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = a.N, T = a.T, K = a.K, d = a.d, Kk = a.Kk;
  Lds l = carve(gsm, N, Tc, 0, Kk, true);
  float* S = l.S;
  const float* H = a.H + (size_t)b * N * T * 32;
  const float* V = a.V + (size_t)b * N * K;
  const float* dout = a.dout + (size_t)b * N * K;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
  const float* w = a.w + (size_t)b * a.w_bstride;
  for (int q = tid; q < Kk; q += GB_THR) {
    const int e = a.kept[(size_t)b * Kk + q];
    l.ksrc[q] = node_of(a.ei[e], N); l.ktgt[q] = node_of(a.ei[a.ei_stride + e], N); l.kw[q] = w[e];
    l.dwacc[q] = 0.f;
  }
  for (int i = tid; i < N * 16; i += GB_THR) l.dmacc[i] = 0.f;
  __syncthreads();
  build_lists(l.ksrc, Kk, N, l.soff, l.slist, tid);
  build_lists(l.ktgt, Kk, N, l.toff, l.tlist, tid);
  float* dV = a.dV + (size_t)b * N * K;
  float* dH = a.dH + (size_t)b * N * T * 32;
  // time steps [t0, t0 + tc) resident per pass; local step tt = t - t0, rows of the [N][Tc] arrays Tc apart
  for (int t0 = 0; t0 < T; t0 += Tc) {
    const int tc = min(Tc, T - t0);
    for (int i = tid; i < N * tc; i += GB_THR) {
      const int n = i / tc, tt = i - n * tc;
      l.beta[n * Tc + tt] = a.beta_save[((size_t)b * N + n) * T + t0 + tt];
    }
    __syncthreads();
    softmax_stats(l, N, tc, Tc, tid);
    auto weight = [&](int e, int tt) {
      const int n = l.ksrc[e];
      return expf(l.beta[l.ktgt[e] * Tc + tt] * l.kw[e] - l.mx[n * Tc + tt]) * l.inv[n * Tc + tt];
    };
    auto dwgt = [&](int e, int tt) {                                 // d loss / d weight[e][t] = sum_c dout[src][td+c] * V[tgt][td+c]
      const float* po = dout + (size_t)l.ksrc[e] * K + (t0 + tt) * d;
      const float* pv = V + (size_t)l.ktgt[e] * K + (t0 + tt) * d;
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += po[c] * pv[c];
      return s;
    };
    // S[n][t] = sum over n's out-edges of weight * dweight
    for (int i = tid; i < N * tc; i += GB_THR) {
      const int n = i / tc, tt = i - n * tc;
      float s = 0.f;
      for (int q = l.soff[n]; q < l.soff[n + 1]; ++q) { const int e = l.slist[q]; s += weight(e, tt) * dwgt(e, tt); }
      S[n * Tc + tt] = s;
    }
    __syncthreads();
    // dV[i][k] = sum over kept edges INTO i of weight[e][t(k)] * dout[src(e)][k]      (columns of this chunk)
    const int kc = tc * d;
    for (int i = tid; i < N * kc; i += GB_THR) {
      const int n = i / kc, kk = i - n * kc, k = t0 * d + kk, tt = kk / d;
      float acc = 0.f;
      for (int q = l.toff[n]; q < l.toff[n + 1]; ++q) { const int e = l.tlist[q]; acc += weight(e, tt) * dout[(size_t)l.ksrc[e] * K + k]; }
      dV[(size_t)n * K + k] = acc;
    }
    // dbeta[i][t] = sum over kept edges into i of w[e] * dg[e][t],  dg = weight * (dweight - S[src])
    // dH[i][t][c] = dbeta * aa[c] / 32;  dmap[i][c<16] = sum_t dbeta * H[i][t][c] / 32
    for (int i = tid; i < N * tc; i += GB_THR) {
      const int n = i / tc, tt = i - n * tc, t = t0 + tt;
      float s = 0.f;
      for (int q = l.toff[n]; q < l.toff[n + 1]; ++q) {
        const int e = l.tlist[q];
        s += l.kw[e] * (weight(e, tt) * (dwgt(e, tt) - S[l.ksrc[e] * Tc + tt]));
      }
      const float db = s * (1.0f / 32.0f);
      float* ph = dH + ((size_t)n * T + t) * 32;
#pragma unroll
      for (int c = 0; c < 16; ++c) ph[c] = db * a.map_w[n * 16 + c];
#pragma unroll
      for (int c = 0; c < 16; ++c) ph[16 + c] = db * pt[t * 16 + c];
      l.db[n * Tc + tt] = db;
    }
    __syncthreads();
    for (int i = tid; i < N * 16; i += GB_THR) {                     // same thread for a given (n, c) in every pass: ordered sum over t
      const int n = i >> 4, c = i & 15;
      float s = l.dmacc[i];
      for (int tt = 0; tt < tc; ++tt) s += l.db[n * Tc + tt] * H[((size_t)n * T + t0 + tt) * 32 + c];
      l.dmacc[i] = s;
    }
    // d loss / d w[e] = sum_t dg[e][t] * beta[tgt][t] for kept edges
    if (a.dw)
      for (int q = tid; q < Kk; q += GB_THR) {
        float s = l.dwacc[q];
        for (int tt = 0; tt < tc; ++tt) s += weight(q, tt) * (dwgt(q, tt) - S[l.ksrc[q] * Tc + tt]) * l.beta[l.ktgt[q] * Tc + tt];
        l.dwacc[q] = s;
      }
    __syncthreads();
  }
  for (int i = tid; i < N * 16; i += GB_THR) a.dmap_part[(size_t)b * N * 16 + i] = l.dmacc[i];
  if (a.dw) {                                                        // 0 for pruned edges
    float* dw = a.dw + (size_t)b * a.E;
    for (int e = tid; e < a.E; e += GB_THR) dw[e] = 0.f;
    __syncthreads();
    for (int q = tid; q < Kk; q += GB_THR) dw[a.kept[(size_t)b * Kk + q]] = l.dwacc[q];
  }
}

// ================================================================================================================================
// Round 6: the same operator on 16 waves per sample (rounds 2-5: 4) with every (edge, step) quantity formed ONCE.
// Why: the captured use_beta step (bench.py --use-beta) spent 41 % of its 2.19 ms in the two kernels above -- 247 + 641 us for
// 84 + 168 MB of operands, 3-4 % of the HBM roofline.  They are latency-bound scalar loops on one wave per SIMD: per-node edge lists
// built by 34 threads scanning 578 edges each (four times in the backward), exp() of the same (edge, step) score recomputed by each
// of the d channels of the aggregation and by four of the backward's passes, 4-byte loads of V / dout / H.  Here:
//   * 1024 threads; H, V, dout, dV, dH move as 16-byte accesses; V is staged in LDS forward where it fits;
//   * per-node lists by a wave per node (ballot + prefix popcount over the kept edges: the same order, ascending pruning position);
//   * forward: max, normaliser and the aggregation of a (source, step) pair by ONE thread for the d = 4 channels of the step;
//   * backward: per chunk of steps, weight[e][t] and dweight[e][t] are computed once into LDS ([kept edges][chunk]); S, dV, dbeta,
//     d map_weights and d edge weight read them.
// Every sum keeps the order of the kernels above (lists in pruning order, steps ascending, channels ascending), so the results are
// the same BITS (tests/test_graph_beta_gpu.py::test_v2_kernels_equal_v1_bit_for_bit; RD_BETA_V1=1 runs the old kernels).
constexpr int GB2_THR = 1024, GB2_NW = GB2_THR / 64;

// per-node lists of kept positions (ascending), one WAVE per node: off [N+1], list [Kk]
__device__ void build_lists2(const int* key, int Kk, int N, int* off, int* list, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int n = wave; n < N; n += GB2_NW) {
    int c = 0;
    for (int base = 0; base < Kk; base += 64) {
      const int q = base + lane;
      c += __popcll(__ballot(q < Kk && key[q] == n));
    }
    if (lane == 0) off[n + 1] = c;
  }
  __syncthreads();
  if (tid == 0) { off[0] = 0; for (int n = 0; n < N; ++n) off[n + 1] += off[n]; }
  __syncthreads();
  for (int n = wave; n < N; n += GB2_NW) {
    int w = off[n];
    for (int base = 0; base < Kk; base += 64) {
      const int q = base + lane;
      const bool hit = q < Kk && key[q] == n;
      const unsigned long long mk = __ballot(hit);
      if (hit) list[w + __popcll(mk & ((1ull << lane) - 1ull))] = q;
      w += __popcll(mk);
    }
  }
  __syncthreads();
}

struct Lds2 {
  float* beta; unsigned long long* keys; int *ksrc, *ktgt; float* kw; int *soff, *slist, *toff, *tlist;
  float *V;                                  // forward: [N][K] staged values (or null)
  float *mx, *inv, *S, *db, *dmacc, *dwacc, *W, *DG;   // backward
};
size_t lds2_fwd(int N, int T, int K, int P2, int Kk, bool stage_v) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return r((size_t)N * T * 4) + r((size_t)P2 * 8) + 6 * r((size_t)Kk * 4) + r((size_t)(N + 1) * 4) + (stage_v ? r((size_t)N * K * 4) : 0);
}
size_t lds2_bwd(int N, int Tc, int Kk) {
  auto r = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return 5 * r((size_t)N * Tc * 4) + 7 * r((size_t)Kk * 4) + 2 * r((size_t)(N + 1) * 4) + r((size_t)N * 16 * 4) + r((size_t)Kk * 4) +
         2 * r((size_t)Kk * Tc * 4);
}
int bwd2_chunk(int N, int T, int Kk) {       // steps per pass: the largest that fits, then evened out over the passes
  int tc = T;
  while (tc > 1 && lds2_bwd(N, tc, Kk) > 160 * 1024) --tc;
  const int np = (T + tc - 1) / tc;
  return (T + np - 1) / np;
}

template <bool STAGE_V>
__global__ __launch_bounds__(GB2_THR) void k_graph_beta_fwd2(BetaArgs a, int P2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = a.N, T = a.T, K = a.K, E = a.E, Kk = a.Kk;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = gsm + off; off += (bytes + 15) & ~(size_t)15; return p; };
  float* beta = (float*)take((size_t)N * T * 4);
  unsigned long long* keys = (unsigned long long*)take((size_t)P2 * 8);
  int* ksrc = (int*)take((size_t)Kk * 4); int* ktgt = (int*)take((size_t)Kk * 4); float* kw = (float*)take((size_t)Kk * 4);
  int* slist = (int*)take((size_t)Kk * 4); int* soff = (int*)take((size_t)(N + 1) * 4);
  int* ltg = (int*)take((size_t)Kk * 4); float* lkw = (float*)take((size_t)Kk * 4);      // target / weight of the kept edges in SOURCE-LIST order
  float* Vs = STAGE_V ? (float*)take((size_t)N * K * 4) : nullptr;
  const float* H = a.H + (size_t)b * N * T * 32;
  const float* V = a.V + (size_t)b * N * K;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
  const float* w = a.w + (size_t)b * a.w_bstride;
  if (STAGE_V)
    for (int i = tid; i < N * K / 4; i += GB2_THR) reinterpret_cast<float4*>(Vs)[i] = reinterpret_cast<const float4*>(V)[i];
  // ---- beta[i][t]: 32 products in the order of the first kernel (map_weights' 16 channels, then p_t's) ----
  for (int i = tid; i < N * T; i += GB2_THR) {
    const int n = i / T, t = i - n * T;
    float h[32], mw[16], pp[16];
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) *reinterpret_cast<float4*>(h + 4 * c4) = *reinterpret_cast<const float4*>(H + (size_t)i * 32 + 4 * c4);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      *reinterpret_cast<float4*>(mw + 4 * c4) = *reinterpret_cast<const float4*>(a.map_w + n * 16 + 4 * c4);
      *reinterpret_cast<float4*>(pp + 4 * c4) = *reinterpret_cast<const float4*>(pt + t * 16 + 4 * c4);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += h[c] * mw[c];
#pragma unroll
    for (int c = 0; c < 16; ++c) s += h[16 + c] * pp[c];
    const float bt = s * (1.0f / 32.0f);
    beta[i] = bt;
    a.beta_save[(size_t)b * N * T + i] = bt;
  }
  __syncthreads();
  for (int e = tid; e < P2; e += GB2_THR) {
    unsigned long long key = ~0ull;
    if (e < E) {
      const int tg = node_of(a.ei[a.ei_stride + e], N);
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += beta[tg * T + t] * w[e];
      s = s / (float)T;
      key = ((unsigned long long)sortable_desc(s) << 32) | (unsigned)e;
    }
    keys[e] = key;
  }
  __syncthreads();
  // bitonic network as above.  Element i is handled by thread i % 1024, i.e. a 64-element block of keys by ONE wave: a step with
  // partner distance j < 64 stays inside the wave (its LDS operations are processed in issue order: no workgroup barrier), only
  // steps with j >= 64 -- and the last step of a level whose successor starts at distance >= 64 -- synchronise the workgroup:
  // 21 barriers instead of 66 for 2048 keys.
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P2; i += GB2_THR) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long x = keys[i], y = keys[p];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { keys[i] = y; keys[p] = x; }
        }
      }
      if (j >= 64 || (j == 1 && k >= 64)) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-local: stores of this step before the next step's loads
    }
  __syncthreads();
  for (int q = tid; q < Kk; q += GB2_THR) {
    const int e = (int)(keys[q] & 0xFFFFFFFFu);
    const int sr = node_of(a.ei[e], N), tg = node_of(a.ei[a.ei_stride + e], N);
    ksrc[q] = sr; ktgt[q] = tg; kw[q] = w[e];
    a.kept[(size_t)b * Kk + q] = e;
    a.ei_out[(size_t)b * 2 * Kk + q] = sr; a.ei_out[(size_t)b * 2 * Kk + Kk + q] = tg;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += beta[tg * T + t] * w[e];
    a.alpha_out[(size_t)b * Kk + q] = s / (float)T;
  }
  __syncthreads();
  build_lists2(ksrc, Kk, N, soff, slist, tid);
  for (int q = tid; q < Kk; q += GB2_THR) { const int e = slist[q]; ltg[q] = ktgt[e]; lkw[q] = kw[e]; }
  __syncthreads();
  // ---- (source n, step t): max, normaliser, then the d = 4 channels of the step ----
  // (consecutive threads = consecutive steps of one source: the list entries are broadcast reads, the scores consecutive floats)
  float* out = a.out + (size_t)b * N * K;
  const float* Vr = STAGE_V ? Vs : V;
  for (int i = tid; i < N * T; i += GB2_THR) {
    const int n = i / T, t = i - n * T;
    const int q0 = soff[n], q1 = soff[n + 1];
    float m = -INFINITY;
    for (int q = q0; q < q1; ++q) m = fmaxf(m, beta[ltg[q] * T + t] * lkw[q]);
    float z = 0.f;
    for (int q = q0; q < q1; ++q) z += expf(beta[ltg[q] * T + t] * lkw[q] - m);
    const float iv = 1.0f / (z + 1e-16f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = q0; q < q1; ++q) {
      const int tg = ltg[q];
      const float wi = expf(beta[tg * T + t] * lkw[q] - m) * iv;
      const float4 v = *reinterpret_cast<const float4*>(Vr + (size_t)tg * K + 4 * t);
      acc.x += wi * v.x; acc.y += wi * v.y; acc.z += wi * v.z; acc.w += wi * v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)n * K + 4 * t) = acc;
  }
}

__global__ __launch_bounds__(GB2_THR) void k_graph_beta_bwd2(BetaArgs a, int Tc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int N = a.N, T = a.T, K = a.K, Kk = a.Kk;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = gsm + off; off += (bytes + 15) & ~(size_t)15; return p; };
  float* beta = (float*)take((size_t)N * Tc * 4); float* mx = (float*)take((size_t)N * Tc * 4); float* inv = (float*)take((size_t)N * Tc * 4);
  float* S = (float*)take((size_t)N * Tc * 4); float* dbv = (float*)take((size_t)N * Tc * 4);
  int* ksrc = (int*)take((size_t)Kk * 4); int* ktgt = (int*)take((size_t)Kk * 4); float* kw = (float*)take((size_t)Kk * 4);
  int* slist = (int*)take((size_t)Kk * 4); int* tlist = (int*)take((size_t)Kk * 4);
  int* soff = (int*)take((size_t)(N + 1) * 4); int* toff = (int*)take((size_t)(N + 1) * 4);
  float* dmacc = (float*)take((size_t)N * 16 * 4); float* dwacc = (float*)take((size_t)Kk * 4);
  float* W = (float*)take((size_t)Kk * Tc * 4); float* DG = (float*)take((size_t)Kk * Tc * 4);
  int* ltg = (int*)take((size_t)Kk * 4); float* lkw = (float*)take((size_t)Kk * 4);      // source-list order (as in the forward)
  const float* H = a.H + (size_t)b * N * T * 32;
  const float* V = a.V + (size_t)b * N * K;
  const float* dout = a.dout + (size_t)b * N * K;
  const float* pt = a.p_t + (size_t)b * a.pt_bstride;
  const float* w = a.w + (size_t)b * a.w_bstride;
  for (int q = tid; q < Kk; q += GB2_THR) {
    const int e = a.kept[(size_t)b * Kk + q];
    ksrc[q] = node_of(a.ei[e], N); ktgt[q] = node_of(a.ei[a.ei_stride + e], N); kw[q] = w[e];
    dwacc[q] = 0.f;
  }
  for (int i = tid; i < N * 16; i += GB2_THR) dmacc[i] = 0.f;
  __syncthreads();
  build_lists2(ksrc, Kk, N, soff, slist, tid);
  build_lists2(ktgt, Kk, N, toff, tlist, tid);
  for (int q = tid; q < Kk; q += GB2_THR) { const int e = slist[q]; ltg[q] = ktgt[e]; lkw[q] = kw[e]; }
  __syncthreads();
  float* dV = a.dV + (size_t)b * N * K;
  float* dH = a.dH + (size_t)b * N * T * 32;
  for (int t0 = 0; t0 < T; t0 += Tc) {
    const int tc = min(Tc, T - t0);
    for (int i = tid; i < N * tc; i += GB2_THR) {
      const int n = i / tc, tt = i - n * tc;
      beta[n * Tc + tt] = a.beta_save[((size_t)b * N + n) * T + t0 + tt];
    }
    __syncthreads();
    for (int i = tid; i < N * tc; i += GB2_THR) {                    // softmax statistics per (source, step): the first kernels' order
      const int n = i / tc, tt = i - n * tc;
      float m = -INFINITY;
      for (int q = soff[n]; q < soff[n + 1]; ++q) m = fmaxf(m, beta[ltg[q] * Tc + tt] * lkw[q]);
      float z = 0.f;
      for (int q = soff[n]; q < soff[n + 1]; ++q) z += expf(beta[ltg[q] * Tc + tt] * lkw[q] - m);
      mx[n * Tc + tt] = m; inv[n * Tc + tt] = 1.0f / (z + 1e-16f);
    }
    __syncthreads();
    // weight[e][t] and d loss / d weight[e][t] = sum_c dout[src][4t+c] V[tgt][4t+c], once
    for (int i = tid; i < Kk * tc; i += GB2_THR) {
      const int e = i / tc, tt = i - e * tc;
      const int n = ksrc[e], tg = ktgt[e];
      const float wi = expf(beta[tg * Tc + tt] * kw[e] - mx[n * Tc + tt]) * inv[n * Tc + tt];
      const float4 po = *reinterpret_cast<const float4*>(dout + (size_t)n * K + 4 * (t0 + tt));
      const float4 pv = *reinterpret_cast<const float4*>(V + (size_t)tg * K + 4 * (t0 + tt));
      float s = 0.f;
      s += po.x * pv.x; s += po.y * pv.y; s += po.z * pv.z; s += po.w * pv.w;
      W[e * Tc + tt] = wi; DG[e * Tc + tt] = s;
    }
    __syncthreads();
    for (int i = tid; i < N * tc; i += GB2_THR) {                    // S[n][t] = sum over n's out-edges of weight * dweight
      const int n = i / tc, tt = i - n * tc;
      float s = 0.f;
      for (int q = soff[n]; q < soff[n + 1]; ++q) { const int e = slist[q]; s += W[e * Tc + tt] * DG[e * Tc + tt]; }
      S[n * Tc + tt] = s;
    }
    __syncthreads();
    for (int i = tid; i < Kk * tc; i += GB2_THR) {                   // dg[e][t] = weight * (dweight - S[src])
      const int e = i / tc, tt = i - e * tc;
      DG[e * Tc + tt] = W[e * Tc + tt] * (DG[e * Tc + tt] - S[ksrc[e] * Tc + tt]);
    }
    __syncthreads();
    for (int i = tid; i < N * tc; i += GB2_THR) {
      const int n = i / tc, tt = i - n * tc, t = t0 + tt;
      // dV[n][4t..] = sum over kept edges INTO n of weight * dout[src][4t..];  dbeta = sum of w[e] * dg[e][t]
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float s = 0.f;
      for (int q = toff[n]; q < toff[n + 1]; ++q) {
        const int e = tlist[q];
        const float wi = W[e * Tc + tt];
        const float4 po = *reinterpret_cast<const float4*>(dout + (size_t)ksrc[e] * K + 4 * t);
        acc.x += wi * po.x; acc.y += wi * po.y; acc.z += wi * po.z; acc.w += wi * po.w;
        s += kw[e] * DG[e * Tc + tt];
      }
      *reinterpret_cast<float4*>(dV + (size_t)n * K + 4 * t) = acc;
      const float db = s * (1.0f / 32.0f);
      float* ph = dH + ((size_t)n * T + t) * 32;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float4 mw = *reinterpret_cast<const float4*>(a.map_w + n * 16 + 4 * c4);
        *reinterpret_cast<float4*>(ph + 4 * c4) = make_float4(db * mw.x, db * mw.y, db * mw.z, db * mw.w);
      }
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float4 pp = *reinterpret_cast<const float4*>(pt + t * 16 + 4 * c4);
        *reinterpret_cast<float4*>(ph + 16 + 4 * c4) = make_float4(db * pp.x, db * pp.y, db * pp.z, db * pp.w);
      }
      dbv[n * Tc + tt] = db;
    }
    __syncthreads();
    for (int i = tid; i < N * 16; i += GB2_THR) {                    // same thread for a given (n, c) in every pass: ordered sum over t
      const int n = i >> 4, c = i & 15;
      float s = dmacc[i];
      for (int tt = 0; tt < tc; ++tt) s += dbv[n * Tc + tt] * H[((size_t)n * T + t0 + tt) * 32 + c];
      dmacc[i] = s;
    }
    if (a.dw)
      for (int q = tid; q < Kk; q += GB2_THR) {
        float s = dwacc[q];
        for (int tt = 0; tt < tc; ++tt) s += DG[q * Tc + tt] * beta[ktgt[q] * Tc + tt];
        dwacc[q] = s;
      }
    __syncthreads();
  }
  for (int i = tid; i < N * 16; i += GB2_THR) a.dmap_part[(size_t)b * N * 16 + i] = dmacc[i];
  if (a.dw) {
    float* dw = a.dw + (size_t)b * a.E;
    for (int e = tid; e < a.E; e += GB2_THR) dw[e] = 0.f;
    __syncthreads();
    for (int q = tid; q < Kk; q += GB2_THR) dw[a.kept[(size_t)b * Kk + q]] = dwacc[q];
  }
}

// mean pairwise L2 distance between the B columns of alpha_all [E, B]  (code/models_rd.py:345-346: cdist(a.T, a.T).mean()).
// part[b] = sum_c ||alpha[:, b] - alpha[:, c]||, fixed order; k_distance_reduce sums the rows and divides by B*B.
__global__ __launch_bounds__(256) void k_distance_rows(const float* __restrict__ alpha, int E, int B, float* __restrict__ part) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < B; c += 256) {
    float s = 0.f;
    // (same order of sums as rounds 2-5; the loads of eight edges are requested before the first is consumed: the loop was one
    // dependent L2 round trip per edge -- 141 us at B = 256, E = 578, a quarter of the time the use_beta step spent outside the
    // graph operator)
    int e = 0;
    for (; e + 8 <= E; e += 8) {
      float xb[8], xc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { xb[u] = alpha[(size_t)(e + u) * B + b]; xc[u] = alpha[(size_t)(e + u) * B + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float df = xb[u] - xc[u]; s += df * df; }
    }
    for (; e < E; ++e) { const float df = alpha[(size_t)e * B + b] - alpha[(size_t)e * B + c]; s += df * df; }
    acc += sqrtf(s);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) part[b] = red[0];
}
__global__ __launch_bounds__(256) void k_distance_reduce(const float* __restrict__ part, int B, float* __restrict__ out) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0] / ((float)B * (float)B);
}

int next_pow2(int x) { return beta_next_pow2(x); }

int check_beta(int B, int N, int K, int T, int d, int E) {
  RD_REQUIRE(B >= 0 && N > 0 && K > 0 && T > 0 && d > 0 && E >= 0, "bad dims");
  RD_REQUIRE(T * d == K, "K (%d) must equal T*d_ob (%d*%d)", K, T, d);
  if (d * 8 != 32) return fail(RD_EUNSUPPORTED, "the beta branch needs d_ob == 4 (increase_dim output viewed as [T, 32], Ob_propagation.py:165)");
  return RD_OK;
}

// RD_BETA_V1=1 (tests, A/B): rounds 2-5's 4-wave kernels instead of round 6's.  Read per call.
bool beta_v1() { const char* e = getenv("RD_BETA_V1"); return e && atoi(e) != 0; }
// RD_BETA_LARGE=1 (tests): the workspace form also for graphs the LDS-staged kernels take.  Read per call.
bool force_large() { const char* e = getenv("RD_BETA_LARGE"); return e && atoi(e) != 0; }
// does one workgroup's LDS hold the graph?  (forward: every step's scores + the sort keys; backward: the edge lists + one chunk of steps)
bool fits_lds(int N, int T, int E, bool bwd) {
  if (N > GB_MAXN || E > GB_MAXE || force_large()) return false;
  const int Kk = (int)((double)E * 0.5), Kc = Kk > 0 ? Kk : 1;
  if (bwd) return lds_bytes(N, bwd_chunk(N, T, Kc), 0, Kc, true) <= 160 * 1024;
  return lds_bytes(N, T, next_pow2(E > 1 ? E : 2), Kc, false) <= 160 * 1024;
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" int32_t rd_graph_beta_kept(int32_t E) { return (int32_t)((double)E * 0.5); }      /* K = int(E * 0.5), :180 */

// bytes of workspace rd_graph_beta_fwd / _bwd need for these dimensions: 0 where the LDS-staged kernels take the graph
extern "C" size_t rd_graph_beta_workspace_bytes(int32_t B, int32_t N, int32_t K, int32_t T, int32_t E) {
  (void)K;
  if (B <= 0 || N <= 0 || T <= 0 || E < 0) return 0;
  if (fits_lds(N, T, E, false) && fits_lds(N, T, E, true)) return 0;
  return beta_large_ws_bytes(B, N, T, E);
}

extern "C" int rd_graph_beta_fwd(int32_t B, int32_t N, int32_t K, int32_t T, int32_t d_ob, int32_t E, const float* V,
                                 const float* H, const float* map_weights, const float* p_t, int64_t pt_bstride,
                                 const int64_t* edge_index, int64_t row_stride, const float* edge_weights, int64_t w_bstride,
                                 float* out, int64_t* edge_index_out, float* alpha_out, float* beta_save, int32_t* kept,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_beta(B, N, K, T, d_ob, E);
  if (rc) return rc;
  if (B == 0) return RD_OK;
  RD_REQUIRE(V && H && map_weights && p_t && edge_index && edge_weights && out && edge_index_out && alpha_out && beta_save && kept,
             "NULL tensor");
  BetaArgs a{};
  a.V = V; a.H = H; a.map_w = map_weights; a.p_t = p_t; a.pt_bstride = pt_bstride; a.ei = edge_index; a.ei_stride = row_stride;
  a.w = edge_weights; a.w_bstride = w_bstride; a.out = out; a.ei_out = edge_index_out; a.alpha_out = alpha_out;
  a.beta_save = beta_save; a.kept = kept;
  a.B = B; a.N = N; a.K = K; a.T = T; a.d = d_ob; a.E = E; a.Kk = rd_graph_beta_kept(E);
  if (!fits_lds(N, T, E, false)) return beta_large_fwd(a, workspace, workspace_bytes, (hipStream_t)stream);
  const int P2 = next_pow2(E > 1 ? E : 2);
  if (!beta_v1() && (K & 3) == 0) {                              // round 6: 16 waves, every (edge, step) quantity once; the same bits
    const int Kc = a.Kk > 0 ? a.Kk : 1;
    const bool stage = lds2_fwd(N, T, K, P2, Kc, true) <= 160 * 1024;
    const size_t lds2 = lds2_fwd(N, T, K, P2, Kc, stage);
    if (stage) { RD_LDS_ATTR(k_graph_beta_fwd2<true>, 160 * 1024); hipLaunchKernelGGL(k_graph_beta_fwd2<true>, dim3(B), dim3(GB2_THR), lds2, (hipStream_t)stream, a, P2); }
    else { RD_LDS_ATTR(k_graph_beta_fwd2<false>, 160 * 1024); hipLaunchKernelGGL(k_graph_beta_fwd2<false>, dim3(B), dim3(GB2_THR), lds2, (hipStream_t)stream, a, P2); }
    return check_launch("k_graph_beta_fwd2");
  }
  const size_t lds = lds_bytes(N, T, P2, a.Kk > 0 ? a.Kk : 1, false);
  RD_LDS_ATTR(k_graph_beta_fwd, 160 * 1024);
  hipLaunchKernelGGL(k_graph_beta_fwd, dim3(B), dim3(GB_THR), lds, (hipStream_t)stream, a, P2);
  return check_launch("k_graph_beta_fwd");
}

extern "C" int rd_graph_beta_bwd(int32_t B, int32_t N, int32_t K, int32_t T, int32_t d_ob, int32_t E, const float* V,
                                 const float* H, const float* map_weights, const float* p_t, int64_t pt_bstride,
                                 const int64_t* edge_index, int64_t row_stride, const float* edge_weights, int64_t w_bstride,
                                 const float* beta_save, const int32_t* kept, const float* dout, float* dV, float* dH,
                                 float* dmap_part, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_beta(B, N, K, T, d_ob, E);
  if (rc) return rc;
  if (B == 0) return RD_OK;
  RD_REQUIRE(V && H && map_weights && p_t && edge_index && edge_weights && beta_save && kept && dout && dV && dH && dmap_part,
             "NULL tensor");
  BetaArgs a{};
  a.V = V; a.H = H; a.map_w = map_weights; a.p_t = p_t; a.pt_bstride = pt_bstride; a.ei = edge_index; a.ei_stride = row_stride;
  a.w = edge_weights; a.w_bstride = w_bstride; a.beta_save = const_cast<float*>(beta_save); a.kept = const_cast<int32_t*>(kept);
  a.dout = dout; a.dV = dV; a.dH = dH; a.dmap_part = dmap_part; a.dw = dw;
  a.B = B; a.N = N; a.K = K; a.T = T; a.d = d_ob; a.E = E; a.Kk = rd_graph_beta_kept(E);
  if (!fits_lds(N, T, E, true)) return beta_large_bwd(a, workspace, workspace_bytes, (hipStream_t)stream);
  const int Kc = a.Kk > 0 ? a.Kk : 1;
  if (!beta_v1() && (K & 3) == 0 && lds2_bwd(N, 1, Kc) <= 160 * 1024) {
    const int Tc2 = bwd2_chunk(N, T, Kc);
    RD_LDS_ATTR(k_graph_beta_bwd2, 160 * 1024);
    hipLaunchKernelGGL(k_graph_beta_bwd2, dim3(B), dim3(GB2_THR), lds2_bwd(N, Tc2, Kc), (hipStream_t)stream, a, Tc2);
    return check_launch("k_graph_beta_bwd2");
  }
  const int Tc = bwd_chunk(N, T, Kc);
  const size_t lds = lds_bytes(N, Tc, 0, Kc, true);
  RD_LDS_ATTR(k_graph_beta_bwd, 160 * 1024);
  hipLaunchKernelGGL(k_graph_beta_bwd, dim3(B), dim3(GB_THR), lds, (hipStream_t)stream, a, Tc);
  return check_launch("k_graph_beta_bwd");
}

extern "C" int rd_structure_distance(int32_t E, int32_t B, const float* alpha_all, float* workspace, float* distance,
                                     void* stream) {
  RD_REQUIRE(E >= 0 && B > 0, "bad dims E=%d B=%d", E, B);
  RD_REQUIRE(alpha_all && workspace && distance, "NULL tensor");
  hipLaunchKernelGGL(k_distance_rows, dim3(B), dim3(256), 0, (hipStream_t)stream, alpha_all, E, B, workspace);
  hipLaunchKernelGGL(k_distance_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, B, distance);
  return check_launch("rd_structure_distance");
}
