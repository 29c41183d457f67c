// rd_graph.hip -- sensor-graph construction (integer work, bit-exact), per-target edge softmax
// with wavefront-shuffle reductions, and the positional-encoding / padding-mask kernel.
#include "rd_common.h"
#include "rd_plan.h"
#include "rd_rng.h"

namespace rd {
namespace {

// code/models_rd.py:307-311.  One 1024-thread workgroup: thread i owns the contiguous, row-major
// segment [i*per, (i+1)*per) of the F*F adjacency, counts its non-zeros, an LDS scan turns the
// counts into output offsets, and each thread then emits its edges in order -- so the edge list
// has exactly torch.nonzero's row-major order.
__global__ __launch_bounds__(1024) void k_graph_build(const float* __restrict__ gs, int F,
                                                      float* __restrict__ adj,
                                                      int64_t* __restrict__ edge_index,
                                                      float* __restrict__ edge_weights,
                                                      int32_t* __restrict__ n_edges) {
  __shared__ int scan[1024];
  const int tid = threadIdx.x;
  const long total = (long)F * F;
  const long per = (total + 1023) / 1024;
  const long beg = min(total, tid * per), end = min(total, beg + per);
  int cnt = 0;
  for (long e = beg; e < end; ++e) {
    const int r = (int)(e / F), c = (int)(e - (long)r * F);
    const float v = (r == c) ? 1.0f : gs[e];
    adj[e] = v;
    cnt += (v != 0.0f);
  }
  scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan
    const int v = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  long o = scan[tid] - cnt;
  for (long e = beg; e < end; ++e) {
    const int r = (int)(e / F), c = (int)(e - (long)r * F);
    const float v = (r == c) ? 1.0f : gs[e];
    if (v != 0.0f) {
      edge_index[o] = r;              // row 0: source j
      edge_index[total + o] = c;      // row 1: target i (row stride F*F)
      edge_weights[o] = v;
      ++o;
    }
  }
  if (tid == 1023) *n_edges = scan[1023];
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// torch_geometric.utils.softmax(edge_weights, index=target) on the dense adjacency
// (code/Ob_propagation.py:195): one wavefront per target node i, lanes stride over the sources j.
__global__ __launch_bounds__(256) void k_edge_softmax(const float* __restrict__ adj, int F,
                                                      float* __restrict__ gamma,
                                                      float* __restrict__ ssum) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= F) return;
  float m = -INFINITY;
  for (int j = lane; j < F; j += 64) {
    const float w = adj[(long)j * F + i];
    if (w != 0.f) m = fmaxf(m, w);
  }
  m = wave_max(m);
  float den = 0.f;
  for (int j = lane; j < F; j += 64) {
    const float w = adj[(long)j * F + i];
    if (w != 0.f) den += expf(w - m);
  }
  den = wave_sum(den) + 1e-16f;
  float tot = 0.f;
  for (int j = lane; j < F; j += 64) {
    const float w = adj[(long)j * F + i];
    const float g = (w != 0.f) ? expf(w - m) / den : 0.f;
    gamma[(long)j * F + i] = g;
    tot += g;
  }
  tot = wave_sum(tot);
  if (lane == 0) ssum[i] = tot;
}

// code/models_rd.py:28-38 (PE) and :298-299 (padding mask).  Thread per (t,b): 2*H transcendental
// evaluations written straight into the PE columns of the concat buffer (no torch.cat pass).
__global__ __launch_bounds__(256) void k_pe_mask(const float* __restrict__ times,
                                                 const int64_t* __restrict__ lengths,
                                                 const float* __restrict__ ts, float* __restrict__ z,
                                                 uint8_t* __restrict__ mask, int T, int B, int D,
                                                 int Dm, int H, const int32_t* __restrict__ sp_row0, const int32_t* __restrict__ sp_len) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)T * B) return;
  const int t = (int)(i / B), b = (int)(i - (long)t * B);
  mask[(long)b * T + t] = (uint8_t)((int64_t)t >= lengths[b]);
  long zrow = i;
  if (sp_row0) {                                   // token plan: the step's row is sample b's first row + t; padded steps have none
    if (t >= sp_len[b]) return;
    zrow = (long)sp_row0[b] + t;
  }
  const float tm = times[i];
  float* row = z + zrow * D + Dm;
  for (int k = 0; k < H; ++k) {
    const float a = tm / ts[k];
    float sn, cs;
    sincosf(a, &sn, &cs);            // same call as the fused stage (rd_msgpass_fused.hip): the two paths stay bit-equal
    row[k] = sn;
    row[H + k] = cs;
  }
}

// torch_geometric.utils.softmax over an explicit edge list (duplicates allowed), normalised by
// row `norm_row` of edge_index (1 = target: Observation_progation default / TransformerConv,
// code/Ob_propagation.py:195, code/transformer_conv.py:201; 0 = source: the use_beta branch,
// code/Ob_propagation.py:184).  One wavefront per node scans the list three times.
// blockIdx.y = graph of a batch (strides 0: shared)
__global__ __launch_bounds__(256) void k_edge_softmax_list(const int64_t* __restrict__ idx, int E,
                                                           const float* __restrict__ w, int N,
                                                           float* __restrict__ gamma_e,
                                                           float* __restrict__ ssum, long idx_bstride, long w_bstride,
                                                           float p_drop, uint64_t seed, const uint64_t* cell) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  idx += (long)blockIdx.y * idx_bstride; w += (long)blockIdx.y * w_bstride;
  gamma_e += (long)blockIdx.y * E; ssum += (long)blockIdx.y * N;
  // optional dropout of the coefficients AFTER the softmax (code/Ob_propagation.py:196, code/transformer_conv.py:203): keep with
  // probability 1 - p, scale 1 / (1 - p); a pure function of (seed + cell, edge) like every mask of the library
  const bool drop = p_drop > 0.f;
  const uint64_t seed_eff = drop ? eff_seed(seed, cell) : 0;
  const float inv_keep = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  float m = -INFINITY;
  for (int e = lane; e < E; e += 64)
    if (idx[e] == n) m = fmaxf(m, w[e]);
  m = wave_max(m);
  float den = 0.f;
  for (int e = lane; e < E; e += 64)
    if (idx[e] == n) den += expf(w[e] - m);
  den = wave_sum(den) + 1e-16f;
  float tot = 0.f;
  for (int e = lane; e < E; e += 64)
    if (idx[e] == n) {
      float g = expf(w[e] - m) / den;
      if (drop) {
        const float4 u = uniform4(seed_eff, SITE_EDGE_COEFF, (uint64_t)blockIdx.y * ((uint64_t)(E + 3) >> 2) + (uint64_t)(e >> 2));
        const float ue = (e & 3) == 0 ? u.x : (e & 3) == 1 ? u.y : (e & 3) == 2 ? u.z : u.w;
        g = ue >= p_drop ? g * inv_keep : 0.f;
      }
      gamma_e[e] = g;
      tot += g;
    }
  tot = wave_sum(tot);
  if (lane == 0) ssum[n] = tot;
}

// dense coefficient matrix of an edge list: G[j*N + i] = sum of gamma_e over the edges (j -> i), duplicates added in EDGE ORDER
// (the scatter-add of code/transformer_conv.py:205 via PyG aggregate, made deterministic).  One wavefront per target i; the edge
// list is walked by the whole wave (uniform loads), the lane that owns source j = src % 64 adds into its LDS row slot.
__global__ __launch_bounds__(256) void k_edge_gamma_dense(const int64_t* __restrict__ src, const int64_t* __restrict__ tgt, int E,
                                                          const float* __restrict__ gamma_e, int N, float* __restrict__ G) {
  extern __shared__ float grow[];                       // [4 waves][N]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wv;
  float* row = grow + (size_t)wv * N;
  if (i < N) {
    for (int j = lane; j < N; j += 64) row[j] = 0.f;
    for (int e = 0; e < E; ++e) {
      if ((int)tgt[e] != i) continue;                   // wave-uniform
      const int j = (int)src[e];
      if (lane == (j & 63) && j >= 0 && j < N) row[j] += gamma_e[e];   // one lane per source: edge order, no atomics
    }
    for (int j = lane; j < N; j += 64) G[(long)j * N + i] = row[j];
  }
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" int rd_graph_build(int32_t F, const float* global_structure, float* adj_out,
                              int64_t* edge_index, float* edge_weights, int32_t* n_edges,
                              void* stream) {
  RD_REQUIRE(F > 0 && F <= 4096, "F=%d out of range", F);
  RD_REQUIRE(global_structure && adj_out && edge_index && edge_weights && n_edges, "NULL tensor");
  hipLaunchKernelGGL(k_graph_build, dim3(1), dim3(1024), 0, (hipStream_t)stream, global_structure,
                     F, adj_out, edge_index, edge_weights, n_edges);
  return check_launch("k_graph_build");
}

extern "C" int rd_edge_softmax(int32_t F, const float* adj, float* gamma, float* ssum, void* stream) {
  RD_REQUIRE(F > 0 && F <= 4096, "F=%d out of range", F);
  RD_REQUIRE(adj && gamma && ssum, "NULL tensor");
  hipLaunchKernelGGL(k_edge_softmax, dim3(cdiv(F, 4)), dim3(256), 0, (hipStream_t)stream, adj, F,
                     gamma, ssum);
  return check_launch("k_edge_softmax");
}

extern "C" int rd_pe_mask(const rd_shape* s, const float* times, const int64_t* lengths,
                          const float* timescales, float* z, uint8_t* mask, void* stream) {
  RD_REQUIRE(s && s->T > 0 && s->F > 0 && s->d_ob > 0 && s->B >= 0, "bad rd_shape");
  RD_REQUIRE(s->d_pe > 0 && (s->d_pe % 2) == 0, "d_pe must be even and positive");
  if (s->B == 0) return RD_OK;                       // empty batch: nothing to do (and torch hands out NULL data)
  RD_REQUIRE(times && lengths && timescales && z && mask, "NULL tensor");
  const long n = (long)s->T * s->B;
  const int Dm = s->F * s->d_ob, D = Dm + s->d_pe;
  const int32_t* tp = token_plan();                  // registered plan: the PE rows follow it (rd_plan.h brow / blen)
  hipLaunchKernelGGL(k_pe_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     times, lengths, timescales, z, mask, s->T, s->B, D, Dm, s->d_pe / 2,
                     tp ? tp + plan::brow_base(s->B, s->T) : nullptr, tp ? tp + plan::blen_base(s->B, s->T) : nullptr);
  return check_launch("k_pe_mask");
}

extern "C" int rd_edge_softmax_list(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride,
                                    int32_t norm_row, const float* edge_weights, float* gamma_e,
                                    float* ssum, void* stream) {
  RD_REQUIRE(N > 0 && E >= 0, "bad N=%d E=%d", N, E);
  RD_REQUIRE(norm_row == 0 || norm_row == 1, "norm_row must be 0 (source) or 1 (target)");
  RD_REQUIRE(edge_index && edge_weights && gamma_e && ssum, "NULL tensor");
  hipLaunchKernelGGL(k_edge_softmax_list, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream,
                     edge_index + (long)norm_row * row_stride, E, edge_weights, N, gamma_e, ssum, 0L, 0L, 0.f, (uint64_t)0, nullptr);
  return check_launch("k_edge_softmax_list");
}

// the same with dropout of the coefficients after the softmax (training mode of an operator built with dropout > 0)
extern "C" int rd_edge_softmax_list_dropout(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride, int32_t norm_row,
                                            const float* edge_weights, float p_drop, uint64_t seed, float* gamma_e, float* ssum,
                                            void* stream) {
  RD_REQUIRE(N > 0 && E >= 0, "bad N=%d E=%d", N, E);
  RD_REQUIRE(norm_row == 0 || norm_row == 1, "norm_row must be 0 (source) or 1 (target)");
  RD_REQUIRE(edge_index && edge_weights && gamma_e && ssum, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  hipLaunchKernelGGL(k_edge_softmax_list, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream,
                     edge_index + (long)norm_row * row_stride, E, edge_weights, N, gamma_e, ssum, 0L, 0L, p_drop, seed, seed_cell());
  return check_launch("k_edge_softmax_list");
}

// B graphs in one launch: edge lists `batch_stride` int64 apart (0: one shared list), weights `w_bstride` floats apart (0: shared);
// gamma_e [B,E], ssum [B,N].  The per-sample pruned edge lists of the use_beta branch feed layer 2 through this
// (code/models_rd.py:331-335: edge_index_layer2 / edge_weights_layer2 differ per sample).
extern "C" int rd_edge_softmax_list_batched(int32_t B, int32_t N, int32_t E, const int64_t* edge_index, int64_t batch_stride,
                                            int64_t row_stride, int32_t norm_row, const float* edge_weights, int64_t w_bstride,
                                            float* gamma_e, float* ssum, void* stream) {
  RD_REQUIRE(B >= 0 && N > 0 && E >= 0, "bad B=%d N=%d E=%d", B, N, E);
  RD_REQUIRE(norm_row == 0 || norm_row == 1, "norm_row must be 0 (source) or 1 (target)");
  if (B == 0) return RD_OK;
  RD_REQUIRE(B <= 65535, "B=%d exceeds the grid's y extent", B);
  RD_REQUIRE(edge_index && edge_weights && gamma_e && ssum, "NULL tensor");
  hipLaunchKernelGGL(k_edge_softmax_list, dim3(cdiv(N, 4), B), dim3(256), 0, (hipStream_t)stream,
                     edge_index + (long)norm_row * row_stride, E, edge_weights, N, gamma_e, ssum, (long)batch_stride, (long)w_bstride,
                     0.f, (uint64_t)0, nullptr);
  return check_launch("k_edge_softmax_list");
}

extern "C" int rd_edge_gamma_dense(int32_t N, int32_t E, const int64_t* edge_index, int64_t row_stride, const float* gamma_e,
                                   float* gamma_dense, void* stream) {
  RD_REQUIRE(N > 0 && N <= 4096 && E >= 0, "bad N=%d E=%d", N, E);
  RD_REQUIRE(edge_index && gamma_e && gamma_dense, "NULL tensor");
  hipLaunchKernelGGL(k_edge_gamma_dense, dim3(cdiv(N, 4)), dim3(256), (size_t)4 * N * sizeof(float), (hipStream_t)stream, edge_index,
                     edge_index + row_stride, E, gamma_e, N, gamma_dense);
  return check_launch("k_edge_gamma_dense");
}

// out[i,c] = sum_j gamma[j,i] * V[j,c] (+ skip[i,c])  -- the source-valued aggregate of
// TransformerConv (code/transformer_conv.py:158,168-175,205-206) on a dense coefficient matrix.
extern "C" int rd_aggregate_fwd(int32_t N, int32_t C, const float* gamma, const float* V,
                                const float* skip, float* out, void* stream) {
  RD_REQUIRE(N > 0 && C > 0, "bad N=%d C=%d", N, C);
  RD_REQUIRE(gamma && V && out, "NULL tensor");
  GemmArgs g{};
  g.M = N; g.N = C; g.K = N; g.nsplit = 1;
  g.A = gamma; g.sa_m = 1; g.sa_k = N;     // A(i,j) = gamma[j*N + i]
  g.B = V; g.sb_n = 1; g.sb_k = C;         // B(c,j) = V[j*C + c]
  g.C = out; g.sc_m = C;
  g.residual = skip; g.res_m = C;
  return launch_gemm(g, (hipStream_t)stream);
}

// The same for B feature matrices V [B,N,C] that share one coefficient matrix (one batched product; the legacy `Raindrop` model,
// code/models_rd.py:155-165, calls the operator once per sample of a batch with the same graph).
extern "C" int rd_aggregate_batched_fwd(int32_t B, int32_t N, int32_t C, const float* gamma, const float* V, const float* skip,
                                        float* out, void* stream) {
  RD_REQUIRE(B >= 0 && N > 0 && C > 0, "bad B=%d N=%d C=%d", B, N, C);
  if (B == 0) return RD_OK;
  RD_REQUIRE(gamma && V && out, "NULL tensor");
  if (B == 1) return rd_aggregate_fwd(N, C, gamma, V, skip, out, stream);
  GemmArgs g{};
  g.M = N; g.N = C; g.K = N; g.nsplit = 1;
  g.A = gamma; g.sa_m = 1; g.sa_k = N;
  g.B = V; g.sb_n = 1; g.sb_k = C;
  g.C = out; g.sc_m = C;
  g.nbatch = B; g.batch_inner = 1; g.a_bo = 0; g.b_bo = (long)N * C; g.c_bo = (long)N * C;
  g.residual = skip; g.res_m = C; g.res_batched = 1;
  return launch_gemm(g, (hipStream_t)stream);
}
extern "C" int rd_aggregate_batched_bwd(int32_t B, int32_t N, int32_t C, const float* gamma, const float* dout, float* dV,
                                        void* stream) {
  RD_REQUIRE(B >= 0 && N > 0 && C > 0, "bad B=%d N=%d C=%d", B, N, C);
  if (B == 0) return RD_OK;
  RD_REQUIRE(gamma && dout && dV, "NULL tensor");
  if (B == 1) return rd_aggregate_bwd(N, C, gamma, dout, dV, stream);
  GemmArgs g{};
  g.M = N; g.N = C; g.K = N; g.nsplit = 1;
  g.A = gamma; g.sa_m = N; g.sa_k = 1;
  g.B = dout; g.sb_n = 1; g.sb_k = C;
  g.C = dV; g.sc_m = C;
  g.nbatch = B; g.batch_inner = 1; g.a_bo = 0; g.b_bo = (long)N * C; g.c_bo = (long)N * C;
  return launch_gemm(g, (hipStream_t)stream);
}

// dV[j,c] = sum_i gamma[j,i] * dout[i,c]
extern "C" int rd_aggregate_bwd(int32_t N, int32_t C, const float* gamma, const float* dout,
                                float* dV, void* stream) {
  RD_REQUIRE(N > 0 && C > 0, "bad N=%d C=%d", N, C);
  RD_REQUIRE(gamma && dout && dV, "NULL tensor");
  GemmArgs g{};
  g.M = N; g.N = C; g.K = N; g.nsplit = 1;
  g.A = gamma; g.sa_m = N; g.sa_k = 1;     // A(j,i) = gamma[j*N + i]
  g.B = dout; g.sb_n = 1; g.sb_k = C;      // B(c,i) = dout[i*C + c]
  g.C = dV; g.sc_m = C;
  return launch_gemm(g, (hipStream_t)stream);
}
