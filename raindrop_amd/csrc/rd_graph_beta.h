// rd_graph_beta.h -- argument block shared by the two forms of the use_beta graph operator (rd_graph_beta.hip: one workgroup per
// sample graph, graph staged in LDS; rd_graph_beta_large.hip: any graph size, state in a caller-provided workspace).
#pragma once
#include "rd_common.h"

namespace rd {

struct BetaArgs {
  const float *V, *H;                    // [B,N,K] relu(lin_value(x)),  [B,N,T*32] increase_dim(x)
  const float *map_w, *p_t;              // [N,16], [B or 1][T,16]
  const int64_t* ei; int64_t ei_stride;  // edge_index rows (source; target), shared by the batch
  const float* w; long w_bstride;        // [E] edge weights (per-sample stride, 0 = shared)
  long pt_bstride;
  float* out;                            // [B,N,K]
  int64_t* ei_out; float* alpha_out;     // [B][2,Kk] kept edges in pruning order, [B][Kk] mean kept score
  float* beta_save;                      // [B,N,T]
  int32_t* kept;                         // [B][Kk] original edge ids in pruning order (for backward)
  // backward
  const float* dout; float *dV, *dH, *dmap_part, *dw;   // dmap_part [B,N,16]; dw [B,E] or null
  int B, N, K, T, d, E, Kk;
};

// edge endpoint -> node index that is always legal (raindrop_amd.ops.graph_beta validates the range and raises like the
// reference's index_select; the kernels must not read out of range whatever they are handed)
__device__ __forceinline__ int node_of(int64_t v, int N) { return v < 0 ? 0 : (v >= N ? N - 1 : (int)v); }

__device__ __forceinline__ unsigned sortable_desc(float x) {          // larger float -> smaller key
  unsigned u = __float_as_uint(x);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                     // ascending order-preserving map
  return ~u;
}

inline int beta_next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// rd_graph_beta_large.hip
size_t beta_large_ws_bytes(int B, int N, int T, int E);
int beta_large_fwd(const BetaArgs& a, void* ws, size_t ws_bytes, hipStream_t st);
int beta_large_bwd(const BetaArgs& a, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace rd
