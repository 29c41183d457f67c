// rd_trailing.h -- the step's TRAILING launches as device bodies that can also ride inside another kernel's launch.
//
// Two kinds of launches of the training step only produce parameter gradients that nothing later in the backward chain reads: the
// slice reduce of a layer's weight-gradient stream (k_twg_reduce, rd_tile_wgrad.hip) and the classifier head's weight-gradient tiles
// + loss mean (k_head_wgrad, rd_head.hip).  Each is 5-15 us of launch boundary and latency for ~10 MB of traffic, on the critical
// path of a single-stream step.  A forked graph branch was measured four times and lost every time (NOTES.md); what works is to
// give the work to workgroups of the NEXT launch that would otherwise not exist: the backward row-local chain (rd_encfuse.hip)
// fills 178-266 of 256 CUs with one workgroup each, so extra workgroups appended to its grid run on the idle CUs while the chain
// works.  The bodies below are called by the stand-alone kernels and, as "riders", by k_enc_pre_bwd.
// Host side: rd_set_defer_trailing(1) makes rd_head_train / the weight-gradient stream park their trailing launch (trailing_park);
// the next chain launch picks it up (trailing_take), rd_flush_trailing launches whatever is still parked.
#pragma once
#include "rd_common.h"

namespace rd {

typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));

// ---- weight-gradient stream: problems, partials, reduce ------------------------------------------------------------------------
struct TwProb {
  const __bf16 *tA, *tB;                           // dY tiles [S][nctA][2][512], X tiles [S][nctB][2][512]
  float* part;                                     // [TW_SLICES][16 nctA][ldp]
  float *dW, *db;                                  // [N][K], [N] (db may be null)
  int nctA, nctB, N, K, nbk, nmem, ldp;
  int wg0;                                         // first workgroup of the problem in the grid (multiple of 8)
  int q0, nq;                                      // reduce kernel: first quad-thread group of the problem, count
  const int32_t* s32x; int Sx;                     // this problem's own chunk count (device / bound) or null: the launch's
  int hd, hdp, H, D;                               // hd != 0: rows of the A tiles are head-padded ((which, head) blocks of hdp, hd real)
};
// column sums riding on the reduce launch: the LayerNorm dgamma | dbeta partials of the layer ([M rows][N], out1 = first n1 sums)
struct TwColsum { const float* x; int M, N, n1; float *out1, *out2; };
struct TwArgs { TwProb p[4]; int n, S; const __bf16* ones; TwColsum cs[2]; int ncs, nblk_w;
                const int32_t* s32; };             // device count of live 32-row chunks (token plan, rd_plan.h: plan[1]) or null

constexpr int TWR_THR = 1024;

// dW, db = sum over the 8 slices in slice order.  Thread pair (2 lanes) per output quad: lane 0 sums slices 0..3, lane 1
// slices 4..7, combined in that order.  Blocks >= nblk_w: column sums (same arithmetic and order as k_colsum_small,
// rd_gemm.hip: 64 columns x 16 row groups, four interleaved accumulators, fixed-order combine).  1024 threads; red: [16][64] floats of LDS.
__device__ __forceinline__ void twg_reduce_body(const TwArgs& a, int block, float (*red)[64]) {
  if (block >= a.nblk_w) {
    const int cb = block - a.nblk_w;
    const int bpj0 = (a.cs[0].N + 63) / 64;
    const TwColsum J = cb < bpj0 ? a.cs[0] : a.cs[1];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = (cb < bpj0 ? cb : cb - bpj0) * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < J.N) {
      int r = rg;
      for (; r + 48 < J.M; r += 64) {
        s0 += J.x[(long)r * J.N + c]; s1 += J.x[(long)(r + 16) * J.N + c];
        s2 += J.x[(long)(r + 32) * J.N + c]; s3 += J.x[(long)(r + 48) * J.N + c];
      }
      for (; r < J.M; r += 16) s0 += J.x[(long)r * J.N + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < J.N) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) v += red[q][cl];
      if (c < J.n1) J.out1[c] = v; else J.out2[c - J.n1] = v;
    }
    return;
  }
  const int g = (int)((block * (long)TWR_THR + threadIdx.x) >> 1), half = threadIdx.x & 1;
  TwProb P = a.p[0];                                // q0 are multiples of 512: a workgroup never straddles two problems
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.n && block * (TWR_THR / 2) >= a.p[i].q0) P = a.p[i];
  const int e = g - P.q0;
  const bool live = e < P.nq;
  const int qpr = P.ldp >> 2;
  const int ec = live ? e : 0;
  const int n = ec / qpr, k = 4 * (ec - n * qpr);
  int nr = n;                                       // row of dW / db this partial row belongs to
  bool rok = true;
  if (P.hd) {                                       // head-padded rows: (which, head, c) -> which D + head hd + c, c < hd
    const int blk = P.H * P.hdp, which = n / blk, rem = n - which * blk, hh = rem / P.hdp, c = rem - hh * P.hdp;
    rok = c < P.hd;
    nr = which * P.D + hh * P.hd + c;
  }
  const bool is_w = rok && k < P.K, is_b = rok && (k == 16 * P.nctB) && P.db != nullptr;
  const size_t stride = (size_t)16 * P.nctA * P.ldp;
  const float* p = P.part + (size_t)n * P.ldp + k + (size_t)(4 * half) * stride;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live && (is_w || is_b)) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  const float4 r = make_float4(__shfl_down(s.x, 1, 2), __shfl_down(s.y, 1, 2), __shfl_down(s.z, 1, 2), __shfl_down(s.w, 1, 2));
  if (live && half == 0) {
    s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
    if (is_w) *reinterpret_cast<float4*>(P.dW + (size_t)nr * P.K + k) = s;
    else if (is_b) P.db[nr] = s.x;
  }
}

// ---- classifier head: dW[n,k] = sum_b u[b,n] v[b,k], db[n] = sum_b u[b,n]: 16 x 16 output tile per 256-thread group ------------
struct HwJob { const float* u; const float* v; float *dW, *db; int ldu, ldv, N, K, tiles_k, blk0; };
struct HwArgs { HwJob j[3]; int n, B, ntiles; const float* lossr; float* loss; };

// tile `tile` (< a.ntiles; beyond: the group only keeps the barriers) by the 256 threads t = 0..255 of a group; us / vs: the
// group's [256][16] / [256][17] floats of LDS.  EVERY thread of the workgroup must call this (it contains workgroup barriers).
__device__ __forceinline__ void head_wgrad_body(const HwArgs& a, int tile, int t, float (*us)[16], float (*vs)[17]) {
  const bool on = tile < a.ntiles;
  HwJob J = a.j[0];
#pragma unroll
  for (int i = 1; i < 3; ++i)
    if (i < a.n && tile >= a.j[i].blk0) J = a.j[i];
  const int local = on ? tile - J.blk0 : 0;
  const int tn = local / J.tiles_k, tk = local - tn * J.tiles_k;
  const int ni = t >> 4, ki = t & 15;
  const int n = 16 * tn + ni, k = 16 * tk + ki;
  float acc = 0.f, bacc = 0.f;
  for (int bb = 0; bb < a.B; bb += 256) {
    float ur[16], vr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i, row = e >> 4, col = e & 15, b = bb + row;
      ur[i] = 0.f; vr[i] = 0.f;
      if (on && b < a.B && 16 * tn + col < J.N) ur[i] = J.u[(long)b * J.ldu + 16 * tn + col];
      if (on && b < a.B && 16 * tk + col < J.K) vr[i] = J.v[(long)b * J.ldv + 16 * tk + col];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = t + 256 * i, row = e >> 4, col = e & 15;
      us[row][col] = ur[i]; vs[row][col] = vr[i];
    }
    __syncthreads();
#pragma unroll 8
    for (int b = 0; b < 256; ++b) { acc += us[b][ni] * vs[b][ki]; bacc += us[b][ni]; }
  }
  if (on && n < J.N && k < J.K) J.dW[(long)n * J.K + k] = acc;
  if (on && tk == 0 && ki == 0 && n < J.N && J.db) J.db[n] = bacc;
  if (tile == 0 && t < 64 && a.loss) {                // loss = mean of the per-sample losses (lane-strided, then lanes in order)
    float s = 0.f;
    for (int b = t; b < a.B; b += 64) s += a.lossr[b];
    s = wave_sum64_dpp(s);
    if (t == 0) *a.loss = s / (float)a.B;
  }
}
constexpr size_t HW_GROUP_LDS = (size_t)(256 * 16 + 256 * 17) * sizeof(float);     // bytes of LDS per 256-thread group

// ---- riders --------------------------------------------------------------------------------------------------------------------
struct RiderArgs { int kind, nblocks; TwArgs tw; HwArgs hw; };     // kind 0: none, 1: head weight gradients (4 tiles per block), 2: slice reduce
constexpr int RIDER_NONE = 0, RIDER_HEAD = 1, RIDER_TWG = 2;

// block `rb` of the rider by a 1024-thread workgroup; lds: >= 4 * HW_GROUP_LDS bytes
__device__ __forceinline__ void rider_body(const RiderArgs& r, int rb, unsigned char* lds) {
  if (r.kind == RIDER_TWG) {
    twg_reduce_body(r.tw, rb, reinterpret_cast<float (*)[64]>(lds));
  } else if (r.kind == RIDER_HEAD) {
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
    float* base = reinterpret_cast<float*>(lds + (size_t)grp * HW_GROUP_LDS);
    head_wgrad_body(r.hw, 4 * rb + grp, t, reinterpret_cast<float (*)[16]>(base), reinterpret_cast<float (*)[17]>(base + 256 * 16));
  }
}

// host side (rd_api.hip): the parked trailing launch of this host thread
bool trailing_deferred();
// park `r` (kind != 0).  If something is parked already it is launched stand-alone on `st` first.
int trailing_park(const RiderArgs& r, hipStream_t st);
// hand the parked launch over (kind 0 when nothing is parked) and clear the slot
RiderArgs trailing_take();
// stand-alone launch of a rider (rd_api.hip)
int trailing_launch(const RiderArgs& r, hipStream_t st);

}  // namespace rd
