// rd_tile_wgrad.hip -- weight gradients of the temporal encoder's four dense layers as ONE streaming launch per layer:
//   dW[n, k] = sum over the M = T*B token rows m of dY[m, n] * X[m, k],   db[n] = sum_m dY[m, n]
// (autograd of the nn.Linear / in_proj products of torch's TransformerEncoderLayer as used at code/models_rd.py:235-237).
//
// Both operands arrive as split-bf16 ROW TILES -- [chunk s of 32 rows][column tile j of 16][hi, lo][64 lanes][8], every
// kilobyte one MFMA operand fragment with the 32 rows as the reduction index -- exported by the row-block GEMM that
// consumed the same tensor as its A operand (rd_rowgemm.hip: the split planes are in LDS there anyway; a transposing
// LDS read turns them into tiles).  X tiles (x, attn, x1, h) are written by the forward, dY tiles (dqkv, dout, du, df) by
// the input-gradient products of the backward.  So this kernel is the same pure stream as the message-passing weight
// gradient (rd_msgpass_dw.hip): one 16-byte load per lane and tile part -> v_mfma_f32_16x16x32_bf16, no conversion, no
// transposition, no LDS and no barrier in the main loop.  It replaces four split-K GEMMs + four reduces per layer, each
// of which converted both fp32 operands once per 64x64 output tile (PMC: 90 MB fetched for 19-38 MB of operands, waves
// waiting 58 % of their cycles).
//
// Decomposition: a problem's [16 nctA x 16 (nctB + 1)] output (the extra column tile multiplies a constant "ones" tile:
// its column 0 is the bias gradient) is cut into NA x NB-tile blocks; the S chunks into 8 interleaved slices, slice z
// on XCD z (workgroup id % 8), so every XCD fetches one eighth of every operand exactly once into its L2.  Workgroup =
// (problem, block, slice): its 4 waves each own the WHOLE block (NA*NB*4 accumulator registers) and take every 4th
// chunk of the slice through a ring of three register buffers; they are summed through LDS in wave order at the end.
// k_twg_reduce adds the 8 slice partials in slice order (deterministic; no floating-point atomics).
#include "rd_common.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TILE = 512;                          // bf16 elements of one tile part (64 lanes x 8)
constexpr int TW_THR = 256, TW_NA = 5, TW_NB = 6, TW_SLICES = 8;
constexpr int TW_LDC = 16 * TW_NB + 4;             // fp32 row stride of a wave's block in LDS
constexpr int TW_LDS = 4 * 16 * TW_NA * TW_LDC * 4;   // 128 KB: the four waves' blocks for the final sum

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

struct Frag { bf16x8 ah[TW_NA], al[TW_NA], bh[TW_NB], bl[TW_NB]; };

// ONE: the single-product arithmetic mode (RD_PREC_BF16: hi * hi only).  The lo halves of the tiles are neither read nor multiplied --
// half the operand bytes of a kernel that is a pure stream, one third of its MFMAs.
template <bool ONE>
__global__ __launch_bounds__(TW_THR) void k_twg(TwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TwProb P = a.p[0];                                // uniform selects (a dynamic index would move the table to scratch)
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.n && (int)blockIdx.x >= a.p[i].wg0) P = a.p[i];
  const int local = blockIdx.x - P.wg0, sl = local & 7, mi = local >> 3;     // slice == XCD of this workgroup
  if (mi >= P.nmem) return;
  const int bn = mi / P.nbk, bk = mi - bn * P.nbk;
  const int nctA = P.nctA, nctB = P.nctB;
  const int32_t* sp = P.s32x ? P.s32x : a.s32;
  const int Sb = P.s32x ? P.Sx : a.S;
  const int S = sp ? min(Sb, __builtin_amdgcn_readfirstlane(*sp)) : Sb;           // chunks beyond the live rows hold nothing (never exported)
  const int ntile = S > sl ? (S - sl + TW_SLICES - 1) / TW_SLICES : 0;        // chunks of this slice: sl, sl + 8, ...

  // operand tile pointers of chunk 0 (+ lane offset) and their per-chunk strides.  The B column tile with index nctB is
  // the constant "ones" tile (stride 0).  Tiles beyond an operand's range map to a valid tile; their products are
  // computed and dropped at the store.
  const size_t stepA = (size_t)nctA * 2 * TILE, stepB = (size_t)nctB * 2 * TILE;
  const __bf16 *pa[TW_NA], *pb[TW_NB]; size_t sb[TW_NB];
#pragma unroll
  for (int i = 0; i < TW_NA; ++i) pa[i] = P.tA + (size_t)min(TW_NA * bn + i, nctA - 1) * 2 * TILE + lane * 8;
#pragma unroll
  for (int i = 0; i < TW_NB; ++i) {
    const int kt = TW_NB * bk + i;
    if (kt == nctB) { pb[i] = a.ones + lane * 8; sb[i] = 0; }
    else { pb[i] = P.tB + (size_t)min(kt, nctB - 1) * 2 * TILE + lane * 8; sb[i] = stepB; }
  }
  // chunk i of this wave is slice-local index wave + 4 i; i >= nst is a GHOST: its A operands come from a zero tile,
  // so every wave runs the same branch-free trip count and the s_waitcnt bookkeeping stays exact (rd_msgpass_dw.hip)
  const int nst = ntile > wave ? (ntile - wave + 3) / 4 : 0;
  const __bf16* zt = a.ones + TILE + lane * 8;                        // [ones hi][zeros][zeros]
  auto load = [&](Frag& f, int i) {
    const bool ghost = i >= nst;
    const size_t s = (size_t)(ghost ? 0 : sl + TW_SLICES * (wave + 4 * i));
#pragma unroll
    for (int t = 0; t < TW_NA; ++t) {
      const __bf16* qa = ghost ? zt : pa[t] + s * stepA;
      f.ah[t] = *reinterpret_cast<const bf16x8*>(qa);
      if (!ONE) f.al[t] = *reinterpret_cast<const bf16x8*>(qa + TILE);
    }
#pragma unroll
    for (int t = 0; t < TW_NB; ++t) {
      const __bf16* qb = pb[t] + s * sb[t];
      f.bh[t] = *reinterpret_cast<const bf16x8*>(qb);
      if (!ONE) f.bl[t] = *reinterpret_cast<const bf16x8*>(qb + TILE);
    }
  };
  f32x4 acc[TW_NA][TW_NB];
#pragma unroll
  for (int ni = 0; ni < TW_NA; ++ni)
#pragma unroll
    for (int ki = 0; ki < TW_NB; ++ki) acc[ni][ki] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const Frag& f) {
    if (!ONE) {
#pragma unroll
      for (int ni = 0; ni < TW_NA; ++ni)
#pragma unroll
        for (int ki = 0; ki < TW_NB; ++ki)
          acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[ni], f.bh[ki], acc[ni][ki], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < TW_NA; ++ni)
#pragma unroll
        for (int ki = 0; ki < TW_NB; ++ki)
          acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[ni], f.bl[ki], acc[ni][ki], 0, 0, 0);
    }
#pragma unroll
    for (int ni = 0; ni < TW_NA; ++ni)
#pragma unroll
      for (int ki = 0; ki < TW_NB; ++ki)
        acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[ni], f.bh[ki], acc[ni][ki], 0, 0, 0);
  };

  const int nmax = (ntile + 3) / 4;                                   // chunk count of wave 0 (the largest)
  Frag f0, f1, f2;
  load(f0, 0); load(f1, 1);
  for (int it = 0; it < nmax; it += 3) {
    load(f2, it + 2); mma(f0);
    load(f0, it + 3); mma(f1);
    load(f1, it + 4); mma(f2);
  }

  // ---- in-workgroup sum of the four waves' blocks (fixed wave order) -> slice partial ----
  constexpr int BR = 16 * TW_NA, BC = 16 * TW_NB;
  float* Cs = reinterpret_cast<float*>(tsm) + (size_t)wave * BR * TW_LDC;
#pragma unroll
  for (int ni = 0; ni < TW_NA; ++ni)
#pragma unroll
    for (int ki = 0; ki < TW_NB; ++ki)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        Cs[(16 * ni + 4 * (lane >> 4) + i) * TW_LDC + 16 * ki + (lane & 15)] = acc[ni][ki][i];
  __syncthreads();
  const float* C0 = reinterpret_cast<const float*>(tsm);
  const int nrows = min(BR, 16 * nctA - BR * bn), ncols = min(BC, P.ldp - BC * bk);   // multiples of 16
  float* out = P.part + ((size_t)sl * 16 * nctA + BR * bn) * P.ldp + BC * bk;
  const int qpr = ncols >> 2;
  for (int e = tid; e < nrows * qpr; e += TW_THR) {
    const int r = e / qpr, c4 = e - r * qpr;
    const float* q = C0 + r * TW_LDC + 4 * c4;
    float4 v = *reinterpret_cast<const float4*>(q);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(q + (size_t)w * BR * TW_LDC);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)r * P.ldp + 4 * c4) = v;
  }
}

// dW, db = sum over the 8 slices in slice order.  Thread pair (2 lanes) per output quad: lane 0 sums slices 0..3, lane 1
// slices 4..7, combined in that order.  Workgroups >= nblk_w: column sums (same arithmetic and order as k_colsum_small,
// rd_gemm.hip: 64 columns x 16 row groups, four interleaved accumulators, fixed-order combine).
constexpr int TWR_THR = 1024;
__global__ __launch_bounds__(TWR_THR) void k_twg_reduce(TwArgs a) {
  if ((int)blockIdx.x >= a.nblk_w) {
    __shared__ float red[16][64];
    const int cb = blockIdx.x - a.nblk_w;
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
  const int g = (int)((blockIdx.x * (long)TWR_THR + threadIdx.x) >> 1), half = threadIdx.x & 1;
  TwProb P = a.p[0];                                // q0 are multiples of 512: a workgroup never straddles two problems
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.n && (int)blockIdx.x * (TWR_THR / 2) >= a.p[i].q0) P = a.p[i];
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

}  // namespace

// ---- host interface (rd_temporal.hip) ---------------------------------------------------------------------
size_t tile_elems(long M, int cols) { return (size_t)cdiv((int)M, 32) * cdiv(cols, 16) * 2 * TILE; }   // bf16 elements of a tile tensor
size_t tile_wgrad_ones_elems() { return 3 * TILE; }   // [ones hi][zeros][zeros], written by k_wsplit (rd_rowgemm.hip)
size_t tile_wgrad_part_floats(int N, int K) {
  return (size_t)TW_SLICES * 16 * cdiv(N, 16) * 16 * (cdiv(K, 16) + 1);
}
bool tile_wgrad_ok(int N, int K) {
  const char* e = getenv("RD_TILE_WGRAD");           // read per call (tests compare both paths in one process)
  const bool on = !(e && atoi(e) == 0);
  return on && precision() != RD_PREC_FP32 && (K % 4) == 0 && N >= 16 && K >= 16;
}

struct TileWgradJob { const void *tA, *tB; float* part; float *dW, *db; int N, K; const int32_t* s32; int S; int hd, hdp, H, D; };
// up to four products over the same M rows; part_i: tile_wgrad_part_floats(N_i, K_i) floats.  cs: 0..2 column-sum jobs
// (x [M rows][N] contiguous -> out1[0..n1), out2[0..N-n1)) carried by the reduce launch.
struct TileColsumJob { const float* x; int M, N, n1; float *out1, *out2; };
int launch_tile_wgrad(long M, int njobs, const TileWgradJob* jobs, const void* ones, int ncs, const TileColsumJob* cs,
                      hipStream_t st, const int32_t* s32) {
  if (njobs < 1 || njobs > 4) return fail(RD_EINVAL, "tile_wgrad: 1..4 jobs");
  TwArgs a{};
  a.n = njobs; a.S = cdiv((int)M, 32); a.ones = (const __bf16*)ones; a.s32 = s32;
  int wg = 0, q = 0;
  for (int i = 0; i < njobs; ++i) {
    TwProb& P = a.p[i];
    const TileWgradJob& j = jobs[i];
    P.tA = (const __bf16*)j.tA; P.tB = (const __bf16*)j.tB; P.part = j.part; P.dW = j.dW; P.db = j.db;
    P.N = j.N; P.K = j.K; P.nctA = cdiv(j.N, 16); P.nctB = cdiv(j.K, 16);
    P.nbk = cdiv(P.nctB + 1, TW_NB); P.nmem = cdiv(P.nctA, TW_NA) * P.nbk; P.ldp = 16 * (P.nctB + 1);
    P.wg0 = wg; wg += 8 * P.nmem;
    P.q0 = q; P.nq = j.N * (P.ldp >> 2); q += (P.nq + 511) / 512 * 512;
    P.s32x = j.s32; P.Sx = j.S; P.hd = j.hd; P.hdp = j.hdp; P.H = j.H; P.D = j.D;
  }
  // a reduce launch of an earlier call may still be reading partials / LayerNorm partial matrices on the side branch: this call
  // rewrites workspace of the same kind (callers with one workspace per layer never wait here for long)
  int rcj = side_join(st);
  if (rcj) return rcj;
  if (precision() == RD_PREC_BF16) {
    RD_LDS_ATTR(k_twg<true>, TW_LDS);
    hipLaunchKernelGGL(k_twg<true>, dim3(wg), dim3(TW_THR), TW_LDS, st, a);
  } else {
    RD_LDS_ATTR(k_twg<false>, TW_LDS);
    hipLaunchKernelGGL(k_twg<false>, dim3(wg), dim3(TW_THR), TW_LDS, st, a);
  }
  int rc = check_launch("k_twg");
  if (rc) return rc;
  if (ncs < 0 || ncs > 2) return fail(RD_EINVAL, "tile_wgrad: 0..2 column-sum jobs");
  a.ncs = ncs; a.nblk_w = cdiv(2 * q, TWR_THR);
  int ncb = 0;
  for (int i = 0; i < ncs; ++i) {
    a.cs[i].x = cs[i].x; a.cs[i].M = cs[i].M; a.cs[i].N = cs[i].N; a.cs[i].n1 = cs[i].n1; a.cs[i].out1 = cs[i].out1; a.cs[i].out2 = cs[i].out2;
    ncb += cdiv(cs[i].N, 64);
  }
  // the reduce only produces parameter gradients: it goes to the side branch (rd_common.h side_fork) when one is registered
  hipLaunchKernelGGL(k_twg_reduce, dim3(a.nblk_w + ncb), dim3(TWR_THR), 0, side_fork(st), a);
  return check_launch("k_twg_reduce");
}

}  // namespace rd
