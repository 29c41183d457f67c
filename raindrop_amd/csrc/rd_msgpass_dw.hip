// rd_msgpass_dw.hip -- weight gradients of the two Observation_progation layers on the fused K1 path:
//   dW_l[n, k] = sum over the B*F graph rows r of dZ_l[r, n] * In_l[r, k],   db_l[n] = sum_r dZ_l[r, n]
// (backward of `lin_value`, code/Ob_propagation.py:200 via :42; In_1 = X, In_2 = Y1).
//
// Both operands arrive as split-bf16 ROW tiles (rd_k1_layout.h) written by the forward / backward kernels of
// rd_msgpass_fused.hip: every 1-KB tile is one MFMA operand fragment, so this kernel is a pure stream:
// one 16-byte load per lane and tile part -> v_mfma_f32_16x16x32_bf16, no conversion, no transposition, no LDS in
// the main loop, no padding in the reduction (272 tiles of 32 rows = 8704 rows at P19).
//
// Decomposition: the [K x (K+16)] output of a layer (the extra 16 columns hold the bias gradient in column 0: a
// constant "ones" operand tile) is cut into 4 x 4-tile blocks; the S reduction tiles into `nslice` interleaved slices;
// workgroup = (layer, block, slice).  Its 4 waves each own the WHOLE 4 x 4 block (64 accumulator registers) and
// take every 4th reduction tile, double-buffered in registers: no barrier and no LDS until the final in-workgroup
// sum (wave order: deterministic).  What bounds it: every operand tile is read by the 4 workgroups of a block row /
// column, 16 KB per reduction tile and workgroup through the address unit (16 cycles per 1-KB wave-load) ~ 8.7 k
// cycles, against 6.5 k cycles of MFMA per wave.  The 16 workgroups of one (layer, slice) group land on the same XCD
// (block % 8) and walk the reduction in step, so each tile crosses the fabric once per XCD.  Small blocks buy few
// slices: 8 partials per layer (3.9 MB) instead of 32.  k_dw_reduce sums the partials in slice order and also
// folds the per-sample dR_u partials of the backward kernel.
#include "rd_common.h"
#include "rd_trailing.h"
#include "rd_k1_layout.h"
#include "rd_plan.h"

namespace rd {
namespace {

using k1::TILE;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// RD_DW_WAVES: 4 = one wave per SIMD, three register buffers (two tiles in flight under the third's products; 312 registers);
// 8 = two waves per SIMD at <= 256 registers, two buffers each (the same tiles in flight per SIMD; one wave's requests issue under
// the other's products).  Measured in round 5 (profiles/r05_kdw_waves_ab.txt).
#ifndef RD_DW_SPREAD
#define RD_DW_SPREAD 1
#endif
#ifndef RD_DW_WAVES
#define RD_DW_WAVES 4
#endif
constexpr int DW_NW = RD_DW_WAVES;
constexpr int DW_THR = 64 * DW_NW;
constexpr int DW_LDC = 68;                         // fp32 row stride of a wave's 64 x 64 block in LDS
constexpr int DW_LDS = DW_NW * 64 * DW_LDC * 4;    // 68 KB (4 waves) / 136 KB (8): the waves' blocks for the final sum

struct DwArgs {
  const __bf16 *tpX, *tpY1, *tpD1, *tpD2, *ones;
  float* part;                    // [nslice][2][K][ldp]
  int S, K, nct, nbn, nbk, nslice, ldp;
  const int32_t* plan; int B, q, rem, per;     // token plan (or null) and the row-tile grouping of rd_k1_layout.h
};

struct Frag { bf16x8 ah[4], al[4], bh[4], bl[4]; };

__global__ __launch_bounds__(DW_THR) void k_dw(DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  RD_TOUCH_CODE(RD_TL_DW);                                   // own code -> L2 (rd_common.h; all of the 7 536-byte kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- which (layer, slice, block) ----
  const int nmem = a.nbn * a.nbk;
  const int xcd = blockIdx.x & 7, y = blockIdx.x >> 3;
  const int mi = y % nmem, gidx = (y / nmem) * 8 + xcd;
  if (gidx >= 2 * a.nslice) return;
  const int layer = gidx / a.nslice, sl = gidx - layer * a.nslice;
  const int bn = mi / a.nbk, bk = mi - bn * a.nbk;
  const int nct = a.nct;
  // slice sl = reduction tiles sl, sl + nslice, sl + 2 nslice, ...: with nslice == 8 these are the samples b = sl (mod 8),
  // i.e. the row tiles the backward kernel's workgroups on XCD sl wrote a moment ago -- and this group runs on XCD sl
  const int nsl = a.nslice;
  // Reduction tiles this block needs.  Layer 2 with a token plan: dZ2's columns at a sample's padded steps are exactly zero
  // (and not exported), so the 64-column block bn (steps 16 bn ..) sums only over the samples longer than 16 bn -- the row
  // tiles are stored by RANK (descending length), i.e. over a prefix: nr samples -> nr*q main tiles + ceil(nr / per) leftover
  // tiles (a leftover tile holds `per` consecutive ranks; its shorter members wrote zeros up to the longest one's block).
  int nmain = a.B * a.q, ntot = a.S;
  if (a.plan && layer == 1) {
    const int t0 = min(16 * bn, __builtin_amdgcn_readfirstlane(a.plan[plan::I_T]));
    const int nr = __builtin_amdgcn_readfirstlane(a.plan[plan::cnt_base(a.B) + t0]);
    nmain = nr * a.q;
    ntot = nmain + (a.rem ? (nr + a.per - 1) / a.per : 0);
  }
  const int lbase = a.B * a.q - nmain;                                // reduction index i >= nmain -> tile i + lbase
  const int ntile = ntot > sl ? (ntot - sl + nsl - 1) / nsl : 0;      // tiles of this slice
  const int s0 = 0, s1 = ntile;                                       // wave tiling below runs over slice-local indices
  const __bf16* tA = layer ? a.tpD2 : a.tpD1;
  const __bf16* tB = layer ? a.tpY1 : a.tpX;

  // Operand addressing (as rd_tile_wgrad.hip, where the measurement is described): every tile part is read at [wave-uniform base] +
  // lane * 16 bytes, the bases are scalar byte offsets from the zero tile + tile * stride.  The k-tile with index nct is the
  // constant "ones" tile (column 0 = 1: its output column is the bias gradient), same source every step.  Tiles beyond the
  // operand's range map to a valid tile; their products are computed and dropped.
  // Tile i of this wave is reduction tile s0 + wave + DW_NW i; i >= nst is a GHOST tile: its A operands come from the zero tile
  // (an AND with an opaque mask, not a select of two addresses: the compiler made five branches per tile of that, which kept the
  // tile's loads and its MFMAs in separate basic blocks -- a burst of 16 loads, then 48 MFMAs, instead of both units busy), so
  // every wave runs the same branch-free trip count and the compiler's s_waitcnt bookkeeping stays exact.
  const unsigned lane16 = (unsigned)lane * 16u;
  const char* zb = reinterpret_cast<const char*>(a.ones + TILE);        // [ones hi][zeros][zeros]: the zero tile
  const long step = (long)nct * 2 * TILE * 2;                           // bytes per reduction tile
  long dA[4], dB[4], sB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dA[i] = reinterpret_cast<const char*>(tA + (size_t)min(4 * bn + i, nct - 1) * 2 * TILE) - zb;
    const int kt = 4 * bk + i;
    if (kt == nct) { dB[i] = reinterpret_cast<const char*>(a.ones) - zb; sB[i] = 0; }
    else { dB[i] = reinterpret_cast<const char*>(tB + (size_t)min(kt, nct - 1) * 2 * TILE) - zb; sB[i] = step; }
  }
  const int nst = s1 - s0 > wave ? (s1 - s0 - wave + DW_NW - 1) / DW_NW : 0;      // this wave's tile count
  auto load = [&](Frag& f, int i) {
    const bool ghost = i >= nst;
    int live32 = __builtin_amdgcn_readfirstlane(ghost ? 0 : -1);
    asm volatile("" : "+s"(live32));
    const long live = (long)live32;
    const int ri = sl + nsl * (wave + DW_NW * i);                     // reduction index -> tile (ghosts: tile 0)
    const long s = ghost ? 0 : (long)(ri < nmain ? ri : ri + lbase);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const char* qa = zb + ((dA[t] + s * step) & live) + lane16;
      f.ah[t] = *reinterpret_cast<const bf16x8*>(qa);
      f.al[t] = *reinterpret_cast<const bf16x8*>(qa + TILE * 2);
      const char* qb = zb + (dB[t] + s * sB[t]) + lane16;
      f.bh[t] = *reinterpret_cast<const bf16x8*>(qb);
      f.bl[t] = *reinterpret_cast<const bf16x8*>(qb + TILE * 2);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki) acc[ni][ki] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
        acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.al[ni], f.bh[ki], acc[ni][ki], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
        acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[ni], f.bl[ki], acc[ni][ki], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
        acc[ni][ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.ah[ni], f.bh[ki], acc[ni][ki], 0, 0, 0);
  };

  const int nmax = (s1 - s0 + DW_NW - 1) / DW_NW;                      // tile count of wave 0 (the largest)
  if constexpr (DW_NW == 4) {
    // ring of three register buffers, two tiles (32 KB per wave) in flight under the MFMAs of the third
    // a stage = the 16 loads of tile i + 2 and the 48 MFMAs of tile i, interleaved one load per three MFMAs
    auto spread = [&]() {
#if RD_DW_SPREAD
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // one VMEM read
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);             // three MFMAs
      }
#endif
    };
    Frag f0, f1, f2;
    load(f0, 0); load(f1, 1);
    for (int it = 0; it < nmax; it += 3) {
      load(f2, it + 2); mma(f0); spread();
      load(f0, it + 3); mma(f1); spread();
      load(f1, it + 4); mma(f2); spread();
    }
  } else {
    // two buffers per wave, two waves per SIMD
    Frag f0, f1;
    load(f0, 0);
    for (int it = 0; it < nmax; it += 2) {
      load(f1, it + 1); mma(f0);
      load(f0, it + 2); mma(f1);
    }
  }

  // ---- in-workgroup sum of the waves' blocks (fixed wave order) -> partial ----
  float* Cs = reinterpret_cast<float*>(dsm) + (size_t)wave * 64 * DW_LDC;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        Cs[(16 * ni + 4 * (lane >> 4) + i) * DW_LDC + 16 * ki + (lane & 15)] = acc[ni][ki][i];
  __syncthreads();
  const float* C0 = reinterpret_cast<const float*>(dsm);
  const int nrows = min(64, a.K - 64 * bn), ncols = min(64, a.ldp - 64 * bk);   // multiples of 16
  float* out = a.part + ((size_t)(sl * 2 + layer) * a.K + 64 * bn) * a.ldp + 64 * bk;
  const int qpr = ncols >> 2;
  for (int e = tid; e < nrows * qpr; e += DW_THR) {
    const int r = e / qpr, c4 = e - r * qpr;
    const float* q = C0 + r * DW_LDC + 4 * c4;
    float4 v = *reinterpret_cast<const float4*>(q);
#pragma unroll
    for (int w = 1; w < DW_NW; ++w) {
      const float4 u = *reinterpret_cast<const float4*>(q + (size_t)w * 64 * DW_LDC);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)r * a.ldp + 4 * c4) = v;
  }
}

// dW_l, db_l = sum over slices (in slice order); dR_u = sum over samples of the backward kernel's partials.
struct RedArgs {
  const float* part; int nslice, K, ldp, nct;
  float *dW1, *db1, *dW2, *db2;
  const float* rupart; float* dRu; int B, Fd;
  int nblk_dw;
};

// 1024-thread workgroups (round 4): block b < ceil(nblk_dw / 4) = FOUR of the former 256-thread blocks of the slice sum (same
// thread -> element map, same order: bit-identical), then the dR_u blocks (their first 256 threads), then RIDERS (rd_trailing.h):
// the parked slice reduce of encoder layer 0's weight-gradient stream, which nothing picked up, runs here instead of as a launch
// of its own.
__global__ __launch_bounds__(1024) void k_dw_reduce(RedArgs a, RiderArgs rider, int nmain) {
  __shared__ float rlds[16][64];
  RD_TOUCH_CODE(RD_TL_DW_REDUCE);                                   // own code -> L2, the rider's body included (all of the 10 096-byte kernel)
  if ((int)blockIdx.x >= nmain) { rider_body(rider, (int)blockIdx.x - nmain, reinterpret_cast<unsigned char*>(rlds)); return; }
  const int nb4 = (a.nblk_dw + 3) >> 2;
  const int tid = threadIdx.x & 255;
  const int vblock = (int)blockIdx.x < nb4 ? (int)blockIdx.x * 4 + (int)(threadIdx.x >> 8) : a.nblk_dw + ((int)blockIdx.x - nb4);
  if ((int)blockIdx.x < nb4) {
    if (vblock >= a.nblk_dw) return;
    // 4 neighbouring lanes share one output quad: lane p sums slices [p*chunk, (p+1)*chunk) in order, then the four
    // sub-sums are added in lane order (fixed order: deterministic); 4x the loads in flight of one thread per quad
    const int qpr = a.ldp >> 2;
    const long e = ((long)vblock * 256 + tid) >> 2;
    const int p4 = tid & 3;
    const bool live = e < (long)2 * a.K * qpr;
    const long ec = live ? e : 0;
    const int layer = (int)(ec / ((long)a.K * qpr));
    const int rem = (int)(ec - (long)layer * a.K * qpr);
    const int n = rem / qpr, c4 = rem - n * qpr;
    const int k = 4 * c4;
    const bool is_w = k < a.K, is_b = k == 16 * a.nct;
    const size_t stride = (size_t)2 * a.K * a.ldp;
    const float* p = a.part + ((size_t)layer * a.K + n) * a.ldp + k;
    const int chunk = (a.nslice + 3) >> 2;
    const int sl0 = p4 * chunk, sl1 = min(a.nslice, sl0 + chunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live && (is_w || is_b)) {
      int sl = sl0;
      for (; sl + 8 <= sl1; sl += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(sl + u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; sl < sl1; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)sl * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    float4 t = s;
#pragma unroll
    for (int o = 1; o < 4; ++o) {                       // lane p4 == 0 accumulates sub-sums 1, 2, 3 in that order
      const float4 r = make_float4(__shfl_down(s.x, o, 4), __shfl_down(s.y, o, 4), __shfl_down(s.z, o, 4), __shfl_down(s.w, o, 4));
      t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
    }
    if (live && p4 == 0) {
      if (is_w) *reinterpret_cast<float4*>((layer ? a.dW2 : a.dW1) + (size_t)n * a.K + k) = t;
      else if (is_b) (layer ? a.db2 : a.db1)[n] = t.x;
    }
    return;
  }
  // ---- dR_u: workgroup handles 32 columns; 8 groups of (its first 256) threads split the samples, combined in fixed order ----
  float (*red)[32] = reinterpret_cast<float (*)[32]>(rlds);
  const bool act = threadIdx.x < 256;
  const int c0 = (vblock - a.nblk_dw) * 32;
  const int pgrp = tid >> 5, c = tid & 31;
  float v = 0.f;
  if (act && c0 + c < a.Fd) {
    const float* rp = a.rupart + c0 + c;
    int b = pgrp;
    for (; b + 56 < a.B; b += 64) {                    // 8 independent loads in flight, added in order
      float w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = rp[(size_t)(b + 8 * u) * a.Fd];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += w[u];
    }
    for (; b < a.B; b += 8) v += rp[(size_t)b * a.Fd];
  }
  if (act) red[pgrp][c] = v;
  __syncthreads();
  if (threadIdx.x < 32 && c0 + (int)threadIdx.x < a.Fd) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][threadIdx.x];
    a.dRu[c0 + threadIdx.x] = s;
  }
}

}  // namespace

int fused_dw(const k1::Layout& L, const k1::DwPlan& P, const void* tpX, const void* tpY1, const void* tpD1,
             const void* tpD2, const void* ones, float* part, const float* rupart, float* dW1, float* db1, float* dW2, float* db2,
             float* dRu, hipStream_t st) {
  DwArgs a{};
  a.tpX = (const __bf16*)tpX; a.tpY1 = (const __bf16*)tpY1; a.tpD1 = (const __bf16*)tpD1; a.tpD2 = (const __bf16*)tpD2;
  a.ones = (const __bf16*)ones;
  a.part = part; a.S = L.S; a.K = L.K; a.nct = L.nct; a.nbn = P.nbn; a.nbk = P.nbk; a.nslice = P.nslice; a.ldp = P.ldp;
  a.plan = token_plan(); a.B = L.B; a.q = L.q; a.rem = L.rem; a.per = L.per;
  const int nmem = P.nbn * P.nbk, ngroups = 2 * P.nslice;
  const int grid = 8 * nmem * cdiv(ngroups, 8);
  RD_LDS_ATTR(k_dw, DW_LDS);
  hipLaunchKernelGGL(k_dw, dim3(grid), dim3(DW_THR), DW_LDS, st, a);
  int rc = check_launch("k_dw");
  if (rc) return rc;
  RedArgs r{};
  r.part = part; r.nslice = P.nslice; r.K = L.K; r.ldp = P.ldp; r.nct = L.nct;
  r.dW1 = dW1; r.db1 = db1; r.dW2 = dW2; r.db2 = db2;
  r.rupart = rupart; r.dRu = dRu; r.B = L.B; r.Fd = L.F * 4;
  r.nblk_dw = (int)(((long)2 * L.K * (P.ldp >> 2) * 4 + 255) / 256);
  // a parked slice reduce (encoder layer 0's: rd_trailing.h) rides here; anything else parked runs on its own first
  RiderArgs rider = trailing_take();
  if (rider.kind != RIDER_NONE && rider.kind != RIDER_TWG) { if ((rc = trailing_launch(rider, st))) return rc; rider.kind = RIDER_NONE; }
  const int nmain = cdiv(r.nblk_dw, 4) + cdiv(r.Fd, 32);
  hipLaunchKernelGGL(k_dw_reduce, dim3(nmain + (rider.kind != RIDER_NONE ? rider.nblocks : 0)), dim3(1024), 0, st, r, rider, nmain);
  return check_launch("k_dw_reduce");
}

}  // namespace rd
