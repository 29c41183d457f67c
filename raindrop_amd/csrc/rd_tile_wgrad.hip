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
#include "rd_trailing.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TILE = 512;                          // bf16 elements of one tile part (64 lanes x 8)
constexpr int TW_THR = 256, TW_NA = 5, TW_NB = 6, TW_SLICES = 8;
constexpr int TW_LDC = 16 * TW_NB + 4;             // fp32 row stride of a wave's block in LDS
constexpr int TW_LDS = 4 * 16 * TW_NA * TW_LDC * 4;   // 128 KB: the four waves' blocks for the final sum

#ifndef RD_TWG_SPREAD
#define RD_TWG_SPREAD 1
#endif
struct Frag { bf16x8 ah[TW_NA], al[TW_NA], bh[TW_NB], bl[TW_NB]; };

// ONE: the single-product arithmetic mode (RD_PREC_BF16: hi * hi only).  The lo halves of the tiles are neither read nor multiplied --
// half the operand bytes of a kernel that is a pure stream, one third of its MFMAs.
template <bool ONE>
__global__ __launch_bounds__(TW_THR) void k_twg(TwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  RD_TOUCH_CODE(ONE ? RD_TL_TWG_ONE : RD_TL_TWG);                 // own code -> L2 (rd_common.h; 8 860 / 10 592 bytes: all of it)
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

  // Operand addressing: every tile part is read at [wave-uniform base] + lane * 16 bytes.  The bases are SCALAR (byte offsets from the
  // zero tile, one per operand tile of the block, + chunk * stride), so a load is `global_load_dwordx4 v, v_lane16, s[base]` with no
  // vector address arithmetic.  The B column tile with index nctB is the constant "ones" tile (stride 0).  Tiles beyond an operand's
  // range map to a valid tile; their products are computed and dropped at the store.
  // Chunk i of this wave is slice-local index wave + 4 i; i >= nst is a GHOST: its A operands come from the zero tile -- selected
  // by an AND with an opaque all-ones / zero mask, not a ?: (round 2-5a: the compiler turned the select of the two addresses into
  // a branch around the 64-bit multiply, five per chunk, which cut every chunk's loads and MFMAs into separate basic blocks: the
  // wave issued its 22 loads in one burst -- stalling on the address unit's queue while the other three waves' bursts drained --
  // and only then its 90 MFMAs; address unit (22 KB per wave and chunk at 64 B/clk = 1408 cycles per round of the four waves) and
  // matrix cores (90 x 16 = 1440) took turns instead of overlapping).  Measured in the step, same box, two traces each: k_twg
  // 21.7 / 21.1 -> 20.7 / 20.5 us, k_dw 11.10 / 11.03 -> 10.56 / 10.88 (profiles/r05_spread_loads_ab.txt) -- less than the model
  // promised: at 74.7 MB per launch the stream is within ~15 % of what HBM delivers to a kernel of this length.
  const unsigned lane16 = (unsigned)lane * 16u;
  const char* zb = reinterpret_cast<const char*>(a.ones + TILE);                  // [ones hi][zeros][zeros]: the zero tile
  const long stepA = (long)nctA * 2 * TILE * 2, stepB = (long)nctB * 2 * TILE * 2;  // bytes per chunk
  long dA[TW_NA], dB[TW_NB], sB[TW_NB];
#pragma unroll
  for (int i = 0; i < TW_NA; ++i)
    dA[i] = reinterpret_cast<const char*>(P.tA + (size_t)min(TW_NA * bn + i, nctA - 1) * 2 * TILE) - zb;
#pragma unroll
  for (int i = 0; i < TW_NB; ++i) {
    const int kt = TW_NB * bk + i;
    if (kt == nctB) { dB[i] = reinterpret_cast<const char*>(a.ones) - zb; sB[i] = 0; }
    else { dB[i] = reinterpret_cast<const char*>(P.tB + (size_t)min(kt, nctB - 1) * 2 * TILE) - zb; sB[i] = stepB; }
  }
  const int nst = ntile > wave ? (ntile - wave + 3) / 4 : 0;
  auto load = [&](Frag& f, int i) {
    const bool ghost = i >= nst;
    int live32 = __builtin_amdgcn_readfirstlane(ghost ? 0 : -1);
    asm volatile("" : "+s"(live32));                                    // opaque: stays an AND (see above)
    const long live = (long)live32;
    const long s = ghost ? 0 : sl + TW_SLICES * (wave + 4 * i);
#pragma unroll
    for (int t = 0; t < TW_NA; ++t) {
      const char* qa = zb + ((dA[t] + s * stepA) & live) + lane16;
      f.ah[t] = *reinterpret_cast<const bf16x8*>(qa);
      if (!ONE) f.al[t] = *reinterpret_cast<const bf16x8*>(qa + TILE * 2);
    }
#pragma unroll
    for (int t = 0; t < TW_NB; ++t) {
      const char* qb = zb + (dB[t] + s * sB[t]) + lane16;
      f.bh[t] = *reinterpret_cast<const bf16x8*>(qb);
      if (!ONE) f.bl[t] = *reinterpret_cast<const bf16x8*>(qb + TILE * 2);
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
  // One stage = the loads of chunk i + 2 and the products of chunk i, INTERLEAVED: one tile-part load per NM / NL MFMAs, so the
  // wave's requests reach the address unit spread over the stage and its own MFMAs run under the other waves' loads.
  constexpr int NL = (ONE ? 1 : 2) * (TW_NA + TW_NB), NM = (ONE ? 1 : 3) * TW_NA * TW_NB, PER = NM / NL;
  auto spread = [&]() {
#if RD_TWG_SPREAD
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);               // one VMEM read
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);             // PER MFMAs
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NL, 0);
#endif
  };
  Frag f0, f1, f2;
  load(f0, 0); load(f1, 1);
  for (int it = 0; it < nmax; it += 3) {
    load(f2, it + 2); mma(f0); spread();
    load(f0, it + 3); mma(f1); spread();
    load(f1, it + 4); mma(f2); spread();
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

__global__ __launch_bounds__(TWR_THR) void k_twg_reduce(TwArgs a) {
  __shared__ float red[16][64];
  twg_reduce_body(a, (int)blockIdx.x, red);
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
  // the reduce only produces parameter gradients.  Deferred mode (rd_set_defer_trailing): it is parked and rides in the next
  // backward chain launch's idle workgroups (rd_trailing.h); else it goes to the side branch when one is registered, else here.
  if (trailing_deferred()) {
    RiderArgs r{};
    r.kind = RIDER_TWG; r.nblocks = a.nblk_w + ncb; r.tw = a;
    return trailing_park(r, st);
  }
  hipLaunchKernelGGL(k_twg_reduce, dim3(a.nblk_w + ncb), dim3(TWR_THR), 0, side_fork(st), a);
  return check_launch("k_twg_reduce");
}

int launch_twg_reduce_standalone(const TwArgs& a, int nblocks, hipStream_t st) {
  hipLaunchKernelGGL(k_twg_reduce, dim3(nblocks), dim3(TWR_THR), 0, st, a);
  return check_launch("k_twg_reduce");
}

}  // namespace rd
