// rd_rowgemm.hip -- "row-block x all columns" dense layer for the temporal encoder.
//
// The encoder's products are tall and skinny (M = T*B = 15360 tokens, N,K <= 456): a 64x64-tiled
// GEMM re-reads its operands 4-8x below L1 and is bound by that traffic.  Here ONE
// workgroup owns 64 complete rows: the activation rows are read from HBM exactly once, split to
// bf16 hi/lo planes in LDS, and every wave streams its slice of the PRE-SPLIT weight planes from L2
// straight into registers as MFMA B operands -- the same machinery as the fused message-passing
// kernel (rd_msgpass_fused.hip), so the weight stream is the only repeated traffic.
//
//   C[m, n] = epilogue( sum_k A[m, k] * Wp[n, k] ),   Wp = weight planes [NP][KP] (hi, lo), rows = n
// forward  (x W^T):  Wp = split(W)      [N rows, K cols]
// dgrad    (dy W):   Wp = split(W^T)    [K rows, N cols]
// Split-bf16 arithmetic (hi*hi + hi*lo + lo*hi, fp32 accumulate) as everywhere else.
#include "rd_common.h"
#include <type_traits>
#include "rd_plan.h"
#include "rd_rng.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// workgroup: WV wavefronts (template parameter, 8 or 16) x RG_ROWS rows (64; 32 for the long reduction K = 3D)
// per kernel instance: NJ column tiles per wave and round, RG_CPR = 8 * NJ * 16 output columns per round, stage row stride RG_CPR + 4

static unsigned long long* g_rg_stamps = nullptr;   // debug only (tools/rowgemm_timing.py); passed as a kernel argument
#define RGSTAMP(i)                                                                          \
  do {                                                                                      \
    if (a.stamps && blockIdx.x < 8 && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (i)] = clock64();       \
  } while (0)

struct RowGemmArgs {
  const float* A; long lda;                         // [M, K] fp32
  const __bf16* Wh; const __bf16* Wl;               // Wh: native operand tiles [NP16/16][KP/32][hi,lo][64][8]; Wl unused
  float* C; long ldc;
  int M, N, K, KP;
  const float* bias; int relu;
  const float* posmask; long pm_ld; float cscale;
  const float* residual; long res_ld;
  float drop_p; uint64_t drop_seed; uint32_t drop_site; const uint64_t* seed_cell;
  unsigned long long* stamps;
  int one_product;                                  // RD_PREC_BF16
  // LayerNorm epilogue (N <= one round of columns): s = residual + dropout(A W^T + bias) -> ln_s (saved for the backward),
  // C = LayerNorm(s) * ln_g + ln_b, (mean, rstd) -> ln_stats.  Replaces the separate add+LayerNorm kernel of the encoder
  // layer (rd_temporal.hip k_add_ln_fwd_v: same lane <-> column assignment, same reduction order, same dropout quads).
  const float* ln_g; const float* ln_b; float* ln_s; float* ln_stats;
  // Tile export (rd_tile_wgrad.hip): the split A planes, transposed into MFMA operand tiles whose reduction index is the
  // ROW -- [chunk of 32 rows][column tile of 16][hi, lo][64][8] -- the operand format of the weight-gradient stream.
  __bf16* xt; int xt_nct;
  // LayerNorm-backward prologue (template flag LNB): the A operand is not read but COMPUTED -- A = dropout-masked gradient of the
  // LayerNorm input, from dy (lnb_dy), the saved pre-norm sum (lnb_s), (mean, rstd) (lnb_stats) and gamma (lnb_g); the
  // unmasked gradient goes to lnb_ds (the residual branch), per-workgroup dgamma | dbeta partials to lnb_part
  // [workgroup][2K].  Same arithmetic and lane <-> column assignment as k_ln_bwd_v (rd_temporal.hip), which it replaces
  // together with the write + re-read of its second output.
  const float *lnb_dy, *lnb_s, *lnb_stats, *lnb_g; float *lnb_ds, *lnb_part; float lnb_p; uint32_t lnb_site; uint64_t lnb_seed;
  // Live row count on the device (token plan, rd_plan.h: plan[0]) or null.  The grid is sized for the padded M; a workgroup
  // whose rows are all beyond *mlive exits (after zeroing its LayerNorm partial), the one that straddles it treats the rest as the tail.
  const int32_t* mlive;
};

// one weight matrix -> hi/lo planes; transpose != 0 writes split(W^T): rows k, cols n
struct SplitJob { const float* W; int N, K, transpose; __bf16* hi; __bf16* lo; int rows, cols_p; };
constexpr int WS_MAXJOBS = 48, WS_MAXONES = 4;
struct SplitJobs { SplitJob j[WS_MAXJOBS]; int n; __bf16* ones[WS_MAXONES];   // ones: constant tiles of the weight-gradient streams (or null)
                   // optional: the step's token plan (rd_plan.h) by the workgroup (0, n) of the same launch (rd_step_begin)
                   const int64_t* plan_lengths; int32_t* plan_out; int plan_B, plan_T, plan_first; uint64_t* seed_cell; uint64_t seed_delta;
                   // optional: the optimizer's device step state (rd_set_adam_state), advanced once per launch next to the seed bump
                   double* adam_state; float adam_b1, adam_b2; };

// Output: NATIVE MFMA operand tiles [ntile = rows/16][kc = cols_p/32][hi, lo][64 lanes][8] (one contiguous kilobyte per
// wave-load; rd_k1_layout.h has the measurement: 61 B/clk/CU against 16 B/clk for a row-major plane).  `hi` is the
// base of the tile array, `lo` is unused (kept for the job layout).  One workgroup row per job; a wave converts one
// (ntile, kc) tile per iteration.
__global__ __launch_bounds__(256) void k_wsplit(SplitJobs jobs) {
  RD_TOUCH_CODE_FIRST(RD_TL_WSPLIT, blockIdx.y * gridDim.x + blockIdx.x, 64);   // own code -> L2 by the first workgroups (rd_common.h; 5 220-byte kernel)
  // rd_step_begin: one block row is the token plan (+ seed bump): its workgroups each take a share of the samples
  // (rd_plan.h: token_plan_part).  First row by default (RD_PLAN_FIRST=0: last, as in rounds 3-4 when the plan was ONE
  // workgroup's ~10-us chain that started behind ~2600 split workgroups: the launch took the sum of both).
  const int job = (jobs.plan_out && jobs.plan_first) ? (int)blockIdx.y - 1 : (int)blockIdx.y;
  if (job < 0 || job == jobs.n) {
    extern __shared__ __attribute__((aligned(16))) int wsm_plan[];
    if (blockIdx.x == 0 && threadIdx.x == 64 && jobs.adam_state) adam_state_advance(jobs.adam_state, jobs.adam_b1, jobs.adam_b2);
    plan::token_plan_part(jobs.plan_lengths, jobs.plan_out, jobs.plan_B, jobs.plan_T, jobs.seed_cell, jobs.seed_delta, wsm_plan, blockIdx.x, gridDim.x);
    return;
  }
  const SplitJob jb = jobs.j[job];
  const int ntile = jb.rows >> 4, nkc = jb.cols_p >> 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, G = lane >> 4;
  const int src_rows = jb.transpose ? jb.K : jb.N, src_cols = jb.transpose ? jb.N : jb.K;
  if (blockIdx.x == 0 && job == 0 && wave < WS_MAXONES && jobs.ones[wave]) {
    // [ones hi: column 0 = 1][zeros][zeros] (rd_tile_wgrad.hip: the B operand whose output column is the bias gradient)
    bf16x8 o, z;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = (__bf16)(c == 0 ? 1.f : 0.f); z[e] = (__bf16)0.f; }
    __bf16* on = jobs.ones[wave];
    *reinterpret_cast<bf16x8*>(on + lane * 8) = o;
    *reinterpret_cast<bf16x8*>(on + 512 + lane * 8) = z;
    *reinterpret_cast<bf16x8*>(on + 1024 + lane * 8) = z;
  }
  for (int t = blockIdx.x * 4 + wave; t < ntile * nkc; t += gridDim.x * 4) {
    const int j = t / nkc, kc = t - j * nkc;
    const int r = 16 * j + c, c0 = 32 * kc + 8 * G;          // plane row (free index), first plane column (reduction)
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = c0 + e;
      x[e] = (r < src_rows && cc < src_cols) ? (jb.transpose ? jb.W[(long)cc * jb.K + r] : jb.W[(long)r * jb.K + cc]) : 0.f;
    }
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) { h[e] = (__bf16)x[e]; l[e] = (__bf16)(x[e] - (float)h[e]); }
    __bf16* dst = jb.hi + ((size_t)t * 2) * 512 + lane * 8;
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + 512) = l;
  }
}

__device__ __forceinline__ float wave_sum64(float v) { return wave_sum64_dpp(v); }

template <int KC, int RG_NJ>
struct RPanel { bf16x8 h[RG_NJ][KC], l[RG_NJ][KC]; };

template <int KC, int RG_NJ, int RG_WAVES>
__device__ __forceinline__ void rg_load_panel(RPanel<KC, RG_NJ>& p, const __bf16* __restrict__ Wt, int tile0, int ntiles, int wave,
                                              int lane) {
#pragma unroll
  for (int jj = 0; jj < RG_NJ; ++jj) {
    const int j = tile0 + wave + RG_WAVES * jj;
    const __bf16* t = Wt + (size_t)(j < ntiles ? j : 0) * (KC * 2 * 512) + lane * 8;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      p.h[jj][kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 0) * 512);
      p.l[jj][kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 1) * 512);
    }
  }
}

// reduction steps [K0, K1) of the wave's tiles into a panel of KP = K1 - K0 (or more) steps: the K = 3D product holds HALF of its
// 15-step panel at a time (64 instead of 120 registers -> two workgroups per CU, see k_rowgemm)
template <int KC, int KP, int RG_NJ, int RG_WAVES, int K0, int K1>
__device__ __forceinline__ void rg_load_panel_part(RPanel<KP, RG_NJ>& p, const __bf16* __restrict__ Wt, int tile0, int ntiles, int wave,
                                                   int lane) {
#pragma unroll
  for (int jj = 0; jj < RG_NJ; ++jj) {
    const int j = tile0 + wave + RG_WAVES * jj;
    const __bf16* t = Wt + (size_t)(j < ntiles ? j : 0) * (KC * 2 * 512) + lane * 8;
#pragma unroll
    for (int kc = K0; kc < K1; ++kc) {
      p.h[jj][kc - K0] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 0) * 512);
      p.l[jj][kc - K0] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 1) * 512);
    }
  }
}

template <int KC, int RG_ROWS, int RG_NJ, bool LN, bool LNB, int WV>
__global__ __launch_bounds__(64 * WV) void k_rowgemm(RowGemmArgs a) {
  RD_TOUCH_CODE_X(RD_TL_ROWGEMM, blockIdx.x, 512);
  // SPLIT (the K = 3D = 456 input gradient of the QKV projection): the 15-step weight panel is 120 VGPRs -- with them the kernel
  // needs 178, ONE 8-wave workgroup per CU, and the 266 32-row workgroups of 8497 live rows run in two rounds on 256 CUs (the
  // second for 10 of them).  Holding half a panel at a time keeps the kernel under 128 registers: two workgroups per CU, one round.
  constexpr bool SPLIT = KC == 15 && !LN && !LNB;
  constexpr int KH = SPLIT ? 8 : KC;                 // steps of the resident (half) panel
  constexpr int RG_WAVES = WV, RG_THR = 64 * WV;
  constexpr int RT = RG_ROWS / 16, RG_CPR = RG_WAVES * RG_NJ * 16, RG_LDS_STAGE = RG_CPR + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  constexpr int KPc = KC * 32, LDA = KPc + 16;      // bf16 elements per A-plane row: + 32 bytes (conflict-free fragment reads, rd_encfuse.hip LDD)
  __bf16* Ah = reinterpret_cast<__bf16*>(rsm);
  __bf16* Al = Ah + RG_ROWS * LDA;
  float* stage = reinterpret_cast<float*>(Al + RG_ROWS * LDA);         // [64][260] fp32
  float* lnred = stage + RG_ROWS * RG_LDS_STAGE;                       // LNB only: [8 waves][2K] dgamma | dbeta partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * RG_ROWS;
  const int ntiles = (a.N + 15) >> 4;
  const int nrounds = (ntiles + RG_WAVES * RG_NJ - 1) / (RG_WAVES * RG_NJ);
  if (a.mlive) {                                     // uniform: scalar load
    a.M = min(a.M, __builtin_amdgcn_readfirstlane(*a.mlive));
    if (m0 >= a.M) {
      if constexpr (LNB)
        for (int i = tid; i < 2 * a.K; i += RG_THR) a.lnb_part[(long)blockIdx.x * 2 * a.K + i] = 0.f;
      return;
    }
  }

  RGSTAMP(0);
  uint64_t seed = a.drop_seed;
  // ---- A rows -> split planes (zero padded); 16-byte loads, K % 4 == 0.  64 rows x KPc/4 quads is exactly
  // KC quads per thread: all of them are requested first, THEN the weight panel (loads return in issue
  // order: the rows are needed now, the panel only at the first MFMA), then the rows are split into LDS
  // while the panel streams in.
  RPanel<KH, RG_NJ> pw;
  if constexpr (LNB) {
    // ---- LayerNorm backward per row (wave w: rows 8w .. 8w+7 of the block; lane l: columns 4l .. 4l+3) -> split planes ----
    constexpr int RPW = RG_ROWS / RG_WAVES;
    const int c = 4 * lane;
    const bool cok = c < a.K, cpl = c < KPc;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gg = zero4;
    if (cok) gg = *reinterpret_cast<const float4*>(a.lnb_g + c);
    float4 sraw[RPW], dvr[RPW]; float mean_r[RPW], rstd_r[RPW];
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
      const long row = m0 + wave * RPW + it;
      const bool rok = row < a.M;
      mean_r[it] = rok ? a.lnb_stats[2 * row] : 0.f; rstd_r[it] = rok ? a.lnb_stats[2 * row + 1] : 0.f;
      sraw[it] = zero4; dvr[it] = zero4;
      if (rok && cok) {
        sraw[it] = *reinterpret_cast<const float4*>(a.lnb_s + row * a.K + c);
        dvr[it] = *reinterpret_cast<const float4*>(a.lnb_dy + row * a.K + c);
      }
    }
    rg_load_panel_part<KC, KH, RG_NJ, RG_WAVES, 0, KH>(pw, a.Wh, 0, ntiles, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    uint64_t lseed = a.lnb_seed;
    if (a.seed_cell) { const uint64_t cv = load_uniform_u64(a.seed_cell); lseed += cv; seed += cv; }
    const float inv_keep_l = 1.0f / (1.0f - a.lnb_p);
    float4 ag = zero4, ab = zero4;
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
      const int rl = wave * RPW + it;
      const long row = m0 + rl;
      float4 dr = zero4;
      if (row < a.M) {                                   // wave-uniform
        const float mean = mean_r[it], rstd = rstd_r[it];
        const float4 dv = dvr[it];
        float4 xh = make_float4((sraw[it].x - mean) * rstd, (sraw[it].y - mean) * rstd, (sraw[it].z - mean) * rstd,
                                (sraw[it].w - mean) * rstd);
        if (!cok) xh = zero4;
        const float4 dg = make_float4(dv.x * gg.x, dv.y * gg.y, dv.z * gg.z, dv.w * gg.w);
        const float c1 = wave_sum64((dg.x + dg.y) + (dg.z + dg.w)) / a.K;
        const float c2 = wave_sum64((dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w)) / a.K;
        if (cok) {
          const float4 v = make_float4(rstd * (dg.x - c1 - xh.x * c2), rstd * (dg.y - c1 - xh.y * c2),
                                       rstd * (dg.z - c1 - xh.z * c2), rstd * (dg.w - c1 - xh.w * c2));
          *reinterpret_cast<float4*>(a.lnb_ds + row * a.K + c) = v;
          dr = v;
          if (a.lnb_p > 0.f) {
            const float4 u = uniform4(lseed, a.lnb_site, ((uint64_t)row * a.K + c) >> 2);
            dr.x *= u.x >= a.lnb_p ? inv_keep_l : 0.f; dr.y *= u.y >= a.lnb_p ? inv_keep_l : 0.f;
            dr.z *= u.z >= a.lnb_p ? inv_keep_l : 0.f; dr.w *= u.w >= a.lnb_p ? inv_keep_l : 0.f;
          }
          ag.x += dv.x * xh.x; ag.y += dv.y * xh.y; ag.z += dv.z * xh.z; ag.w += dv.w * xh.w;
          ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
        }
      }
      if (cpl) {
        bf16x4 h, l;
        const float x[4] = {dr.x, dr.y, dr.z, dr.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[q] = (__bf16)x[q]; l[q] = (__bf16)(x[q] - (float)h[q]); }
        *reinterpret_cast<bf16x4*>(Ah + rl * LDA + c) = h;
        *reinterpret_cast<bf16x4*>(Al + rl * LDA + c) = l;
      }
    }
    if (cok) {
      *reinterpret_cast<float4*>(lnred + wave * 2 * a.K + c) = ag;
      *reinterpret_cast<float4*>(lnred + wave * 2 * a.K + a.K + c) = ab;
    }
  } else {
    constexpr int kq = KPc / 4;                                        // float4 slots per row (incl. pad)
    constexpr int NIT = (RG_ROWS * kq + RG_THR - 1) / RG_THR;          // KC at 64 rows, ceil(KC / 2) at 32
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * RG_THR;
      const int r = i / kq, k = 4 * (i - r * kq);
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < RG_ROWS && m0 + r < a.M && k < a.K) v[it] = *reinterpret_cast<const float4*>(a.A + (long)(m0 + r) * a.lda + k);
    }
    rg_load_panel_part<KC, KH, RG_NJ, RG_WAVES, 0, KH>(pw, a.Wh, 0, ntiles, wave, lane);
    __builtin_amdgcn_sched_barrier(0);       // keep every request above the first use (the scheduler otherwise
                                             // waits for the rows before it has requested the panel)
    if (a.seed_cell) seed += load_uniform_u64(a.seed_cell);   // scalar path: not queued behind the panel
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * RG_THR;
      const int r = i / kq, k = 4 * (i - r * kq);
      bf16x4 h, l;
      const float x[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) { h[c] = (__bf16)x[c]; l[c] = (__bf16)(x[c] - (float)h[c]); }
      if (r < RG_ROWS) {
        *reinterpret_cast<bf16x4*>(Ah + r * LDA + k) = h;
        *reinterpret_cast<bf16x4*>(Al + r * LDA + k) = l;
      }
    }
  }
  RGSTAMP(1);
  lds_barrier();                       // LDS ordering only: do not drain the weight panel (rd_common.h)
  RGSTAMP(2);
  if constexpr (LNB) {                 // this workgroup's dgamma | dbeta partial: the 8 waves in fixed order
    const int K2 = 2 * a.K;
    for (int i = tid; i < K2; i += RG_THR)
    {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < RG_WAVES; ++w) v += lnred[w * K2 + i];
      a.lnb_part[(long)blockIdx.x * K2 + i] = v;
    }
  }
  if (a.xt) {
    // ds_read_b64_tr_b16 (tools/probe_tr16.hip): in a 16-lane group lane i passes the address of 4 consecutive shorts --
    // row i>>2, columns 4(i&3).. of a 4 x 16 block -- and receives column i of the block.  Two reads give lane
    // (column c = lane & 15, G = lane >> 4) the 8 rows 8G .. 8G+7 of its column: one tile part, stored as a contiguous KB.
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef short v8s __attribute__((ext_vector_type(8)));
    const int nct = a.xt_nct, ntp = (RG_ROWS / 32) * nct * 2;
    const int i16 = lane & 15, G = lane >> 4;
    for (int t = wave; t < ntp; t += RG_WAVES) {
      const int plane = t & 1, cj = t >> 1;
      const int c = cj / nct, j = cj - c * nct;
      if (m0 + 32 * c >= a.M) continue;
      const __bf16* src = (plane ? Al : Ah) + (32 * c + 8 * G + (i16 >> 2)) * LDA + 16 * j + 4 * (i16 & 3);
      const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(src));
      const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(src + 4 * LDA));
      const v8s o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<v8s*>(a.xt + (((size_t)(m0 / 32 + c) * nct + j) * 2 + plane) * 512 + lane * 8) = o;
    }
  }

  const float inv_keep = 1.0f / (1.0f - a.drop_p);
  const int aoff = (lane & 15) * LDA + 8 * (lane >> 4);
  for (int rd = 0; rd < nrounds; ++rd) {
    const int tile0 = rd * RG_WAVES * RG_NJ;
    f32x4 acc[RG_NJ][RT];
#pragma unroll
    for (int jj = 0; jj < RG_NJ; ++jj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[jj][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the single-product mode (RD_PREC_BF16) is decided OUTSIDE the reduction loop: a branch inside it cut every step into its own
    // basic block, and the A-fragment reads of step kc+1 could not be scheduled above the products of step kc
    auto steps = [&](auto k0_tag, auto k1_tag, auto three_tag) {
      constexpr int K0 = decltype(k0_tag)::value, K1 = decltype(k1_tag)::value;
      constexpr bool THREE = decltype(three_tag)::value;
#pragma unroll
      for (int kc = K0; kc < K1; ++kc) {
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDA + aoff + kc * 32);
          if (THREE) al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDA + aoff + kc * 32);
        }
        if (THREE) {
#pragma unroll
          for (int jj = 0; jj < RG_NJ; ++jj)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], pw.h[jj][kc - K0], acc[jj][rt], 0, 0, 0);
#pragma unroll
          for (int jj = 0; jj < RG_NJ; ++jj)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], pw.l[jj][kc - K0], acc[jj][rt], 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < RG_NJ; ++jj)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], pw.h[jj][kc - K0], acc[jj][rt], 0, 0, 0);
      }
    };
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, KH>; using IK = std::integral_constant<int, KC>;
    if (!a.one_product) steps(I0{}, IH{}, std::true_type{}); else steps(I0{}, IH{}, std::false_type{});
    if constexpr (SPLIT) {
      __builtin_amdgcn_sched_barrier(0);               // the second half goes into the registers the first half has released
      rg_load_panel_part<KC, KH, RG_NJ, RG_WAVES, KH, KC>(pw, a.Wh, tile0, ntiles, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      if (!a.one_product) steps(IH{}, IK{}, std::true_type{}); else steps(IH{}, IK{}, std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
    }
    if (rd == 0) RGSTAMP(3);
    // next round's weights stream while this round's epilogue runs
    if (rd + 1 < nrounds) rg_load_panel_part<KC, KH, RG_NJ, RG_WAVES, 0, KH>(pw, a.Wh, tile0 + RG_WAVES * RG_NJ, ntiles, wave, lane);
    // ---- accumulators -> stage tile (column = position inside this round's 256-column window) ----
#pragma unroll
    for (int jj = 0; jj < RG_NJ; ++jj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[(rt * 16 + 4 * (lane >> 4) + r) * RG_LDS_STAGE + (wave + RG_WAVES * jj) * 16 + (lane & 15)] = acc[jj][rt][r];
    if (rd == 0) RGSTAMP(4);
    lds_barrier();
    if (rd == 0) RGSTAMP(5);
    if constexpr (LN) {
      // ---- LayerNorm epilogue: wave w owns rows w, w + 8, ...; lane l owns columns 4l .. 4l+3 of the row
      constexpr int RPW = RG_ROWS / RG_WAVES;
      const int c = 4 * lane;
      const bool cok = c < a.N;
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xr[RPW];
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int m = m0 + wave + RG_WAVES * q;
        xr[q] = zero4;
        if (cok && m < a.M) xr[q] = *reinterpret_cast<const float4*>(a.residual + (long)m * a.res_ld + c);
      }
      float4 gg = zero4, bb = zero4, bs = zero4;       // (no `cond ? *p : zero4`: that becomes a pointer select + flat load)
      if (cok) {
        gg = *reinterpret_cast<const float4*>(a.ln_g + c);
        bb = *reinterpret_cast<const float4*>(a.ln_b + c);
        if (a.bias) bs = *reinterpret_cast<const float4*>(a.bias + c);
      }
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int rl = wave + RG_WAVES * q;
        const long m = m0 + rl;
        if (m >= a.M) continue;
        float4 t = zero4;
        if (cok) t = *reinterpret_cast<const float4*>(stage + rl * RG_LDS_STAGE + c);
        t.x += bs.x; t.y += bs.y; t.z += bs.z; t.w += bs.w;
        if (a.drop_p > 0.f) {
          const float4 u = uniform4(seed, a.drop_site, ((uint64_t)m * a.N + c) >> 2);
          t.x *= u.x >= a.drop_p ? inv_keep : 0.f; t.y *= u.y >= a.drop_p ? inv_keep : 0.f;
          t.z *= u.z >= a.drop_p ? inv_keep : 0.f; t.w *= u.w >= a.drop_p ? inv_keep : 0.f;
        }
        const float4 sv = make_float4(xr[q].x + t.x, xr[q].y + t.y, xr[q].z + t.z, xr[q].w + t.w);   // 0 beyond N
        if (cok) *reinterpret_cast<float4*>(a.ln_s + m * a.ldc + c) = sv;
        const float mean = wave_sum64((sv.x + sv.y) + (sv.z + sv.w)) / a.N;
        float4 d = make_float4(sv.x - mean, sv.y - mean, sv.z - mean, sv.w - mean);
        if (!cok) d = zero4;
        const float rstd = rsqrtf(wave_sum64((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) / a.N + 1e-5f);
        if (cok)
          *reinterpret_cast<float4*>(a.C + m * a.ldc + c) =
              make_float4(d.x * rstd * gg.x + bb.x, d.y * rstd * gg.y + bb.y, d.z * rstd * gg.z + bb.z, d.w * rstd * gg.w + bb.w);
        if (lane == 0) { a.ln_stats[2 * m] = mean; a.ln_stats[2 * m + 1] = rstd; }
      }
      RGSTAMP(6);
      break;                                                            // one round by construction
    } else {
    // ---- epilogue over the stage tile: thread = (row, 4 consecutive columns); all global reads first
    const int n_base = tile0 * 16;
    const int ncols = min(RG_CPR, a.N - n_base);                       // valid columns this round (multiple of 4)
    const int qpr = ncols >> 2;
    for (int e = tid; e < RG_ROWS * qpr; e += RG_THR) {
      const int rl = e / qpr, q = e - rl * qpr;
      const int m = m0 + rl, n = n_base + 4 * q;
      if (m >= a.M) continue;
      float4 pm = make_float4(1.f, 1.f, 1.f, 1.f), rs = make_float4(0.f, 0.f, 0.f, 0.f), bs = rs;
      if (a.posmask) pm = *reinterpret_cast<const float4*>(a.posmask + (long)m * a.pm_ld + n);
      if (a.residual) rs = *reinterpret_cast<const float4*>(a.residual + (long)m * a.res_ld + n);
      if (a.bias) bs = *reinterpret_cast<const float4*>(a.bias + n);
      const float4 s4 = *reinterpret_cast<const float4*>(stage + rl * RG_LDS_STAGE + 4 * q);
      float v[4] = {s4.x + bs.x, s4.y + bs.y, s4.z + bs.z, s4.w + bs.w};
      const float pmv[4] = {pm.x, pm.y, pm.z, pm.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
      float4 du = make_float4(1.f, 1.f, 1.f, 1.f);
      if (a.drop_p > 0.f) du = uniform4(seed, a.drop_site, ((uint64_t)m * a.N + n) >> 2);
      const float uu[4] = {du.x, du.y, du.z, du.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = v[c];
        if (a.relu) x = fmaxf(x, 0.f);
        if (a.posmask) x = pmv[c] > 0.f ? x : 0.f;
        if (a.cscale != 0.f) x *= a.cscale;
        if (a.drop_p > 0.f) x = uu[c] >= a.drop_p ? x * inv_keep : 0.f;
        v[c] = x + rsv[c];
      }
      *reinterpret_cast<float4*>(a.C + (long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (rd == 0) RGSTAMP(6);
    if (rd + 1 < nrounds) lds_barrier();                                // stage tile is rewritten next round
    }
  }
  RGSTAMP(7);
}

template <int KC, int ROWS, int NJ, bool LN = false, bool LNB = false, int WV = 8>
int launch_rowgemm_kc(const RowGemmArgs& a, hipStream_t st) {
  const size_t lds = (size_t)2 * ROWS * (KC * 32 + 16) * sizeof(__bf16) + (size_t)ROWS * (WV * NJ * 16 + 4) * sizeof(float) +
                     (LNB ? (size_t)WV * 2 * KC * 32 * sizeof(float) : 0);
  RD_LDS_ATTR((k_rowgemm<KC, ROWS, NJ, LN, LNB, WV>), lds);
  hipLaunchKernelGGL((k_rowgemm<KC, ROWS, NJ, LN, LNB, WV>), dim3(cdiv(a.M, ROWS)), dim3(64 * WV), lds, st, a);
  return check_launch("k_rowgemm");
}

}  // namespace

extern "C" void rd_debug_set_rowgemm_stamps(void* p) {   // not part of the ABI
  g_rg_stamps = (unsigned long long*)p;
}

// ---- host interface (used by rd_temporal.hip) -----------------------------------------------------
bool rowgemm_ok(int N, int K, long lda, long ldc) {
  static const bool enabled = [] { const char* e = getenv("RD_ROWGEMM"); return !(e && atoi(e) == 0); }();
  const int kc = (K + 31) / 32;
  return enabled && precision() != RD_PREC_FP32 && (N % 4) == 0 && (K % 4) == 0 && (lda % 4) == 0 && (ldc % 4) == 0 &&
         (kc == 5 || kc == 9 || kc == 15);
}
size_t rowgemm_plane_elems(int rows, int cols) { return (size_t)((rows + 15) / 16 * 16) * ((cols + 31) / 32 * 32); }

// split up to 8 weight matrices with one launch; job i: W [N_i, K_i] -> planes at hi_i / lo_i
// up to WS_MAXJOBS matrices and WS_MAXONES constant-tile buffers in ONE launch (a whole training step's weights: rd_step_prepare)
int launch_wsplit_specs(int njobs, const WsplitSpec* specs, int nones, void* const* ones, hipStream_t st) {
  return launch_wsplit_plan(njobs, specs, nones, ones, nullptr, nullptr, 0, 0, nullptr, 0, st);
}
// the same launch with one extra workgroup that builds the token plan of the step (plan_out == null: none)
int launch_wsplit_plan(int njobs, const WsplitSpec* specs, int nones, void* const* ones, const int64_t* lengths, int32_t* plan_out,
                       int B, int T, uint64_t* seed_cell_dev, uint64_t delta, hipStream_t st) {
  if (njobs < 1 || njobs > WS_MAXJOBS || nones < 0 || nones > WS_MAXONES) return fail(RD_EINVAL, "wsplit: %d jobs, %d constant tiles", njobs, nones);
  SplitJobs jobs{};
  jobs.n = njobs;
  static const int plan_first = [] { const char* e = getenv("RD_PLAN_FIRST"); return !(e && atoi(e) == 0); }();
  jobs.plan_first = plan_first;
  jobs.plan_lengths = lengths; jobs.plan_out = plan_out; jobs.plan_B = B; jobs.plan_T = T; jobs.seed_cell = seed_cell_dev; jobs.seed_delta = delta;
  if (plan_out) { const AdamCellReg ac = adam_cell(); jobs.adam_state = ac.state; jobs.adam_b1 = ac.b1; jobs.adam_b2 = ac.b2; }
  for (int i = 0; i < nones; ++i) jobs.ones[i] = (__bf16*)ones[i];
  for (int i = 0; i < njobs; ++i) {
    SplitJob& j = jobs.j[i];
    j.W = specs[i].W; j.N = specs[i].N; j.K = specs[i].K; j.transpose = specs[i].transpose; j.hi = (__bf16*)specs[i].tiles; j.lo = nullptr;
    const int rows = j.transpose ? j.K : j.N, cols = j.transpose ? j.N : j.K;
    j.rows = (rows + 15) / 16 * 16; j.cols_p = (cols + 31) / 32 * 32;
  }
  const size_t lds = plan_out ? plan::lds_bytes(B, T) : 0;
  if (lds > 64 * 1024) return fail(RD_EINVAL, "rd_step_begin: B too large for the token plan workgroups (%d, %d)", B, T);
  // workgroups per job: a job is 25-150 tiles of one wave-iteration each; RD_WSPLIT_GX (A/B only) overrides the default
  static const int gx = [] { const char* e = getenv("RD_WSPLIT_GX"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }();
  hipLaunchKernelGGL(k_wsplit, dim3(gx, njobs + (plan_out ? 1 : 0)), dim3(256), lds, st, jobs);
  return check_launch("k_wsplit");
}

int launch_wsplit(int njobs, const float* const* W, const int* N, const int* K, const int* transpose, __bf16* const* hi,
                  __bf16* const* lo, void* ones, hipStream_t st) {
  (void)lo;
  WsplitSpec specs[WS_MAXJOBS];
  if (njobs > WS_MAXJOBS) return fail(RD_EINVAL, "wsplit: too many jobs");
  for (int i = 0; i < njobs; ++i) specs[i] = WsplitSpec{W[i], N[i], K[i], transpose[i], hi[i]};
  void* on[1] = {ones};
  return launch_wsplit_specs(njobs, specs, ones ? 1 : 0, on, st);
}

// 32-row workgroups (480 instead of 240 at P19: two or more per CU, so one's loads overlap another's MFMAs instead of every
// workgroup of the launch moving through load -> multiply -> store in lockstep).  Bits: 1 plain K<=160, 2 plain K<=288,
// 4 LayerNorm epilogue, 8 LayerNorm-backward prologue.  MEASURED in-step (same box, ms/step): none 1.077, bits 1|2 1.047,
// 1|2|4 0.992,
// all four 0.966 -- but on a "fast" box of the same pool 0.7205 without vs 0.7268 with all four: which is better depends on the
// device's state, so raindrop_amd.step.TrainStep measures both when it captures its graph (rd_set_rowgemm_rows32).
static int g_rows32 = -1;                            // < 0: environment / default
static int rows32_mask() {
  if (g_rows32 >= 0) return g_rows32;
  static const int m = [] { const char* e = getenv("RD_RG_ROWS32"); return e ? atoi(e) : 15; }();
  return m;
}
// 16-wave workgroups (one column tile per wave and round instead of two): same bit assignment as rows32
static int g_waves16 = -1;
static int waves16_mask() {
  if (g_waves16 >= 0) return g_waves16;
  // MEASURED in-step (ms/step, workgroup height autotuned in each run): fast box 0 -> 0.687, 3 -> 0.682, 12 -> 0.666, 15 -> 0.663-0.678;
  // slow box 12 -> 0.828, 15 -> 0.836 (8 waves everywhere: 0.867).  Default: the LayerNorm-fused variants on 16 waves.
  static const int m = [] { const char* e = getenv("RD_RG_WAVES16"); return e ? atoi(e) : 12; }();
  return m;
}
extern "C" int rd_set_rowgemm_waves16(int32_t mask) {
  g_waves16 = mask < 0 ? -1 : (mask & 15);
  return RD_OK;
}
extern "C" int rd_set_rowgemm_rows32(int32_t mask) {   // tuning knob, see include/raindrop_hip.h
  g_rows32 = mask < 0 ? -1 : (mask & 15);
  return RD_OK;
}

// The next launch_rowgemm / launch_rowgemm_ln call also exports its A operand as row tiles (nct = ceil(K / 16) column
// tiles per 32-row chunk) to `tiles`; one-shot.
static thread_local void* g_export = nullptr;
void rowgemm_export_next(void* tiles) { g_export = tiles; }
// device pointer to the live row count of the following launches (sticky until reset to null by the caller)
static thread_local const int32_t* g_mlive = nullptr;
void rowgemm_set_mlive(const int32_t* p) { g_mlive = p; }
static void take_export(RowGemmArgs& a) { a.xt = (__bf16*)g_export; a.xt_nct = (a.K + 15) / 16; g_export = nullptr; a.mlive = g_mlive; }

// s = residual + dropout(A Wp^T + bias) -> s_out;  y = LayerNorm(s) g + b;  stats[m] = (mean, rstd).  N <= 256, N % 4 == 0,
// all row strides N.  Same values as launch_rowgemm(.. -> o) followed by the add+LayerNorm kernel.
bool rowgemm_ln_ok(int N, int K) { return rowgemm_ok(N, K, K, N) && N <= 256 && ((K + 31) / 32 == 5 || (K + 31) / 32 == 9); }
int launch_rowgemm_ln(long M, int N, int K, const float* A, const void* Wh, const float* bias, const float* residual,
                      const float* ln_g, const float* ln_b, float* s_out, float* y, float* stats, float drop_p,
                      uint64_t drop_seed, uint32_t drop_site, hipStream_t st) {
  RowGemmArgs a{};
  a.A = A; a.lda = K; a.Wh = (const __bf16*)Wh; a.C = y; a.ldc = N;
  a.M = (int)M; a.N = N; a.K = K; a.KP = (K + 31) / 32 * 32;
  a.bias = bias; a.residual = residual; a.res_ld = N;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_site = drop_site; a.seed_cell = seed_cell();
  a.stamps = g_rg_stamps;
  a.one_product = precision() == RD_PREC_BF16;
  a.ln_g = ln_g; a.ln_b = ln_b; a.ln_s = s_out; a.ln_stats = stats;
  take_export(a);
  const bool r32 = rows32_mask() & 4, w16 = waves16_mask() & 4;
  if (a.KP == 160) {
    if (w16) return r32 ? launch_rowgemm_kc<5, 32, 1, true, false, 16>(a, st) : launch_rowgemm_kc<5, 64, 1, true, false, 16>(a, st);
    return r32 ? launch_rowgemm_kc<5, 32, 2, true>(a, st) : launch_rowgemm_kc<5, 64, 2, true>(a, st);
  }
  if (w16) return r32 ? launch_rowgemm_kc<9, 32, 1, true, false, 16>(a, st) : launch_rowgemm_kc<9, 64, 1, true, false, 16>(a, st);
  return r32 ? launch_rowgemm_kc<9, 32, 2, true>(a, st) : launch_rowgemm_kc<9, 64, 2, true>(a, st);
}

// C = epi(A Wp^T) with A = LayerNorm-backward of (dy, s, stats, gamma) computed in the prologue (K = LayerNorm width <= 160);
// ds_out [M,K]: gradient of the pre-norm sum; part [ceil(M/64)][2K]: dgamma | dbeta partials.  Epilogue: posmask/cscale only.
bool rowgemm_lnb_ok(int N, int K) { return rowgemm_ok(N, K, K, N) && (K + 31) / 32 == 5; }
int rowgemm_lnb_part_rows(long M) { return (rows32_mask() & 8) ? (int)((M + 31) / 32) : (int)((M + 63) / 64); }
int launch_rowgemm_lnb(long M, int N, int K, const float* dy, const float* s, const float* stats, const float* g, float* ds_out,
                       float* part, float p_drop, uint64_t seed, uint32_t site, const void* Wh, float* C, long ldc,
                       const float* posmask, long pm_ld, float cscale, hipStream_t st) {
  RowGemmArgs a{};
  a.lda = K; a.Wh = (const __bf16*)Wh; a.C = C; a.ldc = ldc;
  a.M = (int)M; a.N = N; a.K = K; a.KP = (K + 31) / 32 * 32;
  a.posmask = posmask; a.pm_ld = pm_ld; a.cscale = cscale;
  a.seed_cell = seed_cell();
  a.stamps = g_rg_stamps;
  a.one_product = precision() == RD_PREC_BF16;
  a.lnb_dy = dy; a.lnb_s = s; a.lnb_stats = stats; a.lnb_g = g; a.lnb_ds = ds_out; a.lnb_part = part;
  a.lnb_p = p_drop; a.lnb_site = site; a.lnb_seed = seed;
  take_export(a);
  const bool r32 = rows32_mask() & 8;
  if (waves16_mask() & 8)
    return r32 ? launch_rowgemm_kc<5, 32, 1, false, true, 16>(a, st) : launch_rowgemm_kc<5, 64, 1, false, true, 16>(a, st);
  return r32 ? launch_rowgemm_kc<5, 32, 2, false, true>(a, st) : launch_rowgemm_kc<5, 64, 2, false, true>(a, st);
}

// C[M,N] = epi(A[M,K] Wp^T): Wp planes [ceil16(N)][ceil32(K)]
int launch_rowgemm(long M, int N, int K, const float* A, long lda, const void* Wh, const void* Wl, float* C, long ldc,
                   const float* bias, int relu, const float* posmask, long pm_ld, float cscale, const float* residual,
                   long res_ld, float drop_p, uint64_t drop_seed, uint32_t drop_site, hipStream_t st) {
  RowGemmArgs a{};
  a.A = A; a.lda = lda; a.Wh = (const __bf16*)Wh; a.Wl = (const __bf16*)Wl; a.C = C; a.ldc = ldc;
  a.M = (int)M; a.N = N; a.K = K; a.KP = (K + 31) / 32 * 32;
  a.bias = bias; a.relu = relu; a.posmask = posmask; a.pm_ld = pm_ld; a.cscale = cscale;
  a.residual = residual; a.res_ld = res_ld;
  a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_site = drop_site; a.seed_cell = seed_cell();
  a.stamps = g_rg_stamps;
  a.one_product = precision() == RD_PREC_BF16;
  take_export(a);
  const int kc = a.KP / 32;
  const int rows32 = rows32_mask();
  const int w16 = waves16_mask();
  if (kc == 5 && (w16 & 1))
    return (rows32 & 1) ? launch_rowgemm_kc<5, 32, 1, false, false, 16>(a, st) : launch_rowgemm_kc<5, 64, 1, false, false, 16>(a, st);
  if (kc == 9 && (w16 & 2))
    return (rows32 & 2) ? launch_rowgemm_kc<9, 32, 1, false, false, 16>(a, st) : launch_rowgemm_kc<9, 64, 1, false, false, 16>(a, st);
  if (kc == 5 && (rows32 & 1)) return launch_rowgemm_kc<5, 32, 2>(a, st);
  if (kc == 9 && (rows32 & 2)) return launch_rowgemm_kc<9, 32, 2>(a, st);
  if (kc == 5) return launch_rowgemm_kc<5, 64, 2>(a, st);
  if (kc == 9) return launch_rowgemm_kc<9, 64, 2>(a, st);
  // K = 3D (QKV dgrad): 32 rows keep planes + stage inside 160 KB; one column tile per wave keeps the 15-step panel in 120 VGPRs
  if (kc == 15) return launch_rowgemm_kc<15, 32, 1>(a, st);
  return fail(RD_EUNSUPPORTED, "rowgemm: K=%d not built", K);
}

}  // namespace rd
