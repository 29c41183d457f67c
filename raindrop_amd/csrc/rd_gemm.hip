// rd_gemm.hip -- generic fp32 GEMM on the gfx950 f32-input MFMA, with the fused epilogues the
// Raindrop path needs (bias / ReLU / per-sensor aggregate scale / ReLU-mask / residual /
// [B,F,K] -> [T,B,F*d] scatter) and a deterministic split-K form for weight gradients.
//
// Replaces the torch call sites: F.linear inside Observation_progation.message
// (code/Ob_propagation.py:200), nn.TransformerEncoderLayer's in_proj / out_proj / linear1 /
// linear2 (code/models_rd.py:235-237,358), emb / mlp_static (code/models_rd.py:294,385) and
// their autograd backward (mm / addmm on dy, W, x).
//
// Tiling: 64x64 output tile per 256-thread workgroup (4 wavefronts, each a 32x32 quadrant =
// 2x2 MFMA tiles of 16x16), K stepped by 32 through LDS.  v_mfma_f32_16x16x4_f32 is exact fp32
// (one rounding per product, bitwise an fmaf chain), so results match a plain fp32 reference to
// summation-order rounding.  Within each 16-wide K chunk lane l consumes k = 4*(l>>4)+j on MFMA
// step j for BOTH operands, so each lane fetches its four k values with one ds_read_b128.
#include <stdlib.h>

#include "rd_common.h"
#include <type_traits>
#include "rd_rng.h"

namespace rd {

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDT = BK + 4;   // LDT*4 B = 144 B rows: 16-B aligned, conflict-free b128

// ---- global -> register staging --------------------------------------------------------------
// KC (k contiguous in memory): thread owns rows {tid/8, 32+tid/8}, k quad (tid%8)*4.
// MC (row contiguous in memory): thread owns row tid%64, k = tid/64 + 4*i, i<8.
template <bool KC>
__device__ __forceinline__ void stage_load(float (&r)[8], const float* __restrict__ P, long s_row,
                                           long s_k, int row0, int nrows, int k0, int kend, int tid,
                                           bool vec_ok) {
  if (KC) {
    const int kq = k0 + (tid & 7) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = row0 + p * 32 + (tid >> 3);
      const bool rok = row < nrows;
      if (rok && vec_ok && kq + 3 < kend) {
        const float4 v = *reinterpret_cast<const float4*>(P + (long)row * s_row + kq);
        r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          r[p * 4 + j] = (rok && kq + j < kend) ? P[(long)row * s_row + kq + j] : 0.f;
      }
    }
  } else {
    const int row = row0 + (tid & 63);
    const bool rok = row < nrows;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + (tid >> 6) + 4 * i;
      r[i] = (rok && k < kend) ? P[(long)row * s_row + (long)k * s_k] : 0.f;
    }
  }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const float (&r)[8], float* __restrict__ T, int tid) {
  if (KC) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float4 v = make_float4(r[p * 4 + 0], r[p * 4 + 1], r[p * 4 + 2], r[p * 4 + 3]);
      *reinterpret_cast<float4*>(T + (p * 32 + (tid >> 3)) * LDT + (tid & 7) * 4) = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) T[(tid & 63) * LDT + (tid >> 6) + 4 * i] = r[i];
  }
}

// acc[i][j][r] is C(m, n) with m = mw + 16*i + 4*(lane>>4) + r, n = nw + 16*j + (lane&15)
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x4 (&acc)[2][2], int mw, int nw, int z, int lane) {
  float* Cz = g.C + (long)z * g.sc_split;
  const bool raw = g.nsplit > 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = nw + j * 16 + (lane & 15);
      if (n >= g.N) continue;
      const float bias = (!raw && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mw + i * 16 + 4 * (lane >> 4) + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r];
        if (!raw) {
          v += bias;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.rowscale) v *= g.rowscale[m % g.rs_period];
          if (g.posmask) v = (g.posmask[(long)m * g.pm_m + n] > 0.f) ? v : 0.f;
          if (g.cscale != 0.f) v *= g.cscale;
          if (g.drop_p > 0.f)
            v *= dropout_scale(eff_seed(g.drop_seed, g.seed_cell), g.drop_site, (uint64_t)m * g.N + n, g.drop_p,
                               1.0f / (1.0f - g.drop_p));
          if (g.residual) v += g.residual[(long)m * g.res_m + n];
        }
        if (g.scatter && !raw) {
          const int b = m / g.sF, f = m - b * g.sF;
          const int t = n / g.sd, c = n - t * g.sd;
          g.C[((long)t * g.sB + b) * g.ldz + f * g.sd + c] = v;
        } else {
          Cz[(long)m * g.sc_m + n] = v;
        }
      }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  RD_TOUCH_CODE_X(RD_TL_GEMM, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  int z = blockIdx.z;
  if (g.nbatch > 1) {                                 // batched: z selects the problem, no split
    const int o = z / g.batch_inner, i = z - o * g.batch_inner;
    g.A += o * g.a_bo + i * g.a_bi; g.B += o * g.b_bo + i * g.b_bi; g.C += o * g.c_bo + i * g.c_bi;
    if (g.residual && g.res_batched) g.residual += o * g.c_bo + i * g.c_bi;
    z = 0;
  }
  const int kbeg = z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  const bool a_vec = A_KC && ((g.sa_m & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool b_vec = B_KC && ((g.sb_n & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float ra[8], rb[8];
  float rsum = 0.f;
  const bool do_rowsum = !A_KC && g.rowsum != nullptr && blockIdx.x == 0;
  if (kbeg < kend) {
    stage_load<A_KC>(ra, g.A, g.sa_m, g.sa_k, m0, g.M, kbeg, kend, tid, a_vec);
    stage_load<B_KC>(rb, g.B, g.sb_n, g.sb_k, n0, g.N, kbeg, kend, tid, b_vec);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    if (do_rowsum) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rsum += ra[i];
    }
    stage_store<A_KC>(ra, As, tid);
    stage_store<B_KC>(rb, Bs, tid);
    __syncthreads();
    if (k0 + BK < kend) {   // next tile's global loads fly while this tile is multiplied
      stage_load<A_KC>(ra, g.A, g.sa_m, g.sa_k, m0, g.M, k0 + BK, kend, tid, a_vec);
      stage_load<B_KC>(rb, g.B, g.sb_n, g.sb_k, n0, g.N, k0 + BK, kend, tid, b_vec);
    }
#pragma unroll
    for (int kc = 0; kc < BK; kc += 16) {
      float4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const float4*>(As + (wy * 32 + i * 16 + (lane & 15)) * LDT + kc + 4 * (lane >> 4));
        bf[i] = *reinterpret_cast<const float4*>(Bs + (wx * 32 + i * 16 + (lane & 15)) * LDT + kc + 4 * (lane >> 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  if (do_rowsum) {
    float* red = As;
    red[wave * 64 + lane] = rsum;
    __syncthreads();
    if (tid < 64 && m0 + tid < g.M)
      g.rowsum[(long)z * g.rowsum_split + m0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
  }
  epilogue(g, acc, m0 + wy * 32, n0 + wx * 32, z, lane);
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant: every fp32 operand x is split on the fly into hi = bf16(x), lo = bf16(x - hi)
// when its tile is stored to LDS, and each product is evaluated as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the dropped lo*lo term is ~2^-16 relative).
// 3 bf16 MFMAs replace 8 f32 MFMAs per 16x16x32 block: 16x the MFMA rate at 3x the count = 5.3x.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BK2 = 64, LDB = BK2 + 16;  // bf16 elements; 160-byte rows (32 bytes mod 64: conflict-free fragment reads, rd_encfuse.hip LDD)

// Tile = (32*MI) x (32*NI) outputs per 256-thread workgroup (2x2 waves, each 16*MI x 16*NI), K step 64.
// Staging of a ROWS x 64 fp32 tile (ROWS = 32*MI or 32*NI):
//   KC (k contiguous in memory): thread owns rows {tid/16 + 16p}, k quad (tid%16)*4  -> float4 loads;
//   MC (row contiguous in memory, ROWS in {64,128}): thread owns row tid%ROWS and 64/(256/ROWS)
//      consecutive k -> dword loads coalesced along the rows, 16-B LDS stores.
template <bool KC, int ROWS>
struct Stage {
  static constexpr int NREG = ROWS / 4;                      // fp32 values per thread per tile
  static constexpr int TPR = 256 / ROWS;                     // MC: threads per row
  static constexpr int KPT = 64 / (TPR > 0 ? TPR : 1);       // MC: consecutive k per thread
  // `ok`: validity bits of r[] (KC staging only).  The k-contiguous loads are UNCONDITIONAL from clamped, always
  // legal addresses and the zero padding is applied in store(): a conditional load is a phi of {0, value} and
  // its two arms (16-byte / scalar tail) write the same registers, which made the compiler wait for the A tile
  // before it had requested the B tile (one extra memory round trip per workgroup in every prologue).
  __device__ static __forceinline__ void load(float (&r)[NREG], unsigned long long& ok, const float* __restrict__ P,
                                              long s_row, long s_k, int row0, int nrows, int k0, int kend, int tid,
                                              bool vec_ok) {
    ok = 0ull;
    if (KC) {
      const int kq = k0 + (tid & 15) * 4;
      if (vec_ok) {                 // uniform.  Row stride % 4 == 0 and 16-byte aligned base: a quad that starts below
        const bool kok = kq < kend; // kend stays inside its row's stride even when K % 4 != 0 (tail masked below)
        const int kc = kok ? kq : k0;
        const unsigned long long kbits = (kq < kend ? 1ull : 0ull) | (kq + 1 < kend ? 2ull : 0ull) |
                                         (kq + 2 < kend ? 4ull : 0ull) | (kq + 3 < kend ? 8ull : 0ull);
#pragma unroll
        for (int p = 0; p < ROWS / 16; ++p) {
          const int row = row0 + p * 16 + (tid >> 4);
          const bool rok = row < nrows;
          const float4 v = *reinterpret_cast<const float4*>(P + (long)(rok ? row : row0) * s_row + kc);
          r[p * 4 + 0] = v.x; r[p * 4 + 1] = v.y; r[p * 4 + 2] = v.z; r[p * 4 + 3] = v.w;
          if (rok) ok |= kbits << (4 * p);
        }
      } else {
#pragma unroll
        for (int p = 0; p < ROWS / 16; ++p) {
          const int row = row0 + p * 16 + (tid >> 4);
          const bool rok = row < nrows;
          const float* src = P + (long)(rok ? row : row0) * s_row;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool kok = kq + j < kend;
            r[p * 4 + j] = src[kok ? kq + j : k0];
            if (rok && kok) ok |= 1ull << (4 * p + j);
          }
        }
      }
    } else {
      const int row = row0 + (tid % ROWS);
      const bool rok = row < nrows;
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        const int k = k0 + (tid / ROWS) * KPT + i;
        r[i] = (rok && k < kend) ? P[(long)row * s_row + (long)k * s_k] : 0.f;
      }
    }
  }
  // LO = false (one-product bf16 mode, where the lo plane is never read): the residual's arithmetic and its LDS store are skipped
  template <bool LO = true>
  __device__ static __forceinline__ void store(const float (&r)[NREG], unsigned long long ok, __bf16* __restrict__ Th,
                                               __bf16* __restrict__ Tl, int tid) {
    if (KC) {
#pragma unroll
      for (int p = 0; p < ROWS / 16; ++p) {
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = ((ok >> (4 * p + j)) & 1ull) ? r[p * 4 + j] : 0.f;
          h[j] = (__bf16)x;
          if (LO) l[j] = (__bf16)(x - (float)h[j]);
        }
        const int o = (p * 16 + (tid >> 4)) * LDB + (tid & 15) * 4;
        *reinterpret_cast<bf16x4*>(Th + o) = h;
        if (LO) *reinterpret_cast<bf16x4*>(Tl + o) = l;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NREG / 8; ++q) {
        bf16x8 h, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = r[q * 8 + j];
          h[j] = (__bf16)x;
          if (LO) l[j] = (__bf16)(x - (float)h[j]);
        }
        const int o = (tid % ROWS) * LDB + (tid / ROWS) * KPT + q * 8;
        *reinterpret_cast<bf16x8*>(Th + o) = h;
        if (LO) *reinterpret_cast<bf16x8*>(Tl + o) = l;
      }
    }
  }
};

// Tile epilogue through LDS.  The MFMA accumulator layout gives a lane 4 consecutive ROWS of one
// column, i.e. 64-byte store segments; the tile is therefore transposed through LDS (the operand
// planes are dead by now) so that each thread owns 4 consecutive COLUMNS of one row: bias /
// mask / residual reads and the output stores become 16-byte accesses in 256..640-byte runs, and the
// Philox dropout mask costs one evaluation per 4 elements.
template <int MI, int NI, int WY = 2, int WX = 2, int NTHR = 256>   // WY x WX waves hold 16 MI x 16 NI outputs each; NTHR threads store
__device__ __forceinline__ void epilogue_t(const GemmArgs& g, f32x4 (&acc)[MI][NI], float* stage, const float* bias_s,
                                           int m0, int n0, int wy, int wx, int z, int tid, int lane, bool writer = true) {
  constexpr int TM = 16 * MI * WY, TN = 16 * NI * WX, LDSG = TN + 4;
  if (writer) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stage[(wy * 16 * MI + i * 16 + 4 * (lane >> 4) + r) * LDSG + wx * 16 * NI + j * 16 + (lane & 15)] = acc[i][j][r];
  }
  __syncthreads();
  float* Cz = g.C + (long)z * g.sc_split;
  const bool raw = g.nsplit > 1;
  const bool vec = ((g.N & 3) == 0) && ((g.sc_m & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cz) & 15) == 0) &&
                   !g.scatter && (!g.posmask || (((g.pm_m & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.posmask) & 15) == 0))) &&
                   (!g.residual || (((g.res_m & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.residual) & 15) == 0)));
  const bool svec = g.scatter && g.sd == 4 && ((g.N & 3) == 0) && ((g.ldz & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  const float inv_keep = 1.0f / (1.0f - g.drop_p);
  constexpr int QPR = TN / 4;                         // column quads per tile row
  constexpr int ITER = TM * QPR / NTHR;
  // Straight path for the common epilogues (bias, ReLU, row scale, gate mask, constant scale; 16-byte rows or the d_ob = 4
  // scatter): every global read -- row scales included -- is requested in one unbranched pass before the first is used.  The
  // general loop below interleaves loads, flag tests and stores per iteration; measured 11 k cycles per 64 x 128 tile against
  // a 25 k-cycle main loop in the panel kernel.
  if (!raw && (vec || (svec && !g.posmask)) && !g.residual && !(g.drop_p > 0.f)) {
    // scatter: element offset of the thread's (b, f) cell at step 0 and the sample's live steps (same two registers per row as
    // before the token plan: the step stride carries the layout -- sB * ldz on the padded layout, ldz on the plan's)
    float rsc[ITER]; float4 pm[ITER]; int boff[ITER], lnq[ITER];
    const long tstride = g.sp_row0 ? g.ldz : (long)g.sB * g.ldz;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * NTHR;
      const int rl = e / QPR, q = e - rl * QPR;
      const int mc = min(m0 + rl, g.M - 1), nc = min(n0 + 4 * q, g.N - 4);        // clamped: always legal addresses
      rsc[it] = g.rowscale ? g.rowscale[mc % g.rs_period] : 1.f;
      pm[it] = g.posmask ? *reinterpret_cast<const float4*>(g.posmask + (long)mc * g.pm_m + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
      boff[it] = 0; lnq[it] = 0x7fffffff;
      if (svec) {
        const int bb = mc / g.sF, ff = mc - bb * g.sF;
        int r0 = bb;
        if (g.sp_row0) { r0 = g.sp_row0[bb]; lnq[it] = g.sp_len[bb]; }                    // uniform flag; requested with the rest
        boff[it] = (int)(r0 * g.ldz) + ff * 4;
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int e = tid + it * NTHR;
      const int rl = e / QPR, q = e - rl * QPR;
      const int m = m0 + rl, n = n0 + 4 * q;
      if (m >= g.M || n >= g.N) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(stage + rl * LDSG + 4 * q);
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      const float pmv[4] = {pm[it].x, pm[it].y, pm[it].z, pm[it].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = v[c];
        if (g.bias) x += bias_s[4 * q + c];
        if (g.relu) x = fmaxf(x, 0.f);
        x *= rsc[it];
        if (g.posmask) x = (pmv[c] > 0.f) ? x : 0.f;
        if (g.cscale != 0.f) x *= g.cscale;
        v[c] = x;
      }
      if (svec && (n >> 2) >= lnq[it]) continue;          // token plan: padded steps have no row
      float* dst = svec ? g.C + (long)(n >> 2) * tstride + boff[it] : Cz + (long)m * g.sc_m + n;
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  // every global read of the epilogue (mask / residual) is issued before any of them is consumed:
  // one memory round trip for the whole tile instead of one per row group
  float4 pm4[ITER], rs4[ITER];
  float rsc_r[ITER];                                  // row scale of each of this thread's rows (a load per row inside the
#pragma unroll                                        // store loop below was one dependent round trip per iteration)
  for (int it = 0; it < ITER; ++it) {
    const int e = tid + it * NTHR;
    const int rl = e / QPR, q = e - rl * QPR;
    const int m = m0 + rl, n = n0 + 4 * q;
    pm4[it] = make_float4(1.f, 1.f, 1.f, 1.f); rs4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    rsc_r[it] = (!raw && g.rowscale && m < g.M) ? g.rowscale[m % g.rs_period] : 1.f;
    if (!raw && m < g.M && n < g.N) {
      if (g.posmask) {
        if (vec) pm4[it] = *reinterpret_cast<const float4*>(g.posmask + (long)m * g.pm_m + n);
        else {
          float t4[4] = {0.f, 0.f, 0.f, 0.f};
          for (int c = 0; c < 4 && n + c < g.N; ++c) t4[c] = g.posmask[(long)m * g.pm_m + n + c];
          pm4[it] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
      if (g.residual) {
        if (vec) rs4[it] = *reinterpret_cast<const float4*>(g.residual + (long)m * g.res_m + n);
        else {
          float t4[4] = {0.f, 0.f, 0.f, 0.f};
          for (int c = 0; c < 4 && n + c < g.N; ++c) t4[c] = g.residual[(long)m * g.res_m + n + c];
          rs4[it] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int e = tid + it * NTHR;
    const int rl = e / QPR, q = e - rl * QPR;
    const int m = m0 + rl, n = n0 + 4 * q;
    if (m >= g.M || n >= g.N) continue;
    const float4 a4 = *reinterpret_cast<const float4*>(stage + rl * LDSG + 4 * q);
    float v[4] = {a4.x, a4.y, a4.z, a4.w};
    const int nv = min(4, g.N - n);
    if (!raw) {
      const float rsc = rsc_r[it];
      float4 du = make_float4(1.f, 1.f, 1.f, 1.f);
      if (g.drop_p > 0.f) {
        if (vec) {
          du = uniform4(eff_seed(g.drop_seed, g.seed_cell), g.drop_site, ((uint64_t)m * g.N + n) >> 2);
        } else {
          float t4[4];
          for (int c = 0; c < 4; ++c)
            t4[c] = dropout_scale(eff_seed(g.drop_seed, g.seed_cell), g.drop_site, (uint64_t)m * g.N + n + c, g.drop_p, 1.f) > 0.f ? 1.f : 0.f;
          du = make_float4(t4[0], t4[1], t4[2], t4[3]);     // 1 = keep (>= p), 0 = drop (< p)
        }
      }
      const float uu[4] = {du.x, du.y, du.z, du.w};
      const float pmv[4] = {pm4[it].x, pm4[it].y, pm4[it].z, pm4[it].w};
      const float rsv[4] = {rs4[it].x, rs4[it].y, rs4[it].z, rs4[it].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c >= nv) break;
        float x = v[c];
        if (g.bias) x += bias_s[4 * q + c];
        if (g.relu) x = fmaxf(x, 0.f);
        x *= rsc;
        if (g.posmask) x = (pmv[c] > 0.f) ? x : 0.f;
        if (g.cscale != 0.f) x *= g.cscale;
        if (g.drop_p > 0.f) x = (uu[c] >= g.drop_p) ? x * inv_keep : 0.f;
        if (g.residual) x += rsv[c];
        v[c] = x;
      }
    }
    if (g.scatter && !raw) {
      const int b = m / g.sF, f = m - b * g.sF;
      if (svec) {                                     // d_ob = 4: the column quad is one (t, b, f) cell of z
        *reinterpret_cast<float4*>(g.C + ((long)(n >> 2) * g.sB + b) * g.ldz + f * 4) = make_float4(v[0], v[1], v[2], v[3]);
        continue;
      }
      for (int c = 0; c < nv; ++c) {                  // (the token plan's scatter takes the straight path above: launch_gemm checks)
        const int t = (n + c) / g.sd, cc = (n + c) - t * g.sd;
        g.C[((long)t * g.sB + b) * g.ldz + f * g.sd + cc] = v[c];
      }
    } else if (vec) {
      *reinterpret_cast<float4*>(Cz + (long)m * g.sc_m + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int c = 0; c < nv; ++c) Cz[(long)m * g.sc_m + n + c] = v[c];
    }
  }
}

// debug only (tools/gemm_timing.py).  The pointer travels as a KERNEL ARGUMENT (an SGPR): a __device__ global
// would cost a dependent global load plus s_waitcnt vmcnt(0) at every stamp site even when stamps are off.
static unsigned long long* g_gemm_stamps = nullptr;
#define GSTAMP(i)                                                                                    \
  do {                                                                                               \
    if (g.stamps && blockIdx.z == 0 && blockIdx.x < 8 && threadIdx.x == 0)                           \
      g.stamps[blockIdx.x * 8 + (i)] = clock64();                                             \
  } while (0)

// NPRE > 0: the whole K range of the workgroup (<= 64*NPRE) is requested up front, tile by tile into
// registers, together with the bias: global-memory latency under load is ~3 us on this chip, and a
// workgroup of these tall-skinny products lives for only a handful of tiles, so every dependent
// round trip removed is ~10 % of the kernel.  NPRE == 0: generic loop, one tile of lookahead.
template <bool A_KC, bool B_KC, int MI, int NI, int NPRE>
__global__ __launch_bounds__(256) void k_gemm_bf16x3(GemmArgs g) {
  constexpr int TM = 32 * MI, TN = 32 * NI;
  RD_TOUCH_CODE_X(RD_TL_GEMM_X3, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  GSTAMP(0);
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __bf16* Ah = reinterpret_cast<__bf16*>(gsm);
  __bf16* Al = Ah + TM * LDB;
  __bf16* Bh = Al + TM * LDB;
  __bf16* Bl = Bh + TN * LDB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs (each with its own
  // L2), so the column blocks that share one A row block are given ids that are equal mod 8: the row
  // block is fetched into ONE L2 and re-read there by its other column blocks, and every XCD keeps
  // its own copy of the (small) weight matrix.  Pure placement hint: any mapping is correct.
  const int ncb = (g.N + TN - 1) / TN;
  int lin = blockIdx.x;
  const int xcd = lin & 7, jj = lin >> 3;
  // (measured: helps the single-pass products by ~10 %, hurts the split-K weight-gradient products by
  // ~40 % -- there the z-slices already spread one row block over the XCDs -- so those keep row-major order)
  const bool swz = g.nsplit <= 1 && g.xcd_swizzle;
  // Split-K products (weight gradients): ALL output tiles of one reduction slice z run on XCD z % 8, so the slice's
  // operand rows are fetched from HBM into one L2 once and re-read there by the other tiles (with the plain
  // (tile, z) grid every XCD held a third of the tiles of EVERY slice and fetched most of both operands: PMC
  // 90 MB fetched for 19-38 MB of operands).  Grid: 8 * ceil(nz / 8) * tiles workgroups in x, surplus slices exit.
  int z = blockIdx.z;
  if (g.slice_xcd) {
    const int tiles = ncb * ((g.M + TM - 1) / TM);
    z = (jj / tiles) * 8 + xcd;
    lin = jj - (jj / tiles) * tiles;
    if (z >= g.nsplit * (g.A2 != nullptr ? 2 : 1)) return;
  }
  const int rblk = swz ? 8 * (jj / ncb) + xcd : lin / ncb;
  const int cblk = swz ? jj - (jj / ncb) * ncb : lin - (lin / ncb) * ncb;
  const int m0 = rblk * TM, n0 = cblk * TN;
  if (m0 >= g.M) return;                              // padding blocks of the last group of 8 row blocks
  if (g.nbatch > 1) {                                 // batched: z selects the problem, no split
    const int o = z / g.batch_inner, i = z - o * g.batch_inner;
    g.A += o * g.a_bo + i * g.a_bi; g.B += o * g.b_bo + i * g.b_bi; g.C += o * g.c_bo + i * g.c_bi;
    if (g.residual && g.res_batched) g.residual += o * g.c_bo + i * g.c_bi;
    z = 0;
  }
  if (g.A2 != nullptr && z >= g.nsplit) {             // second problem of a batched pair (uniform per block)
    z -= g.nsplit;
    g.A = g.A2; g.B = g.B2; g.C = g.C2; g.rowsum = g.rowsum2;
  }
  const int kbeg = z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const bool a_vec = A_KC && ((g.sa_m & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool b_vec = B_KC && ((g.sb_n & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  using SA = Stage<A_KC, TM>;
  using SB = Stage<B_KC, TN>;
  constexpr int NBUF = NPRE > 0 ? NPRE : 1;
  float ra[NBUF][SA::NREG], rb[NBUF][SB::NREG];
  unsigned long long oka[NBUF], okb[NBUF];               // validity bits of the k-contiguous staging (see Stage::load)
  float rsum = 0.f;                                   // MC staging: this thread's row is tid % TM
  const bool do_rowsum = !A_KC && g.rowsum != nullptr && cblk == 0;
  constexpr size_t PLANES_B = (size_t)2 * (TM + TN) * LDB * sizeof(__bf16), STAGE_B = (size_t)TM * (TN + 4) * sizeof(float);
  float* bias_s = reinterpret_cast<float*>(gsm + (PLANES_B > STAGE_B ? PLANES_B : STAGE_B));   // [TN], above planes and stage
  if (NPRE > 0) {
#pragma unroll
    for (int t = 0; t < NBUF; ++t)
      if (kbeg + t * BK2 < kend) {
        SA::load(ra[t], oka[t], g.A, g.sa_m, g.sa_k, m0, g.M, kbeg + t * BK2, kend, tid, a_vec);
        SB::load(rb[t], okb[t], g.B, g.sb_n, g.sb_k, n0, g.N, kbeg + t * BK2, kend, tid, b_vec);
      }
  } else if (kbeg < kend) {
    SA::load(ra[0], oka[0], g.A, g.sa_m, g.sa_k, m0, g.M, kbeg, kend, tid, a_vec);
    SB::load(rb[0], okb[0], g.B, g.sb_n, g.sb_k, n0, g.N, kbeg, kend, tid, b_vec);
  }
  if (tid < TN) bias_s[tid] = (g.bias && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;

  // `three` products (split-bf16) or hi*hi only (RD_PREC_BF16), decided outside the reduction loop: a branch inside it made every
  // step its own basic block and kept the next step's fragment reads below this step's products
  auto mma_steps = [&](auto three_tag) {
    constexpr bool THREE = decltype(three_tag)::value;
#pragma unroll
    for (int kc = 0; kc < BK2; kc += 32) {
      bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int oa = (wy * 16 * MI + i * 16 + (lane & 15)) * LDB + kc + 8 * (lane >> 4);
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + oa);
        if (THREE) al[i] = *reinterpret_cast<const bf16x8*>(Al + oa);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int ob = (wx * 16 * NI + j * 16 + (lane & 15)) * LDB + kc + 8 * (lane >> 4);
        bh[j] = *reinterpret_cast<const bf16x8*>(Bh + ob);
        if (THREE) bl[j] = *reinterpret_cast<const bf16x8*>(Bl + ob);
      }
      // three passes over independent accumulators: no back-to-back MFMA on the same registers
      if (THREE) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
  };
  auto mma_tile = [&]() {
    if (!g.one_product) mma_steps(std::true_type{});
    else mma_steps(std::false_type{});
  };
  auto stage_store = [&](int t) {                      // one-product mode never reads the lo planes: not computed, not stored
    if (!g.one_product) { SA::store(ra[t], oka[t], Ah, Al, tid); SB::store(rb[t], okb[t], Bh, Bl, tid); }
    else { SA::template store<false>(ra[t], oka[t], Ah, Al, tid); SB::template store<false>(rb[t], okb[t], Bh, Bl, tid); }
  };

  if (NPRE > 0) {
#pragma unroll
    for (int t = 0; t < NBUF; ++t) {
      if (kbeg + t * BK2 < kend) {                    // uniform
        if (do_rowsum) {
#pragma unroll
          for (int i = 0; i < SA::NREG; ++i) rsum += ra[t][i];
        }
        stage_store(t);
        __syncthreads();
        if (t == 0) GSTAMP(1);
        mma_tile();
        __syncthreads();
      }
    }
  } else {
    for (int k0 = kbeg; k0 < kend; k0 += BK2) {
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < SA::NREG; ++i) rsum += ra[0][i];
      }
      stage_store(0);
      __syncthreads();
      if (k0 == kbeg) GSTAMP(1);
      if (k0 + BK2 < kend) {
        SA::load(ra[0], oka[0], g.A, g.sa_m, g.sa_k, m0, g.M, k0 + BK2, kend, tid, a_vec);
        SB::load(rb[0], okb[0], g.B, g.sb_n, g.sb_k, n0, g.N, k0 + BK2, kend, tid, b_vec);
      }
      mma_tile();
      __syncthreads();
    }
  }
  GSTAMP(2);
  float* stage = reinterpret_cast<float*>(gsm);
  if (do_rowsum) {                                    // TPR threads share a row; combine in fixed order
    float* red = stage;
    red[tid] = rsum;
    __syncthreads();
    if (tid < TM && m0 + tid < g.M) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 256 / TM; ++q) v += red[q * TM + tid];
      g.rowsum[(long)z * g.rowsum_split + m0 + tid] = v;
    }
    __syncthreads();
  }
  epilogue_t<MI, NI>(g, acc, stage, bias_s, m0, n0, wy, wx, z, tid, lane);
  GSTAMP(3);
}

template <bool A_KC, bool B_KC, int MI, int NI, int NPRE>
int launch_bf16x3_n(const GemmArgs& g, hipStream_t st) {
  constexpr int TM = 32 * MI, TN = 32 * NI;
  const size_t planes = (size_t)2 * (TM + TN) * LDB * sizeof(__bf16);
  const size_t stage = (size_t)TM * (TN + 4) * sizeof(float);       // epilogue transpose tile (aliases the planes)
  const size_t lds = (planes > stage ? planes : stage) + TN * sizeof(float);   // + bias
  const int nz = g.nbatch > 1 ? g.nbatch : (g.nsplit > 1 ? g.nsplit : 1) * (g.A2 ? 2 : 1);
  dim3 grid((g.nsplit > 1 ? cdiv(g.M, TM) : 8 * cdiv(cdiv(g.M, TM), 8)) * cdiv(g.N, TN), 1, nz);
  if (g.slice_xcd) grid = dim3(8 * cdiv(nz, 8) * cdiv(g.M, TM) * cdiv(g.N, TN), 1, 1);
  if (lds > 48 * 1024)
    RD_LDS_ATTR((k_gemm_bf16x3<A_KC, B_KC, MI, NI, NPRE>), lds);
  hipLaunchKernelGGL((k_gemm_bf16x3<A_KC, B_KC, MI, NI, NPRE>), grid, dim3(256), lds, st, g);
  return check_launch("k_gemm_bf16x3");
}

template <bool A_KC, bool B_KC, int MI, int NI>
int launch_bf16x3(const GemmArgs& g, hipStream_t st) {
  // Measured on MI355X (tools/gemm_timing.py, tools/bw_probe.py): requesting all K tiles up front
  // (NPRE = tiles) is neutral at K=152 and 40 % slower at K=272 (register-limited occupancy), because
  // these products are bound by operand RE-READS at the L2 level (each 64x64 tile loads 39+39 KB to
  // write 16 KB), not by dependent latency; the looped form with one tile of lookahead stays.
  // ... except for the handful-of-workgroups products of the classifier head (M = B = 256 rows: 12 workgroups, 3 K tiles):
  // those are pure dependent-latency chains (measured 14-35 us for a 256 x 186 x 186 product), so the whole K range is
  // requested up front.
  static const bool npre_on = [] { const char* e = getenv("RD_GEMM_NPRE"); return !(e && atoi(e) == 0); }();
  if (MI == 2 && NI == 2 && npre_on) {
    const long blocks = (long)cdiv(g.M, 64) * cdiv(g.N, 64) * (g.nsplit > 1 ? g.nsplit : 1) * (g.A2 ? 2 : 1);
    const int kspan = g.nsplit > 1 ? g.k_per_split : g.K;
    if (blocks <= 128 && kspan <= 256) return launch_bf16x3_n<A_KC, B_KC, 2, 2, 4>(g, st);
  }
  return launch_bf16x3_n<A_KC, B_KC, MI, NI, 0>(g, st);
}

// ------------------------------------------------------------------------------------------------
// Panel product: C = epilogue(A W'), A [M,K] k-contiguous fp32, W' given as NATIVE operand tiles (k_wsplit, rd_rowgemm.hip:
// [n tile][k tile][hi, lo][64 lanes][8] -- a B fragment is one contiguous KB per wave, straight from L2 into registers, no
// LDS, no conversion).  Built for the square products of the unfused message passing (K = N = T d_ob: 860 at P12, 2400 at
// PAM), where k_gemm_bf16x3 converts BOTH operands through LDS every 64 k with two barriers around 24 MFMAs per wave.
// Here: workgroup = 64 rows x 64 NJ columns; four waves side by side (each: all 64 rows x 16 NJ columns, 12 RT NJ MFMAs per
// 32 k) -- twice: a second group of four waves takes the odd 64-k chunks (split K inside the workgroup: with ~1 workgroup per CU
// that is the second wave per SIMD that overlaps one group's loads and conversions with the other's products) and hands its
// accumulators over through LDS at the end.  Only A is staged (hi/lo planes, double buffered per group: ONE LDS-only barrier per
// round), the next chunk of A and of the weight fragments is in flight during the products.  Same epilogue as the tiled kernel (bias, ReLU, row scale, masks,
// dropout, residual, the [T,B,ldz] scatter), same XCD-aware tile order.
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(512) void k_gemm_panel(GemmArgs g) {
  RD_TOUCH_CODE_X(RD_TL_GEMM_PANEL, blockIdx.x, 512);
  constexpr int RT = 4, TM = 16 * RT, TN = 64 * NJ, PLANE = TM * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __bf16* Pb = reinterpret_cast<__bf16*>(gsm);                  // [2 groups][2 buffers][hi, lo][TM][LDB]
  constexpr size_t PLANES_B = (size_t)8 * PLANE * sizeof(__bf16), STAGE_B = (size_t)TM * (TN + 4) * sizeof(float);
  constexpr size_t RED_B = (size_t)4 * RT * NJ * 4 * 64 * sizeof(float);
  static_assert(STAGE_B + RED_B <= PLANES_B, "epilogue stage + group-1 accumulators must fit the dead planes");
  float* red = reinterpret_cast<float*>(gsm + STAGE_B);         // group 1's accumulators, lane for lane
  float* bias_s = reinterpret_cast<float*>(gsm + PLANES_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2, wv = wave & 3, gt = tid & 255;     // two groups of four waves: even / odd 64-k chunks
  // XCD-aware AND balanced tile order: workgroup ids round-robin over the 8 XCDs (own L2 each); XCD x takes the contiguous
  // run [x per, (x+1) per) of the row-major tile list, per = ceil(tiles / 8).  The column blocks of one A row block then share
  // an L2 (a row block is cut between two XCDs at most once) and no XCD gets more than `per` tiles: with whole row blocks per
  // XCD (the tiled kernel's order) P12's 36 x 7 tiles came out as 35 / 28 per XCD of 32 CUs -- a second round for 3 tiles.
  const int ncb = (g.N + TN - 1) / TN, tiles = ncb * ((g.M + TM - 1) / TM), per = (tiles + 7) / 8;
  const int lin = g.xcd_swizzle ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (lin >= tiles || (g.xcd_swizzle && (int)(blockIdx.x >> 3) >= per)) return;
  const int rblk = lin / ncb, cblk = lin - rblk * ncb;
  const int m0 = rblk * TM, n0 = cblk * TN;
#define PSTAMP(i)                                                                                                   \
  do {                                                                                                              \
    if (g.stamps && threadIdx.x == 0 && (blockIdx.x * 8) / gridDim.x != ((blockIdx.x - 1) * 8) / gridDim.x)         \
      g.stamps[((blockIdx.x * 8) / gridDim.x) * 8 + (i)] = clock64();                                                \
  } while (0)
  PSTAMP(0);
  const int nkc = g.bt_nkc;
  const __bf16* Bt = reinterpret_cast<const __bf16*>(g.Btiles);
  size_t toff[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) toff[jj] = (size_t)min(n0 / 16 + wv * NJ + jj, g.bt_ntile - 1) * nkc * 1024 + lane * 8;
  using SA = Stage<true, TM>;
  float ra[SA::NREG]; unsigned long long oka;
  const int nch = (g.K + BK2 - 1) / BK2;
  const int niter = (nch + 1) / 2;                    // both groups run the same number of rounds (the barriers are shared)
  f32x4 acc[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct BFr { bf16x8 h[NJ][2], l[NJ][2]; };
  // Every look-ahead load is UNCONDITIONAL (past the end: the last chunk again, with the A rows masked to zero): behind a
  // branch, the compiler's wait-count bookkeeping has to assume the path that issued nothing, and then waits for the NEWEST
  // loads -- the ones just requested -- before the products on the current fragments.
  auto load_b = [&](BFr& b, int c) __attribute__((always_inline)) {
    const int cc = min(c, nch - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const __bf16* t = Bt + toff[jj] + (size_t)min(2 * cc + ks, nkc - 1) * 1024;
        b.h[jj][ks] = *reinterpret_cast<const bf16x8*>(t);
        b.l[jj][ks] = *reinterpret_cast<const bf16x8*>(t + 512);
      }
  };
  auto load_a = [&](int c) __attribute__((always_inline)) {
    SA::load(ra, oka, g.A, g.sa_m, 1, m0, g.M, min(c, nch - 1) * BK2, g.K, gt, true);
    if (c >= nch) oka = 0ull;                         // no such chunk: the planes get zeros
  };
  const int aoff = (lane & 15) * LDB + 8 * (lane >> 4);
  auto products = [&](const __bf16* Ah, const __bf16* Al, const BFr& b, auto three_tag) __attribute__((always_inline)) {
    constexpr bool THREE = decltype(three_tag)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[RT], al[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDB + aoff + ks * 32);
        if (THREE) al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDB + aoff + ks * 32);
      }
      if (THREE) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.l[jj][ks], acc[rt][jj], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
    }
  };
  __bf16* Pg = Pb + (size_t)grp * 4 * PLANE;          // this group's two buffers
  // round i: the group's chunk 2 i + grp on buffer i & 1 with fragments `cur`; the chunk of round i+1 arrives meanwhile
  // (split-bf16 | plain bf16 is decided once, outside the loop: k_gemm_panel_wide below has the reason)
  auto step = [&](int i, const BFr& cur, BFr& nxt, auto three_tag) __attribute__((always_inline)) {
    __bf16* Ah = Pg + (size_t)(i & 1) * 2 * PLANE;
    load_b(nxt, 2 * (i + 1) + grp);
    products(Ah, Ah + PLANE, cur, three_tag);
    __bf16* Nh = Pg + (size_t)((i + 1) & 1) * 2 * PLANE;
    SA::template store<decltype(three_tag)::value>(ra, oka, Nh, Nh + PLANE, gt);   // round i+1's chunk
    load_a(2 * (i + 2) + grp);
    lds_barrier();                                    // buffer i & 1 is free for round i+2; buffer (i+1) & 1 is complete
  };
  BFr b0, b1;
  auto run = [&](auto three_tag) __attribute__((always_inline)) {
    SA::template store<decltype(three_tag)::value>(ra, oka, Pg, Pg + PLANE, gt);
    load_a(2 + grp);
    lds_barrier();
    PSTAMP(1);
    int i = 0;
    for (; i + 1 < niter; i += 2) { step(i, b0, b1, three_tag); step(i + 1, b1, b0, three_tag); }
    if (i < niter) step(i, b0, b1, three_tag);
  };
  load_a(grp);
  load_b(b0, grp);
  if (tid < TN) bias_s[tid] = (g.bias && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;
  if (!g.one_product) run(std::true_type{});
  else run(std::false_type{});
  PSTAMP(2);
  // group 1 hands its accumulators over, lane for lane; group 0 adds them (fixed order) and owns the epilogue's stage writes
  if (grp == 1) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        *reinterpret_cast<f32x4*>(red + ((((size_t)wv * RT + rt) * NJ + jj) * 64 + lane) * 4) = acc[rt][jj];
  }
  lds_barrier();
  if (grp == 0) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(red + ((((size_t)wv * RT + rt) * NJ + jj) * 64 + lane) * 4);
        acc[rt][jj] += o;
      }
  }
  epilogue_t<RT, NJ, 1, 4, 512>(g, acc, reinterpret_cast<float*>(gsm), bias_s, m0, n0, 0, wv, 0, tid, lane, grp == 0);
  PSTAMP(3);
#undef PSTAMP
}

// ------------------------------------------------------------------------------------------------
// Wide panel product (round 6): the same product on a 128 x 256 workgroup tile -- eight waves side by side, each ALL 128 rows x 32
// columns (RT = 8 row tiles x NJ = 2: 48 MFMAs per 32 k and wave), no split of K inside the workgroup.  Why: an ablation of
// k_gemm_panel at the 2048-wide shapes (profiles/r06_panel_ablation.txt) showed its parts ADD UP instead of overlapping -- MFMAs 49 %
// of the time, the fp32 -> split-bf16 conversion of A 23 %, the weight-fragment loads 14 %, the A loads 12 % -- and all but the
// MFMAs scale with 1 / tile width (A is converted once per COLUMN block) or 1 / tile height (a weight fragment feeds RT MFMAs per
// product).  Twice the width and twice the height halve all three per MFMA.  One workgroup per CU (82 KB of planes, <= 256
// registers at two waves per SIMD), so it pays only where the tile count fits the chip's 256 CUs well: panel_wide_pays() below.
// A is staged by both halves of the workgroup (threads 0-255: rows 0-63, 256-511: rows 64-127; double-buffered hi/lo planes, one
// LDS-only barrier per 64 k); the epilogue runs twice over 64-row halves (its transpose stage must fit the dead planes).
// Sums over k run in chunk order here (k_gemm_panel: even / odd chunks in two accumulator sets): same products, another fp32
// rounding order -- both within the split-bf16 bound the tests hold every product to.
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(512) void k_gemm_panel_wide(GemmArgs g) {
  RD_TOUCH_CODE_X(RD_TL_GEMM_PANEL_WIDE, blockIdx.x, 512);
  constexpr int RT = 8, TM = 16 * RT, HM = TM / 2, TN = 128 * NJ, PLANE = TM * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __bf16* Pb = reinterpret_cast<__bf16*>(gsm);                  // [2 buffers][hi, lo][TM][LDB]
  constexpr size_t PLANES_B = (size_t)4 * PLANE * sizeof(__bf16), STAGE_B = (size_t)HM * (TN + 4) * sizeof(float);
  static_assert(STAGE_B <= PLANES_B, "the epilogue's half-tile stage must fit the dead planes");
  float* bias_s = reinterpret_cast<float*>(gsm + PLANES_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = tid >> 8, gt = tid & 255;
  const int ncb = (g.N + TN - 1) / TN, tiles = ncb * ((g.M + TM - 1) / TM), per = (tiles + 7) / 8;     // XCD order: as k_gemm_panel
  const int lin = g.xcd_swizzle ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (lin >= tiles || (g.xcd_swizzle && (int)(blockIdx.x >> 3) >= per)) return;
  const int rblk = lin / ncb, cblk = lin - rblk * ncb;
  const int m0 = rblk * TM, n0 = cblk * TN;
  const int nkc = g.bt_nkc;
  const __bf16* Bt = reinterpret_cast<const __bf16*>(g.Btiles);
  size_t toff[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) toff[jj] = (size_t)min(n0 / 16 + wave * NJ + jj, g.bt_ntile - 1) * nkc * 1024 + lane * 8;
  using SA = Stage<true, HM>;                                   // each half of the workgroup stages its 64 rows
  float ra[SA::NREG]; unsigned long long oka;
  const int nch = (g.K + BK2 - 1) / BK2;
  f32x4 acc[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct BFr { bf16x8 h[NJ][2], l[NJ][2]; };
  auto load_b = [&](BFr& b, int c) __attribute__((always_inline)) {                            // unconditional look-ahead (past the end: the last chunk again)
    const int cc = min(c, nch - 1);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const __bf16* t = Bt + toff[jj] + (size_t)min(2 * cc + ks, nkc - 1) * 1024;
        b.h[jj][ks] = *reinterpret_cast<const bf16x8*>(t);
        b.l[jj][ks] = *reinterpret_cast<const bf16x8*>(t + 512);
      }
  };
  auto load_a = [&](int c) __attribute__((always_inline)) {
    SA::load(ra, oka, g.A, g.sa_m, 1, m0 + half * HM, g.M, min(c, nch - 1) * BK2, g.K, gt, true);
    if (c >= nch) oka = 0ull;
  };
  // `three_tag` (split-bf16: three products, hi and lo planes | plain bf16: one product, hi planes only) is decided ONCE, outside the
  // loop: with the branch inside, the products and the next chunk's conversion were separate basic blocks -- a burst of MFMAs, then a
  // burst of VALU -- and the ablation's parts added up; in one block the conversion issues in the MFMAs' shadow
  auto store_a = [&](__bf16* Ph, auto three_tag) __attribute__((always_inline)) {               // Ph: the buffer's hi plane; this half's rows
    __bf16* h = Ph + (size_t)half * HM * LDB;
    SA::template store<decltype(three_tag)::value>(ra, oka, h, h + PLANE, gt);
  };
  const int aoff = (lane & 15) * LDB + 8 * (lane >> 4);
  auto products = [&](const __bf16* Ah, const __bf16* Al, const BFr& b, auto three_tag) __attribute__((always_inline)) {
    constexpr bool THREE = decltype(three_tag)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[RT], al[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDB + aoff + ks * 32);
        if (THREE) al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDB + aoff + ks * 32);
      }
      if (THREE) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.l[jj][ks], acc[rt][jj], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
    }
  };
  // round i: chunk i on buffer i & 1 with fragments `cur`; chunk i+1 is converted into the other buffer, chunk i+2's loads leave
  auto step = [&](int i, const BFr& cur, BFr& nxt, auto three_tag) __attribute__((always_inline)) {
    __bf16* Ah = Pb + (size_t)(i & 1) * 2 * PLANE;
    load_b(nxt, i + 1);
    products(Ah, Ah + PLANE, cur, three_tag);
    store_a(Pb + (size_t)((i + 1) & 1) * 2 * PLANE, three_tag);
    load_a(i + 2);
    lds_barrier();
  };
  BFr b0, b1;
  auto run = [&](auto three_tag) __attribute__((always_inline)) {
    store_a(Pb, three_tag);
    load_a(1);
    lds_barrier();
    int i = 0;
    for (; i + 1 < nch; i += 2) { step(i, b0, b1, three_tag); step(i + 1, b1, b0, three_tag); }
    if (i < nch) step(i, b0, b1, three_tag);
  };
  load_a(0);
  load_b(b0, 0);
  if (tid < TN) bias_s[tid] = (g.bias && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;
  if (!g.one_product) run(std::true_type{});
  else run(std::false_type{});
  // epilogue over the two 64-row halves of the tile (the stage aliases the planes: every wave is past its last fragment read)
  float* stage = reinterpret_cast<float*>(gsm);
  epilogue_t<RT / 2, NJ, 1, 8, 512>(g, reinterpret_cast<f32x4(&)[RT / 2][NJ]>(acc[0]), stage, bias_s, m0, n0, 0, wave, 0, tid, lane);
  __syncthreads();
  epilogue_t<RT / 2, NJ, 1, 8, 512>(g, reinterpret_cast<f32x4(&)[RT / 2][NJ]>(acc[RT / 2]), stage, bias_s, m0 + HM, n0, 0, wave, 0, tid, lane);
}

// the panel form applies: pre-split weight tiles given, A k-contiguous with 16-byte rows, one plain pass
static bool panel_ok(const GemmArgs& g) {
  static const bool on = [] { const char* e = getenv("RD_GEMM_PANEL"); return !(e && atoi(e) == 0); }();
  return on && g.Btiles && g.sa_k == 1 && (g.sa_m & 3) == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0 && g.nsplit <= 1 &&
         g.nbatch <= 1 && !g.A2 && !g.rowsum && g.K >= 64;
}
// ------------------------------------------------------------------------------------------------
// Producer / consumer panel product (round 6): 128 x 128 workgroup tile, eight waves with TWO ROLES.  Waves 0-3 (one per SIMD) are
// consumers: each all 128 rows x 32 columns, weight fragments straight from L2, A fragments from LDS, nothing but MFMAs and fragment
// reads.  Waves 4-7 (the other wave of each SIMD) are producers: they load the NEXT 64-k chunk of A, convert it to split-bf16 and
// store the planes -- VALU work that issues between the consumer's MFMAs on the same SIMD instead of in front of them.  The ablation
// (profiles/r06_panel_ablation.txt) had shown conversion, loads and MFMAs of k_gemm_panel adding up; here they belong to different
// waves by construction.  One LDS-only barrier per chunk (double-buffered planes).  Tile 128 x 128: the conversion count per element
// of the 64 x 128 form, the weight-fragment traffic per MFMA of the wide one, and P12's 72 x 7 = 504 tiles fill two rounds of 256.
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(512) void k_gemm_panel_pc(GemmArgs g) {
  RD_TOUCH_CODE_X(RD_TL_GEMM_PANEL_PC, blockIdx.x, 512);
  constexpr int RT = 8, TM = 16 * RT, HM = TM / 2, TN = 64 * NJ, PLANE = TM * LDB;
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __bf16* Pb = reinterpret_cast<__bf16*>(gsm);                  // [2 buffers][hi, lo][TM][LDB]
  constexpr size_t PLANES_B = (size_t)4 * PLANE * sizeof(__bf16), STAGE_B = (size_t)HM * (TN + 4) * sizeof(float);
  static_assert(STAGE_B <= PLANES_B, "the epilogue's half-tile stage must fit the dead planes");
  float* bias_s = reinterpret_cast<float*>(gsm + PLANES_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), gt = tid & 255;
  const bool consumer = wave < 4;
  const int cw = wave & 3;
  const int ncb = (g.N + TN - 1) / TN, tiles = ncb * ((g.M + TM - 1) / TM), per = (tiles + 7) / 8;     // XCD order: as k_gemm_panel
  const int lin = g.xcd_swizzle ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (lin >= tiles || (g.xcd_swizzle && (int)(blockIdx.x >> 3) >= per)) return;
  const int rblk = lin / ncb, cblk = lin - rblk * ncb;
  const int m0 = rblk * TM, n0 = cblk * TN;
  const int nkc = g.bt_nkc;
  const int nch = (g.K + BK2 - 1) / BK2;
  if (tid < TN) bias_s[tid] = (g.bias && n0 + tid < g.N) ? g.bias[n0 + tid] : 0.f;
  f32x4 acc[RT][NJ];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!consumer) {
    // ---- producers: 256 threads stage 128 rows x 64 k per chunk; chunk c + 1 is in registers while chunk c is converted ----
    using SA = Stage<true, TM>;
    float ra[SA::NREG]; unsigned long long oka;
    auto load_a = [&](int c) __attribute__((always_inline)) {
      SA::load(ra, oka, g.A, g.sa_m, 1, m0, g.M, min(c, nch - 1) * BK2, g.K, gt, true);
      if (c >= nch) oka = 0ull;
    };
    auto run = [&](auto three_tag) __attribute__((always_inline)) {
      load_a(0);
      SA::template store<decltype(three_tag)::value>(ra, oka, Pb, Pb + PLANE, gt);
      load_a(1);
      lds_barrier();                                             // buffer 0 complete
      for (int i = 0; i < nch; ++i) {
        __bf16* Nh = Pb + (size_t)((i + 1) & 1) * 2 * PLANE;     // consumers read buffer i & 1 now; (i + 1) & 1 was released by the barrier
        SA::template store<decltype(three_tag)::value>(ra, oka, Nh, Nh + PLANE, gt);
        load_a(i + 2);
        lds_barrier();
      }
    };
    if (!g.one_product) run(std::true_type{}); else run(std::false_type{});
  } else {
    // ---- consumers ----
    const __bf16* Bt = reinterpret_cast<const __bf16*>(g.Btiles);
    size_t toff[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) toff[jj] = (size_t)min(n0 / 16 + cw * NJ + jj, g.bt_ntile - 1) * nkc * 1024 + lane * 8;
    struct BFr { bf16x8 h[NJ][2], l[NJ][2]; };
    auto load_b = [&](BFr& b, int c) __attribute__((always_inline)) {
      const int cc = min(c, nch - 1);
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const __bf16* t = Bt + toff[jj] + (size_t)min(2 * cc + ks, nkc - 1) * 1024;
          b.h[jj][ks] = *reinterpret_cast<const bf16x8*>(t);
          b.l[jj][ks] = *reinterpret_cast<const bf16x8*>(t + 512);
        }
    };
    const int aoff = (lane & 15) * LDB + 8 * (lane >> 4);
    auto products = [&](const __bf16* Ah, const __bf16* Al, const BFr& b, auto three_tag) __attribute__((always_inline)) {
      constexpr bool THREE = decltype(three_tag)::value;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDB + aoff + ks * 32);
          if (THREE) al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDB + aoff + ks * 32);
        }
        if (THREE) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.l[jj][ks], acc[rt][jj], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int jj = 0; jj < NJ; ++jj) acc[rt][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], b.h[jj][ks], acc[rt][jj], 0, 0, 0);
      }
    };
    BFr b0, b1;
    auto run = [&](auto three_tag) __attribute__((always_inline)) {
      load_b(b0, 0);
      lds_barrier();                                             // buffer 0 complete
      int i = 0;
      for (; i + 1 < nch; i += 2) {
        load_b(b1, i + 1);
        products(Pb, Pb + PLANE, b0, three_tag);
        lds_barrier();
        load_b(b0, i + 2);
        products(Pb + 2 * PLANE, Pb + 3 * PLANE, b1, three_tag);
        lds_barrier();
      }
      if (i < nch) {
        load_b(b1, i + 1);
        products(Pb, Pb + PLANE, b0, three_tag);
        lds_barrier();
      }
    };
    if (!g.one_product) run(std::true_type{}); else run(std::false_type{});
  }
  // epilogue over the two 64-row halves (the stage aliases the planes: behind the loop's last barrier nobody reads them).  All 512
  // threads store; the consumers (waves 0-3) hold the accumulators.
  float* stage = reinterpret_cast<float*>(gsm);
  epilogue_t<RT / 2, NJ, 1, 4, 512>(g, reinterpret_cast<f32x4(&)[RT / 2][NJ]>(acc[0]), stage, bias_s, m0, n0, 0, cw, 0, tid, lane, consumer);
  __syncthreads();
  epilogue_t<RT / 2, NJ, 1, 4, 512>(g, reinterpret_cast<f32x4(&)[RT / 2][NJ]>(acc[RT / 2]), stage, bias_s, m0 + HM, n0, 0, cw, 0, tid, lane, consumer);
}

// Which panel form.  All three run one workgroup per CU, so a launch costs (rounds over the chip's 256 CUs) x (time of one round), and
// the time of a round relative to the 64 x 128 form's was measured: 128 x 256 (four tiles' worth) 2.8-3.0, 128 x 128 producer /
// consumer (two tiles' worth) 1.6-1.8 (profiles/r06_panel_wide.txt).  A bigger tile has to be >= 5 % cheaper to be taken; its coarser
// quantisation is priced in by the rounds.  RD_PANEL_WIDE / RD_PANEL_PC = 0 / 1 exclude / force a form (read per call: tests, A/B);
// RD_PANEL_WIDE_COST / RD_PANEL_PC_COST (percent) override the measured ratios.
enum PanelForm { PANEL_SMALL = 0, PANEL_WIDE = 1, PANEL_PC = 2 };
static PanelForm panel_form(const GemmArgs& g) {
  const char* fw = getenv("RD_PANEL_WIDE");
  const char* fp = getenv("RD_PANEL_PC");
  const int force_w = fw ? atoi(fw) : -1, force_p = fp ? atoi(fp) : -1;
  static const int cost_w = [] { const char* e = getenv("RD_PANEL_WIDE_COST"); return e ? atoi(e) : 280; }();
  static const int cost_p = [] { const char* e = getenv("RD_PANEL_PC_COST"); return e ? atoi(e) : 180; }();
  if (force_p == 1) return PANEL_PC;
  if (force_w == 1) return PANEL_WIDE;
  auto rounds = [&](int tm, int tn) { return ((long)cdiv(g.M, tm) * cdiv(g.N, tn) + 255) / 256; };
  const long cs = rounds(64, 128) * 100, cw = rounds(128, 256) * cost_w, cp = rounds(128, 128) * cost_p;
  PanelForm best = PANEL_SMALL; long bc = cs * 95;                 // in units of 1 / 100: a bigger tile must beat 0.95 x small
  if (force_w != 0 && cw * 100 < bc) { best = PANEL_WIDE; bc = cw * 100; }
  if (force_p != 0 && cp * 100 < bc) { best = PANEL_PC; bc = cp * 100; }
  return best;
}
static int launch_panel_wide(const GemmArgs& g, hipStream_t st) {
  constexpr int NJ = 2, TM = 128, TN = 128 * NJ;
  const size_t lds = (size_t)4 * TM * LDB * sizeof(__bf16) + TN * sizeof(float);
  const dim3 grid(8 * cdiv(cdiv(g.M, TM) * cdiv(g.N, TN), 8));
  RD_LDS_ATTR((k_gemm_panel_wide<NJ>), lds);
  hipLaunchKernelGGL((k_gemm_panel_wide<NJ>), grid, dim3(512), lds, st, g);
  return check_launch("k_gemm_panel_wide");
}
static int launch_panel_pc(const GemmArgs& g, hipStream_t st) {
  constexpr int NJ = 2, TM = 128, TN = 64 * NJ;
  const size_t lds = (size_t)4 * TM * LDB * sizeof(__bf16) + TN * sizeof(float);
  const dim3 grid(8 * cdiv(cdiv(g.M, TM) * cdiv(g.N, TN), 8));
  RD_LDS_ATTR((k_gemm_panel_pc<NJ>), lds);
  hipLaunchKernelGGL((k_gemm_panel_pc<NJ>), grid, dim3(512), lds, st, g);
  return check_launch("k_gemm_panel_pc");
}
static int launch_panel(const GemmArgs& g, hipStream_t st) {
  const PanelForm form = panel_form(g);
  if (form == PANEL_WIDE) return launch_panel_wide(g, st);
  if (form == PANEL_PC) return launch_panel_pc(g, st);
  constexpr int NJ = 2, TM = 64, TN = 64 * NJ;
  const size_t lds = (size_t)8 * TM * LDB * sizeof(__bf16) + TN * sizeof(float);      // planes (stage and hand-over alias them) + bias
  const dim3 grid(8 * cdiv(cdiv(g.M, TM) * cdiv(g.N, TN), 8));
  RD_LDS_ATTR((k_gemm_panel<NJ>), lds);
  hipLaunchKernelGGL((k_gemm_panel<NJ>), grid, dim3(512), lds, st, g);
  return check_launch("k_gemm_panel");
}

// padded work / tile efficiency: bigger tiles re-read less and amortise the barrier, but waste
// more on ragged edges; pick the cheapest legal (MI, NI).
template <bool A_KC, bool B_KC>
int dispatch_bf16x3(const GemmArgs& g, hipStream_t st) {
  // (a 128 x 160 tile was a candidate until round 3: its two instantiations were the last kernels of the library with register
  // spills -- 656 bytes of scratch per lane -- and no shape of the path ever selected them)
  static const int cand[][2] = {{2, 2}, {4, 2}, {4, 4}, {2, 4}, {2, 5}};
  if (g.tile_hint == 1) return launch_bf16x3<A_KC, B_KC, 4, 4>(g, st);
  int best = 0; double bcost = 1e300;
  for (int c = 0; c < 5; ++c) {
    const int mi = cand[c][0], ni = cand[c][1];
    if (ni == 5 && !B_KC) continue;                  // 160-row staging only exists for k-contiguous operands
    const double tm = 32.0 * mi, tn = 32.0 * ni;
    const double padded = (double)cdiv(g.M, (int)tm) * tm * cdiv(g.N, (int)tn) * tn;
    const double blocks = (double)cdiv(g.M, (int)tm) * cdiv(g.N, (int)tn) * (g.nsplit > 1 ? g.nsplit : 1);
    // measured on MI355X: with this (unpipelined) main loop the 64x64 tile at 4 workgroups/CU beats
    // the fatter tiles at 1-2 workgroups/CU by 1.5-3x on every shape of the path, so bigger tiles
    // are only chosen when they are essentially free of padding AND the grid stays >= 8 blocks/CU.
    double cost = padded * (1.0 + 16.0 / tm + 16.0 / tn);
    if (blocks < 2048 && (mi > 2 || ni > 2)) cost *= 4.0;
    if (blocks < 192) cost *= 192.0 / blocks;        // do not starve the 256 CUs
    if (cost < bcost) { bcost = cost; best = c; }
  }
  switch (best) {
    case 0: return launch_bf16x3<A_KC, B_KC, 2, 2>(g, st);
    case 1: return launch_bf16x3<A_KC, B_KC, 4, 2>(g, st);
    case 2: return launch_bf16x3<A_KC, B_KC, 4, 4>(g, st);
    case 3: return launch_bf16x3<A_KC, B_KC, 2, 4>(g, st);
    default: return launch_bf16x3<A_KC, true, 2, 5>(g, st);
  }
}

__global__ void k_splitk_reduce2(const float* __restrict__ part, int nsplit, long stride, long e1,
                                 float* __restrict__ out1, long e2, float* __restrict__ out2) {
  const long elems = e1 + e2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < elems;
       i += (long)gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < nsplit; z += 4) {                     // 4 independent chains, combined in fixed order
      s0 += part[(long)z * stride + i]; s1 += part[(long)(z + 1) * stride + i];
      s2 += part[(long)(z + 2) * stride + i]; s3 += part[(long)(z + 3) * stride + i];
    }
    for (; z < nsplit; ++z) s0 += part[(long)z * stride + i];
    const float v = (s0 + s1) + (s2 + s3);
    if (i < e1) out1[i] = v; else out2[i - e1] = v;
  }
}

__global__ void k_splitk_reduce(const float* __restrict__ part, int nsplit, long elems,
                                float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < elems;
       i += (long)gridDim.x * blockDim.x) {
    float s = part[i];
    for (int z = 1; z < nsplit; ++z) s += part[(long)z * elems + i];   // fixed order
    out[i] = s;
  }
}

// column sums, stage 1: block (bx, by) sums rows [by*RPB, ..) of columns bx*64..+63 (coalesced),
// 4 row-groups per block combined through LDS in fixed order.
constexpr int CS_RPB = 512;
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ x, int M, int N,
                                                     long ldx, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CS_RPB, r1 = min(M, r0 + CS_RPB);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rg; r < r1; r += 4) s += x[(long)r * ldx + c];
  red[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < N)
    part[(long)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                     (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// single-stage column sum for short matrices: 1024 threads = 64 columns x 16 row groups, all of a
// thread's loads independent (deep memory-level parallelism), fixed-order LDS combine.
__global__ __launch_bounds__(1024) void k_colsum_small(const float* __restrict__ x, int M, int N, long ldx,
                                                       float* __restrict__ out, int n1, float* __restrict__ out2) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < N) {
    int r = rg;
    for (; r + 48 < M; r += 64) {
      s0 += x[(long)r * ldx + c]; s1 += x[(long)(r + 16) * ldx + c];
      s2 += x[(long)(r + 32) * ldx + c]; s3 += x[(long)(r + 48) * ldx + c];
    }
    for (; r < M; r += 16) s0 += x[(long)r * ldx + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < N) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][cl];
    if (c < n1) out[c] = v; else out2[c - n1] = v;
  }
}

}  // namespace

extern "C" void rd_debug_set_gemm_stamps(void* p) {      // not part of the ABI
  g_gemm_stamps = (unsigned long long*)p;
}

int launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return RD_OK;
  // a row-sum request rides on the row-contiguous staging; with a 1-wide operand both strides are 1
  const bool akc = (a.sa_k == 1) && !(a.rowsum && a.sa_m == 1), bkc = (a.sb_k == 1);
  if (!akc && a.sa_m != 1) return fail(RD_EINVAL, "gemm: A needs a unit stride");
  if (!bkc && a.sb_n != 1) return fail(RD_EINVAL, "gemm: B needs a unit stride");
  if (a.nbatch > 1 && (a.nsplit > 1 || a.A2 || a.rowsum || a.scatter || a.batch_inner < 1))
    return fail(RD_EINVAL, "gemm: the batched form takes plain single-pass products only");
  // the token plan's scatter exists on the straight epilogue of the bf16 kernels only (16-byte cells, no mask / residual / dropout)
  if (a.sp_row0 && !(a.scatter && a.sd == 4 && (a.N & 3) == 0 && (a.ldz & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                     !a.posmask && !a.residual && !(a.drop_p > 0.f) && a.nsplit <= 1 && a.sp_len && precision() != RD_PREC_FP32))
    return fail(RD_EUNSUPPORTED, "gemm: the plan-following scatter needs d_ob = 4, 16-byte cells and a plain epilogue in a bf16 mode");
  dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nbatch > 1 ? a.nbatch : (a.nsplit > 1 ? a.nsplit : 1));
  GemmArgs g = a;
  g.seed_cell = seed_cell();
  g.stamps = g_gemm_stamps;
  static const int xcd_env = [] { const char* e = getenv("RD_GEMM_XCD"); return e ? atoi(e) : 1; }();
  g.xcd_swizzle = xcd_env;
  if (g.nsplit <= 1) { g.nsplit = 1; g.k_per_split = g.K > 0 ? g.K : 1; g.sc_split = 0; }
  if (precision() != RD_PREC_FP32) {
    g.one_product = precision() == RD_PREC_BF16;
    static const int sx_env = [] { const char* e = getenv("RD_SPLITK_XCD"); return e ? atoi(e) : 1; }();
    g.slice_xcd = (g.nsplit > 1 && sx_env) ? 1 : 0;
    if (g.nsplit > 1 && (g.k_per_split % BK2) != 0) return fail(RD_EINVAL, "gemm: k_per_split must be a multiple of 64");
    if (panel_ok(g)) return launch_panel(g, st);
    if (akc && bkc) return dispatch_bf16x3<true, true>(g, st);
    if (akc && !bkc) return dispatch_bf16x3<true, false>(g, st);
    if (!akc && bkc) return dispatch_bf16x3<false, true>(g, st);
    return dispatch_bf16x3<false, false>(g, st);
  }
  if (akc && bkc) hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, st, g);
  else if (akc && !bkc) hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, st, g);
  else if (!akc && bkc) hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, st, g);
  return check_launch("k_gemm");
}

// Weight-gradient products are HBM-traffic-bound (PMC: the 64x64-tile form fetches 4x its operands).
// RD_WGRAD_TILE=128 runs them on 128x128 tiles (half the re-reads) -- measured SLOWER with the present
// unpipelined main loop (K1 backward 79 -> 89 us: 2 workgroups/CU cannot cover the load latency), so
// 64 stays the default until the main loop is software-pipelined.
static int wgrad_tile() {
  static const int t = [] { const char* e = getenv("RD_WGRAD_TILE"); const int v = e ? atoi(e) : 64;
                            return v == 128 ? 128 : 64; }();
  return t;
}

static int g_splitk_want = [] { const char* e = getenv("RD_SPLITK_WANT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 640; }();
extern "C" void rd_debug_set_splitk_want(int v) { g_splitk_want = v > 0 ? v : 640; }   // not part of the ABI

int splitk_plan(long red, int rows, int cols, int* k_per_split) {
  const int tile = (rows >= 96 && cols >= 96) ? wgrad_tile() : 64;
  const int tiles = cdiv(rows, tile) * cdiv(cols, tile);
  const int want = cdiv(tile == 128 ? 256 : g_splitk_want, tiles);  // workgroups over the whole product (640: ~2.5 per CU;
                                                                      // in-step A/B 320 -> 640: 1.182 -> 1.148 ms/step, 1024: 1.144)
  const int r = red > 0 ? (int)red : 1;
  int per = (int)align_up((size_t)cdiv(r, want), 64);
  *k_per_split = per;
  return cdiv(r, per);
}

// dW[N,K] = dy[M,N]^T x[M,K] and db[N] = sum_m dy[m,:] with ONE pass over dy: the split-K product
// accumulates the row sums of its A operand (= dy^T) on the side.  ws: nsplit*(N*K + N) floats.
long wgrad_ws_floats(long M, int N, int K) {
  int kps; const int ns = splitk_plan(M, N, K, &kps);
  return (long)ns * ((long)N * K + N);
}
int launch_wgrad(long M, int N, int K, const float* dy, long lddy, const float* x, long ldx, float* dW,
                 float* db, float* ws, hipStream_t st) {
  int kps; const int ns = splitk_plan(M, N, K, &kps);
  GemmArgs t{};
  t.M = N; t.N = K; t.K = (int)M;
  t.A = dy; t.sa_m = 1; t.sa_k = lddy;
  t.B = x; t.sb_n = 1; t.sb_k = ldx;
  t.nsplit = ns; t.k_per_split = kps;
  t.tile_hint = (N >= 96 && K >= 96 && wgrad_tile() == 128) ? 1 : 0;
  const long stride = (long)N * K + N;                 // split z: [dW partial | db partial]
  int rc;
  if (ns > 1) {
    t.C = ws; t.sc_m = K; t.sc_split = stride;
    t.rowsum = db ? ws + (long)N * K : nullptr; t.rowsum_split = stride;
    if ((rc = launch_gemm(t, st))) return rc;
    return launch_splitk_reduce_pair(ws, dW, db, nullptr, nullptr, nullptr, ns, stride, (long)N * K, N, st);
  }
  t.C = dW; t.sc_m = K; t.rowsum = db; t.rowsum_split = 0;
  return launch_gemm(t, st);
}

int launch_wgrad2(long M, int N, int K, const float* dyA, const float* xA, float* dWA, float* dbA,
                  const float* dyB, const float* xB, float* dWB, float* dbB, float* ws, hipStream_t st) {
  int kps; const int ns = splitk_plan(M, N, K, &kps);
  const long stride = (long)N * K + N;
  if (ns <= 1 || precision() == RD_PREC_FP32) {        // the batched form exists for the bf16 MFMA kernel only
    int rc = launch_wgrad(M, N, K, dyA, N, xA, K, dWA, dbA, ws, st);
    if (rc) return rc;
    return launch_wgrad(M, N, K, dyB, N, xB, K, dWB, dbB, ws + (long)ns * stride, st);
  }
  float* wsB = ws + (long)ns * stride;
  GemmArgs t{};
  t.M = N; t.N = K; t.K = (int)M;
  t.A = dyA; t.sa_m = 1; t.sa_k = N;
  t.B = xA; t.sb_n = 1; t.sb_k = K;
  t.nsplit = ns; t.k_per_split = kps;
  t.tile_hint = (N >= 96 && K >= 96 && wgrad_tile() == 128) ? 1 : 0;
  t.C = ws; t.sc_m = K; t.sc_split = stride;
  t.rowsum = ws + (long)N * K; t.rowsum_split = stride;
  t.A2 = dyB; t.B2 = xB; t.C2 = wsB; t.rowsum2 = wsB + (long)N * K;
  int rc;
  if ((rc = launch_gemm(t, st))) return rc;
  return launch_splitk_reduce_pair(ws, dWA, dbA, wsB, dWB, dbB, ns, stride, (long)N * K, N, st);
}

int launch_splitk_reduce2(const float* part, int nsplit, long stride, long e1, float* out1, long e2, float* out2,
                          hipStream_t st) {
  const long elems = e1 + e2;
  if (elems <= 0) return RD_OK;
  int blocks = (int)((elems + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_splitk_reduce2, dim3(blocks), dim3(256), 0, st, part, nsplit, stride, e1, out1, e2, out2);
  return check_launch("k_splitk_reduce2");
}

int launch_splitk_reduce(const float* part, int nsplit, long elems, float* out, hipStream_t st) {
  if (elems <= 0) return RD_OK;
  int blocks = (int)((elems + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, st, part, nsplit, elems, out);
  return check_launch("k_splitk_reduce");
}

int launch_colsum2(const float* x, int M, int N, long ldx, float* out1, int n1, float* out2, float* ws, hipStream_t st) {
  if (M <= 8192) {
    hipLaunchKernelGGL(k_colsum_small, dim3(cdiv(N, 64)), dim3(1024), 0, st, x, M, N, ldx, out1, n1, out2);
    return check_launch("k_colsum_small");
  }
  int rc = launch_colsum(x, M, N, ldx, ws, ws + N, st);          // long matrices: two-stage, then split
  if (rc) return rc;
  RD_HIP(hipMemcpyAsync(out1, ws, sizeof(float) * n1, hipMemcpyDeviceToDevice, st));
  RD_HIP(hipMemcpyAsync(out2, ws + n1, sizeof(float) * (N - n1), hipMemcpyDeviceToDevice, st));
  return RD_OK;
}

long colsum_ws_floats(int M, int N) { return (long)cdiv(M, CS_RPB) * N; }

int launch_colsum(const float* x, int M, int N, long ldx, float* out, float* ws, hipStream_t st) {
  if (N <= 0) return RD_OK;
  if (M <= 8192) {
    hipLaunchKernelGGL(k_colsum_small, dim3(cdiv(N, 64)), dim3(1024), 0, st, x, M, N, ldx, out, N, (float*)nullptr);
    return check_launch("k_colsum_small");
  }
  const int nby = cdiv(M, CS_RPB);
  hipLaunchKernelGGL(k_colsum_part, dim3(cdiv(N, 64), nby), dim3(256), 0, st, x, M, N, ldx, ws);
  int rc = check_launch("k_colsum_part");
  if (rc) return rc;
  return launch_splitk_reduce(ws, nby, N, out, st);
}

namespace {
// ---- wide fixed-order reduce: 64 element quads x 16 split groups per workgroup ---------------------
struct RedJob { const float* part; float* out1; float* out2; };
struct RedArgs { RedJob j[2]; int nsplit; long stride, e1, e2; };

__global__ __launch_bounds__(1024) void k_reduce_wide(RedArgs a) {
  __shared__ float4 red[16][64];
  const RedJob job = a.j[blockIdx.y];
  const int ql = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const long q = (long)blockIdx.x * 64 + ql, nq = (a.e1 + a.e2) >> 2;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < nq) {
#pragma unroll 4
    for (int z = sg; z < a.nsplit; z += 16) {
      const float4 t = *reinterpret_cast<const float4*>(job.part + (long)z * a.stride + 4 * q);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[sg][ql] = s;
  __syncthreads();
  if (sg == 0 && q < nq) {
    float4 t = red[0][ql];
#pragma unroll
    for (int g = 1; g < 16; ++g) { const float4 u = red[g][ql]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const long i = 4 * q;
    if (i < a.e1) *reinterpret_cast<float4*>(job.out1 + i) = t;
    else *reinterpret_cast<float4*>(job.out2 + (i - a.e1)) = t;
  }
}

}  // namespace

// fixed-order sum of `nsplit` partials ([e1 | e2] floats each, `stride` apart) for one or two problems:
// the wide kernel (16 split groups x float4) when the layout allows 16-byte accesses, else the scalar one
int launch_splitk_reduce_pair(const float* partA, float* out1A, float* out2A, const float* partB, float* out1B,
                              float* out2B, int nsplit, long stride, long e1, long e2, hipStream_t st) {
  const bool two = partB != nullptr;
  const bool same_e2 = !two || ((out2A != nullptr) == (out2B != nullptr));
  const long e2a = out2A ? e2 : 0;
  static const bool wide_on = [] { const char* e = getenv("RD_REDUCE_WIDE"); return !(e && atoi(e) == 0); }();
  const bool vec = wide_on && same_e2 && ((stride | e1 | e2a) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(partA) | reinterpret_cast<uintptr_t>(out1A) | reinterpret_cast<uintptr_t>(out2A) |
                     reinterpret_cast<uintptr_t>(partB) | reinterpret_cast<uintptr_t>(out1B) | reinterpret_cast<uintptr_t>(out2B)) & 15) == 0;
  if (!vec) {
    int rc = launch_splitk_reduce2(partA, nsplit, stride, e1, out1A, out2A ? e2 : 0, out2A, st);
    if (rc || !two) return rc;
    return launch_splitk_reduce2(partB, nsplit, stride, e1, out1B, out2B ? e2 : 0, out2B, st);
  }
  RedArgs r{};
  r.j[0] = RedJob{partA, out1A, out2A};
  r.j[1] = RedJob{partB, out1B, out2B};
  r.nsplit = nsplit; r.stride = stride; r.e1 = e1; r.e2 = e2a;
  const long nq = (r.e1 + r.e2) >> 2;
  if (nq <= 0) return RD_OK;
  hipLaunchKernelGGL(k_reduce_wide, dim3((unsigned)((nq + 63) / 64), two ? 2 : 1), dim3(1024), 0, st, r);
  return check_launch("k_reduce_wide");
}

}  // namespace rd
