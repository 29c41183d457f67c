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
#include "rd_common.h"
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

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int z = blockIdx.z;
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
  if (kbeg < kend) {
    stage_load<A_KC>(ra, g.A, g.sa_m, g.sa_k, m0, g.M, kbeg, kend, tid, a_vec);
    stage_load<B_KC>(rb, g.B, g.sb_n, g.sb_k, n0, g.N, kbeg, kend, tid, b_vec);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
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

  // ---- epilogue: acc[i][j][r] is C(m, n) with m = .. + 4*(lane>>4) + r, n = .. + (lane&15)
  float* Cz = g.C + (long)z * g.sc_split;
  const bool raw = g.nsplit > 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wx * 32 + j * 16 + (lane & 15);
      if (n >= g.N) continue;
      const float bias = (!raw && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wy * 32 + i * 16 + 4 * (lane >> 4) + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r];
        if (!raw) {
          v += bias;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.rowscale) v *= g.rowscale[m % g.rs_period];
          if (g.posmask) v = (g.posmask[(long)m * g.pm_m + n] > 0.f) ? v : 0.f;
          if (g.cscale != 0.f) v *= g.cscale;
          if (g.drop_p > 0.f)
            v *= dropout_scale(g.drop_seed, g.drop_site, (uint64_t)m * g.N + n, g.drop_p,
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

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0) return RD_OK;
  const bool akc = (a.sa_k == 1), bkc = (a.sb_k == 1);
  if (!akc && a.sa_m != 1) return fail(RD_EINVAL, "gemm: A needs a unit stride");
  if (!bkc && a.sb_n != 1) return fail(RD_EINVAL, "gemm: B needs a unit stride");
  dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.nsplit > 1 ? a.nsplit : 1);
  GemmArgs g = a;
  if (g.nsplit <= 1) { g.nsplit = 1; g.k_per_split = g.K > 0 ? g.K : 1; g.sc_split = 0; }
  if (akc && bkc) hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, st, g);
  else if (akc && !bkc) hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, st, g);
  else if (!akc && bkc) hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, st, g);
  return check_launch("k_gemm");
}

int launch_splitk_reduce(const float* part, int nsplit, long elems, float* out, hipStream_t st) {
  if (elems <= 0) return RD_OK;
  int blocks = (int)((elems + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, st, part, nsplit, elems, out);
  return check_launch("k_splitk_reduce");
}

long colsum_ws_floats(int M, int N) { return (long)cdiv(M, CS_RPB) * N; }

int launch_colsum(const float* x, int M, int N, long ldx, float* out, float* ws, hipStream_t st) {
  if (N <= 0) return RD_OK;
  const int nby = cdiv(M, CS_RPB);
  hipLaunchKernelGGL(k_colsum_part, dim3(cdiv(N, 64), nby), dim3(256), 0, st, x, M, N, ldx, ws);
  int rc = check_launch("k_colsum_part");
  if (rc) return rc;
  return launch_splitk_reduce(ws, nby, N, out, st);
}

}  // namespace rd
