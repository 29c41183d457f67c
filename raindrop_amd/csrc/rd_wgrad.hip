// rd_wgrad.hip -- weight-gradient product for tall-skinny layers, operands read ONCE.
//
//   dW[N,K] = dy[M,N]^T x[M,K],  db[N] = sum_m dy[m,:]        (M = tokens or nodes, N,K <= a few hundred)
//
// replaces autograd's `grad_weight = grad_output.t().mm(input)` / `grad_bias = grad_output.sum(0)` of
// every nn.Linear on the path (code/models_rd.py:235-237 encoder layers, code/Ob_propagation.py:80
// lin_value).  The tiled split-K form (rd_gemm.hip) gives each workgroup a 64x64 output tile, so both
// operands are re-fetched once per tile column/row (PMC: 3.7x the operand bytes).  Here a workgroup owns
// a SLAB of rows and the WHOLE output (or an n-block of it): each operand element is loaded exactly once,
// the [n-block x K] accumulator lives in the registers of 8 waves (TN x TK 16x16 tiles per wave), and
// the per-slab partials are combined by a wide fixed-order reduce (no floating-point atomics).
//
// Per 64-row slab: thread c owns operand COLUMN c (dy columns first, then x columns): 64 independent
// dword loads (one HBM round trip, each wave instruction a contiguous 256-byte row segment), split into
// bf16 hi/lo and written as eight 16-byte LDS stores per plane -- the planes are [column][row], i.e.
// already the k-contiguous layout the MFMA operands want, so the "transpose" of dy^T costs nothing.
// The next slab's loads are in flight while the MFMAs of the current one run.
#include "rd_common.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_THR = 512, WG_SLAB = 64;
constexpr int WG_LD = WG_SLAB + 8;          // bf16 per plane row: 144 B, odd multiple of 16 B -> conflict-free b128

static unsigned long long* g_wg_stamps = nullptr;   // debug only (tools/wgrad_timing.py); passed as a kernel argument
#define WGSTAMP(i)                                                                          \
  do {                                                                                      \
    if (a.stamps && blockIdx.x < 8 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)   \
      a.stamps[a.cfg * 128 + blockIdx.x * 16 + (i)] = clock64();                                          \
  } while (0)

struct WgArgs {
  const float *dy, *x;
  const float *dy2, *x2;                    // second problem of identical shape (blockIdx.z == 1) or null
  float *part, *part2;                      // per slab group g: [N*K dW partial | N db partial], `stride` floats apart
  long lddy, ldx, stride;
  int M, N, K;
  int rows_per_wg;                          // multiple of 64
  int nb_tiles;                             // 16-row n tiles per n-block (blockIdx.y)
  int want_rowsum;
  int cfg;                                  // configuration index (debug stamps)
  unsigned long long* stamps;
};

template <int TN, int TK, int WN, int WK>
__global__ __launch_bounds__(WG_THR) void k_wgrad_slab(WgArgs a) {
  static_assert(WN * WK == WG_THR / 64, "eight waves");
  constexpr int NB = WN * TN * 16;          // plane rows reserved for dy columns
  constexpr int KB = WK * TK * 16;          // plane rows reserved for x columns
  extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
  __bf16* Ph = reinterpret_cast<__bf16*>(wsm);                 // [(NB + KB)][WG_LD]
  __bf16* Pl = Ph + (NB + KB) * WG_LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave / WK, wk = wave - wn * WK;
  if (blockIdx.z) { a.dy = a.dy2; a.x = a.x2; a.part = a.part2; }
  const int n0 = blockIdx.y * a.nb_tiles * 16;
  const int nvalid = min(a.nb_tiles * 16, a.N - n0);
  const int mbeg = blockIdx.x * a.rows_per_wg, mend = min(a.M, mbeg + a.rows_per_wg);

  // operand columns: whole waves take dy columns (64 per wave), the remaining waves take x columns, so
  // base pointer and row stride are wave-uniform (scalar address arithmetic, one 32-bit lane offset)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int dyw = (nvalid + 63) >> 6;
  const bool is_dy = wave_u < dyw;
  const int col = (is_dy ? wave_u : wave_u - dyw) * 64 + lane;
  const int ncols = is_dy ? nvalid : a.K;
  const bool active = col < ncols;
  const int colc = active ? col : ncols - 1;                 // idle lanes re-read the last column (no branch)
  const long ld = is_dy ? a.lddy : a.ldx;
  const float* base = is_dy ? a.dy + n0 : a.x;
  const int prow = is_dy ? col : NB + col;

  float v[WG_SLAB];
  auto load_slab = [&](int m0) {
    const float* p = base + (long)m0 * ld;
    if (m0 + WG_SLAB <= mend) {
#pragma unroll
      for (int r = 0; r < WG_SLAB; ++r) { v[r] = p[colc]; p += ld; }
    } else {
#pragma unroll
      for (int r = 0; r < WG_SLAB; ++r) { v[r] = (m0 + r < mend) ? p[colc] : 0.f; p += ld; }
    }
  };

  f32x4 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rsum = 0.f;

  const int foff = (lane & 15) * WG_LD + 8 * (lane >> 4);
  const __bf16* Ah = Ph + wn * TN * 16 * WG_LD + foff;
  const __bf16* Al = Pl + wn * TN * 16 * WG_LD + foff;
  const __bf16* Bh = Ph + (NB + wk * TK * 16) * WG_LD + foff;
  const __bf16* Bl = Pl + (NB + wk * TK * 16) * WG_LD + foff;

  WGSTAMP(0);
  load_slab(mbeg);
  for (int m0 = mbeg; m0 < mend; m0 += WG_SLAB) {
    if (active) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int o = 0; o < WG_SLAB / 8; ++o) {
        bf16x8 h, l;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float xv = v[8 * o + c];
          h[c] = (__bf16)xv;
          l[c] = (__bf16)(xv - (float)h[c]);
          if (c & 1) s1 += xv; else s0 += xv;
        }
        *reinterpret_cast<bf16x8*>(Ph + prow * WG_LD + 8 * o) = h;
        *reinterpret_cast<bf16x8*>(Pl + prow * WG_LD + 8 * o) = l;
      }
      rsum += s0 + s1;
    }
    if (m0 == mbeg) WGSTAMP(1);
    __syncthreads();
    if (m0 == mbeg) WGSTAMP(2);
    if (m0 + WG_SLAB < mend) load_slab(m0 + WG_SLAB);          // in flight during the MFMAs
#pragma unroll
    for (int ch = 0; ch < WG_SLAB / 32; ++ch) {
      bf16x8 ah[TN], al[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + i * 16 * WG_LD + ch * 32);
        al[i] = *reinterpret_cast<const bf16x8*>(Al + i * 16 * WG_LD + ch * 32);
      }
#pragma unroll
      for (int j = 0; j < TK; ++j) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + j * 16 * WG_LD + ch * 32);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + j * 16 * WG_LD + ch * 32);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
      }
    }
    if (m0 == mbeg) WGSTAMP(3);
    __syncthreads();
    if (m0 == mbeg) WGSTAMP(4);
  }
  WGSTAMP(5);

  // ---- partial of this slab group: dW rows n0.., all K columns; accumulator element r of tile (i,j) is
  // row 4*(lane>>4)+r, column lane&15
  float* out = a.part + (long)blockIdx.x * a.stride;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int nl = (wn * TN + i) * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const int k = (wk * TK + j) * 16 + (lane & 15);
      if (k < a.K) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nl + r < nvalid) out[(long)(n0 + nl + r) * a.K + k] = acc[i][j][r];
      }
    }
  }
  if (a.want_rowsum && is_dy && active) out[(long)a.N * a.K + n0 + col] = rsum;
  WGSTAMP(6);
}

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

struct Cfg { int TN, TK, WN, WK; };
constexpr Cfg CFGS[] = {{5, 5, 2, 4}, {5, 5, 4, 2}, {5, 3, 2, 4}, {4, 5, 4, 2}, {4, 4, 2, 4}};
constexpr int NCFG = sizeof(CFGS) / sizeof(CFGS[0]);

// 64-row slabs per workgroup; 0 = off (the default).  MEASURED (P19, B=256, same box, hipGraph step): tiled split-K
// 1.41 ms/step, this form 1.59 (2 slabs) / 1.56 (3) / 1.65 (1).  Per-phase stamps inside the step
// (tools/wgrad_in_step.py): first load+split 11.5 k cycles, MFMA 8.5 k per slab, partial store 9 k -- with the
// whole output in registers there is ONE workgroup per CU and nothing overlaps those phases; the tiled form
// re-reads operands 3.7x but keeps 4 workgroups per CU in flight.  Kept (tested, off) as the starting point
// for a wave-specialised version that loads/splits and multiplies concurrently.
int g_slabs = [] { const char* e = getenv("RD_WGRAD_SLABS"); return e ? atoi(e) : 0; }();

// cheapest configuration (in 16x16 tile slots) that holds the product; -1: none
int pick_cfg(int N, int K, int* nblocks, int* nb_tiles) {
  int best = -1, best_cost = 0;
  const int nt = cdiv(N, 16);
  for (int c = 0; c < NCFG; ++c) {
    const int nbt = CFGS[c].WN * CFGS[c].TN, kbt = CFGS[c].WK * CFGS[c].TK;
    if (K > kbt * 16) continue;
    const int nb = cdiv(nt, nbt);
    const int per = cdiv(nt, nb);                       // balanced n-blocks
    if (cdiv(per * 16, 64) + cdiv(K, 64) > WG_THR / 64) continue;   // one operand column per thread, whole waves per operand
    const int cost = nb * nbt * kbt;
    if (best < 0 || cost < best_cost) { best = c; best_cost = cost; *nblocks = nb; *nb_tiles = per; }
  }
  return best;
}

template <int TN, int TK, int WN, int WK>
int launch_cfg(const WgArgs& a, dim3 grid, hipStream_t st) {
  const size_t lds = (size_t)2 * (WN * TN + WK * TK) * 16 * WG_LD * sizeof(__bf16);
  RD_LDS_ATTR((k_wgrad_slab<TN, TK, WN, WK>), lds);
  hipLaunchKernelGGL((k_wgrad_slab<TN, TK, WN, WK>), grid, dim3(WG_THR), lds, st, a);
  return check_launch("k_wgrad_slab");
}

}  // namespace

extern "C" void rd_debug_set_wgrad_slabs(int v) { g_slabs = v; }   // not part of the ABI
extern "C" void rd_debug_set_wgrad_stamps(void* p) {               // not part of the ABI
  g_wg_stamps = (unsigned long long*)p;
}

bool wgrad_slab_ok(long M, int N, int K) {
  int nb, per;
  return g_slabs > 0 && precision() == RD_PREC_BF16X3 && M >= 1024 && M < (1L << 30) && (N % 4) == 0 && N >= 16 && K >= 16 &&
         pick_cfg(N, K, &nb, &per) >= 0;
}
static int slab_groups(long M) { return cdiv((int)M, WG_SLAB * g_slabs); }
long wgrad_slab_ws_floats(long M, int N, int K) { return (long)slab_groups(M) * ((long)N * K + N); }

// one or two (dy2 != null) products of identical shape; ws: (dy2 ? 2 : 1) * wgrad_slab_ws_floats floats, 16-byte aligned
int launch_wgrad_slab(long M, int N, int K, const float* dy, long lddy, const float* x, long ldx, float* dW, float* db,
                      const float* dy2, const float* x2, float* dW2, float* db2, float* ws, hipStream_t st) {
  int nblocks = 1, nb_tiles = 1;
  const int c = pick_cfg(N, K, &nblocks, &nb_tiles);
  if (c < 0) return fail(RD_EINVAL, "wgrad_slab: no configuration for N=%d K=%d", N, K);
  const int G = slab_groups(M);
  const long stride = (long)N * K + N;
  WgArgs a{};
  a.dy = dy; a.x = x; a.dy2 = dy2; a.x2 = x2; a.lddy = lddy; a.ldx = ldx;
  a.part = ws; a.part2 = ws + (long)G * stride; a.stride = stride;
  a.M = (int)M; a.N = N; a.K = K; a.rows_per_wg = WG_SLAB * g_slabs; a.nb_tiles = nb_tiles;
  a.want_rowsum = (db != nullptr || db2 != nullptr) ? 1 : 0;
  a.cfg = c; a.stamps = g_wg_stamps;
  const dim3 grid(G, nblocks, dy2 ? 2 : 1);
  int rc;
  switch (c) {
    case 0: rc = launch_cfg<5, 5, 2, 4>(a, grid, st); break;
    case 1: rc = launch_cfg<5, 5, 4, 2>(a, grid, st); break;
    case 2: rc = launch_cfg<5, 3, 2, 4>(a, grid, st); break;
    case 3: rc = launch_cfg<4, 5, 4, 2>(a, grid, st); break;
    default: rc = launch_cfg<4, 4, 2, 4>(a, grid, st); break;
  }
  if (rc) return rc;
  return launch_splitk_reduce_pair(a.part, dW, db, dy2 ? a.part2 : nullptr, dW2, db2, G, stride, (long)N * K, N, st);
}

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
