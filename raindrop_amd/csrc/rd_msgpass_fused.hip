// rd_msgpass_fused.hip -- kernel K1, fused LDS-resident form for small sensor graphs
// (F <= 64 sensors, K = T*d_ob <= 256 and a multiple of 16: the P19 shape, K = 240).
//
// One workgroup owns one sample.  Its sensor graph node features X [F, K] are built in LDS straight
// from src (observation embedding, code/models_rd.py:290-296 + the [T,F*d] -> [F,T*d] re-layout of
// :326-327), both Observation_progation layers run back to back with the layer-1 output never
// leaving the CU, and the result is written in the [T,B,D] layout the temporal stage consumes
// (code/models_rd.py:338-342) -- so the ~40 launches per sample of the reference loop
// (code/models_rd.py:322-343, code/Ob_propagation.py:157-228) become one launch per batch.
//
// Arithmetic: the two K x K contractions use split-bf16 on v_mfma_f32_16x16x32_bf16
// (x = hi + lo in bf16; hi*hi + hi*lo + lo*hi, fp32 accumulate).  The weights are split ONCE per
// step by k_wprep into bf16 planes (both orientations, K padded to 256) that stay L2-resident and
// are streamed by every workgroup as MFMA B operands; activations are split when they are
// written to LDS.  Per sample the MFMA work is (RT*16 rows) x K x 256 x 3 products x 2 layers.
//
// Layout in LDS (RT = ceil(F/16) row tiles): four bf16 planes [RT*16][264] (X hi/lo, Y1 hi/lo;
// 528-B rows keep ds_read_b128 conflict-free) plus an fp32 staging tile aliased onto the X planes
// for the coalesced [F,K] <-> [T,F*d] transposes.
#include "rd_common.h"
#include "rd_rng.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KP = 256;          // padded reduction length of every plane
constexpr int LDX = KP + 8;      // bf16 elements per LDS plane row (528 B)
constexpr int LDS_F = 244;
constexpr int NTHR = 512;         // 8 wavefronts: wave w owns output column tiles {w, w+8}
constexpr int NWAVE = NTHR / 64, NJ = 2;       // fp32 staging row stride (conflict-free for both access orders)

struct FusedArgs {
  const float *src, *R_u, *b1, *b2, *ssum;
  const float *times, *tscale; const int64_t* lengths; uint8_t* mask; int d_pe;   // optional PE / mask (fwd)
  const __bf16* wplanes;         // [layer 2][orient 2][hi/lo 2][K rows][KP]
  float *xsave, *y1save, *z;
  const float *dz;               // bwd
  float *dz2save, *dz1save, *rupart;
  int B, T, F, d, K, ldz;
  float p_drop; uint64_t seed; const uint64_t* seed_cell;
  unsigned long long* stamps;    // debug: per-phase clock64() of wave 0 of the first 8 workgroups
};

#define RD_STAMP(i)                                                                              \
  do {                                                                                           \
    if (a.stamps && blockIdx.x < 8 && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (i)] = clock64(); \
  } while (0)

__device__ __forceinline__ const __bf16* plane(const FusedArgs& a, int layer, int orient, int part) {
  return a.wplanes + ((size_t)((layer * 2 + orient) * 2 + part)) * a.K * KP;
}

__device__ __forceinline__ void split_store4(__bf16* ph, __bf16* pl, const float (&v)[4]) {
  bf16x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) { h[j] = (__bf16)v[j]; l[j] = (__bf16)(v[j] - (float)h[j]); }
  *reinterpret_cast<bf16x4*>(ph) = h;
  *reinterpret_cast<bf16x4*>(pl) = l;
}

// W [K,K] fp32 -> four bf16 planes: orient 0 rows n (k contiguous, == W), orient 1 rows k (== W^T).
__global__ __launch_bounds__(256) void k_wprep(const float* __restrict__ W1, const float* __restrict__ W2,
                                               __bf16* __restrict__ planes, int K) {
  const int layer = blockIdx.y >> 1, orient = blockIdx.y & 1;
  const float* W = layer ? W2 : W1;
  __bf16* ph = planes + ((size_t)((layer * 2 + orient) * 2 + 0)) * K * KP;
  __bf16* pl = ph + (size_t)K * KP;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < K * KP; i += gridDim.x * 256) {
    const int r = i / KP, c = i - r * KP;
    float x = 0.f;
    if (c < K) x = orient ? W[(size_t)c * K + r] : W[(size_t)r * K + c];
    const __bf16 h = (__bf16)x;
    ph[i] = h;
    pl[i] = (__bf16)(x - (float)h);
  }
}

// The wave's weight panel: column tiles {w, w+8} x 256 k x hi/lo = 128 VGPRs per lane, requested
// in one burst so a GEMM exposes ONE L2 round trip; issued a whole phase before it is consumed
// (during the observation embedding / the previous layer's epilogue) so that trip is hidden too.
struct Panel {
  bf16x8 h[NJ][KP / 32], l[NJ][KP / 32];
};

__device__ __forceinline__ void load_panel(Panel& p, const __bf16* __restrict__ Bh,
                                           const __bf16* __restrict__ Bl, int nct, int wave, int lane) {
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    const size_t boff = (size_t)(16 * (j < nct ? j : 0) + (lane & 15)) * KP + 8 * (lane >> 4);
#pragma unroll
    for (int kc = 0; kc < KP / 32; ++kc) {
      p.h[jj][kc] = *reinterpret_cast<const bf16x8*>(Bh + boff + kc * 32);
      p.l[jj][kc] = *reinterpret_cast<const bf16x8*>(Bl + boff + kc * 32);
    }
  }
}

// one column tile (jj) of the panel: 16 loads per lane.  The backward kernel requests the first half right
// after its 44 gather loads and the second once those have been consumed: s_waitcnt counts at most 63
// outstanding operations in issue order, so "wait for my activations" is only expressible while fewer
// than 64 younger loads are in flight.
template <int JJ>
__device__ __forceinline__ void load_panel_half(Panel& p, const __bf16* __restrict__ Bh,
                                                const __bf16* __restrict__ Bl, int nct, int wave, int lane) {
  const int j = wave + NWAVE * JJ;
  const size_t boff = (size_t)(16 * (j < nct ? j : 0) + (lane & 15)) * KP + 8 * (lane >> 4);
#pragma unroll
  for (int kc = 0; kc < KP / 32; ++kc) {
    p.h[JJ][kc] = *reinterpret_cast<const bf16x8*>(Bh + boff + kc * 32);
    p.l[JJ][kc] = *reinterpret_cast<const bf16x8*>(Bl + boff + kc * 32);
  }
}

// lds_barrier() (rd_common.h): every barrier of these kernels orders LDS traffic only -- no thread reads global
// memory that another thread of the same launch wrote, and a full __syncthreads() would drain the weight panel.
// Keeps the first USE of a prefetched value below this point (volatile asm statements stay in program
// order, so below the preceding lds_barrier): otherwise the scheduler folds the consumer's arithmetic up to
// the load to save registers and waits for the data before the weight panel has even been requested.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
// Split value -> hi/lo planes with ONE 4-byte LDS store per lane: the MFMA accumulator layout puts columns
// n (even lane) and n+1 (odd lane) of a row in neighbouring lanes; the pair swaps one half through a DPP
// quad permute (no LDS traffic), the even lane stores (hi[n], hi[n+1]) to the hi plane and the odd lane
// (lo[n], lo[n+1]) to the lo plane.  Replaces two 2-byte stores per lane, which the epilogue stamps showed
// to dominate (96 sub-dword stores per lane and layer).  Must be called by all 64 lanes; n = column of this lane.
__device__ __forceinline__ void store_split_pair(__bf16* Ph, __bf16* Pl, int row, int n, int lane, float y) {
  const __bf16 h = (__bf16)y;
  const __bf16 l = (__bf16)(y - (float)h);
  const unsigned hb = __builtin_bit_cast(unsigned short, h), lb = __builtin_bit_cast(unsigned short, l);
  const bool odd = lane & 1;
  const unsigned send = odd ? hb : lb;
  const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);   // lane ^ 1
  const unsigned word = odd ? (recv | (lb << 16)) : (hb | (recv << 16));
  __bf16* P = odd ? Pl : Ph;
  *reinterpret_cast<unsigned*>(P + row * LDX + (n & ~1)) = word;
}

// zero what the products read but no phase writes: pad columns [K, KP) of rows [0, crow) and whole pad rows
// [prow, rows) of one plane, with 16-byte stores (K % 16 == 0; a row is 528 bytes)
__device__ __forceinline__ void zero_plane_pads(__bf16* P, int rows, int prow, int crow, int K, int tid) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  char* base = reinterpret_cast<char*>(P);
  const int nrow16 = (rows - prow) * LDX * (int)sizeof(__bf16) / 16;
  float4* q = reinterpret_cast<float4*>(base + (size_t)prow * LDX * sizeof(__bf16));
  for (int i = tid; i < nrow16; i += NTHR) q[i] = z;
  const int n16 = (KP - K) / 8;                           // 16-byte chunks of pad columns per row
  for (int i = tid; i < crow * n16; i += NTHR) {
    const int r = i / n16, c = i - r * n16;
    *reinterpret_cast<float4*>(base + ((size_t)r * LDX + K) * sizeof(__bf16) + 16 * c) = z;
  }
}


// acc[jj][rt] (16x16 tiles) += A[rows rt*16.., KP] * panel^T ; A planes (hi/lo) in LDS.
// The three split products are issued as three sweeps over independent accumulators.
template <int RT>
__device__ __forceinline__ void mma_panel(f32x4 (&acc)[NJ][RT], const __bf16* Ah, const __bf16* Al,
                                          const Panel& p, int lane) {
  const int aoff = (lane & 15) * LDX + 8 * (lane >> 4);
#pragma unroll
  for (int kc = 0; kc < KP / 32; ++kc) {
    bf16x8 ah[RT], al[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + aoff + kc * 32);
      al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + aoff + kc * 32);
    }
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], p.h[jj][kc], acc[jj][rt], 0, 0, 0);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.l[jj][kc], acc[jj][rt], 0, 0, 0);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.h[jj][kc], acc[jj][rt], 0, 0, 0);
  }
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NJ][RT]) {
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[jj][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// zero `bytes` of LDS (multiple of 16) with 16-byte stores, no index arithmetic
__device__ __forceinline__ void zero_lds(void* p, int bytes, int tid) {
  float4* q = reinterpret_cast<float4*>(p);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < bytes / 16; i += NTHR) q[i] = z;
}


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(NTHR) void k_msg_fwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int ROWS = RT * 16;
  __bf16* Xh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Xl = Xh + ROWS * LDX;
  __bf16* Yh = Xl + ROWS * LDX;
  __bf16* Yl = Yh + ROWS * LDX;
  float* Ys = reinterpret_cast<float*>(smem_raw);        // fp32 [F][LDS_F] staging, aliases X planes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int T = a.T, F = a.F, d = a.d, K = a.K, B = a.B;
  const int nct = K / 16;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  (void)d;

  RD_STAMP(0);
  Panel pw;
  float srow[RT][4];                                     // aggregate coefficient of this lane's rows
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * (lane >> 4) + r;
      srow[rt][r] = row < F ? a.ssum[row] : 0.f;
    }
  float bias1[NJ], bias2[NJ];                            // both layers' biases: ahead of the weight stream
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int n = 16 * min(wave + NWAVE * jj, nct - 1) + (lane & 15);
    bias1[jj] = a.b1[n]; bias2[jj] = a.b2[n];
  }

  // ---- observation embedding -> X planes (+ fp32 copy for the weight-gradient pass) ------------
  // thread -> (f, t) with t fastest: LDS / xsave writes are contiguous; the strided src reads hit each
  // 128-B line F times within the workgroup (L1).  The loads of the first batch (the only one when
  // F*T <= 2048) are requested BEFORE the layer-1 weight panel: loads return in issue order, and the
  // embedding is on the critical path while the panel is not needed before the first MFMA.
  constexpr int UNR = 4;
  const int total = F * T;
  uint64_t seed_eff = a.seed;
  float v[UNR]; int fi[UNR], ti[UNR]; float4 ru[UNR], uu[UNR];
  auto embed_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = min(base + u * NTHR, total - 1);          // clamped duplicates rewrite the same cell
      fi[u] = i / T; ti[u] = i - fi[u] * T;
      v[u] = a.src[((size_t)ti[u] * B + b) * (2 * F) + fi[u]];
      ru[u] = *reinterpret_cast<const float4*>(a.R_u + fi[u] * 4);
    }
  };
  auto embed_consume = [&]() {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int f = fi[u], t = ti[u];
      pin(v[u]); pin(ru[u]);
      float x[4] = {fmaxf(v[u] * ru[u].x, 0.f), fmaxf(v[u] * ru[u].y, 0.f), fmaxf(v[u] * ru[u].z, 0.f), fmaxf(v[u] * ru[u].w, 0.f)};
      if (a.p_drop > 0.f) {                                   // wave-uniform
        x[0] = uu[u].x >= a.p_drop ? x[0] * inv_keep : 0.f; x[1] = uu[u].y >= a.p_drop ? x[1] * inv_keep : 0.f;
        x[2] = uu[u].z >= a.p_drop ? x[2] * inv_keep : 0.f; x[3] = uu[u].w >= a.p_drop ? x[3] * inv_keep : 0.f;
      }
      split_store4(Xh + f * LDX + 4 * t, Xl + f * LDX + 4 * t, x);
      *reinterpret_cast<float4*>(a.xsave + ((size_t)b * F + f) * K + 4 * t) = make_float4(x[0], x[1], x[2], x[3]);
    }
  };
  // dropout masks: one Philox call = the 4 channel masks of a (t, f) cell; ~1 k cycles of integer multiplies per
  // call and wave, evaluated while the loads above are in flight (they do not depend on the loaded data)
  auto embed_masks = [&]() {
    if (a.p_drop > 0.f) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        uu[u] = uniform4(seed_eff, SITE_OBS_EMBED, ((uint64_t)ti[u] * B + b) * F + fi[u]);
        pin(uu[u]);                                           // materialised here, above the barrier
      }
    }
  };
  embed_issue(tid);
  load_panel(pw, plane(a, 0, 0, 0), plane(a, 0, 0, 1), nct, wave, lane);   // 32 loads per lane, behind the 24 above
  // device seed cell (rd_set_seed_cell) on the scalar path: a vector load here would sit behind the panel
  if (a.seed_cell) seed_eff += load_uniform_u64(a.seed_cell);
  RD_STAMP(10);
  embed_masks();
  RD_STAMP(11);
  // pads only: X rows >= F and columns >= K (the embedding writes the rest); Y columns >= K (its pad rows are
  // written as zeros by the layer-1 epilogue)
  zero_plane_pads(Xh, ROWS, F, F, K, tid); zero_plane_pads(Xl, ROWS, F, F, K, tid);
  zero_plane_pads(Yh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(Yl, ROWS, ROWS, ROWS, K, tid);
  RD_STAMP(12);
  lds_barrier();
  RD_STAMP(13);
  embed_consume();
  for (int base = tid + NTHR * UNR; base < total; base += NTHR * UNR) { embed_issue(base); embed_masks(); embed_consume(); }
  RD_STAMP(1);
  lds_barrier();
  RD_STAMP(2);

  // ---- layer 1: Y1 = relu(X W1^T + b1) * ssum ----------------------------------------------------
  f32x4 acc[NJ][RT];
  zero_acc<RT>(acc);
  mma_panel<RT>(acc, Xh, Xl, pw, lane);
  RD_STAMP(3);
  // layer-2 weights start streaming while the epilogue below runs
  load_panel(pw, plane(a, 1, 0, 0), plane(a, 1, 0, 1), nct, wave, lane);
  // branch-free epilogue: pad rows carry srow == 0 and land in the planes' pad rows
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform
      const int n = 16 * j + (lane & 15);
      const float bias = bias1[jj];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;
          const float y = fmaxf(acc[jj][rt][r] + bias, 0.f) * srow[rt][r];
          store_split_pair(Yh, Yl, row, n, lane, y);
        }
    }
  }
  RD_STAMP(4);
  lds_barrier();
  RD_STAMP(5);
  // Y1 for the backward pass, written row-contiguously from the planes (hi + lo is exactly the
  // value the split-bf16 products of the backward pass would reconstruct anyway)
  {
    const int kq = K / 4;
    for (int i = tid; i < F * kq; i += NTHR) {
      const int f = i / kq, k = 4 * (i - f * kq);
      const bf16x4 h = *reinterpret_cast<const bf16x4*>(Yh + f * LDX + k);
      const bf16x4 l = *reinterpret_cast<const bf16x4*>(Yl + f * LDX + k);
      *reinterpret_cast<float4*>(a.y1save + ((size_t)b * F + f) * K + k) =
          make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                      (float)h[3] + (float)l[3]);
    }
  }

  // ---- layer 2: Y2 = relu(Y1 W2^T + b2) * ssum -> fp32 staging -----------------------------------
  zero_acc<RT>(acc);
  mma_panel<RT>(acc, Yh, Yl, pw, lane);
  RD_STAMP(6);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform
      const int n = 16 * j + (lane & 15);
      const float bias = bias2[jj];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;      // rows >= F land in the staging tile's slack
          Ys[row * LDS_F + n] = fmaxf(acc[jj][rt][r] + bias, 0.f) * srow[rt][r];
        }
    }
  }
  RD_STAMP(7);
  lds_barrier();
  RD_STAMP(8);
  // ---- [F, T*d] -> z[t, b, f*d + c]: consecutive threads write consecutive addresses; the thread's
  // (f, c) is fixed and t advances by the number of rows the block covers (no divisions in the loop)
  {
    const int Fd = F * 4;
    const int tpb = NTHR / Fd;                           // time steps covered per pass
    const int t0 = tid / Fd, fc = tid - t0 * Fd;
    if (tpb > 0) {
      if (t0 < tpb) {
        const int f = fc >> 2, c = fc & 3;
        for (int t = t0; t < T; t += tpb)
          a.z[((size_t)t * B + b) * a.ldz + fc] = Ys[f * LDS_F + t * 4 + c];
      }
    } else {
      for (int i = tid; i < T * Fd; i += NTHR) {
        const int t = i / Fd, q = i - t * Fd;
        a.z[((size_t)t * B + b) * a.ldz + q] = Ys[(q >> 2) * LDS_F + t * 4 + (q & 3)];
      }
    }
  }
  // ---- positional encoding + padding mask of this sample (code/models_rd.py:28-38,298-299) ----
  if (a.times != nullptr) {
    const int H = a.d_pe >> 1;
    for (int i = tid; i < T * H; i += NTHR) {
      const int t = i / H, k = i - t * H;
      const float ang = a.times[(size_t)t * B + b] / a.tscale[k];
      float* row = a.z + ((size_t)t * B + b) * a.ldz + F * 4;
      row[k] = sinf(ang);
      row[H + k] = cosf(ang);
    }
    const int64_t len = a.lengths[b];
    for (int t = tid; t < T; t += NTHR) a.mask[(size_t)b * T + t] = (uint8_t)((int64_t)t >= len);
  }
  RD_STAMP(9);
}

// ------------------------------------------------------------------------------------------------
// backward (activation side): dZ2 -> dZ1 -> dX -> per-sample dR_u partial.  The weight gradients
// dW_l = dZ_l^T In_l reduce over all B*F rows and run as split-K GEMMs on the saved dZ tensors.
// ------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(NTHR) void k_msg_bwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int ROWS = RT * 16;
  __bf16* Dh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Dl = Dh + ROWS * LDX;
  __bf16* Eh = Dl + ROWS * LDX;
  __bf16* El = Eh + ROWS * LDX;
  float* St = reinterpret_cast<float*>(Eh);              // fp32 [ROWS][LDS_F] staging, aliases the E planes
  float* Sx = reinterpret_cast<float*>(Dh);              // second staging tile (dX), aliases the D planes
  unsigned char* Mk = smem_raw + (size_t)4 * ROWS * LDX * sizeof(__bf16);   // [ROWS][KP] bytes: Y1 > 0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int T = a.T, F = a.F, K = a.K, B = a.B;
  const int nct = K / 16;
  const int Fd = F * 4;
  const int kq = K / 4;

  RD_STAMP(0);
  Panel pw;
  float srow[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + 4 * (lane >> 4) + r;
      srow[rt][r] = row < F ? a.ssum[row] : 0.f;
    }

  // ---- dZ2 = dz * ssum * (z > 0), read coalesced in [t, f*d+c] order, transposed through LDS; and the
  // ReLU gate of layer 1 as bytes from the saved Y1.  All of a thread's loads (<= 44) are requested in
  // one burst BEFORE the W2^T panel (loads return in issue order), then half the panel; the other half
  // follows once the gather has been consumed.
  const int tpb = NTHR / Fd;                                // >= 2 (F <= 64)
  const int t0 = tid / Fd, fc = tid - t0 * Fd;
  const bool gact = t0 < tpb;
  const int gf = fc >> 2, gc = fc & 3;
  constexpr int GU = 20;
  float zz[GU], dd[GU];
  auto gather_issue = [&](int tb) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int t = min(tb + u * tpb, T - 1);                   // clamped duplicates are not stored
      const size_t zi = ((size_t)t * B + b) * a.ldz + fc;
      zz[u] = a.z[zi]; dd[u] = a.dz[zi];
    }
  };
  auto gather_consume = [&](int tb, float sf) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int t = tb + u * tpb;
      pin(zz[u]); pin(dd[u]);
      if (t < T) St[gf * LDS_F + t * 4 + gc] = (zz[u] > 0.f) ? dd[u] * sf : 0.f;
    }
  };
  constexpr int YU = 4;
  float4 yv[YU];
  const int ncell = F * kq;
  auto gate_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < YU; ++u) {
      const int i = min(base + u * NTHR, ncell - 1);
      const int f = i / kq, k = 4 * (i - f * kq);
      yv[u] = *reinterpret_cast<const float4*>(a.y1save + ((size_t)b * F + f) * K + k);
    }
  };
  auto gate_consume = [&](int base) {
#pragma unroll
    for (int u = 0; u < YU; ++u) {
      const int i = base + u * NTHR;
      pin(yv[u]);
      if (i < ncell) {
        const int f = i / kq, k = 4 * (i - f * kq);
        *reinterpret_cast<uchar4*>(Mk + f * KP + k) = make_uchar4(yv[u].x > 0.f, yv[u].y > 0.f, yv[u].z > 0.f, yv[u].w > 0.f);
      }
    }
  };
  float sf = 0.f;
  if (gact) { sf = a.ssum[gf]; gather_issue(t0); }
  gate_issue(tid);
  load_panel_half<0>(pw, plane(a, 1, 1, 0), plane(a, 1, 1, 1), nct, wave, lane);
  RD_STAMP(10);
  zero_lds(smem_raw, 4 * ROWS * LDX * (int)sizeof(__bf16) + ROWS * KP, tid);
  RD_STAMP(11);
  lds_barrier();
  RD_STAMP(12);
  if (gact) {
    gather_consume(t0, sf);
    for (int tb = t0 + GU * tpb; tb < T; tb += GU * tpb) { gather_issue(tb); gather_consume(tb, sf); }
  }
  RD_STAMP(13);
  gate_consume(tid);
  for (int base = tid + YU * NTHR; base < ncell; base += YU * NTHR) { gate_issue(base); gate_consume(base); }
  load_panel_half<1>(pw, plane(a, 1, 1, 0), plane(a, 1, 1, 1), nct, wave, lane);
  RD_STAMP(1);
  lds_barrier();
  RD_STAMP(2);
  for (int i = tid; i < F * kq; i += NTHR) {
    const int f = i / kq, k = 4 * (i - f * kq);
    const float4 v = *reinterpret_cast<const float4*>(St + f * LDS_F + k);
    const float x[4] = {v.x, v.y, v.z, v.w};
    split_store4(Dh + f * LDX + k, Dl + f * LDX + k, x);
    *reinterpret_cast<float4*>(a.dz2save + ((size_t)b * F + f) * K + k) = v;
  }
  lds_barrier();
  zero_lds(Eh, 2 * ROWS * LDX * (int)sizeof(__bf16), tid);   // staging (aliased) is dead: clear the E planes
  RD_STAMP(3);
  lds_barrier();

  // ---- dZ1 = (dZ2 W2) * ssum * (Y1 > 0) ----------------------------------------------------------
  f32x4 acc[NJ][RT];
  zero_acc<RT>(acc);
  mma_panel<RT>(acc, Dh, Dl, pw, lane);
  RD_STAMP(4);
  load_panel(pw, plane(a, 0, 1, 0), plane(a, 0, 1, 1), nct, wave, lane);   // W1^T streams during the epilogue
  // inputs of the dR_u pass (independent of both products) ride behind the weight stream
  constexpr int XU = 4;
  const int ncells = F * T;
  float4 xs[XU]; float svv[XU];
  auto ru_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = min(base + u * NTHR, ncells - 1);
      const int f = i / T, t = i - f * T;
      xs[u] = *reinterpret_cast<const float4*>(a.xsave + ((size_t)b * F + f) * K + 4 * t);
      svv[u] = a.src[((size_t)t * B + b) * (2 * F) + f];
    }
  };
  ru_issue(tid);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform; body is branch-free
      const int n = 16 * j + (lane & 15);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;      // pad rows: srow == 0 and gate == 0
          const float g = Mk[row * KP + n] ? acc[jj][rt][r] * srow[rt][r] : 0.f;
          store_split_pair(Eh, El, row, n, lane, g);
        }
    }
  }
  RD_STAMP(5);
  lds_barrier();
  // dZ1 for the weight-gradient pass, row-contiguous from the planes
  for (int i = tid; i < F * kq; i += NTHR) {
    const int f = i / kq, k = 4 * (i - f * kq);
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(Eh + f * LDX + k);
    const bf16x4 l = *reinterpret_cast<const bf16x4*>(El + f * LDX + k);
    *reinterpret_cast<float4*>(a.dz1save + ((size_t)b * F + f) * K + k) =
        make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2],
                    (float)h[3] + (float)l[3]);
  }

  // ---- dX = dZ1 W1 -> fp32 staging (the D planes are dead) -------------------------------------
  zero_acc<RT>(acc);
  mma_panel<RT>(acc, Eh, El, pw, lane);
  RD_STAMP(6);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {
      const int n = 16 * j + (lane & 15);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Sx[(rt * 16 + 4 * (lane >> 4) + r) * LDS_F + n] = acc[jj][rt][r];
    }
  }
  RD_STAMP(7);
  lds_barrier();
  // ---- dR_u[f*4+c] = sum_t dX[f, 4t+c] * (X > 0) * src[t,b,f] * keep -----------------------------
  // pass 1 (thread per (f,t) cell, in place): P = dX * gate * src * keep
  const float keep = 1.0f / (1.0f - a.p_drop);
  auto ru_consume = [&](int base) {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int i = base + u * NTHR;
      pin(xs[u]); pin(svv[u]);
      if (i < ncells) {
        const int f = i / T, t = i - f * T;
        const float sv = svv[u] * keep;
        float4 dx = *reinterpret_cast<float4*>(Sx + f * LDS_F + 4 * t);
        dx.x = xs[u].x > 0.f ? dx.x * sv : 0.f; dx.y = xs[u].y > 0.f ? dx.y * sv : 0.f;
        dx.z = xs[u].z > 0.f ? dx.z * sv : 0.f; dx.w = xs[u].w > 0.f ? dx.w * sv : 0.f;
        *reinterpret_cast<float4*>(Sx + f * LDS_F + 4 * t) = dx;
      }
    }
  };
  ru_consume(tid);
  for (int base = tid + XU * NTHR; base < ncells; base += XU * NTHR) { ru_issue(base); ru_consume(base); }
  RD_STAMP(8);
  lds_barrier();
  // pass 2 (thread per (f,c)): fixed-order sum over t
  for (int i = tid; i < Fd; i += NTHR) {
    const int f = i >> 2, c = i & 3;
    float v = 0.f;
    for (int t = 0; t < T; ++t) v += Sx[f * LDS_F + 4 * t + c];
    a.rupart[(size_t)b * Fd + i] = v;
  }
  RD_STAMP(9);
}

template <int RT>
int launch_fused(const FusedArgs& a, bool bwd, hipStream_t st) {
  const size_t lds = (size_t)4 * RT * 16 * LDX * sizeof(__bf16);
  if (!bwd) {
    RD_LDS_ATTR((k_msg_fwd_fused<RT>), lds);
    hipLaunchKernelGGL(k_msg_fwd_fused<RT>, dim3(a.B), dim3(NTHR), lds, st, a);
    return check_launch("k_msg_fwd_fused");
  }
  const size_t ldsb = lds + (size_t)RT * 16 * KP;                  // + ReLU gate bytes
  RD_LDS_ATTR((k_msg_bwd_fused<RT>), ldsb);
  hipLaunchKernelGGL(k_msg_bwd_fused<RT>, dim3(a.B), dim3(NTHR), ldsb, st, a);
  return check_launch("k_msg_bwd_fused");
}

}  // namespace

static unsigned long long* g_stamps = nullptr;
extern "C" void rd_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }   // not part of the ABI

bool fused_msgpass_ok(const rd_shape* s) {
  const int K = s->T * s->d_ob;
  // staging tile [F][244] fp32 must fit inside two bf16 planes [RT*16][264]; d_ob == 4 only
  return precision() == RD_PREC_BF16X3 && s->d_ob == 4 && s->F <= 64 && K <= 240 && (K % 16) == 0 && K >= 16;
}

size_t fused_wplanes_bytes(const rd_shape* s) {
  const size_t K = (size_t)s->T * s->d_ob;
  return align_up(8 * K * KP * sizeof(__bf16), 256);
}

int fused_wprep(const rd_shape* s, const float* W1, const float* W2, void* planes, hipStream_t st) {
  const int K = s->T * s->d_ob;
  hipLaunchKernelGGL(k_wprep, dim3(32, 4), dim3(256), 0, st, W1, W2, (__bf16*)planes, K);
  return check_launch("k_wprep");
}

int fused_msgpass_fwd(const rd_shape* s, const float* src, const float* R_u, const float* b1, const float* b2,
                      const float* ssum, const void* planes, float p_drop, uint64_t seed, float* xsave,
                      float* y1save, float* z, int ldz, hipStream_t st, const float* times,
                      const int64_t* lengths, const float* tscale, uint8_t* mask) {
  FusedArgs a{};
  a.times = times; a.lengths = lengths; a.tscale = tscale; a.mask = mask; a.d_pe = s->d_pe;
  a.src = src; a.R_u = R_u; a.b1 = b1; a.b2 = b2; a.ssum = ssum; a.wplanes = (const __bf16*)planes;
  a.xsave = xsave; a.y1save = y1save; a.z = z; a.ldz = ldz;
  a.B = s->B; a.T = s->T; a.F = s->F; a.d = s->d_ob; a.K = s->T * s->d_ob;
  a.p_drop = p_drop; a.seed = seed; a.seed_cell = seed_cell(); a.stamps = g_stamps;
  switch (cdiv(s->F, 16)) {
    case 1: return launch_fused<1>(a, false, st);
    case 2: return launch_fused<2>(a, false, st);
    case 3: return launch_fused<3>(a, false, st);
    default: return launch_fused<4>(a, false, st);
  }
}

int fused_msgpass_bwd(const rd_shape* s, const float* src, const float* ssum, const void* planes, float p_drop,
                      const float* xsave, const float* y1save, const float* z, const float* dz, int ldz,
                      float* dz2save, float* dz1save, float* rupart, hipStream_t st) {
  FusedArgs a{};
  a.src = src; a.ssum = ssum; a.wplanes = (const __bf16*)planes;
  a.xsave = const_cast<float*>(xsave); a.y1save = const_cast<float*>(y1save); a.z = const_cast<float*>(z);
  a.dz = dz; a.ldz = ldz; a.dz2save = dz2save; a.dz1save = dz1save; a.rupart = rupart;
  a.B = s->B; a.T = s->T; a.F = s->F; a.d = s->d_ob; a.K = s->T * s->d_ob;
  a.p_drop = p_drop; a.stamps = g_stamps;
  switch (cdiv(s->F, 16)) {
    case 1: return launch_fused<1>(a, true, st);
    case 2: return launch_fused<2>(a, true, st);
    case 3: return launch_fused<3>(a, true, st);
    default: return launch_fused<4>(a, true, st);
  }
}

}  // namespace rd
