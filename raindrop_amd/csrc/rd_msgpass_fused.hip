// rd_msgpass_fused.hip -- kernel K1, fused LDS-resident form for small sensor graphs
// (F <= 48 sensors, K = T*d_ob <= 240 and a multiple of 16: the P19 shape, K = 240).
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
// step by k_wprep into native MFMA operand tiles (rd_k1_layout.h: every wave-load is one contiguous
// kilobyte -- 4x the L2->CU rate of a row-major plane) that stay L2-resident; activations are split
// when they are written to LDS.
//
// What the backward pass needs is handed over in the form it is consumed in (rd_k1_layout.h):
//   * X and Y1 (the layer inputs of dW_l = dZ_l^T In_l) as split-bf16 ROW tiles, transposed on the way
//     out, so the weight-gradient kernel (rd_msgpass_dw.hip) streams pure MFMA operands;
//   * the three ReLU gates (X > 0, Y1 > 0, Y2 > 0) as bit masks: 5 KB per sample instead of re-reading
//     96 KB of activations.
// The backward kernel here does the activation-side chain dz -> dZ2 -> dZ1 -> dX -> dR_u and emits dZ2, dZ1
// as row tiles; no fp32 copy of X, Y1, dZ1 or dZ2 exists in HBM.
//
// Layout in LDS (RT = ceil(F/16) row tiles): four bf16 planes [RT*16][264] (X hi/lo, Y1 hi/lo;
// 528-B rows keep ds_read_b128 conflict-free) plus fp32 staging tiles [F][244] aliased on top of the plane
// pair that is dead at that point (coalesced [F,K] <-> [T,F*d] transposes and the row-tile transposes).
#include <stdlib.h>

#include "rd_common.h"
#include "rd_k1_layout.h"
#include "rd_plan.h"
#include "rd_rng.h"

// RD_ABL: bit mask of phases compiled OUT (timing ablations, tools/k1_ablate.sh; results are garbage):
//   1 weight panel loads   2 MFMAs   4 row-tile exports   8 z scatter + PE / dR_u tail   16 embedding / dz gather arithmetic
#ifndef RD_ABL
#define RD_ABL 0
#endif

namespace rd {
namespace {

using k1::KP;
using k1::NKC;
using k1::TILE;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int LDX = KP + 8;      // bf16 elements per LDS plane row (528 B)
constexpr int LDS_F = 244;       // fp32 staging row stride (conflict-free for both access orders)
// MEASURED in-step, library variants A/B'd in one call: 1024 threads is neutral on the pool's fast boxes (0.714 vs 0.714 ms/step)
// and 3 % faster on its slow ones (0.911 -> 0.884), where these instruction-issue-bound kernels stretch the most.
#ifndef RD_K1_NTHR
#define RD_K1_NTHR 1024
#endif
constexpr int NTHR = RD_K1_NTHR; // 512: 8 wavefronts, wave w owns output column tiles {w, w+8}; 1024: 16 wavefronts, one tile each
constexpr int NWAVE = NTHR / 64, NJ = 16 / NWAVE;
constexpr int CPT = 2048 / NTHR; // cells (t, f) per thread and batch of loads (F*T <= 2048 needs one batch)

struct FusedArgs {
  const float *src, *R_u, *b1, *b2, *ssum;
  const float *times, *tscale; const int64_t* lengths; uint8_t* mask; int d_pe;   // optional PE / mask (fwd)
  const __bf16* wt;              // weight tiles [layer 2][orient 2][nct][NKC][hi/lo][64][8]
  __bf16* ones;                  // bwd writes the constant bias-gradient operand tile of rd_msgpass_dw.hip here
  __bf16 *tpX, *tpY1, *tpD1, *tpD2;   // row tiles of X, Y1 (fwd writes) and dZ1, dZ2 (bwd writes)
  uint16_t *m1, *m2; uint8_t* mx;     // gates: Y1 > 0, Y2 > 0 (bit per element), X > 0 (byte per (f,t))
  float* z;
  const float* dz;               // bwd
  float* rupart;
  int B, T, F, K, ldz, nct, q, rem, per;
  float p_drop; uint64_t seed; const uint64_t* seed_cell;
  unsigned long long* stamps;    // debug: per-phase clock64() of every wave of the first 4 workgroups
  const int32_t* plan;           // token plan (rd_plan.h) or null: which steps of a sample are live, and where its z rows are
  int* lin;                      // [B] per slot: 1 + last step with a non-zero observation (fwd writes, bwd reads)
};

// Which sample a workgroup owns and where its rows live.  Padded layout (no plan): workgroup i = sample i, every step is
// live, z row of step t = t*B + b.  With a plan: workgroup i owns the sample of RANK i (longest first); `b` indexes the
// caller's tensors (src, times, lengths, mask), `sb` = the rank indexes everything these kernels hand to each other (row
// tiles, gate bits, per-sample partials); steps t >= L are padding and neither written nor read; z row of step t = row0 + t.
struct Tok { int b, sb, L, row0, rstride; };
__device__ __forceinline__ Tok tok_of(const FusedArgs& a) {
  Tok k;
  if (a.plan) {
    const int r = blockIdx.x;
    k.sb = r;
    k.b = __builtin_amdgcn_readfirstlane(a.plan[plan::order_base(a.B) + r]);
    k.L = __builtin_amdgcn_readfirstlane(a.plan[plan::len_base(a.B) + r]);
    k.row0 = __builtin_amdgcn_readfirstlane(a.plan[plan::off_base() + r]);
    k.rstride = 1;
  } else {
    k.b = k.sb = blockIdx.x; k.L = a.T; k.row0 = blockIdx.x; k.rstride = a.B;
  }
  return k;
}

// Shape numbers of a launch.  The kernels are instantiated with the sensor and step counts as COMPILE-TIME constants for the shapes
// that matter (P19: F = 34, T = 60) and with FC = TC = 0 (read from the arguments) for every other shape of the envelope: these
// kernels are instruction-issue bound, and a quarter of the generic instantiation's instructions was index arithmetic on runtime
// shape numbers (integer divisions, 32-bit multiplies at quarter rate) that folds to shifts and adds here.
// A specialised instantiation also fixes ldz = 4 F + 16 and d_pe = 16 (the model's layout, code/models_rd.py:216,354); the
// host picks it only then, and only while every element index of the launch fits 31 bits (the index arithmetic is 32-bit there).
struct Dim { int B, T, F, K, nct, q, rem, per, ldz, H; };
template <int FC, int TC>
__device__ __forceinline__ Dim dims_of(const FusedArgs& a) {
  Dim d;
  d.B = a.B; d.T = TC ? TC : a.T; d.F = FC ? FC : a.F;
  d.K = 4 * d.T; d.nct = d.K / 16; d.q = d.F / 32; d.rem = d.F % 32; d.per = d.rem ? 32 / d.rem : 1;
  d.ldz = FC ? 4 * FC + 16 : a.ldz; d.H = FC ? 8 : (a.d_pe >> 1);
  return d;
}
// 24-bit multiply (full rate; v_mul_lo_u32 issues at quarter rate): both factors < 2^24, product < 2^32
__device__ __forceinline__ unsigned m24(unsigned x, unsigned y) { return __umul24(x, y); }

#define RD_STAMP(i)                                                                              \
  do {                                                                                           \
    if (a.stamps && blockIdx.x < 4 && (threadIdx.x & 63) == 0)                                   \
      a.stamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = clock64();                    \
  } while (0)

__device__ __forceinline__ const __bf16* wtiles(const FusedArgs& a, const Dim& d, int layer, int orient) {
  return a.wt + (size_t)((layer * 2 + orient) * d.nct) * (NKC * 2 * TILE);
}

// bulk-output stores (row tiles, z).  A write-through (sc1, inline asm) variant was measured: no gain at the kernel
// boundary and it broke parity, so these are plain stores.
__device__ __forceinline__ void st16(void* p, const bf16x8& v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ void st16f(void* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st8(void* p, const bf16x4& v) { *reinterpret_cast<bf16x4*>(p) = v; }
__device__ __forceinline__ void st2(__bf16* p, __bf16 v) { *p = v; }

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { h[e] = (__bf16)v[e]; l[e] = (__bf16)(v[e] - (float)h[e]); }
}

__device__ __forceinline__ void split_store4(__bf16* ph, __bf16* pl, const float (&v)[4]) {
  bf16x4 h, l;
#pragma unroll
  for (int j = 0; j < 4; ++j) { h[j] = (__bf16)v[j]; l[j] = (__bf16)(v[j] - (float)h[j]); }
  *reinterpret_cast<bf16x4*>(ph) = h;
  *reinterpret_cast<bf16x4*>(pl) = l;
}

// W [K,K] fp32 -> native operand tiles, both orientations (rd_k1_layout.h).  One workgroup per (layer, orient,
// column tile j); wave w converts reduction steps kc = w, w+4.  Every store is a contiguous kilobyte per wave.
__global__ __launch_bounds__(256) void k_wprep(const float* __restrict__ W1, const float* __restrict__ W2,
                                               __bf16* __restrict__ wt, int K, int nct) {
  const int j = blockIdx.x % nct, lo_ = blockIdx.x / nct, orient = lo_ & 1, layer = lo_ >> 1;
  const float* W = layer ? W2 : W1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, G = lane >> 4;
  __bf16* base = wt + (size_t)blockIdx.x * (NKC * 2 * TILE) + lane * 8;
  const int fr = 16 * j + c;                         // free index: n (orient 0) or k (orient 1)
  for (int kc = wave; kc < NKC; kc += 4) {
    const int r0 = 32 * kc + 8 * G;                  // first reduction index of this lane (multiple of 8; K % 8 == 0)
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 < K) {
      if (orient == 0) {
        const float4 u = *reinterpret_cast<const float4*>(W + (size_t)fr * K + r0);
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)fr * K + r0 + 4);
        x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = W[(size_t)(r0 + e) * K + fr];
      }
    }
    bf16x8 h, l;
    split8(x, h, l);
    *reinterpret_cast<bf16x8*>(base + (kc * 2 + 0) * TILE) = h;
    *reinterpret_cast<bf16x8*>(base + (kc * 2 + 1) * TILE) = l;
  }
}

// The wave's weight panel: column tiles {w, w+8} x 256 k x hi/lo = 128 VGPRs per lane, requested in one
// burst of 32 contiguous 1-KB wave-loads and issued a whole phase before it is consumed.
struct Panel {
  bf16x8 h[NJ][NKC], l[NJ][NKC];
};

// reduction steps [KC0, KC1) of both column tiles
template <int KC0, int KC1>
__device__ __forceinline__ void load_panel_kc(Panel& p, const __bf16* __restrict__ wl, int nct, int wave, int lane) {
  if (RD_ABL & 1) return;
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = __builtin_amdgcn_readfirstlane(wave) + NWAVE * jj;
    if (j >= nct) continue;                            // a wave without this column tile requests nothing (wave-uniform)
    const __bf16* t = wl + (size_t)j * (NKC * 2 * TILE) + lane * 8;
#pragma unroll
    for (int kc = KC0; kc < KC1; ++kc) {
      p.h[jj][kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 0) * TILE);
      p.l[jj][kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 1) * TILE);
    }
  }
}
__device__ __forceinline__ void load_panel(Panel& p, const __bf16* __restrict__ wl, int nct, int wave, int lane) {
  load_panel_kc<0, NKC>(p, wl, nct, wave, lane);
}

// lds_barrier() (rd_common.h): every barrier of these kernels orders LDS traffic only -- no thread reads global
// memory that another thread of the same launch wrote, and a full __syncthreads() would drain the weight panel.
// pin(): keeps the first USE of a prefetched value below this point (volatile asm statements stay in program
// order, so below the preceding lds_barrier): otherwise the scheduler folds the consumer's arithmetic up to
// the load to save registers and waits for the data before the weight panel has even been requested.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin(float4& x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
__device__ __forceinline__ void pin(unsigned& x) { asm volatile("" : "+v"(x)); }
// Split value -> hi/lo planes with ONE 4-byte LDS store per lane: the MFMA accumulator layout puts columns
// n (even lane) and n+1 (odd lane) of a row in neighbouring lanes; the pair swaps one half through a DPP
// quad permute (no LDS traffic), the even lane stores (hi[n], hi[n+1]) to the hi plane and the odd lane
// (lo[n], lo[n+1]) to the lo plane.  Must be called by all 64 lanes; n = column of this lane.
__device__ __forceinline__ void store_split_pair(__bf16* Ph, __bf16* Pl, int row, int n, int lane, __bf16 h, __bf16 l) {
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

// acc[jj][rt] (16x16 tiles) += A[rows rt*16.., 0 .. 32 NKC) * panel^T ; A planes (hi/lo) in LDS.
// The A fragments are ROLLED: as soon as the three split products of (step kc, row tile rt) are issued, the fragments of
// (kc + 1, rt) are requested into the same registers, so the LDS reads of the next reduction step travel under the MFMAs of the
// other row tiles instead of in front of the step (measured round 3: the products ran at ~50 % of the matrix pipe with
// read-then-multiply steps -- all waves of a SIMD read, then all multiply).  RD_K1_ROLL=0 builds the old order (A/B).
// `mid()` runs after step NKC/2 - 1: the registers of the first half of the panel are free there, so the NEXT layer's first
// half-panel streams in underneath the second half of this product.
// kclim (wave-uniform): reduction steps kc >= kclim are skipped -- the caller knows the A operand is exactly zero there
// (padded / unobserved time steps), so the skipped products are x0: bit-safe.
#ifndef RD_K1_ROLL
#define RD_K1_ROLL 1
#endif
template <int RT, typename Mid>
__device__ __forceinline__ void mma_mid(f32x4 (&acc)[NJ][RT], const __bf16* Ah, const __bf16* Al, Panel& p, int lane, int kclim, Mid&& mid) {
  const int aoff = (lane & 15) * LDX + 8 * (lane >> 4);
  if (RD_ABL & 2) { mid(); return; }
  bf16x8 ah[RT], al[RT];
  if (RD_K1_ROLL) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + aoff);
      al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + aoff);
    }
  }
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    if (kc < kclim) {                                              // wave-uniform
      if (!RD_K1_ROLL) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + aoff + kc * 32);
          al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + aoff + kc * 32);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], p.h[jj][kc], acc[jj][rt], 0, 0, 0);
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.l[jj][kc], acc[jj][rt], 0, 0, 0);
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.h[jj][kc], acc[jj][rt], 0, 0, 0);
        }
        if (RD_K1_ROLL && kc + 1 < NKC) {                          // in-bounds whatever kclim is: the planes hold NKC steps
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + aoff + (kc + 1) * 32);
          al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + aoff + (kc + 1) * 32);
        }
      }
      if (RD_K1_ROLL && kc + 1 < NKC) {
        // pin the issue order of this step: three MFMAs, the two fragment reads they free, ... (the scheduler otherwise
        // collects all six reads behind the ninth MFMA and waits for them at once)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * NJ, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
    }
    if (kc == NKC / 2 - 1) {
      __builtin_amdgcn_sched_barrier(0);                            // the scheduler otherwise sinks these loads below the second half
      mid();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
template <int RT>
__device__ __forceinline__ void mma_panel(f32x4 (&acc)[NJ][RT], const __bf16* Ah, const __bf16* Al,
                                          Panel& p, int lane, int kclim = NKC) {
  mma_mid<RT>(acc, Ah, Al, p, lane, kclim, [] {});
}

// srow[rt][r] = ssum[16 rt + 4 g + r] (0 beyond F): one 16-byte load per row tile when F % 4 == 0 puts the quad inside
// the array, scalar loads otherwise (every VMEM instruction costs the address unit 16 cycles whatever its width)
template <int RT>
__device__ __forceinline__ void load_srow(float (&srow)[RT][4], const float* __restrict__ ssum, int F, int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row0 = rt * 16 + 4 * (lane >> 4);
    if ((F & 3) == 0 && (reinterpret_cast<uintptr_t>(ssum) & 15) == 0) {
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 < F) q = *reinterpret_cast<const float4*>(ssum + row0);
      srow[rt][0] = q.x; srow[rt][1] = q.y; srow[rt][2] = q.z; srow[rt][3] = q.w;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) srow[rt][r] = row0 + r < F ? ssum[row0 + r] : 0.f;
    }
  }
}

template <int RT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NJ][RT]) {
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[jj][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// cell index i -> (t, f) = (i / F, i % F) for i < 2^15, F <= 64: exact with one fp32 reciprocal multiply + fix-up
// (an integer division is ~40 instructions; the embedding does four per thread)
__device__ __forceinline__ void cell_tf(int i, int F, int& t, int& f) {
  t = (int)((float)i * (1.0f / (float)F));
  f = i - t * F;
  if (f < 0) { f += F; --t; }
  if (f >= F) { f -= F; ++t; }
}

// zero `bytes` of LDS (multiple of 16) with 16-byte stores, no index arithmetic
__device__ __forceinline__ void zero_lds(void* p, int bytes, int tid) {
  float4* q = reinterpret_cast<float4*>(p);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < bytes / 16; i += NTHR) q[i] = z;
}

// ------------------------------------------------------------------------------------------------
// row tiles for the weight-gradient kernel (rd_k1_layout.h)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ __bf16* tp_tile(__bf16* tp, int nct, int s, int j) {
  return tp + ((size_t)s * nct + j) * (2 * TILE);
}

// Staging-sourced: S = fp32 [F][LDS_F] tile of sample b in LDS -> row tiles.  A slot = 8 consecutive graph rows of
// one column: read down the column (conflict-free across the 16 columns of a lane group), split, one 16-byte
// store per part; consecutive lanes write consecutive 16-byte slots of a tile.
// jlim / jlimL: only column tiles j < jlim of the main tiles and j < jlimL of the leftover rows are written (the consumer
// reads no further: rd_msgpass_dw.hip skips the column blocks that are all padding for a sample).
__device__ __forceinline__ void tstore_stage(const float* S, __bf16* tp, const Dim& a, int b, int tid, int jlim, int jlimL) {
  if (RD_ABL & 4) return;
  const int nct = a.nct;
  const int nslot = a.q * jlim * 64;
  for (int idx = tid; idx < nslot; idx += NTHR) {
    const int L = idx & 63, t2 = idx >> 6;
    const int m = a.q == 1 ? 0 : t2 / jlim, j = t2 - m * jlim;
    const float* p = S + (32 * m + 8 * (L >> 4)) * LDS_F + 16 * j + (L & 15);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = p[e * LDS_F];
    bf16x8 h, l;
    split8(v, h, l);
    __bf16* dst = tp_tile(tp, nct, b * a.q + m, j) + L * 8;
    st16(dst, h);
    st16(dst + TILE, l);
  }
  if (a.rem) {
    const int s = a.B * a.q + b / a.per, slot = b % a.per;
    const int KL = 16 * jlimL;
    // thread -> (leftover row li = tid / 256 + 4 pass, column n = tid % 256): K <= 256, no division
    for (int li = tid >> 8; li < a.rem; li += NTHR >> 8) {
      const int n = tid & 255;
      if (n < KL) {
        const float x = S[(32 * a.q + li) * LDS_F + n];
        const int r = slot * a.rem + li;
        const __bf16 h = (__bf16)x, l = (__bf16)(x - (float)h);
        __bf16* dst = tp_tile(tp, nct, s, n >> 4) + ((n & 15) + 16 * (r >> 3)) * 8 + (r & 7);
        st2(dst, h); st2(dst + TILE, l);
      }
    }
  }
}

// positions of a leftover tile that no sample covers: zeros, written by the workgroup whose sample sits at position 0
__device__ __forceinline__ void tzero_uncovered(__bf16* tp, const Dim& a, int b, int tid) {
  if (a.rem == 0 || (b % a.per) != 0) return;
  const int first = b / a.per * a.per;
  const int nvalid = min(a.per, a.B - first);
  const int r0 = nvalid * a.rem;
  const int s = a.B * a.q + b / a.per;
  const __bf16 zero = (__bf16)0.f;
  for (int rr = tid >> 8; rr < 32 - r0; rr += NTHR >> 8) {
    const int n = tid & 255, r = r0 + rr;
    if (n < a.K) {
      __bf16* dst = tp_tile(tp, a.nct, s, n >> 4) + ((n & 15) + 16 * (r >> 3)) * 8 + (r & 7);
      st2(dst, zero); st2(dst + TILE, zero);
    }
  }
}

// Accumulator-sourced, main tiles: the MFMA C layout gives each lane 4 consecutive rows of one column = half a slot.
// h/l[i] = split value at (row 16 rt + 4 g + i, column 16 j + c).  Row tiles rt >= 2 q hold leftover rows: those are
// stored from the LDS planes after the epilogue's barrier (tstore_leftover_planes), two store instructions per wave
// instead of sixteen mostly-masked ones.
__device__ __forceinline__ void tstore_acc(__bf16* tp, const Dim& a, int b, int j, int rt, int lane,
                                           const __bf16 (&h)[4], const __bf16 (&l)[4]) {
  if (RD_ABL & 4) return;
  const int c = lane & 15, g = lane >> 4;
  if (rt < 2 * a.q) {                                    // uniform: this row tile is half of a main tile
    bf16x4 hv, lv;
#pragma unroll
    for (int i = 0; i < 4; ++i) { hv[i] = h[i]; lv[i] = l[i]; }
    __bf16* dst = tp_tile(tp, a.nct, b * a.q + (rt >> 1), j) + (c + 16 * (2 * (rt & 1) + (g >> 1))) * 8 + 4 * (g & 1);
    st8(dst, hv);
    st8(dst + TILE, lv);
  }
}
// leftover rows (32 q + li, li < rem) of a tensor whose split planes [row][LDX] are complete in LDS
__device__ __forceinline__ void tstore_leftover_planes(const __bf16* Ph, const __bf16* Pl, __bf16* tp, const Dim& a,
                                                       int b, int tid) {
  if (a.rem == 0 || (RD_ABL & 4)) return;
  const int s = a.B * a.q + b / a.per, slot = b % a.per;
  for (int li = tid >> 8; li < a.rem; li += NTHR >> 8) {
    const int n = tid & 255, r = slot * a.rem + li;
    if (n < a.K) {
      __bf16* dst = tp_tile(tp, a.nct, s, n >> 4) + ((n & 15) + 16 * (r >> 3)) * 8 + (r & 7);
      st2(dst, Ph[(32 * a.q + li) * LDX + n]);
      st2(dst + TILE, Pl[(32 * a.q + li) * LDX + n]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Schedule.  The 8 waves form two groups (waves 0-3 / 4-7; wave w and w+4 share a SIMD).  Wherever a phase consists
// of a matrix-core part and a VALU / LDS / memory part, the two groups run the parts in OPPOSITE order, so each
// SIMD's matrix pipe works for one wave while its partner wave splits, stores or issues loads (measured: a
// GEMM of the two waves of a SIMD is pipe-bound, 2 x 144 MFMA x 16 cycles; a wave blocks while it issues a
// 32-load weight panel, ~2 k cycles for half the workgroup).
template <int RT, int FC, int TC>
__global__ __launch_bounds__(NTHR) void k_msg_fwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RD_TOUCH_CODE(14336);                                  // own code -> L2 (rd_common.h; the smallest instantiation is 15.5 KB)
  constexpr int ROWS = RT * 16;
  constexpr bool ALIAS = RT > 3;                         // F > 48: the fp32 copy of X has to share the Y planes' space
  constexpr size_t PLANES = (size_t)4 * ROWS * LDX * sizeof(__bf16);
  constexpr size_t XS_BYTES = ALIAS ? 0 : (size_t)ROWS * LDS_F * sizeof(float);
  __bf16* Xh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Xl = Xh + ROWS * LDX;
  __bf16* Yh = Xl + ROWS * LDX;
  __bf16* Yl = Yh + ROWS * LDX;
  float* Ys = reinterpret_cast<float*>(smem_raw);        // fp32 [F][LDS_F] staging of Y2, aliases the X planes
  float* Xs = ALIAS ? reinterpret_cast<float*>(Yh) : reinterpret_cast<float*>(smem_raw + PLANES);   // fp32 copy of X
  uint16_t* M1 = reinterpret_cast<uint16_t*>(smem_raw + PLANES + XS_BYTES);                          // [ROWS][16]
  uint16_t* M2 = M1 + ROWS * 16;
  int* LinW = reinterpret_cast<int*>(M2 + ROWS * 16);     // [NWAVE] per-wave "1 + last observed step" (the slot the backward uses for ssum)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool grpB = __builtin_amdgcn_readfirstlane(wave) >= NWAVE / 2;   // scalar: the group branches are real branches
  const Tok tk = tok_of(a);
  const int b = tk.b, sb = tk.sb, L = tk.L;
  const Dim dm = dims_of<FC, TC>(a);
  const int T = dm.T, F = dm.F, K = dm.K, B = dm.B;
  const int nct = dm.nct;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  // column tile j of a layer's output = time steps 4j .. 4j+3.  Layer 2's output at padded steps (t >= L) is never read
  // (rd_plan.h): a wave whose column tiles are all padding skips its layer-2 weight stream, product and epilogue.
  const bool live2 = 4 * __builtin_amdgcn_readfirstlane(wave) < L;

  RD_STAMP(0);
  Panel pw;
  float srow[RT][4];                                     // aggregate coefficient of this lane's rows
  load_srow<RT>(srow, a.ssum, F, lane);
  float bias1[NJ], bias2[NJ];                            // both layers' biases: ahead of the weight stream
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int n = 16 * min(wave + NWAVE * jj, nct - 1) + (lane & 15);
    bias1[jj] = a.b1[n]; bias2[jj] = a.b2[n];
  }

  // ---- observation embedding -> X planes (+ fp32 copy for the row-tile transpose, + gate byte) ----
  // thread -> cell (t, f) with f fastest: a wave-load of src covers two or three 136-byte row segments (with t
  // fastest it touched 64 cache lines, and the 64 such loads of the workgroup cost more tag look-ups than the
  // whole weight panel).  LDS: the 16-byte fp32 stores are conflict-free at stride 244, the 8-byte plane
  // stores 2-way.
  constexpr int UNR = CPT;
  const int total = F * T;
  uint64_t seed_eff = a.seed;
  float v[UNR]; int fi[UNR], ti[UNR]; float4 ru[UNR]; unsigned km[UNR];     // km: keep bits of the 4 channels
  int lin_w = 0;                                          // wave-uniform: 1 + last step at which this wave saw a non-zero observation
  auto embed_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = min(base + u * NTHR, total - 1);          // clamped duplicates rewrite the same cell
      cell_tf(i, F, ti[u], fi[u]);
      v[u] = a.src[(size_t)(m24(m24(ti[u], B) + b, 2 * F) + fi[u])];
      ru[u] = *reinterpret_cast<const float4*>(a.R_u + fi[u] * 4);
    }
  };
  auto embed_consume = [&]() {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int f = fi[u], t = ti[u];
      pin(v[u]); pin(ru[u]);
      {   // cells are ordered by time step (f fastest): the highest lane holding a non-zero value has the wave's latest step
        const unsigned long long nz = __ballot(v[u] != 0.f);
        if (nz) lin_w = max(lin_w, 1 + __builtin_amdgcn_readlane(t, 63 - __builtin_clzll(nz)));
      }
      float x[4] = {fmaxf(v[u] * ru[u].x, 0.f), fmaxf(v[u] * ru[u].y, 0.f), fmaxf(v[u] * ru[u].z, 0.f), fmaxf(v[u] * ru[u].w, 0.f)};
      if (a.p_drop > 0.f) {                                   // wave-uniform
        x[0] = (km[u] & 1) ? x[0] * inv_keep : 0.f; x[1] = (km[u] & 2) ? x[1] * inv_keep : 0.f;
        x[2] = (km[u] & 4) ? x[2] * inv_keep : 0.f; x[3] = (km[u] & 8) ? x[3] * inv_keep : 0.f;
      }
      split_store4(Xh + m24(f, LDX) + 4 * t, Xl + m24(f, LDX) + 4 * t, x);
      *reinterpret_cast<float4*>(Xs + m24(f, LDS_F) + 4 * t) = make_float4(x[0], x[1], x[2], x[3]);
      a.mx[(size_t)(m24(sb, total) + m24(t, F) + f)] =        // [slot][t][f]: consecutive lanes, consecutive bytes
          (uint8_t)((x[0] > 0.f ? 1 : 0) | (x[1] > 0.f ? 2 : 0) | (x[2] > 0.f ? 4 : 0) | (x[3] > 0.f ? 8 : 0));
    }
  };
  // dropout masks: one Philox call = the 4 channel masks of a (t, f) cell; ~0.5 k cycles of integer multiplies per
  // call and wave, no memory traffic
  auto embed_masks = [&]() {
    if (a.p_drop > 0.f) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const float4 q4 = uniform4(seed_eff, SITE_OBS_EMBED, (uint64_t)(m24(m24(ti[u], B) + b, F) + fi[u]));
        km[u] = (q4.x >= a.p_drop ? 1u : 0u) | (q4.y >= a.p_drop ? 2u : 0u) | (q4.z >= a.p_drop ? 4u : 0u) | (q4.w >= a.p_drop ? 8u : 0u);
        pin(km[u]);
      }
    }
  };
  if (!(RD_ABL & 16)) embed_issue(tid);
  // device seed cell (rd_set_seed_cell) on the scalar path: a vector load would queue behind the panel
  if (a.seed_cell) seed_eff += load_uniform_u64(a.seed_cell);
  // pads only: X rows >= F and columns >= K (the embedding writes the rest)
  zero_plane_pads(Xh, ROWS, F, F, K, tid); zero_plane_pads(Xl, ROWS, F, F, K, tid);
  if (!ALIAS) { zero_plane_pads(Yh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(Yl, ROWS, ROWS, ROWS, K, tid); }
  // group B does its arithmetic BEFORE requesting its weight panel, group A after: one-sided uniform branches around
  // a single panel load (an if/else with the panel in both arms makes the allocator spill the panel)
  if (grpB && !(RD_ABL & 16)) { embed_masks(); embed_consume(); }
  RD_STAMP(10);
  load_panel(pw, wtiles(a, dm, 0, 0), nct, wave, lane);
  RD_STAMP(11);
  if (!grpB && !(RD_ABL & 16)) { embed_masks(); embed_consume(); }
  if (!(RD_ABL & 16))
    for (int base = tid + NTHR * UNR; base < total; base += NTHR * UNR) { embed_issue(base); embed_masks(); embed_consume(); }
  if (lane == 0) LinW[wave] = lin_w;
  RD_STAMP(1);
  lds_barrier();
  RD_STAMP(2);
  // X[:, 4t..4t+3] is exactly zero for every step t >= lin (no sensor observed: relu(0 * R_u) = 0, dropout or not), so
  // layer 1's reduction stops after the last 32-column chunk that holds an observed step
  int lin = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) lin = max(lin, LinW[w]);
  lin = __builtin_amdgcn_readfirstlane(lin);
  const int kclim1 = (4 * lin + 31) >> 5;
  if (tid == 0) {
    a.lin[sb] = lin;
    if (a.plan && lin > L) atomicMax(const_cast<int*>(a.plan) + plan::I_SLACK, lin - L);
  }

  // ---- layer 1: Y1 = relu(X W1^T + b1) * ssum;  X leaves as row tiles for dW1 (reads the fp32 copy) --------
  f32x4 acc[NJ][RT];
  zero_acc<RT>(acc);
  if (grpB) {                                                // group B transposes while group A multiplies ...
    tstore_stage(Xs, a.tpX, dm, sb, tid, nct, nct);
    tzero_uncovered(a.tpX, dm, sb, tid);
    tzero_uncovered(a.tpY1, dm, sb, tid);
  }
  RD_STAMP(13);
  mma_mid<RT>(acc, Xh, Xl, pw, lane, kclim1, [&] {
    if (live2) load_panel_kc<0, NKC / 2>(pw, wtiles(a, dm, 1, 0), nct, wave, lane);   // layer-2 weights, first half of the reduction
  });
  RD_STAMP(3);
  if (live2) load_panel_kc<NKC / 2, NKC>(pw, wtiles(a, dm, 1, 0), nct, wave, lane);   // second half
  RD_STAMP(12);
  if (!grpB) {                                               // ... and the other way round
    tstore_stage(Xs, a.tpX, dm, sb, tid, nct, nct);
    tzero_uncovered(a.tpX, dm, sb, tid);
    tzero_uncovered(a.tpY1, dm, sb, tid);
  }
  if (ALIAS) {                                               // the fp32 copy of X lives in the Y planes: everybody must be done with it
    lds_barrier();
    zero_plane_pads(Yh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(Yl, ROWS, ROWS, ROWS, K, tid);
  }
  RD_STAMP(14);
  // branch-free epilogue: pad rows carry srow == 0 and land in the planes' pad rows
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform
      const int n = 16 * j + (lane & 15);
      const float bias = bias1[jj];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        __bf16 hh[4], ll[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;
          const float y = fmaxf(acc[jj][rt][r] + bias, 0.f) * srow[rt][r];
          hh[r] = (__bf16)y; ll[r] = (__bf16)(y - (float)hh[r]);
          store_split_pair(Yh, Yl, row, n, lane, hh[r], ll[r]);
          const unsigned long long bal = __ballot(y > 0.f);
          if ((lane & 15) == 0) M1[row * 16 + j] = (uint16_t)(bal >> (16 * (lane >> 4)));
        }
        tstore_acc(a.tpY1, dm, sb, j, rt, lane, hh, ll);
      }
    }
  }
  RD_STAMP(4);
  lds_barrier();
  RD_STAMP(5);
  // gate bits of layer 1 -> global (rows < F: 32 bytes each); leftover rows of Y1 -> row tiles
  for (int i = tid; i < 2 * F; i += NTHR)
    reinterpret_cast<uint4*>(a.m1 + (size_t)m24(sb, F * 16))[i] = reinterpret_cast<const uint4*>(M1)[i];
  tstore_leftover_planes(Yh, Yl, a.tpY1, dm, sb, tid);

  // ---- layer 2: Y2 = relu(Y1 W2^T + b2) * ssum -> fp32 staging (live column tiles only) ------------
  zero_acc<RT>(acc);
  if (live2) mma_panel<RT>(acc, Yh, Yl, pw, lane);
  RD_STAMP(6);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct && 4 * j < L) {                                 // wave-uniform
      const int n = 16 * j + (lane & 15);
      const float bias = bias2[jj];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;      // rows >= F land in the staging tile's slack
          const float y = fmaxf(acc[jj][rt][r] + bias, 0.f) * srow[rt][r];
          Ys[row * LDS_F + n] = y;
          const unsigned long long bal = __ballot(y > 0.f);
          if ((lane & 15) == 0) M2[row * 16 + j] = (uint16_t)(bal >> (16 * (lane >> 4)));
        }
    }
  }
  RD_STAMP(7);
  lds_barrier();
  RD_STAMP(8);
  for (int i = tid; i < 2 * F; i += NTHR)
    reinterpret_cast<uint4*>(a.m2 + (size_t)m24(sb, F * 16))[i] = reinterpret_cast<const uint4*>(M2)[i];
  // ---- [F, T*d] -> z[row(t), f*d + c]: thread -> (t, f) with f fastest moves the 4 channels of a cell as one 16-byte
  // LDS read (conflict-free at stride 244) and one 16-byte store; consecutive lanes write consecutive addresses.
  // Live steps only (t < L; L == T on the padded layout).
  const int ldz = dm.ldz;
  if (RD_ABL & 8) return;
  if ((ldz & 3) == 0) {
    for (int i = tid; i < L * F; i += NTHR) {
      int t, f;
      cell_tf(i, F, t, f);
      st16f(a.z + (size_t)(m24(tk.row0 + m24(t, tk.rstride), ldz) + 4 * f), *reinterpret_cast<const float4*>(Ys + m24(f, LDS_F) + 4 * t));
    }
  } else {
    const int Fd = F * 4;
    for (int i = tid; i < L * Fd; i += NTHR) {
      const int t = i / Fd, q = i - t * Fd;
      a.z[(size_t)(m24(tk.row0 + m24(t, tk.rstride), ldz) + q)] = Ys[(q >> 2) * LDS_F + t * 4 + (q & 3)];
    }
  }
  // ---- positional encoding + padding mask of this sample (code/models_rd.py:28-38,298-299) ----
  if (a.times != nullptr) {
    const int H = dm.H;
    for (int i = tid; i < L * H; i += NTHR) {
      const int t = i / H, k = i - t * H;
      const float ang = a.times[(size_t)(m24(t, B) + b)] / a.tscale[k];
      float* row = a.z + (size_t)(m24(tk.row0 + m24(t, tk.rstride), ldz) + F * 4);
      float sn, cs;
      sincosf(ang, &sn, &cs);                                  // one argument reduction for the pair
      row[k] = sn;
      row[H + k] = cs;
    }
    const int64_t len = a.lengths[b];
    for (int t = tid; t < T; t += NTHR) a.mask[(size_t)(m24(b, T) + t)] = (uint8_t)((int64_t)t >= len);
  }
  RD_STAMP(9);
}

// ------------------------------------------------------------------------------------------------
// backward (activation side): dZ2 -> dZ1 -> dX -> per-sample dR_u partial; dZ2 and dZ1 leave as row tiles.
// The weight gradients dW_l = dZ_l^T In_l reduce over all B*F rows: rd_msgpass_dw.hip.
// Same two-group schedule as the forward kernel.
// ------------------------------------------------------------------------------------------------
template <int RT, int FC, int TC>
__global__ __launch_bounds__(NTHR) void k_msg_bwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RD_TOUCH_CODE(11264);                                  // own code -> L2 (the smallest instantiation is 12.5 KB)
  constexpr int ROWS = RT * 16;
  constexpr bool ALIAS = RT > 3;
  constexpr size_t PLANES = (size_t)4 * ROWS * LDX * sizeof(__bf16);
  constexpr size_t ST_BYTES = ALIAS ? 0 : (size_t)ROWS * LDS_F * sizeof(float);
  __bf16* Dh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Dl = Dh + ROWS * LDX;
  __bf16* Eh = Dl + ROWS * LDX;
  __bf16* El = Eh + ROWS * LDX;
  float* St = ALIAS ? reinterpret_cast<float*>(Eh) : reinterpret_cast<float*>(smem_raw + PLANES);   // fp32 [F][LDS_F]: dZ2
  float* Sx = reinterpret_cast<float*>(Dh);              // second staging tile (dX), aliases the D planes
  uint16_t* M1 = reinterpret_cast<uint16_t*>(smem_raw + PLANES + ST_BYTES);   // [ROWS][16]: Y1 > 0
  uint16_t* M2 = M1 + ROWS * 16;                                               // [ROWS][16]: Y2 > 0
  float* Ss = reinterpret_cast<float*>(M2 + ROWS * 16);                        // [ROWS]: ssum
  float* Rp = reinterpret_cast<float*>(Eh);              // dR_u partial sums [groups][F*4], aliases the E planes (dead by then)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool grpB = __builtin_amdgcn_readfirstlane(wave) >= NWAVE / 2;   // scalar: the group branches are real branches
  const Tok tk = tok_of(a);
  const int b = tk.b, sb = tk.sb, L = tk.L;
  const Dim dm = dims_of<FC, TC>(a);
  const int T = dm.T, F = dm.F, K = dm.K, B = dm.B;
  const int nct = dm.nct;
  const int Fd = F * 4;
  const int kq = K / 4;
  // dz is exactly zero at the padded steps t >= L (rd_plan.h) -- on the padded layout L == T.  So dZ2's columns >= 4L are zero:
  // the product dZ2 W2 stops after the last 32-column chunk with a live step, and rd_msgpass_dw.hip reads dZ2's column blocks
  // (64 columns = 16 steps) only for samples that are live there -- only those blocks are exported.
  // dX feeds dR_u alone, through the gate X > 0, which is closed at every step >= lin (no observation): the wave whose
  // column tile is past lin skips the dZ1 W1 product.
  const int kclim2 = (4 * L + 31) >> 5;
  const int lin = __builtin_amdgcn_readfirstlane(a.lin[sb]);
  const bool liveX = 4 * __builtin_amdgcn_readfirstlane(wave) < lin;
  int jlim = nct, jlimL = nct;
  if (a.plan) {
    jlim = min(nct, 4 * ((L + 15) >> 4));
    const int Lg = __builtin_amdgcn_readfirstlane(a.plan[plan::len_base(B) + (sb / dm.per) * dm.per]);   // longest sample of this leftover tile
    jlimL = min(nct, 4 * ((Lg + 15) >> 4));
  }

  RD_STAMP(0);
  if (blockIdx.x == 0 && tid < 192) {                                 // constant operand tiles [ones][zeros][zeros]: column 0 of the 16 is one
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (tid < 64 && (tid & 15) == 0) ? (__bf16)1.f : (__bf16)0.f;
    *reinterpret_cast<bf16x8*>(a.ones + tid * 8) = o;
  }
  Panel pw;
  float srow[RT][4];
  load_srow<RT>(srow, a.ssum, F, lane);

  // ---- dZ2 = dz * ssum * (Y2 > 0), dz read coalesced in [t, f*d+c] order and transposed through LDS; the gate
  // bits of both layers come from the forward pass (2 x 32 bytes per graph row).
  // thread -> cell (t, f), f fastest: one 16-byte load per cell (the 4 channels), one 16-byte LDS store
  constexpr int GU = CPT;
  const int total = F * T;
  const int ldz = dm.ldz;
  const bool vec4 = (ldz & 3) == 0;
  float4 dd[GU]; int gt[GU], gfi[GU];
  auto gather_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = min(base + u * NTHR, total - 1);            // clamped duplicates rewrite the same cell
      cell_tf(i, F, gt[u], gfi[u]);
      const int tl = min(gt[u], max(L - 1, 0));                 // padded steps: a legal address, zeroed at the consumer
      const float* p = a.dz + (size_t)(m24(tk.row0 + m24(tl, tk.rstride), ldz) + 4 * gfi[u]);
      if (vec4) dd[u] = *reinterpret_cast<const float4*>(p);
      else dd[u] = make_float4(p[0], p[1], p[2], p[3]);
    }
  };
  auto gather_consume = [&]() {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      pin(dd[u]);
      const int f = gfi[u], k = 4 * gt[u];
      const float sf = Ss[f];
      unsigned bits = (unsigned)M2[f * 16 + (k >> 4)] >> (k & 15);           // 4 consecutive gate bits (k % 4 == 0)
      if (gt[u] >= L) bits = 0;
      *reinterpret_cast<float4*>(St + m24(f, LDS_F) + k) =
          make_float4((bits & 1) ? dd[u].x * sf : 0.f, (bits & 2) ? dd[u].y * sf : 0.f, (bits & 4) ? dd[u].z * sf : 0.f,
                      (bits & 8) ? dd[u].w * sf : 0.f);
    }
  };
  uint4 mw = make_uint4(0, 0, 0, 0);
  if (tid < 4 * F)                                           // threads [0,2F): M1 rows, [2F,4F): M2 rows
    mw = tid < 2 * F ? reinterpret_cast<const uint4*>(a.m1 + (size_t)m24(sb, F * 16))[tid]
                     : reinterpret_cast<const uint4*>(a.m2 + (size_t)m24(sb, F * 16))[tid - 2 * F];
  float sfv = 0.f;
  if (tid >= NTHR - 64 && tid - (NTHR - 64) < F) sfv = a.ssum[tid - (NTHR - 64)];      // last wave: ssum -> LDS
  if (!(RD_ABL & 16)) gather_issue(tid);
  load_panel(pw, wtiles(a, dm, 1, 1), nct, wave, lane);          // W2^T panel queues behind the gather
  RD_STAMP(10);
  // D planes: zero the pads (rows >= F, columns >= K); the staging tile is fully written for rows < F, columns < K
  zero_plane_pads(Dh, ROWS, F, F, K, tid); zero_plane_pads(Dl, ROWS, F, F, K, tid);
  if (!ALIAS) { zero_plane_pads(Eh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(El, ROWS, ROWS, ROWS, K, tid); }
  if (tid < 4 * F) {
    pin(mw.x); pin(mw.y); pin(mw.z); pin(mw.w);
    if (tid < 2 * F) reinterpret_cast<uint4*>(M1)[tid] = mw;
    else reinterpret_cast<uint4*>(M2)[tid - 2 * F] = mw;
  }
  if (tid >= NTHR - 64 && tid - (NTHR - 64) < F) { pin(sfv); Ss[tid - (NTHR - 64)] = sfv; }
  RD_STAMP(11);
  lds_barrier();
  RD_STAMP(12);
  if (!(RD_ABL & 16)) {
    gather_consume();
    for (int base = tid + GU * NTHR; base < total; base += GU * NTHR) { gather_issue(base); gather_consume(); }
  }
  RD_STAMP(13);
  lds_barrier();
  RD_STAMP(1);
  // staging -> D planes (row-major, the A operand of the next product)
  for (int i = tid; i < F * kq; i += NTHR) {
    int f, kk;
    cell_tf(i, kq, f, kk);                                     // i < 2^15, kq <= 64
    const int k = 4 * kk;
    const float4 v = *reinterpret_cast<const float4*>(St + m24(f, LDS_F) + k);
    const float x[4] = {v.x, v.y, v.z, v.w};
    split_store4(Dh + m24(f, LDX) + k, Dl + m24(f, LDX) + k, x);
  }
  RD_STAMP(2);
  lds_barrier();
  RD_STAMP(3);

  // ---- dZ1 = (dZ2 W2) * ssum * (Y1 > 0); dZ2 leaves as row tiles for dW2 (reads the staging tile) ---------
  // inputs of the dR_u pass: thread -> (time group tg, sensor f), f fastest (coalesced); its cells are t = tg, tg+TG, ...
  const int TG = NTHR / F;                                   // >= 8
  const int rtg = tid / F, rf = tid - rtg * F;
  const bool ract = rtg < TG;
  constexpr int XU = CPT;
  unsigned xb[XU]; float svv[XU];
  auto ru_issue = [&](int tb) {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int t = min(tb + u * TG, T - 1);
      xb[u] = a.mx[(size_t)(m24(m24(sb, T) + t, F) + rf)];
      svv[u] = a.src[(size_t)(m24(m24(t, B) + b, 2 * F) + rf)];
    }
  };
  f32x4 acc[NJ][RT];
  zero_acc<RT>(acc);
  if (grpB) {
    tstore_stage(St, a.tpD2, dm, sb, tid, jlim, jlimL);
    tzero_uncovered(a.tpD2, dm, sb, tid);
    tzero_uncovered(a.tpD1, dm, sb, tid);
  }
  mma_mid<RT>(acc, Dh, Dl, pw, lane, kclim2, [&] {
    if (liveX) load_panel_kc<0, NKC / 2>(pw, wtiles(a, dm, 0, 1), nct, wave, lane);   // W1^T, first half of the reduction
  });
  RD_STAMP(4);
  if (liveX) load_panel_kc<NKC / 2, NKC>(pw, wtiles(a, dm, 0, 1), nct, wave, lane);
  if (ract) ru_issue(rtg);
  RD_STAMP(14);
  if (!grpB) {
    tstore_stage(St, a.tpD2, dm, sb, tid, jlim, jlimL);
    tzero_uncovered(a.tpD2, dm, sb, tid);
    tzero_uncovered(a.tpD1, dm, sb, tid);
  }
  if (ALIAS) lds_barrier();                                    // staging tile lives in the E planes: everybody must be done with it
  if (ALIAS) { zero_plane_pads(Eh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(El, ROWS, ROWS, ROWS, K, tid); }
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform; body is branch-free
      const int n = 16 * j + (lane & 15);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        __bf16 hh[4], ll[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rt * 16 + 4 * (lane >> 4) + r;      // pad rows: srow == 0
          const bool open = (M1[row * 16 + j] >> (lane & 15)) & 1;
          const float g = open ? acc[jj][rt][r] * srow[rt][r] : 0.f;
          hh[r] = (__bf16)g; ll[r] = (__bf16)(g - (float)hh[r]);
          store_split_pair(Eh, El, row, n, lane, hh[r], ll[r]);
        }
        tstore_acc(a.tpD1, dm, sb, j, rt, lane, hh, ll);
      }
    }
  }
  RD_STAMP(5);
  lds_barrier();
  RD_STAMP(15);
  tstore_leftover_planes(Eh, El, a.tpD1, dm, sb, tid);

  // ---- dX = dZ1 W1 -> fp32 staging (the D planes are dead); observed column tiles only -----------
  zero_acc<RT>(acc);
  if (liveX) mma_panel<RT>(acc, Eh, El, pw, lane);
  RD_STAMP(6);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct && 4 * j < lin) {
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
  // pass 1: thread (tg, f) sums its time steps (fixed order) into Rp[tg][f*4 + c]
  const float keep = 1.0f / (1.0f - a.p_drop);
  if (RD_ABL & 8) return;
  if (ract) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tb = rtg; tb < T; tb += XU * TG) {
      if (tb != rtg) ru_issue(tb);
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        const int t = tb + u * TG;
        pin(xb[u]); pin(svv[u]);
        if (t < lin) {                                          // lin <= T; beyond it the gate is closed and the staging tile unwritten
          const float sv = svv[u] * keep;
          const float4 dx = *reinterpret_cast<const float4*>(Sx + m24(rf, LDS_F) + 4 * t);
          s.x += (xb[u] & 1) ? dx.x * sv : 0.f; s.y += (xb[u] & 2) ? dx.y * sv : 0.f;
          s.z += (xb[u] & 4) ? dx.z * sv : 0.f; s.w += (xb[u] & 8) ? dx.w * sv : 0.f;
        }
      }
    }
    *reinterpret_cast<float4*>(Rp + (size_t)rtg * Fd + 4 * rf) = s;
  }
  RD_STAMP(8);
  lds_barrier();
  // pass 2 (thread per (f,c)): fixed-order sum over the time groups
  for (int i = tid; i < Fd; i += NTHR) {
    float v = 0.f;
    for (int g = 0; g < TG; ++g) v += Rp[g * Fd + i];
    a.rupart[(size_t)(m24(sb, Fd) + i)] = v;
  }
  RD_STAMP(9);
  // ---- warm the weight-gradient stream's COLD operands.  rd_msgpass_dw.hip runs right behind this kernel and streams four tile
  // tensors: dZ1 / dZ2 are written here (Infinity-Cache-hot), X / Y1 were written by the forward, a whole encoder forward +
  // backward ago (~700 MB of traffic: long evicted) -- in the step k_dw took 21 us against 12 in the isolated loop (slow box).
  // The LAST instructions of the last wave touch one dword per 128-byte line of this sample's X and Y1 tiles: fire-and-forget
  // loads (inline asm: the destination registers are never read and nothing follows that could reuse them), so no wave waits and no
  // register is held through the kernel (eight live registers across it spilled).  The lines are in the memory-side cache when
  // k_dw asks; workgroups finish at different times, so most touches are well ahead of it.
  if (a.tpX && __builtin_amdgcn_readfirstlane(wave) == NWAVE - 1) {
    const int bytes = dm.q * nct * 2 * TILE * (int)sizeof(__bf16);      // main tiles of this sample: contiguous
    const char* bx = reinterpret_cast<const char*>(tp_tile(const_cast<__bf16*>(a.tpX), nct, sb * dm.q, 0));
    const char* by = reinterpret_cast<const char*>(tp_tile(const_cast<__bf16*>(a.tpY1), nct, sb * dm.q, 0));
    // The compiler does not know these loads are asynchronous: every destination must be a register NOTHING else writes before the
    // wave ends (the first version let it recycle one as the next address temporary -- the returning load then corrupted the
    // address: a memory fault at F = 48).  Eight distinct outputs, kept alive to the end by the empty asm below.
    float t[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = min((lane + 64 * u) * 128, max(bytes - 128, 0));
      asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(t[2 * u]), "=&v"(t[2 * u + 1]) : "v"(bx + o), "v"(by + o) : "memory");
    }
    asm volatile("" ::"v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]), "v"(t[5]), "v"(t[6]), "v"(t[7]));
  }
}

template <int RT, int FC, int TC>
int launch_fused(const FusedArgs& a, bool bwd, hipStream_t st) {
  const size_t lds = (size_t)4 * RT * 16 * LDX * sizeof(__bf16) + (RT > 3 ? 0 : (size_t)RT * 16 * LDS_F * sizeof(float)) +
                     (size_t)2 * RT * 16 * 16 * sizeof(uint16_t) + (size_t)RT * 16 * sizeof(float);
  if (!bwd) {
    RD_LDS_ATTR((k_msg_fwd_fused<RT, FC, TC>), lds);
    hipLaunchKernelGGL((k_msg_fwd_fused<RT, FC, TC>), dim3(a.B), dim3(NTHR), lds, st, a);
    return check_launch("k_msg_fwd_fused");
  }
  RD_LDS_ATTR((k_msg_bwd_fused<RT, FC, TC>), lds);
  hipLaunchKernelGGL((k_msg_bwd_fused<RT, FC, TC>), dim3(a.B), dim3(NTHR), lds, st, a);
  return check_launch("k_msg_bwd_fused");
}

// RD_K1_SPECIALIZE=0 runs the P19 shape on the generic instantiation (A/B, and the parity test of the two)
int launch_fused_shape(const FusedArgs& a, const k1::Layout& L, bool bwd, hipStream_t st) {
  if (a.ldz >= 1024) return fail(RD_EUNSUPPORTED, "fused message passing: ldz (%d) must be < 1024", a.ldz);
  const char* e = getenv("RD_K1_SPECIALIZE");
  const bool model_layout = a.ldz == 4 * L.F + 16 && (bwd || a.times == nullptr || a.d_pe == 16);
  if (L.F == 34 && L.T == 60 && model_layout && !(e && atoi(e) == 0)) return launch_fused<3, 34, 60>(a, bwd, st);
  switch (L.RT) {
    case 1: return launch_fused<1, 0, 0>(a, bwd, st);
    case 2: return launch_fused<2, 0, 0>(a, bwd, st);
    case 3: return launch_fused<3, 0, 0>(a, bwd, st);
    // (RT = 4, 48 < F <= 64: its runtime-shape instantiation spilled 68-88 bytes per lane and no dataset has such a sensor count --
    // round 4 narrowed the envelope to F <= 48, fused_msgpass_ok; those shapes take the panel-product path)
    default: return fail(RD_EUNSUPPORTED, "fused message passing: F = %d > 48", L.F);
  }
}

void fill_layout(FusedArgs& a, const k1::Layout& L) {
  a.B = L.B; a.T = L.T; a.F = L.F; a.K = L.K; a.nct = L.nct; a.q = L.q; a.rem = L.rem; a.per = L.per;
}

}  // namespace

static unsigned long long* g_stamps = nullptr;
extern "C" void rd_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; }   // not part of the ABI

bool fused_msgpass_ok(const rd_shape* s) {
  // RD_K1_FUSED=0 routes the fused envelope through the generic tiled path (read per call: the parity tests compare
  // the two paths in one process)
  const char* e = getenv("RD_K1_FUSED");
  const bool enabled = !(e && atoi(e) == 0);
  const int K = s->T * s->d_ob;
  // staging tile [F][244] fp32 must fit inside two bf16 planes [RT*16][264]; d_ob == 4 only
  // index arithmetic of the kernels is 32-bit with 24-bit multiplies: B*T rows < 2^22 (with ldz < 1024, checked at the call)
  return enabled && precision() == RD_PREC_BF16X3 && s->d_ob == 4 && s->F <= 48 && K <= 240 && (K % 16) == 0 && K >= 16 &&
         (long)s->B * s->T < (1L << 22);
}

int fused_wprep(const k1::Layout& L, const float* W1, const float* W2, void* wt, hipStream_t st) {
  hipLaunchKernelGGL(k_wprep, dim3(4 * L.nct), dim3(256), 0, st, W1, W2, (__bf16*)wt, L.K, L.nct);
  return check_launch("k_wprep");
}

int fused_msgpass_fwd(const k1::Layout& L, const float* src, const float* R_u, const float* b1, const float* b2,
                      const float* ssum, const void* wt, float p_drop, uint64_t seed, void* tpX, void* tpY1,
                      void* m1, void* m2, void* mx, float* z, int ldz, hipStream_t st, const float* times,
                      const int64_t* lengths, const float* tscale, uint8_t* mask, int d_pe) {
  FusedArgs a{};
  fill_layout(a, L);
  a.times = times; a.lengths = lengths; a.tscale = tscale; a.mask = mask; a.d_pe = d_pe;
  a.src = src; a.R_u = R_u; a.b1 = b1; a.b2 = b2; a.ssum = ssum; a.wt = (const __bf16*)wt;
  a.tpX = (__bf16*)tpX; a.tpY1 = (__bf16*)tpY1; a.m1 = (uint16_t*)m1; a.m2 = (uint16_t*)m2; a.mx = (uint8_t*)mx;
  a.z = z; a.ldz = ldz;
  a.p_drop = p_drop; a.seed = seed; a.seed_cell = seed_cell(); a.stamps = g_stamps;
  a.plan = token_plan(); a.lin = reinterpret_cast<int*>(reinterpret_cast<char*>(mx) + k1::lin_offset(L.B, L.T, L.F));
  return launch_fused_shape(a, L, false, st);
}

int fused_msgpass_bwd(const k1::Layout& L, const float* src, const float* ssum, const void* wt, float p_drop,
                      const void* m1, const void* m2, const void* mx, const float* dz, int ldz, void* tpD1, void* tpD2,
                      void* ones, float* rupart, hipStream_t st, const void* tpX, const void* tpY1) {
  FusedArgs a{};
  fill_layout(a, L);
  {   // the forward's row tiles, only touched here (RD_K1_WARM=0: not at all; A/B)
    static const bool warm = [] { const char* e = getenv("RD_K1_WARM"); return !(e && atoi(e) == 0); }();
    a.tpX = warm ? (__bf16*)const_cast<void*>(tpX) : nullptr; a.tpY1 = warm ? (__bf16*)const_cast<void*>(tpY1) : nullptr;
  }
  a.src = src; a.ssum = ssum; a.wt = (const __bf16*)wt;
  a.m1 = (uint16_t*)const_cast<void*>(m1); a.m2 = (uint16_t*)const_cast<void*>(m2); a.mx = (uint8_t*)const_cast<void*>(mx);
  a.dz = dz; a.ldz = ldz; a.tpD1 = (__bf16*)tpD1; a.tpD2 = (__bf16*)tpD2; a.ones = (__bf16*)ones; a.rupart = rupart;
  a.p_drop = p_drop; a.stamps = g_stamps;
  a.plan = token_plan();
  a.lin = reinterpret_cast<int*>(reinterpret_cast<char*>(const_cast<void*>(mx)) + k1::lin_offset(L.B, L.T, L.F));
  return launch_fused_shape(a, L, true, st);
}

}  // namespace rd
