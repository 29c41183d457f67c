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
// Layout in LDS (RT = ceil(F/16) row tiles): four bf16 planes [RT*16][256] (X hi/lo, Y1 hi/lo), rows of exactly 512 bytes whose
// 16-byte chunks are XOR-swizzled by the row (pofs): the MFMA A-fragment read -- lane (r, G) takes 16 bytes of row r at chunk
// 4 kc + G -- is serviced by gfx950 in 16-lane groups {r 0-3 G, r 12-15 G, r 4-11 G+1} over 64 banks, and NO padded row stride keeps
// those 16 accesses on distinct bank quads (rounds 1-4 used 528-byte rows: tools/probe_ldsfrag.hip measures 127 B/clk/CU for that,
// 224 for the swizzle, 32 for plain 512-byte rows -- the two K x K products were LDS-bound, 6.2 k cycles of fragment reads against
// 4.9 k cycles of MFMAs).  One fp32 staging tile [F][244] aliases the plane pair that is dead at that point (the [F,K] <-> [T,F*d]
// transposes); the row tiles leave straight from the planes through transposing LDS reads (ds_read_b64_tr_b16).
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

// bytes of its own code each kernel requests into L2 at its start (rd_common.h touch_own_code; checked against the built library:
// raindrop_amd/build.py CODE_TOUCH)
// RD_K1_BLATE: group B requests the first weight panel of a kernel BEHIND the first barrier instead of under the cold loads the
// kernel starts with.  Measured (stamps in the step, round 5): the vector memory pipe is in order -- panel requests issued behind the
// cold observation / dz loads (~6 k cycles) are not accepted until those return, so a wave that requests its panel before it
// consumes them reaches the barrier ~1.8 k cycles later; group A's half of the panel fits under the latency, group B's does not.
#ifndef RD_K1_BLATE
#define RD_K1_BLATE 1
#endif
// RD_K1_ALATE (experiment): group A behind the barrier too -- nothing but the kernel's small first loads in the address unit's queue
#ifndef RD_K1_ALATE
#define RD_K1_ALATE 0
#endif
// the P19 instantiation <3, 34, 60>: ALL of its 13 172 / 10 512 bytes (an uncovered tail is fetched cold, line by line, on the pool's slow boxes)

namespace rd {
namespace {

using k1::KP;
using k1::NKC;
using k1::TILE;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int LDX = KP;          // bf16 elements per LDS plane row (512 B); chunks of 8 elements swizzled by the row: pofs()
constexpr int LDS_F = 244;       // fp32 staging row stride (conflict-free for both access orders)
// element offset of (row, col) in a split plane: 16-byte chunk (col >> 3) lands at chunk ^ (row & 15) within its half row
__device__ __forceinline__ int pofs(int row, int col) {
  return row * LDX + (col & 128) + ((((col >> 3) ^ row) & 15) << 3) + (col & 7);
}
// MEASURED in-step, library variants A/B'd in one call: 1024 threads is neutral on the pool's fast boxes (0.714 vs 0.714 ms/step)
// and 3 % faster on its slow ones (0.911 -> 0.884), where these instruction-issue-bound kernels stretch the most.
#ifndef RD_K1_NTHR
#define RD_K1_NTHR 1024
#endif
constexpr int NTHR = RD_K1_NTHR; // 512: 8 wavefronts, wave w owns output column tiles {w, w+8}; 1024: 16 wavefronts, one tile each
constexpr int NWAVE = NTHR / 64, NJ = 16 / NWAVE;
constexpr int CPT = 2048 / NTHR; // cells (t, f) per thread and batch of loads (F*T <= 2048 needs one batch)
constexpr int GTHR = NTHR / 2, GWAVE = NWAVE / 2;   // threads / waves of one of the two groups (A: waves [0, GWAVE), B: the rest)

struct FusedArgs {
  const float *src, *R_u, *b1, *b2, *ssum;
  const float *times, *tscale; const int64_t* lengths; uint8_t* mask; int d_pe;   // optional PE / mask (fwd)
  const __bf16* wt;              // weight tiles [layer 2][orient 2][nct][NKC][hi/lo][64][8]
  __bf16* ones;                  // bwd writes the constant bias-gradient operand tile of rd_msgpass_dw.hip here
  __bf16 *tpX, *tpY1, *tpD1, *tpD2;   // row tiles of X, Y1 (fwd writes) and dZ1, dZ2 (bwd writes)
  // gates (rd_k1_layout.h): m1 = Y1 > 0 as 64-bit lane masks [slot][column tile][RT][4] of the epilogue that made them (the backward's
  // epilogue has the same lane -> element map and applies them as v_cndmask operands); m2 = Y2 > 0 and mx = X > 0 as one byte per
  // (slot, t, f) cell, bit c = channel c
  uint64_t* m1; uint8_t *m2, *mx;
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

// Debug stamps (tools/k1_stamps.py; a.stamps is null in production).  Per kernel K1_STAMP_WORDS 64-bit words:
//   [4 workgroups][16 waves][16 phases] clock64() of lane 0 -- the first four workgroups (with a token plan: the four LONGEST samples);
//   then [<= 1024 workgroups][4]: start clock64 / start wall_clock64 (thread 0), end clock64 / end wall_clock64 (max over the waves).
// The backward kernel's block sits K1_STAMP_WORDS behind the forward's.
constexpr int K1_STAMP_WORDS = 4 * 16 * 16 + 4 * 1024;
#define RD_STAMP(i)                                                                              \
  do {                                                                                           \
    if (a.stamps && blockIdx.x < 4 && (threadIdx.x & 63) == 0)                                   \
      a.stamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (i)] = clock64();                   \
  } while (0)
#define RD_STAMP_WG_START()                                                                      \
  do {                                                                                           \
    if (a.stamps && threadIdx.x == 0 && blockIdx.x < 1024) {                                     \
      a.stamps[1024 + 4 * blockIdx.x] = clock64(); a.stamps[1024 + 4 * blockIdx.x + 1] = wall_clock64(); }   \
  } while (0)
#define RD_STAMP_WG_END()                                                                        \
  do {                                                                                           \
    if (a.stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 1024) {                              \
      atomicMax(a.stamps + 1024 + 4 * blockIdx.x + 2, (unsigned long long)clock64());            \
      atomicMax(a.stamps + 1024 + 4 * blockIdx.x + 3, (unsigned long long)wall_clock64()); }     \
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
// four consecutive columns (n0 % 4 == 0) of one row -> hi / lo planes: two packed conversions and one 8-byte LDS store per plane
__device__ __forceinline__ void store_split_quad(__bf16* Ph, __bf16* Pl, int row, int n0, const float (&y)[4]) {
  const int o = pofs(row, n0);
  split_store4(Ph + o, Pl + o, y);
}
// 64-bit lane mask (a __ballot) -> lanes 2 k, 2 k + 1 of a collector register; lanes [0, 2 n) of it are stored by ONE instruction later
// (v_writelane_b32 by inline asm: this compiler has no builtin for it; K is a compile-time constant after unrolling.  The mask comes
// straight out of a v_cmp, and a VALU write of an SGPR needs 4 wait states before v_writelane reads it -- the compiler's hazard
// recognizer does not look inside an asm statement, and without the s_nop the LOW half arrived stale on the device: half of the
// columns of dZ1 gated by garbage, nondeterministically.)
template <int K>
__device__ __forceinline__ void collect_mask(int& gv, unsigned long long m) {
  asm("s_nop 3\n\tv_writelane_b32 %0, %1, %3\n\tv_writelane_b32 %0, %2, %4"
      : "+v"(gv) : "s"((unsigned)(m & 0xffffffffull)), "s"((unsigned)(m >> 32)), "n"(2 * K), "n"(2 * K + 1));
}
// x where this lane's bit of the uniform mask is set, 0 elsewhere: one v_cndmask with the mask as its SGPR-pair operand
__device__ __forceinline__ float gate_by_mask(float x, unsigned long long m) {
  float r;
  asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(x), "s"(m));
  return r;
}
// the RT * 4 masks of one (sample slot, column tile): lane l < RT * 8 requests dword l (one wave-wide load, issued with the kernel's first
// loads; the compiler's own vmcnt bookkeeping covers it), unpack_masks moves them to SGPR pairs with v_readlane where the epilogue
// needs them.  (A scalar s_load with its wait in the same asm statement stalled every wave for 3-7 k cycles at the kernel's start:
// the masks are cold -- the forward wrote them a whole encoder pass ago.)
template <int RT>
__device__ __forceinline__ int request_masks(const uint64_t* p, int lane) {
  int v = 0;
  if (lane < RT * 8) v = reinterpret_cast<const int*>(p)[lane];
  return v;
}
template <int RT>
__device__ __forceinline__ void unpack_masks(int v, unsigned long long (&m)[RT * 4]) {
#pragma unroll
  for (int k = 0; k < RT * 4; ++k)
    m[k] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(v, 2 * k) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(v, 2 * k + 1) << 32);
}

// zero what the products read but no phase writes: pad columns [K, KP) of rows [0, crow) and whole pad rows
// [prow, rows) of one plane, with 16-byte stores (K % 16 == 0; a row is 512 bytes, a whole row of zeros is its own swizzle)
__device__ __forceinline__ void zero_plane_pads(__bf16* P, int rows, int prow, int crow, int K, int tid) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nrow16 = (rows - prow) * LDX * (int)sizeof(__bf16) / 16;
  float4* q = reinterpret_cast<float4*>(P + (size_t)prow * LDX);
  for (int i = tid; i < nrow16; i += NTHR) q[i] = z;
  const int n16 = (KP - K) / 8;                           // 16-byte chunks of pad columns per row
  for (int i = tid; i < crow * n16; i += NTHR) {
    const int r = i / n16, c = i - r * n16;
    *reinterpret_cast<float4*>(P + pofs(r, K + 8 * c)) = z;
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
  // fragment of (step kc, row tile rt): row 16 rt + r, chunk 4 kc + G -> swizzled chunk ((kc & 3) << 2 | G) ^ r in half (kc >> 2) of
  // the row (pofs): four lane offsets, one per kc & 3; row tile, half row and the hi / lo plane are immediates
  int aoffk[4];
  {
    const int r = lane & 15, g0 = ((lane >> 4) ^ r) & 15;
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) aoffk[k3] = r * LDX + ((g0 << 3) ^ (k3 << 5));
  }
#define RD_AOFF(kc) (aoffk[(kc) & 3] + ((kc) >> 2) * 128)
  if (RD_ABL & 2) { mid(); return; }
  bf16x8 ah[RT], al[RT];
  if (RD_K1_ROLL) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + RD_AOFF(0));
      al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + RD_AOFF(0));
    }
  }
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    if (kc < kclim) {                                              // wave-uniform
      if (!RD_K1_ROLL) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + RD_AOFF(kc));
          al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + RD_AOFF(kc));
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
          // the WEIGHT fragment is the A operand: the accumulator then holds, per lane (c = lane & 15, g = lane >> 4), the FOUR
          // CONSECUTIVE COLUMNS 16 j + 4 g .. of row 16 rt + c -- 8 contiguous bytes of a row-major plane, 16 of the fp32 staging
          // tile -- instead of four rows of one column (rounds 1-4: a DPP pair exchange + a 4-byte store per element, 27
          // instructions each; the epilogues were a third of the kernel's instructions)
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.h[jj][kc], al[rt], acc[jj][rt], 0, 0, 0);
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.l[jj][kc], ah[rt], acc[jj][rt], 0, 0, 0);
          acc[jj][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.h[jj][kc], ah[rt], acc[jj][rt], 0, 0, 0);
        }
        if (RD_K1_ROLL && kc + 1 < NKC) {                          // in-bounds whatever kclim is: the planes hold NKC steps
          ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * LDX + RD_AOFF(kc + 1));
          al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * LDX + RD_AOFF(kc + 1));
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
#undef RD_AOFF
}
template <int RT>
__device__ __forceinline__ void mma_panel(f32x4 (&acc)[NJ][RT], const __bf16* Ah, const __bf16* Al,
                                          Panel& p, int lane, int kclim = NKC) {
  mma_mid<RT>(acc, Ah, Al, p, lane, kclim, [] {});
}

// srow[rt] = ssum[16 rt + c] (0 beyond F), the aggregate coefficient of the lane's row in each row tile, from the workgroup's LDS copy
// Ss[RT * 16] (written before the first barrier): read where an epilogue needs it.  (Rounds 2-4 carried twelve coefficients per lane
// -- four rows per row tile -- from the kernel's start through both products: 12 of 128 registers.)
template <int RT>
__device__ __forceinline__ void load_srow(float (&srow)[RT], const float* Ss, int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) srow[rt] = Ss[rt * 16 + (lane & 15)];
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

// Plane-sourced, main tiles: the split planes of the tensor are complete in LDS, and a row tile part IS a transposed 32 x 16 block of
// one plane -- lane (c = lane & 15, G = lane >> 4) holds rows 8 G .. 8 G + 7 of column 16 j + c.  ds_read_b64_tr_b16
// (tools/probe_tr16.hip): in a 16-lane group lane i passes the address of row i >> 2, columns 4 (i & 3) .. of a 4 x 16 block and
// receives column i; two reads = the lane's 8 rows, one 16-byte store per lane, a contiguous kilobyte per wave.  No fp32 copy of the
// tensor, no conversion (rounds 2-4 kept an fp32 tile [F][244] for this: 8 strided LDS reads + 24 conversions per slot, 46 KB of LDS).
// Task t = (main tile m, column tile j < jlim, part): dealt to the `nw` waves of the calling group (gw = the wave's index in it).
__device__ __forceinline__ void tstore_planes_main(const __bf16* Ph, const __bf16* Pl, __bf16* tp, const Dim& a, int b, int gw, int nw,
                                                   int lane, int jlim) {
  if (RD_ABL & 4) return;
  typedef short v4s __attribute__((ext_vector_type(4)));
  typedef short v8s __attribute__((ext_vector_type(8)));
  const int i16 = lane & 15, G = lane >> 4;
  const int ntask = a.q * jlim * 2;
  for (int t = gw; t < ntask; t += nw) {
    const int part = t & 1, mj = t >> 1;
    const int m = a.q == 1 ? 0 : mj / jlim, j = mj - m * jlim;
    const int row = 32 * m + 8 * G + (i16 >> 2), col = 16 * j + 4 * (i16 & 3);
    const __bf16* P = part ? Pl : Ph;
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(P + pofs(row, col)));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(P + pofs(row + 4, col)));
    const v8s o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<v8s*>(tp_tile(tp, a.nct, b * a.q + m, j) + part * TILE + lane * 8) = o;
  }
}

// positions of a leftover tile that no sample covers: zeros, written by the workgroup whose sample sits at position 0
// (t / nthr: the calling threads' index and count -- a whole workgroup or one group of it)
__device__ __forceinline__ void tzero_uncovered(__bf16* tp, const Dim& a, int b, int t, int nthr) {
  if (a.rem == 0 || (b % a.per) != 0) return;
  const int first = b / a.per * a.per;
  const int nvalid = min(a.per, a.B - first);
  const int r0 = nvalid * a.rem;
  const int s = a.B * a.q + b / a.per;
  const __bf16 zero = (__bf16)0.f;
  for (int rr = t >> 8; rr < 32 - r0; rr += nthr >> 8) {
    const int n = t & 255, r = r0 + rr;
    if (n < a.K) {
      __bf16* dst = tp_tile(tp, a.nct, s, n >> 4) + ((n & 15) + 16 * (r >> 3)) * 8 + (r & 7);
      st2(dst, zero); st2(dst + TILE, zero);
    }
  }
}

// leftover rows (32 q + li, li < rem) of a tensor whose split planes are complete in LDS, columns [0, KL); by the calling threads
// t of nthr (a whole workgroup or one group of it; nthr a multiple of 256)
__device__ __forceinline__ void tstore_leftover_planes(const __bf16* Ph, const __bf16* Pl, __bf16* tp, const Dim& a,
                                                       int b, int t, int nthr, int KL) {
  if (a.rem == 0 || (RD_ABL & 4)) return;
  const int s = a.B * a.q + b / a.per, slot = b % a.per;
  for (int li = t >> 8; li < a.rem; li += nthr >> 8) {
    const int n = t & 255, r = slot * a.rem + li;
    if (n < KL) {
      __bf16* dst = tp_tile(tp, a.nct, s, n >> 4) + ((n & 15) + 16 * (r >> 3)) * 8 + (r & 7);
      const int o = pofs(32 * a.q + li, n);
      st2(dst, Ph[o]);
      st2(dst + TILE, Pl[o]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Schedule.  The waves form two groups (A: the first half, B: the second; waves w, w + 4, w + 8, w + 12 share a SIMD, so every SIMD
// holds both).  Wherever a phase consists of a matrix-core part and a VALU / LDS / memory part, the two groups run the parts in
// OPPOSITE order, so each SIMD's matrix pipe works for one group while the other splits, stores or issues loads.  Round 5, from the
// phase stamps of all 16 waves inside the captured step (tools/k1_stamps.py --step): the first barrier was released at 9.3 k cycles
// although the observations (cold: ~6 k cycles) were consumed by 7.5 k -- group B requested its weight panel AFTER consuming them,
// behind group A's in the address unit's queue; every wave now requests it before it waits for the observations.  Behind the
// barrier group B does the work that depends on nothing a product makes (positional encoding: off the kernel's tail) while the
// matrix pipe serves group A's product first (older waves win the arbitration), and group A exports X's row tiles behind its
// product while group B multiplies.
template <int RT, int FC, int TC>
__global__ __launch_bounds__(NTHR) void k_msg_fwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RD_TOUCH_CODE(FC == 34 && TC == 60 ? RD_TL_K1_FWD_P19 : RD_TL_K1_FWD);   // own code -> L2 (rd_common.h)
  constexpr int ROWS = RT * 16;
  constexpr size_t PLANES = (size_t)4 * ROWS * LDX * sizeof(__bf16);
  __bf16* Xh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Xl = Xh + ROWS * LDX;
  __bf16* Yh = Xl + ROWS * LDX;
  __bf16* Yl = Yh + ROWS * LDX;
  float* Ys = reinterpret_cast<float*>(smem_raw);        // fp32 [F][LDS_F] staging of Y2, aliases the X planes (F * 976 <= 2 * ROWS * 512)
  float* Ss = reinterpret_cast<float*>(smem_raw + PLANES);   // [ROWS]: ssum (0 beyond F)
  int* LinW = reinterpret_cast<int*>(Ss + ROWS);          // [NWAVE] per-wave "1 + last observed step"
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool grpB = __builtin_amdgcn_readfirstlane(wave) >= GWAVE;       // scalar: the group branches are real branches
  const int gt = tid & (GTHR - 1), gw = __builtin_amdgcn_readfirstlane(wave) & (GWAVE - 1);   // index inside the group
  const Tok tk = tok_of(a);
  const int b = tk.b, sb = tk.sb, L = tk.L;
  const Dim dm = dims_of<FC, TC>(a);
  const int T = dm.T, F = dm.F, K = dm.K, B = dm.B;
  const int nct = dm.nct;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  // column tile j of a layer's output = time steps 4j .. 4j+3.  Layer 2's output at padded steps (t >= L) is never read
  // (rd_plan.h): a wave whose column tiles are all padding skips its layer-2 weight stream, product and epilogue.
  const bool live2 = 4 * __builtin_amdgcn_readfirstlane(wave) < L;

  RD_STAMP_WG_START();
#ifdef RD_K1_EMPTY                                       // timing aid: what the bare launch of this grid costs (results are garbage)
  RD_STAMP_WG_END(); return;
#endif
  RD_STAMP(0);
  Panel pw;
  float sfv = 0.f;                                       // ssum -> LDS (last wave; the epilogues read their rows' coefficients there)
  const int sfi = tid - (NTHR - 64);
  if (sfi >= 0 && sfi < F) sfv = a.ssum[sfi];
  float4 bias1[NJ], bias2[NJ];                           // both layers' biases of the lane's four columns: ahead of the weight stream
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int n0 = 16 * min(wave + NWAVE * jj, nct - 1) + 4 * (lane >> 4);
    bias1[jj] = *reinterpret_cast<const float4*>(a.b1 + n0); bias2[jj] = *reinterpret_cast<const float4*>(a.b2 + n0);
  }

  // ---- observation embedding -> X planes (+ gate byte) ----
  // thread -> cell (t, f) with f fastest: a wave-load of src covers two or three 136-byte row segments (with t
  // fastest it touched 64 cache lines, and the 64 such loads of the workgroup cost more tag look-ups than the
  // whole weight panel).
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
      const int o = pofs(f, 4 * t);                           // the cell's 4 channels: half a 16-byte chunk of row f
      split_store4(Xh + o, Xl + o, x);
      a.mx[(size_t)(m24(sb, total) + m24(t, F) + f)] =        // [slot][t][f]: consecutive lanes, consecutive bytes
          (uint8_t)((x[0] > 0.f ? 1 : 0) | (x[1] > 0.f ? 2 : 0) | (x[2] > 0.f ? 4 : 0) | (x[3] > 0.f ? 8 : 0));
    }
  };
  // dropout masks: one generator call = the 4 channel masks of a (t, f) cell; no memory traffic
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
  // positional encoding + padding mask (code/models_rd.py:28-38,298-299) depend on nothing the products make: group B computes
  // them behind the first barrier, under group A's product; the first pass's operands are requested here, behind the observations
  const bool pe_on = a.times != nullptr;
  const int H = dm.H, npe = L * H;
  float pe_time = 0.f, pe_ts = 1.f;
  if (pe_on && grpB) {
    const int i = min(gt, max(npe - 1, 0));
    const int t = i / H, k = i - t * H;
    pe_time = a.times[(size_t)(m24(t, B) + b)]; pe_ts = a.tscale[k];
  }
  // device seed cell (rd_set_seed_cell) and the sample's length on the scalar path: a vector load would queue behind the panel
  if (a.seed_cell) seed_eff += load_uniform_u64(a.seed_cell);
  int64_t len_b = 0;
  if (pe_on) len_b = (int64_t)load_uniform_u64(reinterpret_cast<const uint64_t*>(a.lengths + b));
  // pads only: X rows >= F and columns >= K (the embedding writes the rest); Y1's pad columns (its epilogue writes every row)
  zero_plane_pads(Xh, ROWS, F, F, K, tid); zero_plane_pads(Xl, ROWS, F, F, K, tid);
  zero_plane_pads(Yh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(Yl, ROWS, ROWS, ROWS, K, tid);
  RD_STAMP(10);
  // every wave requests its W1 panel under the observations' latency (cold in the step: ~6 k cycles), BEFORE it consumes them:
  // the request only queues in the address unit, the barrier below waits for LDS traffic alone
  if (!(RD_ABL & 16)) embed_masks();
  if (!RD_K1_ALATE && (!RD_K1_BLATE || !grpB)) load_panel(pw, wtiles(a, dm, 0, 0), nct, wave, lane);
  RD_STAMP(11);
  if (!(RD_ABL & 16)) {
    embed_consume();
    for (int base = tid + NTHR * UNR; base < total; base += NTHR * UNR) { embed_issue(base); embed_masks(); embed_consume(); }
  }
  if (lane == 0) LinW[wave] = lin_w;
  if (sfi >= 0 && sfi < ROWS) { pin(sfv); Ss[sfi] = sfv; }
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

  // ---- layer 1: Y1 = relu(X W1^T + b1) * ssum;  X leaves as row tiles for dW1 (group B, straight from the planes) --------
  f32x4 acc[NJ][RT];
  zero_acc<RT>(acc);
  if (RD_K1_ALATE && !grpB) load_panel(pw, wtiles(a, dm, 0, 0), nct, wave, lane);
  if (grpB) {                                                // group B: the positional encoding while group A multiplies ...
    if (RD_K1_BLATE || RD_K1_ALATE) load_panel(pw, wtiles(a, dm, 0, 0), nct, wave, lane);
    if (pe_on && !(RD_ABL & 8)) {
      pin(pe_time); pin(pe_ts);
      for (int i = gt; i < npe; i += GTHR) {
        const int t = i / H, k = i - t * H;
        float tm = pe_time, ts = pe_ts;
        if (i != gt) { tm = a.times[(size_t)(m24(t, B) + b)]; ts = a.tscale[k]; }
        const float ang = tm / ts;
        float* row = a.z + (size_t)(m24(tk.row0 + m24(t, tk.rstride), dm.ldz) + F * 4);
        float sn, cs;
        sincosf(ang, &sn, &cs);                                // one argument reduction for the pair
        row[k] = sn;
        row[H + k] = cs;
      }
      for (int t = gt; t < T; t += GTHR) a.mask[(size_t)(m24(b, T) + t)] = (uint8_t)((int64_t)t >= len_b);
    }
  }
  RD_STAMP(13);
  mma_mid<RT>(acc, Xh, Xl, pw, lane, kclim1, [&] {
    if (live2) load_panel_kc<0, NKC / 2>(pw, wtiles(a, dm, 1, 0), nct, wave, lane);   // layer-2 weights, first half of the reduction
  });
  RD_STAMP(3);
  if (live2) load_panel_kc<NKC / 2, NKC>(pw, wtiles(a, dm, 1, 0), nct, wave, lane);   // second half
  RD_STAMP(12);
  if (!grpB) {                                               // ... and group A, whose product the matrix pipe served first, exports X's row tiles
    tstore_planes_main(Xh, Xl, a.tpX, dm, sb, gw, GWAVE, lane, nct);
    tstore_leftover_planes(Xh, Xl, a.tpX, dm, sb, gt, GTHR, K);
    tzero_uncovered(a.tpX, dm, sb, gt, GTHR);
    tzero_uncovered(a.tpY1, dm, sb, gt, GTHR);
  }
  RD_STAMP(14);
  // branch-free epilogue: pad rows carry srow == 0 and land in the planes' pad rows.  Lane (c, g): row 16 rt + c, columns 16 j + 4 g ..
  float srow[RT];                                        // aggregate coefficient of this lane's row in each row tile
  load_srow<RT>(srow, Ss, lane);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct) {                                              // wave-uniform
      const int n0 = 16 * j + 4 * (lane >> 4);
      const float4 bias = bias1[jj];
      int gv = 0;                                               // gate masks Y1 > 0 of this column tile: [RT][4] x 64 bits, two lanes each
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float y[4] = {fmaxf(acc[jj][rt][0] + bias.x, 0.f) * srow[rt], fmaxf(acc[jj][rt][1] + bias.y, 0.f) * srow[rt],
                            fmaxf(acc[jj][rt][2] + bias.z, 0.f) * srow[rt], fmaxf(acc[jj][rt][3] + bias.w, 0.f) * srow[rt]};
        store_split_quad(Yh, Yl, rt * 16 + (lane & 15), n0, y);
        // (rt is a compile-time constant after unrolling; the collector slots are immediates)
        if (rt == 0) { collect_mask<0>(gv, __ballot(y[0] > 0.f)); collect_mask<1>(gv, __ballot(y[1] > 0.f)); collect_mask<2>(gv, __ballot(y[2] > 0.f)); collect_mask<3>(gv, __ballot(y[3] > 0.f)); }
        if (rt == 1) { collect_mask<4>(gv, __ballot(y[0] > 0.f)); collect_mask<5>(gv, __ballot(y[1] > 0.f)); collect_mask<6>(gv, __ballot(y[2] > 0.f)); collect_mask<7>(gv, __ballot(y[3] > 0.f)); }
        if (rt == 2) { collect_mask<8>(gv, __ballot(y[0] > 0.f)); collect_mask<9>(gv, __ballot(y[1] > 0.f)); collect_mask<10>(gv, __ballot(y[2] > 0.f)); collect_mask<11>(gv, __ballot(y[3] > 0.f)); }
      }
      if (lane < RT * 8) reinterpret_cast<int*>(a.m1 + (size_t)(m24(sb, nct) + j) * (RT * 4))[lane] = gv;
    }
  }
  RD_STAMP(4);
  lds_barrier();
  RD_STAMP(5);
  // Y1 -> row tiles for dW2, straight from the planes (every wave takes its share of the 2 q nct tile parts; the leftover rows as before)
  tstore_planes_main(Yh, Yl, a.tpY1, dm, sb, __builtin_amdgcn_readfirstlane(wave), NWAVE, lane, nct);
  tstore_leftover_planes(Yh, Yl, a.tpY1, dm, sb, tid, NTHR, K);

  // ---- layer 2: Y2 = relu(Y1 W2^T + b2) * ssum -> fp32 staging (live column tiles only) ------------
  zero_acc<RT>(acc);
  if (live2) mma_panel<RT>(acc, Yh, Yl, pw, lane);
  RD_STAMP(6);
  load_srow<RT>(srow, Ss, lane);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct && 4 * j < L) {                                 // wave-uniform
      const int n0 = 16 * j + 4 * (lane >> 4);
      const float4 bias = bias2[jj];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {                         // rows >= F land in the staging tile's slack
        *reinterpret_cast<float4*>(Ys + (rt * 16 + (lane & 15)) * LDS_F + n0) =
            make_float4(fmaxf(acc[jj][rt][0] + bias.x, 0.f) * srow[rt], fmaxf(acc[jj][rt][1] + bias.y, 0.f) * srow[rt],
                        fmaxf(acc[jj][rt][2] + bias.z, 0.f) * srow[rt], fmaxf(acc[jj][rt][3] + bias.w, 0.f) * srow[rt]);
      }
    }
  }
  RD_STAMP(7);
  lds_barrier();
  RD_STAMP(8);
  // ---- [F, T*d] -> z[row(t), f*d + c]: thread -> (t, f) with f fastest moves the 4 channels of a cell as one 16-byte
  // LDS read (conflict-free at stride 244) and one 16-byte store; consecutive lanes write consecutive addresses.
  // Live steps only (t < L; L == T on the padded layout).  The cell's gate byte (Y2 > 0, bit c = channel c; what the backward's dz
  // gather needs, in ITS thread order) leaves with it: [slot][t][f], consecutive lanes -> consecutive bytes.
  const int ldz = dm.ldz;
  if (RD_ABL & 8) return;
  for (int i = tid; i < L * F; i += NTHR) {
    int t, f;
    cell_tf(i, F, t, f);
    const float4 y = *reinterpret_cast<const float4*>(Ys + m24(f, LDS_F) + 4 * t);
    float* dst = a.z + (size_t)(m24(tk.row0 + m24(t, tk.rstride), ldz) + 4 * f);
    if ((ldz & 3) == 0) st16f(dst, y);
    else { dst[0] = y.x; dst[1] = y.y; dst[2] = y.z; dst[3] = y.w; }
    a.m2[(size_t)(m24(sb, total) + m24(t, F) + f)] =
        (uint8_t)((y.x > 0.f ? 1 : 0) | (y.y > 0.f ? 2 : 0) | (y.z > 0.f ? 4 : 0) | (y.w > 0.f ? 8 : 0));
  }
  RD_STAMP(9);
  RD_STAMP_WG_END();
}

// ------------------------------------------------------------------------------------------------
// backward (activation side): dZ2 -> dZ1 -> dX -> per-sample dR_u partial; dZ2 and dZ1 leave as row tiles.
// The weight gradients dW_l = dZ_l^T In_l reduce over all B*F rows: rd_msgpass_dw.hip.
// Same two-group schedule as the forward kernel.
// ------------------------------------------------------------------------------------------------
template <int RT, int FC, int TC>
__global__ __launch_bounds__(NTHR) void k_msg_bwd_fused(FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RD_TOUCH_CODE(FC == 34 && TC == 60 ? RD_TL_K1_BWD_P19 : RD_TL_K1_BWD);   // own code -> L2 (rd_common.h)
  constexpr int ROWS = RT * 16;
  constexpr size_t PLANES = (size_t)4 * ROWS * LDX * sizeof(__bf16);
  __bf16* Dh = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* Dl = Dh + ROWS * LDX;
  __bf16* Eh = Dl + ROWS * LDX;
  __bf16* El = Eh + ROWS * LDX;
  float* Sx = reinterpret_cast<float*>(Dh);              // fp32 [F][LDS_F] staging tile of dX, aliases the D planes (dead by then)
  float* Ss = reinterpret_cast<float*>(smem_raw + PLANES);                    // [ROWS]: ssum (0 beyond F)
  float* Rp = reinterpret_cast<float*>(Eh);              // dR_u partial sums [groups][F*4], aliases the E planes (dead by then)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool grpB = __builtin_amdgcn_readfirstlane(wave) >= GWAVE;       // scalar: the group branches are real branches
  const int gt = tid & (GTHR - 1), gw = __builtin_amdgcn_readfirstlane(wave) & (GWAVE - 1);
  const Tok tk = tok_of(a);
  const int b = tk.b, sb = tk.sb, L = tk.L;
  const Dim dm = dims_of<FC, TC>(a);
  const int T = dm.T, F = dm.F, K = dm.K, B = dm.B;
  const int nct = dm.nct;
  const int Fd = F * 4;
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

  RD_STAMP_WG_START();
#ifdef RD_K1_EMPTY
  RD_STAMP_WG_END(); return;
#endif
  RD_STAMP(0);
  if (blockIdx.x == 0 && tid < 192) {                                 // constant operand tiles [ones][zeros][zeros]: column 0 of the 16 is one
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (tid < 64 && (tid & 15) == 0) ? (__bf16)1.f : (__bf16)0.f;
    *reinterpret_cast<bf16x8*>(a.ones + tid * 8) = o;
  }
  Panel pw;

  // ---- dZ2 = dz * ssum * (Y2 > 0): dz read coalesced in [t, f*d+c] order, gated, split and stored straight into the D planes
  // (row f, columns 4t .. 4t+3: half a 16-byte chunk, like the forward's embedding); the gate bits of both layers come from the
  // forward pass (2 x 32 bytes per graph row) and wait in LDS.  (Rounds 2-4 went through an fp32 tile [F][244] and a second pass.)
  // thread -> cell (t, f), f fastest: one 16-byte load per cell (the 4 channels)
  constexpr int GU = CPT;
  const int total = F * T;
  const int ldz = dm.ldz;
  const bool vec4 = (ldz & 3) == 0;
  float4 dd[GU]; int gtt[GU], gfi[GU]; unsigned gb[GU]; float gs[GU];
  auto gather_issue = [&](int base) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int i = min(base + u * NTHR, total - 1);            // clamped duplicates rewrite the same cell
      cell_tf(i, F, gtt[u], gfi[u]);
      const int tl = min(gtt[u], max(L - 1, 0));                // padded steps: a legal address, zeroed at the consumer
      const float* p = a.dz + (size_t)(m24(tk.row0 + m24(tl, tk.rstride), ldz) + 4 * gfi[u]);
      if (vec4) dd[u] = *reinterpret_cast<const float4*>(p);
      else dd[u] = make_float4(p[0], p[1], p[2], p[3]);
      gb[u] = a.m2[(size_t)(m24(sb, total) + m24(tl, F) + gfi[u])];      // the cell's gate byte (the forward's scatter wrote it in this order)
      gs[u] = a.ssum[gfi[u]];
    }
  };
  auto gather_consume = [&]() {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      pin(dd[u]); pin(gb[u]); pin(gs[u]);
      const int f = gfi[u], k = 4 * gtt[u];
      const float sf = gs[u];
      unsigned bits = gb[u];
      if (gtt[u] >= L) bits = 0;
      const float x[4] = {(bits & 1) ? dd[u].x * sf : 0.f, (bits & 2) ? dd[u].y * sf : 0.f, (bits & 4) ? dd[u].z * sf : 0.f,
                          (bits & 8) ? dd[u].w * sf : 0.f};
      const int o = pofs(f, k);
      split_store4(Dh + o, Dl + o, x);
    }
  };
  float sfv = 0.f;
  if (tid >= NTHR - 64 && tid - (NTHR - 64) < F) sfv = a.ssum[tid - (NTHR - 64)];      // last wave: ssum -> LDS (the epilogue's coefficients)
  if (!(RD_ABL & 16)) gather_issue(tid);
  if (!RD_K1_ALATE && (!RD_K1_BLATE || !grpB)) load_panel(pw, wtiles(a, dm, 1, 1), nct, wave, lane);          // W2^T panel queues behind the gather
  // the gate masks Y1 > 0 of this wave's column tile: requested under the gather's latency
  const int g1v = request_masks<RT>(a.m1 + (size_t)(m24(sb, nct) + min(__builtin_amdgcn_readfirstlane(wave), nct - 1)) * (RT * 4), lane);
  RD_STAMP(10);
  // D planes: zero the pads (rows >= F, columns >= K; the gather writes the rest); E planes: pad columns (the epilogue writes every row)
  zero_plane_pads(Dh, ROWS, F, F, K, tid); zero_plane_pads(Dl, ROWS, F, F, K, tid);
  zero_plane_pads(Eh, ROWS, ROWS, ROWS, K, tid); zero_plane_pads(El, ROWS, ROWS, ROWS, K, tid);
  if (tid >= NTHR - 64 && tid - (NTHR - 64) < ROWS) { pin(sfv); Ss[tid - (NTHR - 64)] = sfv; }   // 0 beyond F
  RD_STAMP(11);
  if (!(RD_ABL & 16)) {
    gather_consume();
    for (int base = tid + GU * NTHR; base < total; base += GU * NTHR) { gather_issue(base); gather_consume(); }
  }
  RD_STAMP(13);
  lds_barrier();
  RD_STAMP(1);

  // ---- dZ1 = (dZ2 W2) * ssum * (Y1 > 0); dZ2 leaves as row tiles for dW2 (group B, straight from the planes) ---------
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
  if (RD_K1_ALATE || (RD_K1_BLATE && grpB)) load_panel(pw, wtiles(a, dm, 1, 1), nct, wave, lane);
  RD_STAMP(2);
  mma_mid<RT>(acc, Dh, Dl, pw, lane, kclim2, [&] {
    if (liveX) load_panel_kc<0, NKC / 2>(pw, wtiles(a, dm, 0, 1), nct, wave, lane);   // W1^T, first half of the reduction
  });
  RD_STAMP(4);
  if (liveX) load_panel_kc<NKC / 2, NKC>(pw, wtiles(a, dm, 0, 1), nct, wave, lane);
  if (ract) ru_issue(rtg);
  if (!grpB) {                                               // group A (served first by the matrix pipe) exports dZ2 while group B multiplies
    tstore_planes_main(Dh, Dl, a.tpD2, dm, sb, gw, GWAVE, lane, jlim);
    tstore_leftover_planes(Dh, Dl, a.tpD2, dm, sb, gt, GTHR, 16 * jlimL);
    tzero_uncovered(a.tpD2, dm, sb, gt, GTHR);
    tzero_uncovered(a.tpD1, dm, sb, gt, GTHR);
  }
  RD_STAMP(14);
  float srow[RT];
  load_srow<RT>(srow, Ss, lane);
  static_assert(NJ == 1, "the gate masks are loaded for one column tile per wave");
  unsigned long long g1m[RT * 4];
  unpack_masks<RT>(g1v, g1m);
  {
    const int j = wave;
    if (j < nct) {                                              // wave-uniform; body is branch-free
      const int n0 = 16 * j + 4 * (lane >> 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {                         // pad rows: srow == 0
        const float gq[4] = {gate_by_mask(acc[0][rt][0] * srow[rt], g1m[rt * 4 + 0]), gate_by_mask(acc[0][rt][1] * srow[rt], g1m[rt * 4 + 1]),
                             gate_by_mask(acc[0][rt][2] * srow[rt], g1m[rt * 4 + 2]), gate_by_mask(acc[0][rt][3] * srow[rt], g1m[rt * 4 + 3])};
        store_split_quad(Eh, El, rt * 16 + (lane & 15), n0, gq);
      }
    }
  }
  RD_STAMP(5);
  lds_barrier();
  RD_STAMP(15);
  // dZ1 -> row tiles for dW1, straight from the planes
  tstore_planes_main(Eh, El, a.tpD1, dm, sb, __builtin_amdgcn_readfirstlane(wave), NWAVE, lane, nct);
  tstore_leftover_planes(Eh, El, a.tpD1, dm, sb, tid, NTHR, K);

  // ---- dX = dZ1 W1 -> fp32 staging (the D planes are dead); observed column tiles only -----------
  zero_acc<RT>(acc);
  if (liveX) mma_panel<RT>(acc, Eh, El, pw, lane);
  RD_STAMP(6);
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int j = wave + NWAVE * jj;
    if (j < nct && 4 * j < lin) {
      const int n = 16 * j + 4 * (lane >> 4);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        *reinterpret_cast<float4*>(Sx + (rt * 16 + (lane & 15)) * LDS_F + n) =
            make_float4(acc[jj][rt][0], acc[jj][rt][1], acc[jj][rt][2], acc[jj][rt][3]);
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
  RD_STAMP_WG_END();
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
  const size_t lds = (size_t)4 * RT * 16 * LDX * sizeof(__bf16) + (size_t)RT * 16 * sizeof(float) + (size_t)NWAVE * sizeof(int);   // planes, ssum, LinW (forward)
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
  // staging tile [F][244] fp32 must fit inside two bf16 planes [RT*16][256]; d_ob == 4 only
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
  a.tpX = (__bf16*)tpX; a.tpY1 = (__bf16*)tpY1; a.m1 = (uint64_t*)m1; a.m2 = (uint8_t*)m2; a.mx = (uint8_t*)mx;
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
  a.m1 = (uint64_t*)const_cast<void*>(m1); a.m2 = (uint8_t*)const_cast<void*>(m2); a.mx = (uint8_t*)const_cast<void*>(mx);
  a.dz = dz; a.ldz = ldz; a.tpD1 = (__bf16*)tpD1; a.tpD2 = (__bf16*)tpD2; a.ones = (__bf16*)ones; a.rupart = rupart;
  a.p_drop = p_drop; a.stamps = g_stamps ? g_stamps + K1_STAMP_WORDS : nullptr;
  a.plan = token_plan();
  a.lin = reinterpret_cast<int*>(reinterpret_cast<char*>(const_cast<void*>(mx)) + k1::lin_offset(L.B, L.T, L.F));
  return launch_fused_shape(a, L, true, st);
}

}  // namespace rd
