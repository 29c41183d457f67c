// rd_encfuse.hip -- the row-local half of a post-norm encoder layer as ONE launch per direction.
//
// nn.TransformerEncoderLayer (torch/nn/modules/transformer.py:799-983, used at code/models_rd.py:235-237,358) is, apart from
// the attention core, a chain of per-token operations:
//   forward   attn -> out_proj -> +x, dropout, LayerNorm1 -> x1 -> linear1, ReLU, dropout -> h -> linear2 -> +x1, dropout,
//             LayerNorm2 -> y
//   backward  dy -> LayerNorm2' -> (df, ds2) -> linear2' (gated) -> du -> linear1' + ds2 -> dx1 -> LayerNorm1' -> (dout, ds1)
//             -> out_proj' -> d attn
// As separate row-block products (rd_rowgemm.hip) each link was its own launch: three forward, three backward, every one a
// 15-20 us chain (load rows -> split -> multiply -> stage -> epilogue -> store) over ONE round of workgroups, with the
// intermediate written to HBM and read back by the next.  Here a workgroup keeps its token rows in LDS through the whole chain:
// the rows are read once, only what the backward pass (or the attention core) needs is written, and the three weight panels
// stream from L2 behind one another.  Same arithmetic, same dropout quads and same summation order of every product as the
// kernels it replaces (k_rowgemm<.., LN> / <.., LNB>); the LayerNorm row sums are grouped differently (rounding-level differences).
//
// Round 5 (second half): the products run with the WEIGHT fragment as the MFMA's A operand, so an accumulator is a column quad of a
// row (one 16-byte store / one dropout quad / one gate byte) -- see mma.  The two nhid-wide products no longer pass through a
// fp32 stage: their epilogues (bias + ReLU + dropout + split in the forward chain, the h > 0 gate in the backward chain) run on
// the accumulators, per wave, while the other waves of the SIMD still multiply -- one barrier, one stage round trip and the index
// arithmetic of a flat 3.4-quads-per-thread loop less per chain; the stage shrank to the D-wide products' 164 columns, and the
// LDS it freed keeps x1 (forward) and ds2 (backward) as fp32 rows where rounds 3-4 wrote them to memory and read them back; the
// last product of the backward chain stores d attn straight from its accumulators.
//
// Rows per workgroup.  A workgroup needs ~100-150 KB of LDS, so ONE fits a CU and a launch is a whole number of rounds over the
// `ncu` CUs; these chains are latency-bound (a round costs ~20 us whether it carries 32 or 48 rows per workgroup), so the
// kernels pick their block height ON THE DEVICE from the live row count: 32 rows while that gives at most one workgroup per CU,
// 48 rows (three row tiles) when 32 would spill a few blocks into a second round (P19, B = 256: 8000-8500 live rows against
// 32 x 256 = 8192 -- half of all batches).  The grid is sized for 32-row blocks of the padded row count; blocks beyond the live
// rows exit at once.
//
// Envelope: ceil(D / 32) == 5 and ceil(nhid / 32) == 9 (P19: D = 152, nhid = 272), D % 4 == 0, nhid % 4 == 0, bf16 modes.
#include <stdlib.h>

#include "rd_common.h"
#include "rd_rng.h"
#include "rd_trailing.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// own-code touch lengths (bytes behind the s_getpc_b64) of the twelve instantiations: kernel size - prologue - 64, rounded down to 128
constexpr int EF_WV = 16, EF_THR = 64 * EF_WV;
constexpr int EF_RTMAX = 3;                          // row tiles of the tall variant
constexpr int KCD = 5, KCH = 9;                      // reduction steps of 32 for D and nhid
constexpr int KPD = 32 * KCD, KPH = 32 * KCH;        // 160, 288
// bf16 plane row strides: columns + 32 BYTES.  gfx950 services a ds_read_b128 in 16-lane groups {r 0-3 G, r 12-15 G, r 4-11 G+1} over 64
// banks: the MFMA fragment read (lane: row r, 16-byte chunk G) is conflict-free exactly when the row stride is 32 bytes mod 64
// (tools/lds_conflicts.py, measured by tools/probe_ldsfrag.hip: 235 B/clk/CU against 127 for the "+ 8 elements" of rounds 1-4).
constexpr int LDD = KPD + 16, LDH = KPH + 16;
constexpr int STG = KPD + 8;                         // fp32 stage row stride: the D-wide products' output columns + 32 bytes (the
                                                     // nhid-wide products finish on their accumulators: no stage)

template <int KC>
struct Panel { bf16x8 h[KC], l[KC]; };

// native operand tiles [tile j][kc][hi, lo][64 lanes][8] (k_wsplit, rd_rowgemm.hip); reduction steps [KC0, KC1) of tile j
// (a long panel is requested in two halves to bound the registers in flight)
// ALL: every wave is known to own a tile (the nhid-wide products: nhid > 256 = 16 tiles, encfuse_ok) -- no test.  The test is
// a branch around the loads, and behind a branch the compiler's wait-count bookkeeping assumes the path that requested nothing:
// the LayerNorm epilogue that runs under the streaming panel then waited (vmcnt(0)) for the panel instead of its own, older rows.
template <int KC, int KC0 = 0, int KC1 = KC, bool ALL = false>
__device__ __forceinline__ void load_panel(Panel<KC>& p, const __bf16* __restrict__ Wt, int ntiles, int j, int lane) {
  // a wave without a column tile requests nothing (wave-uniform): 6 of the 16 waves have none in the D-wide products, and their
  // 10-18 KB of loads each only lengthened the address unit's queue in front of everybody's stores
  if (!ALL && __builtin_amdgcn_readfirstlane(j) >= ntiles) return;
  const __bf16* t = Wt + (size_t)j * (KC * 2 * 512) + lane * 8;
#pragma unroll
  for (int kc = KC0; kc < KC1; ++kc) {
    p.h[kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 0) * 512);
    p.l[kc] = *reinterpret_cast<const bf16x8*>(t + (kc * 2 + 1) * 512);
  }
}

// acc[rt] = (A[rows 16 rt .., :] * panel^T)^T, split-bf16 (lo*hi + hi*lo + hi*hi), A planes in LDS.  The WEIGHT fragment is the MFMA's
// A operand and the row fragment its B operand (both fragments have the same lane layout: index lane & 15, reduction steps
// 8 (lane >> 4) ..), so the accumulator comes out transposed: lane (i = lane & 15, G = lane >> 4) holds the four CONSECUTIVE COLUMNS
// 16 j + 4 G .. + 3 of row 16 rt + i -- one column quad, i.e. one 16-byte stage store, one dropout quad, one gate byte -- instead of
// four rows of one column.  Same products in the same order: the same bits (rd_msgpass_fused.hip, round 5).
// One (hi, lo) fragment pair is live per row tile and step: the three products of (kc, rt) are issued, then the pair of (kc + 1, rt)
// is requested into the same registers (the rolled order of rd_msgpass_fused.hip's mma_mid) -- the LDS reads travel under the other
// row tiles' MFMAs, and the tall variant keeps 8 fragment registers fewer alive beside its 72-register panels.
// mid() / mid2(): issued once behind reduction step MIDK / MIDK2 -- global loads that have no registers to live in before the panel's
// first steps are consumed (the tall variant: the last steps of a 72-register panel, the next phase's saved rows).
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <int KC, int RT, int MIDK = -1, typename Mid = NoMid, int MIDK2 = -1, typename Mid2 = NoMid>
__device__ __forceinline__ void mma(f32x4 (&acc)[RT], const __bf16* Ah, const __bf16* Al, int lda, const Panel<KC>& p, int lane, int one,
                                    Mid mid = Mid(), Mid2 mid2 = Mid2()) {
  const int aoff = (lane & 15) * lda + 8 * (lane >> 4);
  bf16x8 ah[RT], al[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (one) {                                         // RD_PREC_BF16 (hi * hi only): read-then-multiply steps
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * lda + aoff + kc * 32);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.h[kc], ah[rt], acc[rt], 0, 0, 0);
      if (kc == MIDK) mid();
      if (kc == MIDK2) mid2();
    }
    return;
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * lda + aoff);
    al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * lda + aoff);
  }
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.h[kc], al[rt], acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.l[kc], ah[rt], acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p.h[kc], ah[rt], acc[rt], 0, 0, 0);
      if (kc + 1 < KC) {
        ah[rt] = *reinterpret_cast<const bf16x8*>(Ah + rt * 16 * lda + aoff + (kc + 1) * 32);
        al[rt] = *reinterpret_cast<const bf16x8*>(Al + rt * 16 * lda + aoff + (kc + 1) * 32);
      }
    }
    if (kc + 1 < KC) {                               // pin the issue order: three MFMAs, the two reads they free, ...
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    if (kc == MIDK || kc == MIDK2) {
      __builtin_amdgcn_sched_barrier(0);
      if (kc == MIDK) mid();
      if (kc == MIDK2) mid2();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// accumulators of column tile j (transposed form, see mma) -> stage[row][16 j + 4 G ..]: one 16-byte store per row tile.  Rows of
// 672 bytes = 42 chunks: in gfx950's 16-lane groups {rows 0-3, 12-15 of chunk G; rows 4-11 of chunk G + 1} the chunks 10 r (+ 1) mod 16
// of the 256-byte bank window are all different (tools/lds_conflicts.py's model, tests/test_lds_layouts.py; with KPD + 4 floats the
// group was 2-way conflicted).  The LayerNorm passes' reads of the stage (a 16-lane row reads 16 consecutive chunks of one stage row)
// stay 2-way against the neighbouring row -- three reads per lane and pass.
template <int RT>
__device__ __forceinline__ void to_stage(float* stage, const f32x4 (&acc)[RT], int j, int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<f32x4*>(stage + (rt * 16 + (lane & 15)) * STG + j * 16 + 4 * (lane >> 4)) = acc[rt];
}

__device__ __forceinline__ void split_store4(__bf16* ph, __bf16* pl, const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  bf16x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) { h[q] = (__bf16)x[q]; l[q] = (__bf16)(x[q] - (float)h[q]); }
  *reinterpret_cast<bf16x4*>(ph) = h;
  *reinterpret_cast<bf16x4*>(pl) = l;
}

// split planes [16 RT][lda] (complete, behind a barrier) -> row tiles [chunk of 32 rows][column tile j][hi, lo][64][8] of the
// weight-gradient stream (rd_tile_wgrad.hip).  A block may start in the middle of a chunk (48-row blocks), so the unit is a
// 16-row group = lanes 32 half .. 32 half + 31 of a tile part.  ds_read_b64_tr_b16 (rd_rowgemm.hip has the lane map): in a
// 16-lane group lane i passes the address of row i >> 2, columns 4 (i & 3) .. of a 4 x 16 block and receives column i; two reads
// give lane (column i, group g) the rows 8 g .. 8 g + 7 of its column.  Lanes 0-31 work on the hi plane, 32-63 on the lo plane.
template <int RT>
__device__ __forceinline__ void export_tiles(const __bf16* Ph, const __bf16* Pl, int lda, __bf16* xt, int nct, int m0, int M, int wave,
                                             int lane) {
  typedef short v4s __attribute__((ext_vector_type(4)));
  typedef short v8s __attribute__((ext_vector_type(8)));
  const int i16 = lane & 15, g = (lane >> 4) & 1, plane = lane >> 5;
  const int nchunk = (M + 31) >> 5;                    // chunks beyond the live rows are never read (and may not exist)
  for (int t = wave; t < nct * RT; t += EF_WV) {
    const int rt = t / nct, j = t - rt * nct;
    const int grp = (m0 >> 4) + rt;                    // global 16-row group
    if ((grp >> 1) >= nchunk) continue;
    const __bf16* src = (plane ? Pl : Ph) + (16 * rt + 8 * g + (i16 >> 2)) * lda + 16 * j + 4 * (i16 & 3);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(src));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(src + 4 * lda));
    const v8s o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    const int slot = 32 * (grp & 1) + 16 * g + i16;    // lane slot inside the tile part
    *reinterpret_cast<v8s*>(xt + (((size_t)(grp >> 1) * nct + j) * 2 + plane) * 512 + slot * 8) = o;
  }
}

// A block without a live row still owns the 16-row groups it would have exported; where such a group falls into the last live
// 32-row chunk (48-row blocks can end in the middle of one) the weight-gradient stream reads it: zeros, not last step's rows.
template <int RT>
__device__ __forceinline__ void zero_dead_groups(__bf16* xt, int nct, int m0, int M, int tid) {
  if (!xt) return;
  const int nchunk = (M + 31) >> 5;
  for (int rt = 0; rt < RT; ++rt) {
    const int grp = (m0 >> 4) + rt;
    if ((grp >> 1) >= nchunk) continue;
    // 32 lanes x 16 bytes per (column tile, plane): thread -> (j, plane, slot)
    for (int i = tid; i < nct * 2 * 32; i += EF_THR) {
      const int slot = i & 31, jp = i >> 5;
      *reinterpret_cast<float4*>(xt + ((size_t)(grp >> 1) * nct * 2 + jp) * 512 + (32 * (grp & 1) + slot) * 8) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// The nhid-wide planes' columns beyond the last column TILE (16 ntH .. KPH: the reduction of the product that reads them runs over
// KPH): zero, once per kernel -- the epilogues that fill the planes write whole tiles (and zeros beyond nhid inside the last one).
template <int RT>
__device__ __forceinline__ void zero_pad_columns(__bf16* Hh, __bf16* Hl, int ntH, int tid) {
  const int c0 = 16 * ntH, nq = (KPH - c0) >> 2;               // uniform
  for (int i = tid; i < 16 * RT * nq; i += EF_THR) {
    const int r = i / nq, q = i - r * nq;
    bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    *reinterpret_cast<bf16x4*>(Hh + r * LDH + c0 + 4 * q) = z;
    *reinterpret_cast<bf16x4*>(Hl + r * LDH + c0 + 4 * q) = z;
  }
}

// Dropout keep bits of a LayerNorm epilogue, made by the waves that IDLE in the phase before it -- the six waves without a column tile
// while the other ten run a D-wide product (out_proj, linear2, linear1'), the waves without a LayerNorm pass while the others wait
// for their rows at the start of the backward chain -- instead of inside the LayerNorm passes, which are bound by the vector ALUs
// (three Threefry calls = 141 of a pass's ~450 vector instructions, three waves per SIMD in a pass).  Used where the phase before is
// long enough to hide ~100 instructions per quad on two helper waves per SIMD: LayerNorm2 (behind linear2, 7.5 k cycles) and
// LayerNorm1' (behind linear1', 7.5 k).  Measured slower, and therefore still Threefry calls inside the pass: LayerNorm1 (out_proj
// is 4.1 k cycles: +1.2 k there for 0.9 k), LayerNorm2' (nothing in front of it but the first phase's wait for the rows: +3.9 k for
// 1.8 k by the four pass-less waves); all of a kernel's dropout decisions by all waves in its first phase: +7.4 k for 5.6 k.
// mk: [16 RT rows][MKQ] bytes, bit c of byte q = element 4 q + c of the row is kept (same quads, same comparison as
// uniform4(..) >= p: the masks, and with them every result, are unchanged); idx / nthr: this thread's index among the helpers.
constexpr int MKQ = KPD / 4;                          // mask bytes per row (one per column quad, D <= KPD)
// NW: helper waves (idx = 0 .. 64 NW - 1).  A rolled loop, one quad per trip: four quads per trip as independent straight-line chains
// (to hide a lone wave's dependent-issue latency) measured SLOWER -- the helpers are bound by vector-ALU throughput (~100
// instructions per quad with the index arithmetic), and the fixed trip count added wasted quads.
template <int RT, int NW>
__device__ __forceinline__ void drop_masks(uint8_t* mk, uint64_t seed, uint32_t site, int D, int m0, float p, int idx) {
  const int qpr = D >> 2, per = 16 * RT * qpr;
  for (int e = idx; e < per; e += 64 * NW) {
    const int r = e / qpr, q = e - r * qpr;
    const float4 v = uniform4(seed, site, (uint64_t)(m0 + r) * qpr + q);
    mk[r * MKQ + q] = (uint8_t)((v.x >= p ? 1 : 0) | (v.y >= p ? 2 : 0) | (v.z >= p ? 4 : 0) | (v.w >= p ? 8 : 0));
  }
}

__device__ __forceinline__ float wsum(float v) { return wave_sum64_dpp(v); }

static unsigned long long* g_ef_stamps = nullptr;    // debug (tools/encfuse_timing.py): clock64 per phase, every wave of workgroup 0
#define EFSTAMP(i)                                                                                       \
  do {                                                                                                   \
    if (a.stamps && blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.stamps[(threadIdx.x >> 6) * 16 + (i)] = clock64(); \
  } while (0)

// block height for `M` live rows on `ncu` CUs: 2 row tiles unless that needs a second round and 3 do not
__device__ __forceinline__ int pick_rt(int M, int ncu) { return (M > 32 * ncu && M <= 48 * ncu) ? 3 : 2; }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct PostFwdArgs {
  const float* attn; const float* x;                // [M, D] each: attention output, layer input (residual of LayerNorm1)
  const __bf16 *Wo, *W1, *W2;                       // operand tiles of out_proj [D,D], linear1 [H,D], linear2 [D,H]
  const float *bo, *b1, *b2, *g1, *be1, *g2, *be2;
  float *s1, *x1, *st1, *h, *s2, *y, *st2;          // saved pre-norm sums, normalised rows, (mean, rstd), FFN hidden, output
  uint8_t* hgate;                                   // LEAN form: [M][H / 4] gate bytes (bit c: h[4 q + c] > 0) instead of the fp32 h and x1
  __bf16 *xt_attn, *xt_x1, *xt_h;                   // row tiles for the weight gradients (null: not wanted)
  int M, D, H, ncu;
  float p; uint64_t seed; uint32_t site_ao, site_fh, site_fo; const uint64_t* seed_cell;
  const int32_t* mlive;
  int one;
  unsigned long long* stamps;
};

// ---- LayerNorm passes: FOUR rows per wave pass, one per 16-lane DPP row -------------------------------------------------------
// Lane (g = lane >> 4, i = lane & 15) works on row 4 pass + g and on the column quads i, i + 16, i + 32 of it (D <= 188; at
// D = 152 the third quad exists for i < 6).  With one row per pass (64 lanes, 38 of them on the 38 quads of a row) a LayerNorm
// phase was ~180 instructions per row, three rows per wave, four waves per SIMD: 10-11 k cycles of pure issue per phase.  Here the
// row sums are 4-step rotate-add butterflies inside a DPP row (every lane of the row ends with the same bits), a pass costs about
// twice a row's instructions for four rows, and 4 RT of the 16 waves run one pass each.
constexpr int LNQ = 3;                                // column quads per lane

__device__ __forceinline__ float row16_allsum(float v) {
#define RD_ROR_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  RD_ROR_ADD(0x128);   // row_ror:8
  RD_ROR_ADD(0x124);   // row_ror:4
  RD_ROR_ADD(0x122);   // row_ror:2
  RD_ROR_ADD(0x121);   // row_ror:1
#undef RD_ROR_ADD
  return v;
}
// EVEN 16-lane rows: x + (the same lane of the next row), i.e. rows 0 + 1 and 2 + 3; the odd rows' results are not used.
// v_permlane16_swap_b32 (gfx950) exchanges the odd rows of its first operand with the even rows of its second: the even rows of
// a + b are right whichever of the two directions the hardware completes (tools/probe_permlane_swap.hip: measured).
__device__ __forceinline__ float rowpair_sum(float x) {
  float a = x, b = x;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// residual / saved rows of a pass: quad k of row m0 + 4 wave + g from a [M, D] tensor (zero beyond D or M)
__device__ __forceinline__ void ln_load3(float4 (&r)[LNQ], const float* src, int D, int m0, int M, int wave, int lane) {
  const long m = m0 + 4 * wave + (lane >> 4);
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * ((lane & 15) + 16 * k);
    r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < D && m < M) r[k] = *reinterpret_cast<const float4*>(src + m * D + c);
  }
}

// One pass of the LayerNorm epilogue: sv = residual + dropout(stage + bias); writes the pre-norm sum, the normalised row and the
// statistics; normalised rows also go to the split planes (Ph != null; zeros beyond D / M: the next product reads them) and, as fp32,
// to `keep` ([rows][KPD] in LDS; null: not wanted).
// bs / gg / bb: the bias, gamma, beta vectors in LDS (zero padded to KPD).
// PRE: the dropout keep bits come from mk (drop_masks); otherwise from Threefry calls here (seed, site).
template <bool PRE>
__device__ __forceinline__ void ln_rows4(const float* stage, const float4 (&res)[LNQ], const float* bs, const float* gg, const float* bb, int D,
                                         int m0, int M, int wave, int lane, float p, float inv_keep, const uint8_t* mk, uint64_t seed,
                                         uint32_t site, float* s_out, float* y_out, float* stats, __bf16* Ph, __bf16* Pl, float* keep = nullptr) {
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int g = lane >> 4, i16 = lane & 15, rl = 4 * wave + g;
  const long m = m0 + rl;
  const bool rok = m < M;
  float4 sv[LNQ];
  float part = 0.f;
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * (i16 + 16 * k);
    const bool ok = rok && c < D;
    float4 t = zero4;
    if (c < D) {
      t = *reinterpret_cast<const float4*>(stage + rl * STG + c);
      const float4 b4 = *reinterpret_cast<const float4*>(bs + c);
      t.x += b4.x; t.y += b4.y; t.z += b4.z; t.w += b4.w;
    }
    if (p > 0.f && c < D) {
      if constexpr (PRE) {                            // keep bits made by the idle waves of the phase before (drop_masks)
        const unsigned kb = mk[rl * MKQ + (c >> 2)];
        t.x *= (kb & 1) ? inv_keep : 0.f; t.y *= (kb & 2) ? inv_keep : 0.f;
        t.z *= (kb & 4) ? inv_keep : 0.f; t.w *= (kb & 8) ? inv_keep : 0.f;
      } else {
        const float4 u = uniform4(seed, site, ((uint64_t)m * D + c) >> 2);
        t.x *= u.x >= p ? inv_keep : 0.f; t.y *= u.y >= p ? inv_keep : 0.f;
        t.z *= u.z >= p ? inv_keep : 0.f; t.w *= u.w >= p ? inv_keep : 0.f;
      }
    }
    sv[k] = zero4;
    if (ok) {
      sv[k] = make_float4(res[k].x + t.x, res[k].y + t.y, res[k].z + t.z, res[k].w + t.w);
      *reinterpret_cast<float4*>(s_out + m * D + c) = sv[k];
    }
    part += (sv[k].x + sv[k].y) + (sv[k].z + sv[k].w);
  }
  const float mean = row16_allsum(part) / D;
  float4 d[LNQ];
  float vpart = 0.f;
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * (i16 + 16 * k);
    d[k] = zero4;
    if (rok && c < D) d[k] = make_float4(sv[k].x - mean, sv[k].y - mean, sv[k].z - mean, sv[k].w - mean);
    vpart += (d[k].x * d[k].x + d[k].y * d[k].y) + (d[k].z * d[k].z + d[k].w * d[k].w);
  }
  const float rstd = rsqrtf(row16_allsum(vpart) / D + 1e-5f);
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * (i16 + 16 * k);
    float4 o = zero4;
    if (rok && c < D) {
      const float4 g4 = *reinterpret_cast<const float4*>(gg + c), b4 = *reinterpret_cast<const float4*>(bb + c);
      o = make_float4(d[k].x * rstd * g4.x + b4.x, d[k].y * rstd * g4.y + b4.y, d[k].z * rstd * g4.z + b4.z, d[k].w * rstd * g4.w + b4.w);
      if (y_out) *reinterpret_cast<float4*>(y_out + m * D + c) = o;
    }
    if (Ph && c < KPD) split_store4(Ph + rl * LDD + c, Pl + rl * LDD + c, o);
    if (keep && c < KPD) *reinterpret_cast<float4*>(keep + rl * KPD + c) = o;   // fp32 copy for the next LayerNorm's residual (this lane reads it back)
  }
  if (i16 == 0 && rok) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
}

// DC / HC: model width and FFN width as compile-time constants (0: read from the arguments).  These chains are instruction-issue
// bound; with the widths known the row / quad index arithmetic (64-bit multiplies at quarter rate, divisions) folds away.
// LEAN (the training step on the token plan, whose backward is the fused chain + the tile stream): what only the UNFUSED backward
// or the chain itself would read is not written -- the FFN hidden h leaves as row tiles (weight gradient) and as one gate BYTE
// per column quad (the backward needs h > 0, nothing else: 9.2 MB of fp32 written and read back per layer at P19), and the
// normalised x1 (residual of LayerNorm2) is recomputed from the saved pre-norm sum and statistics instead of stored and re-read.
template <int RT, int DC, int HC, bool LEAN, bool ONE>
__device__ __forceinline__ void post_fwd_body(const PostFwdArgs& a, unsigned char* esm, int M) {
  constexpr int ROWS = 16 * RT;
  __bf16* Ah = reinterpret_cast<__bf16*>(esm);                 // [ROWS][LDD]: attn, then x1
  __bf16* Al = Ah + ROWS * LDD;
  __bf16* Hh = Al + ROWS * LDD;                                // [ROWS][LDH]: h
  __bf16* Hl = Hh + ROWS * LDH;
  float* stage = reinterpret_cast<float*>(Hl + ROWS * LDH);    // [ROWS][STG]
  // per-column vectors [bo | g1 | be1 | b2 | g2 | be2] (KPD each) and b1 (KPH), zero padded: fetched ONCE, in the first phase, so that
  // no epilogue has a global load of its own -- those queue behind the weight panel requested just before them (loads return in order)
  float* cst = stage + ROWS * STG;
  // x1 (LayerNorm1's output, the residual of LayerNorm2) as fp32 [ROWS][KPD]: written and read back by the SAME lane (pass layout), so
  // no barrier orders it.  Rounds 3-4 re-read it from memory (LEAN: the pre-norm sum, re-normalised); the stage of the nhid-wide
  // product that stood here is gone (h_epilogue).
  float* x1r = cst + 6 * KPD + KPH;
  uint8_t* mk = reinterpret_cast<uint8_t*>(x1r + ROWS * KPD);  // dropout keep bits of LayerNorm2 (drop_masks: by idle waves)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: `if (wave < ..)` is a branch, not a mask
  const int m0 = blockIdx.x * ROWS;
  const int D = DC ? DC : a.D, H = HC ? HC : a.H;
  const int ntD = (D + 15) >> 4, ntH = (H + 15) >> 4;
  if (m0 >= M) {
    zero_dead_groups<RT>(a.xt_attn, ntD, m0, M, tid); zero_dead_groups<RT>(a.xt_x1, ntD, m0, M, tid); zero_dead_groups<RT>(a.xt_h, ntH, m0, M, tid);
    return;
  }
  uint64_t seed = a.seed;
  const float inv_keep = 1.0f / (1.0f - a.p);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  EFSTAMP(0);
  if (a.stamps && tid == 0) { a.stamps[256 + 2 * blockIdx.x] = wall_clock64(); a.stamps[256 + 2 * blockIdx.x + 1] = clock64(); }
  zero_pad_columns<RT>(Hh, Hl, ntH, tid);                      // columns 16 ntH .. KPH of the h planes (linear2 reduces over KPH)
  // ---- attention rows -> split planes (zero padded to KPD columns, rows beyond M zero) ----
  constexpr int kq = KPD / 4;
  constexpr int NIT = (ROWS * kq + EF_THR - 1) / EF_THR;
  float4 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * EF_THR;
    const int r = i / kq, k = 4 * (i - r * kq);
    v[it] = zero4;
    if (r < ROWS && m0 + r < M && k < D) v[it] = *reinterpret_cast<const float4*>(a.attn + (long)(m0 + r) * D + k);
  }
  Panel<KCD> po;
  load_panel<KCD>(po, a.Wo, ntD, wave, lane);
  // The per-column vectors, requested BEHIND the rows and the panel and written to LDS behind the row split (first read: LayerNorm1).
  // 23 wave-wide slots of 64 elements -- 3 per D-wide vector, 5 for b1 --, wave w takes slot w and (w < 7) slot 16 + w: the vector of
  // a slot is wave-uniform (scalar pointer select: a per-LANE select among the seven pointers became a pointer table in scratch),
  // every load is unconditional from a clamped address.  Rounds 3-4 fetched them first thing, one vector after the other, each
  // load waited for before its LDS store: four dependent round trips (the first one cold: ~2.6 k cycles, ~0.85 k each after
  // that) in front of the row loads of waves 0-4, which every other wave then waited for at the first barrier.
  float cval[2]; int cdst[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int sl = min(wave + 16 * r, 22);                     // scalar (round 2: waves 7 .. 15 repeat slot 22, not stored)
    const int vec = sl < 18 ? sl / 3 : 6, part = sl < 18 ? sl - 3 * vec : sl - 18;
    const float* src = vec == 0 ? a.bo : vec == 1 ? a.g1 : vec == 2 ? a.be1 : vec == 3 ? a.b2 : vec == 4 ? a.g2 : vec == 5 ? a.be2 : a.b1;
    const int lim = vec < 6 ? D : H, n = vec < 6 ? KPD : KPH, e = 64 * part + lane;
    cval[r] = src[min(e, lim - 1)];
    if (e >= lim) cval[r] = 0.f;
    cdst[r] = (e < n && wave + 16 * r <= 22) ? vec * KPD + e : -1;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (a.seed_cell) seed += load_uniform_u64(a.seed_cell);
  // the second-round (tile, row tile) unit of this wave in the nhid-wide product, if any (see linear1 below)
  const int unit = EF_WV - 1 - wave;                           // scalar
  const bool more = unit < (ntH - EF_WV) * RT;                 // scalar
  const int j2 = EF_WV + unit / RT, rt2 = unit - (unit / RT) * RT;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * EF_THR;
    const int r = i / kq, k = 4 * (i - r * kq);
    if (r < ROWS) split_store4(Ah + r * LDD + k, Al + r * LDD + k, v[it]);
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
    if (cdst[r] >= 0) cst[cdst[r]] = cval[r];
  EFSTAMP(1);
  lds_barrier();
  EFSTAMP(2);
  if (a.xt_attn) export_tiles<RT>(Ah, Al, LDD, a.xt_attn, ntD, m0, M, wave, lane);

  // ---- out_proj ----
  constexpr int HW0 = 10, NHW = EF_WV - HW0;                    // helper waves 10 .. 15: no column tile in a D-wide product (D <= 160)
  const int hidx = (wave - HW0) * 64 + lane;
  if (wave < ntD) {
    f32x4 acc[RT];
    mma<KCD, RT>(acc, Ah, Al, LDD, po, lane, (int)ONE);
    to_stage<RT>(stage, acc, wave, lane);
  }
  EFSTAMP(3);
  constexpr int NPASS = 4 * RT;                                // LayerNorm passes of 4 rows: waves 0 .. NPASS - 1 run one each
  const bool lnw = wave < NPASS;
  float4 xr[LNQ];                                              // residual rows of LayerNorm1, in the pass layout (ln_rows4)
  if (lnw) ln_load3(xr, a.x, D, m0, M, wave, lane);
  lds_barrier();                                               // stage complete; every wave is done reading the attn planes
  EFSTAMP(4);
  Panel<KCD> p1;                                               // linear1 (column tiles 0..15) streams in under the LayerNorm epilogue
  load_panel<KCD, 0, KCD, true>(p1, a.W1, ntH, wave, lane);    // (requested BEHIND the barrier: issuing it blocks a wave for a while)
  // ---- + bias, dropout, + x, LayerNorm1 -> s1, x1 (global), x1 planes ----
  if (lnw) ln_rows4<false>(stage, xr, cst, cst + KPD, cst + 2 * KPD, D, m0, M, wave, lane, a.p, inv_keep, nullptr, seed, a.site_ao, a.s1, LEAN ? nullptr : a.x1,
                    a.st1, Ah, Al, x1r);
  EFSTAMP(5);
  lds_barrier();
  EFSTAMP(6);
  if (a.xt_x1) export_tiles<RT>(Ah, Al, LDD, a.xt_x1, ntD, m0, M, wave, lane);

  // ---- linear1 -> + bias, ReLU, dropout -> h: the epilogue runs on the ACCUMULATORS (no fp32 stage, no barrier between product and
  // epilogue) -- a lane holds the column quad 16 j + 4 G .. + 3 of row 16 rt + i (mma), which is one dropout quad, one gate byte (LEAN)
  // or one 16-byte store of h, and one 8-byte store into each split plane.  Rounds 2-4: product -> stage -> barrier -> a flat loop of
  // 3.4 quads per thread -> barrier (9 k cycles for the loop alone; product + epilogue are 10.9 k now). ----
  Panel<KCH> p2;                                               // linear2 streams in under the (last) epilogue
  {
    const int G = lane >> 4, i16 = lane & 15, qpr = H >> 2;
    auto h_epi1 = [&](const f32x4& acc, int j, int rt) {       // one row tile of column tile j
      const int n = 16 * j + 4 * G;                            // < KPH (j < 18)
      const float4 bs = *reinterpret_cast<const float4*>(cst + 6 * KPD + n);
      const int rl = 16 * rt + i16, m = m0 + rl;
      float4 o = zero4;
      if (m < M && n < H) {
        o = make_float4(fmaxf(acc[0] + bs.x, 0.f), fmaxf(acc[1] + bs.y, 0.f), fmaxf(acc[2] + bs.z, 0.f), fmaxf(acc[3] + bs.w, 0.f));
        if (a.p > 0.f) {
          const float4 u = uniform4(seed, a.site_fh, ((uint64_t)m * H + n) >> 2);
          o.x = u.x >= a.p ? o.x * inv_keep : 0.f; o.y = u.y >= a.p ? o.y * inv_keep : 0.f;
          o.z = u.z >= a.p ? o.z * inv_keep : 0.f; o.w = u.w >= a.p ? o.w * inv_keep : 0.f;
        }
        if (LEAN)
          a.hgate[(long)m * qpr + (n >> 2)] = (uint8_t)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        else
          *reinterpret_cast<float4*>(a.h + (long)m * H + n) = o;
      }
      split_store4(Hh + rl * LDH + n, Hl + rl * LDH + n, o);
    };
    auto h_epilogue = [&](const f32x4 (&acc)[RT], int j) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) h_epi1(acc[rt], j, rt);
    };
    f32x4 acc[RT];
    mma<KCD, RT>(acc, Ah, Al, LDD, p1, lane, (int)ONE);           // column tile `wave` (every wave has one: nhid > 256)
    // nhid > 256: the column tiles beyond the 16th (P19: tile 16) are split into (tile, row tile) UNITS, one each for the LAST waves
    // (P19, tall blocks: waves 15, 14, 13 -- which have no D-wide tile, no LayerNorm pass and therefore nothing else to carry).  A
    // whole second tile on wave 0 (rounds 3-5) was 5 k cycles of product + epilogue that the other 15 waves spent at the barrier
    // (stamps in the step: 23.3 k .. 28.5 k).  The unit's panel goes into the registers the first product has just released, as ONE
    // batch of requests in front of the first tile's epilogue (which covers its way from L2); a wave without a unit requests the
    // first part of linear2's panel there instead.
    // (linear2's 72-register panel in two parts: steps 0 .. PS - 1 travel under the epilogue, the rest is requested behind it --
    // whole, it spilled 24-32 registers across the epilogue; the product consumes the steps in order, PS of them cover the rest)
    constexpr int PS = 5;
    // (`more` is a scalar condition: one of the two panels is live, not both)
    // (two straight-line paths, not `if (more) request A else request B` + a common epilogue: behind that join the register
    // allocator kept BOTH panels' registers apart -- 112 registers for one live panel)
    if (more) {
      load_panel<KCD>(p1, a.W1, ntH, j2, lane);
      __builtin_amdgcn_sched_barrier(0);
      h_epilogue(acc, wave);
      f32x4 acc1[1];
      mma<KCD, 1>(acc1, Ah + rt2 * 16 * LDD, Al + rt2 * 16 * LDD, LDD, p1, lane, (int)ONE);
      __builtin_amdgcn_sched_barrier(0);
      load_panel<KCH, 0, PS>(p2, a.W2, ntD, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      h_epi1(acc1[0], j2, rt2);
    } else {
      load_panel<KCH, 0, PS>(p2, a.W2, ntD, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      h_epilogue(acc, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_panel<KCH, PS, KCH>(p2, a.W2, ntD, wave, lane);
  }
  EFSTAMP(7);
  lds_barrier();
  EFSTAMP(8);
  if (a.xt_h) export_tiles<RT>(Hh, Hl, LDH, a.xt_h, ntH, m0, M, wave, lane);

  // ---- linear2; LayerNorm2's dropout decisions by the waves without a column tile ----
  if (wave < ntD) {
    f32x4 acc[RT];
    mma<KCH, RT>(acc, Hh, Hl, LDH, p2, lane, (int)ONE);
    to_stage<RT>(stage, acc, wave, lane);
  } else if (a.p > 0.f && wave >= HW0) {
    drop_masks<RT, NHW>(mk, seed, a.site_fo, D, m0, a.p, hidx);
  }
  EFSTAMP(9);
  lds_barrier();
  EFSTAMP(10);
  // ---- + bias, dropout, + x1, LayerNorm2 -> s2, y ----
  if (lnw) {                                                   // the residual x1: this lane's own fp32 copy (zero beyond D / M)
    const int rl = 4 * wave + (lane >> 4);
#pragma unroll
    for (int k = 0; k < LNQ; ++k) {
      const int c = 4 * ((lane & 15) + 16 * k);
      xr[k] = zero4;
      if (c < KPD) xr[k] = *reinterpret_cast<const float4*>(x1r + rl * KPD + c);
    }
  }
  if (lnw) ln_rows4<true>(stage, xr, cst + 3 * KPD, cst + 4 * KPD, cst + 5 * KPD, D, m0, M, wave, lane, a.p, inv_keep, mk, 0, 0, a.s2, a.y, a.st2,
                    nullptr, nullptr);
  EFSTAMP(11);
  if (a.stamps && tid == 0) { a.stamps[256 + 1024 + 2 * blockIdx.x] = wall_clock64(); a.stamps[256 + 1024 + 2 * blockIdx.x + 1] = clock64(); }
}

// ONE (round 6): RD_PREC_BF16's one-product form as an instantiation of its own.  As a runtime flag every product of the chain began
// with a branch -- a basic-block boundary the scheduler does not move the next phase's loads or the previous epilogue's stores across
// (in-step A/B against a build with the flag folded to 0: post_fwd 22.1 -> 21.3 us, pre_bwd 23.8 -> 22.6).
template <int DC, int HC, bool LEAN, bool ONE>
__global__ __launch_bounds__(EF_THR) void k_enc_post_fwd(PostFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
  // own code -> L2 (rd_common.h), ALL of it, per instantiation (raindrop_amd/build.py CODE_TOUCH checks each against the linked kernel): the
  // tall body -- the one the step runs -- is laid out LAST, and with one length below the smallest instantiation its LayerNorm2 sat in
  // the uncovered 3 KB: 9 k cycles instead of 5 k on a box whose instruction fetch does not look ahead
  RD_TOUCH_CODE(ONE ? (DC == 152 ? (LEAN ? RD_TL_EF_POST_P19L_B : RD_TL_EF_POST_P19_B) : DC == 160 ? (LEAN ? RD_TL_EF_POST_P12L_B : RD_TL_EF_POST_P12_B)
                                 : (LEAN ? RD_TL_EF_POST_RTL_B : RD_TL_EF_POST_RT_B))
                    : (DC == 152 ? (LEAN ? RD_TL_EF_POST_P19L : RD_TL_EF_POST_P19) : DC == 160 ? (LEAN ? RD_TL_EF_POST_P12L : RD_TL_EF_POST_P12)
                                 : (LEAN ? RD_TL_EF_POST_RTL : RD_TL_EF_POST_RT)));
  int M = a.M;
  if (a.mlive) M = min(M, __builtin_amdgcn_readfirstlane(*a.mlive));
  if (pick_rt(M, a.ncu) == 3) post_fwd_body<3, DC, HC, LEAN, ONE>(a, esm, M);
  else post_fwd_body<2, DC, HC, LEAN, ONE>(a, esm, M);
}

constexpr size_t post_fwd_lds(int rt) {
  return (size_t)2 * 16 * rt * LDD * 2 + (size_t)2 * 16 * rt * LDH * 2 + (size_t)16 * rt * STG * 4 + (size_t)(6 * KPD + KPH) * 4 +
         (size_t)16 * rt * KPD * 4 + (size_t)16 * rt * MKQ;
}
static_assert(post_fwd_lds(EF_RTMAX) <= 160 * 1024, "one workgroup's LDS");

// ------------------------------------------------------------------------------------------------
// backward chain: dy -> LayerNorm2' -> linear2' (gated by h > 0) -> linear1' + ds2 -> LayerNorm1' -> out_proj' -> d attn
// ------------------------------------------------------------------------------------------------
struct PreBwdArgs {
  const float* dy;                                  // [M, D] gradient of the layer output
  const float *s2, *st2, *g2;                       // LayerNorm2: saved pre-norm sum, (mean, rstd), gamma
  const float* h;                                   // [M, H] FFN hidden after ReLU and dropout (gate: h > 0)
  const uint8_t* hgate;                             // LEAN form: [M][H / 4] gate bytes written by the LEAN forward chain (h is null)
  const float *s1, *st1, *g1;                       // LayerNorm1
  const __bf16 *W2t, *W1t, *Wot;                    // operand tiles of linear2^T, linear1^T, out_proj^T
  float *ds2, *ds1, *da;                            // [M, D]: gradients of the LayerNorm2 / LayerNorm1 inputs (residual branches), of attn
  float *part2, *part1;                             // [grid][2D] dgamma | dbeta partials of LayerNorm2, LayerNorm1
  __bf16 *xt_df, *xt_du, *xt_dout;                  // row tiles for the weight gradients
  int M, D, H, ncu;
  float p; uint64_t seed; uint32_t site_fo, site_ao; const uint64_t* seed_cell;
  const int32_t* mlive;
  int one;
  unsigned long long* stamps;
};

// LayerNorm backward, one pass of four rows (the layout of ln_rows4).  dyq / sq: the rows' dy and saved pre-norm quads (zero beyond
// D or M); mean / rstd: the lane's row statistics.  Writes the unmasked gradient quads to ds_glob ([M][K]; null: not wanted) and / or to
// `keep` (fp32 [rows][KPD] in LDS, read back by the same lane; only the quads below K of live rows are written), the dropout-masked ones
// (what the next product consumes) as split planes, and this pass's dgamma | dbeta partials, summed over row pairs in registers
// (rowpair_sum), into lnred[2 wave + (g >> 1)].  Same arithmetic as k_rowgemm<.., LNB> (rd_rowgemm.hip) up to the order of the sums.
// PRE: the dropout keep bits come from mk (drop_masks); otherwise from Threefry calls here (seed, site).
template <bool PRE>
__device__ __forceinline__ void lnb_rows4(const float4 (&dyq)[LNQ], const float4 (&sq)[LNQ], float mean, float rstd, const float* gg, int K,
                                          int m0, int M, int wave, int lane, float p, float inv_keep, const uint8_t* mk, uint64_t seed,
                                          uint32_t site, float* ds_glob, float* keep, __bf16* Ph, __bf16* Pl, float* lnred) {
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int g = lane >> 4, i16 = lane & 15, rl = 4 * wave + g;
  const long row = m0 + rl;
  const bool rok = row < M;
  float4 dg[LNQ], xh[LNQ];
  float p1 = 0.f, p2 = 0.f;
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * (i16 + 16 * k);
    dg[k] = zero4; xh[k] = zero4;
    if (rok && c < K) {
      const float4 g4 = *reinterpret_cast<const float4*>(gg + c);
      xh[k] = make_float4((sq[k].x - mean) * rstd, (sq[k].y - mean) * rstd, (sq[k].z - mean) * rstd, (sq[k].w - mean) * rstd);
      dg[k] = make_float4(dyq[k].x * g4.x, dyq[k].y * g4.y, dyq[k].z * g4.z, dyq[k].w * g4.w);
    }
    p1 += (dg[k].x + dg[k].y) + (dg[k].z + dg[k].w);
    p2 += (dg[k].x * xh[k].x + dg[k].y * xh[k].y) + (dg[k].z * xh[k].z + dg[k].w * xh[k].w);
  }
  const float c1 = row16_allsum(p1) / K, c2 = row16_allsum(p2) / K;
  float* slot = lnred + (size_t)(2 * wave + (g >> 1)) * 2 * KPD;
#pragma unroll
  for (int k = 0; k < LNQ; ++k) {
    const int c = 4 * (i16 + 16 * k);
    float4 dr = zero4;
    if (rok && c < K) {
      const float4 v = make_float4(rstd * (dg[k].x - c1 - xh[k].x * c2), rstd * (dg[k].y - c1 - xh[k].y * c2),
                                   rstd * (dg[k].z - c1 - xh[k].z * c2), rstd * (dg[k].w - c1 - xh[k].w * c2));
      if (ds_glob) *reinterpret_cast<float4*>(ds_glob + row * K + c) = v;
      if (keep) *reinterpret_cast<float4*>(keep + rl * KPD + c) = v;
      dr = v;
      if (p > 0.f) {
        if constexpr (PRE) {                          // keep bits made by the idle waves of the phase before (drop_masks)
          const unsigned kb = mk[rl * MKQ + (c >> 2)];
          dr.x *= (kb & 1) ? inv_keep : 0.f; dr.y *= (kb & 2) ? inv_keep : 0.f;
          dr.z *= (kb & 4) ? inv_keep : 0.f; dr.w *= (kb & 8) ? inv_keep : 0.f;
        } else {
          const float4 u = uniform4(seed, site, ((uint64_t)row * K + c) >> 2);
          dr.x *= u.x >= p ? inv_keep : 0.f; dr.y *= u.y >= p ? inv_keep : 0.f;
          dr.z *= u.z >= p ? inv_keep : 0.f; dr.w *= u.w >= p ? inv_keep : 0.f;
        }
      }
    }
    if (c < KPD) split_store4(Ph + rl * LDD + c, Pl + rl * LDD + c, dr);
    // dgamma | dbeta of this row (dyq is zero beyond D / M), added to the neighbouring row's in registers: all 64 lanes take part
    const float4 ag = make_float4(rowpair_sum(dyq[k].x * xh[k].x), rowpair_sum(dyq[k].y * xh[k].y), rowpair_sum(dyq[k].z * xh[k].z),
                                  rowpair_sum(dyq[k].w * xh[k].w));
    const float4 ab = make_float4(rowpair_sum(dyq[k].x), rowpair_sum(dyq[k].y), rowpair_sum(dyq[k].z), rowpair_sum(dyq[k].w));
    if ((g & 1) == 0 && c < K) {
      *reinterpret_cast<float4*>(slot + c) = ag;
      *reinterpret_cast<float4*>(slot + KPD + c) = ab;
    }
  }
}

template <int RT, int DC, int HC, bool LEAN, bool ONE>
__device__ __forceinline__ void pre_bwd_body(const PreBwdArgs& a, unsigned char* esm, int M) {
  constexpr int ROWS = 16 * RT;
  __bf16* Ah = reinterpret_cast<__bf16*>(esm);                 // [ROWS][LDD]: df, then dout
  __bf16* Al = Ah + ROWS * LDD;
  __bf16* Hh = Al + ROWS * LDD;                                // [ROWS][LDH]: du
  __bf16* Hl = Hh + ROWS * LDH;
  float* stage = reinterpret_cast<float*>(Hl + ROWS * LDH);    // [ROWS][STG]
  float* cst = stage + ROWS * STG;                             // [g2 | g1] (KPD each, zero padded)
  // ds2 (gradient of LayerNorm2's input: the residual branch around the FFN) as fp32 [ROWS][KPD]: written by LayerNorm2' and read back
  // by the same lane in front of LayerNorm1' -- it never leaves the CU (rounds 3-4: a 5-MB round trip through memory per layer)
  float* dsr = cst + 2 * KPD;
  uint8_t* mk = reinterpret_cast<uint8_t*>(dsr + ROWS * KPD);  // dropout keep bits of LayerNorm1' (drop_masks: by idle waves)
  // [2 NPASS row pairs][2 KPD] dgamma | dbeta partials of one LayerNorm.  LayerNorm2's live in the (then idle) stage; LayerNorm1's alias
  // the du planes, which are dead by then (both uses are closed by a barrier before / after the memory is reused)
  float* lnred2 = stage;
  float* lnred1 = reinterpret_cast<float*>(Hh);
  static_assert((size_t)2 * 4 * RT * 2 * KPD * 4 <= (size_t)2 * ROWS * LDH * 2, "lnred must fit inside the du planes");
  static_assert((size_t)2 * 4 * RT * 2 * KPD * 4 <= (size_t)ROWS * STG * 4, "lnred must fit inside the stage");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: `if (wave < ..)` is a branch, not a mask
  const int m0 = blockIdx.x * ROWS;
  const int D = DC ? DC : a.D, H = HC ? HC : a.H;
  const int ntD = (D + 15) >> 4, ntH = (H + 15) >> 4;
  if (m0 >= M) {                                               // no live row: the partial sums of this block are zero
    for (int i = tid; i < 2 * D; i += EF_THR) { a.part2[(long)blockIdx.x * 2 * D + i] = 0.f; a.part1[(long)blockIdx.x * 2 * D + i] = 0.f; }
    zero_dead_groups<RT>(a.xt_df, ntD, m0, M, tid); zero_dead_groups<RT>(a.xt_du, ntH, m0, M, tid); zero_dead_groups<RT>(a.xt_dout, ntD, m0, M, tid);
    return;
  }
  uint64_t seed = a.seed;
  const float inv_keep = 1.0f / (1.0f - a.p);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  EFSTAMP(0);
  // ---- rows of dy, s2 and their statistics, in the LayerNorm pass layout (four rows per wave, waves 0 .. NPASS - 1) ----
  constexpr int NPASS = 4 * RT;
  const bool lnw = wave < NPASS;
  float4 dyq[LNQ], sq[LNQ]; float mean_l = 0.f, rstd_l = 0.f;
  auto load_stats = [&](const float* st) {
    const long row = m0 + 4 * wave + (lane >> 4);
    mean_l = 0.f; rstd_l = 0.f;
    if (lnw && row < M) { mean_l = st[2 * row]; rstd_l = st[2 * row + 1]; }
  };
#pragma unroll
  for (int k = 0; k < LNQ; ++k) { dyq[k] = zero4; sq[k] = zero4; }
  if (lnw) { ln_load3(sq, a.s2, D, m0, M, wave, lane); ln_load3(dyq, a.dy, D, m0, M, wave, lane); }
  load_stats(a.st2);
  Panel<KCD> pw;
  load_panel<KCD, 0, KCD, true>(pw, a.W2t, ntH, wave, lane);
  // the two gamma vectors: requested BEHIND the rows (rounds 3-4: in front of them and waited for -- a second cold round trip in
  // front of the LayerNorm waves' row loads), unconditional from clamped addresses, each from its own kernel argument
  float gv2 = a.g2[min(tid, D - 1)], gv1 = a.g1[min(tid, D - 1)];
  __builtin_amdgcn_sched_barrier(0);
  if (a.seed_cell) seed += load_uniform_u64(a.seed_cell);
  if (tid < KPD) { cst[tid] = tid < D ? gv2 : 0.f; cst[KPD + tid] = tid < D ? gv1 : 0.f; }
  lds_barrier();                                               // cst visible
  EFSTAMP(1);
  if (lnw) lnb_rows4<false>(dyq, sq, mean_l, rstd_l, cst, D, m0, M, wave, lane, a.p, inv_keep, nullptr, seed, a.site_fo, nullptr, dsr, Ah, Al, lnred2);
  zero_pad_columns<RT>(Hh, Hl, ntH, tid);                      // columns 16 ntH .. KPH of the du planes (linear1' reduces over KPH)
  EFSTAMP(2);
  lds_barrier();
  EFSTAMP(3);
  for (int i = tid; i < 2 * D; i += EF_THR) {                  // this block's dgamma | dbeta partial: the 16 waves in fixed order
    const int col = i < D ? i : KPD + (i - D);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 2 * NPASS; ++w) v += lnred2[w * 2 * KPD + col];
    a.part2[(long)blockIdx.x * 2 * D + i] = v;
  }
  // the FFN hidden's gate (h > 0) for this wave's column tile, in the accumulator layout of the product that consumes it (lane: row
  // 16 rt + i, column quad 4 j + G): requested here (behind LayerNorm2', whose registers are free again), consumed behind the
  // product; a second-round unit's gate is requested behind the first tile's epilogue, into the same registers.  UNCONDITIONAL from
  // clamped addresses (rows >= M have a zero gradient): a conditional load is a phi of {0, value} and the compiler waited for each
  // one right behind its request
  const int G = lane >> 4, i16 = lane & 15;
  float4 hv[LEAN ? 1 : RT]; uint8_t hb[LEAN ? RT : 1];
  auto load_gate1 = [&](int j, int rt, int slot) {
    const long row = min(m0 + 16 * rt + i16, M - 1);
    const int q = min(4 * j + G, (H >> 2) - 1);
    if (LEAN) hb[slot] = a.hgate[row * (H >> 2) + q];
    else hv[slot] = *reinterpret_cast<const float4*>(a.h + row * H + 4 * q);
  };
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) load_gate1(wave, rt, rt);
  if (a.xt_df) export_tiles<RT>(Ah, Al, LDD, a.xt_df, ntD, m0, M, wave, lane);
  // ---- du = (df W2) gated by h > 0, * keep: the gate runs on the ACCUMULATORS (a lane holds a column quad of a row, see mma) and
  // writes the split planes directly -- no fp32 stage, no barrier between product and epilogue (see the forward chain's h_epilogue) ----
  Panel<KCH> p1;                                               // linear1^T streams in under the (last) epilogue
  // its 72 registers in parts (see the forward chain): steps [0, PS) under the epilogue, [PS, PS2) behind it, and [PS2, KCH) INSIDE the
  // product, once two steps are consumed (the product runs at the register limit: 72 + 12 accumulator + 24 fragment registers in the
  // tall variant; the low one carries LayerNorm1's saved rows through it): five steps of three waves' MFMAs (~2 k cycles) cover their
  // way from L2
  constexpr int PS = (RT >= 3 && !LEAN) ? 3 : 5, PS2 = 7;    // (fp32 gates take 9 registers more than gate bytes)
  {
    const float ks = a.p > 0.f ? inv_keep : 1.0f;
    auto du_epi1 = [&](const f32x4& acc, int j, int rt, int slot) {   // one row tile of column tile j, gate in hb / hv[slot]
      const int n = 16 * j + 4 * G;                            // < KPH
      const int rl = 16 * rt + i16;
      float4 o = zero4;
      if (n < H) {
        if (LEAN) {
          const uint8_t b = hb[slot];
          o = make_float4((b & 1) ? acc[0] * ks : 0.f, (b & 2) ? acc[1] * ks : 0.f, (b & 4) ? acc[2] * ks : 0.f, (b & 8) ? acc[3] * ks : 0.f);
        } else {
          const float4 h4 = hv[slot];
          o = make_float4(h4.x > 0.f ? acc[0] * ks : 0.f, h4.y > 0.f ? acc[1] * ks : 0.f, h4.z > 0.f ? acc[2] * ks : 0.f,
                          h4.w > 0.f ? acc[3] * ks : 0.f);
        }
      }
      split_store4(Hh + rl * LDH + n, Hl + rl * LDH + n, o);
    };
    auto du_epilogue = [&](const f32x4 (&acc)[RT], int j) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) du_epi1(acc[rt], j, rt, rt);
    };
    f32x4 acc[RT];
    mma<KCD, RT>(acc, Ah, Al, LDD, pw, lane, (int)ONE);           // column tile `wave` (every wave has one: nhid > 256)
    const int unit = EF_WV - 1 - wave;                         // scalar; second-round (tile, row tile) units: see the forward chain
    const bool more = unit < (ntH - EF_WV) * RT;
    if (more) {                                                // two straight-line paths: see the forward chain
      const int j2 = EF_WV + unit / RT, rt2 = unit - (unit / RT) * RT;
      load_panel<KCD>(pw, a.W2t, ntH, j2, lane);
      __builtin_amdgcn_sched_barrier(0);
      du_epilogue(acc, wave);
      __builtin_amdgcn_sched_barrier(0);
      load_gate1(j2, rt2, 0);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc1[1];
      mma<KCD, 1>(acc1, Ah + rt2 * 16 * LDD, Al + rt2 * 16 * LDD, LDD, pw, lane, (int)ONE);
      __builtin_amdgcn_sched_barrier(0);
      load_panel<KCH, 0, PS>(p1, a.W1t, ntD, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      du_epi1(acc1[0], j2, rt2, 0);
    } else {
      load_panel<KCH, 0, PS>(p1, a.W1t, ntD, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
      du_epilogue(acc, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
    load_panel<KCH, PS, PS2>(p1, a.W1t, ntD, wave, lane);      // (tall variant: the last two steps inside the product, below)
  }
  // LayerNorm1's saved rows: requested here (the tall variant: inside the next product, once four reduction steps of its 72-register
  // panel are consumed -- carried through the whole product they spilled), consumed after the product
  auto load_ln1 = [&]() {
    load_stats(a.st1);
#pragma unroll
    for (int k = 0; k < LNQ; ++k) sq[k] = zero4;
    if (lnw) ln_load3(sq, a.s1, D, m0, M, wave, lane);
  };
  if constexpr (RT < 3) load_ln1();
  EFSTAMP(4);
  lds_barrier();                                               // du planes complete; everybody is done with the stage (lnred2)
  EFSTAMP(5);
  if (a.xt_du) export_tiles<RT>(Hh, Hl, LDH, a.xt_du, ntH, m0, M, wave, lane);
  // ---- dx1 = du W1 + ds2 ----
  if (wave < ntD) {
    f32x4 acc[RT];
    auto rest = [&]() { load_panel<KCH, PS2, KCH>(p1, a.W1t, ntD, wave, lane); };
    if constexpr (RT >= 3) mma<KCH, RT, 1, decltype(rest), 3, decltype(load_ln1)>(acc, Hh, Hl, LDH, p1, lane, (int)ONE, rest, load_ln1);
    else mma<KCH, RT, 1, decltype(rest)>(acc, Hh, Hl, LDH, p1, lane, (int)ONE, rest);
    to_stage<RT>(stage, acc, wave, lane);
  } else {                                                     // no column tile: LayerNorm1's dropout decisions meanwhile
    if constexpr (RT >= 3) load_ln1();
    if (a.p > 0.f && wave >= 10) drop_masks<RT, EF_WV - 10>(mk, seed, a.site_ao, D, m0, a.p, (wave - 10) * 64 + lane);   // waves 10 .. 15: D <= 160
  }
  EFSTAMP(6);
  lds_barrier();                                               // stage complete; the du planes are dead (lnred1 may be written)
  EFSTAMP(7);
  Panel<KCD> po;
  load_panel<KCD>(po, a.Wot, ntD, wave, lane);
  if (lnw) {
    const int rl = 4 * wave + (lane >> 4);
#pragma unroll
    for (int k = 0; k < LNQ; ++k) {
      const int cq = 4 * ((lane & 15) + 16 * k);
      dyq[k] = zero4;
      if (cq < D && m0 + rl < M) {
        const float4 t = *reinterpret_cast<const float4*>(stage + rl * STG + cq);
        const float4 r4 = *reinterpret_cast<const float4*>(dsr + rl * KPD + cq);   // ds2: this lane's own copy
        dyq[k] = make_float4(t.x + r4.x, t.y + r4.y, t.z + r4.z, t.w + r4.w);
      }
    }
    lnb_rows4<true>(dyq, sq, mean_l, rstd_l, cst + KPD, D, m0, M, wave, lane, a.p, inv_keep, mk, 0, 0, a.ds1, nullptr, Ah, Al, lnred1);
  }
  EFSTAMP(8);
  lds_barrier();
  EFSTAMP(9);
  for (int i = tid; i < 2 * D; i += EF_THR) {
    const int col = i < D ? i : KPD + (i - D);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 2 * NPASS; ++w) v += lnred1[w * 2 * KPD + col];
    a.part1[(long)blockIdx.x * 2 * D + i] = v;
  }
  if (a.xt_dout) export_tiles<RT>(Ah, Al, LDD, a.xt_dout, ntD, m0, M, wave, lane);
  // ---- d attn = dout Wo: straight from the accumulators (a lane holds four consecutive columns of a row: one 16-byte store; the 16
  // lanes of a group write 64 contiguous bytes of 16 rows, the neighbouring column tile's wave the other half of each line) ----
  if (wave < ntD) {
    f32x4 acc[RT];
    mma<KCD, RT>(acc, Ah, Al, LDD, po, lane, (int)ONE);
    const int c = 16 * wave + 4 * G;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const long m = m0 + 16 * rt + i16;
      if (m < M && c < D) *reinterpret_cast<f32x4*>(a.da + m * D + c) = acc[rt];
    }
  }
  EFSTAMP(10);
}

// Workgroups >= nmain are RIDERS (rd_trailing.h): they run a parked trailing launch -- the head's weight-gradient tiles, the
// previous layer's slice reduce -- on the CUs this chain leaves idle (178-266 workgroups of one per CU on 256 CUs).
template <int DC, int HC, bool LEAN, bool ONE>
__global__ __launch_bounds__(EF_THR) void k_enc_pre_bwd(PreBwdArgs a, RiderArgs rider, int nmain) {
  extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
  // own code -> L2, the riders' bodies included, all of it, per instantiation (see k_enc_post_fwd)
  RD_TOUCH_CODE(ONE ? (DC == 152 ? (LEAN ? RD_TL_EF_PRE_P19L_B : RD_TL_EF_PRE_P19_B) : DC == 160 ? (LEAN ? RD_TL_EF_PRE_P12L_B : RD_TL_EF_PRE_P12_B)
                                 : (LEAN ? RD_TL_EF_PRE_RTL_B : RD_TL_EF_PRE_RT_B))
                    : (DC == 152 ? (LEAN ? RD_TL_EF_PRE_P19L : RD_TL_EF_PRE_P19) : DC == 160 ? (LEAN ? RD_TL_EF_PRE_P12L : RD_TL_EF_PRE_P12)
                                 : (LEAN ? RD_TL_EF_PRE_RTL : RD_TL_EF_PRE_RT)));
  if ((int)blockIdx.x >= nmain) { rider_body(rider, (int)blockIdx.x - nmain, esm); return; }
  int M = a.M;
  if (a.mlive) M = min(M, __builtin_amdgcn_readfirstlane(*a.mlive));
  if (pick_rt(M, a.ncu) == 3) pre_bwd_body<3, DC, HC, LEAN, ONE>(a, esm, M);
  else pre_bwd_body<2, DC, HC, LEAN, ONE>(a, esm, M);
}

// Widths compiled in for the two datasets that fit these kernels: 1 = P19 (152, 272), 2 = P12 (160, 288); 0 = runtime widths.
// RD_ENC_SPECIALIZE=0: always the runtime-width instantiation (A/B and the parity test of the two)
static int ef_specialize(int D, int H) {
  const char* e = getenv("RD_ENC_SPECIALIZE");
  if (e && atoi(e) == 0) return 0;
  return (D == 152 && H == 272) ? 1 : ((D == 160 && H == 288) ? 2 : 0);
}

constexpr size_t pre_bwd_lds(int rt) {
  return (size_t)2 * 16 * rt * LDD * 2 + (size_t)2 * 16 * rt * LDH * 2 + (size_t)16 * rt * STG * 4 + (size_t)2 * KPD * 4 +
         (size_t)16 * rt * KPD * 4 + (size_t)16 * rt * MKQ;
}
static_assert(pre_bwd_lds(EF_RTMAX) <= 160 * 1024, "one workgroup's LDS");

int device_cus() {
  static const int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
    return v;
  }();
  return n;
}

}  // namespace

extern "C" void rd_debug_set_encfuse_stamps(void* p) { g_ef_stamps = (unsigned long long*)p; }   // not part of the ABI

bool encfuse_ok(int D, int H) {
  const char* e = getenv("RD_ENC_FUSE");               // read per call (tests compare the fused and the unfused path in one process)
  const bool enabled = !(e && atoi(e) == 0);
  // widths: the two compiled-in pairs (P19, P12).  The runtime-width instantiation of the backward chain spills 84 bytes per lane
  // (DESIGN rule 11): it runs only when asked for explicitly (RD_ENC_SPECIALIZE=0: the A/B and the parity test of the two);
  // any other width in the envelope takes the separate row-block launches.
  const char* sp = getenv("RD_ENC_SPECIALIZE");
  const bool runtime_widths = sp && atoi(sp) == 0;
  return enabled && precision() != RD_PREC_FP32 && (D + 31) / 32 == KCD && (H + 31) / 32 == KCH && (D % 4) == 0 && (H % 4) == 0 &&
         (runtime_widths || ef_specialize(D, H) != 0);
}

// RD_ENC_FUSE_TALL=0 pins the block height to 32 rows (A/B of the device-side choice)
static int ef_ncu() {
  const char* e = getenv("RD_ENC_FUSE_TALL");           // read per call
  const bool tall = !(e && atoi(e) == 0);
  return tall ? device_cus() : (1 << 20);
}

int launch_enc_post_fwd(long M, int D, int H, const float* attn, const float* x, const void* Wo, const void* W1, const void* W2,
                        const float* bo, const float* b1, const float* b2, const float* g1, const float* be1, const float* g2,
                        const float* be2, float* s1, float* x1, float* st1, float* h, float* s2, float* y, float* st2,
                        void* xt_attn, void* xt_x1, void* xt_h, float p, uint64_t seed, uint32_t site_ao, uint32_t site_fh,
                        uint32_t site_fo, const int32_t* mlive, void* hgate, hipStream_t st) {
  PostFwdArgs a{};
  a.attn = attn; a.x = x; a.Wo = (const __bf16*)Wo; a.W1 = (const __bf16*)W1; a.W2 = (const __bf16*)W2;
  a.bo = bo; a.b1 = b1; a.b2 = b2; a.g1 = g1; a.be1 = be1; a.g2 = g2; a.be2 = be2;
  a.s1 = s1; a.x1 = x1; a.st1 = st1; a.h = h; a.s2 = s2; a.y = y; a.st2 = st2;
  a.xt_attn = (__bf16*)xt_attn; a.xt_x1 = (__bf16*)xt_x1; a.xt_h = (__bf16*)xt_h;
  a.M = (int)M; a.D = D; a.H = H; a.ncu = ef_ncu(); a.p = p; a.seed = seed; a.site_ao = site_ao; a.site_fh = site_fh; a.site_fo = site_fo;
  a.seed_cell = seed_cell(); a.mlive = mlive; a.one = precision() == RD_PREC_BF16; a.stamps = g_ef_stamps;
  constexpr size_t lds = post_fwd_lds(EF_RTMAX);
  const int spec = ef_specialize(D, H);
  a.hgate = (uint8_t*)hgate;
#define RD_POST_FWD1(DCV, HCV, LEANV, ONEV)                                                                          \
  do { RD_LDS_ATTR((k_enc_post_fwd<DCV, HCV, LEANV, ONEV>), lds);                                                    \
       hipLaunchKernelGGL((k_enc_post_fwd<DCV, HCV, LEANV, ONEV>), dim3(cdiv((int)M, 32)), dim3(EF_THR), lds, st, a); } while (0)
#define RD_POST_FWD(DCV, HCV)                                                                                        \
  do {                                                                                                               \
    if (hgate) { if (a.one) RD_POST_FWD1(DCV, HCV, true, true); else RD_POST_FWD1(DCV, HCV, true, false); }          \
    else { if (a.one) RD_POST_FWD1(DCV, HCV, false, true); else RD_POST_FWD1(DCV, HCV, false, false); }              \
  } while (0)
  if (spec == 1) RD_POST_FWD(152, 272);
  else if (spec == 2) RD_POST_FWD(160, 288);
  else RD_POST_FWD(0, 0);
#undef RD_POST_FWD
#undef RD_POST_FWD1
  return check_launch("k_enc_post_fwd");
}

int encfuse_part_rows(long M) { return (int)((M + 31) / 32); }

int launch_enc_pre_bwd(long M, int D, int H, const float* dy, const float* s2, const float* st2, const float* g2, const float* h,
                       const float* s1, const float* st1, const float* g1, const void* W2t, const void* W1t, const void* Wot,
                       float* ds2, float* ds1, float* da, float* part2, float* part1, void* xt_df, void* xt_du, void* xt_dout, float p,
                       uint64_t seed, uint32_t site_fo, uint32_t site_ao, const int32_t* mlive, const void* hgate, hipStream_t st) {
  PreBwdArgs a{};
  a.dy = dy; a.s2 = s2; a.st2 = st2; a.g2 = g2; a.h = h; a.s1 = s1; a.st1 = st1; a.g1 = g1;
  a.W2t = (const __bf16*)W2t; a.W1t = (const __bf16*)W1t; a.Wot = (const __bf16*)Wot;
  a.ds2 = ds2; a.ds1 = ds1; a.da = da; a.part2 = part2; a.part1 = part1;
  a.xt_df = (__bf16*)xt_df; a.xt_du = (__bf16*)xt_du; a.xt_dout = (__bf16*)xt_dout;
  a.M = (int)M; a.D = D; a.H = H; a.ncu = ef_ncu(); a.p = p; a.seed = seed; a.site_fo = site_fo; a.site_ao = site_ao;
  a.seed_cell = seed_cell(); a.mlive = mlive; a.one = precision() == RD_PREC_BF16;
  a.stamps = g_ef_stamps ? g_ef_stamps + 4096 : nullptr;   // debug: the backward chain's stamps behind the forward chain's (8192 words)
  constexpr size_t lds = pre_bwd_lds(EF_RTMAX);
  static_assert(lds >= 4 * HW_GROUP_LDS, "the riders' LDS must fit the chain's");   // (otherwise: launch with the larger of the two)
  const int spec = ef_specialize(D, H);
  a.hgate = (const uint8_t*)hgate;
  const RiderArgs rider = trailing_take();             // a parked trailing launch (or kind 0) rides in this one
  const int nmain = cdiv((int)M, 32), grid = nmain + (rider.kind != RIDER_NONE ? rider.nblocks : 0);
#define RD_PRE_BWD1(DCV, HCV, LEANV, ONEV)                                                                          \
  do { RD_LDS_ATTR((k_enc_pre_bwd<DCV, HCV, LEANV, ONEV>), lds);                                                    \
       hipLaunchKernelGGL((k_enc_pre_bwd<DCV, HCV, LEANV, ONEV>), dim3(grid), dim3(EF_THR), lds, st, a, rider, nmain); } while (0)
#define RD_PRE_BWD(DCV, HCV)                                                                                        \
  do {                                                                                                              \
    if (hgate) { if (a.one) RD_PRE_BWD1(DCV, HCV, true, true); else RD_PRE_BWD1(DCV, HCV, true, false); }           \
    else { if (a.one) RD_PRE_BWD1(DCV, HCV, false, true); else RD_PRE_BWD1(DCV, HCV, false, false); }               \
  } while (0)
  if (spec == 1) RD_PRE_BWD(152, 272);
  else if (spec == 2) RD_PRE_BWD(160, 288);
  else RD_PRE_BWD(0, 0);
#undef RD_PRE_BWD
#undef RD_PRE_BWD1
  return check_launch("k_enc_pre_bwd");
}

}  // namespace rd
