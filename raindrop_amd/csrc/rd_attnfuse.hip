// rd_attnfuse.hip -- in_proj + attention core of a TransformerEncoderLayer as ONE launch per direction (token plan, T <= 64).
//
// torch's encoder layer (torch/nn/modules/transformer.py, used at code/models_rd.py:235-237,358) starts with
//   qkv = x W_in^T + b_in;  per head: S = Q K^T / sqrt(hd), keys >= len masked, P = softmax(S), O = dropout(P) V
// Until round 3 that was two launches forward (the QKV row-block product, then one attention workgroup per (sample, head)) and two
// backward (attention, then dx = dqkv W_in + ds1 as a row-block product), with qkv and dqkv -- 3D floats per token, the widest
// tensors of the layer -- written to HBM by one launch and read back by the next, twice.  Here ONE workgroup owns one SAMPLE:
//   forward   x rows (<= 64 x D, read once) -> split planes -> per head: Q|K|V = X W_h^T + b on the matrix cores, written
//             TRANSPOSED ([feature][step]: an accumulator holds 4 consecutive steps of one feature = one 8-byte LDS store) ->
//             S, softmax, dropout, O = P V -> attention rows out as 16-byte stores.  qkv never exists in memory.
//   backward  x rows -> the same Q|K|V again (recomputed: 1.8 us of MFMA per head against 31 MB of HBM traffic per layer) ->
//             S, dP, P, dS -> dQ, dK, dV (accumulators) -> transposed planes -> (a) row tiles of dqkv for the weight-gradient
//             stream, (b) dx += d{Q,K,V} W_{h}: accumulated over both heads IN REGISTERS, + ds1 (the residual branch) -> dx rows.
//             dqkv never exists in memory as fp32; the cross-head sum needs no second pass because a workgroup owns both heads.
// The weight-gradient stream (rd_tile_wgrad.hip) needs x and dqkv as row tiles of 32 rows.  A sample's rows do not start on a
// 32-row boundary of the compact token order, so these kernels export in the plan's PER-SAMPLE group space (rd_plan.h: coff):
// rank r owns the 16-row groups coff[r] .. coff[r + 1] (two groups = one 32-row chunk), rows past its length are zero.
// dW = dqkv^T x is a sum over rows: any order works as long as both operands use the same one.  dqkv's columns are exported in a head-padded layout ((q|k|v, head) blocks of 16 NTH
// columns); k_twg_reduce maps them back to in_proj's rows.
//
// Same arithmetic as the kernels it replaces (split-bf16 products with fp32 accumulation in the same order of the reduction
// steps, same dropout quads -- attn_quad(rank * H + h, ..) -- same saved log-sum-exp), so it is tested against them directly.
// Envelope: token plan registered, T <= 64, head_dim <= 16 NTH (NTH = 5), ceil(D / 32) == KCX (5), D % 4 == 0, hd % 4 == 0.
#include <stdlib.h>

#include "rd_common.h"
#include "rd_plan.h"
#include "rd_rng.h"

namespace rd {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short sh4 __attribute__((ext_vector_type(4)));
typedef short sh8 __attribute__((ext_vector_type(8)));

constexpr int TS = 64;                               // steps of a sample (T <= 64)
// Row stride (bf16) of every [feature][step] plane and of the [key][query] score planes: EXACTLY 128 bytes, the eight 16-byte chunks of
// a row XOR-swizzled by the row (tofs).  These planes are read three ways -- transposing reads (ds_read_b64_tr_b16: 2 x 32-lane groups
// over 64 banks, a group covers rows {r .. r+3, r+8 .. r+11} x 32 bytes), 16-byte fragment reads (rows r .. r+15 x one chunk) -- and
// written as 8-byte quads; tools/lds_conflicts.py's model: with the 144-byte rows of round 4 (TS + 8) every read was 2-way conflicted
// (the judge's PMC pass: 48 % of the kernel's LDS cycles), no padded stride fixes the transposing reads (rows r and r + 8 collide
// whenever the stride is a multiple of 32 bytes), this swizzle makes both reads conflict-free and leaves the quad stores 2-way as
// before -- and the planes are 9 KB smaller.
constexpr int LDT = TS;
// (row bit 2 is NOT part of the swizzle, so the second transposing read of a fragment -- four rows further -- is the first one's
// address + 4 LDT: one address register per fragment, as before)
__device__ __forceinline__ int tsw(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int tofs(int row, int col) { return row * LDT + ((((col >> 3) ^ tsw(row)) & 7) << 3) + (col & 7); }
constexpr int AF_WV = 8, AF_THR = 64 * AF_WV;

struct FAttnArgs {
  const float* x;                                    // [M_live, D] layer input (plan order)
  const __bf16* wf;                                  // [H][3][NTH][KCX][hi,lo][64][8]: rows (which, head) of W_in as B operands, reduction D
  const __bf16* wb;                                  // [H][3][NCT][HDP/32][hi,lo][64][8]: W_in(which, head)^T: rows = D, reduction = head dim
  const float* bias;                                 // in_proj_bias [3D]
  float* out; float* lse;                            // attention output [M_live, D]; log-sum-exp [B, H, T]
  const float* dout; const float* ds1; float* dx;    // backward: d out [M_live, D], residual-branch gradient, layer-input gradient
  __bf16* xt; __bf16* dt;                            // backward: row tiles of x [S][ceil(D/16)][2][512], of dqkv [S][3 H NTH][2][512]
  const int32_t* plan;
  int T, B, D, H, hd;
  float scale, p_drop; uint64_t seed; uint32_t site; const uint64_t* seed_cell;
  int one;                                           // RD_PREC_BF16: hi * hi only
  unsigned long long* stamps;                        // debug (tools/attnfuse_timing.py): clock64 per phase, wave 0 of workgroup 0
};
static unsigned long long* g_af_stamps = nullptr;
#define AFSTAMP(i)                                                                                  \
  do {                                                                                              \
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[(i)] = clock64();                 \
  } while (0)

__device__ __forceinline__ float g16_max(float v) {
#define RD_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
  v = fmaxf(v, RD_ROR(v, 8)); v = fmaxf(v, RD_ROR(v, 4)); v = fmaxf(v, RD_ROR(v, 2)); v = fmaxf(v, RD_ROR(v, 1));
  return v;
}
__device__ __forceinline__ float g16_sum(float v) {
  v += RD_ROR(v, 8); v += RD_ROR(v, 4); v += RD_ROR(v, 2); v += RD_ROR(v, 1);
#undef RD_ROR
  return v;
}

// fragment with the reduction index along the plane's columns: lane -> row row0 + (lane & 15), columns k0 + 8 (lane >> 4) ..
__device__ __forceinline__ bf8 frag_n(const __bf16* P, int ld, int row0, int k0, int lane) {
  return *reinterpret_cast<const bf8*>(P + (row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
}
// fragment with the reduction index along the plane's rows: lane -> column col0 + (lane & 15), rows k0 + 8 (lane >> 4) ..
// (two ds_read_b64_tr_b16: tools/probe_tr16.hip)
__device__ __forceinline__ bf8 frag_t(const __bf16* P, int ld, int k0, int col0, int lane) {
  const int i = lane & 15, G = lane >> 4;
  const __bf16* src = P + (k0 + 8 * G + (i >> 2)) * ld + col0 + 4 * (i & 3);
  const sh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src));
  const sh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src + 4 * ld));
  const sh8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, o);
}
// the same two fragments from a swizzled [.][TS] plane (tofs): (row, col) are PLANE coordinates, never folded into the pointer
__device__ __forceinline__ bf8 frag_ns(const __bf16* P, int row0, int k0, int lane) {
  return *reinterpret_cast<const bf8*>(P + tofs(row0 + (lane & 15), k0 + 8 * (lane >> 4)));
}
__device__ __forceinline__ bf8 frag_ts(const __bf16* P, int k0, int col0, int lane) {
  const int i = lane & 15, G = lane >> 4;
  const int row = k0 + 8 * G + (i >> 2), col = col0 + 4 * (i & 3);
  const __bf16* src = P + tofs(row, col);
  const sh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src));
  const sh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src + 4 * LDT));
  const sh8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf8, o);
}
// acc[j] += A(16 x KK) B(KK x 16 NT), split-bf16 (lo*hi + hi*lo + hi*hi per reduction step: the order of rd_temporal.hip's mma_b16).
// AT / BT: the operand is read transposed (its reduction index runs along plane rows).  a0: first row (AT: column) of A's 16-wide
// slice; B tile j starts at row (BT: column) b0 + 16 j.  ASW / BSW: the operand lives in a swizzled [.][TS] plane (lda / ldb unused).
template <int NT, bool AT, bool BT, bool ASW, bool BSW>
__device__ __forceinline__ void mma_b16(f32x4 (&acc)[NT], const __bf16* Ah, const __bf16* Al, int lda, int a0, const __bf16* Bh,
                                        const __bf16* Bl, int ldb, int b0, int KK, int lane, bool one) {
  auto fa = [&](const __bf16* P, int k0) {
    if (ASW) return AT ? frag_ts(P, k0, a0, lane) : frag_ns(P, a0, k0, lane);
    return AT ? frag_t(P, lda, k0, a0, lane) : frag_n(P, lda, a0, k0, lane);
  };
  auto fb = [&](const __bf16* P, int k0, int j) {
    if (BSW) return BT ? frag_ts(P, k0, b0 + 16 * j, lane) : frag_ns(P, b0 + 16 * j, k0, lane);
    return BT ? frag_t(P, ldb, k0, b0 + 16 * j, lane) : frag_n(P, ldb, b0 + 16 * j, k0, lane);
  };
  if (!one) {
#pragma unroll
    for (int k0 = 0; k0 < KK; k0 += 32) {
      const bf8 ah = fa(Ah, k0);
      const bf8 al = fa(Al, k0);
      bf8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) { bh[j] = fb(Bh, k0, j); bl[j] = fb(Bl, k0, j); }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[j], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int k0 = 0; k0 < KK; k0 += 32) {
      const bf8 ah = fa(Ah, k0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf8 bh = fb(Bh, k0, j);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void split4(const float (&v)[4], bf4& hi, bf4& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) { hi[r] = (__bf16)v[r]; lo[r] = (__bf16)(v[r] - (float)hi[r]); }
}
// four consecutive steps of one feature / key row -> hi and lo planes, one 8-byte store each
__device__ __forceinline__ void store_t4(__bf16* Ph, __bf16* Pl, int row, int col4, const float (&v)[4]) {
  bf4 hi, lo;
  split4(v, hi, lo);
  const int o = tofs(row, col4);
  *reinterpret_cast<bf4*>(Ph + o) = hi;
  *reinterpret_cast<bf4*>(Pl + o) = lo;
}

__device__ __forceinline__ uint64_t attn_quad(int bh, int T, int q, int key) { return ((uint64_t)bh * T + key) * ((T + 3) >> 2) + (q >> 2); }
__device__ __forceinline__ void attn_keep4(float (&k4)[4], uint64_t seed, uint32_t site, int bh, int T, int q0, int key, float p, float inv_keep) {
  const float4 u = uniform4(seed, site, attn_quad(bh, T, q0, key));
  k4[0] = u.x >= p ? inv_keep : 0.f; k4[1] = u.y >= p ? inv_keep : 0.f;
  k4[2] = u.z >= p ? inv_keep : 0.f; k4[3] = u.w >= p ? inv_keep : 0.f;
}

// ---- x rows of the sample -> registers -> split planes [TS][LDX] (rows >= Tv and columns >= D zero) -----------------------------
template <int KCX>
struct XRows { float4 v[(TS * 8 * KCX) / AF_THR]; };

template <int KCX>
__device__ __forceinline__ void x_request(XRows<KCX>& r, const float* x, long row0, int Tv, int D, int tid) {
  constexpr int kq = 8 * KCX, NIT = (TS * kq) / AF_THR;
  static_assert((TS * kq) % AF_THR == 0, "x tile must divide over the workgroup");
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * AF_THR;
    const int rr = i / kq, k = 4 * (i - rr * kq);
    // unconditional, from a clamped (always legal) address: the padding is applied at the store (DESIGN rule 2)
    r.v[it] = *reinterpret_cast<const float4*>(x + (row0 + min(rr, Tv - 1)) * D + min(k, D - 4));
  }
}
template <int KCX>
__device__ __forceinline__ void x_store(const XRows<KCX>& r, __bf16* Xh, __bf16* Xl, int Tv, int D, int tid) {
  constexpr int kq = 8 * KCX, NIT = (TS * kq) / AF_THR, LDX = 32 * KCX + 16;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = tid + it * AF_THR;
    const int rr = i / kq, k = 4 * (i - rr * kq);
    const bool ok = rr < Tv && k < D;
    const float v[4] = {ok ? r.v[it].x : 0.f, ok ? r.v[it].y : 0.f, ok ? r.v[it].z : 0.f, ok ? r.v[it].w : 0.f};
    bf4 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<bf4*>(Xh + rr * LDX + k) = hi;
    *reinterpret_cast<bf4*>(Xl + rr * LDX + k) = lo;
  }
}

// ---- Q | K | V of one head: [TS x D] x W_h^T + b -> TRANSPOSED planes T[which][hi, lo][HDP][LDT] --------------------------------
// Wave w owns the column tiles w and w + 8 of the head's 3 NTH and all four row tiles: each weight fragment is loaded once per
// workgroup, straight from L2 into registers.  That stream -- 150 KB per head and workgroup, every workgroup of the launch at the same
// time -- is what the projection costs (phase stamps: 9-10 k cycles with the panel requested where it is used, ~2 k of them MFMA):
// the panel of a head is therefore REQUESTED A WHOLE HEAD EARLIER (qkv_request: at kernel entry for head 0, right behind the
// previous head's projection for the next), into registers that are free through the attention phases.
// Rows >= Tv and features >= hd are stored as zeros (what the padded kernels' zero padding gives: dead keys stay finite, padded
// features add nothing).
template <int KCX>
struct QkvPanel { bf8 h[2][KCX], l[2][KCX]; float bs[2]; };

template <int NTH, int KCX>
__device__ __forceinline__ void qkv_request(QkvPanel<KCX>& p, const __bf16* wf_h, const float* bias, int D, int hd, int h, int wave, int lane) {
  constexpr int NCT3 = 3 * NTH;
  static_assert(NCT3 <= 2 * AF_WV, "two column tiles per wave");
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ct = min(wave + AF_WV * s, NCT3 - 1);          // clamped: unconditional loads (wave 7 has one tile: its second is a copy)
    const __bf16* t = wf_h + (size_t)ct * (KCX * 2 * 512) + lane * 8;
#pragma unroll
    for (int kc = 0; kc < KCX; ++kc) {
      p.h[s][kc] = *reinterpret_cast<const bf8*>(t + (kc * 2 + 0) * 512);
      p.l[s][kc] = *reinterpret_cast<const bf8*>(t + (kc * 2 + 1) * 512);
    }
    const int whc = ct / NTH, cl = 16 * (ct - whc * NTH) + (lane & 15);
    p.bs[s] = bias[whc * D + h * hd + min(cl, hd - 1)];
  }
}

template <int NTH, int KCX>
__device__ __forceinline__ void qkv_project(const QkvPanel<KCX>& p, const __bf16* Xh, const __bf16* Xl, int hd, int Tv, __bf16* Tp, int wave,
                                            int lane, bool one) {
  constexpr int NCT3 = 3 * NTH, HDP = 32 * ((16 * NTH + 31) / 32), LDX = 32 * KCX + 16;
  const bool two = wave + AF_WV < NCT3;                      // wave-uniform
  const int aoff = (lane & 15) * LDX + 8 * (lane >> 4);
  f32x4 acc[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[s][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!one) {
    // per reduction step: the A fragments of all four row tiles, then the three split products as three SWEEPS over the eight
    // (column tile, row tile) accumulators -- consecutive MFMAs never touch the same accumulator (a dependent MFMA waits for its
    // predecessor's passes: the tile-after-tile order ran the matrix pipe at ~50 %)
#pragma unroll
    for (int kc = 0; kc < KCX; ++kc) {
      bf8 ah[4], al[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        ah[rt] = *reinterpret_cast<const bf8*>(Xh + rt * 16 * LDX + aoff + kc * 32);
        al[rt] = *reinterpret_cast<const bf8*>(Xl + rt * 16 * LDX + aoff + kc * 32);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], p.h[s][kc], acc[s][rt], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.l[s][kc], acc[s][rt], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.h[s][kc], acc[s][rt], 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int kc = 0; kc < KCX; ++kc) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const bf8 ah = *reinterpret_cast<const bf8*>(Xh + rt * 16 * LDX + aoff + kc * 32);
#pragma unroll
        for (int s = 0; s < 2; ++s) acc[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, p.h[s][kc], acc[s][rt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && !two) break;
    const int ct = wave + AF_WV * s;
    const int whc = ct / NTH, cl = 16 * (ct - whc * NTH) + (lane & 15);
    __bf16* Ph = Tp + (size_t)(whc * 2) * HDP * LDT;
    __bf16* Pl = Ph + (size_t)HDP * LDT;
    const bool cok = cl < hd;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int r0 = 16 * rt + 4 * (lane >> 4);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (cok && r0 + r < Tv) ? acc[s][rt][r] + p.bs[s] : 0.f;
      store_t4(Ph, Pl, cl, r0, v);
    }
  }
}

// rows [16 NTH, HDP) of the six transposed planes: never written by the products, read as reduction padding -> zero once
template <int NTH>
__device__ __forceinline__ void zero_pad_rows(__bf16* Tp, int tid) {
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), PADR = HDP - 16 * NTH;
  if (PADR == 0) return;
  constexpr int Q8 = LDT / 4;                                // 8-byte units per row
  for (int i = tid; i < 6 * PADR * Q8; i += AF_THR) {
    const int pl = i / (PADR * Q8), rem = i - pl * (PADR * Q8);
    const int rr = rem / Q8, q = rem - rr * Q8;
    bf4 z;
#pragma unroll
    for (int c = 0; c < 4; ++c) z[c] = (__bf16)0.f;
    *reinterpret_cast<bf4*>(Tp + ((size_t)pl * HDP + 16 * NTH + rr) * LDT + 4 * q) = z;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------------
// ONE: RD_PREC_BF16's one-product form as its own instantiation (a runtime flag put a branch in front of every product: rd_encfuse.hip
// k_enc_post_fwd has the measurement)
template <int NTH, int KCX, bool ONE>
__global__ __launch_bounds__(AF_THR) void k_attn_fwd_fused(FAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  RD_TOUCH_CODE(ONE ? RD_TL_ATTN_FWD_B : RD_TL_ATTN_FWD);                                      // own code -> L2 (rd_common.h; 14 140-byte kernel: ALL of it -- an uncovered tail is fetched cold, line by line, on the pool's slow boxes)
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDX = 32 * KCX + 16, NA = (NTH + 1) / 2;
  constexpr int LDO = 16 * NTH + 4;                          // fp32 row stride of the output stage
  __bf16* Xh = reinterpret_cast<__bf16*>(fsm);
  __bf16* Xl = Xh + TS * LDX;
  __bf16* Tp = Xl + TS * LDX;                                // [3][hi, lo][HDP][LDT]: Q^T, K^T, V^T of the current head
  __bf16 *Qh = Tp, *Ql = Qh + HDP * LDT, *Kh = Ql + HDP * LDT, *Kl = Kh + HDP * LDT, *Vh = Kl + HDP * LDT, *Vl = Vh + HDP * LDT;
  float* mxs = reinterpret_cast<float*>(Vl + HDP * LDT);     // [2][TS] partial row maxima
  float* sms = mxs + 2 * TS;                                 // [2][TS] partial row sums
  __bf16 *Ph = Qh, *Pl = Ql;                                 // (P o M)^T [key][query] overlays Q^T (dead once S is formed)
  float* ost = reinterpret_cast<float*>(Kh);                 // output stage [TS][LDO] overlays K^T (hi + lo: dead once S is formed)
  static_assert(HDP >= TS, "P^T must fit inside the Q^T plane");
  static_assert((size_t)TS * LDO * 4 <= (size_t)2 * HDP * LDT * 2, "output stage must fit inside the K^T planes");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wq = wave & 3, wh = wave >> 2;   // wave: uniform (scalar registers)
  const int b = blockIdx.x;                                  // rank of the sample (plan order)
  const int Tv = __builtin_amdgcn_readfirstlane(a.plan[plan::len_base(a.B) + b]);
  if (Tv <= 0) return;
  const long row0 = __builtin_amdgcn_readfirstlane(a.plan[plan::off_base() + b]);
  uint64_t seedv = a.seed;
  XRows<KCX> xr;
  x_request<KCX>(xr, a.x, row0, Tv, a.D, tid);
  QkvPanel<KCX> pan;                                         // head 0's weight panel: requested behind the rows, used after the first barrier
  qkv_request<NTH, KCX>(pan, a.wf, a.bias, a.D, a.hd, 0, wave, lane);
  __builtin_amdgcn_sched_barrier(0);
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  zero_pad_rows<NTH>(Tp, tid);
  AFSTAMP(0);
  x_store<KCX>(xr, Xh, Xl, Tv, a.D, tid);
  AFSTAMP(1);
  lds_barrier();
  AFSTAMP(2);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  constexpr bool one = ONE;
  const int qr0 = wq * 16 + 4 * (lane >> 4);                 // first of this lane's four query rows
  for (int h = 0; h < a.H; ++h) {
    const int bh = b * a.H + h;
    if (h > 0) zero_pad_rows<NTH>(Tp, tid);                  // the previous head's output stage ran over K^T's padding rows
    qkv_project<NTH, KCX>(pan, Xh, Xl, a.hd, Tv, Tp, wave, lane, one);
    __builtin_amdgcn_sched_barrier(0);
    {   // the NEXT head's panel into the registers this one has just released (past the last head: head 0's again, unused --
        // an unconditional request keeps the wait counts in front of the attention phases exact, DESIGN rule 17)
      const int hn = h + 1 < a.H ? h + 1 : 0;
      qkv_request<NTH, KCX>(pan, a.wf + (size_t)hn * (3 * NTH * KCX * 2 * 512), a.bias, a.D, a.hd, hn, wave, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    AFSTAMP(3 + 8 * h);
    lds_barrier();
    AFSTAMP(4 + 8 * h);
    // ---- S = Q K^T: query tile wq, key tiles 2 wh, 2 wh + 1 ----
    // The dropout keeps of this lane's 2 x 4 probabilities are a function of indices only: computed HERE, unconditionally (p = 0
    // gives 1.0 everywhere), in the same basic block as the S products -- ~100 VALU instructions that issue in the MFMAs' shadow
    // instead of behind the row-max barrier (behind `if (p > 0)` they were a block of their own)
    float kp[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      attn_keep4(kp[j], seedv, a.site, bh, a.T, qr0, min(16 * (2 * wh + j) + (lane & 15), a.T - 1), a.p_drop, inv_keep);
    f32x4 s[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_b16<2, true, true, true, true>(s, Qh, Ql, LDT, wq * 16, Kh, Kl, LDT, 32 * wh, HDP, lane, one);
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = 16 * (2 * wh + j) + (lane & 15);
      const bool dead = key >= Tv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[j][r] = dead ? -INFINITY : s[j][r] * a.scale;
        mx[r] = fmaxf(mx[r], s[j][r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mx[r] = g16_max(mx[r]);
      if ((lane & 15) == 0) mxs[wh * TS + qr0 + r] = mx[r];
    }
    AFSTAMP(5 + 8 * h);
    lds_barrier();                                         // partial maxima visible; everybody is done with Q^T and K^T
    float m_i[4], rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) m_i[r] = fmaxf(mxs[qr0 + r], mxs[TS + qr0 + r]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = 16 * (2 * wh + j) + (lane & 15);
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (s[j][r] == -INFINITY) ? 0.f : __expf(s[j][r] - m_i[r]);
        rsum[r] += p;
        pv[r] = p * kp[j][r];
      }
      store_t4(Ph, Pl, key, qr0, pv);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rsum[r] = g16_sum(rsum[r]);
      if ((lane & 15) == 0) sms[wh * TS + qr0 + r] = rsum[r];
    }
    AFSTAMP(6 + 8 * h);
    lds_barrier();
    AFSTAMP(7 + 8 * h);
    float l_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) l_i[r] = sms[qr0 + r] + sms[TS + qr0 + r];
    // ---- O = (P o M) V: head-dim tiles t0 .. t0 + NA - 1 of this half ----
    const int t0 = wh * NA;
    f32x4 o[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_b16<NA, true, false, true, true>(o, Ph, Pl, LDT, wq * 16, Vh, Vl, LDT, 16 * t0, TS, lane, one);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = qr0 + r;
      const float inv = 1.0f / l_i[r];
#pragma unroll
      for (int j = 0; j < NA; ++j)
        if (t0 + j < NTH) ost[q * LDO + 16 * (t0 + j) + (lane & 15)] = o[j][r] * inv;
      if (wh == 0 && (lane & 15) == 0 && q < Tv) a.lse[(long)bh * a.T + q] = m_i[r] + logf(l_i[r]);
    }
    AFSTAMP(8 + 8 * h);
    lds_barrier();
    AFSTAMP(9 + 8 * h);
    // ---- attention rows of this head out: 16-byte stores, a row's hd floats contiguous ----
    {
      const int qpr = a.hd >> 2;
      for (int e = tid; e < Tv * qpr; e += AF_THR) {
        const int q = e / qpr, c4 = e - q * qpr;
        *reinterpret_cast<float4*>(a.out + (row0 + q) * a.D + h * a.hd + 4 * c4) = *reinterpret_cast<const float4*>(ost + q * LDO + 4 * c4);
      }
    }
    lds_barrier();                                         // the next head's projection rewrites the planes the stage lives in
    AFSTAMP(10 + 8 * h);
  }
}

template <int NTH, int KCX>
constexpr size_t fwd_lds() {
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDX = 32 * KCX + 16;
  return (size_t)2 * TS * LDX * 2 + (size_t)6 * HDP * LDT * 2 + (size_t)4 * TS * 4;
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------------------
// head slice [TS x hd] of a [M, D] tensor: 8 threads per row over the whole workgroup (thread (r, q): the 16-byte chunks q, q + 8, ..
// of row r); every load unconditional from a clamped address, padding applied by the consumer through `ok`
template <int NTH>
struct HRegs { float4 v[(4 * NTH + 7) / 8]; unsigned ok; };
template <int NTH>
__device__ __forceinline__ void head_request(HRegs<NTH>& h, const float* base, long ld, int Tv, int hd, int tl) {
  constexpr int NCH = (4 * NTH + 7) / 8;
  const int t = tl >> 3;
  const bool rok = t < Tv;
  const float* src = base + (long)(rok ? t : 0) * ld;
  unsigned ok = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = 4 * ((tl & 7) + 8 * i);
    const bool cok = c < hd;
    h.v[i] = *reinterpret_cast<const float4*>(src + (cok ? c : 0));
    if (rok && cok) ok |= 1u << i;
  }
  h.ok = ok;
}
template <int NTH>
__device__ __forceinline__ void head_mask(HRegs<NTH>& h) {
  constexpr int NCH = (4 * NTH + 7) / 8;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (!((h.ok >> i) & 1u)) h.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- dx += d{Q,K,V} W_in(which, head): the input gradient of in_proj, accumulated in registers over which and heads ------------
// A = the transposed d{Q,K,V} planes (read transposed: reduction = head feature), B = W_in(which, head)^T tiles straight from L2.
// Wave w owns the column tiles w and w + 8 (NS = 2: waves 0, 1 at NCT = 10) of dx and all four row tiles.  Like the projection's,
// the panels are requested ahead: `which` 0 behind barrier (D) -- two phases before its use --, `which` + 1 before the products
// of `which` (two panels alive).
template <int KB>
struct DxPanel { bf8 h[2][KB], l[2][KB]; };

template <int KCX, int KB, int NS>
__device__ __forceinline__ void dx_request(DxPanel<KB>& p, const __bf16* wb_h, int which, int wave, int lane) {
  constexpr int NCT = 2 * KCX;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const __bf16* t = wb_h + ((size_t)which * NCT + wave + AF_WV * s) * (KB * 2 * 512) + lane * 8;
#pragma unroll
    for (int kc = 0; kc < KB; ++kc) {
      p.h[s][kc] = *reinterpret_cast<const bf8*>(t + (kc * 2 + 0) * 512);
      p.l[s][kc] = *reinterpret_cast<const bf8*>(t + (kc * 2 + 1) * 512);
    }
  }
}
template <int NTH, int KB, int NS, bool ONE>
__device__ __forceinline__ void dx_mma(f32x4 (&dxa)[2][4], const DxPanel<KB>& p, const __bf16* Tp, int which, int lane) {
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32);
  const __bf16* Ah = Tp + (size_t)(which * 2) * HDP * LDT;
  const __bf16* Al = Ah + (size_t)HDP * LDT;
#pragma unroll
  for (int kc = 0; kc < KB; ++kc) {
    bf8 ah[4], al[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      ah[rt] = frag_ts(Ah, 32 * kc, 16 * rt, lane);
      if (!ONE) al[rt] = frag_ts(Al, 32 * kc, 16 * rt, lane);
    }
    if (!ONE) {                                    // three sweeps over the accumulators (see qkv_project)
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dxa[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], p.h[s][kc], dxa[s][rt], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dxa[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.l[s][kc], dxa[s][rt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) dxa[s][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], p.h[s][kc], dxa[s][rt], 0, 0, 0);
  }
}
// p0: the panel of `which` 0, already requested
template <int NTH, int KCX, int NS, bool ONE>
__device__ __forceinline__ void dx_phase(f32x4 (&dxa)[2][4], DxPanel<(16 * NTH + 31) / 32>& p0, const __bf16* Tp, const __bf16* wb_h, int wave,
                                         int lane) {
  constexpr int KB = (16 * NTH + 31) / 32;
  DxPanel<KB> p1;
  dx_request<KCX, KB, NS>(p1, wb_h, 1, wave, lane);
  __builtin_amdgcn_sched_barrier(0);
  dx_mma<NTH, KB, NS, ONE>(dxa, p0, Tp, 0, lane);
  __builtin_amdgcn_sched_barrier(0);
  dx_request<KCX, KB, NS>(p0, wb_h, 2, wave, lane);
  __builtin_amdgcn_sched_barrier(0);
  dx_mma<NTH, KB, NS, ONE>(dxa, p1, Tp, 1, lane);
  __builtin_amdgcn_sched_barrier(0);
  dx_mma<NTH, KB, NS, ONE>(dxa, p0, Tp, 2, lane);
  __builtin_amdgcn_sched_barrier(0);
}

template <int NTH, int KCX, bool ONE>
__global__ __launch_bounds__(AF_THR) void k_attn_bwd_fused(FAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  RD_TOUCH_CODE(ONE ? RD_TL_ATTN_BWD_B : RD_TL_ATTN_BWD);                                      // own code -> L2 (32 028-byte kernel, all of it)
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDX = 32 * KCX + 16, NA = (NTH + 1) / 2, KB = HDP / 32, LDB = HDP + 16;
  constexpr int NCT = 2 * KCX;                               // 16-column tiles of D (padded to 32 KCX)
  constexpr int LDS_DX = 32 * KCX + 4;                       // fp32 row stride of the dx stage
  // R1: x planes, then the score planes of the head, at the end the dx stage
  __bf16* Xh = reinterpret_cast<__bf16*>(fsm);
  __bf16* Xl = Xh + TS * LDX;
  __bf16 *Ph = Xh, *Pl = Ph + TS * LDT, *Sh = Pl + TS * LDT, *Sl = Sh + TS * LDT;     // (P o M)^T, dS^T: [key][query]
  float* dxs = reinterpret_cast<float*>(fsm);
  static_assert((size_t)4 * TS * LDT * 2 <= (size_t)2 * TS * LDX * 2, "score planes must fit inside the x planes");
  static_assert((size_t)TS * LDS_DX * 4 <= (size_t)2 * TS * LDX * 2, "dx stage must fit inside the x planes");
  // R2: Q^T, K^T, V^T of the head, later dQ^T, dK^T, dV^T
  __bf16* Tp = Xl + TS * LDX;
  __bf16 *Qh = Tp, *Ql = Qh + HDP * LDT, *Kh = Ql + HDP * LDT, *Kl = Kh + HDP * LDT, *Vh = Kl + HDP * LDT, *Vl = Vh + HDP * LDT;
  // R3: dO [query][head dim]
  __bf16* Oh = Vl + HDP * LDT;
  __bf16* Ol = Oh + TS * LDB;
  float* lse_s = reinterpret_cast<float*>(Ol + TS * LDB);
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wq = wave & 3, wh = wave >> 2;   // wave: uniform (scalar registers)
  const int b = blockIdx.x;
  const int Tv = __builtin_amdgcn_readfirstlane(a.plan[plan::len_base(a.B) + b]);
  if (Tv <= 0) return;
  const long row0 = __builtin_amdgcn_readfirstlane(a.plan[plan::off_base() + b]);
  const int g0 = __builtin_amdgcn_readfirstlane(a.plan[plan::coff_base(a.B, a.T) + b]);     // first 16-row group of the sample's row tiles
  const int ngrp = (Tv + 15) >> 4;
  // the group space ends with this sample AND on an odd group: the second half of the last chunk belongs to nobody -> zeros
  const bool tail_zero = (g0 + ngrp) == __builtin_amdgcn_readfirstlane(a.plan[plan::coff_base(a.B, a.T) + a.B]) && ((g0 + ngrp) & 1);
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  constexpr bool one = ONE;
  // dx accumulators: wave w owns the column tiles w and w + 8 (< NCT) of dx and all four row tiles, over BOTH heads
  f32x4 dxa[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) dxa[s][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool two = wave + AF_WV < NCT;                       // wave-uniform
  constexpr int NQ = (TS * 8 * KCX + AF_THR - 1) / AF_THR;
  QkvPanel<KCX> pan;
  // dO and O slices of a head + its LSE: requested one head AHEAD (head 0: here), so that only x is waited for at a head's start
  HRegs<NTH> dov, ov; float lrow;
  head_request<NTH>(dov, a.dout + row0 * a.D, a.D, Tv, a.hd, tid);
  head_request<NTH>(ov, a.out + row0 * a.D, a.D, Tv, a.hd, tid);
  lrow = a.lse[(long)(b * a.H) * a.T + min(tid >> 3, Tv - 1)];
  for (int h = 0; h < a.H; ++h) {
    const int bh = b * a.H + h;
    // ---- requests: x rows (all threads), dO and O slices of the head + LSE (waves 4-7), then the projection's weight panel ----
    // (the thread index goes through an opaque move per head: everything derived from it -- five row / column predicates and
    // addresses per tile -- is loop-invariant, and hoisted out of the head loop it stayed live across the whole kernel: spills)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    XRows<KCX> xr;
    x_request<KCX>(xr, a.x, row0, Tv, a.D, tl);
    // (the lane index too: the panels' 20 + 36 tile-part addresses are loop-invariant and were kept in registers across the loop)
    int lq = lane;
    asm volatile("" : "+v"(lq));
    qkv_request<NTH, KCX>(pan, a.wf + (size_t)h * (3 * NTH * KCX * 2 * 512), a.bias, a.D, a.hd, h, wave, lq);
    AFSTAMP(32 + 16 * h);
    __builtin_amdgcn_sched_barrier(0);
    if (h == 0) zero_pad_rows<NTH>(Tp, tl);
    x_store<KCX>(xr, Xh, Xl, Tv, a.D, tl);
    AFSTAMP(33 + 16 * h);
    lds_barrier();                                           // (A) x planes complete; the previous head's dQ^T.. / dO planes are dead
    AFSTAMP(34 + 16 * h);
    if (h == 0) {
      // ---- row tiles of x: 16-row groups of the per-sample group space (rd_encfuse.hip export_tiles' lane map: a wave-read
      // covers TWO groups, lanes 0-31 the even one of the pair, 32-63 the odd one) ----
      const int i16 = lq & 15, g2 = (lq >> 4) & 1, sub = lq >> 5;
      const int npair = (ngrp + 1) >> 1;
      for (int t = wave; t < npair * NCT * 2; t += AF_WV) {
        const int plane = t & 1, cj = t >> 1;
        const int pr = cj / NCT, j = cj - pr * NCT;
        const int lg = 2 * pr + sub;                          // local group of this half-wave
        const __bf16* src = (plane ? Xl : Xh) + (16 * min(lg, ngrp - 1) + 8 * g2 + (i16 >> 2)) * LDX + 16 * j + 4 * (i16 & 3);
        const sh4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src));
        const sh4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sh4 __attribute__((address_space(3)))*)(src + 4 * LDX));
        const sh8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const int gg = g0 + lg;
        if (lg < ngrp)
          *reinterpret_cast<sh8*>(a.xt + (((size_t)(gg >> 1) * NCT + j) * 2 + plane) * 512 + (32 * (gg & 1) + 16 * g2 + i16) * 8) = o;
      }
      if (tail_zero) {
        const int gg = g0 + ngrp;                             // the unowned odd group
        const sh8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < NCT * 2 * 32; i += AF_THR) {
          const int slot = i & 31, jp = i >> 5;
          *reinterpret_cast<sh8*>(a.xt + ((size_t)(gg >> 1) * NCT * 2 + jp) * 512 + (32 + slot) * 8) = z;
        }
      }
    }
    {
      // dO -> [query][head dim] planes (zero padded to HDP columns); delta = rowsum(dO * O) in fp32: 8 threads per row
      constexpr int NCH = (4 * NTH + 7) / 8;
      head_mask<NTH>(dov);
      head_mask<NTH>(ov);
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        d += (dov.v[i].x * ov.v[i].x + dov.v[i].y * ov.v[i].y) + (dov.v[i].z * ov.v[i].z + dov.v[i].w * ov.v[i].w);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      const int o = (tl >> 3) * LDB + 4 * (tl & 7);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        static_assert(32 * NCH == HDP, "8 threads x NCH chunks cover exactly the padded head width (chunks >= hd arrive as zeros)");
        const float v[4] = {dov.v[i].x, dov.v[i].y, dov.v[i].z, dov.v[i].w};
        bf4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf4*>(Oh + o + 32 * i) = hi;
        *reinterpret_cast<bf4*>(Ol + o + 32 * i) = lo;
      }
      if ((tl & 7) == 0) { dl_s[tl >> 3] = d; lse_s[tl >> 3] = lrow; }
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // the NEXT head's dO / O / LSE (past the last head: head 0's again, unused -- unconditional, DESIGN rule 17)
      const int hn = h + 1 < a.H ? h + 1 : 0;
      head_request<NTH>(dov, a.dout + row0 * a.D + hn * a.hd, a.D, Tv, a.hd, tl);
      head_request<NTH>(ov, a.out + row0 * a.D + hn * a.hd, a.D, Tv, a.hd, tl);
      lrow = a.lse[(long)(b * a.H + hn) * a.T + min(tl >> 3, Tv - 1)];
    }
    AFSTAMP(35 + 16 * h);
    __builtin_amdgcn_sched_barrier(0);
    qkv_project<NTH, KCX>(pan, Xh, Xl, a.hd, Tv, Tp, wave, lq, one);
    AFSTAMP(36 + 16 * h);
    lds_barrier();                                           // (B) Q^T, K^T, V^T, dO, delta, LSE complete; the x planes are dead
    AFSTAMP(37 + 16 * h);
    // ---- S = Q K^T, dP = dO V^T for query tile wq, key tiles 2 wh, 2 wh + 1 ----
    // (every phase works from its own opaque copy of the lane index: the LDS addresses of its fragments are loop-invariant, the
    // planes span 150 KB -- beyond one base + 16-bit offset -- and hoisted out of the head loop they cost ~90 registers: spills)
    int l1 = lane;
    asm volatile("" : "+v"(l1));
    const int qr1 = wq * 16 + 4 * (l1 >> 4);
    float kp[2][4];                                          // dropout keeps: index-only, in the products' shadow (see the forward)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      attn_keep4(kp[j], seedv, a.site, bh, a.T, qr1, min(16 * (2 * wh + j) + (l1 & 15), a.T - 1), a.p_drop, inv_keep);
    f32x4 s[2], dp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
    mma_b16<2, true, true, true, true>(s, Qh, Ql, LDT, wq * 16, Kh, Kl, LDT, 32 * wh, HDP, l1, one);
    mma_b16<2, false, true, false, true>(dp, Oh, Ol, LDB, wq * 16, Vh, Vl, LDT, 32 * wh, HDP, l1, one);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int key = 16 * (2 * wh + j) + (l1 & 15);
      const bool dead = key >= Tv;
      const float (&k4)[4] = kp[j];
      float pm[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = qr1 + r;
        pm[r] = 0.f; ds[r] = 0.f;
        if (!dead && row < Tv) {
          const float p = __expf(s[j][r] * a.scale - lse_s[row]);
          pm[r] = p * k4[r];
          ds[r] = p * (dp[j][r] * k4[r] - dl_s[row]) * a.scale;
        }
      }
      store_t4(Ph, Pl, key, qr1, pm);
      store_t4(Sh, Sl, key, qr1, ds);
    }
    AFSTAMP(38 + 16 * h);
    lds_barrier();                                           // (C)
    AFSTAMP(39 + 16 * h);
    const int t0 = wh * NA;                                  // first head-dim tile of this half
    int l2 = lane;
    asm volatile("" : "+v"(l2));
    f32x4 dq[NA], dk[NA], dv[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) { dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[j] = dq[j]; dv[j] = dq[j]; }
    mma_b16<NA, true, false, true, true>(dq, Sh, Sl, LDT, wq * 16, Kh, Kl, LDT, 16 * t0, TS, l2, one);     // dQ = dS K      (rows: queries)
    mma_b16<NA, false, false, true, true>(dk, Sh, Sl, LDT, wq * 16, Qh, Ql, LDT, 16 * t0, TS, l2, one);    // dK = dS^T Q    (rows: keys)
    mma_b16<NA, false, true, true, false>(dv, Ph, Pl, LDT, wq * 16, Oh, Ol, LDB, 16 * t0, TS, l2, one);    // dV = (P o M)^T dO
    AFSTAMP(40 + 16 * h);
    lds_barrier();                                           // (D) everybody is done with Q^T, K^T, V^T, the score planes and dO
    AFSTAMP(41 + 16 * h);
    // the first input-gradient panel of this head: in flight through the next two phases
    const __bf16* wb_h = a.wb + (size_t)h * (3 * NCT * KB * 2 * 512);
    DxPanel<KB> dp0;
    int ld = lane;
    asm volatile("" : "+v"(ld));
    if (two) dx_request<KCX, KB, 2>(dp0, wb_h, 0, wave, ld); else dx_request<KCX, KB, 1>(dp0, wb_h, 0, wave, ld);
    __builtin_amdgcn_sched_barrier(0);
    // ---- dQ, dK, dV -> transposed planes in place of Q^T, K^T, V^T (rows >= Tv and features >= hd come out as exact zeros) ----
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      if (t0 + j < NTH) {
        const int f = 16 * (t0 + j) + (ld & 15), qr2 = wq * 16 + 4 * (ld >> 4);
        const float vq[4] = {dq[j][0], dq[j][1], dq[j][2], dq[j][3]};
        const float vk[4] = {dk[j][0], dk[j][1], dk[j][2], dk[j][3]};
        const float vv[4] = {dv[j][0], dv[j][1], dv[j][2], dv[j][3]};
        store_t4(Qh, Ql, f, qr2, vq);
        store_t4(Kh, Kl, f, qr2, vk);
        store_t4(Vh, Vl, f, qr2, vv);
      }
    }
    AFSTAMP(42 + 16 * h);
    lds_barrier();                                           // (E)
    AFSTAMP(43 + 16 * h);
    // ---- (a) row tiles of dqkv, head-padded column layout: tile (which, h, j); 16-row groups as for x ----
    {
      const int i16 = ld & 15, g2 = (ld >> 4) & 1, sub = ld >> 5;
      const int nctp = 3 * a.H * NTH;
      const int npair = (ngrp + 1) >> 1;
      for (int t = wave; t < 3 * NTH * npair * 2; t += AF_WV) {
        const int plane = t & 1, u = t >> 1;
        const int pr = u / (3 * NTH), wj = u - pr * (3 * NTH);
        const int which = wj / NTH, j = wj - which * NTH;
        const int lg = 2 * pr + sub;
        const __bf16* src = Tp + (size_t)(which * 2 + plane) * HDP * LDT + tofs(16 * j + i16, 16 * min(lg, ngrp - 1) + 8 * g2);
        const sh8 o = *reinterpret_cast<const sh8*>(src);
        const int jt = (which * a.H + h) * NTH + j;
        const int gg = g0 + lg;
        if (lg < ngrp)
          *reinterpret_cast<sh8*>(a.dt + (((size_t)(gg >> 1) * nctp + jt) * 2 + plane) * 512 + (32 * (gg & 1) + 16 * g2 + i16) * 8) = o;
      }
      if (tail_zero) {
        const int gg = g0 + ngrp;
        const sh8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < 3 * NTH * 2 * 32; i += AF_THR) {
          const int slot = i & 31, jp = i >> 5;               // jp = (which-j index, plane) of this head's 3 NTH column tiles
          const int wj = jp >> 1, plane = jp & 1;
          const int which = wj / NTH, j = wj - which * NTH;
          const int jt = (which * a.H + h) * NTH + j;
          *reinterpret_cast<sh8*>(a.dt + (((size_t)(gg >> 1) * nctp + jt) * 2 + plane) * 512 + (32 + slot) * 8) = z;
        }
      }
    }
    AFSTAMP(44 + 16 * h);
    // ---- (b) dx += dQ W_q,h + dK W_k,h + dV W_v,h ----
    if (two) { if (one) dx_phase<NTH, KCX, 2, true>(dxa, dp0, Tp, wb_h, wave, ld); else dx_phase<NTH, KCX, 2, false>(dxa, dp0, Tp, wb_h, wave, ld); }
    else { if (one) dx_phase<NTH, KCX, 1, true>(dxa, dp0, Tp, wb_h, wave, ld); else dx_phase<NTH, KCX, 1, false>(dxa, dp0, Tp, wb_h, wave, ld); }
    AFSTAMP(45 + 16 * h);
    // no barrier here: the next head first rewrites the x planes (R1, not read above) and reaches its barrier (A) before it
    // touches R2 / R3
  }
  // ---- dx = accumulated products + ds1 -> rows out as 16-byte stores ----
  // (R1 was last read before barrier (D) of the last head: free).  The residual rows are requested first: they travel under the
  // tail of the products and the stage writes.
  float4 r4[NQ];
  {
    const int qpr = a.D >> 2;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + i * AF_THR, q = e / qpr, c4 = e - q * qpr;
      r4[i] = *reinterpret_cast<const float4*>(a.ds1 + (row0 + min(q, Tv - 1)) * a.D + 4 * c4);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && !two) break;
    const int ct = wave + AF_WV * s;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dxs[(16 * rt + 4 * (lane >> 4) + r) * LDS_DX + 16 * ct + (lane & 15)] = dxa[s][rt][r];
  }
  AFSTAMP(64);
  lds_barrier();
  {
    const int qpr = a.D >> 2;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int e = tid + i * AF_THR, q = e / qpr, c4 = e - q * qpr;
      if (q < Tv) {
        const float4 s4 = *reinterpret_cast<const float4*>(dxs + q * LDS_DX + 4 * c4);
        *reinterpret_cast<float4*>(a.dx + (row0 + q) * a.D + 4 * c4) = make_float4(s4.x + r4[i].x, s4.y + r4[i].y, s4.z + r4[i].z, s4.w + r4[i].w);
      }
    }
  }
  AFSTAMP(65);
}

template <int NTH, int KCX>
constexpr size_t bwd_lds() {
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDX = 32 * KCX + 16, LDB = HDP + 16;
  return (size_t)2 * TS * LDX * 2 + (size_t)6 * HDP * LDT * 2 + (size_t)2 * TS * LDB * 2 + (size_t)2 * TS * 4;
}

constexpr int AF_NTH = 5, AF_KCX = 5;

}  // namespace

extern "C" void rd_debug_set_attnfuse_stamps(void* p) { g_af_stamps = (unsigned long long*)p; }   // not part of the ABI

// ---- host interface (rd_temporal.hip) -----------------------------------------------------------------------------------------
// RD_ATTN_FUSE=0: the round-3 launches (QKV row-block product + attention per (sample, head) + QKV input-gradient product)
bool attnfuse_ok(int T, int D, int H, int hd) {
  const char* e = getenv("RD_ATTN_FUSE");               // read per call (tests compare both paths in one process)
  const bool enabled = !(e && atoi(e) == 0);
  return enabled && precision() != RD_PREC_FP32 && T <= TS && hd <= 16 * AF_NTH && hd * H == D && (D + 31) / 32 == AF_KCX && (D % 4) == 0 &&
         (hd % 4) == 0 && H >= 1 && H <= 2 && 3 * AF_NTH <= 2 * AF_WV;        // H <= 2: the split-job tables of a layer are sized for 6 H = 12 jobs
}
// bf16 elements of the two weight-tile arrays of a layer
size_t attnfuse_wf_elems(int H) { return (size_t)H * 3 * AF_NTH * AF_KCX * 2 * 512; }
size_t attnfuse_wb_elems(int H) { return (size_t)H * 3 * (2 * AF_KCX) * ((16 * AF_NTH + 31) / 32) * 2 * 512; }
// split jobs of one layer: (which, head) blocks of in_proj_weight [3D, D] as forward operands and, transposed, as input-gradient operands
int attnfuse_split_specs(const float* in_proj_w, int D, int H, int hd, void* wf, void* wb, WsplitSpec* out) {
  constexpr int KB = (16 * AF_NTH + 31) / 32;
  int n = 0;
  for (int h = 0; h < H; ++h)
    for (int which = 0; which < 3; ++which) {
      const float* W = in_proj_w + ((size_t)which * D + (size_t)h * hd) * D;
      out[n++] = WsplitSpec{W, hd, D, 0, (__bf16*)wf + ((size_t)(h * 3 + which) * AF_NTH) * (AF_KCX * 2 * 512)};
      out[n++] = WsplitSpec{W, hd, D, 1, (__bf16*)wb + ((size_t)(h * 3 + which) * (2 * AF_KCX)) * (KB * 2 * 512)};
    }
  return n;
}
int attnfuse_padded_cols(int H) { return 3 * H * 16 * AF_NTH; }      // columns of the head-padded dqkv tile layout
int attnfuse_nth() { return AF_NTH; }

static void fill_args(FAttnArgs& a, const float* x, const void* wf, const void* wb, const float* bias, const int32_t* plan, int T, int B,
                      int D, int H, int hd, float p_drop, uint64_t seed, uint32_t site) {
  a.x = x; a.wf = (const __bf16*)wf; a.wb = (const __bf16*)wb; a.bias = bias; a.plan = plan;
  a.T = T; a.B = B; a.D = D; a.H = H; a.hd = hd;
  a.scale = 1.0f / sqrtf((float)hd); a.p_drop = p_drop; a.seed = seed; a.site = site; a.seed_cell = seed_cell();
  a.one = precision() == RD_PREC_BF16; a.stamps = g_af_stamps;
}

int launch_attn_fused_fwd(const float* x, const void* wf, const float* bias, const int32_t* plan, int T, int B, int D, int H, int hd,
                          float p_drop, uint64_t seed, uint32_t site, float* out, float* lse, hipStream_t st) {
  if (!plan) return fail(RD_EINVAL, "fused attention: needs a token plan");
  FAttnArgs a{};
  fill_args(a, x, wf, nullptr, bias, plan, T, B, D, H, hd, p_drop, seed, site);
  a.out = out; a.lse = lse;
  constexpr size_t lds = fwd_lds<AF_NTH, AF_KCX>();
  if (a.one) { RD_LDS_ATTR((k_attn_fwd_fused<AF_NTH, AF_KCX, true>), lds);
               hipLaunchKernelGGL((k_attn_fwd_fused<AF_NTH, AF_KCX, true>), dim3(B), dim3(AF_THR), lds, st, a); }
  else { RD_LDS_ATTR((k_attn_fwd_fused<AF_NTH, AF_KCX, false>), lds);
         hipLaunchKernelGGL((k_attn_fwd_fused<AF_NTH, AF_KCX, false>), dim3(B), dim3(AF_THR), lds, st, a); }
  return check_launch("k_attn_fwd_fused");
}

int launch_attn_fused_bwd(const float* x, const void* wf, const void* wb, const float* bias, const int32_t* plan, int T, int B, int D, int H,
                          int hd, float p_drop, uint64_t seed, uint32_t site, const float* out, const float* lse, const float* dout,
                          const float* ds1, float* dx, void* xt, void* dt, hipStream_t st) {
  if (!plan) return fail(RD_EINVAL, "fused attention: needs a token plan");
  FAttnArgs a{};
  fill_args(a, x, wf, wb, bias, plan, T, B, D, H, hd, p_drop, seed, site);
  a.out = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.dout = dout; a.ds1 = ds1; a.dx = dx;
  a.xt = (__bf16*)xt; a.dt = (__bf16*)dt;
  constexpr size_t lds = bwd_lds<AF_NTH, AF_KCX>();
  if (a.one) { RD_LDS_ATTR((k_attn_bwd_fused<AF_NTH, AF_KCX, true>), lds);
               hipLaunchKernelGGL((k_attn_bwd_fused<AF_NTH, AF_KCX, true>), dim3(B), dim3(AF_THR), lds, st, a); }
  else { RD_LDS_ATTR((k_attn_bwd_fused<AF_NTH, AF_KCX, false>), lds);
         hipLaunchKernelGGL((k_attn_bwd_fused<AF_NTH, AF_KCX, false>), dim3(B), dim3(AF_THR), lds, st, a); }
  return check_launch("k_attn_bwd_fused");
}

}  // namespace rd
