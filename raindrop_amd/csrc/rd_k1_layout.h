// rd_k1_layout.h -- device-memory formats shared by the fused message-passing kernels (rd_msgpass_fused.hip)
// and their weight-gradient kernel (rd_msgpass_dw.hip).  Fused envelope: F <= 48 sensors, d_ob = 4,
// K = T*d_ob <= 240, K % 16 == 0 (the P19 shape).
//
// Everything an MFMA operand is loaded from is stored as NATIVE TILES: one v_mfma_f32_16x16x32_bf16 operand
// fragment = 64 lanes x 8 bf16 = 1 KB, laid out [lane][8] so that one wave-wide 16-byte load (or one LDS-DMA
// instruction) fetches it as ONE contiguous kilobyte.  Measured on MI355X (tools/probe_l2stream.hip): an
// L2-resident 256 KB weight set streams into a CU at 61 B/clk (86 B/clk by LDS-DMA) in this form, but at only
// 16 B/clk when every wave-load touches 16 rows x 64 B of a row-major plane (the round-1 layout).
//
//   lane l of a fragment holds index (l & 15) of the tile's 16 "free" indices and reduction elements
//   8*(l >> 4) .. 8*(l >> 4) + 7 of its 32.
//
// Weight tiles  wt[layer 2][orient 2][j < nct][kc < 8][part 2 (hi, lo)][64][8]:
//   orient 0 (forward  Y = X W^T):  free index n = 16 j + c, reduction k = 32 kc + ...: W[n][k]
//   orient 1 (backward dX = dZ W):  free index k = 16 j + c, reduction n = 32 kc + ...: W[n][k]
//   (zero where the reduction index is >= K).
//
// Row tiles ("t-planes", the operands of dW = dZ^T In, whose reduction runs over the B*F graph rows):
//   tp[tensor][s < S][j < nct][part 2][64][8]: free index = column 16 j + c of the [B*F, K] tensor, reduction =
//   32 graph rows.  With q = F / 32 and rem = F % 32 the rows of sample b are grouped as
//     main tiles      s = b*q + m         rows 32 m .. 32 m + 31 of the sample         (m < q)
//     leftover tiles  s = B*q + b / per   rows 32 q + li (li < rem) at position (b % per)*rem + li, per = 32 / rem
//   so no reduction slot is wasted on padding (P19: 256 main + 16 leftover tiles = 8704 / 32 exactly).  Positions
//   of a leftover tile that no sample covers are written as zeros by the sample that owns position 0.
//
// ReLU gates.  Y1 > 0 (m1): the 64-bit LANE MASKS of the epilogue that made the values, [slot][column tile j][row tile rt][e < 4]: bit
// l = (Y1[16 rt + (l & 15)][16 j + 4 (l >> 4) + e] > 0) -- the backward's epilogue owns the same elements per lane and applies a mask as
// the SGPR-pair operand of one v_cndmask.  Y2 > 0 (m2) and X > 0 (mx): one byte per (slot, t, f) cell with bit c = channel c, the
// order in which the forward's scatter / embedding and the backward's dz gather / dR_u pass walk the cells.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rd {
namespace k1 {

constexpr int KP = 256;            // padded reduction length of the K x K products
constexpr int NKC = KP / 32;       // reduction steps per product
constexpr int TILE = 512;          // bf16 elements of one operand fragment (1 KB)

struct Layout {
  int B, T, F, K, nct, RT, q, rem, per, S;
  size_t wt_bytes, tp_bytes, g1_bytes, g2_bytes, mx_bytes;
};

// the per-sample "1 + last observed step" array rides behind the embedding's gate bytes
inline size_t lin_offset(int B, int T, int F) { return ((size_t)B * F * T + 255) / 256 * 256; }

inline Layout make_layout(int B, int T, int F) {
  Layout L;
  L.B = B; L.T = T; L.F = F; L.K = T * 4; L.nct = L.K / 16; L.RT = (F + 15) / 16;
  L.q = F / 32; L.rem = F % 32; L.per = L.rem ? 32 / L.rem : 1;
  L.S = B * L.q + (L.rem ? (B + L.per - 1) / L.per : 0);
  L.wt_bytes = (size_t)4 * L.nct * NKC * 2 * TILE * 2;
  L.tp_bytes = (size_t)L.S * L.nct * 2 * TILE * 2;
  L.g1_bytes = (size_t)B * L.nct * L.RT * 4 * sizeof(uint64_t);
  L.g2_bytes = (size_t)B * F * T;
  L.mx_bytes = lin_offset(B, T, F) + (size_t)B * sizeof(int);      // gate bytes, then lin[B] (int per sample slot)
  return L;
}

// weight-gradient kernel: 4 x 4 output tiles per workgroup; the (nct)-th k-tile is a constant "ones" tile whose
// column 0 accumulates the bias gradient
struct DwPlan { int nbn, nbk, nslice, ldp; size_t part_floats; };
inline DwPlan make_dw_plan(const Layout& L) {
  DwPlan p;
  p.nbn = (L.nct + 3) / 4; p.nbk = (L.nct + 1 + 3) / 4;
  int ns = 256 / (2 * p.nbn * p.nbk);
  if (ns > L.S / 4) ns = L.S / 4;                  // at least one reduction tile per wave
  if (ns < 1) ns = 1;
  p.nslice = ns;
  p.ldp = 16 * (L.nct + 1);
  p.part_floats = (size_t)ns * 2 * L.K * p.ldp;
  return p;
}

}  // namespace k1
}  // namespace rd
