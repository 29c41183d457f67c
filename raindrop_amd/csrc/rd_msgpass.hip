// rd_msgpass.hip -- inter-sensor message passing (kernel family K1), general path.
//
// Replaces, for the whole batch at once, the reference's per-sample Python loop
//   code/models_rd.py:285-296   observation embedding  h = relu(repeat_interleave(src) * R_u)
//   code/models_rd.py:313-343   for unit in range(batch): reshape -> ob_propagation ->
//                               ob_propagation_layer2 -> reshape back -> output[:, unit, :] = ...
//   code/Ob_propagation.py:157-228  message(): edge softmax, relu(lin_value(x_i)) * gamma;
//                               aggregate(): scatter-add onto the target node
// On the shipped (default) branch the value multiplied by the softmax is the TARGET node's own
// relu(lin_value(x_i)), so summing the messages into target i gives V[i] * sum_j gamma[j,i]
// (SURVEY.md fact 2).  The per-target coefficient sum `ssum` comes from rd_edge_softmax (the
// graph kernel), and lin_value is evaluated once per NODE instead of once per EDGE.
//
// This file holds the shape-generic path (any T, F): the [B*F, K] x [K, K] products run on the
// tiled MFMA GEMM (rd_gemm.hip) with the ReLU / aggregate-scale / layout-scatter fused into its
// epilogue.  The LDS-resident fused kernel for small K lives in rd_msgpass_fused.hip.
#include <stdlib.h>

#include "rd_common.h"
#include "rd_k1_layout.h"
#include "rd_plan.h"
#include "rd_rng.h"

namespace rd {

// fused LDS-resident path (rd_msgpass_fused.hip) and its weight-gradient kernels (rd_msgpass_dw.hip)
bool fused_msgpass_ok(const rd_shape* s);
int fused_wprep(const k1::Layout& L, const float* W1, const float* W2, void* wt, hipStream_t st);
int fused_msgpass_fwd(const k1::Layout& L, const float* src, const float* R_u, const float* b1, const float* b2,
                      const float* ssum, const void* wt, float p_drop, uint64_t seed, void* tpX, void* tpY1,
                      void* m1, void* m2, void* mx, float* z, int ldz, hipStream_t st, const float* times,
                      const int64_t* lengths, const float* tscale, uint8_t* mask, int d_pe);
int fused_msgpass_bwd(const k1::Layout& L, const float* src, const float* ssum, const void* wt, float p_drop,
                      const void* m1, const void* m2, const void* mx, const float* dz, int ldz, void* tpD1, void* tpD2,
                      void* ones, float* rupart, hipStream_t st, const void* tpX = nullptr, const void* tpY1 = nullptr);
int fused_dw(const k1::Layout& L, const k1::DwPlan& P, const void* tpX, const void* tpY1, const void* tpD1,
             const void* tpD2, const void* ones, float* part, const float* rupart, float* dW1, float* db1, float* dW2, float* db2,
             float* dRu, hipStream_t st);
// streamed weight gradients from row tiles (rd_tile_wgrad.hip, rd_tiles_export.hip; declarations as in rd_temporal.hip)
struct TileWgradJob { const void *tA, *tB; float* part; float *dW, *db; int N, K; const int32_t* s32; int S; int hd, hdp, H, D; };
struct TileColsumJob { const float* x; int M, N, n1; float *out1, *out2; };
size_t tile_elems(long M, int cols);
size_t tile_wgrad_ones_elems();
size_t tile_wgrad_part_floats(int N, int K);
bool tile_wgrad_ok(int N, int K);
int launch_rows_to_tiles(long M, int n, const float* const* x, const long* ld, const int* cols, void* const* tiles, hipStream_t st);
int launch_tile_wgrad(long M, int njobs, const TileWgradJob* jobs, const void* ones, int ncs, const TileColsumJob* cs,
                      hipStream_t st, const int32_t* s32);

namespace {

// X[b,f,t*d+c] = relu(src[t,b,f] * R_u[f*d+c])   (code/models_rd.py:290-291 + the layout change
// of :326-327).  One workgroup per (sample, sensor-chunk); writes are contiguous in K.
__global__ __launch_bounds__(256) void k_obs_embed(const float* __restrict__ src,
                                                   const float* __restrict__ R_u,
                                                   float* __restrict__ X, int B, int T, int F, int d,
                                                   float p_drop, uint64_t seed, const uint64_t* cell) {
  seed = eff_seed(seed, cell);
  const int b = blockIdx.x;
  const int K = T * d;
  const long total = (long)F * K;
  for (long o = blockIdx.y * (long)blockDim.x + threadIdx.x; o < total;
       o += (long)gridDim.y * blockDim.x) {
    const int f = (int)(o / K);
    const int k = (int)(o - (long)f * K);
    const int t = k / d, c = k - t * d;
    float v = fmaxf(src[((long)t * B + b) * (2 * F) + f] * R_u[f * d + c], 0.f);
    // nn.Dropout on h (code/models_rd.py:296); the mask is indexed in h's own [T,B,F*d] order
    if (p_drop > 0.f && v > 0.f)
      v *= dropout_scale(seed, SITE_OBS_EMBED, ((uint64_t)t * B + b) * (F * d) + f * d + c, p_drop,
                         1.0f / (1.0f - p_drop));
    X[(long)b * total + o] = v;
  }
}

// dz2[b,f,t*d+c] = dz[t,b,f*d+c] * ssum[f] * (z[t,b,f*d+c] > 0)
// (backward of `out * gamma` + scatter-add + ReLU of layer 2, read through the [T,B,ldz] layout).
__global__ __launch_bounds__(256) void k_msg_dz2(const float* __restrict__ dz,
                                                 const float* __restrict__ z,
                                                 const float* __restrict__ ssum,
                                                 float* __restrict__ dz2, int B, int T, int F, int d,
                                                 long ldz, const int32_t* __restrict__ sp_row0, const int32_t* __restrict__ sp_len) {
  const int b = blockIdx.x;
  const int Fd = F * d;
  const long total = (long)T * Fd;
  // token plan: sample b's step t lives at row sp_row0[b] + t of z / dz; steps >= sp_len[b] have no row and a zero gradient
  const long row0 = sp_row0 ? sp_row0[b] : b;
  const long rstep = sp_row0 ? 1 : B;
  const int len = sp_row0 ? sp_len[b] : T;
  // iterate in the SOURCE order (t, f, c) so the strided global reads are coalesced
  for (long i = blockIdx.y * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.y * blockDim.x) {
    const int t = (int)(i / Fd);
    const int fc = (int)(i - (long)t * Fd);
    const int f = fc / d, c = fc - f * d;
    float g = 0.f;
    if (t < len) {
      const long zi = (row0 + (long)t * rstep) * ldz + fc;
      g = (z[zi] > 0.f) ? dz[zi] * ssum[f] : 0.f;
    }
    dz2[((long)b * F + f) * ((long)T * d) + t * d + c] = g;
  }
}

// per-sample partial of dR_u: part[b][f*d+c] = sum_t dx[b,f,t*d+c] * (x>0) * src[t,b,f].  One wavefront per (b, f) row of
// T*d contiguous floats, lanes along t (coalesced up to the stride d), fixed-order DPP sum over the 64 lanes.
__global__ __launch_bounds__(256) void k_obs_embed_bwd(const float* __restrict__ dx,
                                                       const float* __restrict__ X,
                                                       const float* __restrict__ src,
                                                       float* __restrict__ part, int B, int T, int F,
                                                       int d, float keep_scale) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= (long)B * F) return;
  const int b = (int)(row / F), f = (int)(row - (long)b * F);
  const long base = row * ((long)T * d);
  for (int c = 0; c < d; ++c) {
    float s = 0.f;
    for (int t = lane; t < T; t += 64) {
      const long o = base + (long)t * d + c;
      if (X[o] > 0.f) s += dx[o] * src[((long)t * B + b) * (2 * F) + f];   // X>0 <=> relu open AND kept
    }
    s = wave_sum64_dpp(s);
    if (lane == 0) part[row * d + c] = s * keep_scale;
  }
}

// z[t,b,f*d+c] = Y[b,f,t*d+c] * rs[b,f]  -- the [F,T*d] -> [T,F*d] layout change of code/models_rd.py:338-342 with the
// aggregate coefficient of a per-sample graph folded in (rs == null: 1).  Iterates in the DESTINATION order (coalesced stores).
__global__ __launch_bounds__(256) void k_rows_to_tokens(const float* __restrict__ Y, const float* __restrict__ rs,
                                                        float* __restrict__ z, int B, int T, int F, int d, long ldz, int bwd) {
  const int b = blockIdx.x;
  const int Fd = F * d;
  const long total = (long)T * Fd;
  float* Yw = const_cast<float*>(Y);
  for (long i = blockIdx.y * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.y * blockDim.x) {
    const int t = (int)(i / Fd);
    const int fc = (int)(i - (long)t * Fd);
    const int f = fc / d, c = fc - f * d;
    const long zi = ((long)t * B + b) * ldz + fc;
    const long yi = ((long)b * F + f) * ((long)T * d) + t * d + c;
    const float r = rs ? rs[(long)b * F + f] : 1.f;
    if (bwd) Yw[yi] = z[zi] * r;                       // dY = dz * rs (z holds dz)
    else z[zi] = Y[yi] * r;
  }
}

// Shapes whose two weight gradients dW_l = dZ_l^T in_l ([K, K], reduction over the B*F graph rows) take the conversion pass + tile
// stream instead of the split-K GEMM pair: enough rows to amortise the conversion and a K the stream's 5 x 6-tile blocks fill
// and many more rows than columns (P12: 9216 x 860: 1.70 -> 1.64 ms/step, 1.45 -> 1.37 in the single-product mode; not SYN256's
// 4096 x 2048, where the pass + stream come out 1 % behind the GEMM pair, nor PAM's 1088 rows).  Part of the workspace LAYOUT, so it
// must not depend on the arithmetic mode or the environment.
static bool wgrad_stream_shape(long M, long K) { return M >= 2048 && K >= 512 && (K % 4) == 0 && M >= 4 * K; }

struct MsgWs {
  float *dz2, *dz1, *dx, *splitk, *colsum, *rupart;
  void* tl[4]; float* part[2];      // wgrad_stream_shape: row tiles of dz2, y1, dz1, x; slice partials of dW2, dW1
  size_t bytes;
  int nsplit, kps;
};

// generic (tiled-GEMM) path
MsgWs carve(const rd_shape* s, void* base) {
  const long B = s->B, F = s->F, K = (long)s->T * s->d_ob;
  const long M = B * F;
  MsgWs w;
  // split the B*F reduction of the weight gradients so that ~256+ workgroups exist
  int kps;
  const int nsplit = splitk_plan(M, (int)K, (int)K, &kps);
  w.nsplit = nsplit; w.kps = kps;
  size_t off = 0;
  auto take = [&](size_t nfloats) { float* p = base ? (float*)((char*)base + off) : nullptr;
                                    off += align_up(nfloats * sizeof(float), 256); return p; };
  w.dz2 = take(M * K); w.dz1 = take(M * K); w.dx = take(M * K);
  w.splitk = take((size_t)2 * wgrad_ws_floats(M, (int)K, (int)K));
  w.colsum = take(colsum_ws_floats((int)M, (int)K));
  w.rupart = take(B * F * s->d_ob);
  for (int i = 0; i < 4; ++i) w.tl[i] = nullptr;
  w.part[0] = w.part[1] = nullptr;
  if (wgrad_stream_shape(M, K)) {
    for (int i = 0; i < 4; ++i) w.tl[i] = take((tile_elems(M, (int)K) + 1) / 2);
    for (int i = 0; i < 2; ++i) w.part[i] = take(tile_wgrad_part_floats((int)K, (int)K));
  }
  w.bytes = off;
  return w;
}

// wt: W1, W2, W2^T, W1^T as native operand tiles (k_wsplit; bf16 modes): the B operands of the two forward products and of the two
// input-gradient products (launch_gemm's panel form), written once by the forward and kept for the backward
struct MsgSaved { float *xsave, *y1save; void* wt[4]; void* ones; int ntile, nkc; size_t bytes; };
MsgSaved carve_saved(const rd_shape* s, void* base) {
  const size_t M = (size_t)s->B * s->F, K = (size_t)s->T * s->d_ob;
  MsgSaved v; size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? (void*)((char*)base + off) : nullptr;
                                  off += align_up(bytes, 256); return p; };
  v.xsave = (float*)take(M * K * sizeof(float));
  v.y1save = (float*)take(M * K * sizeof(float));
  const size_t plane = ((K + 15) / 16 * 16) * ((K + 31) / 32 * 32);                    // elements of one plane (rd_rowgemm.hip: k_wsplit)
  for (int i = 0; i < 4; ++i) v.wt[i] = take(2 * plane * 2);                           // hi + lo, bf16
  v.ones = take(tile_wgrad_ones_elems() * 2);                                          // constant tile of the streamed weight gradients
  v.ntile = ((int)K + 15) / 16; v.nkc = ((int)K + 31) / 32;
  v.bytes = off;
  return v;
}

// fused path (rd_k1_layout.h): forward -> backward hand-over and backward scratch
struct FusedSaved { void *wt, *tpX, *tpY1, *m1, *m2, *mx; size_t bytes; };
FusedSaved carve_fused_saved(const k1::Layout& L, void* base) {
  FusedSaved v; size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? (void*)((char*)base + off) : nullptr;
                                  off += align_up(bytes, 256); return p; };
  v.wt = take(L.wt_bytes); v.tpX = take(L.tp_bytes); v.tpY1 = take(L.tp_bytes);
  v.m1 = take(L.g1_bytes); v.m2 = take(L.g2_bytes); v.mx = take(L.mx_bytes);
  v.bytes = off;
  return v;
}
struct FusedWs { void *tpD1, *tpD2, *ones; float *part, *rupart; size_t bytes; };
FusedWs carve_fused_ws(const k1::Layout& L, const k1::DwPlan& P, void* base) {
  FusedWs w; size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? (void*)((char*)base + off) : nullptr;
                                  off += align_up(bytes, 256); return p; };
  w.tpD1 = take(L.tp_bytes); w.tpD2 = take(L.tp_bytes);
  w.part = (float*)take(P.part_floats * sizeof(float));
  w.rupart = (float*)take((size_t)L.B * L.F * 4 * sizeof(float));
  w.ones = take(3 * k1::TILE * 2);
  w.bytes = off;
  return w;
}
bool fused_envelope(const rd_shape* s) {      // shape part of fused_msgpass_ok (sizes must not depend on the precision mode)
  const int K = s->T * s->d_ob;
  return s->d_ob == 4 && s->F <= 48 && K <= 240 && (K % 16) == 0 && K >= 16;
}

int check_shape(const rd_shape* s) {
  RD_REQUIRE(s != nullptr, "rd_shape is NULL");
  RD_REQUIRE(s->B >= 0 && s->T > 0 && s->F > 0 && s->d_ob > 0, "bad rd_shape (B=%d T=%d F=%d d_ob=%d)",
             s->B, s->T, s->F, s->d_ob);
  RD_REQUIRE((long)s->B * s->F * s->T * s->d_ob < (1L << 31), "B*F*T*d_ob exceeds 2^31");
  return RD_OK;
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" size_t rd_msgpass_workspace_bytes(const rd_shape* s) {
  if (!s || s->T <= 0 || s->F <= 0 || s->d_ob <= 0 || s->B < 0) return 0;
  size_t n = carve(s, nullptr).bytes;
  if (fused_envelope(s)) {
    const k1::Layout L = k1::make_layout(s->B, s->T, s->F);
    const size_t f = carve_fused_ws(L, k1::make_dw_plan(L), nullptr).bytes;
    if (f > n) n = f;
  }
  return n;
}

extern "C" size_t rd_msgpass_saved_bytes(const rd_shape* s) {
  if (!s || s->T <= 0 || s->F <= 0 || s->d_ob <= 0 || s->B < 0) return 0;
  size_t n = carve_saved(s, nullptr).bytes;
  if (fused_envelope(s)) {
    const size_t f = carve_fused_saved(k1::make_layout(s->B, s->T, s->F), nullptr).bytes;
    if (f > n) n = f;
  }
  return n;
}

extern "C" int rd_msgpass_fwd(const rd_shape* s, const float* src, const float* R_u, const float* W1,
                              const float* b1, const float* W2, const float* b2, const float* ssum,
                              float p_drop, uint64_t seed, float* z, int32_t ldz, void* saved,
                              size_t saved_bytes, void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;                       // empty batch
  RD_REQUIRE(src && R_u && W1 && b1 && W2 && b2 && ssum && z && saved, "NULL tensor");
  RD_REQUIRE(ldz >= s->F * s->d_ob, "ldz (%d) < F*d_ob", ldz);
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  MsgSaved v = carve_saved(s, saved);
  RD_REQUIRE(saved_bytes >= v.bytes, "saved buffer too small: %zu < %zu", saved_bytes, v.bytes);
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, T = s->T, F = s->F, d = s->d_ob, K = T * d, M = B * F;
  float* xsave = v.xsave; float* y1save = v.y1save;
  if (fused_msgpass_ok(s)) {
    const k1::Layout L = k1::make_layout(B, T, F);
    FusedSaved fv = carve_fused_saved(L, saved);
    RD_REQUIRE(saved_bytes >= fv.bytes, "saved buffer too small: %zu < %zu", saved_bytes, fv.bytes);
    if ((rc = fused_wprep(L, W1, W2, fv.wt, st))) return rc;
    return fused_msgpass_fwd(L, src, R_u, b1, b2, ssum, fv.wt, p_drop, seed, fv.tpX, fv.tpY1, fv.m1, fv.m2, fv.mx, z, ldz,
                             st, nullptr, nullptr, nullptr, nullptr, 0);
  }
  // token plan on the unfused path (round 4): the products run on all B*F graph rows as before -- lin_value mixes ALL of a
  // sensor's time steps, observed ones past `lengths` included -- only the last product's scatter follows the plan: step t of
  // sample b goes to row brow[b] + t, steps >= blen[b] are not stored (nothing ever reads them)
  const int32_t* tp = token_plan();
  {
    const long per = (long)F * K;
    int gy = (int)((per + 255) / 256); if (gy > 64) gy = 64;
    hipLaunchKernelGGL(k_obs_embed, dim3(B, gy), dim3(256), 0, st, src, R_u, xsave, B, T, F, d,
                       p_drop, seed, seed_cell());
    if ((rc = check_launch("k_obs_embed"))) return rc;
  }
  const bool tiles = precision() != RD_PREC_FP32;      // the split weights feed the panel form of launch_gemm (bf16 modes)
  if (tiles) {
    const WsplitSpec specs[4] = {{W1, K, K, 0, v.wt[0]}, {W2, K, K, 0, v.wt[1]}, {W2, K, K, 1, v.wt[2]}, {W1, K, K, 1, v.wt[3]}};
    void* on[1] = {v.ones};
    if ((rc = launch_wsplit_specs(4, specs, 1, on, st))) return rc;
  }
  GemmArgs g{};
  g.M = M; g.N = K; g.K = K; g.nsplit = 1;
  g.A = xsave; g.sa_m = K; g.sa_k = 1;
  g.B = W1; g.sb_n = K; g.sb_k = 1;
  if (tiles) { g.Btiles = v.wt[0]; g.bt_ntile = v.ntile; g.bt_nkc = v.nkc; }
  g.C = y1save; g.sc_m = K;
  g.bias = b1; g.relu = 1; g.rowscale = ssum; g.rs_period = F;
  if ((rc = launch_gemm(g, st))) return rc;
  g.A = y1save; g.B = W2; g.bias = b2;
  if (tiles) g.Btiles = v.wt[1];
  g.C = z; g.scatter = 1; g.sB = B; g.sF = F; g.sd = d; g.ldz = ldz;
  if (tp) { g.sp_row0 = tp + plan::brow_base(B, T); g.sp_len = tp + plan::blen_base(B, T); }
  return launch_gemm(g, st);
}

// PE + padding mask + message passing in one call: on the fused path ONE launch (plus the weight
// split) produces the whole [T,B,D] input of the temporal encoder and the mask.
static int sensor_stage_fwd_impl(const rd_shape* s, const float* src, const float* times, const int64_t* lengths,
                                 const float* timescales, const float* R_u, const float* W1, const float* b1,
                                 const float* W2, const float* b2, const float* ssum, float p_drop, uint64_t seed,
                                 float* z, uint8_t* mask, void* saved, size_t saved_bytes, void* stream, bool prepared) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(times && lengths && timescales && mask, "NULL tensor");
  RD_REQUIRE(s->d_pe > 0 && (s->d_pe % 2) == 0, "d_pe must be even and positive");
  const int ldz = s->F * s->d_ob + s->d_pe;
  if (fused_msgpass_ok(s)) {
    RD_REQUIRE(src && R_u && W1 && b1 && W2 && b2 && ssum && z && saved, "NULL tensor");
    RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
    const k1::Layout L = k1::make_layout(s->B, s->T, s->F);
    FusedSaved fv = carve_fused_saved(L, saved);
    RD_REQUIRE(saved_bytes >= fv.bytes, "saved buffer too small: %zu < %zu", saved_bytes, fv.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (!prepared && (rc = fused_wprep(L, W1, W2, fv.wt, st))) return rc;
    return fused_msgpass_fwd(L, src, R_u, b1, b2, ssum, fv.wt, p_drop, seed, fv.tpX, fv.tpY1, fv.m1, fv.m2, fv.mx, z, ldz,
                             st, times, lengths, timescales, mask, s->d_pe);
  }
  if ((rc = rd_pe_mask(s, times, lengths, timescales, z, mask, stream))) return rc;
  return rd_msgpass_fwd(s, src, R_u, W1, b1, W2, b2, ssum, p_drop, seed, z, ldz, saved, saved_bytes, stream);
}

extern "C" int rd_sensor_stage_fwd(const rd_shape* s, const float* src, const float* times, const int64_t* lengths,
                                   const float* timescales, const float* R_u, const float* W1, const float* b1,
                                   const float* W2, const float* b2, const float* ssum, float p_drop, uint64_t seed,
                                   float* z, uint8_t* mask, void* saved, size_t saved_bytes, void* stream) {
  return sensor_stage_fwd_impl(s, src, times, lengths, timescales, R_u, W1, b1, W2, b2, ssum, p_drop, seed, z, mask, saved, saved_bytes,
                               stream, false);
}
// the same after rd_step_prepare has written the operand tiles of W1, W2 into `saved` (this step's weights): no split launch here
extern "C" int rd_sensor_stage_fwd_prepared(const rd_shape* s, const float* src, const float* times, const int64_t* lengths,
                                            const float* timescales, const float* R_u, const float* W1, const float* b1,
                                            const float* W2, const float* b2, const float* ssum, float p_drop, uint64_t seed,
                                            float* z, uint8_t* mask, void* saved, size_t saved_bytes, void* stream) {
  return sensor_stage_fwd_impl(s, src, times, lengths, timescales, R_u, W1, b1, W2, b2, ssum, p_drop, seed, z, mask, saved, saved_bytes,
                               stream, true);
}

namespace rd {
// the operand-tile jobs of the fused message-passing stage (W1, W2: forward and transposed orientation) for rd_step_prepare;
// 0 when this shape / mode does not run the fused path (its weights are then read as fp32)
int k1_weight_split_specs(const rd_shape* s, const float* W1, const float* W2, void* saved, size_t saved_bytes, WsplitSpec* out) {
  if (!s || !fused_msgpass_ok(s) || !saved) return 0;
  const k1::Layout L = k1::make_layout(s->B, s->T, s->F);
  FusedSaved fv = carve_fused_saved(L, saved);
  if (saved_bytes < fv.bytes) return 0;
  const size_t per = (size_t)L.nct * k1::NKC * 2 * k1::TILE;           // bf16 elements of one (layer, orientation) tile set
  __bf16* wt = (__bf16*)fv.wt;
  const int K = L.K;
  out[0] = WsplitSpec{W1, K, K, 0, wt}; out[1] = WsplitSpec{W1, K, K, 1, wt + per};
  out[2] = WsplitSpec{W2, K, K, 0, wt + 2 * per}; out[3] = WsplitSpec{W2, K, K, 1, wt + 3 * per};
  return 4;
}
}  // namespace rd

extern "C" int rd_msgpass_bwd(const rd_shape* s, const float* src, const float* R_u, const float* W1,
                              const float* W2, const float* ssum, float p_drop, const void* saved,
                              size_t saved_bytes, const float* z, const float* dz, int32_t ldz,
                              float* dW1, float* db1, float* dW2, float* db2, float* dR_u,
                              void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  RD_REQUIRE(dW1 && db1 && dW2 && db2 && dR_u, "NULL gradient output");
  if (s->B == 0) {
    const int K0 = s->T * s->d_ob;
    hipStream_t st0 = (hipStream_t)stream;
    RD_HIP(hipMemsetAsync(dW1, 0, sizeof(float) * K0 * K0, st0)); RD_HIP(hipMemsetAsync(dW2, 0, sizeof(float) * K0 * K0, st0));
    RD_HIP(hipMemsetAsync(db1, 0, sizeof(float) * K0, st0)); RD_HIP(hipMemsetAsync(db2, 0, sizeof(float) * K0, st0));
    RD_HIP(hipMemsetAsync(dR_u, 0, sizeof(float) * s->F * s->d_ob, st0));
    return RD_OK;
  }
  RD_REQUIRE(src && R_u && W1 && W2 && ssum && saved && z && dz, "NULL tensor");
  MsgSaved sv = carve_saved(s, const_cast<void*>(saved));
  RD_REQUIRE(saved_bytes >= sv.bytes, "saved buffer too small");
  const float* xsave = sv.xsave; const float* y1save = sv.y1save;
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  RD_REQUIRE(ldz >= s->F * s->d_ob, "ldz (%d) < F*d_ob", ldz);
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, T = s->T, F = s->F, d = s->d_ob, K = T * d, M = B * F;
  if (B == 0) {
    RD_HIP(hipMemsetAsync(dW1, 0, sizeof(float) * K * K, st)); RD_HIP(hipMemsetAsync(dW2, 0, sizeof(float) * K * K, st));
    RD_HIP(hipMemsetAsync(db1, 0, sizeof(float) * K, st)); RD_HIP(hipMemsetAsync(db2, 0, sizeof(float) * K, st));
    RD_HIP(hipMemsetAsync(dR_u, 0, sizeof(float) * F * d, st));
    return RD_OK;
  }
  if (fused_msgpass_ok(s)) {
    // chain kernel (dz -> dZ2 -> dZ1 -> dX -> dR_u partials; dZ as row tiles) + streamed weight-gradient product + reduce
    const k1::Layout L = k1::make_layout(B, T, F);
    const k1::DwPlan P = k1::make_dw_plan(L);
    FusedSaved fv = carve_fused_saved(L, const_cast<void*>(saved));
    RD_REQUIRE(saved_bytes >= fv.bytes, "saved buffer too small");
    FusedWs fw = carve_fused_ws(L, P, workspace);
    RD_REQUIRE(workspace && workspace_bytes >= fw.bytes, "workspace too small: %zu < %zu", workspace_bytes, fw.bytes);
    if ((rc = fused_msgpass_bwd(L, src, ssum, fv.wt, p_drop, fv.m1, fv.m2, fv.mx, dz, ldz, fw.tpD1, fw.tpD2, fw.ones, fw.rupart, st, fv.tpX, fv.tpY1)))
      return rc;
    return fused_dw(L, P, fv.tpX, fv.tpY1, fw.tpD1, fw.tpD2, fw.ones, fw.part, fw.rupart, dW1, db1, dW2, db2, dR_u, st);
  }
  const int32_t* tp = token_plan();
  MsgWs w = carve(s, workspace);
  RD_REQUIRE(workspace && workspace_bytes >= w.bytes, "workspace too small: %zu < %zu",
             workspace_bytes, w.bytes);
  {
  {
    const long per = (long)F * K;
    int gy = (int)((per + 255) / 256); if (gy > 64) gy = 64;
    hipLaunchKernelGGL(k_msg_dz2, dim3(B, gy), dim3(256), 0, st, dz, z, ssum, w.dz2, B, T, F, d, (long)ldz,
                       tp ? tp + plan::brow_base(B, T) : nullptr, tp ? tp + plan::blen_base(B, T) : nullptr);
    if ((rc = check_launch("k_msg_dz2"))) return rc;
  }
  // dz1 = (dz2 W2) * ssum[f] * (y1 > 0)
  GemmArgs g{};
  g.M = M; g.N = K; g.K = K; g.nsplit = 1;
  g.A = w.dz2; g.sa_m = K; g.sa_k = 1;
  g.B = W2; g.sb_n = 1; g.sb_k = K;          // B(n=k_out, k=n_red) = W2[n_red*K + k_out]
  const bool tiles = precision() != RD_PREC_FP32;     // W2^T / W1^T tiles written by the forward
  if (tiles) { g.Btiles = sv.wt[2]; g.bt_ntile = sv.ntile; g.bt_nkc = sv.nkc; }
  g.C = w.dz1; g.sc_m = K;
  g.rowscale = ssum; g.rs_period = F; g.posmask = y1save; g.pm_m = K;
  if ((rc = launch_gemm(g, st))) return rc;
  // dx = dz1 W1
  GemmArgs h{};
  h.M = M; h.N = K; h.K = K; h.nsplit = 1;
  h.A = w.dz1; h.sa_m = K; h.sa_k = 1;
  h.B = W1; h.sb_n = 1; h.sb_k = K;
  if (tiles) { h.Btiles = sv.wt[3]; h.bt_ntile = sv.ntile; h.bt_nkc = sv.nkc; }
  h.C = w.dx; h.sc_m = K;
  if ((rc = launch_gemm(h, st))) return rc;
  hipLaunchKernelGGL(k_obs_embed_bwd, dim3((unsigned)(((long)B * F + 3) / 4)), dim3(256), 0, st, w.dx, xsave, src, w.rupart, B, T, F, d,
                     1.0f / (1.0f - p_drop));
  if ((rc = check_launch("k_obs_embed_bwd"))) return rc;
  if ((rc = launch_colsum(w.rupart, B, F * d, F * d, dR_u, w.colsum, st))) return rc;
  }
  // weight gradients dW_l = dz_l^T in_l (+ bias gradients as row sums of dz_l^T), split over the B*F rows
  // Large shapes in the bf16 modes (round 4): one conversion pass over the four operands + the tile stream of rd_tile_wgrad.hip,
  // instead of the split-K GEMM pair that converts both fp32 operands once per 64 x 64 output tile (12.5 % of P12's step, 10 % of
  // SYN256's).  RD_K1_WGRAD_STREAM=0: the GEMMs (A/B; read per call).
  const char* ws_env = getenv("RD_K1_WGRAD_STREAM");
  if (w.tl[0] && precision() != RD_PREC_FP32 && tile_wgrad_ok(K, K) && !(ws_env && atoi(ws_env) == 0)) {
    const float* srcs[4] = {w.dz2, y1save, w.dz1, xsave};
    const long ld[4] = {K, K, K, K};
    const int cols[4] = {K, K, K, K};
    if ((rc = launch_rows_to_tiles(M, 4, srcs, ld, cols, w.tl, st))) return rc;
    TileWgradJob jobs[2] = {{w.tl[0], w.tl[1], w.part[0], dW2, db2, K, K, nullptr, 0, 0, 0, 0, 0},       // dz2^T y1
                            {w.tl[2], w.tl[3], w.part[1], dW1, db1, K, K, nullptr, 0, 0, 0, 0, 0}};      // dz1^T x
    return launch_tile_wgrad(M, 2, jobs, sv.ones, 0, nullptr, st, nullptr);
  }
  if ((rc = launch_wgrad2(M, K, K, w.dz2, y1save, dW2, db2, w.dz1, xsave, dW1, db1, w.splitk, st))) return rc;
  return RD_OK;
}

// ------------------------------------------------------------------------------------------------
// Building blocks of the paper-faithful sensor stage (Raindrop_v2(use_beta=True): code/models_rd.py:317-343 with the literal
// at :317 flipped).  There the two graph layers cannot be folded into one fused launch (layer 1 prunes a different edge set per
// sample), so the model composes: observation embedding -> lin_value / increase_dim (rd_linear_fwd) -> rd_graph_beta_fwd ->
// rd_edge_softmax_list_batched -> lin_value of layer 2 (rd_linear_fwd) -> rd_rows_to_tokens_fwd.
// ------------------------------------------------------------------------------------------------
extern "C" int rd_obs_embed_fwd(const rd_shape* s, const float* src, const float* R_u, float p_drop, uint64_t seed, float* X,
                                void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(src && R_u && X, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  const long per = (long)s->F * s->T * s->d_ob;
  int gy = (int)((per + 255) / 256); if (gy > 64) gy = 64;
  hipLaunchKernelGGL(k_obs_embed, dim3(s->B, gy), dim3(256), 0, (hipStream_t)stream, src, R_u, X, s->B, s->T, s->F, s->d_ob, p_drop,
                     seed, seed_cell());
  return check_launch("k_obs_embed");
}

extern "C" size_t rd_obs_embed_bwd_workspace_bytes(const rd_shape* s) {
  if (!s || s->B <= 0) return 256;
  return ((size_t)s->B * s->F * s->d_ob + (size_t)colsum_ws_floats(s->B, s->F * s->d_ob)) * sizeof(float) + 512;
}

extern "C" int rd_obs_embed_bwd(const rd_shape* s, const float* src, const float* X, const float* dX, float p_drop, float* dR_u,
                                void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  RD_REQUIRE(dR_u, "NULL gradient output");
  hipStream_t st = (hipStream_t)stream;
  const int B = s->B, T = s->T, F = s->F, d = s->d_ob;
  if (B == 0) { RD_HIP(hipMemsetAsync(dR_u, 0, sizeof(float) * F * d, st)); return RD_OK; }
  RD_REQUIRE(src && X && dX && workspace, "NULL tensor");
  RD_REQUIRE(workspace_bytes >= rd_obs_embed_bwd_workspace_bytes(s), "workspace too small");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  float* rupart = (float*)workspace;
  float* cws = rupart + align_up((size_t)B * F * d * sizeof(float), 256) / sizeof(float);
  hipLaunchKernelGGL(k_obs_embed_bwd, dim3((unsigned)(((long)B * F + 3) / 4)), dim3(256), 0, st, dX, X, src, rupart, B, T, F, d, 1.0f / (1.0f - p_drop));
  if ((rc = check_launch("k_obs_embed_bwd"))) return rc;
  return launch_colsum(rupart, B, F * d, F * d, dR_u, cws, st);
}

static int rows_tokens(const rd_shape* s, const float* Y, const float* rowscale, float* z, int32_t ldz, void* stream, int bwd) {
  int rc = check_shape(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(Y && z, "NULL tensor");
  RD_REQUIRE(ldz >= s->F * s->d_ob, "ldz (%d) < F*d_ob", ldz);
  const long per = (long)s->F * s->T * s->d_ob;
  int gy = (int)((per + 255) / 256); if (gy > 64) gy = 64;
  hipLaunchKernelGGL(k_rows_to_tokens, dim3(s->B, gy), dim3(256), 0, (hipStream_t)stream, Y, rowscale, z, s->B, s->T, s->F, s->d_ob,
                     (long)ldz, bwd);
  return check_launch("k_rows_to_tokens");
}
extern "C" int rd_rows_to_tokens_fwd(const rd_shape* s, const float* Y, const float* rowscale, float* z, int32_t ldz, void* stream) {
  return rows_tokens(s, Y, rowscale, z, ldz, stream, 0);
}
extern "C" int rd_rows_to_tokens_bwd(const rd_shape* s, const float* dz, int32_t ldz, const float* rowscale, float* dY, void* stream) {
  return rows_tokens(s, dY, rowscale, const_cast<float*>(dz), ldz, stream, 1);
}
