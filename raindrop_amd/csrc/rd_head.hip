// rd_head.hip -- the classifier head of Raindrop_v2 with its loss, forward AND backward, in two launches.
//
//   agg[b]   = sum_t r[t,b,:] (1 - mask[b,t]) / (lengths[b] + 1)                 code/models_rd.py:366-367,379
//   feat[b]  = [agg[b] | emb(static[b])]                                          code/models_rd.py:381-384 (static model)
//   logits   = mlp_static(feat) = W2 relu(W0 feat + b0) + b2                      code/models_rd.py:385 (mlp_static: :263-267)
//   loss     = mean_b CrossEntropy(logits[b], y[b])                               code/Raindrop.py:255,322
// and every gradient of `loss`: dW0, db0, dW2, db2, d emb.weight, d emb.bias and dr [T,B,D] (the masked mean's backward,
// the entry gradient of the encoder stack).
//
// Why: the head is 256 rows x 186 features -- about 50 MFLOP -- but as separate operators it was 15 launches of
// latency-bound kernels (three forward products, the loss, two input-gradient and three weight-gradient products
// with their split-K reduces, the masked mean and its backward): ~114 us of a 0.85 ms step at ~4.5 us of fixed cost per
// launch.  Here a workgroup owns one sample (RB = 1; the template keeps the count) end to end:
//   k_head_rows   masked mean -> emb -> W0 (+ReLU) -> W2 -> softmax/loss -> dlogits -> dhid -> dfeat -> dr.
//                 W0 is read ONCE per workgroup, coalesced, into registers: wave w holds rows j = w, w+16, ... with lane l
//                 owning columns l, l+64, ...; the forward product reduces over lanes (one wave sum per output), the
//                 input gradient reuses the same registers and reduces over the 16 waves through LDS.  fp32 FMA
//                 arithmetic (at these sizes the matrix cores would idle behind the launch latency anyway).
//   k_head_wgrad  the three weight/bias gradients as 16 x 16 output tiles over the B rows (operands staged in LDS), and
//                 the mean of the per-sample losses -- all fixed-order sums, no atomics.
#include "rd_common.h"
#include "rd_plan.h"
#include "rd_trailing.h"

namespace rd {
namespace {

constexpr int HR_THR = 1024, HR_WAVES = 16;
constexpr int HR_RJ = 16, HR_KI = 4;               // most W0 rows per wave / 64-column slots per lane (dh <= 256); k_head_rows<RB, NRJ, NKI> fetches what dh needs
constexpr int HR_LD = 256;                         // row stride of the per-sample vectors in LDS
constexpr int HR_EMBW = 1024, HR_DS = 64;         // static embedding held in LDS when Fe * d_static <= 1024 and d_static <= 64 (else read in place)

struct HeadArgs {
  const float* r; const uint8_t* mask; const int64_t* lengths; const float* stat;
  const float *emb_w, *emb_b, *w0, *b0, *w2, *b2;
  const int64_t* y;
  float *logits, *dr;
  float *feat, *hid, *dhid, *demb, *dlog, *lossr;   // workspace: [B,dh] x3, [B,Fe], [B,C], [B]
  int T, B, D, ds, Fe, dh, C;
  const float* dlog_in;            // mode 2: d loss / d logits [B,C] from the caller (an autograd backward) instead of the cross entropy
  int mode;                        // 0: forward + loss + backward (rd_head_train); 1: forward only, logits out (rd_head_forward);
                                   // 2: forward recomputed + backward from dlog_in (rd_head_backward)
  const int32_t* plan;             // token plan (rd_plan.h) or null: r / dr hold the live rows only, sample b at rows off[rank[b]] + t
  unsigned long long* stamps;      // debug (tools/head_timing.py): clock64 per phase, thread 0 of workgroup 0
  int touch_bytes;                 // > 0: bytes of this kernel's own code requested into L2 at its start (rd_common.h touch_own_code)
};
static unsigned long long* g_head_stamps = nullptr;
#define HSTAMP(i)                                                                           \
  do {                                                                                      \
    if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[(i)] = clock64();         \
  } while (0)

// first row and row step of sample b's time steps in r / dr, and how many of them are live
struct HeadRows { long row0, rstep; int Tv; };
__device__ __forceinline__ HeadRows head_rows(const HeadArgs& a, int b) {
  HeadRows h;
  if (a.plan) {
    h.row0 = a.plan[plan::brow_base(a.B, a.T) + b]; h.rstep = 1; h.Tv = a.plan[plan::blen_base(a.B, a.T) + b];
  } else { h.row0 = b; h.rstep = a.B; h.Tv = a.T; }
  return h;
}

// the same for a workgroup-uniform sample, through the scalar cache, from the plan's by-sample arrays (brow / blen: ONE round trip;
// off[rank[b]] was two dependent ones)
__device__ __forceinline__ HeadRows head_rows_uniform(const HeadArgs& a, int b) {
  HeadRows h;
  if (a.plan) {
    int row0, tv;
    load_uniform_2xi32(a.plan + plan::brow_base(a.B, a.T) + b, a.plan + plan::blen_base(a.B, a.T) + b, row0, tv);
    h.row0 = row0; h.rstep = 1; h.Tv = tv;
  } else { h.row0 = b; h.rstep = a.B; h.Tv = a.T; }
  return h;
}

__device__ __forceinline__ float wsum64(float v) { return wave_sum64_dpp(v); }

// NRJ / NKI: W0 rows per wave and 64-column slots per lane that are actually FETCHED (16 NRJ >= dh, 64 NKI >= dh).  The generic
// <16, 4> form requests 256 x 256 clamped elements whatever dh is -- at P19 (dh = 186) 262 KB of address-unit traffic for a 138-KB
// matrix, and that request phase was the first 5.7 k of the kernel's 30 k cycles (tools/head_timing.py); <12, 3> requests 192 x 192.
template <int RB, int NRJ, int NKI>
__global__ __launch_bounds__(HR_THR) void k_head_rows(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float feat[RB][HR_LD], hid[RB][HR_LD], dhid[RB][HR_LD], dfeat[RB][HR_LD];
  __shared__ __attribute__((aligned(16))) float red[HR_WAVES * RB * HR_LD];     // mean-phase and wave partials
  __shared__ float lg[RB][16], dl[RB][16], invl[RB], b0s[HR_LD];
  // every small operand of the later phases, fetched ONCE at the top beside W0: as global loads inside their phases (static
  // embedding: d_static dependent steps of two loads; logits: three; the label; W2 again in the gradient) they were ~17 dependent
  // round trips during which the workgroup's other waves sat at the next barrier (SQ counters: 80 % of the wave cycles waiting)
  __shared__ float w2s[16 * HR_LD], embb[HR_LD], b2s[16], embws[HR_EMBW], stats[RB][HR_DS];
  __shared__ long long ys[RB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * RB;
  const int T = a.T, B = a.B, D = a.D, dh = a.dh, C = a.C, D4 = D >> 2;

  HSTAMP(0);
  RD_TOUCH_CODE(a.touch_bytes);
  // Request order = the order the data is NEEDED (vector memory returns in order): small operands (they go out while the plan
  // look-up is on its way through the scalar cache), the sample's rows for the masked mean, then W0, which nothing reads before
  // the mean is reduced.  Round 3 had W0 first: the mean then waited for all of W0 (10.4 k cycles into the kernel).
  // ---- small operands: ALL requests first (unconditional, clamped indices), stored to LDS below.  Written as `if (tid < n) lds[tid] =
  // g[tid];` per array each was its own load -> wait -> store block: seven dependent round trips at the top of the kernel.
  const bool emb_lds = a.Fe > 0 && a.Fe * a.ds <= HR_EMBW && a.ds <= HR_DS;
  const int nw2 = C * dh;
  const float pb0 = a.b0[min(tid, dh - 1)], pw2 = a.w2[min(tid, nw2 - 1)], pb2 = a.b2[min(tid, C - 1)];
  const long long py = a.mode == 0 ? a.y[min(b0 + min(tid, RB - 1), B - 1)] : 0;   // uniform
  float pew = 0.f, peb = 0.f, pst = 0.f;
  if (emb_lds) {                                     // uniform
    const int ne = a.Fe * a.ds, ns = RB * a.ds, ts = min(tid, ns - 1), rs = ts / a.ds;
    pew = a.emb_w[min(tid, ne - 1)]; peb = a.emb_b[min(tid, a.Fe - 1)];
    pst = a.stat[(long)min(b0 + rs, B - 1) * a.ds + (ts - rs * a.ds)];
  }
  // ---- masked mean: thread = (sample r, column quad c4, time group tg); four steps per pass requested together (clamped rows, the
  // step's validity applied to the value).  Threads beyond the ntg groups request a duplicate and add nothing.
  const int P = RB * D4;
  const int ntg = min(HR_THR / P, 16);
  const int pair = tid % P, tg = tid / P;
  const int mr = pair / D4, mc4 = pair - mr * D4, mb = b0 + mr;
  const bool mlive = tg < ntg && mb < B;
  // the sample's length (for 1 / (len + 1)) and its rows.  One sample per workgroup: three scalar loads behind ONE wait -- as a
  // vector load of a uniform address the compiler read the length back with v_readfirstlane at once, i.e. behind `vmcnt(0)`.
  long long plen; HeadRows hr;
  if (RB == 1) {
    const int bu = __builtin_amdgcn_readfirstlane(b0);                 // b0 < B: the grid is B workgroups
    uint64_t lv; int row0 = bu, tv = T;
    if (a.plan) load_uniform_u64_2xi32(reinterpret_cast<const uint64_t*>(a.lengths) + bu, a.plan + plan::brow_base(B, T) + bu,
                                       a.plan + plan::blen_base(B, T) + bu, lv, row0, tv);
    else lv = load_uniform_u64(reinterpret_cast<const uint64_t*>(a.lengths) + bu);
    plen = (long long)lv;
    hr.row0 = row0; hr.rstep = a.plan ? 1 : B; hr.Tv = tv;
  } else {
    plen = a.lengths[min(b0 + min(tid, RB - 1), B - 1)];
    hr = head_rows(a, min(mb, B - 1));
  }
  float4 mv[4]; bool mok[4]; unsigned char mk[4] = {0, 0, 0, 0};
  auto request_rows = [&](int t0) {
    if (!a.plan) {                                                     // uniform: the padded layout's per-step mask bytes
#pragma unroll
      for (int u = 0; u < 4; ++u) mk[u] = a.mask[(long)min(mb, B - 1) * T + min(t0 + u * ntg, T - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * ntg, tc = max(min(t, hr.Tv - 1), 0);
      mok[u] = mlive && t < hr.Tv;
      const long row = mok[u] ? hr.row0 + (long)tc * hr.rstep : 0;     // row 0 of r always exists
      mv[u] = *reinterpret_cast<const float4*>(a.r + row * D + 4 * mc4);
    }
  };
  request_rows(min(tg, ntg - 1));
  __builtin_amdgcn_sched_barrier(0);                       // the rows' requests stay IN FRONT of W0's (the fourth was moved behind them)
  // ---- W0 rows of this wave -> registers.  UNCONDITIONAL loads from clamped addresses: rows j >= dh are never used and columns
  // k >= dh only meet zeros (hid) or reach accumulator slots nobody reads (dfeat).  As conditional loads (a phi of {0, value} each)
  // the compiler waited for them one group at a time -- two dozen dependent round trips at the head of a kernel whose 256
  // workgroups all run at once, so its duration IS one workgroup's latency chain.
  float w[NRJ][NKI];
#pragma unroll
  for (int jj = 0; jj < NRJ; ++jj) {
    const int jc = min(wave + HR_WAVES * jj, dh - 1);
#pragma unroll
    for (int i = 0; i < NKI; ++i) w[jj][i] = a.w0[(long)jc * dh + min(lane + 64 * i, dh - 1)];
  }
  HSTAMP(1);
  // ---- small operands -> LDS ----
  {
    if (tid < HR_LD) b0s[tid] = pb0;
    if (tid < nw2) w2s[tid] = pw2;
    if (tid < C) b2s[tid] = pb2;
    if (tid < RB) ys[tid] = py;
    if (emb_lds) {
      const int ne = a.Fe * a.ds;
      if (tid < ne) embws[tid] = pew;
      if (tid < a.Fe) embb[tid] = peb;
      if (tid < RB * a.ds) stats[tid / a.ds][tid % a.ds] = pst;
      for (int i = tid + HR_THR; i < ne; i += HR_THR) embws[i] = a.emb_w[i];          // HR_EMBW <= HR_THR: never runs; kept for safety
    }
    for (int i = tid + HR_THR; i < nw2; i += HR_THR) w2s[i] = a.w2[i];                // C * dh > 1024 only
    if (tid < RB) invl[tid] = (b0 + tid < B) ? 1.0f / (float)(plen + 1) : 0.f;
  }
  // ---- the rows: first pass (in flight since the top), further passes for T > 4 ntg steps ----
  {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // the first pass is NOT a loop iteration: inside one loop the waits are computed for the back edge (only that pass's four
    // requests outstanding = vmcnt(0) at the last use), which on entry would also wait for all of W0
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (mok[u] && !mk[u]) { s.x += mv[u].x; s.y += mv[u].y; s.z += mv[u].z; s.w += mv[u].w; }
    for (int pass = 1; pass * 4 * ntg < T; ++pass) {       // T > 4 ntg steps only (uniform trip count; the flags drop steps past the sample's length)
      request_rows(min(tg, ntg - 1) + pass * 4 * ntg);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (mok[u] && !mk[u]) { s.x += mv[u].x; s.y += mv[u].y; s.z += mv[u].z; s.w += mv[u].w; }
    }
    if (tg < ntg) *reinterpret_cast<float4*>(red + ((size_t)tg * P + pair) * 4) = s;
  }
  HSTAMP(2);
  lds_barrier();                                           // LDS only: W0 stays in flight (no thread reads global memory another wrote)
  HSTAMP(3);
  if (tid < P) {
    const int r = tid / D4, c4 = tid - r * D4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < ntg; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(red + ((size_t)g * P + tid) * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float il = invl[r];
    *reinterpret_cast<float4*>(&feat[r][4 * c4]) = make_float4(s.x * il, s.y * il, s.z * il, s.w * il);
  }
  // ---- static embedding into the right block (code/models_rd.py:381): thread = (r, j) ----
  for (int e = tid; e < RB * a.Fe; e += HR_THR) {         // (the prefetched operands were stored before the barrier above)
    const int r = e / a.Fe, j = e - r * a.Fe, b = b0 + r;
    float s;
    if (emb_lds) {
      s = embb[j];
      if (b < B)
        for (int q = 0; q < a.ds; ++q) s += stats[r][q] * embws[j * a.ds + q];
    } else {
      s = a.emb_b[j];
      if (b < B)
        for (int q = 0; q < a.ds; ++q) s += a.stat[(long)b * a.ds + q] * a.emb_w[(long)j * a.ds + q];
    }
    feat[r][D + j] = s;
  }
  lds_barrier();
  HSTAMP(4);
  // ---- hid = relu(W0 feat + b0): wave w owns outputs j = w, w+16, ...; lanes split the reduction.  Branch-free over the wave's
  // rows: behind `if (j < dh)` every row was its own basic block and the NRJ wave sums (six dependent DPP steps each) ran one after
  // the other -- 7.3 k cycles for 16 rows; as straight-line code the scheduler interleaves the independent chains. ----
  {
    float fv[RB][NKI];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int i = 0; i < NKI; ++i) {
        const int k = lane + 64 * i;
        const float f = feat[r][k];                        // k < HR_LD always; columns >= dh hold nothing
        fv[r][i] = k < dh ? f : 0.f;
      }
    float ps[NRJ][RB];
#pragma unroll
    for (int jj = 0; jj < NRJ; ++jj)
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        float p = 0.f;
#pragma unroll
        for (int i = 0; i < NKI; ++i) p += fv[r][i] * w[jj][i];
        ps[jj][r] = p;
      }
    // the wave sums step by step ACROSS the rows (rd_common.h wave_sum64_dpp, same order of additions per row): as NRJ calls in a
    // row the chains were emitted one after the other on one register
#define HR_DPP_STEP(ctrl, rmask)                                                                                              \
  _Pragma("unroll") for (int jj = 0; jj < NRJ; ++jj) _Pragma("unroll") for (int r = 0; r < RB; ++r)                           \
    ps[jj][r] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ps[jj][r]), ctrl, rmask, 0xf, false))
    HR_DPP_STEP(0x111, 0xf); HR_DPP_STEP(0x112, 0xf); HR_DPP_STEP(0x114, 0xf); HR_DPP_STEP(0x118, 0xf);
    HR_DPP_STEP(0x142, 0xa); HR_DPP_STEP(0x143, 0xc);
#undef HR_DPP_STEP
#pragma unroll
    for (int jj = 0; jj < NRJ; ++jj)
#pragma unroll
      for (int r = 0; r < RB; ++r) ps[jj][r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ps[jj][r]), 63));
#pragma unroll
    for (int jj = 0; jj < NRJ; ++jj) {
      const int j = wave + HR_WAVES * jj;
      if (lane == 0 && j < dh) {
#pragma unroll
        for (int r = 0; r < RB; ++r) hid[r][j] = fmaxf(ps[jj][r] + b0s[j], 0.f);
      }
    }
  }
  lds_barrier();
  HSTAMP(5);
  // ---- logits: wave = (r, c) ----
  for (int o = wave; o < RB * C; o += HR_WAVES) {
    const int r = o / C, c = o - r * C;
    float p = 0.f;
    for (int k = lane; k < dh; k += 64) p += hid[r][k] * w2s[c * dh + k];
    p = wsum64(p);
    if (lane == 0) lg[r][c] = p + b2s[c];
  }
  lds_barrier();
  HSTAMP(6);
  if (a.mode == 1) {                                   // rd_head_forward: the logits are the result
    if (tid < RB * C && b0 + tid / C < B) a.logits[(long)(b0 + tid / C) * C + tid % C] = lg[tid / C][tid % C];
    return;
  }
  // ---- softmax cross entropy per sample, dlogits = (softmax - onehot) / B ----
  if (a.mode == 2) {                                   // rd_head_backward: d loss / d logits comes from the caller
    if (tid < RB * C) {
      const int r = tid / C, c = tid - r * C, b = b0 + r;
      const float d = b < B ? a.dlog_in[(long)b * C + c] : 0.f;
      dl[r][c] = d;
      if (b < B) a.dlog[(long)b * C + c] = d;
    }
  } else if (tid < RB && b0 + tid < B) {
    const int r = tid, b = b0 + r;
    float m = lg[r][0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, lg[r][c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(lg[r][c] - m);
    const float lse = m + logf(se);
    // a label outside [0, C) has no logit to index (torch raises): the sample's loss becomes NaN -- loud, and no out-of-bounds read
    const long yb = (long)ys[r];
    const bool yok = yb >= 0 && yb < C;
    const int t = yok ? (int)yb : 0;
    a.lossr[b] = yok ? lse - lg[r][t] : __builtin_nanf("");
    const float invB = 1.0f / (float)B;
    for (int c = 0; c < C; ++c) {
      const float d = (expf(lg[r][c] - lse) - (c == t ? 1.f : 0.f)) * invB;
      dl[r][c] = d;
      a.dlog[(long)b * C + c] = d;
      a.logits[(long)b * C + c] = lg[r][c];
    }
  } else if (tid < RB) {
    for (int c = 0; c < C; ++c) dl[tid][c] = 0.f;
  }
  lds_barrier();
  HSTAMP(7);
  // ---- dhid = (dlogits W2) gated by hid > 0: thread = (r, j); rows out to the workspace for the weight gradients ----
  for (int e = tid; e < RB * dh; e += HR_THR) {
    const int r = e / dh, j = e - r * dh, b = b0 + r;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dl[r][c] * w2s[c * dh + j];
    const float h = hid[r][j];
    s = h > 0.f ? s : 0.f;
    dhid[r][j] = s;
    if (b < B) {
      a.dhid[(long)b * dh + j] = s;
      a.hid[(long)b * dh + j] = h;
      a.feat[(long)b * dh + j] = feat[r][j];
    }
  }
  lds_barrier();
  HSTAMP(8);
  // ---- dfeat = dhid W0: the same registers; this wave's rows give a partial for every column, waves combined in order ----
  {
    float acc[RB][NKI];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int i = 0; i < NKI; ++i) acc[r][i] = 0.f;
#pragma unroll
    for (int jj = 0; jj < NRJ; ++jj) {
      const int j = wave + HR_WAVES * jj;                  // < HR_LD; rows >= dh contribute zero (their W0 registers hold a clamped duplicate)
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float dv = dhid[r][j];
        const float d = j < dh ? dv : 0.f;
#pragma unroll
        for (int i = 0; i < NKI; ++i) acc[r][i] += d * w[jj][i];
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int i = 0; i < NKI; ++i) red[((size_t)wave * RB + r) * HR_LD + lane + 64 * i] = acc[r][i];
  }
  lds_barrier();
  HSTAMP(9);
  for (int e = tid; e < RB * HR_LD; e += HR_THR) {
    const int r = e / HR_LD, k = e - r * HR_LD;
    float s = 0.f;
    if (k < 64 * NKI) {                                    // columns beyond the fetched slots were never written (and are >= dh)
#pragma unroll
      for (int q = 0; q < HR_WAVES; ++q) s += red[((size_t)q * RB + r) * HR_LD + k];
    }
    dfeat[r][k] = s;
    const int b = b0 + r;
    if (k >= D && k < dh && b < B) a.demb[(long)b * a.Fe + (k - D)] = s;   // gradient of the static embedding's output
  }
  lds_barrier();
  HSTAMP(10);
  // ---- masked mean backward: dr[t,b,:] = valid ? dfeat[b,:D] / (len + 1) : 0   (code/models_rd.py:379, autograd) ----
  for (int e = tid; e < RB * T * D4; e += HR_THR) {
    const int c4 = e % D4, rt = e / D4;
    const int t = rt % T, r = rt / T, b = b0 + r;
    if (b >= B) continue;
    const HeadRows hq = RB == 1 ? hr : head_rows(a, b);                  // RB == 1: the look-up of the top of the kernel
    if (t >= hq.Tv) continue;                                           // token plan: padded steps have no row
    const float il = (!a.plan && a.mask[(long)b * T + t]) ? 0.f : invl[r];
    const float4 v = *reinterpret_cast<const float4*>(&dfeat[r][4 * c4]);
    *reinterpret_cast<float4*>(a.dr + (hq.row0 + (long)t * hq.rstep) * D + 4 * c4) = make_float4(v.x * il, v.y * il, v.z * il, v.w * il);
  }
  HSTAMP(11);
}

// dW[n,k] = sum_b u[b,n] v[b,k], db[n] = sum_b u[b,n]: 16 x 16 output tile per workgroup, thread = one output (rd_trailing.h)
__global__ __launch_bounds__(256) void k_head_wgrad(HwArgs a) {
  __shared__ float us[256][16], vs[256][17];
  head_wgrad_body(a, (int)blockIdx.x, (int)threadIdx.x, us, vs);
}

}  // namespace
}  // namespace rd

using namespace rd;

namespace rd {
int launch_head_wgrad_standalone(const HwArgs& h, hipStream_t st) {
  hipLaunchKernelGGL(k_head_wgrad, dim3(h.ntiles), dim3(256), 0, st, h);
  return check_launch("k_head_wgrad");
}
}  // namespace rd

extern "C" void rd_debug_set_head_stamps(void* p) { g_head_stamps = (unsigned long long*)p; }   // not part of the ABI

extern "C" size_t rd_head_train_workspace_bytes(int32_t B, int32_t dh, int32_t C) {
  if (B < 0 || dh <= 0 || C <= 0) return 0;
  return align_up(((size_t)4 * B * dh + (size_t)B * C + B) * sizeof(float), 256);
}

extern "C" int rd_head_train_supported(int32_t D, int32_t Fe, int32_t C) {
  const char* e = getenv("RD_HEAD_FUSED");
  if (e && atoi(e) == 0) return 0;
  return (D % 4) == 0 && D > 0 && Fe >= 0 && D + Fe <= 256 && C >= 1 && C <= 16 && 2 * (D / 4) <= HR_THR;
}

static int head_launch(int mode, const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                       const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w, const float* emb_b,
                       const float* w0, const float* b0, const float* w2, const float* b2, const int64_t* y, const float* dlog_in,
                       float* loss, float* logits, float* g_emb_w, float* g_emb_b, float* g_w0, float* g_b0, float* g_w2,
                       float* g_b2, float* dr, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int rd_head_train(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                             const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w,
                             const float* emb_b, const float* w0, const float* b0, const float* w2, const float* b2,
                             const int64_t* y, float* loss, float* logits, float* g_emb_w, float* g_emb_b, float* g_w0,
                             float* g_b0, float* g_w2, float* g_b2, float* dr, void* workspace, size_t workspace_bytes,
                             void* stream) {
  RD_REQUIRE(y && loss && logits && g_w0 && g_b0 && g_w2 && g_b2 && dr, "NULL tensor");
  return head_launch(0, s, D, d_static, Fe, C, r, mask, lengths, stat, emb_w, emb_b, w0, b0, w2, b2, y, nullptr, loss, logits, g_emb_w,
                     g_emb_b, g_w0, g_b0, g_w2, g_b2, dr, workspace, workspace_bytes, stream);
}

// The same head as two calls around a loss the CALLER evaluates (the reference's training loop: `criterion(outputs, y)` in
// code/Raindrop.py:319-322, then loss.backward()): rd_head_forward stops at the logits; rd_head_backward recomputes the forward
// phases (a masked mean and two small products: cheaper than saving and re-reading them) and continues from the caller's
// d loss / d logits.  Same kernel, same arithmetic, same workspace as rd_head_train.
extern "C" int rd_head_forward(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                               const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w,
                               const float* emb_b, const float* w0, const float* b0, const float* w2, const float* b2,
                               float* logits, void* workspace, size_t workspace_bytes, void* stream) {
  RD_REQUIRE(logits, "NULL tensor");
  return head_launch(1, s, D, d_static, Fe, C, r, mask, lengths, stat, emb_w, emb_b, w0, b0, w2, b2, nullptr, nullptr, nullptr, logits,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}
extern "C" int rd_head_backward(const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                                const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w,
                                const float* emb_b, const float* w0, const float* b0, const float* w2, const float* b2,
                                const float* dlogits, float* g_emb_w, float* g_emb_b, float* g_w0, float* g_b0, float* g_w2,
                                float* g_b2, float* dr, void* workspace, size_t workspace_bytes, void* stream) {
  RD_REQUIRE(dlogits && g_w0 && g_b0 && g_w2 && g_b2 && dr, "NULL tensor");
  return head_launch(2, s, D, d_static, Fe, C, r, mask, lengths, stat, emb_w, emb_b, w0, b0, w2, b2, nullptr, dlogits, nullptr, nullptr,
                     g_emb_w, g_emb_b, g_w0, g_b0, g_w2, g_b2, dr, workspace, workspace_bytes, stream);
}

static int head_launch(int mode, const rd_shape* s, int32_t D, int32_t d_static, int32_t Fe, int32_t C, const float* r,
                       const uint8_t* mask, const int64_t* lengths, const float* stat, const float* emb_w, const float* emb_b,
                       const float* w0, const float* b0, const float* w2, const float* b2, const int64_t* y, const float* dlog_in,
                       float* loss, float* logits, float* g_emb_w, float* g_emb_b, float* g_w0, float* g_b0, float* g_w2,
                       float* g_b2, float* dr, void* workspace, size_t workspace_bytes, void* stream) {
  RD_REQUIRE(s && s->T > 0 && s->B > 0, "bad shape");
  RD_REQUIRE(rd_head_train_supported(D, Fe, C), "head_train: unsupported sizes D=%d Fe=%d C=%d", D, Fe, C);
  RD_REQUIRE(r && mask && lengths && w0 && b0 && w2 && b2 && workspace, "NULL tensor");
  RD_REQUIRE(Fe == 0 || (stat && emb_w && emb_b && d_static > 0 && (mode == 1 || (g_emb_w && g_emb_b))), "static embedding tensors missing");
  const int B = s->B, dh = D + Fe;
  RD_REQUIRE(workspace_bytes >= rd_head_train_workspace_bytes(B, dh, C), "workspace too small");
  RD_REQUIRE((reinterpret_cast<uintptr_t>(r) & 15) == 0 && (reinterpret_cast<uintptr_t>(dr) & 15) == 0, "r / dr must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  HeadArgs a{};
  a.r = r; a.mask = mask; a.lengths = lengths; a.stat = stat; a.emb_w = emb_w; a.emb_b = emb_b; a.w0 = w0; a.b0 = b0; a.w2 = w2;
  a.b2 = b2; a.y = y; a.logits = logits; a.dr = dr; a.mode = mode; a.dlog_in = dlog_in;
  float* ws = (float*)workspace;
  a.feat = ws; a.hid = a.feat + (size_t)B * dh; a.dhid = a.hid + (size_t)B * dh; a.demb = a.dhid + (size_t)B * dh;
  a.dlog = a.demb + (size_t)B * dh; a.lossr = a.dlog + (size_t)B * C;
  a.T = s->T; a.B = B; a.D = D; a.ds = d_static; a.Fe = Fe; a.dh = dh; a.C = C;
  a.plan = token_plan(); a.stamps = g_head_stamps;
  // own-code touch: 14.5 KB of the 15.4 KB <1, 12, 3> kernel, 16 KB of the 16.9 KB generic one (tests/test_kernel_resources.py keeps the
  // constants below the code lengths); RD_CODE_TOUCH=0 switches it off (A/B)
  static const int touch = [] { const char* e = getenv("RD_CODE_TOUCH"); return e ? atoi(e) : 1; }();
  // one sample per workgroup (B workgroups: every CU busy at B = 256); two samples per workgroup measured inside the noise twice
  // (rounds 2 and 4) and is no longer built (its hid phase spilled at 128 registers).
  // W0 fetched as 192 x 192 where that covers it (P19: dh = 186), else 256 x 256 (RD_HEAD_NARROW=0: always the latter)
  static const int narrow = [] { const char* e = getenv("RD_HEAD_NARROW"); return e ? atoi(e) : 1; }();
  if (narrow && dh <= 192) { a.touch_bytes = touch ? RD_TL_HEAD_P19 : 0; hipLaunchKernelGGL((k_head_rows<1, 12, 3>), dim3(B), dim3(HR_THR), 0, st, a); }
  else { a.touch_bytes = touch ? RD_TL_HEAD : 0; hipLaunchKernelGGL((k_head_rows<1, HR_RJ, HR_KI>), dim3(B), dim3(HR_THR), 0, st, a); }
  int rc = check_launch("k_head_rows");
  if (rc || mode == 1) return rc;
  HwArgs h{};
  h.B = B; h.lossr = a.lossr; h.loss = loss;             // loss == null (rd_head_backward): no loss mean
  int blk = 0, n = 0;
  auto add = [&](const float* u, int ldu, int N, const float* v, int ldv, int K, float* dW, float* db) {
    HwJob& J = h.j[n++];
    J.u = u; J.ldu = ldu; J.N = N; J.v = v; J.ldv = ldv; J.K = K; J.dW = dW; J.db = db;
    J.tiles_k = cdiv(K, 16); J.blk0 = blk; blk += cdiv(N, 16) * J.tiles_k;
  };
  add(a.dhid, dh, dh, a.feat, dh, dh, g_w0, g_b0);                       // d mlp_static[0]
  add(a.dlog, C, C, a.hid, dh, dh, g_w2, g_b2);                          // d mlp_static[2]
  if (Fe) add(a.demb, Fe, Fe, stat, d_static, d_static, g_emb_w, g_emb_b);   // d emb
  h.n = n; h.ntiles = blk;
  // weight gradients + the loss mean: nothing in the backward chain reads them.  Deferred mode: parked for the next backward chain
  // launch's idle workgroups (rd_trailing.h: four tiles per 1024-thread block); else side branch if registered, else here.
  if (trailing_deferred()) {
    RiderArgs r{};
    r.kind = RIDER_HEAD; r.nblocks = cdiv(blk, 4); r.hw = h;
    return trailing_park(r, st);
  }
  hipLaunchKernelGGL(k_head_wgrad, dim3(blk), dim3(256), 0, side_fork(st), h);
  return check_launch("k_head_wgrad");
}
