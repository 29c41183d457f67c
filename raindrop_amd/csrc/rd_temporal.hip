// rd_temporal.hip -- temporal self-attention stage (kernel families K2/K3/K5):
//   * masked multi-head attention forward/backward (flash-style: scores never leave the CU),
//   * residual-add + dropout + LayerNorm forward/backward,
//   * masked mean over time forward/backward,
//   * rd_encoder_layer_{fwd,bwd}: one post-norm nn.TransformerEncoderLayer as a chain of launches.
//
// Replaces nn.TransformerEncoder (code/models_rd.py:235-237,358; torch semantics in
// torch/nn/modules/transformer.py:799-983 and F.multi_head_attention_forward): post-norm,
// ReLU FFN, key-padding mask -> -inf, q.k scaled by 1/sqrt(head_dim), LayerNorm eps 1e-5,
// dropout on attention probabilities / attention output / FFN hidden / FFN output; and the
// masked mean of code/models_rd.py:366-367,379.
//
// All contractions use v_mfma_f32_16x16x4_f32 (exact fp32).  Sequence tiles are 64 steps; a
// P19 sample (T=60) is a single tile, P12 (215) / PAM (600) loop over key tiles with an online
// softmax, so the [T,T] score matrix only ever exists in LDS.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "rd_common.h"
#include "rd_plan.h"
#include "rd_rng.h"

namespace rd {
// row-block dense layer on pre-split weight planes (rd_rowgemm.hip)
bool rowgemm_ok(int N, int K, long lda, long ldc);
size_t rowgemm_plane_elems(int rows, int cols);
int launch_wsplit(int njobs, const float* const* W, const int* N, const int* K, const int* transpose, __bf16* const* hi,
                  __bf16* const* lo, void* ones, hipStream_t st);
// weight gradients from exported row tiles (rd_tile_wgrad.hip)
// s32 / S: this job's own chunk count (device / bound) when its row tiles live in another chunk space than the launch's; hd != 0: the
// A tiles' columns are in the fused attention's head-padded layout ((which, head) blocks of hdp columns, hd of them real)
struct TileWgradJob { const void *tA, *tB; float* part; float *dW, *db; int N, K; const int32_t* s32; int S; int hd, hdp, H, D; };
size_t tile_elems(long M, int cols);
size_t tile_wgrad_ones_elems();
size_t tile_wgrad_part_floats(int N, int K);
bool tile_wgrad_ok(int N, int K);
struct TileColsumJob { const float* x; int M, N, n1; float *out1, *out2; };
int launch_rows_to_tiles(long M, int n, const float* const* x, const long* ld, const int* cols, void* const* tiles, hipStream_t st);   // rd_tiles_export.hip
int launch_tile_wgrad(long M, int njobs, const TileWgradJob* jobs, const void* ones, int ncs, const TileColsumJob* cs,
                      hipStream_t st, const int32_t* s32);
void rowgemm_export_next(void* tiles);
void rowgemm_set_mlive(const int32_t* p);
// row-local chains of the layer as one launch per direction (rd_encfuse.hip)
bool attnfuse_ok(int T, int D, int H, int hd);
size_t attnfuse_wf_elems(int H);
size_t attnfuse_wb_elems(int H);
int attnfuse_split_specs(const float* in_proj_w, int D, int H, int hd, void* wf, void* wb, WsplitSpec* out);
int attnfuse_padded_cols(int H);
int attnfuse_nth();
int launch_attn_fused_fwd(const float* x, const void* wf, const float* bias, const int32_t* plan, int T, int B, int D, int H, int hd,
                          float p_drop, uint64_t seed, uint32_t site, float* out, float* lse, hipStream_t st);
int launch_attn_fused_bwd(const float* x, const void* wf, const void* wb, const float* bias, const int32_t* plan, int T, int B, int D, int H,
                          int hd, float p_drop, uint64_t seed, uint32_t site, const float* out, const float* lse, const float* dout,
                          const float* ds1, float* dx, void* xt, void* dt, hipStream_t st);
bool encfuse_ok(int D, int H);
int launch_enc_post_fwd(long M, int D, int H, const float* attn, const float* x, const void* Wo, const void* W1, const void* W2,
                        const float* bo, const float* b1, const float* b2, const float* g1, const float* be1, const float* g2,
                        const float* be2, float* s1, float* x1, float* st1, float* h, float* s2, float* y, float* st2,
                        void* xt_attn, void* xt_x1, void* xt_h, float p, uint64_t seed, uint32_t site_ao, uint32_t site_fh,
                        uint32_t site_fo, const int32_t* mlive, void* hgate, hipStream_t st);
int encfuse_part_rows(long M);
int launch_enc_pre_bwd(long M, int D, int H, const float* dy, const float* s2, const float* st2, const float* g2, const float* h,
                       const float* s1, const float* st1, const float* g1, const void* W2t, const void* W1t, const void* Wot,
                       float* ds2, float* ds1, float* da, float* part2, float* part1, void* xt_df, void* xt_du, void* xt_dout, float p,
                       uint64_t seed, uint32_t site_fo, uint32_t site_ao, const int32_t* mlive, const void* hgate, hipStream_t st);
bool rowgemm_lnb_ok(int N, int K);
int rowgemm_lnb_part_rows(long M);
int launch_rowgemm_lnb(long M, int N, int K, const float* dy, const float* s, const float* stats, const float* g, float* ds_out,
                       float* part, float p_drop, uint64_t seed, uint32_t site, const void* Wh, float* C, long ldc,
                       const float* posmask, long pm_ld, float cscale, hipStream_t st);
bool rowgemm_ln_ok(int N, int K);
int launch_rowgemm_ln(long M, int N, int K, const float* A, const void* Wh, const float* bias, const float* residual,
                      const float* ln_g, const float* ln_b, float* s_out, float* y, float* stats, float drop_p,
                      uint64_t drop_seed, uint32_t drop_site, hipStream_t st);
int launch_rowgemm(long M, int N, int K, const float* A, long lda, const void* Wh, const void* Wl, float* C, long ldc,
                   const float* bias, int relu, const float* posmask, long pm_ld, float cscale, const float* residual,
                   long res_ld, float drop_p, uint64_t drop_seed, uint32_t drop_site, hipStream_t st);

namespace {

constexpr int TS = 64;          // sequence tile (queries and keys)
constexpr int LDP = TS + 4;     // row stride of the 64x64 score / probability tiles in LDS

// acc[j] (16x16 tiles, j < NT) += A[16 x KK] * B[KK x 16*NT]
// A(i,k) at a[i*a_si + k*a_sk], B(k,n) at b[k*b_sk + n*b_sn]; all LDS.  MFMA lane map: lane l
// supplies A(i = l&15, k = k0 + (l>>4)) and B(k = k0 + (l>>4), n = l&15).
template <int NT>
__device__ __forceinline__ void mma_f32(f32x4 (&acc)[NT], const float* a, int a_si, int a_sk,
                                        const float* b, int b_sk, int b_sn, int KK, int lane) {
  const float* ap = a + (lane & 15) * a_si + (lane >> 4) * a_sk;
  const float* bp = b + (lane >> 4) * b_sk + (lane & 15) * b_sn;
  for (int k0 = 0; k0 < KK; k0 += 4) {
    const float av = ap[k0 * a_sk];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[k0 * b_sk + 16 * j * b_sn], acc[j], 0, 0, 0);
  }
}

__device__ __forceinline__ float wave_sum64(float v) { return wave_sum64_dpp(v); }   // rd_common.h: same order as the row-block kernels
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// max / sum over each 16-lane row of the wavefront, every lane of the row gets the result: four rotate-within-row DPP steps
// (row_ror:8, 4, 2, 1) instead of four ds_bpermute round trips
#define RD_ROW_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float group16_max(float v) {
  v = fmaxf(v, RD_ROW_ROR(v, 8)); v = fmaxf(v, RD_ROW_ROR(v, 4)); v = fmaxf(v, RD_ROW_ROR(v, 2)); v = fmaxf(v, RD_ROW_ROR(v, 1));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
  v += RD_ROW_ROR(v, 8); v += RD_ROW_ROR(v, 4); v += RD_ROW_ROR(v, 2); v += RD_ROW_ROR(v, 1);
  return v;
}

struct AttnArgs {
  const float* qkv;      // [T,B,3D]
  const uint8_t* mask;   // [B,T], 1 = padded key
  float* out;            // fwd: attention output [T,B,D]; bwd: same tensor (read)
  float* lse;            // [B,H,T] log-sum-exp of the scaled, masked scores
  const float* dout;     // bwd: grad of out [T,B,D]
  float* dqkv;           // bwd: [T,B,3D]
  float* delta;          // bwd: [B,H,T] rowsum(dout * out)
  int T, B, D, H, hd;
  float scale, p_drop; uint64_t seed; uint32_t site; const uint64_t* seed_cell;
  unsigned long long* stamps;   // debug only (tools/attn_timing.py)
  const int32_t* plan;          // token plan (rd_plan.h) or null; honoured by the single-tile split-bf16 kernels (k_attn_*_one_b16w)
};
static unsigned long long* g_attn_stamps = nullptr;

// Where the rows of one sample live.  Padded layout: [T,B,*] tensors, row of step t = t*B + b, key validity from the mask.
// Token plan: workgroup index b is a RANK, its rows are the contiguous block off[b] .. off[b] + len[b]; steps >= len do
// not exist (as keys they are what the mask would have removed; as queries nothing reads them).
struct AttnRows { long row0, rstep; int Tv; };
__device__ __forceinline__ AttnRows attn_rows(const AttnArgs& a, int b) {
  AttnRows r;
  if (a.plan) {
    r.row0 = __builtin_amdgcn_readfirstlane(a.plan[plan::off_base() + b]); r.rstep = 1;
    r.Tv = __builtin_amdgcn_readfirstlane(a.plan[plan::len_base(a.B) + b]);
  } else { r.row0 = b; r.rstep = a.B; r.Tv = a.T; }
  return r;
}
#define ASTAMP(i)                                                                                         \
  do {                                                                                                    \
    if (a.stamps && blockIdx.x < 8 && threadIdx.x == 0) a.stamps[blockIdx.x * 16 + (i)] = clock64();      \
  } while (0)

// Dropout mask of the attention probabilities: element (bh, q, key) is component (q & 3) of the Philox
// quad (bh*T + key)*ceil(T/4) + (q >> 2): the four query rows a lane holds in an MFMA accumulator
// (q = q4*4 + r) come from ONE Philox evaluation.
__device__ __forceinline__ uint64_t attn_quad(int bh, int T, int q, int key) {
  return ((uint64_t)bh * T + key) * ((T + 3) >> 2) + (q >> 2);
}
__device__ __forceinline__ float attn_keep1(uint64_t seed, uint32_t site, int bh, int T, int q, int key, float p,
                                            float inv_keep) {
  const float4 u = uniform4(seed, site, attn_quad(bh, T, q, key));
  const int j = q & 3;
  const float v = j == 0 ? u.x : (j == 1 ? u.y : (j == 2 ? u.z : u.w));
  return v >= p ? inv_keep : 0.f;
}
// keep-scales of the 4 consecutive query rows q4*4 .. q4*4+3 for one key
__device__ __forceinline__ void attn_keep4(float (&k4)[4], uint64_t seed, uint32_t site, int bh, int T, int q0,
                                           int key, float p, float inv_keep) {
  const float4 u = uniform4(seed, site, attn_quad(bh, T, q0, key));
  k4[0] = u.x >= p ? inv_keep : 0.f; k4[1] = u.y >= p ? inv_keep : 0.f;
  k4[2] = u.z >= p ? inv_keep : 0.f; k4[3] = u.w >= p ? inv_keep : 0.f;
}

// rows [t0, t0+64) x cols [0, hd) of one head of q / k / v / out / dout: 4 threads per row, thread (r, q) holds
// the 16-byte chunks q, q+4, q+8, ... of row r (zero padded to 16*NTH columns).  Loading is split from the LDS
// store so that a kernel can request ALL of its tiles before it consumes the first one: the loop form
// (load, wait, store per chunk) cost one dependent memory round trip per chunk -- 20 to 40 per workgroup.
template <int NTH>
struct HeadRegs { float4 v[NTH]; unsigned ok; };     // ok: bit 4*i+j = component j of chunk i is inside the tile

// VEC: 16-byte loads (head_dim % 4 == 0, row strides % 4 == 0, 16-byte aligned bases -- decided once per kernel
// by the host, attn_vec_ok, and compiled in as the kernels' VEC flag); otherwise four 4-byte loads per chunk.  Every load is UNCONDITIONAL
// from a clamped (always legal) address and the zero padding is applied later by head_mask: a conditional
// load is a phi of {0, loaded value}, which makes the compiler shuffle registers -- and therefore wait --
// right behind the request.
template <int NTH, bool VEC>
__device__ __forceinline__ void head_load_t(HeadRegs<NTH>& h, const float* base, long row_stride, int t0, int T, int hd,
                                            int tid) {
  const int t = t0 + (tid >> 2);
  const bool rok = t < T;
  const float* src = base + (long)(rok ? t : t0) * row_stride;
  unsigned ok = 0;
#pragma unroll
  for (int i = 0; i < NTH; ++i) {
    const int c = 4 * (tid & 3) + 16 * i;
    if (VEC) {
      const bool cok = c < hd;
      h.v[i] = *reinterpret_cast<const float4*>(src + (cok ? c : 0));
      if (rok && cok) ok |= 0xFu << (4 * i);
    } else {
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool cok = c + j < hd;
        e[j] = src[cok ? c + j : 0];
        if (rok && cok) ok |= 1u << (4 * i + j);
      }
      h.v[i] = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
  h.ok = ok;
}
// zero padding (rows >= T, columns >= head_dim); call below the point where the loads may complete
template <int NTH>
__device__ __forceinline__ void head_mask(HeadRegs<NTH>& h) {
#pragma unroll
  for (int i = 0; i < NTH; ++i) {
    if (!((h.ok >> (4 * i)) & 1u)) h.v[i].x = 0.f;
    if (!((h.ok >> (4 * i + 1)) & 1u)) h.v[i].y = 0.f;
    if (!((h.ok >> (4 * i + 2)) & 1u)) h.v[i].z = 0.f;
    if (!((h.ok >> (4 * i + 3)) & 1u)) h.v[i].w = 0.f;
  }
}
template <int NTH>
__device__ __forceinline__ void head_store(HeadRegs<NTH>& h, float* dst, int ldh, int tid) {
  head_mask<NTH>(h);
  float* d = dst + (tid >> 2) * ldh + 4 * (tid & 3);
#pragma unroll
  for (int i = 0; i < NTH; ++i) *reinterpret_cast<float4*>(d + 16 * i) = h.v[i];
}
// sum over the row of a[r,:] * b[r,:], combined over the row's 4 threads (every one of them gets the sum)
template <int NTH>
__device__ __forceinline__ float head_rowdot(const HeadRegs<NTH>& a, const HeadRegs<NTH>& b) {
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < NTH; ++i)
    d += (a.v[i].x * b.v[i].x + a.v[i].y * b.v[i].y) + (a.v[i].z * b.v[i].z + a.v[i].w * b.v[i].w);
  d += __shfl_xor(d, 1);
  d += __shfl_xor(d, 2);
  return d;
}

// ------------------------------------------------------------------------------------------------
// forward: grid (q tiles, B*H).  Wave w owns query rows 16w..16w+15 of the tile.
// ------------------------------------------------------------------------------------------------
template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_fwd(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HDP = 16 * NTH, LDH = HDP + 4;
  float* Qs = smem;
  float* Ks = Qs + TS * LDH;
  float* Vs = Ks + TS * LDH;
  // single key tile (T <= 64, the P19 shape): Q is dead once S is formed, and each wave writes P rows
  // 16w..16w+15 only after reading exactly those Q rows -> P overlays Q (same row stride) and two
  // workgroups fit one CU's LDS (64.5 KB each) instead of one
  const bool one_tile = a.T <= TS && LDH >= LDP;             // a P row (64 keys) must fit a Q row
  float* Ps = one_tile ? Qs : Vs + TS * LDH;
  const int ldp = one_tile ? LDH : LDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int q0 = blockIdx.x * TS;
  const long rs = (long)a.B * 3 * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  constexpr bool vec = VEC;
  HeadRegs<NTH> qv;
  if (vec) {
    head_load_t<NTH, true>(qv, qb, rs, q0, a.T, a.hd, tid);
  } else {
    head_load_t<NTH, false>(qv, qb, rs, q0, a.T, a.hd, tid);
  }
  // device seed cell on the scalar path (a vector load would queue behind the tile loads and be waited for
  // in the middle of the softmax)
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  float m_i[4], l_i[4];
  f32x4 o[NTH];
#pragma unroll
  for (int r = 0; r < 4; ++r) { m_i[r] = -INFINITY; l_i[r] = 0.f; }
#pragma unroll
  for (int j = 0; j < NTH; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float inv_keep = 1.0f / (1.0f - a.p_drop);

  for (int k0 = 0; k0 < a.T; k0 += TS) {
    // every load of this key tile (and, first time round, the query tile requested above) is in flight
    // before the first is consumed
    HeadRegs<NTH> kv, vv;
    if (vec) {
      head_load_t<NTH, true>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    }
    uint8_t mb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mb[j] = a.mask[(long)b * a.T + min(k0 + 16 * j + (lane & 15), a.T - 1)];
    __syncthreads();
    if (k0 == 0) head_store<NTH>(qv, Qs, LDH, tid);
    head_store<NTH>(kv, Ks, LDH, tid);
    head_store<NTH>(vv, Vs, LDH, tid);
    __syncthreads();
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_f32<4>(s, Qs + wave * 16 * LDH, LDH, 1, Ks, 1, LDH, HDP, lane);   // S = Q K^T
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = k0 + 16 * j + (lane & 15);
      const bool dead = key >= a.T || mb[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[j][r] = dead ? -INFINITY : s[j][r] * a.scale;
        mx[r] = fmaxf(mx[r], s[j][r]);
      }
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float mn = fmaxf(m_i[r], group16_max(mx[r]));
      alpha[r] = (m_i[r] == -INFINITY) ? 0.f : expf(m_i[r] - mn);
      m_i[r] = mn;
    }
    float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = k0 + 16 * j + (lane & 15);
      float k4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.p_drop > 0.f)     // F.dropout on the attention probabilities (after the softmax sum)
        attn_keep4(k4, seedv, a.site, bh, a.T, q0 + wave * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + 4 * (lane >> 4) + r;
        const float p = (s[j][r] == -INFINITY) ? 0.f : __expf(s[j][r] - m_i[r]);
        rsum[r] += p;
        Ps[row * ldp + 16 * j + (lane & 15)] = p * k4[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) l_i[r] = l_i[r] * alpha[r] + group16_sum(rsum[r]);
#pragma unroll
    for (int j = 0; j < NTH; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[j][r] *= alpha[r];
    __syncthreads();
    mma_f32<NTH>(o, Ps + wave * 16 * ldp, ldp, 1, Vs, LDH, 1, TS, lane);  // O += P V
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + wave * 16 + 4 * (lane >> 4) + r;
    if (q >= a.T) continue;
    const float inv = 1.0f / l_i[r];
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) a.out[((long)q * a.B + b) * a.D + h * a.hd + c] = o[j][r] * inv;
    }
    if ((lane & 15) == 0) a.lse[(long)bh * a.T + q] = m_i[r] + logf(l_i[r]);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dQ: grid (q tiles, B*H); also writes delta = rowsum(dout * out).
// ------------------------------------------------------------------------------------------------
template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HDP = 16 * NTH, LDH = HDP + 4;
  float* Qs = smem;
  float* dOs = Qs + TS * LDH;
  float* Ks = dOs + TS * LDH;
  float* Vs = Ks + TS * LDH;
  float* Ps = Vs + TS * LDH;
  float* lse_s = Ps + TS * LDP;
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int q0 = blockIdx.x * TS;
  const long rs = (long)a.B * 3 * a.D, ro = (long)a.B * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  const float* dob = a.dout + (long)b * a.D + h * a.hd;
  const float* ob = a.out + (long)b * a.D + h * a.hd;
  constexpr bool vec = VEC;
  {
    HeadRegs<NTH> qv, dov, ov;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, q0, a.T, a.hd, tid);
      head_load_t<NTH, true>(dov, dob, ro, q0, a.T, a.hd, tid);
      head_load_t<NTH, true>(ov, ob, ro, q0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, q0, a.T, a.hd, tid);
      head_load_t<NTH, false>(dov, dob, ro, q0, a.T, a.hd, tid);
      head_load_t<NTH, false>(ov, ob, ro, q0, a.T, a.hd, tid);
    }
    const int r = tid >> 2, q = q0 + r;
    const float l = q < a.T ? a.lse[(long)bh * a.T + q] : 0.f;
    __builtin_amdgcn_sched_barrier(0);                  // every request above, every use below
    head_store<NTH>(qv, Qs, LDH, tid);
    head_store<NTH>(dov, dOs, LDH, tid);
    head_mask<NTH>(ov);
    const float d = head_rowdot<NTH>(dov, ov);          // delta = rowsum(dO * O); rows >= T are zero padded
    if ((tid & 3) == 0) {
      dl_s[r] = d; lse_s[r] = l;
      if (q < a.T) a.delta[(long)bh * a.T + q] = d;
    }
  }
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  f32x4 dq[NTH];
#pragma unroll
  for (int j = 0; j < NTH; ++j) dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int k0 = 0; k0 < a.T; k0 += TS) {
    HeadRegs<NTH> kv, vv;
    if (vec) {
      head_load_t<NTH, true>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    }
    __syncthreads();
    head_store<NTH>(kv, Ks, LDH, tid);
    head_store<NTH>(vv, Vs, LDH, tid);
    __syncthreads();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
    mma_f32<4>(s, Qs + wave * 16 * LDH, LDH, 1, Ks, 1, LDH, HDP, lane);     // S  = Q K^T
    mma_f32<4>(dp, dOs + wave * 16 * LDH, LDH, 1, Vs, 1, LDH, HDP, lane);   // dP = dO V^T
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int key = k0 + 16 * j + (lane & 15);
      const bool dead = key >= a.T || a.mask[(long)b * a.T + key];
      float k4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.p_drop > 0.f)
        attn_keep4(k4, seedv, a.site, bh, a.T, q0 + wave * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 + 4 * (lane >> 4) + r;
        float ds = 0.f;
        if (!dead && q0 + row < a.T) {
          const float p = __expf(s[j][r] * a.scale - lse_s[row]);
          ds = p * (dp[j][r] * k4[r] - dl_s[row]) * a.scale;
        }
        Ps[row * LDP + 16 * j + (lane & 15)] = ds;
      }
    }
    __syncthreads();
    mma_f32<NTH>(dq, Ps + wave * 16 * LDP, LDP, 1, Ks, LDH, 1, TS, lane);   // dQ += dS K
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + wave * 16 + 4 * (lane >> 4) + r;
    if (q >= a.T) continue;
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) a.dqkv[((long)q * a.B + b) * 3 * a.D + h * a.hd + c] = dq[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, dK / dV: grid (key tiles, B*H).  Wave w owns key rows 16w..16w+15; scores are formed
// directly transposed (S^T = K Q^T) so that both products below reduce over the query axis.
// ------------------------------------------------------------------------------------------------
template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HDP = 16 * NTH, LDH = HDP + 4;
  float* Ks = smem;
  float* Vs = Ks + TS * LDH;
  float* Qs = Vs + TS * LDH;
  float* dOs = Qs + TS * LDH;
  float* PTs = dOs + TS * LDH;       // (P o M)^T  [key][q]
  float* DSTs = PTs + TS * LDP;      // dS^T       [key][q]
  float* lse_s = DSTs + TS * LDP;
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int k0 = blockIdx.x * TS;
  const long rs = (long)a.B * 3 * a.D, ro = (long)a.B * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  const float* dob = a.dout + (long)b * a.D + h * a.hd;
  constexpr bool vec = VEC;
  {
    HeadRegs<NTH> kv, vv;
    if (vec) {
      head_load_t<NTH, true>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(kv, qb + a.D, rs, k0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, k0, a.T, a.hd, tid);
    }
    head_store<NTH>(kv, Ks, LDH, tid);
    head_store<NTH>(vv, Vs, LDH, tid);
  }
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  f32x4 dk[NTH], dv[NTH];
#pragma unroll
  for (int j = 0; j < NTH; ++j) { dk[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[j] = dk[j]; }
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  bool dead[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = k0 + wave * 16 + 4 * (lane >> 4) + r;
    dead[r] = key >= a.T || a.mask[(long)b * a.T + min(key, a.T - 1)];
  }
  for (int q0 = 0; q0 < a.T; q0 += TS) {
    HeadRegs<NTH> qv, dov;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, q0, a.T, a.hd, tid);
      head_load_t<NTH, true>(dov, dob, ro, q0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, q0, a.T, a.hd, tid);
      head_load_t<NTH, false>(dov, dob, ro, q0, a.T, a.hd, tid);
    }
    __syncthreads();
    head_store<NTH>(qv, Qs, LDH, tid);
    head_store<NTH>(dov, dOs, LDH, tid);
    if (tid < TS) {
      const int q = q0 + tid;
      lse_s[tid] = q < a.T ? a.lse[(long)bh * a.T + q] : 0.f;
      dl_s[tid] = q < a.T ? a.delta[(long)bh * a.T + q] : 0.f;
    }
    __syncthreads();
    f32x4 st[4], dpt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { st[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dpt[j] = st[j]; }
    mma_f32<4>(st, Ks + wave * 16 * LDH, LDH, 1, Qs, 1, LDH, HDP, lane);     // S^T  = K Q^T
    mma_f32<4>(dpt, Vs + wave * 16 * LDH, LDH, 1, dOs, 1, LDH, HDP, lane);   // dP^T = V dO^T
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qi = 16 * j + (lane & 15);
      const int q = q0 + qi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int krow = wave * 16 + 4 * (lane >> 4) + r;
        float pm = 0.f, ds = 0.f;
        if (!dead[r] && q < a.T) {
          const float p = __expf(st[j][r] * a.scale - lse_s[qi]);
          float keep = 1.f;
          if (a.p_drop > 0.f) keep = attn_keep1(seedv, a.site, bh, a.T, q, k0 + krow, a.p_drop, inv_keep);
          pm = p * keep;
          ds = p * (dpt[j][r] * keep - dl_s[qi]) * a.scale;
        }
        PTs[krow * LDP + qi] = pm;
        DSTs[krow * LDP + qi] = ds;
      }
    }
    __syncthreads();
    mma_f32<NTH>(dv, PTs + wave * 16 * LDP, LDP, 1, dOs, LDH, 1, TS, lane);  // dV += (P o M)^T dO
    mma_f32<NTH>(dk, DSTs + wave * 16 * LDP, LDP, 1, Qs, LDH, 1, TS, lane);  // dK += dS^T Q
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = k0 + wave * 16 + 4 * (lane >> 4) + r;
    if (key >= a.T) continue;
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) {
        float* row = a.dqkv + ((long)key * a.B + b) * 3 * a.D + h * a.hd + c;
        row[a.D] = dk[j][r];
        row[2 * a.D] = dv[j][r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, single-tile form (T <= 64: the P19 shape): one workgroup per (sample, head) forms S, P,
// dP and dS ONCE and produces dQ, dK and dV from them (the two-kernel form above recomputes the
// scores in each kernel and is kept for longer sequences).  Wave w owns query rows 16w..16w+15 for
// S / dP / dQ and key rows 16w..16w+15 for dK / dV; dS and P o M are exchanged through LDS.
// ------------------------------------------------------------------------------------------------
template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_bwd_one(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int HDP = 16 * NTH, LDH = HDP + 4;
  float* Qs = smem;
  float* Ks = Qs + TS * LDH;
  float* Vs = Ks + TS * LDH;
  float* dOs = Vs + TS * LDH;
  float* PMs = dOs + TS * LDH;       // P o M   [q][key]
  float* DSs = PMs + TS * LDP;       // dS      [q][key]
  float* lse_s = DSs + TS * LDP;
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const long rs = (long)a.B * 3 * a.D, ro = (long)a.B * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  const float* dob = a.dout + (long)b * a.D + h * a.hd;
  const float* ob = a.out + (long)b * a.D + h * a.hd;
  // all five tiles (Q, K, V, dO, O), the LSE row and the key mask are requested in one burst
  constexpr bool vec = VEC;
  uint64_t seedv = a.seed;
  uint8_t mb[4];
  {
    HeadRegs<NTH> qv, kv, vv, dov, ov;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(dov, dob, ro, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(ov, ob, ro, 0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(dov, dob, ro, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(ov, ob, ro, 0, a.T, a.hd, tid);
    }
    const int r = tid >> 2;
    const float l = r < a.T ? a.lse[(long)bh * a.T + r] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) mb[j] = a.mask[(long)b * a.T + min(16 * j + (lane & 15), a.T - 1)];
    if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);   // scalar path, under the tile loads
    __builtin_amdgcn_sched_barrier(0);                  // every request above, every use below
    head_store<NTH>(qv, Qs, LDH, tid);
    head_store<NTH>(kv, Ks, LDH, tid);
    head_store<NTH>(vv, Vs, LDH, tid);
    head_store<NTH>(dov, dOs, LDH, tid);
    head_mask<NTH>(ov);
    const float d = head_rowdot<NTH>(dov, ov);          // delta = rowsum(dO * O)
    if ((tid & 3) == 0) { dl_s[r] = d; lse_s[r] = l; }
  }
  __syncthreads();
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  f32x4 s[4], dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
  mma_f32<4>(s, Qs + wave * 16 * LDH, LDH, 1, Ks, 1, LDH, HDP, lane);     // S  = Q K^T
  mma_f32<4>(dp, dOs + wave * 16 * LDH, LDH, 1, Vs, 1, LDH, HDP, lane);   // dP = dO V^T
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 16 * j + (lane & 15);
    const bool dead = key >= a.T || mb[j];
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.p_drop > 0.f)
      attn_keep4(k4, seedv, a.site, bh, a.T, wave * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * (lane >> 4) + r;
      float pm = 0.f, ds = 0.f;
      if (!dead && row < a.T) {
        const float p = __expf(s[j][r] * a.scale - lse_s[row]);
        pm = p * k4[r];
        ds = p * (dp[j][r] * k4[r] - dl_s[row]) * a.scale;
      }
      PMs[row * LDP + key] = pm;
      DSs[row * LDP + key] = ds;
    }
  }
  __syncthreads();
  f32x4 dq[NTH], dk[NTH], dv[NTH];
#pragma unroll
  for (int j = 0; j < NTH; ++j) { dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[j] = dq[j]; dv[j] = dq[j]; }
  mma_f32<NTH>(dq, DSs + wave * 16 * LDP, LDP, 1, Ks, LDH, 1, TS, lane);    // dQ = dS K
  mma_f32<NTH>(dk, DSs + wave * 16, 1, LDP, Qs, LDH, 1, TS, lane);          // dK = dS^T Q   (A(i,k) = dS[k][i])
  mma_f32<NTH>(dv, PMs + wave * 16, 1, LDP, dOs, LDH, 1, TS, lane);         // dV = (P o M)^T dO
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = wave * 16 + 4 * (lane >> 4) + r;
    if (t >= a.T) continue;
    float* row = a.dqkv + ((long)t * a.B + b) * 3 * a.D + h * a.hd;
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) { row[c] = dq[j][r]; row[a.D + c] = dk[j][r]; row[2 * a.D + c] = dv[j][r]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Single-tile attention (T <= 64: the P19 shape) on the split-bf16 matrix path.  The kernels above contract in exact
// fp32 (v_mfma_f32_16x16x4_f32: 1/16 of the bf16 rate, and one 4-byte LDS read per operand and MFMA); here the
// Q / K / V / dO tiles are split ONCE into bf16 hi/lo planes while they are stored to LDS, every contraction is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 (fp32 accumulate, ~2^-16 per product like every dense product of
// the path), and an operand fragment is one 16-byte LDS read -- or, where the reduction index runs along the plane's
// ROWS (P V, dS K, dS^T Q, P^T dO), two transposing reads (ds_read_b64_tr_b16).  P o M and dS are stored TRANSPOSED
// ([key][query]): the four query rows a lane holds in an accumulator become one 8-byte store.
// Same masks, same dropout quads (attn_keep4), same saved LSE as the fp32 kernels; used in the bf16 modes only.
// ------------------------------------------------------------------------------------------------
typedef __bf16 abf8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf4 __attribute__((ext_vector_type(4)));
typedef short as4 __attribute__((ext_vector_type(4)));
typedef short as8 __attribute__((ext_vector_type(8)));
constexpr int LDT = TS + 8;                     // row stride (bf16) of the transposed score planes [key][query]

// fragment with the reduction index along the plane's columns: lane -> row row0 + (lane & 15), columns k0 + 8 (lane >> 4) ..
__device__ __forceinline__ abf8 frag_n(const __bf16* P, int ld, int row0, int k0, int lane) {
  return *reinterpret_cast<const abf8*>(P + (row0 + (lane & 15)) * ld + k0 + 8 * (lane >> 4));
}
// fragment with the reduction index along the plane's rows: lane -> column col0 + (lane & 15), rows k0 + 8 (lane >> 4) ..
__device__ __forceinline__ abf8 frag_t(const __bf16* P, int ld, int k0, int col0, int lane) {
  const int i = lane & 15, G = lane >> 4;
  const __bf16* src = P + (k0 + 8 * G + (i >> 2)) * ld + col0 + 4 * (i & 3);
  const as4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as4 __attribute__((address_space(3)))*)(src));
  const as4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as4 __attribute__((address_space(3)))*)(src + 4 * ld));
  const as8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(abf8, o);
}
// acc[j] += A(16 x KK) B(KK x 16 NT), split-bf16.  AT / BT: operand read transposed (reduction along plane rows).
// a0: first row (AT: first column) of A's 16-wide slice; B tile j starts at row (BT: column) 16 j.
template <int NT, bool AT, bool BT>
__device__ __forceinline__ void mma_b16(f32x4 (&acc)[NT], const __bf16* Ah, const __bf16* Al, int lda, int a0,
                                        const __bf16* Bh, const __bf16* Bl, int ldb, int KK, int lane, bool one) {
  // `one` (RD_PREC_BF16: hi*hi only) is decided outside the reduction loop: inside it every step became its own basic block and
  // the fragment reads of the next step could not move above this step's products
  if (!one) {
#pragma unroll
    for (int k0 = 0; k0 < KK; k0 += 32) {
      const abf8 ah = AT ? frag_t(Ah, lda, k0, a0, lane) : frag_n(Ah, lda, a0, k0, lane);
      const abf8 al = AT ? frag_t(Al, lda, k0, a0, lane) : frag_n(Al, lda, a0, k0, lane);
      abf8 bh[NT], bl[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = BT ? frag_t(Bh, ldb, k0, 16 * j, lane) : frag_n(Bh, ldb, 16 * j, k0, lane);
        bl[j] = BT ? frag_t(Bl, ldb, k0, 16 * j, lane) : frag_n(Bl, ldb, 16 * j, k0, lane);
      }
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
      const abf8 ah = AT ? frag_t(Ah, lda, k0, a0, lane) : frag_n(Ah, lda, a0, k0, lane);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const abf8 bh = BT ? frag_t(Bh, ldb, k0, 16 * j, lane) : frag_n(Bh, ldb, 16 * j, k0, lane);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[j], 0, 0, 0);
      }
    }
  }
}
// head tile (fp32 registers) -> hi/lo planes [64][ldb], columns [16 NTH, hdp) zeroed
template <int NTH>
__device__ __forceinline__ void head_store_b16(HeadRegs<NTH>& h, __bf16* Ph, __bf16* Pl, int ldb, int hdp, int tid) {
  head_mask<NTH>(h);
  const int o = (tid >> 2) * ldb + 4 * (tid & 3);
#pragma unroll
  for (int i = 0; i < NTH; ++i) {
    const float x[4] = {h.v[i].x, h.v[i].y, h.v[i].z, h.v[i].w};
    abf4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) { hi[c] = (__bf16)x[c]; lo[c] = (__bf16)(x[c] - (float)hi[c]); }
    *reinterpret_cast<abf4*>(Ph + o + 16 * i) = hi;
    *reinterpret_cast<abf4*>(Pl + o + 16 * i) = lo;
  }
  if (16 * NTH < hdp) {                                   // at most 16 tail columns (hdp = round-up to 32)
    abf4 z;
#pragma unroll
    for (int c = 0; c < 4; ++c) z[c] = (__bf16)0.f;
    *reinterpret_cast<abf4*>(Ph + o + 16 * NTH) = z;
    *reinterpret_cast<abf4*>(Pl + o + 16 * NTH) = z;
  }
}
// the four accumulator rows of a lane (consecutive queries) -> transposed planes [key][query], one 8-byte store each
__device__ __forceinline__ void store_t4(__bf16* Ph, __bf16* Pl, int key, int q4, const float (&v)[4]) {
  abf4 hi, lo;
#pragma unroll
  for (int r = 0; r < 4; ++r) { hi[r] = (__bf16)v[r]; lo[r] = (__bf16)(v[r] - (float)hi[r]); }
  *reinterpret_cast<abf4*>(Ph + key * LDT + q4) = hi;
  *reinterpret_cast<abf4*>(Pl + key * LDT + q4) = lo;
}

template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_fwd_one_b16(AttnArgs a, int one) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16;
  __bf16* Qh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Ql = Qh + TS * LDB;
  __bf16* Kh = Ql + TS * LDB;
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  constexpr bool OVL = LDB >= LDT;                  // P^T overlays Q (dead once S is formed; barrier below) when it fits
  __bf16* Ph = OVL ? Qh : Vl + TS * LDB;
  __bf16* Pl = Ph + TS * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const long rs = (long)a.B * 3 * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  constexpr bool vec = VEC;
  ASTAMP(0);
  uint64_t seedv = a.seed;
  uint8_t mb[4];
  {
    HeadRegs<NTH> qv, kv, vv;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) mb[j] = a.mask[(long)b * a.T + min(16 * j + (lane & 15), a.T - 1)];
    if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(1);
    head_store_b16<NTH>(qv, Qh, Ql, LDB, HDP, tid);
    head_store_b16<NTH>(kv, Kh, Kl, LDB, HDP, tid);
    head_store_b16<NTH>(vv, Vh, Vl, LDB, HDP, tid);
  }
  ASTAMP(2);
  __syncthreads();
  ASTAMP(3);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  f32x4 s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  mma_b16<4, false, false>(s, Qh, Ql, LDB, wave * 16, Kh, Kl, LDB, HDP, lane, one);      // S = Q K^T
  ASTAMP(4);
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 16 * j + (lane & 15);
    const bool dead = key >= a.T || mb[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[j][r] = dead ? -INFINITY : s[j][r] * a.scale;
      mx[r] = fmaxf(mx[r], s[j][r]);
    }
  }
  float m_i[4], l_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) m_i[r] = group16_max(mx[r]);
  __syncthreads();                                  // every wave is done with Q before P^T overwrites it
  float rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 16 * j + (lane & 15);
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.p_drop > 0.f)
      attn_keep4(k4, seedv, a.site, bh, a.T, wave * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
    float pv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (s[j][r] == -INFINITY) ? 0.f : __expf(s[j][r] - m_i[r]);
      rsum[r] += p;
      pv[r] = p * k4[r];
    }
    store_t4(Ph, Pl, key, wave * 16 + 4 * (lane >> 4), pv);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) l_i[r] = group16_sum(rsum[r]);
  __syncthreads();
  ASTAMP(5);
  f32x4 o[NTH];
#pragma unroll
  for (int j = 0; j < NTH; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  mma_b16<NTH, true, true>(o, Ph, Pl, LDT, wave * 16, Vh, Vl, LDB, TS, lane, one);        // O = P V
  ASTAMP(6);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = wave * 16 + 4 * (lane >> 4) + r;
    if (q >= a.T) continue;
    const float inv = 1.0f / l_i[r];
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) a.out[((long)q * a.B + b) * a.D + h * a.hd + c] = o[j][r] * inv;
    }
    if ((lane & 15) == 0) a.lse[(long)bh * a.T + q] = m_i[r] + logf(l_i[r]);
  }
  ASTAMP(7);
}

template <int NTH, bool VEC>
__global__ __launch_bounds__(256) void k_attn_bwd_one_b16(AttnArgs a, int one) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16;
  __bf16* Qh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Ql = Qh + TS * LDB;
  __bf16* Kh = Ql + TS * LDB;
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  __bf16* Oh = Vl + TS * LDB;                       // dO
  __bf16* Ol = Oh + TS * LDB;
  __bf16* Ph = Ol + TS * LDB;                       // (P o M)^T  [key][query]
  __bf16* Pl = Ph + TS * LDT;
  __bf16* Sh = Pl + TS * LDT;                       // dS^T       [key][query]
  __bf16* Sl = Sh + TS * LDT;
  float* lse_s = reinterpret_cast<float*>(Sl + TS * LDT);
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const long rs = (long)a.B * 3 * a.D, ro = (long)a.B * a.D;
  const float* qb = a.qkv + (long)b * 3 * a.D + h * a.hd;
  const float* dob = a.dout + (long)b * a.D + h * a.hd;
  const float* ob = a.out + (long)b * a.D + h * a.hd;
  constexpr bool vec = VEC;
  ASTAMP(0);
  uint64_t seedv = a.seed;
  uint8_t mb[4];
  {
    HeadRegs<NTH> qv, kv, vv, dov, ov;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(dov, dob, ro, 0, a.T, a.hd, tid);
      head_load_t<NTH, true>(ov, ob, ro, 0, a.T, a.hd, tid);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(kv, qb + a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(dov, dob, ro, 0, a.T, a.hd, tid);
      head_load_t<NTH, false>(ov, ob, ro, 0, a.T, a.hd, tid);
    }
    const int r = tid >> 2;
    const float l = r < a.T ? a.lse[(long)bh * a.T + r] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) mb[j] = a.mask[(long)b * a.T + min(16 * j + (lane & 15), a.T - 1)];
    if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
    __builtin_amdgcn_sched_barrier(0);
    ASTAMP(1);
    head_store_b16<NTH>(qv, Qh, Ql, LDB, HDP, tid);
    head_store_b16<NTH>(kv, Kh, Kl, LDB, HDP, tid);
    head_store_b16<NTH>(vv, Vh, Vl, LDB, HDP, tid);
    head_mask<NTH>(ov);
    head_mask<NTH>(dov);
    const float d = head_rowdot<NTH>(dov, ov);          // delta = rowsum(dO * O), in fp32
    head_store_b16<NTH>(dov, Oh, Ol, LDB, HDP, tid);
    if ((tid & 3) == 0) { dl_s[r] = d; lse_s[r] = l; }
  }
  ASTAMP(2);
  __syncthreads();
  ASTAMP(3);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  f32x4 s[4], dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
  mma_b16<4, false, false>(s, Qh, Ql, LDB, wave * 16, Kh, Kl, LDB, HDP, lane, one);      // S  = Q K^T
  mma_b16<4, false, false>(dp, Oh, Ol, LDB, wave * 16, Vh, Vl, LDB, HDP, lane, one);     // dP = dO V^T
  ASTAMP(4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = 16 * j + (lane & 15);
    const bool dead = key >= a.T || mb[j];
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.p_drop > 0.f)
      attn_keep4(k4, seedv, a.site, bh, a.T, wave * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
    float pm[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + 4 * (lane >> 4) + r;
      pm[r] = 0.f; ds[r] = 0.f;
      if (!dead && row < a.T) {
        const float p = __expf(s[j][r] * a.scale - lse_s[row]);
        pm[r] = p * k4[r];
        ds[r] = p * (dp[j][r] * k4[r] - dl_s[row]) * a.scale;
      }
    }
    store_t4(Ph, Pl, key, wave * 16 + 4 * (lane >> 4), pm);
    store_t4(Sh, Sl, key, wave * 16 + 4 * (lane >> 4), ds);
  }
  ASTAMP(5);
  __syncthreads();
  f32x4 dq[NTH], dk[NTH], dv[NTH];
#pragma unroll
  for (int j = 0; j < NTH; ++j) { dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[j] = dq[j]; dv[j] = dq[j]; }
  mma_b16<NTH, true, true>(dq, Sh, Sl, LDT, wave * 16, Kh, Kl, LDB, TS, lane, one);      // dQ = dS K       (rows: queries)
  mma_b16<NTH, false, true>(dk, Sh, Sl, LDT, wave * 16, Qh, Ql, LDB, TS, lane, one);     // dK = dS^T Q     (rows: keys)
  mma_b16<NTH, false, true>(dv, Ph, Pl, LDT, wave * 16, Oh, Ol, LDB, TS, lane, one);     // dV = (P o M)^T dO
  ASTAMP(6);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = wave * 16 + 4 * (lane >> 4) + r;
    if (t >= a.T) continue;
    float* row = a.dqkv + ((long)t * a.B + b) * 3 * a.D + h * a.hd;
#pragma unroll
    for (int j = 0; j < NTH; ++j) {
      const int c = 16 * j + (lane & 15);
      if (c < a.hd) { row[c] = dq[j][r]; row[a.D + c] = dk[j][r]; row[2 * a.D + c] = dv[j][r]; }
    }
  }
  ASTAMP(7);
}

// Eight-wave form of k_attn_fwd_one_b16: wave = (query row tile wq, half wh).  S: two of the four key tiles per wave, the row
// maximum and the row sum are combined across the two halves through 1 KB of LDS; O = P V: the head-dim tiles split
// between the halves.  Loads: waves 0-3 fetch Q and K, waves 4-7 V.
template <int NTH, bool VEC>
__global__ __launch_bounds__(512) void k_attn_fwd_one_b16w(AttnArgs a, int one) {
  RD_TOUCH_CODE_X(RD_TL_ATTN_FWD_ONE, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16, NA = (NTH + 1) / 2;
  __bf16* Qh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Ql = Qh + TS * LDB;
  __bf16* Kh = Ql + TS * LDB;
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  constexpr bool OVL = LDB >= LDT;
  __bf16* Ph = OVL ? Qh : Vl + TS * LDB;
  __bf16* Pl = Ph + TS * LDT;
  float* mxs = reinterpret_cast<float*>((OVL ? Vl + TS * LDB : Pl + TS * LDT));    // [2][64] partial row maxima
  float* sms = mxs + 2 * TS;                                                       // [2][64] partial row sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wq = wave & 3, wh = wave >> 2, lt = tid & 255;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const AttnRows ar = attn_rows(a, b);
  const int Tv = ar.Tv;
  const long rs = ar.rstep * 3 * a.D, ro = ar.rstep * a.D;
  const float* qb = a.qkv + ar.row0 * 3 * a.D + h * a.hd;
  constexpr bool vec = VEC;
  uint64_t seedv = a.seed;
  uint8_t mb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = min(16 * (2 * wh + j) + (lane & 15), a.T - 1);
    mb[j] = a.plan ? (uint8_t)(key >= Tv) : a.mask[(long)b * a.T + key];
  }
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  if (wh == 0) {
    HeadRegs<NTH> qv, kv;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, true>(kv, qb + a.D, rs, 0, Tv, a.hd, lt);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, false>(kv, qb + a.D, rs, 0, Tv, a.hd, lt);
    }
    __builtin_amdgcn_sched_barrier(0);
    head_store_b16<NTH>(qv, Qh, Ql, LDB, HDP, lt);
    head_store_b16<NTH>(kv, Kh, Kl, LDB, HDP, lt);
  } else {
    HeadRegs<NTH> vv;
    if (vec) head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, 0, Tv, a.hd, lt);
    else head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, 0, Tv, a.hd, lt);
    __builtin_amdgcn_sched_barrier(0);
    head_store_b16<NTH>(vv, Vh, Vl, LDB, HDP, lt);
  }
  __syncthreads();
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  f32x4 s[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  mma_b16<2, false, false>(s, Qh, Ql, LDB, wq * 16, Kh + 32 * wh * LDB, Kl + 32 * wh * LDB, LDB, HDP, lane, one);    // S = Q K^T
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = 16 * (2 * wh + j) + (lane & 15);
    const bool dead = key >= a.T || mb[j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[j][r] = dead ? -INFINITY : s[j][r] * a.scale;
      mx[r] = fmaxf(mx[r], s[j][r]);
    }
  }
  const int row0 = wq * 16 + 4 * (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mx[r] = group16_max(mx[r]);
    if ((lane & 15) == 0) mxs[wh * TS + row0 + r] = mx[r];
  }
  __syncthreads();                                  // partial maxima visible; every wave is done with Q before P^T overwrites it
  float m_i[4], rsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) m_i[r] = fmaxf(mxs[row0 + r], mxs[TS + row0 + r]);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = 16 * (2 * wh + j) + (lane & 15);
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.p_drop > 0.f)
      attn_keep4(k4, seedv, a.site, bh, a.T, row0, min(key, a.T - 1), a.p_drop, inv_keep);
    float pv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = (s[j][r] == -INFINITY) ? 0.f : __expf(s[j][r] - m_i[r]);
      rsum[r] += p;
      pv[r] = p * k4[r];
    }
    store_t4(Ph, Pl, key, row0, pv);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    rsum[r] = group16_sum(rsum[r]);
    if ((lane & 15) == 0) sms[wh * TS + row0 + r] = rsum[r];
  }
  __syncthreads();
  float l_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) l_i[r] = sms[row0 + r] + sms[TS + row0 + r];
  const int t0 = wh * NA;
  f32x4 o[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  mma_b16<NA, true, true>(o, Ph, Pl, LDT, wq * 16, Vh + 16 * t0, Vl + 16 * t0, LDB, TS, lane, one);      // O = P V
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = row0 + r;
    if (q >= Tv) continue;
    const float inv = 1.0f / l_i[r];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int c = 16 * (t0 + j) + (lane & 15);
      if (t0 + j < NTH && c < a.hd) a.out[ar.row0 * a.D + (long)q * ro + h * a.hd + c] = o[j][r] * inv;
    }
    if (wh == 0 && (lane & 15) == 0) a.lse[(long)bh * a.T + q] = m_i[r] + logf(l_i[r]);
  }
}

// Eight-wave form of the kernel above: wave = (row tile wq, half wh).  S / dP: each wave takes two of the four key
// tiles of its query rows; dQ / dK / dV: the head-dim tiles are split between the two waves of a row tile.  Loads: waves
// 0-3 fetch Q, K, V, waves 4-7 dO, O (+ delta, LSE).  Same LDS planes, two waves per SIMD instead of one: the phases of a
// workgroup overlap a little instead of not at all (nothing needs a cross-wave reduction: the backward uses the saved LSE).
template <int NTH, bool VEC>
__global__ __launch_bounds__(512) void k_attn_bwd_one_b16w(AttnArgs a, int one) {
  RD_TOUCH_CODE_X(RD_TL_ATTN_BWD_ONE, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16, NA = (NTH + 1) / 2;
  __bf16* Qh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Ql = Qh + TS * LDB;
  __bf16* Kh = Ql + TS * LDB;
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  __bf16* Oh = Vl + TS * LDB;
  __bf16* Ol = Oh + TS * LDB;
  __bf16* Ph = Ol + TS * LDB;
  __bf16* Pl = Ph + TS * LDT;
  __bf16* Sh = Pl + TS * LDT;
  __bf16* Sl = Sh + TS * LDT;
  float* lse_s = reinterpret_cast<float*>(Sl + TS * LDT);
  float* dl_s = lse_s + TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wq = wave & 3, wh = wave >> 2, lt = tid & 255;
  const int bh = blockIdx.x, b = bh / a.H, h = bh - b * a.H;
  const AttnRows ar = attn_rows(a, b);
  const int Tv = ar.Tv;
  const long rs = ar.rstep * 3 * a.D, ro = ar.rstep * a.D;
  const float* qb = a.qkv + ar.row0 * 3 * a.D + h * a.hd;
  const float* dob = a.dout + ar.row0 * a.D + h * a.hd;
  const float* ob = a.out + ar.row0 * a.D + h * a.hd;
  constexpr bool vec = VEC;
  uint64_t seedv = a.seed;
  uint8_t mb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = min(16 * (2 * wh + j) + (lane & 15), a.T - 1);
    mb[j] = a.plan ? (uint8_t)(key >= Tv) : a.mask[(long)b * a.T + key];
  }
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  if (wh == 0) {                                      // wave-uniform
    HeadRegs<NTH> qv, kv, vv;
    if (vec) {
      head_load_t<NTH, true>(qv, qb, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, true>(kv, qb + a.D, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, true>(vv, qb + 2 * a.D, rs, 0, Tv, a.hd, lt);
    } else {
      head_load_t<NTH, false>(qv, qb, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, false>(kv, qb + a.D, rs, 0, Tv, a.hd, lt);
      head_load_t<NTH, false>(vv, qb + 2 * a.D, rs, 0, Tv, a.hd, lt);
    }
    __builtin_amdgcn_sched_barrier(0);
    head_store_b16<NTH>(qv, Qh, Ql, LDB, HDP, lt);
    head_store_b16<NTH>(kv, Kh, Kl, LDB, HDP, lt);
    head_store_b16<NTH>(vv, Vh, Vl, LDB, HDP, lt);
  } else {
    HeadRegs<NTH> dov, ov;
    if (vec) {
      head_load_t<NTH, true>(dov, dob, ro, 0, Tv, a.hd, lt);
      head_load_t<NTH, true>(ov, ob, ro, 0, Tv, a.hd, lt);
    } else {
      head_load_t<NTH, false>(dov, dob, ro, 0, Tv, a.hd, lt);
      head_load_t<NTH, false>(ov, ob, ro, 0, Tv, a.hd, lt);
    }
    const int r = lt >> 2;
    const float l = r < Tv ? a.lse[(long)bh * a.T + r] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    head_mask<NTH>(ov);
    head_mask<NTH>(dov);
    const float d = head_rowdot<NTH>(dov, ov);
    head_store_b16<NTH>(dov, Oh, Ol, LDB, HDP, lt);
    if ((lt & 3) == 0) { dl_s[r] = d; lse_s[r] = l; }
  }
  __syncthreads();
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  f32x4 s[2], dp[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
  mma_b16<2, false, false>(s, Qh, Ql, LDB, wq * 16, Kh + 32 * wh * LDB, Kl + 32 * wh * LDB, LDB, HDP, lane, one);    // S
  mma_b16<2, false, false>(dp, Oh, Ol, LDB, wq * 16, Vh + 32 * wh * LDB, Vl + 32 * wh * LDB, LDB, HDP, lane, one);   // dP
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = 16 * (2 * wh + j) + (lane & 15);
    const bool dead = key >= a.T || mb[j];
    float k4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.p_drop > 0.f)
      attn_keep4(k4, seedv, a.site, bh, a.T, wq * 16 + 4 * (lane >> 4), min(key, a.T - 1), a.p_drop, inv_keep);
    float pm[4], ds[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wq * 16 + 4 * (lane >> 4) + r;
      pm[r] = 0.f; ds[r] = 0.f;
      if (!dead && row < Tv) {
        const float p = __expf(s[j][r] * a.scale - lse_s[row]);
        pm[r] = p * k4[r];
        ds[r] = p * (dp[j][r] * k4[r] - dl_s[row]) * a.scale;
      }
    }
    store_t4(Ph, Pl, key, wq * 16 + 4 * (lane >> 4), pm);
    store_t4(Sh, Sl, key, wq * 16 + 4 * (lane >> 4), ds);
  }
  __syncthreads();
  const int t0 = wh * NA;                             // first head-dim tile of this wave
  f32x4 dq[NA], dk[NA], dv[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) { dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[j] = dq[j]; dv[j] = dq[j]; }
  mma_b16<NA, true, true>(dq, Sh, Sl, LDT, wq * 16, Kh + 16 * t0, Kl + 16 * t0, LDB, TS, lane, one);     // dQ = dS K
  mma_b16<NA, false, true>(dk, Sh, Sl, LDT, wq * 16, Qh + 16 * t0, Ql + 16 * t0, LDB, TS, lane, one);    // dK = dS^T Q
  mma_b16<NA, false, true>(dv, Ph, Pl, LDT, wq * 16, Oh + 16 * t0, Ol + 16 * t0, LDB, TS, lane, one);    // dV = (P o M)^T dO
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int t = wq * 16 + 4 * (lane >> 4) + r;
    if (t >= Tv) continue;
    float* row = a.dqkv + ar.row0 * 3 * a.D + (long)t * rs + h * a.hd;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int c = 16 * (t0 + j) + (lane & 15);
      if (t0 + j < NTH && c < a.hd) { row[c] = dq[j][r]; row[a.D + c] = dk[j][r]; row[2 * a.D + c] = dv[j][r]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Multi-tile attention (T > 64: P12's 215 steps, PAM's 600) on the split-bf16 matrix path, scores chained in REGISTERS.
// The MFMA accumulator layout (lane l: column l & 15, rows 4 (l >> 4) + r) is, up to a permutation of the reduction index that
// the other operand can follow, the B-operand layout of the next product.  So the kernels form the scores with the OWNED
// dimension in the accumulator columns and feed them straight back:
//   forward / dQ (a wave owns 16 queries):  S^T = K Q^T, dP^T = V dO^T  ->  O^T += V^T (P o M)^T,  dQ^T += K^T dS^T
//   dK / dV      (a wave owns 16 keys):     S   = Q K^T, dP   = dO V^T  ->  dV^T += dO^T (P o M),  dK^T += Q^T dS
// The owned rows' own operand (Q, dO / K, V) lives in registers as B fragments read once from global memory; only the streamed
// 64-row tiles (K, V / Q, dO) go through LDS as hi/lo planes, whose transposed A fragments are two ds_read_b64_tr_b16 of FOUR
// rows each (frag_t2: rows 4G.. of two 16-row blocks -- the permutation the accumulators impose).  No P / dS planes, no
// transposing stores, one barrier less per tile, 53 KB of LDS instead of 98 - 143, 8 waves (128 owned rows) per workgroup, and the
// results leave as 16-byte stores (an accumulator holds 4 consecutive head columns of one row).
//  * a 64-key tile with no live key (padding: most P12 samples are far shorter than 215) contributes exactly nothing -- p = 0,
//    alpha = 1 -- and is skipped: one 64-bit "tile has a live key" word per sample, built from the key mask before the loop;
//  * the NEXT tile's rows are requested while the current tile is computed (register double buffer) and the loop's barriers
//    order LDS only (lds_barrier: no vmcnt drain).
// Same masks, same dropout quads (attn_quad), same saved LSE as the fp32 kernels; padded layout or token plan (AttnRows).
// ------------------------------------------------------------------------------------------------
constexpr int QW = 8;                   // waves per workgroup: 16-row blocks of the owned dimension
constexpr int QROWS = 16 * QW;

// bit i: key tile i of sample b holds at least one live key (tiles >= 64 are never skipped)
__device__ __forceinline__ uint64_t live_key_tiles(const uint8_t* __restrict__ mrow, int T, int lane) {
  const int nt = (T + TS - 1) / TS;
  uint64_t live = 0ull;
  for (int i0 = 0; i0 < nt && i0 < 64; i0 += 4) {
    uint8_t mv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = (i0 + u) * TS + lane;
      const uint8_t m = mrow[min(key, T - 1)];
      mv[u] = key < T ? m : (uint8_t)1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (__ballot(mv[u] == 0) != 0ull && i0 + u < 64) live |= 1ull << (i0 + u);
  }
  return live;
}
// first live tile index >= i, or nt
__device__ __forceinline__ int next_live_tile(uint64_t live, int i, int nt) {
  while (i < nt && i < 64 && !((live >> i) & 1ull)) ++i;
  return i;
}

// The 8 consecutive head columns 32 ks + 8 G .. +7 of ONE row, per reduction step ks: a lane's share of a register-resident
// B operand (lane l: row l & 15 of the wave's 16, G = l >> 4).  Loads are unconditional from clamped addresses (raw_load);
// rows >= T and columns >= head_dim become zero in raw_split.
template <int NKS>
struct RawFrag { float4 v[NKS][2]; };
template <int NKS, bool VEC>
__device__ __forceinline__ void raw_load(RawFrag<NKS>& f, const float* __restrict__ row, int hd, int G) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int c = 32 * ks + 8 * G;
    if (VEC) {
      f.v[ks][0] = *reinterpret_cast<const float4*>(row + (c < hd ? c : 0));
      f.v[ks][1] = *reinterpret_cast<const float4*>(row + (c + 4 < hd ? c + 4 : 0));
    } else {
      float e[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) e[u] = row[c + u < hd ? c + u : 0];
      f.v[ks][0] = make_float4(e[0], e[1], e[2], e[3]);
      f.v[ks][1] = make_float4(e[4], e[5], e[6], e[7]);
    }
  }
}
template <int NKS>
__device__ __forceinline__ void raw_zero(RawFrag<NKS>& f, bool rok, int hd, int G) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    float* e = reinterpret_cast<float*>(&f.v[ks][0]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (!(rok && 32 * ks + 8 * G + u < hd)) e[u] = 0.f;
  }
}
template <int NKS>
__device__ __forceinline__ void raw_split(const RawFrag<NKS>& f, abf8 (&fh)[NKS], abf8 (&fl)[NKS]) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float* e = reinterpret_cast<const float*>(&f.v[ks][0]);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      fh[ks][u] = (__bf16)e[u];
      fl[ks][u] = (__bf16)(e[u] - (float)fh[ks][u]);
    }
  }
}
// two accumulator tiles (16-row blocks 2p and 2p+1 of the 64-row tile) -> the B operand of reduction step p:
// slots 0..3 = rows 32 p + 4 G + r, slots 4..7 = rows 32 p + 16 + 4 G + r (the order frag_t2 reads the other operand in)
__device__ __forceinline__ void pack_b(const float (&x0)[4], const float (&x1)[4], abf8& bh, abf8& bl) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bh[r] = (__bf16)x0[r]; bl[r] = (__bf16)(x0[r] - (float)bh[r]);
    bh[4 + r] = (__bf16)x1[r]; bl[4 + r] = (__bf16)(x1[r] - (float)bh[4 + r]);
  }
}
// A fragment with the reduction index along the plane's rows, in the accumulator-imposed order: lane -> column col0 + (l & 15),
// rows ra + 4 G .. +3 then rb + 4 G .. +3
__device__ __forceinline__ abf8 frag_t2(const __bf16* P, int ld, int ra, int rb, int col0, int lane) {
  const int i = lane & 15, G = lane >> 4;
  const int o = (4 * G + (i >> 2)) * ld + col0 + 4 * (i & 3);
  const as4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as4 __attribute__((address_space(3)))*)(P + ra * ld + o));
  const as4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((as4 __attribute__((address_space(3)))*)(P + rb * ld + o));
  const as8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(abf8, v);
}
template <bool ONE>
__device__ __forceinline__ void mfma3(f32x4& acc, const abf8& ah, const abf8& al, const abf8& bh, const abf8& bl) {
  if (!ONE) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}
// acc[j] (rows = rows 16 j .. of the 64-row plane tile, columns = the wave's 16 owned rows) += plane(64 x HDP) . regs^T
template <int NKS, bool ONE>
__device__ __forceinline__ void mma_plane_regs(f32x4 (&acc)[4], const __bf16* Ph, const __bf16* Pl, int ld, const abf8 (&bh)[NKS],
                                               const abf8 (&bl)[NKS], int lane) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const abf8 ah = frag_n(Ph, ld, 16 * j, 32 * ks, lane);
      abf8 al = ah;
      if (!ONE) al = frag_n(Pl, ld, 16 * j, 32 * ks, lane);
      mfma3<ONE>(acc[j], ah, al, bh[ks], bl[ks]);
    }
}
// acc[ct] (rows = head columns 16 ct .., columns = the wave's 16 owned rows) += plane^T(HDP x 64) . x, x = the four accumulator
// tiles xs[j][r] of the 64 streamed rows
template <int NTH, bool ONE>
__device__ __forceinline__ void mma_planeT_acc(f32x4 (&acc)[NTH], const __bf16* Ph, const __bf16* Pl, int ld, const float (&xs)[4][4],
                                               int lane) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    abf8 bh, bl;
    pack_b(xs[2 * p], xs[2 * p + 1], bh, bl);
#pragma unroll
    for (int ct = 0; ct < NTH; ++ct) {
      const abf8 ah = frag_t2(Ph, ld, 32 * p, 32 * p + 16, 16 * ct, lane);
      abf8 al = ah;
      if (!ONE) al = frag_t2(Pl, ld, 32 * p, 32 * p + 16, 16 * ct, lane);
      mfma3<ONE>(acc[ct], ah, al, bh, bl);
    }
  }
}
// sum / max over the four 16-lane rows of the wavefront (lanes l, l^16, l^32, l^48): every lane gets the result
__device__ __forceinline__ float rows4_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
__device__ __forceinline__ float rows4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32)); return v; }
// Dropout keep flags in the TRANSPOSED score layout (a lane: ONE query q, keys kb .. kb+3): the quad of (key, queries 4m .. 4m+3)
// is evaluated once, by the lane whose (l & 3) names the key, and the four lanes of a quad exchange their 4-bit results through
// DPP quad_perm -- the same number of generator calls as the query-major kernels.  Returns bit r = keep (q, kb + r).
__device__ __forceinline__ unsigned keep_bits_t(uint64_t seed, uint32_t site, int bh, int T, int q, int kb, int lane, float p) {
  const float4 u = uniform4(seed, site, attn_quad(bh, T, q, min(kb + (lane & 3), T - 1)));
  const int mine = (u.x >= p ? 1 : 0) | (u.y >= p ? 2 : 0) | (u.z >= p ? 4 : 0) | (u.w >= p ? 8 : 0);   // queries 4m .. 4m+3 of MY key
  const int sh = lane & 3;                                                                           // my query's component
  unsigned bits = 0;
  bits |= ((unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x00, 0xf, 0xf, false) >> sh & 1u) << 0;
  bits |= ((unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x55, 0xf, 0xf, false) >> sh & 1u) << 1;
  bits |= ((unsigned)__builtin_amdgcn_update_dpp(0, mine, 0xaa, 0xf, 0xf, false) >> sh & 1u) << 2;
  bits |= ((unsigned)__builtin_amdgcn_update_dpp(0, mine, 0xff, 0xf, 0xf, false) >> sh & 1u) << 3;
  return bits;
}
// a lane's 4 consecutive head columns 16 ct + 4 G .. of row `dst` (16-byte store when VEC)
template <bool VEC>
__device__ __forceinline__ void store_cols4(float* __restrict__ dst, int c, int hd, const f32x4& v, float s) {
  if (VEC) {
    if (c < hd) *reinterpret_cast<float4*>(dst + c) = make_float4(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c + r < hd) dst[c + r] = v[r] * s;
  }
}

template <int NTH, bool VEC, bool ONE>
__global__ __launch_bounds__(64 * QW) void k_attn_fwd_b16(AttnArgs a) {
  RD_TOUCH_CODE_X(RD_TL_ATTN_FWD_B16, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16, NKS = HDP / 32;
  __bf16* Kh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  uint32_t* mk = reinterpret_cast<uint32_t*>(Vl + TS * LDB);   // key-mask bytes of the current tile (1 = dead)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int qw0 = blockIdx.x * QROWS + wave * 16;             // the wave's first query
  const int q = qw0 + (lane & 15);
  // token plan (round 4): workgroup index b is a RANK, the sample's Tv = len rows sit at row0 .. (AttnRows); keys >= Tv are what
  // the mask would have removed, query / key super-tiles past Tv do not exist
  const AttnRows ar = attn_rows(a, b);
  const int Tv = ar.Tv;
  if ((int)blockIdx.x * QROWS >= Tv) return;                  // (padded layout: Tv == T, never)
  const long rs = ar.rstep * 3 * a.D;
  const float* qb = a.qkv + ar.row0 * 3 * a.D + h * a.hd;
  const uint8_t* mrow = a.mask + (long)b * a.T;
  const int nt = (Tv + TS - 1) / TS;
  // requests: my query row, key tile 0 (threads 0..255: K, 256..511: V; processed whether live or not -- a dead tile is an identity)
  RawFrag<NKS> qraw;
  raw_load<NKS, VEC>(qraw, qb + (long)min(q, Tv - 1) * rs, a.hd, G);
  const bool isv = tid >= 256;
  const int lt = tid & 255;
  const float* kvb = qb + (isv ? 2 : 1) * a.D;
  HeadRegs<NTH> kvr;
  head_load_t<NTH, VEC>(kvr, kvb, rs, 0, Tv, a.hd, lt);
  uint8_t mbyte = a.plan ? (uint8_t)0 : mrow[min(tid & 63, Tv - 1)];
  const uint64_t live = a.plan ? ~0ull : (live_key_tiles(mrow, Tv, lane) | 1ull);
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  abf8 qh[NKS], ql[NKS];
  raw_zero<NKS>(qraw, q < Tv, a.hd, G);
  raw_split<NKS>(qraw, qh, ql);
  float m_i = -INFINITY, l_i = 0.f;                            // l_i: this lane's keys only; combined over the 4 lane rows at the end
  f32x4 o[NTH];                                               // O^T: rows = head columns, column = my query
#pragma unroll
  for (int ct = 0; ct < NTH; ++ct) o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int kt = 0;
  while (kt < nt) {
    const int k0 = kt * TS;
    lds_barrier();                                            // the previous tile's products have read the planes
    head_store_b16<NTH>(kvr, isv ? Vh : Kh, isv ? Vl : Kl, LDB, HDP, lt);
    if (tid < TS) reinterpret_cast<uint8_t*>(mk)[tid] = (k0 + tid < Tv) ? mbyte : (uint8_t)1;
    const int kn = next_live_tile(live, kt + 1, nt);
    if (kn < nt) {                                            // in flight during this tile's products
      head_load_t<NTH, VEC>(kvr, kvb, rs, kn * TS, Tv, a.hd, lt);
      if (!a.plan) mbyte = mrow[min(kn * TS + (tid & 63), Tv - 1)];
    }
    lds_barrier();
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma_plane_regs<NKS, ONE>(s, Kh, Kl, LDB, qh, ql, lane);    // S^T = K Q^T: rows = keys 16 j + 4 G + r, column = my query
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t dm = mk[4 * j + G];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[j][r] = ((dm >> (8 * r)) & 0xffu) ? -INFINITY : s[j][r] * a.scale;
        mx = fmaxf(mx, s[j][r]);
      }
    }
    const float mn = fmaxf(m_i, rows4_max(mx));
    const float alpha = (m_i == -INFINITY) ? 0.f : expf(m_i - mn);
    m_i = mn;
    float rsum = 0.f, pm[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned kb = 0xfu;
      if (a.p_drop > 0.f) kb = keep_bits_t(seedv, a.site, bh, a.T, q, k0 + 16 * j + 4 * G, lane, a.p_drop);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = (s[j][r] == -INFINITY) ? 0.f : __expf(s[j][r] - m_i);
        rsum += p;
        pm[j][r] = ((kb >> r) & 1u) ? (a.p_drop > 0.f ? p * inv_keep : p) : 0.f;   // F.dropout on the probabilities (after the softmax sum)
      }
    }
    l_i = l_i * alpha + rsum;
#pragma unroll
    for (int ct = 0; ct < NTH; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[ct][r] *= alpha;
    mma_planeT_acc<NTH, ONE>(o, Vh, Vl, LDB, pm, lane);        // O^T += V^T (P o M)^T
    kt = kn;
  }
  const float l = rows4_sum(l_i);
  if (q < Tv) {
    const float inv = 1.0f / l;
    float* orow = a.out + (ar.row0 + (long)q * ar.rstep) * a.D + h * a.hd;
#pragma unroll
    for (int ct = 0; ct < NTH; ++ct) store_cols4<VEC>(orow, 16 * ct + 4 * G, a.hd, o[ct], inv);
    if (G == 0) a.lse[(long)bh * a.T + q] = m_i + logf(l);
  }
}

// dQ (and delta = rowsum(dO * O)): grid (query super-tiles, B*H); a wave owns 16 queries
template <int NTH, bool VEC, bool ONE>
__global__ __launch_bounds__(64 * QW) void k_attn_bwd_dq_b16(AttnArgs a) {
  RD_TOUCH_CODE_X(RD_TL_ATTN_BWD_DQ, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16, NKS = HDP / 32;
  __bf16* Kh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Kl = Kh + TS * LDB;
  __bf16* Vh = Kl + TS * LDB;
  __bf16* Vl = Vh + TS * LDB;
  uint32_t* mk = reinterpret_cast<uint32_t*>(Vl + TS * LDB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int qw0 = blockIdx.x * QROWS + wave * 16;
  const int q = qw0 + (lane & 15);
  const AttnRows ar = attn_rows(a, b);                        // token plan: see k_attn_fwd_b16
  const int Tv = ar.Tv;
  if ((int)blockIdx.x * QROWS >= Tv) return;
  const int qc = min(q, Tv - 1);
  const long rs = ar.rstep * 3 * a.D, ro = ar.rstep * a.D;
  const float* qb = a.qkv + ar.row0 * 3 * a.D + h * a.hd;
  const uint8_t* mrow = a.mask + (long)b * a.T;
  const int nt = (Tv + TS - 1) / TS;
  RawFrag<NKS> qraw, doraw, oraw;
  raw_load<NKS, VEC>(qraw, qb + (long)qc * rs, a.hd, G);
  raw_load<NKS, VEC>(doraw, a.dout + ar.row0 * a.D + h * a.hd + (long)qc * ro, a.hd, G);
  raw_load<NKS, VEC>(oraw, a.out + ar.row0 * a.D + h * a.hd + (long)qc * ro, a.hd, G);
  const float lse_q = a.lse[(long)bh * a.T + qc];
  const bool isv = tid >= 256;
  const int lt = tid & 255;
  const float* kvb = qb + (isv ? 2 : 1) * a.D;
  HeadRegs<NTH> kvr;
  head_load_t<NTH, VEC>(kvr, kvb, rs, 0, Tv, a.hd, lt);
  uint8_t mbyte = a.plan ? (uint8_t)0 : mrow[min(tid & 63, Tv - 1)];
  const uint64_t live = a.plan ? ~0ull : (live_key_tiles(mrow, Tv, lane) | 1ull);
  uint64_t seedv = a.seed;
  if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  abf8 qh[NKS], ql[NKS], gh[NKS], gl[NKS];                     // Q and dO rows of my query, as B operands
  const bool qok = q < Tv;
  raw_zero<NKS>(qraw, qok, a.hd, G);
  raw_zero<NKS>(doraw, qok, a.hd, G);
  raw_zero<NKS>(oraw, qok, a.hd, G);
  float dsum = 0.f;                                           // delta = rowsum(dO * O), in fp32
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float* x = reinterpret_cast<const float*>(&doraw.v[ks][0]);
    const float* y = reinterpret_cast<const float*>(&oraw.v[ks][0]);
#pragma unroll
    for (int u = 0; u < 8; ++u) dsum += x[u] * y[u];
  }
  const float dl_q = rows4_sum(dsum);
  if (qok && G == 0) a.delta[(long)bh * a.T + q] = dl_q;
  raw_split<NKS>(qraw, qh, ql);
  raw_split<NKS>(doraw, gh, gl);
  f32x4 dq[NTH];                                              // dQ^T: rows = head columns, column = my query
#pragma unroll
  for (int ct = 0; ct < NTH; ++ct) dq[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int kt = 0;
  while (kt < nt) {
    const int k0 = kt * TS;
    lds_barrier();
    head_store_b16<NTH>(kvr, isv ? Vh : Kh, isv ? Vl : Kl, LDB, HDP, lt);
    if (tid < TS) reinterpret_cast<uint8_t*>(mk)[tid] = (k0 + tid < Tv) ? mbyte : (uint8_t)1;
    const int kn = next_live_tile(live, kt + 1, nt);
    if (kn < nt) {
      head_load_t<NTH, VEC>(kvr, kvb, rs, kn * TS, Tv, a.hd, lt);
      if (!a.plan) mbyte = mrow[min(kn * TS + (tid & 63), Tv - 1)];
    }
    lds_barrier();
    f32x4 s[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
    mma_plane_regs<NKS, ONE>(s, Kh, Kl, LDB, qh, ql, lane);    // S^T  = K Q^T
    mma_plane_regs<NKS, ONE>(dp, Vh, Vl, LDB, gh, gl, lane);   // dP^T = V dO^T
    float ds[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t dm = mk[4 * j + G];
      unsigned kb = 0xfu;
      if (a.p_drop > 0.f) kb = keep_bits_t(seedv, a.site, bh, a.T, q, k0 + 16 * j + 4 * G, lane, a.p_drop);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ds[j][r] = 0.f;
        if (!((dm >> (8 * r)) & 0xffu) && qok) {
          const float p = __expf(s[j][r] * a.scale - lse_q);
          const float k = ((kb >> r) & 1u) ? (a.p_drop > 0.f ? inv_keep : 1.f) : 0.f;
          ds[j][r] = p * (dp[j][r] * k - dl_q) * a.scale;
        }
      }
    }
    mma_planeT_acc<NTH, ONE>(dq, Kh, Kl, LDB, ds, lane);       // dQ^T += K^T dS^T
    kt = kn;
  }
  if (qok) {
    float* row = a.dqkv + (ar.row0 + (long)q * ar.rstep) * 3 * a.D + h * a.hd;
#pragma unroll
    for (int ct = 0; ct < NTH; ++ct) store_cols4<VEC>(row, 16 * ct + 4 * G, a.hd, dq[ct], 1.f);
  }
}

// dK / dV: grid (key super-tiles, B*H); a wave owns 16 keys, K and V rows in registers, the Q / dO tiles stream through LDS.
// A workgroup whose 128 keys are all dead writes zeros and leaves; a wave whose 16 keys are all dead only helps with the tiles.
template <int NTH, bool VEC, bool ONE>
__global__ __launch_bounds__(64 * QW) void k_attn_bwd_dkv_b16(AttnArgs a) {
  RD_TOUCH_CODE_X(RD_TL_ATTN_BWD_DKV, blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y, 512);
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16, NKS = HDP / 32;
  __bf16* Qh = reinterpret_cast<__bf16*>(bsm);
  __bf16* Ql = Qh + TS * LDB;
  __bf16* Oh = Ql + TS * LDB;                       // dO
  __bf16* Ol = Oh + TS * LDB;
  float* lse_s = reinterpret_cast<float*>(Ol + TS * LDB);
  float* dl_s = lse_s + TS;
  __shared__ int any_live_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4;
  const int bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int key = blockIdx.x * QROWS + wave * 16 + (lane & 15);
  const AttnRows ar = attn_rows(a, b);                        // token plan: see k_attn_fwd_b16 (rows >= Tv do not exist: nothing to zero)
  const int Tv = ar.Tv;
  if ((int)blockIdx.x * QROWS >= Tv) return;
  const int kc = min(key, Tv - 1);
  const long rs = ar.rstep * 3 * a.D, ro = ar.rstep * a.D;
  const float* qb = a.qkv + ar.row0 * 3 * a.D + h * a.hd;
  const float* dob = a.dout + ar.row0 * a.D + h * a.hd;
  const uint8_t* mrow = a.mask + (long)b * a.T;
  RawFrag<NKS> kraw, vraw;
  raw_load<NKS, VEC>(kraw, qb + a.D + (long)kc * rs, a.hd, G);
  raw_load<NKS, VEC>(vraw, qb + 2 * a.D + (long)kc * rs, a.hd, G);
  const uint8_t mkey = a.plan ? (uint8_t)0 : mrow[kc];
  // first query tile (threads 0..255: Q, 256..511: dO), lse / delta of its rows (threads 0..63)
  const bool isg = tid >= 256;
  const int lt = tid & 255;
  const float* tb = isg ? dob : qb;
  const long ts = isg ? ro : rs;
  HeadRegs<NTH> tr;
  head_load_t<NTH, VEC>(tr, tb, ts, 0, Tv, a.hd, lt);
  float lq = a.lse[(long)bh * a.T + min(tid & 63, Tv - 1)], dlq = a.delta[(long)bh * a.T + min(tid & 63, Tv - 1)];
  const bool dead = key >= Tv || mkey;
  const bool wave_live = __ballot(!dead) != 0ull;
  if (tid == 0) any_live_s = 0;
  __syncthreads();
  if (wave_live && lane == 0) any_live_s = 1;
  __syncthreads();
  const bool any_live = any_live_s != 0;
  f32x4 dk[NTH], dv[NTH];                                     // dK^T, dV^T: rows = head columns, column = my key
#pragma unroll
  for (int ct = 0; ct < NTH; ++ct) { dk[ct] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[ct] = dk[ct]; }
  if (any_live) {
    abf8 kh[NKS], kl[NKS], vh[NKS], vl[NKS];
    raw_zero<NKS>(kraw, key < Tv, a.hd, G);
    raw_zero<NKS>(vraw, key < Tv, a.hd, G);
    raw_split<NKS>(kraw, kh, kl);
    raw_split<NKS>(vraw, vh, vl);
    uint64_t seedv = a.seed;
    if (a.p_drop > 0.f && a.seed_cell) seedv += load_uniform_u64(a.seed_cell);
    const float inv_keep = 1.0f / (1.0f - a.p_drop);
    for (int q0 = 0; q0 < Tv; q0 += TS) {
      lds_barrier();                                          // the previous query tile's products have read the planes
      head_store_b16<NTH>(tr, isg ? Oh : Qh, isg ? Ol : Ql, LDB, HDP, lt);
      if (tid < TS) { lse_s[tid] = lq; dl_s[tid] = dlq; }
      if (q0 + TS < Tv) {                                     // next query tile, in flight during this one's products
        head_load_t<NTH, VEC>(tr, tb, ts, q0 + TS, Tv, a.hd, lt);
        lq = a.lse[(long)bh * a.T + min(q0 + TS + (tid & 63), Tv - 1)];
        dlq = a.delta[(long)bh * a.T + min(q0 + TS + (tid & 63), Tv - 1)];
      }
      lds_barrier();
      if (!wave_live) continue;                               // uniform per wave; the barriers above are still met
      f32x4 s[4], dp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[j] = s[j]; }
      mma_plane_regs<NKS, ONE>(s, Qh, Ql, LDB, kh, kl, lane);  // S  = Q K^T: rows = queries 16 j + 4 G + r, column = my key
      mma_plane_regs<NKS, ONE>(dp, Oh, Ol, LDB, vh, vl, lane); // dP = dO V^T
      float pm[4][4], ds[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qj = q0 + 16 * j + 4 * G;
        float k4[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.p_drop > 0.f) attn_keep4(k4, seedv, a.site, bh, a.T, qj, kc, a.p_drop, inv_keep);
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + 16 * j + 4 * G);
        const float4 d4 = *reinterpret_cast<const float4*>(dl_s + 16 * j + 4 * G);
        const float lr[4] = {l4.x, l4.y, l4.z, l4.w}, dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pm[j][r] = 0.f; ds[j][r] = 0.f;
          if (!dead && qj + r < Tv) {
            const float p = __expf(s[j][r] * a.scale - lr[r]);
            pm[j][r] = p * k4[r];
            ds[j][r] = p * (dp[j][r] * k4[r] - dr[r]) * a.scale;
          }
        }
      }
      mma_planeT_acc<NTH, ONE>(dv, Oh, Ol, LDB, pm, lane);     // dV^T += dO^T (P o M)
      mma_planeT_acc<NTH, ONE>(dk, Qh, Ql, LDB, ds, lane);     // dK^T += Q^T dS
    }
  }
  if (key < Tv) {
    float* row = a.dqkv + (ar.row0 + (long)key * ar.rstep) * 3 * a.D + h * a.hd;
#pragma unroll
    for (int ct = 0; ct < NTH; ++ct) {
      store_cols4<VEC>(row + a.D, 16 * ct + 4 * G, a.hd, dk[ct], 1.f);
      store_cols4<VEC>(row + 2 * a.D, 16 * ct + 4 * G, a.hd, dv[ct], 1.f);
    }
  }
}

// 16-byte head-tile loads are legal: every row of every head starts on a 16-byte boundary.  Decided HERE and compiled into the
// kernel (template flag): as a runtime flag the compiler merged the two load forms into four 4-byte loads per chunk -- 4x the
// vector-memory instructions of every attention kernel.
static bool attn_vec_ok(const AttnArgs& a) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return (a.hd & 3) == 0 && (a.D & 3) == 0 && al(a.qkv) && al(a.out) && al(a.dout);
}
#define ATTN_LAUNCH(K, grid, block, lds_attr, lds, ...)                                                  \
  do {                                                                                                   \
    if (vec) {                                                                                           \
      RD_LDS_ATTR((K<NTH, true>), lds_attr);                                                             \
      hipLaunchKernelGGL((K<NTH, true>), grid, block, lds, st, __VA_ARGS__);                             \
    } else {                                                                                             \
      RD_LDS_ATTR((K<NTH, false>), lds_attr);                                                            \
      hipLaunchKernelGGL((K<NTH, false>), grid, block, lds, st, __VA_ARGS__);                            \
    }                                                                                                    \
  } while (0)

static bool attn_b16_mt_ok(const AttnArgs& a) {
  const char* e = getenv("RD_ATTN_B16_MT");           // read per call (tests compare both paths in one process)
  return !(e && atoi(e) == 0) && precision() != RD_PREC_FP32 && a.T > TS && a.hd <= 96;
}
#define ATTN_LAUNCH_MT(K, grid, lds, arg)                                                                \
  do {                                                                                                   \
    if (vec && one) { RD_LDS_ATTR((K<NTH, true, true>), lds); hipLaunchKernelGGL((K<NTH, true, true>), grid, dim3(64 * QW), lds, st, arg); }          \
    else if (vec) { RD_LDS_ATTR((K<NTH, true, false>), lds); hipLaunchKernelGGL((K<NTH, true, false>), grid, dim3(64 * QW), lds, st, arg); }         \
    else if (one) { RD_LDS_ATTR((K<NTH, false, true>), lds); hipLaunchKernelGGL((K<NTH, false, true>), grid, dim3(64 * QW), lds, st, arg); }         \
    else { RD_LDS_ATTR((K<NTH, false, false>), lds); hipLaunchKernelGGL((K<NTH, false, false>), grid, dim3(64 * QW), lds, st, arg); }               \
  } while (0)
template <int NTH>
int launch_attn_b16_mt(const AttnArgs& a, int which, hipStream_t st) {
  const bool vec = attn_vec_ok(a);
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16;
  const bool one = precision() == RD_PREC_BF16;
  const dim3 grid(cdiv(a.T, QROWS), a.B * a.H);
  const size_t planes = (size_t)4 * TS * LDB * sizeof(__bf16);
  if (which == 0) {
    ATTN_LAUNCH_MT(k_attn_fwd_b16, grid, planes + TS, a);
    return check_launch("k_attn_fwd_b16");
  }
  if (which == 1) {
    ATTN_LAUNCH_MT(k_attn_bwd_dq_b16, grid, planes + TS, a);
    return check_launch("k_attn_bwd_dq_b16");
  }
  ATTN_LAUNCH_MT(k_attn_bwd_dkv_b16, grid, planes + 2 * TS * sizeof(float), a);
  return check_launch("k_attn_bwd_dkv_b16");
}

static bool attn_b16_ok(const AttnArgs& a) {
  const char* e = getenv("RD_ATTN_B16");              // read per call (tests compare both paths in one process)
  return !(e && atoi(e) == 0) && precision() != RD_PREC_FP32 && a.T <= TS && a.hd <= 96;
}
template <int NTH>
int launch_attn_b16(const AttnArgs& a_in, int which, hipStream_t st) {
  const bool vec = attn_vec_ok(a_in);
  constexpr int HDP = 32 * ((16 * NTH + 31) / 32), LDB = HDP + 16;
  const int one = precision() == RD_PREC_BF16;
  AttnArgs a = a_in;
  a.stamps = g_attn_stamps;
  if (which == 0) {
    const size_t lds = (size_t)(6 * TS * LDB + (LDB >= LDT ? 0 : 2 * TS * LDT)) * sizeof(__bf16);
    static const bool widef = [] { const char* e = getenv("RD_ATTN_FWD_W8"); return !(e && atoi(e) == 0); }();
    if (widef || a.plan) {
      const size_t ldsw = lds + 4 * TS * sizeof(float);
      ATTN_LAUNCH(k_attn_fwd_one_b16w, dim3(a.B * a.H), dim3(512), ldsw, ldsw, a, one);
      return check_launch("k_attn_fwd_one_b16w");
    }
    ATTN_LAUNCH(k_attn_fwd_one_b16, dim3(a.B * a.H), dim3(256), lds, lds, a, one);
    return check_launch("k_attn_fwd_one_b16");
  }
  const size_t lds = (size_t)(8 * TS * LDB + 4 * TS * LDT) * sizeof(__bf16) + 2 * TS * sizeof(float);
  static const bool wide = [] { const char* e = getenv("RD_ATTN_BWD_W8"); return !(e && atoi(e) == 0); }();
  if (wide || a.plan) {
    ATTN_LAUNCH(k_attn_bwd_one_b16w, dim3(a.B * a.H), dim3(512), lds, lds, a, one);
    return check_launch("k_attn_bwd_one_b16w");
  }
  ATTN_LAUNCH(k_attn_bwd_one_b16, dim3(a.B * a.H), dim3(256), lds, lds, a, one);
  return check_launch("k_attn_bwd_one_b16");
}

template <int NTH>
int launch_attn(const AttnArgs& a, int which, hipStream_t st) {
  const bool vec = attn_vec_ok(a);
  constexpr int LDH = 16 * NTH + 4;
  dim3 grid(cdiv(a.T, TS), a.B * a.H);
  size_t lds;
  if (which == 0) {
    lds = sizeof(float) * (3 * TS * LDH + TS * LDP);
    const size_t lds_attr = lds;
    if (a.T <= TS && LDH >= LDP) lds = sizeof(float) * (3 * TS * LDH);   // P overlays Q
    ATTN_LAUNCH(k_attn_fwd, grid, dim3(256), lds_attr, lds, a);
    return check_launch("k_attn_fwd");
  } else if (which == 1) {
    lds = sizeof(float) * (4 * TS * LDH + TS * LDP + 2 * TS);
    ATTN_LAUNCH(k_attn_bwd_dq, grid, dim3(256), lds, lds, a);
    return check_launch("k_attn_bwd_dq");
  }
  if (which == 3) {
    lds = sizeof(float) * (4 * TS * LDH + 2 * TS * LDP + 2 * TS);
    ATTN_LAUNCH(k_attn_bwd_one, dim3(a.B * a.H), dim3(256), lds, lds, a);
    return check_launch("k_attn_bwd_one");
  }
  lds = sizeof(float) * (4 * TS * LDH + 2 * TS * LDP + 2 * TS);
  ATTN_LAUNCH(k_attn_bwd_dkv, grid, dim3(256), lds, lds, a);
  return check_launch("k_attn_bwd_dkv");
}

// ------------------------------------------------------------------------------------------------
// Wide heads (head_dim > 96: the 256-sensor stress shape, D = 1040, 2 heads of 520): the score matrix is MATERIALISED.
// At head_dim 520 and T = 512 the products are large square GEMMs (MFMA-bound), so the flash-style
// tiling above buys nothing: S = scale Q K^T, P = softmax(S + key mask), O = dropout(P) V run as batched tiled GEMMs
// (one problem per (sample, head), rd_gemm.hip) around a row-softmax kernel; P and dropout(P) are saved for the backward
// (2 B H T^2 floats).  Same masks and the same Philox quads (attn_quad) as the tiled kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_softmax_rows(float* __restrict__ S, float* __restrict__ PD, const uint8_t* __restrict__ mask,
                                                      int T, int H, long rows, float p_drop, uint64_t seed, uint32_t site,
                                                      const uint64_t* cell) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  seed = eff_seed(seed, cell);
  const int bh = (int)(row / T), q = (int)(row - (long)bh * T), b = bh / H;
  float* s = S + row * T;
  float m = -INFINITY;
  for (int k = lane; k < T; k += 64)
    if (!mask[(long)b * T + k]) m = fmaxf(m, s[k]);
  m = wave_max64(m);
  float sum = 0.f;
  for (int k = lane; k < T; k += 64)
    if (!mask[(long)b * T + k]) sum += __expf(s[k] - m);
  sum = wave_sum64(sum);
  const float inv = 1.0f / sum, inv_keep = 1.0f / (1.0f - p_drop);
  for (int k = lane; k < T; k += 64) {
    const float p = mask[(long)b * T + k] ? 0.f : __expf(s[k] - m) * inv;
    s[k] = p;
    PD[row * T + k] = p_drop > 0.f ? p * attn_keep1(seed, site, bh, T, q, k, p_drop, inv_keep) : p;
  }
}
// dS = P o (dP - rowsum(P o dP)) * scale with dP = dPD o keep, in place over dPD
__global__ __launch_bounds__(256) void k_softmax_bwd_rows(const float* __restrict__ P, float* __restrict__ dPD, int T, long rows,
                                                          float scale, float p_drop, uint64_t seed, uint32_t site,
                                                          const uint64_t* cell) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= rows) return;
  seed = eff_seed(seed, cell);
  const int bh = (int)(row / T), q = (int)(row - (long)bh * T);
  const float* p = P + row * T;
  float* d = dPD + row * T;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  float delta = 0.f;
  for (int k = lane; k < T; k += 64) {
    float dp = d[k];
    if (p_drop > 0.f) dp *= attn_keep1(seed, site, bh, T, q, k, p_drop, inv_keep);
    d[k] = dp;
    delta += p[k] * dp;
  }
  delta = wave_sum64(delta);
  for (int k = lane; k < T; k += 64) d[k] = p[k] * (d[k] - delta) * scale;
}

static GemmArgs bh_gemm(const AttnArgs& a, int M, int N, int K) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.nsplit = 1; g.nbatch = a.B * a.H; g.batch_inner = a.H;
  return g;
}
// P, PD: [B*H][T][T] each (saved)
int attn_big_fwd(const AttnArgs& a, float* P, float* PD, hipStream_t st) {
  const long rs = (long)a.B * 3 * a.D, TT = (long)a.T * a.T;
  int rc;
  GemmArgs g = bh_gemm(a, a.T, a.T, a.hd);                                  // S = scale Q K^T
  g.A = a.qkv; g.sa_m = rs; g.sa_k = 1; g.a_bo = 3 * a.D; g.a_bi = a.hd;
  g.B = a.qkv + a.D; g.sb_n = rs; g.sb_k = 1; g.b_bo = 3 * a.D; g.b_bi = a.hd;
  g.C = P; g.sc_m = a.T; g.c_bo = (long)a.H * TT; g.c_bi = TT; g.cscale = a.scale;
  if ((rc = launch_gemm(g, st))) return rc;
  const long rows = (long)a.B * a.H * a.T;
  hipLaunchKernelGGL(k_softmax_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, PD, a.mask, a.T, a.H, rows, a.p_drop,
                     a.seed, a.site, a.seed_cell);
  if ((rc = check_launch("k_softmax_rows"))) return rc;
  g = bh_gemm(a, a.T, a.hd, a.T);                                           // O = PD V
  g.A = PD; g.sa_m = a.T; g.sa_k = 1; g.a_bo = (long)a.H * TT; g.a_bi = TT;
  g.B = a.qkv + 2 * a.D; g.sb_n = 1; g.sb_k = rs; g.b_bo = 3 * a.D; g.b_bi = a.hd;
  g.C = a.out; g.sc_m = (long)a.B * a.D; g.c_bo = a.D; g.c_bi = a.hd;
  return launch_gemm(g, st);
}
// dS: [B*H][T][T] workspace
int attn_big_bwd(const AttnArgs& a, const float* P, const float* PD, float* dS, hipStream_t st) {
  const long rs = (long)a.B * 3 * a.D, ro = (long)a.B * a.D, TT = (long)a.T * a.T;
  int rc;
  GemmArgs g = bh_gemm(a, a.T, a.T, a.hd);                                  // dPD = dO V^T
  g.A = a.dout; g.sa_m = ro; g.sa_k = 1; g.a_bo = a.D; g.a_bi = a.hd;
  g.B = a.qkv + 2 * a.D; g.sb_n = rs; g.sb_k = 1; g.b_bo = 3 * a.D; g.b_bi = a.hd;
  g.C = dS; g.sc_m = a.T; g.c_bo = (long)a.H * TT; g.c_bi = TT;
  if ((rc = launch_gemm(g, st))) return rc;
  g = bh_gemm(a, a.T, a.hd, a.T);                                           // dV = PD^T dO
  g.A = PD; g.sa_m = 1; g.sa_k = a.T; g.a_bo = (long)a.H * TT; g.a_bi = TT;
  g.B = a.dout; g.sb_n = 1; g.sb_k = ro; g.b_bo = a.D; g.b_bi = a.hd;
  g.C = a.dqkv + 2 * a.D; g.sc_m = rs; g.c_bo = 3 * a.D; g.c_bi = a.hd;
  if ((rc = launch_gemm(g, st))) return rc;
  const long rows = (long)a.B * a.H * a.T;
  hipLaunchKernelGGL(k_softmax_bwd_rows, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dS, a.T, rows, a.scale, a.p_drop,
                     a.seed, a.site, a.seed_cell);
  if ((rc = check_launch("k_softmax_bwd_rows"))) return rc;
  g = bh_gemm(a, a.T, a.hd, a.T);                                           // dQ = dS K
  g.A = dS; g.sa_m = a.T; g.sa_k = 1; g.a_bo = (long)a.H * TT; g.a_bi = TT;
  g.B = a.qkv + a.D; g.sb_n = 1; g.sb_k = rs; g.b_bo = 3 * a.D; g.b_bi = a.hd;
  g.C = a.dqkv; g.sc_m = rs; g.c_bo = 3 * a.D; g.c_bi = a.hd;
  if ((rc = launch_gemm(g, st))) return rc;
  g = bh_gemm(a, a.T, a.hd, a.T);                                           // dK = dS^T Q
  g.A = dS; g.sa_m = 1; g.sa_k = a.T; g.a_bo = (long)a.H * TT; g.a_bi = TT;
  g.B = a.qkv; g.sb_n = 1; g.sb_k = rs; g.b_bo = 3 * a.D; g.b_bi = a.hd;
  g.C = a.dqkv + a.D; g.sc_m = rs; g.c_bo = 3 * a.D; g.c_bi = a.hd;
  return launch_gemm(g, st);
}

int dispatch_attn(const AttnArgs& a, int which, hipStream_t st) {
  if (a.plan && !(((which == 0 || which == 3) && attn_b16_ok(a)) || (which <= 2 && attn_b16_mt_ok(a))))
    return fail(RD_EUNSUPPORTED, "token plan: only the split-bf16 attention kernels (head_dim <= 96, bf16 modes) read it");
  if ((which == 0 || which == 3) && attn_b16_ok(a)) {
    switch (cdiv(a.hd, 16)) {
      case 1: return launch_attn_b16<1>(a, which, st);
      case 2: return launch_attn_b16<2>(a, which, st);
      case 3: return launch_attn_b16<3>(a, which, st);
      case 4: return launch_attn_b16<4>(a, which, st);
      case 5: return launch_attn_b16<5>(a, which, st);
      default: return launch_attn_b16<6>(a, which, st);
    }
  }
  if (which <= 2 && attn_b16_mt_ok(a)) {
    switch (cdiv(a.hd, 16)) {
      case 1: return launch_attn_b16_mt<1>(a, which, st);
      case 2: return launch_attn_b16_mt<2>(a, which, st);
      case 3: return launch_attn_b16_mt<3>(a, which, st);
      case 4: return launch_attn_b16_mt<4>(a, which, st);
      case 5: return launch_attn_b16_mt<5>(a, which, st);
      default: return launch_attn_b16_mt<6>(a, which, st);
    }
  }
  switch (cdiv(a.hd, 16)) {
    case 1: return launch_attn<1>(a, which, st);
    case 2: return launch_attn<2>(a, which, st);
    case 3: return launch_attn<3>(a, which, st);
    case 4: return launch_attn<4>(a, which, st);
    case 5: return launch_attn<5>(a, which, st);
    case 6: return launch_attn<6>(a, which, st);
    default: return fail(RD_EUNSUPPORTED, "attention head_dim %d > 96 not built", a.hd);
  }
}

// ------------------------------------------------------------------------------------------------
// s = x + dropout(r);  y = LayerNorm(s) * g + b.   One wavefront per row; three passes over the
// row (each lane re-reads only what it wrote), exact two-pass variance like torch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_add_ln_fwd(const float* __restrict__ x, const float* __restrict__ r,
                                                    const float* __restrict__ g, const float* __restrict__ bta,
                                                    float* __restrict__ s_out, float* __restrict__ y,
                                                    float* __restrict__ stats, int M, int D, float p_drop,
                                                    uint64_t seed, uint32_t site, const uint64_t* cell) {
  seed = eff_seed(seed, cell);
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (row >= M) return;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  const float* xr = x + row * D; const float* rr = r + row * D;
  float* sr = s_out + row * D; float* yr = y + row * D;
  float sum = 0.f;
  for (int c = lane; c < D; c += 64) {
    float rv = rr[c];
    if (p_drop > 0.f) rv *= dropout_scale(seed, site, (uint64_t)row * D + c, p_drop, inv_keep);
    const float s = xr[c] + rv;
    sr[c] = s;
    sum += s;
  }
  const float mean = wave_sum64(sum) / D;
  float var = 0.f;
  for (int c = lane; c < D; c += 64) { const float d = sr[c] - mean; var += d * d; }
  const float rstd = rsqrtf(wave_sum64(var) / D + 1e-5f);
  for (int c = lane; c < D; c += 64) yr[c] = (sr[c] - mean) * rstd * g[c] + bta[c];
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// register form for D <= 64*NV: two rows per wavefront, every load of both rows issued before the first
// use (one memory round trip per wave instead of three dependent passes per row); the row lives in VGPRs
// between the passes.  Same per-lane summation order as the generic kernel above.
constexpr int LNF_RPW = 2;
template <int NV>
__global__ __launch_bounds__(256) void k_add_ln_fwd_r(const float* __restrict__ x, const float* __restrict__ r,
                                                      const float* __restrict__ g, const float* __restrict__ bta,
                                                      float* __restrict__ s_out, float* __restrict__ y,
                                                      float* __restrict__ stats, int M, int D, float p_drop,
                                                      uint64_t seed, uint32_t site, const uint64_t* cell) {
  seed = eff_seed(seed, cell);
  const int lane = threadIdx.x & 63;
  const long row0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * LNF_RPW;
  if (row0 >= M) return;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  float xv[LNF_RPW][NV], rv[LNF_RPW][NV], gg[NV], bb[NV];
#pragma unroll
  for (int q = 0; q < LNF_RPW; ++q)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool ok = (row0 + q < M) && c < D;
      xv[q][i] = ok ? x[(row0 + q) * D + c] : 0.f;
      rv[q][i] = ok ? r[(row0 + q) * D + c] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    gg[i] = c < D ? g[c] : 0.f; bb[i] = c < D ? bta[c] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < LNF_RPW; ++q) {
    const long row = row0 + q;
    if (row >= M) break;
    float sv[NV], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      sv[i] = 0.f;
      if (c < D) {
        float t = rv[q][i];
        if (p_drop > 0.f) t *= dropout_scale(seed, site, (uint64_t)row * D + c, p_drop, inv_keep);
        sv[i] = xv[q][i] + t;
        s_out[row * D + c] = sv[i];
        sum += sv[i];
      }
    }
    const float mean = wave_sum64(sum) / D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + 64 * i < D) { const float d = sv[i] - mean; var += d * d; }
    const float rstd = rsqrtf(wave_sum64(var) / D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < D) y[row * D + c] = (sv[i] - mean) * rstd * gg[i] + bb[i];
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// float4 form for D % 4 == 0, D <= 256: lane l owns columns 4l..4l+3 of a row, so the four keep decisions
// of an element quad come from ONE Philox evaluation (the scalar-column kernels evaluate it once per
// element, 4x redundantly, and are bound by those integer multiplies, not by memory); 16-byte accesses,
// LNV_RPW rows per wavefront with every load issued before the first use.  Same masks as the other forms.
constexpr int LNV_RPW = 4;
__global__ __launch_bounds__(256) void k_add_ln_fwd_v(const float* __restrict__ x, const float* __restrict__ r,
                                                      const float* __restrict__ g, const float* __restrict__ bta,
                                                      float* __restrict__ s_out, float* __restrict__ y,
                                                      float* __restrict__ stats, int M, int D, float p_drop,
                                                      uint64_t seed, uint32_t site, const uint64_t* cell) {
  RD_TOUCH_CODE_X(RD_TL_ADD_LN_FWD, blockIdx.x, 512);
  seed = eff_seed(seed, cell);
  const int lane = threadIdx.x & 63;
  const long row0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * LNV_RPW;
  if (row0 >= M) return;
  const int c = 4 * lane;
  const bool cok = c < D;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 xv[LNV_RPW], rv[LNV_RPW];
#pragma unroll
  for (int q = 0; q < LNV_RPW; ++q) {
    const bool ok = cok && (row0 + q < M);
    xv[q] = zero4; rv[q] = zero4;              // (`ok ? *p : zero4` compiles to a pointer select + flat load from scratch)
    if (ok) {
      xv[q] = *reinterpret_cast<const float4*>(x + (row0 + q) * D + c);
      rv[q] = *reinterpret_cast<const float4*>(r + (row0 + q) * D + c);
    }
  }
  float4 gg = zero4, bb = zero4;
  if (cok) { gg = *reinterpret_cast<const float4*>(g + c); bb = *reinterpret_cast<const float4*>(bta + c); }
#pragma unroll
  for (int q = 0; q < LNV_RPW; ++q) {
    const long row = row0 + q;
    if (row >= M) break;
    float4 t = rv[q];
    if (p_drop > 0.f) {
      const float4 u = uniform4(seed, site, ((uint64_t)row * D + c) >> 2);
      t.x *= u.x >= p_drop ? inv_keep : 0.f; t.y *= u.y >= p_drop ? inv_keep : 0.f;
      t.z *= u.z >= p_drop ? inv_keep : 0.f; t.w *= u.w >= p_drop ? inv_keep : 0.f;
    }
    const float4 sv = make_float4(xv[q].x + t.x, xv[q].y + t.y, xv[q].z + t.z, xv[q].w + t.w);   // 0 beyond D
    if (cok) *reinterpret_cast<float4*>(s_out + row * D + c) = sv;
    const float mean = wave_sum64((sv.x + sv.y) + (sv.z + sv.w)) / D;
    float4 d = make_float4(sv.x - mean, sv.y - mean, sv.z - mean, sv.w - mean);
    if (!cok) d = zero4;
    const float rstd = rsqrtf(wave_sum64((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) / D + 1e-5f);
    if (cok)
      *reinterpret_cast<float4*>(y + row * D + c) =
          make_float4(d.x * rstd * gg.x + bb.x, d.y * rstd * gg.y + bb.y, d.z * rstd * gg.z + bb.z, d.w * rstd * gg.w + bb.w);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

static bool ln_vec_ok(int D, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  static const bool on = [] { const char* v = getenv("RD_LN_VEC"); return !(v && atoi(v) == 0); }();
  const uintptr_t m = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                      reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f);
  return on && (D % 4) == 0 && D <= 256 && (m & 15) == 0;
}

int launch_add_ln_fwd(const float* x, const float* r, const float* g, const float* bta, float* s_out, float* y,
                      float* stats, int M, int D, float p_drop, uint64_t seed, uint32_t site, hipStream_t st) {
  if (ln_vec_ok(D, x, r, g, bta, s_out, y)) {
    hipLaunchKernelGGL(k_add_ln_fwd_v, dim3(cdiv(M, 4 * LNV_RPW)), dim3(256), 0, st, x, r, g, bta, s_out, y, stats, M, D,
                       p_drop, seed, site, seed_cell());
    return check_launch("k_add_ln_fwd_v");
  }
  const int nv = cdiv(D, 64);
  const int nb = cdiv(M, 4 * LNF_RPW);
#define RD_LNF(NV) hipLaunchKernelGGL(k_add_ln_fwd_r<NV>, dim3(nb), dim3(256), 0, st, x, r, g, bta, s_out, y, stats, M, D, \
                                      p_drop, seed, site, seed_cell())
  if (nv == 1) RD_LNF(1); else if (nv == 2) RD_LNF(2); else if (nv == 3) RD_LNF(3); else if (nv == 4) RD_LNF(4);
  else hipLaunchKernelGGL(k_add_ln_fwd, dim3(cdiv(M, 4)), dim3(256), 0, st, x, r, g, bta, s_out, y, stats, M, D, p_drop,
                          seed, site, seed_cell());
#undef RD_LNF
  return check_launch("k_add_ln_fwd");
}

// backward: ds = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));  dr = ds o dropout mask;
// per-block partial sums of dgamma = sum dy*xhat and dbeta = sum dy (64 rows per block).
constexpr int LN_RPB = 16;    // rows per block (4 per wavefront): ~1000 blocks at 15k tokens keep every CU busy
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ s,
                                                const float* __restrict__ stats, const float* __restrict__ g,
                                                float* __restrict__ ds_out, float* __restrict__ dr_out,
                                                float* __restrict__ part, int M, int D, float p_drop,
                                                uint64_t seed, uint32_t site, const uint64_t* cell) {
  seed = eff_seed(seed, cell);
  extern __shared__ float red[];             // [4][2*D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  for (int c = threadIdx.x; c < 8 * D; c += 256) red[c] = 0.f;
  __syncthreads();
  float* myg = red + wave * 2 * D;
  for (int i = 0; i < LN_RPB / 4; ++i) {
    const long row = (long)blockIdx.x * LN_RPB + wave * (LN_RPB / 4) + i;
    if (row >= M) break;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* dyr = dy + row * D; const float* sr = s + row * D;
    float c1 = 0.f, c2 = 0.f;
    for (int c = lane; c < D; c += 64) {
      const float xh = (sr[c] - mean) * rstd, dg = dyr[c] * g[c];
      c1 += dg; c2 += dg * xh;
    }
    c1 = wave_sum64(c1) / D; c2 = wave_sum64(c2) / D;
    for (int c = lane; c < D; c += 64) {
      const float xh = (sr[c] - mean) * rstd, d = dyr[c];
      const float v = rstd * (d * g[c] - c1 - xh * c2);
      ds_out[row * D + c] = v;
      float dv = v;
      if (p_drop > 0.f) dv *= dropout_scale(seed, site, (uint64_t)row * D + c, p_drop, inv_keep);
      dr_out[row * D + c] = dv;
      myg[c] += d * xh;          // each lane owns its columns: no race within the wave
      myg[D + c] += d;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    part[(long)blockIdx.x * 2 * D + c] = (red[c] + red[2 * D + c]) + (red[4 * D + c] + red[6 * D + c]);
}

// register-accumulating form for D <= 64*NV: each lane keeps its columns' dgamma/dbeta partials in
// VGPRs across the wave's rows; the LDS is touched once per block for the 4-wave combine.
template <int NV>
__global__ __launch_bounds__(256) void k_ln_bwd_r(const float* __restrict__ dy, const float* __restrict__ s,
                                                  const float* __restrict__ stats, const float* __restrict__ g,
                                                  float* __restrict__ ds_out, float* __restrict__ dr_out,
                                                  float* __restrict__ part, int M, int D, float p_drop,
                                                  uint64_t seed, uint32_t site, const uint64_t* cell) {
  RD_TOUCH_CODE_X(RD_TL_LN_BWD_R, blockIdx.x, 512);
  seed = eff_seed(seed, cell);
  extern __shared__ float red[];             // [4][2*D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  float gg[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    gg[i] = c < D ? g[c] : 0.f; ag[i] = 0.f; ab[i] = 0.f;
  }
  // every load of the wave's rows is issued before the first use (one memory round trip per wave)
  constexpr int RPW = LN_RPB / 4;
  const long rbase = (long)blockIdx.x * LN_RPB + wave * RPW;
  float sraw[RPW][NV], dvr[RPW][NV], mean_r[RPW], rstd_r[RPW];
#pragma unroll
  for (int it = 0; it < RPW; ++it) {
    const long row = rbase + it;
    const bool rok = row < M;
    mean_r[it] = rok ? stats[2 * row] : 0.f; rstd_r[it] = rok ? stats[2 * row + 1] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool ok = rok && c < D;
      sraw[it][i] = ok ? s[row * D + c] : 0.f;
      dvr[it][i] = ok ? dy[row * D + c] : 0.f;
    }
  }
#pragma unroll
  for (int it = 0; it < RPW; ++it) {
    const long row = rbase + it;
    if (row >= M) break;
    const float mean = mean_r[it], rstd = rstd_r[it];
    float xh[NV], dv[NV];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < D;
      xh[i] = ok ? (sraw[it][i] - mean) * rstd : 0.f;
      dv[i] = dvr[it][i];
      const float dg = dv[i] * gg[i];
      c1 += dg; c2 += dg * xh[i];
    }
    c1 = wave_sum64(c1) / D; c2 = wave_sum64(c2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      if (c < D) {
        const float v = rstd * (dv[i] * gg[i] - c1 - xh[i] * c2);
        ds_out[row * D + c] = v;
        float dvv = v;
        if (p_drop > 0.f) dvv *= dropout_scale(seed, site, (uint64_t)row * D + c, p_drop, inv_keep);
        dr_out[row * D + c] = dvv;
        ag[i] += dv[i] * xh[i];
        ab[i] += dv[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D) { red[wave * 2 * D + c] = ag[i]; red[wave * 2 * D + D + c] = ab[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    part[(long)blockIdx.x * 2 * D + c] = (red[c] + red[2 * D + c]) + (red[4 * D + c] + red[6 * D + c]);
}

// float4 form of the backward (see k_add_ln_fwd_v): one Philox evaluation per lane and row, 16-byte accesses
__global__ __launch_bounds__(256) void k_ln_bwd_v(const float* __restrict__ dy, const float* __restrict__ s,
                                                  const float* __restrict__ stats, const float* __restrict__ g,
                                                  float* __restrict__ ds_out, float* __restrict__ dr_out,
                                                  float* __restrict__ part, int M, int D, float p_drop,
                                                  uint64_t seed, uint32_t site, const uint64_t* cell) {
  RD_TOUCH_CODE_X(RD_TL_LN_BWD_V, blockIdx.x, 512);
  seed = eff_seed(seed, cell);
  extern __shared__ float red[];             // [4][2*D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_keep = 1.0f / (1.0f - p_drop);
  constexpr int RPW = LN_RPB / 4;
  const int c = 4 * lane;
  const bool cok = c < D;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gg = zero4;
  if (cok) gg = *reinterpret_cast<const float4*>(g + c);
  float4 ag = zero4, ab = zero4;
  const long rbase = (long)blockIdx.x * LN_RPB + wave * RPW;
  float4 sraw[RPW], dvr[RPW]; float mean_r[RPW], rstd_r[RPW];
#pragma unroll
  for (int it = 0; it < RPW; ++it) {
    const long row = rbase + it;
    const bool rok = row < M;
    mean_r[it] = rok ? stats[2 * row] : 0.f; rstd_r[it] = rok ? stats[2 * row + 1] : 0.f;
    sraw[it] = zero4; dvr[it] = zero4;
    if (rok && cok) {
      sraw[it] = *reinterpret_cast<const float4*>(s + row * D + c);
      dvr[it] = *reinterpret_cast<const float4*>(dy + row * D + c);
    }
  }
#pragma unroll
  for (int it = 0; it < RPW; ++it) {
    const long row = rbase + it;
    if (row >= M) break;
    const float mean = mean_r[it], rstd = rstd_r[it];
    const float4 dv = dvr[it];
    float4 xh = make_float4((sraw[it].x - mean) * rstd, (sraw[it].y - mean) * rstd, (sraw[it].z - mean) * rstd,
                            (sraw[it].w - mean) * rstd);
    if (!cok) xh = zero4;
    const float4 dg = make_float4(dv.x * gg.x, dv.y * gg.y, dv.z * gg.z, dv.w * gg.w);
    const float c1 = wave_sum64((dg.x + dg.y) + (dg.z + dg.w)) / D;
    const float c2 = wave_sum64((dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w)) / D;
    if (cok) {
      const float4 v = make_float4(rstd * (dg.x - c1 - xh.x * c2), rstd * (dg.y - c1 - xh.y * c2),
                                   rstd * (dg.z - c1 - xh.z * c2), rstd * (dg.w - c1 - xh.w * c2));
      *reinterpret_cast<float4*>(ds_out + row * D + c) = v;
      float4 dr = v;
      if (p_drop > 0.f) {
        const float4 u = uniform4(seed, site, ((uint64_t)row * D + c) >> 2);
        dr.x *= u.x >= p_drop ? inv_keep : 0.f; dr.y *= u.y >= p_drop ? inv_keep : 0.f;
        dr.z *= u.z >= p_drop ? inv_keep : 0.f; dr.w *= u.w >= p_drop ? inv_keep : 0.f;
      }
      *reinterpret_cast<float4*>(dr_out + row * D + c) = dr;
      ag.x += dv.x * xh.x; ag.y += dv.y * xh.y; ag.z += dv.z * xh.z; ag.w += dv.w * xh.w;
      ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
    }
  }
  if (cok) {
    *reinterpret_cast<float4*>(red + wave * 2 * D + c) = ag;
    *reinterpret_cast<float4*>(red + wave * 2 * D + D + c) = ab;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256)
    part[(long)blockIdx.x * 2 * D + i] = (red[i] + red[2 * D + i]) + (red[4 * D + i] + red[6 * D + i]);
}

int launch_ln_bwd(const float* dy, const float* s, const float* stats, const float* g, float* ds_out, float* dr_out,
                  float* part, int M, int D, float p_drop, uint64_t seed, uint32_t site, hipStream_t st) {
  const int lnb = cdiv(M, LN_RPB);
  const size_t lds = sizeof(float) * 8 * D;
  if (ln_vec_ok(D, dy, s, g, ds_out, dr_out, nullptr)) {
    hipLaunchKernelGGL(k_ln_bwd_v, dim3(lnb), dim3(256), lds, st, dy, s, stats, g, ds_out, dr_out, part, M, D, p_drop, seed,
                       site, seed_cell());
    return check_launch("k_ln_bwd_v");
  }
  const int nv = cdiv(D, 64);
#define RD_LN(NV) hipLaunchKernelGGL(k_ln_bwd_r<NV>, dim3(lnb), dim3(256), lds, st, dy, s, stats, g, ds_out, dr_out, \
                                     part, M, D, p_drop, seed, site, seed_cell())
  if (nv == 1) RD_LN(1); else if (nv == 2) RD_LN(2); else if (nv == 3) RD_LN(3); else if (nv == 4) RD_LN(4);
  else hipLaunchKernelGGL(k_ln_bwd, dim3(lnb), dim3(256), lds, st, dy, s, stats, g, ds_out, dr_out, part, M, D,
                          p_drop, seed, site, seed_cell());
#undef RD_LN
  return check_launch("k_ln_bwd");
}

// code/models_rd.py:366-367,379: agg[b,c] = sum_t r[t,b,c] * (1 - mask[b,t]) / (lengths[b] + 1)
__global__ __launch_bounds__(1024) void k_masked_mean_fwd(const float* __restrict__ r, const uint8_t* __restrict__ mask,
                                                          const int64_t* __restrict__ lengths, float* __restrict__ out,
                                                          int T, int B, int D, int ldo) {
  // block = (sample, 64-column chunk); 16 time groups x 64 columns, fixed-order LDS combine
  __shared__ float red[16][64];
  const int b = blockIdx.x, cl = threadIdx.x & 63, tg = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  float s = 0.f;
  if (c < D)
    for (int t = tg; t < T; t += 16)
      if (!mask[(long)b * T + t]) s += r[((long)t * B + b) * D + c];
  red[tg][cl] = s;
  __syncthreads();
  if (tg == 0 && c < D) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][cl];
    out[(long)b * ldo + c] = v / (float)(lengths[b] + 1);
  }
}

__global__ __launch_bounds__(256) void k_masked_mean_bwd(const float* __restrict__ dagg, const uint8_t* __restrict__ mask,
                                                         const int64_t* __restrict__ lengths, float* __restrict__ dr,
                                                         int T, int B, int D, int ldo) {
  const long n = (long)T * B * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    const long tb = i / D;
    const int b = (int)(tb % B), t = (int)(tb / B);
    dr[i] = mask[(long)b * T + t] ? 0.f : dagg[(long)b * ldo + c] / (float)(lengths[b] + 1);
  }
}

// ---- encoder layer bookkeeping -----------------------------------------------------------------
struct EncDims { long M; int D, Hd, nhid, H, T, B; };

EncDims enc_dims(const rd_shape* s) {
  EncDims e;
  e.T = s->T; e.B = s->B; e.D = s->F * s->d_ob + s->d_pe; e.H = s->nhead; e.Hd = e.D / e.H;
  e.nhid = s->nhid; e.M = (long)s->T * s->B;
  return e;
}

// wide heads take the materialised-score attention; RD_ATTN_BIG=1 forces it for every shape (tests: it must agree with the
// tiled kernels, masks included).  Read per call: saved / workspace sizes depend on it.
static bool attn_big(const EncDims& e) {
  const char* v = getenv("RD_ATTN_BIG");
  return e.Hd > 96 || (v && atoi(v) != 0);
}

// chunks of the per-sample group space (rd_plan.h: coff): every sample up to ceil(T / 16) groups of 16 rows, two groups per chunk
static size_t attn_chunks(const EncDims& e) { return ((size_t)e.B * cdiv(e.T, 16) + 1) / 2; }

struct EncSaved { float *qkv, *attn, *lse, *s1, *st1, *x1, *h, *s2, *st2; __bf16* pl[8][2];
                  __bf16* xt[4]; __bf16* ones;        // row tiles of x, attn, x1, h (operands of the weight-gradient stream)
                  __bf16 *afw, *abw;                  // fused attention (rd_attnfuse.hip): per-head in_proj tiles, forward / input-gradient form
                  float *pbig, *pdbig;                // wide heads only: P and dropout(P), [B*H][T][T] each
                  size_t bytes; };
// weight tiles kept from forward to backward: 0 in_proj, 1 out_proj, 2 lin1, 3 lin2, 4 out_proj^T, 5 lin2^T, 6 lin1^T, 7 in_proj^T
EncSaved carve_saved(const EncDims& e, void* base) {
  EncSaved v; size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr;
                              off += align_up(n * sizeof(float), 256); return p; };
  v.qkv = take(e.M * 3 * e.D); v.attn = take(e.M * e.D); v.lse = take((size_t)e.B * e.H * e.T);
  v.s1 = take(e.M * e.D); v.st1 = take(e.M * 2); v.x1 = take(e.M * e.D);
  v.h = take(e.M * e.nhid); v.s2 = take(e.M * e.D); v.st2 = take(e.M * 2);
  const int prow[8] = {3 * e.D, e.D, e.nhid, e.D, e.D, e.nhid, e.D, e.D};      // plane rows = output columns
  const int pcol[8] = {e.D, e.D, e.D, e.nhid, e.D, e.D, e.nhid, 3 * e.D};      // plane cols = reduction length
  for (int i = 0; i < 8; ++i)
    for (int h = 0; h < 2; ++h) v.pl[i][h] = (__bf16*)take((rowgemm_plane_elems(prow[i], pcol[i]) + 1) / 2);
  const int xcols[4] = {e.D, e.D, e.D, e.nhid};
  for (int i = 0; i < 4; ++i) {
    size_t n = tile_elems(e.M, xcols[i]);
    if (i == 0) n = std::max(n, attn_chunks(e) * (size_t)cdiv(e.D, 16) * 1024);     // x tiles in the per-sample chunk space (fused attention)
    v.xt[i] = (__bf16*)take((n + 1) / 2);
  }
  v.ones = (__bf16*)take((tile_wgrad_ones_elems() + 1) / 2);
  v.afw = (__bf16*)take((attnfuse_wf_elems(e.H) + 1) / 2);
  v.abw = (__bf16*)take((attnfuse_wb_elems(e.H) + 1) / 2);
  const size_t big = attn_big(e) ? (size_t)e.B * e.H * e.T * e.T : 0;
  v.pbig = take(big); v.pdbig = take(big);
  v.bytes = off;
  return v;
}

struct EncWs { float *o, *f, *ds2, *df, *du, *dx1, *ds1, *dout, *da, *dqkv, *delta, *lnpart, *lnpart1, *lnred, *splitk, *colsum;
               float* dsbig;                      // wide heads only: dS [B*H][T][T]
               __bf16* dt[4]; float* twpart[4];   // row tiles of df, du, dout, dqkv; slice partials of lin2, lin1, out_proj, in_proj
               size_t bytes; int ns_max; };
EncWs carve_ws(const EncDims& e, void* base) {
  EncWs w; size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)((char*)base + off) : nullptr;
                              off += align_up(n * sizeof(float), 256); return p; };
  w.o = take(e.M * e.D); w.f = take(e.M * e.D);
  w.ds2 = take(e.M * e.D); w.df = take(e.M * e.D); w.du = take(e.M * e.nhid); w.dx1 = take(e.M * e.D);
  w.ds1 = take(e.M * e.D); w.dout = take(e.M * e.D); w.da = take(e.M * e.D); w.dqkv = take(e.M * 3 * e.D);
  w.delta = take((size_t)e.B * e.H * e.T);
  w.lnpart = take((size_t)cdiv((int)e.M, LN_RPB) * 2 * e.D);
  w.lnpart1 = take((size_t)cdiv((int)e.M, LN_RPB) * 2 * e.D);
  w.lnred = take((size_t)2 * e.D + colsum_ws_floats(cdiv((int)e.M, LN_RPB), 2 * e.D));
  size_t sk = 0; int kps;
  const int shapes[4][2] = {{3 * e.D, e.D}, {e.D, e.D}, {e.nhid, e.D}, {e.D, e.nhid}};
  (void)kps;
  for (auto& sh : shapes) { size_t v = (size_t)wgrad_ws_floats(e.M, sh[0], sh[1]); if (v > sk) sk = v; }
  w.splitk = take(sk);
  w.colsum = take(colsum_ws_floats((int)e.M, 3 * e.D > e.nhid ? 3 * e.D : e.nhid));
  const int dcols[4] = {e.D, e.nhid, e.D, 3 * e.D};
  const int padc = attnfuse_padded_cols(e.H);                       // head-padded dqkv columns of the fused attention's export
  for (int i = 0; i < 4; ++i) {
    size_t n = tile_elems(e.M, dcols[i]);
    if (i == 3) n = std::max(n, attn_chunks(e) * (size_t)(padc / 16) * 1024);
    w.dt[i] = (__bf16*)take((n + 1) / 2);
  }
  const int pn[4] = {e.D, e.nhid, e.D, std::max(3 * e.D, padc)}, pk[4] = {e.nhid, e.D, e.D, e.D};
  for (int i = 0; i < 4; ++i) w.twpart[i] = take(tile_wgrad_part_floats(pn[i], pk[i]));
  w.dsbig = take(attn_big(e) ? (size_t)e.B * e.H * e.T * e.T : 0);
  w.bytes = off;
  return w;
}

// LEAN chains (rd_encfuse.hip): on the token plan with the tile-stream weight gradients -- the training step's configuration,
// where forward and backward of a layer are known to take the fused chains -- the forward does not write the fp32 FFN hidden and
// x1: the gate bytes live at the start of the (then unused) h buffer.  RD_ENC_LEAN=0: the round-3 contract (A/B).
static bool enc_lean(const int32_t* tp, bool tw) {
  const char* e = getenv("RD_ENC_LEAN");               // read per call
  return tp && tw && !(e && atoi(e) == 0);
}

// the whole layer runs on row-block products and all four weight gradients can take the tile stream
static bool tile_path(const EncDims& e) {
  return rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) && rowgemm_ok(e.nhid, e.D, e.D, e.nhid) &&
         rowgemm_ok(e.D, e.nhid, e.nhid, e.D) && rowgemm_ok(e.D, 3 * e.D, 3 * e.D, e.D) && tile_wgrad_ok(3 * e.D, e.D) &&
         tile_wgrad_ok(e.D, e.D) && tile_wgrad_ok(e.nhid, e.D) && tile_wgrad_ok(e.D, e.nhid);
}

int linear_fwd(long M, int N, int K, const float* x, const float* W, const float* b, float* y, int relu,
               float p, uint64_t seed, uint32_t site, hipStream_t st) {
  GemmArgs g{};
  g.M = (int)M; g.N = N; g.K = K; g.nsplit = 1;
  g.A = x; g.sa_m = K; g.sa_k = 1; g.B = W; g.sb_n = K; g.sb_k = 1; g.C = y; g.sc_m = N;
  g.bias = b; g.relu = relu; g.drop_p = p; g.drop_seed = seed; g.drop_site = site;
  return launch_gemm(g, st);
}
// dx = dy W (+ residual), optionally gated by posmask > 0 and scaled
// the same products with the weight also given as native operand tiles (k_wsplit): launch_gemm takes the panel form when it applies
static int linear_fwd_t(long M, int N, int K, const float* x, const float* W, const void* tiles, const float* b, float* y, int relu,
                        float p, uint64_t seed, uint32_t site, hipStream_t st) {
  GemmArgs g{};
  g.M = (int)M; g.N = N; g.K = K; g.nsplit = 1;
  g.A = x; g.sa_m = K; g.sa_k = 1; g.B = W; g.sb_n = K; g.sb_k = 1; g.C = y; g.sc_m = N;
  g.Btiles = tiles; g.bt_ntile = cdiv(N, 16); g.bt_nkc = cdiv(K, 32);
  g.bias = b; g.relu = relu; g.drop_p = p; g.drop_seed = seed; g.drop_site = site;
  return launch_gemm(g, st);
}
static int linear_bwd_x_t(long M, int N, int K, const float* dy, const float* W, const void* tiles_t, float* dx, const float* posmask,
                          float cscale, const float* residual, hipStream_t st) {
  GemmArgs g{};
  g.M = (int)M; g.N = K; g.K = N; g.nsplit = 1;
  g.A = dy; g.sa_m = N; g.sa_k = 1; g.B = W; g.sb_n = 1; g.sb_k = K; g.C = dx; g.sc_m = K;
  g.Btiles = tiles_t; g.bt_ntile = cdiv(K, 16); g.bt_nkc = cdiv(N, 32);     // tiles of W^T: rows = K, reduction = N
  g.posmask = posmask; g.pm_m = K; g.cscale = cscale; g.residual = residual; g.res_m = K;
  return launch_gemm(g, st);
}
int linear_bwd_x(long M, int N, int K, const float* dy, const float* W, float* dx, const float* posmask,
                 float cscale, const float* residual, hipStream_t st) {
  GemmArgs g{};
  g.M = (int)M; g.N = K; g.K = N; g.nsplit = 1;
  g.A = dy; g.sa_m = N; g.sa_k = 1; g.B = W; g.sb_n = 1; g.sb_k = K; g.C = dx; g.sc_m = K;
  g.posmask = posmask; g.pm_m = K; g.cscale = cscale; g.residual = residual; g.res_m = K;
  return launch_gemm(g, st);
}
int linear_bwd_w(long M, int N, int K, const float* dy, const float* x, float* dW, float* db, float* splitk,
                 float* colsum, hipStream_t st) {
  (void)colsum;
  return launch_wgrad(M, N, K, dy, N, x, K, dW, db, splitk, st);
}

int check_enc(const rd_shape* s) {
  RD_REQUIRE(s != nullptr, "rd_shape is NULL");
  RD_REQUIRE(s->B >= 0 && s->T > 0 && s->F > 0 && s->d_ob > 0 && s->d_pe >= 0 && s->nhead > 0 && s->nhid > 0,
             "bad rd_shape");
  const int D = s->F * s->d_ob + s->d_pe;
  RD_REQUIRE(D % s->nhead == 0, "D=%d not divisible by nhead=%d", D, s->nhead);
  RD_REQUIRE((long)s->B * s->nhead * s->T * s->T < (1L << 31) || D / s->nhead <= 96, "score tensor exceeds 2^31 elements");
  RD_REQUIRE((long)s->T * s->B * (3L * D > s->nhid ? 3L * D : s->nhid) < (1L << 31), "tensor exceeds 2^31 elements");
  return RD_OK;
}

}  // namespace
}  // namespace rd

using namespace rd;

extern "C" size_t rd_encoder_layer_saved_bytes(const rd_shape* s) {
  if (check_enc(s)) return 0;
  return carve_saved(enc_dims(s), nullptr).bytes;
}
extern "C" size_t rd_encoder_layer_workspace_bytes(const rd_shape* s) {
  if (check_enc(s)) return 0;
  return carve_ws(enc_dims(s), nullptr).bytes;
}

extern "C" void rd_debug_set_attn_stamps(void* p) { g_attn_stamps = (unsigned long long*)p; }   // not part of the ABI

// weights of a layer -> native bf16 hi/lo operand tiles (both orientations) + the constant tiles of the weight-gradient stream
static int enc_split_specs(const EncDims& e, const rd_encoder_weights* w, const EncSaved& v, WsplitSpec* out, bool all8 = false) {
  const float* Ws[8] = {w->in_proj_w, w->out_proj_w, w->lin1_w, w->lin2_w, w->out_proj_w, w->lin2_w, w->lin1_w, w->in_proj_w};
  const int Ns[8] = {3 * e.D, e.D, e.nhid, e.D, e.D, e.D, e.nhid, 3 * e.D};
  const int Ks[8] = {e.D, e.D, e.D, e.nhid, e.D, e.nhid, e.D, e.D};
  const int Tr[8] = {0, 0, 0, 0, 1, 1, 1, 1};
  int njobs = (all8 || rowgemm_ok(e.D, 3 * e.D, 3 * e.D, e.D)) ? 8 : 7;
  for (int i = 0; i < njobs; ++i) out[i] = WsplitSpec{Ws[i], Ns[i], Ks[i], Tr[i], v.pl[i][0]};
  // the fused attention's per-head in_proj tiles (6 H more jobs of a few tiles each; whether a call takes that path depends on the
  // token plan, which a prepare call does not see: split whenever the shape is in its envelope)
  if (!all8 && attnfuse_ok(e.T, e.D, e.H, e.Hd)) njobs += attnfuse_split_specs(w->in_proj_w, e.D, e.H, e.Hd, v.afw, v.abw, out + njobs);
  return njobs;
}
constexpr int ENC_MAX_SPECS = 8 + 12;                  // per layer: 8 orientations + (fused attention, H = 2) 6 H
static int enc_prepare(const EncDims& e, const rd_encoder_weights* w, const EncSaved& v, bool tw, hipStream_t st) {
  WsplitSpec specs[ENC_MAX_SPECS];
  const int n = enc_split_specs(e, w, v, specs);
  void* on[1] = {v.ones};
  return launch_wsplit_specs(n, specs, tw ? 1 : 0, on, st);
}

extern "C" int rd_encoder_layer_prepare(const rd_shape* s, const rd_encoder_weights* w, void* saved, size_t saved_bytes,
                                        void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(w && saved, "NULL tensor");
  const EncDims e = enc_dims(s);
  EncSaved v = carve_saved(e, saved);
  RD_REQUIRE(saved_bytes >= v.bytes, "saved buffer too small");
  const bool rg = rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) &&
                  rowgemm_ok(e.nhid, e.D, e.D, e.nhid) && rowgemm_ok(e.D, e.nhid, e.nhid, e.D);
  if (!rg) return RD_OK;                             // the tiled path reads the fp32 weights directly
  return enc_prepare(e, w, v, tile_path(e), (hipStream_t)stream);
}

namespace rd {
int k1_weight_split_specs(const rd_shape* s, const float* W1, const float* W2, void* saved, size_t saved_bytes, WsplitSpec* out);
bool fused_msgpass_ok(const rd_shape* s);
}

// Every weight matrix of a training step -> matrix-core operand tiles in ONE launch: the encoder layers' (what
// rd_encoder_layer_prepare does per layer) and the message-passing stage's (what rd_sensor_stage_fwd does first).
static int step_prepare_impl(const rd_shape* s, int32_t nlayers, const rd_encoder_weights* const* w, void* const* enc_saved,
                             const size_t* enc_saved_bytes, const float* W1, const float* W2, void* k1_saved, size_t k1_saved_bytes,
                             const int64_t* lengths, int32_t* plan_out, uint64_t* seed_cell_dev, uint64_t delta, void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(nlayers >= 0 && nlayers <= 2 && (nlayers == 0 || (w && enc_saved && enc_saved_bytes)), "rd_step_prepare: 0..2 encoder layers");
  const EncDims e = enc_dims(s);
  const bool rg = rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) &&
                  rowgemm_ok(e.nhid, e.D, e.D, e.nhid) && rowgemm_ok(e.D, e.nhid, e.nhid, e.D);
  const bool tw = rg && tile_path(e);
  WsplitSpec specs[2 * ENC_MAX_SPECS + 8]; void* ones[4]; int n = 0, no = 0;
  if (rg)
    for (int l = 0; l < nlayers; ++l) {
      RD_REQUIRE(w[l] && enc_saved[l], "NULL tensor");
      EncSaved v = carve_saved(e, enc_saved[l]);
      RD_REQUIRE(enc_saved_bytes[l] >= v.bytes, "saved buffer too small");
      n += enc_split_specs(e, w[l], v, specs + n);
      if (tw) ones[no++] = v.ones;
    }
  if (W1 && W2 && k1_saved) n += k1_weight_split_specs(s, W1, W2, k1_saved, k1_saved_bytes, specs + n);
  if (n == 0) return plan_out ? rd_token_plan(s, lengths, plan_out, seed_cell_dev, delta, stream) : RD_OK;
  return launch_wsplit_plan(n, specs, no, ones, lengths, plan_out, s->B, s->T, seed_cell_dev, delta, (hipStream_t)stream);
}
extern "C" int rd_step_prepare(const rd_shape* s, int32_t nlayers, const rd_encoder_weights* const* w, void* const* enc_saved,
                               const size_t* enc_saved_bytes, const float* W1, const float* W2, void* k1_saved, size_t k1_saved_bytes,
                               void* stream) {
  return step_prepare_impl(s, nlayers, w, enc_saved, enc_saved_bytes, W1, W2, k1_saved, k1_saved_bytes, nullptr, nullptr, nullptr, 0, stream);
}
// rd_token_plan + rd_step_prepare as ONE launch: everything a training step needs before its first compute kernel (the plan is one
// workgroup of integer work, the splits ~1500 small tiles: two launches of 6-8 us each were mostly launch latency)
extern "C" int rd_step_begin(const rd_shape* s, const int64_t* lengths, int32_t* plan_out, uint64_t* seed_cell_dev, uint64_t delta,
                             int32_t nlayers, const rd_encoder_weights* const* w, void* const* enc_saved, const size_t* enc_saved_bytes,
                             const float* W1, const float* W2, void* k1_saved, size_t k1_saved_bytes, void* stream) {
  RD_REQUIRE(lengths && plan_out, "NULL tensor");
  return step_prepare_impl(s, nlayers, w, enc_saved, enc_saved_bytes, W1, W2, k1_saved, k1_saved_bytes, lengths, plan_out, seed_cell_dev,
                           delta, stream);
}
// 1 if rd_step_prepare covers the encoder layers / the message-passing stage of this shape in the current arithmetic mode (then
// pass RD_LAYER_WEIGHTS_PREPARED / call rd_sensor_stage_fwd_prepared), else the per-call splits must run
extern "C" int rd_step_prepare_covers(const rd_shape* s, int32_t* encoder, int32_t* sensor_stage) {
  if (check_enc(s)) return RD_EINVAL;
  const EncDims e = enc_dims(s);
  const bool rg = rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) &&
                  rowgemm_ok(e.nhid, e.D, e.D, e.nhid) && rowgemm_ok(e.D, e.nhid, e.nhid, e.D);
  if (encoder) *encoder = rg ? 1 : 0;
  if (sensor_stage) *sensor_stage = fused_msgpass_ok(s) ? 1 : 0;
  return RD_OK;
}

extern "C" int rd_encoder_layer_fwd(const rd_shape* s, int32_t layer, const float* x, const uint8_t* mask,
                                    const rd_encoder_weights* w, float p_drop, uint64_t seed, float* y,
                                    void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;                       // empty batch
  RD_REQUIRE(x && mask && w && y && saved && workspace, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  const EncDims e = enc_dims(s);
  EncSaved v = carve_saved(e, saved);
  EncWs ws = carve_ws(e, workspace);
  RD_REQUIRE(saved_bytes >= v.bytes && workspace_bytes >= ws.bytes, "saved/workspace buffer too small");
  if (e.B == 0) return RD_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool prepared = (layer & RD_LAYER_WEIGHTS_PREPARED) != 0;      // rd_encoder_layer_prepare already ran for these weights
  layer &= 0xffff;
  const uint32_t L = (uint32_t)layer;
  // weights -> bf16 hi/lo planes (both orientations needed by this layer's forward and backward), one launch
  const bool rg = rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) &&
                  rowgemm_ok(e.nhid, e.D, e.D, e.nhid) && rowgemm_ok(e.D, e.nhid, e.nhid, e.D);
  const bool tw = rg && tile_path(e);
  const bool panel = !rg && precision() != RD_PREC_FP32;
  // token plan (rd_plan.h): x, y and every saved / scratch tensor hold the live rows only, in plan order
  const int32_t* tp = token_plan();
  RD_REQUIRE(!tp || (tw && !attn_big(e)), "token plan: this shape / mode does not run on the row-block + tile-stream path");
  struct MliveScope { MliveScope(const int32_t* p) { rowgemm_set_mlive(p); } ~MliveScope() { rowgemm_set_mlive(nullptr); } } mscope(tp);
  // in_proj + attention core as ONE launch (rd_attnfuse.hip): a workgroup owns a sample, qkv never reaches memory
  const bool afuse = tp && tw && encfuse_ok(e.D, e.nhid) && attnfuse_ok(e.T, e.D, e.H, e.Hd);      // (same condition as the backward's)
  if (rg) {
    if (!prepared && (rc = enc_prepare(e, w, v, tw, st))) return rc;
    if (!afuse) {
      if (tw) rowgemm_export_next(v.xt[0]);
      if ((rc = launch_rowgemm(e.M, 3 * e.D, e.D, x, e.D, v.pl[0][0], v.pl[0][1], v.qkv, 3 * e.D, w->in_proj_b, 0, nullptr, 0,
                               0.f, nullptr, 0, 0.f, 0, 0, st))) return rc;
    }
  } else {
    // widths beyond the row-block kernels (SYN256: D = 1040): the tiled family -- in the bf16 modes on its panel form, with the
    // layer's eight weight orientations split here (kept in `saved` for the backward)
    if (panel) {
      WsplitSpec specs[8];
      const int n = enc_split_specs(e, w, v, specs, true);
      void* on[1] = {v.ones};                                // the constant tile of the streamed weight gradients (backward, tw2)
      if ((rc = launch_wsplit_specs(n, specs, 1, on, st))) return rc;
    }
    if ((rc = linear_fwd_t(e.M, 3 * e.D, e.D, x, w->in_proj_w, panel ? v.pl[0][0] : nullptr, w->in_proj_b, v.qkv, 0, 0.f, 0, 0, st))) return rc;
  }
  AttnArgs a{};
  a.qkv = v.qkv; a.mask = mask; a.out = v.attn; a.lse = v.lse;
  a.T = e.T; a.B = e.B; a.D = e.D; a.H = e.H; a.hd = e.Hd;
  a.scale = 1.0f / sqrtf((float)e.Hd); a.p_drop = p_drop; a.seed = seed; a.site = SITE_ATTN_PROB + L; a.seed_cell = seed_cell();
  a.plan = tp;
  if (afuse) {
    if ((rc = launch_attn_fused_fwd(x, v.afw, w->in_proj_b, tp, e.T, e.B, e.D, e.H, e.Hd, p_drop, seed, SITE_ATTN_PROB + L, v.attn, v.lse, st)))
      return rc;
  } else if (attn_big(e)) { if ((rc = attn_big_fwd(a, v.pbig, v.pdbig, st))) return rc; }
  else if ((rc = dispatch_attn(a, 0, st))) return rc;
  // out-projection / second FFN layer with the residual add + LayerNorm in their epilogue (a workgroup owns complete rows)
  static const bool ln_fuse_env = [] { const char* e = getenv("RD_LN_FUSE"); return !(e && atoi(e) == 0); }();
  const bool lnf1 = rg && ln_fuse_env && rowgemm_ln_ok(e.D, e.D), lnf2 = rg && ln_fuse_env && rowgemm_ln_ok(e.D, e.nhid);
  RD_REQUIRE(!tp || (lnf1 && lnf2), "token plan: needs the LayerNorm-epilogue path (RD_LN_FUSE)");
  // out_proj + LayerNorm1 + linear1 + linear2 + LayerNorm2 as ONE launch (rd_encfuse.hip): same arithmetic as the three
  // row-block launches below, the rows stay in LDS in between
  if (lnf1 && lnf2 && encfuse_ok(e.D, e.nhid))
    return launch_enc_post_fwd(e.M, e.D, e.nhid, v.attn, x, v.pl[1][0], v.pl[2][0], v.pl[3][0], w->out_proj_b, w->lin1_b, w->lin2_b,
                               w->norm1_w, w->norm1_b, w->norm2_w, w->norm2_b, v.s1, v.x1, v.st1, v.h, v.s2, y, v.st2,
                               tw ? v.xt[1] : nullptr, tw ? v.xt[2] : nullptr, tw ? v.xt[3] : nullptr, p_drop, seed,
                               SITE_ATTN_OUT + L, SITE_FFN_HID + L, SITE_FFN_OUT + L, tp ? tp + plan::I_MLIVE : nullptr,
                               enc_lean(tp, tw) ? (void*)v.h : nullptr, st);
  if (tw) rowgemm_export_next(v.xt[1]);
  if (lnf1) {
    if ((rc = launch_rowgemm_ln(e.M, e.D, e.D, v.attn, v.pl[1][0], w->out_proj_b, x, w->norm1_w, w->norm1_b, v.s1, v.x1, v.st1,
                                p_drop, seed, SITE_ATTN_OUT + L, st))) return rc;
  } else {
    if (rg) {
      if ((rc = launch_rowgemm(e.M, e.D, e.D, v.attn, e.D, v.pl[1][0], v.pl[1][1], ws.o, e.D, w->out_proj_b, 0, nullptr, 0, 0.f,
                               nullptr, 0, 0.f, 0, 0, st))) return rc;
    } else if ((rc = linear_fwd_t(e.M, e.D, e.D, v.attn, w->out_proj_w, panel ? v.pl[1][0] : nullptr, w->out_proj_b, ws.o, 0, 0.f, 0, 0, st))) return rc;
    if ((rc = launch_add_ln_fwd(x, ws.o, w->norm1_w, w->norm1_b, v.s1, v.x1, v.st1, (int)e.M, e.D, p_drop, seed,
                                SITE_ATTN_OUT + L, st))) return rc;
  }
  if (rg) {
    if (tw) rowgemm_export_next(v.xt[2]);
    if ((rc = launch_rowgemm(e.M, e.nhid, e.D, v.x1, e.D, v.pl[2][0], v.pl[2][1], v.h, e.nhid, w->lin1_b, 1, nullptr, 0, 0.f,
                             nullptr, 0, p_drop, seed, SITE_FFN_HID + L, st))) return rc;
    if (tw) rowgemm_export_next(v.xt[3]);
    if (lnf2)
      return launch_rowgemm_ln(e.M, e.D, e.nhid, v.h, v.pl[3][0], w->lin2_b, v.x1, w->norm2_w, w->norm2_b, v.s2, y, v.st2,
                               p_drop, seed, SITE_FFN_OUT + L, st);
    if ((rc = launch_rowgemm(e.M, e.D, e.nhid, v.h, e.nhid, v.pl[3][0], v.pl[3][1], ws.f, e.D, w->lin2_b, 0, nullptr, 0, 0.f,
                             nullptr, 0, 0.f, 0, 0, st))) return rc;
  } else {
    if ((rc = linear_fwd_t(e.M, e.nhid, e.D, v.x1, w->lin1_w, panel ? v.pl[2][0] : nullptr, w->lin1_b, v.h, 1, p_drop, seed, SITE_FFN_HID + L, st)))
      return rc;
    if ((rc = linear_fwd_t(e.M, e.D, e.nhid, v.h, w->lin2_w, panel ? v.pl[3][0] : nullptr, w->lin2_b, ws.f, 0, 0.f, 0, 0, st))) return rc;
  }
  return launch_add_ln_fwd(v.x1, ws.f, w->norm2_w, w->norm2_b, v.s2, y, v.st2, (int)e.M, e.D, p_drop, seed,
                           SITE_FFN_OUT + L, st);
}

extern "C" int rd_encoder_layer_bwd(const rd_shape* s, int32_t layer, const float* x, const uint8_t* mask,
                                    const rd_encoder_weights* w, float p_drop, uint64_t seed, const void* saved,
                                    size_t saved_bytes, const float* dy, float* dx, const rd_encoder_grads* g,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  RD_REQUIRE(x && mask && w && saved && dy && dx && g && workspace, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  const EncDims e = enc_dims(s);
  EncSaved v = carve_saved(e, const_cast<void*>(saved));
  EncWs ws = carve_ws(e, workspace);
  RD_REQUIRE(saved_bytes >= v.bytes && workspace_bytes >= ws.bytes, "saved/workspace buffer too small");
  hipStream_t st = (hipStream_t)stream;
  if (e.B == 0) return RD_OK;
  const uint32_t L = (uint32_t)layer;
  const float keep = 1.0f / (1.0f - p_drop);
  const int lnb = cdiv((int)e.M, LN_RPB);
  hipStream_t sw = st;
  const bool rg = rowgemm_ok(3 * e.D, e.D, e.D, 3 * e.D) && rowgemm_ok(e.D, e.D, e.D, e.D) &&
                  rowgemm_ok(e.nhid, e.D, e.D, e.nhid) && rowgemm_ok(e.D, e.nhid, e.nhid, e.D);
  // tw: the input-gradient products below export their A operand (df, du, dout, dqkv) as row tiles and the four weight
  // gradients run as one streaming launch at the end of the layer (rd_tile_wgrad.hip), instead of four split-K GEMMs;
  // its reduce launch also column-sums the two LayerNorm partial matrices
  const bool tw = rg && tile_path(e);
  const bool panel = !rg && precision() != RD_PREC_FP32;      // the forward split the eight weight orientations into `saved`
  // tw2 (round 4): widths beyond the row-block kernels (SYN256: D = 1040).  The four weight gradients as the same tile stream,
  // fed by ONE stand-alone conversion pass over the eight operands (rd_tiles_export.hip) instead of four split-K GEMMs that
  // convert both fp32 operands once per 64 x 64 output tile (52 % of SYN256's step: 11.2 -> 7.65 ms with this).  Narrow layers keep the
  // GEMMs (PAM, D = 84: 2.96 vs 3.04 ms -- the conversion pass costs more than it saves).  RD_TILE_WGRAD_GENERIC=0: the GEMMs (A/B).
  const char* tw2_v = getenv("RD_TILE_WGRAD_GENERIC");        // read per call (tests compare both paths in one process)
  const bool tw2_env = !(tw2_v && atoi(tw2_v) == 0);
  const bool tw2 = panel && tw2_env && e.M >= 1024 && e.D >= 256 && tile_wgrad_ok(3 * e.D, e.D) && tile_wgrad_ok(e.D, e.D) && tile_wgrad_ok(e.nhid, e.D) &&
                   tile_wgrad_ok(e.D, e.nhid);
  const int32_t* tp = token_plan();
  RD_REQUIRE(!tp || (tw && !attn_big(e)), "token plan: this shape / mode does not run on the row-block + tile-stream path");
  struct MliveScope { MliveScope(const int32_t* p) { rowgemm_set_mlive(p); } ~MliveScope() { rowgemm_set_mlive(nullptr); } } mscope(tp);
  // lnf: the LayerNorm backward runs as the PROLOGUE of the input-gradient product that consumes its output (its second
  // output, the dropout-masked gradient, is only ever that product's A operand and -- as row tiles -- the weight gradient's)
  static const bool lnb_env = [] { const char* v = getenv("RD_LNB_FUSE"); return !(v && atoi(v) == 0); }();
  const bool lnf = tw && lnb_env && rowgemm_lnb_ok(e.nhid, e.D) && rowgemm_lnb_ok(e.D, e.D);
  RD_REQUIRE(!tp || lnf, "token plan: needs the LayerNorm-backward prologue path (RD_LNB_FUSE)");
  // the whole row-local chain (LayerNorm2' .. out_proj') as ONE launch (rd_encfuse.hip); same arithmetic as the three launches below
  const bool fuse = lnf && encfuse_ok(e.D, e.nhid);
  const int lnrows = fuse ? encfuse_part_rows(e.M) : (lnf ? rowgemm_lnb_part_rows(e.M) : lnb);
  if (fuse && (rc = launch_enc_pre_bwd(e.M, e.D, e.nhid, dy, v.s2, v.st2, w->norm2_w, v.h, v.s1, v.st1, w->norm1_w, v.pl[5][0], v.pl[6][0],
                                       v.pl[4][0], ws.ds2, ws.ds1, ws.da, ws.lnpart, ws.lnpart1, ws.dt[0], ws.dt[1], ws.dt[2], p_drop, seed,
                                       SITE_FFN_OUT + L, SITE_ATTN_OUT + L, tp ? tp + plan::I_MLIVE : nullptr,
                                       enc_lean(tp, tw) ? (const void*)v.h : nullptr, st))) return rc;
  // ---- LayerNorm 2:  ds2 (residual path), df = ds2 o mask(ffn out) -------------------------------
  if (!lnf && (rc = launch_ln_bwd(dy, v.s2, v.st2, w->norm2_w, ws.ds2, ws.df, ws.lnpart, (int)e.M, e.D, p_drop, seed,
                                  SITE_FFN_OUT + L, st))) return rc;
  // lnpart is a [blocks, 2D] matrix (dgamma | dbeta per block): column-sum it in fixed order
  if (!tw && (rc = launch_colsum2(ws.lnpart, lnb, 2 * e.D, 2 * e.D, g->norm2_w, e.D, g->norm2_b, ws.lnred, st))) return rc;
  // ---- FFN ---------------------------------------------------------------------------------------
  if (!tw && !tw2 && (rc = linear_bwd_w(e.M, e.D, e.nhid, ws.df, v.h, g->lin2_w, g->lin2_b, ws.splitk, ws.colsum, sw))) return rc;
  if (fuse) {
  } else if (rg) {                                                 // du = (df W2) gated by h>0, * keep
    if (tw) rowgemm_export_next(ws.dt[0]);
    if (lnf) {
      if ((rc = launch_rowgemm_lnb(e.M, e.nhid, e.D, dy, v.s2, v.st2, w->norm2_w, ws.ds2, ws.lnpart, p_drop, seed, SITE_FFN_OUT + L,
                                   v.pl[5][0], ws.du, e.nhid, v.h, e.nhid, p_drop > 0.f ? keep : 0.f, st))) return rc;
    } else if ((rc = launch_rowgemm(e.M, e.nhid, e.D, ws.df, e.D, v.pl[5][0], v.pl[5][1], ws.du, e.nhid, nullptr, 0, v.h, e.nhid,
                                    p_drop > 0.f ? keep : 0.f, nullptr, 0, 0.f, 0, 0, st))) return rc;
  } else if ((rc = linear_bwd_x_t(e.M, e.D, e.nhid, ws.df, w->lin2_w, panel ? v.pl[5][0] : nullptr, ws.du, v.h, p_drop > 0.f ? keep : 0.f, nullptr, st)))
    return rc;
  if (!tw && !tw2 && (rc = linear_bwd_w(e.M, e.nhid, e.D, ws.du, v.x1, g->lin1_w, g->lin1_b, ws.splitk, ws.colsum, sw))) return rc;
  if (fuse) {
  } else if (rg) {
    if (tw) rowgemm_export_next(ws.dt[1]);
    if ((rc = launch_rowgemm(e.M, e.D, e.nhid, ws.du, e.nhid, v.pl[6][0], v.pl[6][1], ws.dx1, e.D, nullptr, 0, nullptr, 0, 0.f,
                             ws.ds2, e.D, 0.f, 0, 0, st))) return rc;
  } else if ((rc = linear_bwd_x_t(e.M, e.nhid, e.D, ws.du, w->lin1_w, panel ? v.pl[6][0] : nullptr, ws.dx1, nullptr, 0.f, ws.ds2, st))) return rc;
  // ---- LayerNorm 1 -------------------------------------------------------------------------------
  if (!lnf && (rc = launch_ln_bwd(ws.dx1, v.s1, v.st1, w->norm1_w, ws.ds1, ws.dout, tw ? ws.lnpart1 : ws.lnpart, (int)e.M, e.D,
                                  p_drop, seed, SITE_ATTN_OUT + L, st))) return rc;
  if (!tw && (rc = launch_colsum2(ws.lnpart, lnb, 2 * e.D, 2 * e.D, g->norm1_w, e.D, g->norm1_b, ws.lnred, st))) return rc;
  // ---- attention output projection ---------------------------------------------------------------
  if (!tw && !tw2 && (rc = linear_bwd_w(e.M, e.D, e.D, ws.dout, v.attn, g->out_proj_w, g->out_proj_b, ws.splitk, ws.colsum, sw)))
    return rc;
  if (fuse) {
  } else if (rg) {
    if (tw) rowgemm_export_next(ws.dt[2]);
    if (lnf) {
      if ((rc = launch_rowgemm_lnb(e.M, e.D, e.D, ws.dx1, v.s1, v.st1, w->norm1_w, ws.ds1, ws.lnpart1, p_drop, seed,
                                   SITE_ATTN_OUT + L, v.pl[4][0], ws.da, e.D, nullptr, 0, 0.f, st))) return rc;
    } else if ((rc = launch_rowgemm(e.M, e.D, e.D, ws.dout, e.D, v.pl[4][0], v.pl[4][1], ws.da, e.D, nullptr, 0, nullptr, 0, 0.f,
                                    nullptr, 0, 0.f, 0, 0, st))) return rc;
  } else if ((rc = linear_bwd_x_t(e.M, e.D, e.D, ws.dout, w->out_proj_w, panel ? v.pl[4][0] : nullptr, ws.da, nullptr, 0.f, nullptr, st))) return rc;
  // ---- attention core ----------------------------------------------------------------------------
  AttnArgs a{};
  a.qkv = v.qkv; a.mask = mask; a.out = v.attn; a.lse = v.lse; a.dout = ws.da; a.dqkv = ws.dqkv; a.delta = ws.delta;
  a.T = e.T; a.B = e.B; a.D = e.D; a.H = e.H; a.hd = e.Hd;
  a.scale = 1.0f / sqrtf((float)e.Hd); a.p_drop = p_drop; a.seed = seed; a.site = SITE_ATTN_PROB + L; a.seed_cell = seed_cell();
  a.plan = tp;
  // fused form (rd_attnfuse.hip): attention backward + dx = dqkv W_in + ds1 + the row tiles of x and dqkv, one launch
  const bool afuse = tp && tw && fuse && attnfuse_ok(e.T, e.D, e.H, e.Hd);
  if (afuse) {
    if ((rc = launch_attn_fused_bwd(x, v.afw, v.abw, w->in_proj_b, tp, e.T, e.B, e.D, e.H, e.Hd, p_drop, seed, SITE_ATTN_PROB + L, v.attn, v.lse,
                                    ws.da, ws.ds1, dx, v.xt[0], ws.dt[3], st))) return rc;
  } else if (attn_big(e)) {
    if ((rc = attn_big_bwd(a, v.pbig, v.pdbig, ws.dsbig, st))) return rc;
  } else if (e.T <= TS) {
    if ((rc = dispatch_attn(a, 3, st))) return rc;          // single tile: S, P, dP, dS formed once
  } else {
    if ((rc = dispatch_attn(a, 1, st))) return rc;
    if ((rc = dispatch_attn(a, 2, st))) return rc;
  }
  // ---- input projection --------------------------------------------------------------------------
  if (!tw && !tw2 && (rc = linear_bwd_w(e.M, 3 * e.D, e.D, ws.dqkv, x, g->in_proj_w, g->in_proj_b, ws.splitk, ws.colsum, sw)))
    return rc;
  if (afuse) {
  } else if (tw) rowgemm_export_next(ws.dt[3]);
  if (afuse) rc = RD_OK;
  else if (rg && rowgemm_ok(e.D, 3 * e.D, 3 * e.D, e.D))      // dx = dqkv W_in + ds1: row-block form, K = 3D (was 70 us as a tiled GEMM)
    rc = launch_rowgemm(e.M, e.D, 3 * e.D, ws.dqkv, 3 * e.D, v.pl[7][0], v.pl[7][1], dx, e.D, nullptr, 0, nullptr, 0, 0.f, ws.ds1,
                        e.D, 0.f, 0, 0, st);
  else
    rc = linear_bwd_x_t(e.M, 3 * e.D, e.D, ws.dqkv, w->in_proj_w, panel ? v.pl[7][0] : nullptr, dx, nullptr, 0.f, ws.ds1, st);
  if (rc) return rc;
  if (tw) {
    TileWgradJob jobs[4] = {
        {ws.dt[3], v.xt[0], ws.twpart[3], g->in_proj_w, g->in_proj_b, 3 * e.D, e.D, nullptr, 0, 0, 0, 0, 0},      // dqkv^T x
        {ws.dt[1], v.xt[2], ws.twpart[1], g->lin1_w, g->lin1_b, e.nhid, e.D, nullptr, 0, 0, 0, 0, 0},             // du^T x1
        {ws.dt[0], v.xt[3], ws.twpart[0], g->lin2_w, g->lin2_b, e.D, e.nhid, nullptr, 0, 0, 0, 0, 0},             // df^T h
        {ws.dt[2], v.xt[1], ws.twpart[2], g->out_proj_w, g->out_proj_b, e.D, e.D, nullptr, 0, 0, 0, 0, 0}};       // dout^T attn
    if (afuse) {   // x and dqkv tiles come from the fused attention backward: per-sample chunk space, head-padded dqkv columns
      jobs[0].N = attnfuse_padded_cols(e.H); jobs[0].s32 = tp + plan::I_SCHUNK; jobs[0].S = (int)attn_chunks(e);
      jobs[0].hd = e.Hd; jobs[0].hdp = 16 * attnfuse_nth(); jobs[0].H = e.H; jobs[0].D = e.D;
    }
    const TileColsumJob cs[2] = {{ws.lnpart, lnrows, 2 * e.D, e.D, g->norm2_w, g->norm2_b},
                                 {ws.lnpart1, lnrows, 2 * e.D, e.D, g->norm1_w, g->norm1_b}};
    return launch_tile_wgrad(e.M, 4, jobs, v.ones, 2, cs, st, tp ? tp + plan::I_S32 : nullptr);
  }
  if (tw2) {
    // every operand of the four products is complete and still in place (ws.* are this call's, v.* the forward's): one conversion
    // launch, then the stream.  The LayerNorm column sums went their own way above (launch_colsum2).
    const float* src[8] = {ws.df, ws.du, ws.dout, ws.dqkv, x, v.attn, v.x1, v.h};
    const long ld[8] = {e.D, e.nhid, e.D, 3L * e.D, e.D, e.D, e.D, e.nhid};
    const int cols[8] = {e.D, e.nhid, e.D, 3 * e.D, e.D, e.D, e.D, e.nhid};
    void* dst[8] = {ws.dt[0], ws.dt[1], ws.dt[2], ws.dt[3], v.xt[0], v.xt[1], v.xt[2], v.xt[3]};
    if ((rc = launch_rows_to_tiles(e.M, 8, src, ld, cols, dst, st))) return rc;
    TileWgradJob jobs[4] = {
        {ws.dt[3], v.xt[0], ws.twpart[3], g->in_proj_w, g->in_proj_b, 3 * e.D, e.D, nullptr, 0, 0, 0, 0, 0},      // dqkv^T x
        {ws.dt[1], v.xt[2], ws.twpart[1], g->lin1_w, g->lin1_b, e.nhid, e.D, nullptr, 0, 0, 0, 0, 0},             // du^T x1
        {ws.dt[0], v.xt[3], ws.twpart[0], g->lin2_w, g->lin2_b, e.D, e.nhid, nullptr, 0, 0, 0, 0, 0},             // df^T h
        {ws.dt[2], v.xt[1], ws.twpart[2], g->out_proj_w, g->out_proj_b, e.D, e.D, nullptr, 0, 0, 0, 0, 0}};       // dout^T attn
    return launch_tile_wgrad(e.M, 4, jobs, v.ones, 0, nullptr, st, nullptr);
  }
  return rc;
}

// The attention core alone (torch F.multi_head_attention_forward between in_proj and out_proj, as used by the encoder layer of
// code/models_rd.py:235-237,358): qkv [T,B,3D] -> out [T,B,D], lse [B,H,T]; backward: dout -> dqkv.  Same kernels and dispatch as
// inside rd_encoder_layer_fwd/bwd; exposed so that this sub-graph (no ReLU gate in it) can be held to a tight bound against the
// oracle by itself (tests/test_gpu_parity.py::test_attention_core_vs_float64).
extern "C" int rd_attention_fwd(const rd_shape* s, int32_t layer, const float* qkv, const uint8_t* mask, float p_drop, uint64_t seed,
                                float* out, float* lse, void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(qkv && mask && out && lse, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  const EncDims e = enc_dims(s);
  RD_REQUIRE(!attn_big(e), "rd_attention_fwd: head_dim %d > 96 runs as materialised products inside the layer only", e.Hd);
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.out = out; a.lse = lse;
  a.T = e.T; a.B = e.B; a.D = e.D; a.H = e.H; a.hd = e.Hd;
  a.scale = 1.0f / sqrtf((float)e.Hd); a.p_drop = p_drop; a.seed = seed; a.site = SITE_ATTN_PROB + (uint32_t)layer; a.seed_cell = seed_cell();
  a.plan = token_plan();
  return dispatch_attn(a, 0, (hipStream_t)stream);
}
// delta_ws: [B,H,T] floats of scratch
extern "C" int rd_attention_bwd(const rd_shape* s, int32_t layer, const float* qkv, const uint8_t* mask, float p_drop, uint64_t seed,
                                const float* out, const float* lse, const float* dout, float* dqkv, float* delta_ws, void* stream) {
  int rc = check_enc(s);
  if (rc) return rc;
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(qkv && mask && out && lse && dout && dqkv && delta_ws, "NULL tensor");
  RD_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop must be in [0,1)");
  const EncDims e = enc_dims(s);
  RD_REQUIRE(!attn_big(e), "rd_attention_bwd: head_dim %d > 96 runs as materialised products inside the layer only", e.Hd);
  AttnArgs a{};
  a.qkv = qkv; a.mask = mask; a.out = const_cast<float*>(out); a.lse = const_cast<float*>(lse); a.dout = dout; a.dqkv = dqkv; a.delta = delta_ws;
  a.T = e.T; a.B = e.B; a.D = e.D; a.H = e.H; a.hd = e.Hd;
  a.scale = 1.0f / sqrtf((float)e.Hd); a.p_drop = p_drop; a.seed = seed; a.site = SITE_ATTN_PROB + (uint32_t)layer; a.seed_cell = seed_cell();
  a.plan = token_plan();
  hipStream_t st = (hipStream_t)stream;
  if (e.T <= TS) return dispatch_attn(a, 3, st);
  if ((rc = dispatch_attn(a, 1, st))) return rc;
  return dispatch_attn(a, 2, st);
}

extern "C" int rd_masked_mean_fwd(const rd_shape* s, int32_t D, const float* r, const uint8_t* mask,
                                  const int64_t* lengths, float* out, int32_t ldo, void* stream) {
  RD_REQUIRE(s && s->T > 0 && s->B >= 0 && D > 0 && ldo >= D, "bad arguments");
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(r && mask && lengths && out, "NULL tensor");
  hipLaunchKernelGGL(k_masked_mean_fwd, dim3(s->B, cdiv(D, 64)), dim3(1024), 0, (hipStream_t)stream, r, mask, lengths,
                     out, s->T, s->B, D, ldo);
  return check_launch("k_masked_mean_fwd");
}

extern "C" int rd_masked_mean_bwd(const rd_shape* s, int32_t D, const float* dout, int32_t ldo,
                                  const uint8_t* mask, const int64_t* lengths, float* dr, void* stream) {
  RD_REQUIRE(s && s->T > 0 && s->B >= 0 && D > 0 && ldo >= D, "bad arguments");
  if (s->B == 0) return RD_OK;
  RD_REQUIRE(dout && mask && lengths && dr, "NULL tensor");
  const long n = (long)s->T * s->B * D;
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_masked_mean_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, mask, lengths, dr,
                     s->T, s->B, D, ldo);
  return check_launch("k_masked_mean_bwd");
}
