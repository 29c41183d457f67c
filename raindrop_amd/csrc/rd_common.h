// rd_common.h -- shared host/device helpers for libraindrop_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/raindrop_hip.h"
#include "rd_touch_gen.h"      // GENERATED own-code touch lengths (raindrop_amd/build.py TOUCH_SITES)

namespace rd {

// thread-local last-error message (rd_last_error()).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
  return RD_OK;
}

// runtime call that must succeed: returns the hipError_t (positive) through the C-ABI error channel
#define RD_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t rd_e_ = (call);                                                            \
    if (rd_e_ != hipSuccess) return rd::fail((int)rd_e_, "%s: %s", #call, hipGetErrorString(rd_e_)); \
  } while (0)
// dynamic-LDS opt-in of a kernel, done once per process (never inside a stream capture after the first call)
#define RD_LDS_ATTR(kernel, bytes)                                                        \
  do {                                                                                    \
    static const hipError_t rd_attr_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (rd_attr_ != hipSuccess) return rd::fail((int)rd_attr_, "hipFuncSetAttribute(%s): %s", #kernel, hipGetErrorString(rd_attr_)); \
  } while (0)

#define RD_REQUIRE(cond, ...)                           \
  do {                                                  \
    if (!(cond)) return rd::fail(RD_EINVAL, __VA_ARGS__); \
  } while (0)

// optional device cell added to every dropout seed (rd_set_seed_cell); nullptr when unset
const uint64_t* seed_cell();

// process-wide arithmetic mode of the dense contractions (rd_set_precision / env RD_PRECISION)
int precision();

// Side branch for TRAILING launches (rd_set_side_stream, per host thread, read at enqueue like the seed cell).  A few launches of
// the step produce only parameter gradients that nothing later in the backward chain reads -- the slice reduces of the weight-
// gradient streams, the head's weight-gradient tiles -- yet in one stream each sits on the critical path for its 5-15 us.
// side_fork(main) orders a registered side stream behind everything enqueued on `main` so far and returns it (or `main` when none
// is registered): the trailing launch goes there and runs beside what follows on `main`.  side_join(main) makes `main` wait for the
// side stream; the OWNER of the stream (raindrop_amd.step.TrainStep) calls rd_side_join at the end of a step / of a captured graph,
// the library itself before it rewrites a buffer a forked launch reads.  Under stream capture both become graph dependencies.
hipStream_t side_fork(hipStream_t main_stream);
int side_join(hipStream_t main_stream);

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// one weight matrix W [N,K] (row-major fp32) -> native matrix-core operand tiles [ceil16(rows)/16][ceil32(cols)/32][hi, lo][64][8]
// (rows = N, cols = K; transpose: rows = K, cols = N, i.e. the tiles of W^T); rd_rowgemm.hip: launch_wsplit_specs
struct WsplitSpec { const float* W; int N, K, transpose; void* tiles; };
int launch_wsplit_specs(int njobs, const WsplitSpec* specs, int nones, void* const* ones, hipStream_t st);
int launch_wsplit_plan(int njobs, const WsplitSpec* specs, int nones, void* const* ones, const int64_t* lengths, int32_t* plan_out,
                       int B, int T, uint64_t* seed_cell_dev, uint64_t delta, hipStream_t st);
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global
// load and store (s_waitcnt vmcnt(0)): with a register-resident weight panel in flight that serialises
// the weight stream with each phase change.  Use ONLY where no thread reads global memory that another
// thread of the same launch wrote before the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// uniform 64-bit load through the scalar cache: does not queue behind the vector loads in flight
__device__ __forceinline__ uint64_t load_uniform_u64(const uint64_t* p) {
  uint64_t v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// uniform 32-bit load through the scalar cache (token-plan look-ups of a workgroup's sample)
__device__ __forceinline__ int load_uniform_i32(const int32_t* p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// two independent uniform loads, one wait
__device__ __forceinline__ void load_uniform_2xi32(const int32_t* p, const int32_t* q, int& x, int& y) {
  asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(x), "=&s"(y) : "s"(p), "s"(q) : "memory");
}

// Device-side Adam step state (rd_optim.hip): {t, beta1^t, beta2^t, lr, -, weight_decay, -, -}; one thread advances it once per step
// in a launch that precedes the update (rd_adam_state_advance, or the plan workgroup of rd_step_begin).
__device__ __forceinline__ void adam_state_advance(double* st, float b1, float b2) {
  st[0] += 1.0; st[1] *= (double)b1; st[2] *= (double)b2;
}
// the cell registered by rd_set_adam_state on this host thread (nullptr: none); read when rd_step_begin is ENQUEUED
struct AdamCellReg { double* state; float b1, b2; };
AdamCellReg adam_cell();

// Request this kernel's OWN code into the L2 of the XCD it runs on: lane t asks for one dword of the t-th 128-byte line after the
// current PC (`bytes` <= the code that follows; lanes past the range sit out), then the requesting waves wait once.  Why: a kernel
// of the step runs once per step, ~0.9 GB of traffic after its previous run -- its code is in no cache, and on the pool's SLOW boxes
// instruction fetch has no look-ahead: every 128 bytes of straight-line code are a serialized ~840-tick trip to memory
// (tools/probe_clocks.hip: 405 ticks per 64-byte line cold against 80 warm; the fast boxes prefetch: 81 / 51).  One parallel
// batch of loads up front turns them into L2 hits.  Data loads and instruction fetches share the L2.
__device__ __forceinline__ void touch_own_code(int tid, int bytes) {
  if (bytes <= 0) return;                                   // uniform
  uint64_t pc;
  asm volatile("s_getpc_b64 %0" : "=s"(pc));
  const int nl = bytes >> 7;
  if (tid < nl) {
    const char* p = reinterpret_cast<const char*>(pc) + (size_t)tid * 128;
    int t;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory");
  }
}

// -DRD_NO_CODE_TOUCH: A/B builds without it (python -m raindrop_amd.build --variant notouch -DRD_NO_CODE_TOUCH).  `first` bounds the
// workgroups that do it in launches of many small workgroups (the first ones dispatched cover every XCD).
#ifdef RD_NO_CODE_TOUCH
#define RD_TOUCH_CODE(bytes) ((void)0)
#define RD_TOUCH_CODE_FIRST(bytes, lin, first) ((void)0)
#else
#define RD_TOUCH_CODE(bytes) ::rd::touch_own_code((int)threadIdx.x, (bytes))
#define RD_TOUCH_CODE_FIRST(bytes, lin, first) ::rd::touch_own_code((int)threadIdx.x, (int)(lin) < (first) ? (bytes) : 0)
#endif

// The kernels OUTSIDE the P19 step (tiled / panel GEMMs, row-block products, multi-tile and padded-layout attention, LayerNorm
// kernels: P12, PAM, SYN256, the eager surface) carry the same prologue (RD_TOUCH_CODE_X).  Written in round 4, measured and made the
// default in round 5 (one call, alternating libraries, a box whose probe says "fast": P12 bf16 1.460 -> 1.428 ms/step, PAM 2.93 ->
// 2.85, SYN256 7.75 -> 7.57, P19 unchanged; profiles/r05_touchall_ab.txt).  -DRD_NO_TOUCH_ALL builds the library without it (A/B).
#if !defined(RD_NO_TOUCH_ALL) && !defined(RD_NO_CODE_TOUCH)
#define RD_TOUCH_CODE_X(bytes, lin, first) ::rd::touch_own_code((int)threadIdx.x, (int)(lin) < (first) ? (bytes) : 0)
#else
#define RD_TOUCH_CODE_X(bytes, lin, first) ((void)0)
#endif

// a uniform 64-bit value and two uniform 32-bit ones, three independent scalar loads behind one wait
__device__ __forceinline__ void load_uniform_u64_2xi32(const uint64_t* l, const int32_t* p, const int32_t* q, uint64_t& v, int& x, int& y) {
  asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dword %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(v), "=&s"(x), "=&s"(y) : "s"(l), "s"(p), "s"(q) : "memory");
}

// Sum over the 64 lanes of a wavefront on the DPP path, result uniform (every lane gets it).  row_shr 1/2/4/8 leave each
// 16-lane row's total in its last lane, row_bcast15 / row_bcast31 carry the totals up to lane 63, v_readlane makes it
// uniform: six VALU instructions with DPP modifiers instead of six LDS-crossbar round trips (`__shfl_xor` compiles to
// ds_bpermute_b32; a dependent chain of those is ~6 x 100 cycles, twice that on the pool's slow boxes).  Fixed order.
__device__ __forceinline__ float wave_sum64_dpp(float v) {
  const int zero = 0;
#define RD_DPP_ADD(ctrl, rmask)                                                                              \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(zero, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false))
  RD_DPP_ADD(0x111, 0xf);   // row_shr:1
  RD_DPP_ADD(0x112, 0xf);   // row_shr:2
  RD_DPP_ADD(0x114, 0xf);   // row_shr:4
  RD_DPP_ADD(0x118, 0xf);   // row_shr:8
  RD_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
  RD_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef RD_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ------------------------------------------------------------------------------------------
// Generic fp32 GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact fp32, == fmaf chain).
//   C(m,n) = epilogue( sum_k A(m,k) * B(n,k) )
// A(m,k) lives at A[m*sa_m + k*sa_k], B(n,k) at B[n*sb_n + k*sb_k]; either stride may be the
// unit one, so NT (x W^T), NN (dy W) and TN (dy^T x) products are the same kernel.
// ------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A; long sa_m, sa_k;
  const float* B; long sb_n, sb_k;
  float* C; long sc_m;            // C(m,n) at C[m*sc_m + n] unless scatter != 0
  int M, N, K;
  int nsplit; int k_per_split; long sc_split;   // split-K: partial z goes to C + z*sc_split, raw
  // optional second problem of identical shape in the same launch (z in [nsplit, 2*nsplit)): the two
  // message-passing layers' weight gradients share one grid
  const float* A2; const float* B2; float* C2; float* rowsum2;
  int tile_hint;                  // 1: force the 128x128 tile (split-K weight gradients: halves operand re-reads)
  int xcd_swizzle;                // set by launch_gemm (env RD_GEMM_XCD, default 1)
  int slice_xcd;                  // set by launch_gemm: split-K slices pinned to XCDs (env RD_SPLITK_XCD, default 1)
  float* rowsum; long rowsum_split;   // optional: rowsum[z*rowsum_split + m] = sum_k A(m,k) over this split (bias
                                  // gradients ride along the weight-gradient product: no second pass over dy)
  // epilogue (ignored when nsplit > 1)
  const float* bias;              // [N]
  const float* rowscale; int rs_period;          // * rowscale[m % rs_period]
  const float* posmask; long pm_m;               // * (posmask[m*pm_m + n] > 0)
  const float* residual; long res_m;             // + residual[m*res_m + n]
  int relu;
  float cscale;                   // * cscale when != 0 (applied with posmask: dropout keep-scale in backward)
  float drop_p; uint64_t drop_seed; uint32_t drop_site; const uint64_t* seed_cell;   // Philox dropout on element m*N+n (after ReLU)
  // scatter == 1: rows are (b,f) pairs of a [B,F,K] tensor, columns (t,c); element goes to the
  // [T,B,ldz] layout z[(t*sB + b)*ldz + f*sd + c]   (code/models_rd.py:338-342)
  int scatter; int sB, sF, sd; long ldz;
  // token plan (rd_plan.h) for the scatter: first row / clamped length of SAMPLE b (plan brow / blen) or null -- element (b, f, t, c)
  // then goes to row sp_row0[b] + t of z and steps t >= sp_len[b] are not stored (those rows do not exist)
  const int32_t *sp_row0, *sp_len;
  unsigned long long* stamps;     // debug phase stamps (set by launch_gemm; null in normal runs)
  int one_product;                // RD_PREC_BF16: hi*hi only (set by launch_gemm)
  // batched form (nbatch > 1, nsplit <= 1): problem z = (o, i) = (z / batch_inner, z % batch_inner) reads A + o*a_bo + i*a_bi,
  // B + o*b_bo + i*b_bi and writes C + o*c_bo + i*c_bi (two-level strides: sample and head of the [T,B,3D] attention tensors)
  int nbatch, batch_inner; long a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;
  int res_batched;                // batched form: the residual is laid out like C (offset by the problem's C offset)
  // optional: B (as used: B(n,k), N x K) already split into native operand tiles by launch_wsplit_specs (rows = N, cols = K);
  // launch_gemm then runs the panel form (k_gemm_panel) when the product qualifies, and ignores B / sb_n / sb_k
  const void* Btiles; int bt_ntile, bt_nkc;       // tile counts: ceil(N / 16), ceil(K / 32)
};
int launch_gemm(const GemmArgs& a, hipStream_t st);
// split of the reduction length `red` of a [rows x cols] weight-gradient product into nsplit chunks
// of k_per_split (a multiple of 64) so that a few hundred workgroups exist; returns nsplit
int splitk_plan(long red, int rows, int cols, int* k_per_split);
// weight + bias gradient of a linear layer in one split-K product (ws: wgrad_ws_floats floats)
long wgrad_ws_floats(long M, int N, int K);
int launch_wgrad(long M, int N, int K, const float* dy, long lddy, const float* x, long ldx, float* dW,
                 float* db, float* ws, hipStream_t st);
// two weight gradients of identical shape in one launch pair (ws: 2 * wgrad_ws_floats floats)
int launch_wgrad2(long M, int N, int K, const float* dyA, const float* xA, float* dWA, float* dbA,
                  const float* dyB, const float* xB, float* dWB, float* dbB, float* ws, hipStream_t st);
// sum `nsplit` partials [nsplit][rows*cols] in fixed order into out
int launch_splitk_reduce(const float* part, int nsplit, long elems, float* out, hipStream_t st);
// same with partials `stride` floats apart; elements [0,e1) go to out1, [e1, e1+e2) to out2
int launch_splitk_reduce2(const float* part, int nsplit, long stride, long e1, float* out1, long e2, float* out2,
                          hipStream_t st);
// one or two problems in one launch; picks the wide (float4 x 16 split groups) kernel when alignment allows
int launch_splitk_reduce_pair(const float* partA, float* out1A, float* out2A, const float* partB, float* out1B,
                              float* out2B, int nsplit, long stride, long e1, long e2, hipStream_t st);
// out[n] = sum_m x[m*ldx + n], deterministic two-stage; ws needs colsum_ws_floats(M,N) floats
long colsum_ws_floats(int M, int N);
int launch_colsum(const float* x, int M, int N, long ldx, float* out, float* ws, hipStream_t st);
// same, columns [0,n1) to out1 and [n1,N) to out2 (M <= 8192)
int launch_colsum2(const float* x, int M, int N, long ldx, float* out1, int n1, float* out2, float* ws, hipStream_t st);

}  // namespace rd
