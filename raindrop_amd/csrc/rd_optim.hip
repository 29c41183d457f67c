// rd_optim.hip -- Adam update over one flat parameter buffer (the optimizer of code/Raindrop.py:256,
// torch.optim.Adam(lr=1e-4) with default betas / eps, no amsgrad).  With the live parameters and
// their gradients held in flat buffers (raindrop_amd/dp.py) the whole update is one elementwise pass.
#include "rd_common.h"

namespace rd {
namespace {

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                              float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  RD_TOUCH_CODE_FIRST(RD_TL_ADAM, blockIdx.x, 64);             // own code -> L2 by the first workgroups (rd_common.h; 3 012-byte kernel)
  const long i4 = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float step_size = lr / bc1;
  if (i4 + 3 < n) {
    float4 pp = *reinterpret_cast<float4*>(p + i4), gg = *reinterpret_cast<const float4*>(g + i4);
    float4 mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gr = G[c] + wd * P[c];
      M[c] = b1 * M[c] + (1.f - b1) * gr;
      V[c] = b2 * V[c] + (1.f - b2) * gr * gr;
      P[c] -= step_size * M[c] / (sqrtf(V[c]) / bc2_sqrt + eps);
    }
    *reinterpret_cast<float4*>(p + i4) = pp; *reinterpret_cast<float4*>(m + i4) = mm;
    *reinterpret_cast<float4*>(v + i4) = vv;
  } else {
    for (long i = i4; i < n; ++i) {
      const float gr = g[i] + wd * p[i];
      m[i] = b1 * m[i] + (1.f - b1) * gr;
      v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
      p[i] -= step_size * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    }
  }
}

// What kind of box is this?  (DESIGN.md "Box variance")  32 KB of straight-line code -- 4096 dependent 8-byte v_fma_f32, nothing the
// compiler can fold --, one wave, clock64 around it: cold (its code in no cache) about 80 ticks per 64-byte line where the hardware
// fetches instructions ahead, 400-450 where it does not; warm 45-55 / 75-95.  tools/probe_clocks.hip is the standalone form.
// The same update with the optimizer's step state on the DEVICE: a captured launch cannot carry the host's bias corrections (they
// change every step).  The state is eight doubles {t, beta1^t, beta2^t, lr, -, weight_decay, -, -} holding the values OF THE STEP
// BEING APPLIED; this kernel only READS it.  The state is advanced -- t += 1, beta^t *= beta -- by an earlier launch of the same
// step: the step's first launch where there is one (rd_set_adam_state registers the cell, rd_step_begin's plan workgroup does it
// next to the dropout seed bump), else rd_adam_state_advance (one thread).  Rounds 5's form advanced a second slot from workgroup 0
// of THIS launch: workgroups dispatched after that store became visible applied step t + 2's corrections (a race wherever the grid
// does not fit the device in one round).  Round 6's first fix -- an arrival ticket, the last workgroup advances -- was
// race-free and cost 4.5 us per step (profiles/r06_adam_ticket_vs_begin.txt: 9.85 against 5.4 us: every workgroup waits for the
// return of a device-scope atomic on ONE line, the last one for a fence and a second atomic on top).
// lr and weight_decay live in the cell too: a learning-rate schedule is an 8-byte copy, not a new capture.
// (beta^t by repeated multiplication differs from pow() by ~t 2^-53 relative: far below the float the correction is rounded to.)
struct AdamState { double t, p1, p2, lr, pad0, wd, pad1, pad2; };
__global__ __launch_bounds__(256) void k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long n, float b1, float b2, float eps,
                                                  const AdamState* __restrict__ state) {
  RD_TOUCH_CODE_FIRST(RD_TL_ADAM_DEV, blockIdx.x, 64);
  const float lr = (float)state->lr, wd = (float)state->wd;
  const float bc1 = (float)(1.0 - state->p1), bc2_sqrt = (float)sqrt(1.0 - state->p2);
  const long i4 = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const float step_size = lr / bc1;
  if (i4 + 3 < n) {
    float4 pp = *reinterpret_cast<float4*>(p + i4), gg = *reinterpret_cast<const float4*>(g + i4);
    float4 mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float gr = G[c4] + wd * P[c4];
      M[c4] = b1 * M[c4] + (1.f - b1) * gr;
      V[c4] = b2 * V[c4] + (1.f - b2) * gr * gr;
      P[c4] -= step_size * M[c4] / (sqrtf(V[c4]) / bc2_sqrt + eps);
    }
    *reinterpret_cast<float4*>(p + i4) = pp; *reinterpret_cast<float4*>(m + i4) = mm;
    *reinterpret_cast<float4*>(v + i4) = vv;
  } else {
    for (long i = i4; i < n; ++i) {
      const float gr = g[i] + wd * p[i];
      m[i] = b1 * m[i] + (1.f - b1) * gr;
      v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
      p[i] -= step_size * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    }
  }
}
__global__ void k_adam_advance(double* state, float b1, float b2) { adam_state_advance(state, b1, b2); }

__global__ __launch_bounds__(64) void k_ifetch_probe(float* out, unsigned long long* t, float a, float b) {
  float x = (float)threadIdx.x;
  const unsigned long long c0 = clock64();
  asm volatile(".rept 4096\n\tv_fma_f32 %0, %0, %1, %2\n\t.endr" : "+v"(x) : "v"(a), "v"(b));
  const unsigned long long c1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}

}  // namespace
}  // namespace rd

using namespace rd;

// not part of the ABI (include/raindrop_hip_debug.h): ticks of the probe block -> ticks_dev[0]; scratch_dev: 64 floats
extern "C" void rd_debug_ifetch_probe(void* ticks_dev, void* scratch_dev, void* stream) {
  hipLaunchKernelGGL(k_ifetch_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (float*)scratch_dev, (unsigned long long*)ticks_dev,
                     1.0000001f, 0.5f);
}

extern "C" int rd_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                            float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                            void* stream) {
  RD_REQUIRE(n >= 0 && step >= 1, "bad n / step");
  RD_REQUIRE(param && grad && exp_avg && exp_avg_sq, "NULL tensor");
  RD_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
               reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
             "buffers must be 16-byte aligned");
  if (n == 0) return RD_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const long threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (long)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  return check_launch("k_adam");
}

// rd_adam_step with the step state on the device: `state` = 64 bytes {t, beta1^t, beta2^t, lr, -, weight_decay, -, -} (doubles) OF THE
// STEP BEING APPLIED (advanced beforehand: rd_adam_state_advance, or the step's first launch after rd_set_adam_state) -- ONE launch a
// hipGraph can replay, read-only on the state.
extern "C" int rd_adam_step_dev(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                float beta1, float beta2, float eps, const void* state, void* stream) {
  RD_REQUIRE(n > 0 && state != nullptr, "bad n / NULL optimizer state");
  RD_REQUIRE(param && grad && exp_avg && exp_avg_sq, "NULL tensor");
  RD_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
               reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(state)) & 15) == 0, "buffers must be 16-byte aligned");
  const long threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (long)n, beta1, beta2, eps, (const AdamState*)state);
  return check_launch("k_adam_dev");
}

extern "C" int rd_adam_state_advance(void* state, float beta1, float beta2, void* stream) {
  RD_REQUIRE(state != nullptr && (reinterpret_cast<uintptr_t>(state) & 15) == 0, "NULL / misaligned optimizer state");
  hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, (double*)state, beta1, beta2);
  return check_launch("k_adam_advance");
}
