// rd_optim.hip -- Adam update over one flat parameter buffer (the optimizer of code/Raindrop.py:256,
// torch.optim.Adam(lr=1e-4) with default betas / eps, no amsgrad).  With the live parameters and
// their gradients held in flat buffers (raindrop_amd/dp.py) the whole update is one elementwise pass.
#include "rd_common.h"

namespace rd {
namespace {

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                              float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  RD_TOUCH_CODE_FIRST(2432, blockIdx.x, 64);             // own code -> L2 by the first workgroups (rd_common.h; 3.0 KB kernel)
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

}  // namespace
}  // namespace rd

using namespace rd;

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
