// probe_rowstride.hip -- cost of per-sample tile loads from a time-major [T,B,C] tensor against a sample-major [B,T,C] one.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_rowstride tools/probe_rowstride.hip
// One workgroup per (sample b, head h) loads NT tiles of 64 rows x 80 floats the way the attention kernels do (4 threads per
// row, 16-byte chunks) and sums them; T = 60, B = 256, C = 456 (qkv of P19).  Prints us per launch for both layouts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* __restrict__ out, int T, int B, int C, long tstride,
                                         long bstride, unsigned long long* stamps) {
  const int tid = threadIdx.x, b = blockIdx.x >> 1, h = blockIdx.x & 1;
  const int t = tid >> 2;
  const float* base = x + ((long)(t < T ? t : 0) * tstride + (long)b * bstride) * C + h * 76;
  float4 v[NT][5];
  unsigned long long c0 = clock64();
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int c = 4 * (tid & 3) + 16 * i;
      v[n][i] = *reinterpret_cast<const float4*>(base + (n % 3) * 152 + (c < 76 ? c : 0));
    }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long c1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int i = 0; i < 5; ++i) s += v[n][i].x + v[n][i].y + v[n][i].z + v[n][i].w;
  unsigned long long c2 = clock64();
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x < 4) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = c2 - c1; }
}
int main() {
  const int T = 60, B = 256, C = 456;
  float *x, *out; unsigned long long* st;
  hipMalloc(&x, sizeof(float) * T * B * C); hipMalloc(&out, sizeof(float) * 512 * 256); hipMalloc(&st, 64);
  hipMemset(x, 0, sizeof(float) * T * B * C);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int layout = 0; layout < 2; ++layout) {
    const long ts = layout ? 1 : B, bs = layout ? T : 1;
    for (int nt = 0; nt < 2; ++nt) {
      for (int it = 0; it < 3; ++it) {
        if (nt) hipLaunchKernelGGL(k<5>, dim3(512), dim3(256), 0, 0, x, out, T, B, C, ts, bs, st);
        else hipLaunchKernelGGL(k<3>, dim3(512), dim3(256), 0, 0, x, out, T, B, C, ts, bs, st);
      }
      hipEventRecord(e0);
      for (int it = 0; it < 20; ++it) {
        if (nt) hipLaunchKernelGGL(k<5>, dim3(512), dim3(256), 0, 0, x, out, T, B, C, ts, bs, st);
        else hipLaunchKernelGGL(k<3>, dim3(512), dim3(256), 0, 0, x, out, T, B, C, ts, bs, st);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[8]; hipMemcpy(h, st, 64, hipMemcpyDeviceToHost);
      printf("%s  tiles=%d  %.2f us/launch   wg0: issue %llu wait+sum %llu cycles   wg1: issue %llu wait+sum %llu\n",
             layout ? "[B,T,C]" : "[T,B,C]", nt ? 5 : 3, ms * 1000 / 20, h[0], h[1], h[2], h[3]);
    }
  }
  return 0;
}
