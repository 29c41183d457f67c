// probe_clocks.hip -- what kind of box is this?  (The pool's MI355X boxes come in two speeds, DESIGN "Box variance".)
//   shader clock as s_memtime sees it (clock64 ticks per 100-MHz wall_clock64 tick), dependent-load latency at three working-set
//   sizes (64 KB: L2 of one XCD, 64 MB: memory-side cache, 2 GB: HBM), and the device-wide read / write stream rate.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_clocks tools/probe_clocks.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_clock(unsigned long long* out) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - w0 < 100000) {}                          // 1 ms at 100 MHz
  out[0] = clock64() - c0; out[1] = wall_clock64() - w0;
}

// one lane chases `steps` dependent indices (each element on its own 128-byte line), starting at line `start`
__global__ void k_chase(const unsigned* __restrict__ next, unsigned start, int steps, unsigned long long* out) {
  unsigned i = start;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int s = 0; s < steps; ++s) i = next[(size_t)i * 32];
  out[0] = wall_clock64() - w0; out[1] = i; out[2] = clock64() - c0;
}

// 32 KB of straight-line code (4096 dependent 8-byte v_fma_f32, nothing the compiler can fold), one wave: cold vs warm instruction fetch
__global__ void k_icache(float* out, unsigned long long* t, float a, float b) {
  float x = (float)threadIdx.x;
  const unsigned long long c0 = clock64();
  asm volatile(".rept 4096\n\tv_fma_f32 %0, %0, %1, %2\n\t.endr" : "+v"(x) : "v"(a), "v"(b));
  const unsigned long long c1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n16, uint4* sink) {
  uint4 a = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[i]; a.x ^= v.x; a.y += v.y; a.z ^= v.z; a.w += v.w;
  }
  if (a.x == 0x1234567u && a.y == 99u) sink[0] = a;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("%s  CUs %d  clockRate %d kHz  memoryClockRate %d kHz  L2 %d KB\n", pr.name, pr.multiProcessorCount, pr.clockRate, pr.memoryClockRate, pr.l2CacheSize / 1024);
  unsigned long long* out; CK(hipMalloc(&out, 64));
  unsigned long long h[2];
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, out); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  }
  printf("shader clock by s_memtime: %.1f MHz (clock64 %llu ticks per %llu wall ticks of 10 ns)\n", (double)h[0] / ((double)h[1] / 100.0), h[0], h[1]);
  // pointer chase: a random cycle over the working set; the timed walk starts in the half the warm-up never touched, and is
  // shorter than the set for the large sizes (every step a first touch: cold latency)
  const size_t sizes[4] = {64ull << 10, 2ull << 20, 64ull << 20, 512ull << 20};
  const char* what[4] = {"64 KB (L2 hit)", "2 MB (L2 hit, other channels)", "64 MB (memory-side cache after a warm walk)", "512 MB (HBM, cold)"};
  for (int si = 0; si < 4; ++si) {
    const size_t nline = sizes[si] / 128;
    std::vector<unsigned> perm(nline); std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 g(7); std::shuffle(perm.begin() + 1, perm.end(), g);
    std::vector<unsigned> nxt(nline * 32, 0u);
    for (size_t i = 0; i < nline; ++i) nxt[(size_t)perm[i] * 32] = perm[(i + 1) % nline];
    unsigned* d; CK(hipMalloc(&d, nline * 128)); CK(hipMemcpy(d, nxt.data(), nline * 128, hipMemcpyHostToDevice));
    const int steps = (int)std::min<size_t>(nline, 100000);
    const int passes = si < 3 ? 2 : 1;                              // small sets: second walk over the same lines = warm
    for (int r = 0; r < passes; ++r) { hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, d, perm[nline / 2], steps, out); CK(hipDeviceSynchronize()); }
    unsigned long long h3[3]; CK(hipMemcpy(h3, out, 24, hipMemcpyDeviceToHost));
    printf("dependent-load latency, %-44s: %6.0f ns = %6.0f s_memtime ticks\n", what[si], (double)h3[0] * 10.0 / steps, (double)h3[2] / steps);
    CK(hipFree(d));
  }
  {  // instruction fetch: the block right after a 1-GB write (L2 and memory-side cache hold none of it), then again at once (warm)
    float* xo; CK(hipMalloc(&xo, 64 * 4));
    uint4* fl; CK(hipMalloc(&fl, 1024ull << 20));
    for (int r = 0; r < 2; ++r) {
      hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, fl, (1024ull << 20) / 16); CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(k_icache, dim3(1), dim3(64), 0, 0, xo, out, 1.0000001f, 0.5f); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
      const unsigned long long cold = h[0];
      hipLaunchKernelGGL(k_icache, dim3(1), dim3(64), 0, 0, xo, out, 1.0000001f, 0.5f); CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
      printf("32 KB of straight-line code (4096 dependent v_fma_f32, one wave): cold %llu ticks (%.0f per 64-byte line), warm %llu ticks (%.0f per line)\n",
             cold, cold / 512.0, h[0], h[0] / 512.0);
    }
    CK(hipFree(fl));
  }
  // streams
  const size_t bytes = 1024ull << 20, n16 = bytes / 16;
  uint4* buf; CK(hipMalloc(&buf, bytes)); uint4* sink; CK(hipMalloc(&sink, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, buf, n16);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_write, dim3(256 * 8), dim3(256), 0, 0, buf, n16);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("write stream, 1 GB x 5: %.2f TB/s\n", 5.0 * bytes / (ms * 1e-3) / 1e12);
    hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(256), 0, 0, buf, n16, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) printf("read stream,  1 GB x 5: %.2f TB/s\n", 5.0 * bytes / (ms * 1e-3) / 1e12);
  }
  return 0;
}
