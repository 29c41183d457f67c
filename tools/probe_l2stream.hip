// probe_l2stream.hip -- how fast can a CU pull an L2-resident weight plane set (the K1 weight stream)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_l2stream tools/probe_l2stream.hip
// Variants: V0 = 16 rows x 64 B per wave-load (the round-1 plane layout), V1 = contiguous 1 KB per wave-load,
// V2 = contiguous, LDS-DMA (global_load_lds_dwordx4) into an LDS ring.  Each workgroup (512 threads) reads
// `bytes` per pass, `npass` passes cycling over 4 buffers; grid sizes 256 / 128 / 64 / 32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NTHR = 512;
constexpr int NLOAD = 32;                     // 16-byte loads per lane per pass: 512 * 32 * 16 = 262144 B

template <int V>
__global__ __launch_bounds__(NTHR) void k_stream(const uint4* __restrict__ buf, int npass, size_t pass_stride16,
                                                 uint4* __restrict__ out, unsigned long long* __restrict__ cyc, int skew) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint4 acc = make_uint4(0, 0, 0, 0);
  if (skew) { for (int i = 0; i < (int)(blockIdx.x % 8) * skew; ++i) __builtin_amdgcn_s_sleep(8); }
  const unsigned long long t0 = clock64();
  for (int p = 0; p < npass; ++p) {
    const uint4* base = buf + (size_t)(p & 3) * pass_stride16;
    uint4 v[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      size_t off16;
      if (V == 0) {
        // plane [rows][256 bf16 = 512 B = 32 x 16 B]: tile j = wave + 8*(i>>4), plane part = (i>>3)&1, kc = i&7
        const int j = wave + 8 * (i >> 4), part = (i >> 3) & 1, kc = i & 7;
        off16 = (size_t)part * (256 * 32) + (size_t)(16 * j + (lane & 15)) * 32 + kc * 4 + (lane >> 4);
      } else {
        off16 = (size_t)(wave * NLOAD + i) * 64 + lane;
      }
      v[i] = base[off16];
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
  }
  const unsigned long long t1 = clock64();
  if (acc.x == 0x12345678u && acc.y == 77u) out[blockIdx.x * NTHR + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS-DMA variant: 8 waves, each wave streams its 32 KB share of the pass through a 2 x 8 KB LDS ring per wave
// (128 KB total), 8 loads in flight; a consumer-less probe (data is not read back: pure transport rate)
__global__ __launch_bounds__(NTHR) void k_stream_lds(const uint4* __restrict__ buf, int npass, size_t pass_stride16,
                                                     uint4* __restrict__ out, unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* ring = smem + (size_t)wave * 16384;
  const unsigned long long t0 = clock64();
  for (int p = 0; p < npass; ++p) {
    const uint4* base = buf + (size_t)(p & 3) * pass_stride16;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const uint4* g = base + (size_t)(wave * NLOAD + i) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                       (void __attribute__((address_space(3)))*)(ring + (i & 15) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  __syncthreads();
  uint4 r = *reinterpret_cast<uint4*>(smem + tid * 16);
  if (r.x == 0x12345678u && r.y == 77u) out[blockIdx.x * NTHR + tid] = r;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// HBM streaming probe: every workgroup reads its OWN `bytes_per_wg` (no sharing) -- what a sample's
// activation gather can expect while the whole chip does the same.
__global__ __launch_bounds__(NTHR) void k_private(const uint4* __restrict__ buf, int n16_per_wg, uint4* __restrict__ out,
                                                  unsigned long long* __restrict__ cyc) {
  const int tid = threadIdx.x;
  const uint4* base = buf + (size_t)blockIdx.x * n16_per_wg;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const unsigned long long t0 = clock64();
  for (int i = tid; i < n16_per_wg; i += NTHR * 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = base[min(i + u * NTHR, n16_per_wg - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
  }
  const unsigned long long t1 = clock64();
  if (acc.x == 0x12345678u && acc.y == 77u) out[blockIdx.x * NTHR + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t pass_bytes = (size_t)NTHR * NLOAD * 16;            // 262144
  const size_t pass_stride16 = pass_bytes / 16;
  uint4* buf; uint4* out; unsigned long long* cyc;
  const size_t big = (size_t)512 << 20;
  CK(hipMalloc(&buf, big));
  CK(hipMemset(buf, 1, big));
  CK(hipMalloc(&out, 256 * NTHR * 16));
  CK(hipMalloc(&cyc, 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned long long> h(256);
  CK(hipFuncSetAttribute((const void*)k_stream_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int npass = 16;
  for (int variant = 0; variant < 4; ++variant) {
    for (int G : {256, 128, 64, 32}) {
      float best = 1e9f; unsigned long long cmin = ~0ull, cmax = 0, csum = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(k_stream<0>, dim3(G), dim3(NTHR), 0, 0, buf, npass, pass_stride16, out, cyc, 0);
        if (variant == 1) hipLaunchKernelGGL(k_stream<1>, dim3(G), dim3(NTHR), 0, 0, buf, npass, pass_stride16, out, cyc, 0);
        if (variant == 2) hipLaunchKernelGGL(k_stream_lds, dim3(G), dim3(NTHR), 131072, 0, buf, npass, pass_stride16, out, cyc);
        if (variant == 3) hipLaunchKernelGGL(k_stream<1>, dim3(G), dim3(NTHR), 0, 0, buf, npass, pass_stride16, out, cyc, 4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy(h.data(), cyc, G * 8, hipMemcpyDeviceToHost));
      for (int i = 0; i < G; ++i) { cmin = h[i] < cmin ? h[i] : cmin; cmax = h[i] > cmax ? h[i] : cmax; csum += h[i]; }
      const double cyc_per_pass = (double)csum / G / npass;
      printf("variant %d G=%3d: kernel %.1f us  (%.2f us/pass)  cycles/pass avg %.0f min %.0f max %.0f  -> %.1f B/clk/CU, aggregate %.2f TB/s\n",
             variant, G, best * 1e3, best * 1e3 / npass, cyc_per_pass, (double)cmin / npass, (double)cmax / npass,
             pass_bytes / cyc_per_pass, (double)G * pass_bytes * npass / (best * 1e-3) / 1e12);
    }
  }
  // private HBM streams: 256 WGs x {64, 128, 256} KB each, buffers far apart and rotated so the MALL does not help
  for (int kb : {64, 128, 256, 1024}) {
    const int n16 = kb * 1024 / 16;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      const uint4* b = buf + ((size_t)rep * 64 << 20) / 16;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_private, dim3(256), dim3(NTHR), 0, 0, b, n16, out, cyc);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    unsigned long long csum = 0; for (int i = 0; i < 256; ++i) csum += h[i];
    printf("private %4d KB/WG: kernel %.1f us, cycles avg %.0f -> %.1f B/clk/CU, aggregate %.2f TB/s (by kernel time)\n", kb, best * 1e3,
           (double)csum / 256, kb * 1024.0 / ((double)csum / 256), 256.0 * kb * 1024 / (best * 1e-3) / 1e12);
  }
  return 0;
}
