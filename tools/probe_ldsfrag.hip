// probe_ldsfrag.hip -- LDS cost of the MFMA A-fragment read pattern of the fused K1 kernels (lane (r = l & 15, G = l >> 4) reads 16 bytes
// of row r at column chunk 4 kc + G of a bf16 plane), for three plane layouts:
//   0: rows of 528 bytes (264 bf16: the round-1..4 layout, "conflict-free" by the 32-bank rule)
//   1: rows of 512 bytes, 16-byte chunk index XOR (row & 15)  (round 5)
//   2: rows of 512 bytes, plain (worst case)
//   4 / 5: rows of 512 + 32 / 320 + 32 bytes, no swizzle (tools/lds_conflicts.py: a stride of 32 bytes mod 256 is conflict-free)
//   3: rows of 336 bytes (the x planes of rd_attnfuse.hip / rd_encfuse.hip: 32 KCX + 8 bf16 with KCX = 5)
// 16 waves per workgroup, one workgroup per CU, every wave reads 3 row tiles x 2 planes x 8 steps per pass like mma_mid does.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_ldsfrag tools/probe_ldsfrag.hip && tools/_build/probe_ldsfrag
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(long long* out, int passes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int RS = MODE == 0 ? 528 : MODE == 3 ? 336 : MODE == 4 ? 544 : MODE == 5 ? 352 : 512, PL = 48 * RS;
  for (int i = threadIdx.x; i < 4 * PL / 16; i += 1024) reinterpret_cast<v4i*>(sm)[i] = (v4i){i, i, i, i};
  __syncthreads();
  const int lane = threadIdx.x & 63, r = lane & 15, G = lane >> 4;
  int off[8];
#pragma unroll
  for (int kc = 0; kc < 8; ++kc) {
    const int chunk = 4 * kc + G;
    const int c2 = MODE == 1 ? ((chunk & 16) | ((chunk ^ r) & 15)) : chunk;
    off[kc] = r * RS + c2 * 16;
  }
  v4i acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int p = 0; p < passes; ++p) {
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        const v4i a = *reinterpret_cast<const v4i*>(sm + rt * 16 * RS + off[kc]);
        const v4i b = *reinterpret_cast<const v4i*>(sm + PL + rt * 16 * RS + off[kc]);
        acc ^= a; acc ^= b;
      }
    asm volatile("" : "+v"(acc));
  }
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  if (acc[0] == 0x12345678 && acc[1] == 7) out[0] = acc[2] + acc[3];
}
template <int MODE> void run(const char* nm, long long* d, int passes) {
  const int lds = 4 * 48 * 544;   // mode 3 (336-byte rows, 5 steps used by the kernel; 8 read here: rows overlap, harmless) fits too
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  long long h[256 * 16];
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), lds, 0, d, passes);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0, sum = 0;
  for (int i = 0; i < 256 * 16; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
  // per pass and CU: 16 waves x 48 ds_read_b128 = 768 KB
  printf("%-44s cycles per pass (48 x ds_read_b128 per wave, 16 waves): mean %.0f  max %.0f  -> %.0f B/clk/CU\n", nm, sum / (256 * 16) / passes,
         mx / passes, 16.0 * 48 * 1024 / (mx / passes));
}
int main() {
  long long* d; hipMalloc(&d, 256 * 16 * 8);
  run<0>("rows of 528 B (264 bf16)", d, 200);
  run<1>("rows of 512 B, chunk ^ (row & 15)", d, 200);
  run<2>("rows of 512 B, plain", d, 200);
  run<3>("rows of 336 B (168 bf16: rd_attnfuse x planes)", d, 200);
  run<4>("rows of 544 B (columns + 32 B, no swizzle)", d, 200);
  run<5>("rows of 352 B (160 columns + 32 B, no swizzle)", d, 200);
  return 0;
}
