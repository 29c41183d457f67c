// Probe (gfx950): v_permlane16_swap_b32 / v_permlane32_swap_b32 for lane-wise sums across the 16-lane rows of a wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_permlane_swap.hip -o tools/_build/probe_permlane_swap && tools/_build/probe_permlane_swap
// What rd_encfuse.hip relies on (rowpair_sum): with a = b = x, after `v_permlane16_swap_b32 a, b` the EVEN rows of a + b hold
// x[row] + x[row + 1].  MEASURED (round 3): that holds; the odd rows of a + b come out as 2 x[row] -- on this part only the second
// operand's even rows (resp. lower half for the 32-lane form) receive data through inline asm with two "+v" operands, the first
// operand's odd rows keep theirs -- so the full four-row all-reduce below is right in row 0 only.  Use the even rows.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float rows4_sum(float x) {
  float a = x, b = x;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = {x0, x0, x2, x2}, b = {x1, x1, x3, x3} (rows of 16 lanes)
  a += b;
  b = a;
  asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = {lo, lo}, b = {hi, hi} (halves of 32 lanes)
  return a + b;
}
__device__ __forceinline__ float rowpair_sum(float x) {
  float a = x, b = x;
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(const float* in, float* out) { out[threadIdx.x] = rows4_sum(in[threadIdx.x]); }
__global__ void k2(const float* in, float* out) { out[threadIdx.x] = rowpair_sum(in[threadIdx.x]); }
int main() {
  float h[64], o[64], *d, *e;
  for (int i = 0; i < 64; ++i) h[i] = (float)(i * i + 1);
  hipMalloc(&d, 256); hipMalloc(&e, 256);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, 256, hipMemcpyDeviceToHost);
  int bad = 0, bad0 = 0;
  for (int l = 0; l < 64; ++l) { const int i = l & 15; const float w = h[i] + h[i + 16] + h[i + 32] + h[i + 48]; if (o[l] != w) { ++bad; if (l < 16) ++bad0; } }
  printf("four-row all-reduce: %d lanes differ (row 0: %d)\n", bad, bad0);
  hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, 256, hipMemcpyDeviceToHost);
  int badp = 0;
  for (int l = 0; l < 64; ++l) if (((l >> 4) & 1) == 0 && o[l] != h[l] + h[l + 16]) { ++badp; printf("lane %d: got %g want %g\n", l, o[l], h[l] + h[l + 16]); }
  printf("row-pair sum in the even rows: %s\n", badp ? "MISMATCH" : "ok");
  return badp != 0;
}
