// probe_tr16.hip -- semantics of ds_read_b64_tr_b16 (LDS transpose read) on gfx950, by experiment.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_tr16 tools/probe_tr16.hip
// LDS holds shorts with value == index; every lane passes its own address; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int addr;   // in shorts
  if (mode == 0) addr = l * 4;                                                  // lane-linear 8-byte pieces
  else if (mode == 1) addr = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 256;   // per 16-lane group: 4 rows (stride 64) x 16 cols
  else addr = (l & 15) * 64 + (l >> 4) * 4;                                     // lane -> row (l&15), 4 cols at (l>>4)*4
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 1) ? "\n" : "   |");
  }
  return 0;
}
