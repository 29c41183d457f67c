// rd_tiles_export.hip -- fp32 [M, cols] row-major -> split-bf16 ROW TILES for the streaming weight-gradient kernel (rd_tile_wgrad.hip):
//   tiles[chunk s of 32 rows][column tile j of 16][hi, lo][64 lanes][8],  lane (i, G) = (lane & 15, lane >> 4) holding
//   X[32 s + 8 G + r][16 j + i], r = 0..7   (rows >= M and columns >= cols: zeros)
// Where a tensor's consumer is a row-block kernel (rd_rowgemm.hip, rd_encfuse.hip, rd_attnfuse.hip) the tiles are a by-product of
// that kernel's LDS planes.  This stand-alone pass serves the widths those kernels do not take -- SYN256's encoder (D = 1040,
// nhid = 2080) runs on the panel / tiled GEMMs, whose split-K weight gradients convert both fp32 operands once per 64 x 64 output
// tile: 52 % of that configuration's step.  One streaming conversion per operand (4 B read, 4 B written per element) and the same
// tile stream as the fused path replace them.
// A 256-thread workgroup takes one 32-row chunk x 64 columns: coalesced 16-byte row loads -> LDS -> each wave assembles one column
// tile (8 strided LDS reads per lane) and stores two contiguous kilobytes.
#include "rd_common.h"

namespace rd {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int EX_MAXJOBS = 8;
struct ExJob { const float* x; long ld; int M, cols; __bf16* tiles; int nct; };       // nct = ceil(cols / 16)
struct ExArgs { ExJob j[EX_MAXJOBS]; int n; };

__global__ __launch_bounds__(256) void k_rows_to_tiles(ExArgs a) {
  __shared__ float sx[32][68];
  const ExJob J = a.j[blockIdx.z];
  const int s = blockIdx.y, c0 = 64 * blockIdx.x;
  if (c0 >= 16 * J.nct || 32 * s >= ((J.M + 31) & ~31)) return;          // (grid is sized for the largest job)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool vec = (J.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(J.x) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int e = tid + 256 * it, r = e >> 4, q = e & 15;                // row of the chunk, 16-byte column group
    const long row = 32L * s + r;
    const int c = c0 + 4 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < J.M) {
      const float* p = J.x + row * J.ld + c;
      if (vec && c + 3 < J.cols) v = *reinterpret_cast<const float4*>(p);
      else {
        if (c < J.cols) v.x = p[0];
        if (c + 1 < J.cols) v.y = p[1];
        if (c + 2 < J.cols) v.z = p[2];
        if (c + 3 < J.cols) v.w = p[3];
      }
    }
    *reinterpret_cast<float4*>(&sx[r][4 * q]) = v;
  }
  __syncthreads();
  const int j = (c0 >> 4) + wave;
  if (j >= J.nct) return;
  const int i = lane & 15, G = lane >> 4;
  bf16x8 h, l;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float x = sx[8 * G + r][16 * wave + i];
    h[r] = (__bf16)x; l[r] = (__bf16)(x - (float)h[r]);
  }
  __bf16* dst = J.tiles + (((size_t)s * J.nct + j) * 2) * 512 + lane * 8;
  *reinterpret_cast<bf16x8*>(dst) = h;
  *reinterpret_cast<bf16x8*>(dst + 512) = l;
}

}  // namespace

// up to 8 tensors in one launch: x_i [M, cols_i] (row stride ld_i floats) -> tiles_i (tile_elems(M, cols_i) bf16 elements)
int launch_rows_to_tiles(long M, int n, const float* const* x, const long* ld, const int* cols, void* const* tiles, hipStream_t st) {
  if (n < 1 || n > EX_MAXJOBS) return fail(RD_EINVAL, "rows_to_tiles: 1..%d tensors", EX_MAXJOBS);
  ExArgs a{};
  a.n = n;
  int maxc = 0;
  for (int i = 0; i < n; ++i) {
    a.j[i] = ExJob{x[i], ld[i], (int)M, cols[i], (__bf16*)tiles[i], cdiv(cols[i], 16)};
    if (cols[i] > maxc) maxc = cols[i];
  }
  hipLaunchKernelGGL(k_rows_to_tiles, dim3(cdiv(cdiv(maxc, 16) * 16, 64), cdiv((int)M, 32), n), dim3(256), 0, st, a);
  return check_launch("k_rows_to_tiles");
}

}  // namespace rd
