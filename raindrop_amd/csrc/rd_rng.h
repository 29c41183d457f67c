// rd_rng.h -- counter-based RNG for dropout masks (Philox4x32-10, Salmon et al. SC'11).
// Stateless: the keep/drop decision for element `idx` of dropout site `site` under `seed` is a
// pure function, so the backward pass regenerates (or re-reads) exactly the forward mask and a
// captured hipGraph only needs its seed cell bumped between replays.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rd {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0; k.y += W1;
  }
  return c;
}

// four uniforms in [0,1) for the element quad `quad` (elements 4*quad .. 4*quad+3) of `site`
__device__ __forceinline__ float4 uniform4(uint64_t seed, uint32_t site, uint64_t quad) {
  const uint4 r = philox4x32_10(make_uint4((uint32_t)quad, (uint32_t)(quad >> 32), site, 0u),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float s = 1.0f / 16777216.0f;
  return make_float4((r.x >> 8) * s, (r.y >> 8) * s, (r.z >> 8) * s, (r.w >> 8) * s);
}

// single element: keep-scale (0 or 1/(1-p)) for element idx
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t site, uint64_t idx, float p,
                                               float inv_keep) {
  const float4 u = uniform4(seed, site, idx >> 2);
  const int j = (int)(idx & 3);
  const float v = j == 0 ? u.x : (j == 1 ? u.y : (j == 2 ? u.z : u.w));
  return v >= p ? inv_keep : 0.f;
}

// effective seed: the by-value seed plus the content of an optional device cell.  A captured hipGraph
// freezes kernel arguments, so per-replay fresh masks come from bumping the cell (rd_seed_cell_advance).
__device__ __forceinline__ uint64_t eff_seed(uint64_t seed, const uint64_t* cell) {
  return cell ? seed + *cell : seed;
}

enum DropSite : uint32_t { SITE_OBS_EMBED = 1, SITE_ATTN_PROB = 16, SITE_ATTN_OUT = 32, SITE_FFN_HID = 48,
                           SITE_FFN_OUT = 64 };   // + layer index for the encoder sites

}  // namespace rd
