// rd_rng.h -- counter-based RNG for dropout masks.
// Stateless: the keep/drop decision for element `idx` of dropout site `site` under `seed` is a
// pure function, so the backward pass regenerates (or re-reads) exactly the forward mask and a
// captured hipGraph only needs its seed cell bumped between replays.
//
// Generator: Threefry-2x32 with 13 rounds (Salmon et al., SC'11: the Crush-resistant round count), one call per
// element quad, 16 random bits per element.  Add / rotate / xor only: ~47 full-rate VALU instructions per call.
// Round 1-2 used Philox4x32-10, whose 40 32-bit integer multiplies per call issue at quarter rate on CDNA (16 cycles
// per wave each): ~800 cycles per call and wave, and the calls sit on the critical path of every latency-bound
// epilogue of the step (LayerNorm / FFN dropout: 3-4 calls per thread between two barriers -- measured with phase
// stamps at 10-15 k cycles per such phase, round 3).  -DRD_RNG_PHILOX builds the old generator (A/B only).
// A keep decision compares a 16-bit uniform k * 2^-16 with p: the keep probability is exact to 2^-16.
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

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

// Threefry-2x32-13 (Random123 threefry2x32_R(13, ..)): counter c, key k -> 64 random bits
__device__ __forceinline__ uint2 threefry2x32_13(uint2 c, uint2 k) {
  const uint32_t ks[3] = {k.x, k.y, 0x1BD11BDAu ^ k.x ^ k.y};
  constexpr int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t x0 = c.x + ks[0], x1 = c.y + ks[1];
#pragma unroll
  for (int r = 0; r < 13; ++r) {
    x0 += x1; x1 = rotl32(x1, R[r & 7]); x1 ^= x0;
    if (((r + 1) & 3) == 0) {
      const int s = (r + 1) >> 2;
      x0 += ks[s % 3]; x1 += ks[(s + 1) % 3] + (uint32_t)s;
    }
  }
  return make_uint2(x0, x1);
}

// four uniforms in [0,1) for the element quad `quad` (elements 4*quad .. 4*quad+3) of `site` (site < 2^16, quad < 2^48)
__device__ __forceinline__ float4 uniform4(uint64_t seed, uint32_t site, uint64_t quad) {
#ifdef RD_RNG_PHILOX
  const uint4 r = philox4x32_10(make_uint4((uint32_t)quad, (uint32_t)(quad >> 32), site, 0u),
                                make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float s = 1.0f / 16777216.0f;
  return make_float4((r.x >> 8) * s, (r.y >> 8) * s, (r.z >> 8) * s, (r.w >> 8) * s);
#else
  const uint2 r = threefry2x32_13(make_uint2((uint32_t)quad, (uint32_t)(quad >> 32) | (site << 16)),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  const float s = 1.0f / 65536.0f;
  return make_float4((float)(r.x & 0xFFFFu) * s, (float)(r.x >> 16) * s, (float)(r.y & 0xFFFFu) * s, (float)(r.y >> 16) * s);
#endif
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

enum DropSite : uint32_t { SITE_OBS_EMBED = 1, SITE_EDGE_COEFF = 2, SITE_ATTN_PROB = 16, SITE_ATTN_OUT = 32, SITE_FFN_HID = 48,
                           SITE_FFN_OUT = 64 };   // + layer index for the encoder sites

}  // namespace rd
