// socmx_philox.h -- the device noise generator shared by the rollout kernels (Philox4x32-10 + Box-Muller) and the
// 16-lane row sum.  Contract: include/socmx.h (socmx_rollout_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace socmx {

// ---- Philox4x32-10 (Salmon et al. SC'11) ---------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one N(0,1) draw for (global row, step, dim) -- contract documented in include/socmx.h
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint32_t grow, uint32_t step,
                                               int dim) {
  uint32_t w[4];
  philox4x32_10(grow, step, (uint32_t)(dim >> 2), (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const int h = (dim >> 1) & 1;
  const float ua = ((float)w[2 * h] + 0.5f) * 2.3283064365386963e-10f;      // 2^-32
  const float ub = ((float)w[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
  const float r = sqrtf(-2.0f * logf(ua));
  float s, c;
  sincospif(2.0f * ub, &s, &c);
  return (dim & 1) ? r * s : r * c;
}

// sum over the 16 lanes of a row group with DPP moves (quad swaps, half-row mirror, row mirror): every lane ends with
// the total, no trip through the LDS crossbar as with ds_bpermute (__shfl_xor)
__device__ __forceinline__ float row16_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

}  // namespace socmx
