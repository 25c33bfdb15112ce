// socmx_rollout_common.h -- what the rollout translation units share: the kernel argument block, the Philox pair helpers and
// the constexpr network instantiations (socmx_rollout.hip: 16-row and 4-row tiles; socmx_rollout1.hip: one row per workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/socmx.h"
#include "socmx_unet.h"
#include "socmx_philox.h"

namespace socmx {

struct RolloutArgs {
  UnetDesc u;
  TileLayout t;
  UnetProgram prog;
  int kind, d, B, K;
  float lmbd;
  uint64_t seed, offset;
  const uint64_t* key_dev;   // device {seed, offset} (socmx_rollout_ex_f32: a replayed hipGraph draws fresh noise) or NULL
  int advance_key;           // SOCMX_ROLLOUT_ADVANCES_KEY: key_dev has a third word (ticket) and the launch adds 1 to the offset
  int64_t row0;
  const float* packed;
  const float *sigma, *A, *P, *Q, *omega, *kappa, *nu;
  const float *x0, *ts, *noise_in;
  float *states, *noises, *controls, *stop_ind, *frac, *lpd, *lps, *ltw;
  float* nabla_v;      // optional (K+1, B, d): the network output at every grid point incl. the terminal one (method.py:272-278)
  // optional (socmx_rollout_ex_f32: act_workspace / act_records; the one-row kernel at d <= 15 only, (K+1) B a multiple of 16): the activation
  // slabs T_X .. T_O1 of the control-network backward's workspace and the 128-byte sign record of every trajectory row (socmx_unet.h)
  float* act_ws;
  uint32_t* act_rec;
  int act_tile_rows;   // 16 x the backward's 16-row tiles = (K+1) B
  int sigma_identity;  // problem->flags & SOCMX_SIGMA_IDENTITY
  int lds_mats;  // float offset (in LDS) of the sigma / A / P copies and the small per-step vectors
  long long* prof;  // diagnostics only (PROF variant): [blocks][64] accumulated s_memtime cycles per phase
  int prof_wave;    // which wave's view is recorded (SOCMX_PROF_WAVE, default 0)
};

__host__ __device__ constexpr int socmx_sde_stride(int d) { return ((d + 15) & ~15) + 1; }

#if defined(__HIPCC__)
// The Philox key of a launch: by value, or read from device memory.  With SOCMX_ROLLOUT_ADVANCES_KEY the launch advances the key
// itself: rollout_key() makes sure the values have ARRIVED in this thread's registers (a barrier alone does not wait for loads in
// flight), and rollout_key_advance() -- called by all threads BEHIND a workgroup barrier that follows every thread's
// rollout_key() -- lets thread 0 take the workgroup's ticket; the workgroup that draws the last one stores offset + 1: every
// workgroup of the launch has read the old value by then.
__device__ __forceinline__ void rollout_key(const RolloutArgs& a, uint64_t& seed, uint64_t& offset) {
  seed = a.key_dev ? a.key_dev[0] : a.seed;
  offset = a.key_dev ? a.key_dev[1] : a.offset;
  if (a.key_dev && a.advance_key) asm volatile("" : : "v"((uint32_t)offset), "v"((uint32_t)(offset >> 32)), "v"((uint32_t)seed) : "memory");
}
__device__ __forceinline__ void rollout_key_advance(const RolloutArgs& a, uint64_t offset) {
  if (a.key_dev && a.advance_key && threadIdx.x == 0) {
    unsigned long long* k = reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(a.key_dev));
    const unsigned long long t = atomicAdd(k + 2, 1ull);
    if (t == (unsigned long long)gridDim.x * gridDim.y * gridDim.z - 1ull) {
      k[1] = offset + 1ull;
      k[2] = 0ull;
    }
  }
}
#endif


// the two halves of philox_normal2, for callers that spread them over two phases of a step
__device__ __forceinline__ void philox_pair_words(uint64_t seed, uint64_t offset, uint32_t grow, uint32_t step,
                                                  int block, int h, uint32_t& wa, uint32_t& wb) {
  uint32_t w[4];
  philox4x32_10(grow, step, (uint32_t)block, (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  wa = h ? w[2] : w[0];
  wb = h ? w[3] : w[1];
}
__device__ __forceinline__ void box_muller_pair(uint32_t wa, uint32_t wb, float& z0, float& z1) {
  const float ua = ((float)wa + 0.5f) * 2.3283064365386963e-10f;
  const float ub = ((float)wb + 0.5f) * 2.3283064365386963e-10f;
  const float r = sqrtf(-2.0f * logf(ua));
  float sn, cs;
  sincospif(2.0f * ub, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

struct DynamicNet { static constexpr int outp = 0; };  // descriptors come from the kernel arguments (any architecture)
// (SOCMX_H*P: the reference's default arch.hdims = [256,128,64] unless this is a variant build, see socmx_unet.h)
typedef StaticNet<16, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 16> DefaultNet;  // d <= 15
typedef StaticNet<80, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 64> Wide64Net;   // the same hidden widths at d = 64 (BASELINE configs[4])
typedef StaticNet<32, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 32> Wide32Net;   // ... and at 17 <= d <= 31 (soc.yaml's default d = 20)


// socmx_rollout1.hip: the one-row kernel (B <= 256, sigma = I, d <= 15, default widths); returns false when this build has no
// such kernel for the architecture (variant libraries)
// (hidden: libsocmx.so is loaded RTLD_GLOBAL and an architecture-variant library beside it -- with default visibility the
//  variant's calls bound to the DEFAULT library's definitions, i.e. to kernels compiled for other widths)
__attribute__((visibility("hidden"))) bool rollout1_available();
__attribute__((visibility("hidden"))) int rollout1_launch(const RolloutArgs& a, bool stopping, void* stream);
__attribute__((visibility("hidden"))) bool rollout1_wide_available();     // ... its 17 <= d <= 31 form (sigma = I)
__attribute__((visibility("hidden"))) int rollout1_wide_launch(const RolloutArgs& a, bool stopping, void* stream);
// socmx_rollout32.hip: two 16-row tiles per workgroup (evaluation bursts, more tiles than CUs: d <= 15, or 17 <= d <= 31 with sigma = I)
__attribute__((visibility("hidden"))) bool rollout32_available(int in0p);
__attribute__((visibility("hidden"))) int rollout32_launch(const RolloutArgs& a, bool stopping, void* stream);

}  // namespace socmx
