// socmx_rollout1p.h -- the per-wave block programs of the packed-fma one-row rollout (socmx_rollout1p.hip) and its weight image:
// shared by the kernel, the pack kernel and the C-ABI translation unit (socmx_rollout.hip: image size, routing).
#pragma once
#include "socmx_rollout_common.h"

namespace socmx {

constexpr int kR1pWaves = 8;
constexpr int kR1pChainBlocks = 22;   // blocks per step of a chain wave (waves 0..3)
constexpr int kR1pSkipBlocks = 21;    // ... of a skip wave (waves 4..7; the last one, res_0, is used by wave 4 only)
constexpr int kR1pWaveBlocks = 22;    // image stride between waves, in blocks of 1024 floats

enum : int { R1P_PK = 0, R1P_STATE = 1, R1P_FRAG = 2 };
// block = (layer in SOCMX_L_* order, unit register j = units 64 j .. 64 j + 63, k16-group kg, form)
//   R1P_PK:    [c][lane][e] = W[unit(64 j + lane)][16 kg + 4 c + e]: v_pk_fma_f32 on (k, k + 1) pairs
//   R1P_STATE: [c][lane][e] = W[64 j + lane][4 c + e] (down_0; res_0: unit lane & 15): DPP fmacs on the state register
//   R1P_FRAG:  the standard fragments (0, 4 kg + c) of up_0: DPP fmacs on the activation register of units 64 kg ..
struct R1pBlk { int layer, j, kg, form; };
__host__ __device__ constexpr R1pBlk r1p_block(int wave, int b) {
  const int q = wave & 3;
  if (wave < 4) {
    if (b == 0) return {0, q, 0, R1P_STATE};                                        // down_0: units 64 q ..
    if (b < 9) return {1, (b - 1) & 1, 4 * q + ((b - 1) >> 1), R1P_PK};             // down_1: k16-group outer, unit register inner
    if (b < 11) return {2, 0, 2 * q + (b - 9), R1P_PK};                             // down_2
    if (b < 13) return {6, b - 11, q, R1P_PK};                                      // up_2
    if (b < 21) return {7, (b - 13) & 3, 2 * q + ((b - 13) >> 2), R1P_PK};          // up_1
    return {8, 0, q, R1P_FRAG};                                                      // up_0
  }
  if (b < 4) return {4, b, 4 * q, R1P_PK};                                           // res_1, k16-group 0 of the quarter
  if (b < 8) return {5, (b - 4) & 1, 2 * q + ((b - 4) >> 1), R1P_PK};               // res_2
  if (b < 12) return {4, b - 8, 4 * q + 1, R1P_PK};                                  // res_1, group 1
  if (b < 20) return {4, (b - 12) & 3, 4 * q + 2 + ((b - 12) >> 2), R1P_PK};        // res_1, groups 2, 3
  return {3, 0, 0, R1P_STATE};                                                       // res_0
}
__host__ __device__ constexpr int r1p_image_floats() { return kR1pWaves * kR1pWaveBlocks * 1024; }

// the default widths at d <= 15 (the shapes both one-row kernels are built for)
__host__ __device__ constexpr bool r1_supported_default() {
  constexpr UnetDesc u = DefaultNet::desc();
  return u.in0p == 16 && u.outp == 16 && u.hp[0] == 256 && u.hp[1] == 128 && u.hp[2] == 64;
}
// does an image of architecture (d, hdims) carry the second (packed-fma) part behind the fragment-ordered one?
inline bool r1p_image_wanted(int d, const int h[3]) {
  return r1_supported_default() && d <= 15 && pad16(h[0]) == 256 && pad16(h[1]) == 128 && pad16(h[2]) == 64;
}

__attribute__((visibility("hidden"))) bool rollout1p_available();
__attribute__((visibility("hidden"))) int rollout1p_pack(const socmx_unet* net, float* pk, void* stream);
__attribute__((visibility("hidden"))) int rollout1p_launch(const RolloutArgs& a, bool stopping, void* stream);

}  // namespace socmx
