// socmx_rollout1p.h -- the per-wave block program of the packed-fma one-row rollout (socmx_rollout1p.hip) and its weight image:
// shared by the kernel, the pack kernel and the C-ABI translation unit (socmx_rollout.hip: image size, routing).
#pragma once
#include "socmx_rollout_common.h"

namespace socmx {

constexpr int kR1pWaves = 8;
constexpr int kR1pBlocks = 22;        // blocks per step of a wave, in the order the wave consumes them
constexpr int kR1pWaveBlocks = 22;    // image stride between waves, in blocks of 1024 floats

// Wave w owns an EIGHTH of every layer's input: r1[32 w ..], r2[16 w ..], r3[8 w ..], o2[16 w ..], o1[32 w ..] (units = positions),
// and multiplies its slice into ALL the units of the layers that read it.  Block = 1024 floats [c (4)][lane (64)][e (4)]:
//   R1P_PK:    W[64 j + lane][16 kg + 4 c + e]                      (unit register j, k16-group kg; v_pk_fma_f32 on (k, k + 1) pairs)
//   R1P_PK8:   up_2 on the wave's EIGHT inputs: pieces 0, 1 = W[lane][8 w + 4 c + e], pieces 2, 3 = W[64 + lane][8 w + 4 (c - 2) + e]
//   R1P_STATE: W[32 w + (lane & 31)][4 c + e] (down_0; res_0: unit lane & 15): DPP fmacs on the state register
//   R1P_FRAG:  up_0's standard fragments (0, 2 w + c), c = 0, 1: DPP fmacs on the activation register of o1[32 w ..]
enum : int { R1P_PK = 0, R1P_STATE = 1, R1P_FRAG = 2, R1P_PK8 = 3 };
struct R1pBlk { int layer, j, kg, form; };
// blocks in CONSUMPTION order (the order decides the stream ring's slots):
//   P0: 0 down_0 | 1, 2 down_1 group 0 (unit registers 0, 1) | 3, 4 down_1 group 1
//   P1: 5, 6 res_1 group 0 (unit registers 0, 1: in the shadow of the partial-sum reads) | 7 down_2 | 8, 9 res_1 group 1 (0, 1)
//   P2: 10, 11 res_2 (0, 1: shadow) | 12 up_2
//   P3: 13, 14 res_1 group 0 (unit registers 2, 3: shadow) | 15, 16 up_1 (0, 1) | 17, 18 up_1 (2, 3) | 19, 20 res_1 group 1 (2, 3)
//   P4: 21 up_0
__host__ __device__ constexpr R1pBlk r1p_block(int w, int b) {
  switch (b) {
    case 0: return {0, 0, 0, R1P_STATE};
    case 1: case 2: return {1, b - 1, 2 * w, R1P_PK};
    case 3: case 4: return {1, b - 3, 2 * w + 1, R1P_PK};
    case 5: case 6: return {4, b - 5, 2 * w, R1P_PK};
    case 7: return {2, 0, w, R1P_PK};
    case 8: case 9: return {4, b - 8, 2 * w + 1, R1P_PK};
    case 10: case 11: return {5, b - 10, w, R1P_PK};
    case 12: return {6, 0, 0, R1P_PK8};
    case 13: case 14: return {4, 2 + (b - 13), 2 * w, R1P_PK};
    case 15: case 16: case 17: case 18: return {7, b - 15, w, R1P_PK};
    case 19: case 20: return {4, 2 + (b - 19), 2 * w + 1, R1P_PK};
    default: return {8, 0, 0, R1P_FRAG};
  }
}
constexpr int kR1pRes0Block = 22;     // wave 7 only (the books): res_0, one more STATE block behind its 22
__host__ __device__ constexpr int r1p_image_floats() { return (kR1pWaves * kR1pWaveBlocks + 1) * 1024; }

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
