// socmx_row1.h -- instruction helpers of the ONE-ROW rollout kernels (socmx_rollout1.hip: v_fmac_f32_dpp matrix-vector stages;
// socmx_rollout1p.hip: the packed-fma form): DPP multiply-adds on 16-lane rows, cross-row sums, raw buffer loads of 1-KiB
// weight fragments.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "socmx_unet.h"

namespace socmx {

// ---- instruction helpers -----------------------------------------------------------------------------------------------
// 16 fmacs of one block: weight register j <-> broadcast position j; two accumulators alternate (no fmac reads the result
// of the one before it).  The leading s_nop covers VALU-write -> DPP-read of the activation register (2 wait states).
#define R1FM(J, A, W) "v_fmac_f32_dpp %" #A ", %2, %" #W " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
// half a block (eight fmacs, positions 8 HI .. 8 HI + 7) into two alternating accumulators
template <int HI>
__device__ __forceinline__ void r1_fmac8(float& a0, float& a1, float x, const float* w) {
  if constexpr (HI == 0) {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(1, 1, 4) R1FM(2, 0, 5) R1FM(3, 1, 6) R1FM(4, 0, 7) R1FM(5, 1, 8) R1FM(6, 0, 9)
            R1FM(7, 1, 10)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
  } else {
    asm("s_nop 1\n\t" R1FM(8, 0, 3) R1FM(9, 1, 4) R1FM(10, 0, 5) R1FM(11, 1, 6) R1FM(12, 0, 7) R1FM(13, 1, 8) R1FM(14, 0, 9)
            R1FM(15, 1, 10)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
  }
}
// sixteen fmacs whose four fragments belong to four different unit blocks: fragment f into accumulator f
__device__ __forceinline__ void r1_fmac16_4acc(float& a0, float& a1, float& a2, float& a3, float x, const float* w) {
#define R1FM4(J, A, W) "v_fmac_f32_dpp %" #A ", %4, %" #W " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t" R1FM4(0, 0, 5) R1FM4(4, 1, 9) R1FM4(8, 2, 13) R1FM4(12, 3, 17) R1FM4(1, 0, 6) R1FM4(5, 1, 10) R1FM4(9, 2, 14)
          R1FM4(13, 3, 18) R1FM4(2, 0, 7) R1FM4(6, 1, 11) R1FM4(10, 2, 15) R1FM4(14, 3, 19) R1FM4(3, 0, 8) R1FM4(7, 1, 12)
              R1FM4(11, 2, 16) R1FM4(15, 3, 20)
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]),
        "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
#undef R1FM4
}
// the same half of TWO blocks that share the activation register (two neuron blocks of one chunk), interleaved: block A's
// fmacs into aA, block B's into aB -- one accumulator per block, and still no fmac reads the result of the one before it
template <int HI>
__device__ __forceinline__ void r1_fmac8x2(float& aA, float& aB, float x, const float* wa, const float* wb) {
  if constexpr (HI == 0) {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(0, 1, 11) R1FM(1, 0, 4) R1FM(1, 1, 12) R1FM(2, 0, 5) R1FM(2, 1, 13) R1FM(3, 0, 6)
            R1FM(3, 1, 14) R1FM(4, 0, 7) R1FM(4, 1, 15) R1FM(5, 0, 8) R1FM(5, 1, 16) R1FM(6, 0, 9) R1FM(6, 1, 17) R1FM(7, 0, 10)
                R1FM(7, 1, 18)
        : "+v"(aA), "+v"(aB)
        : "v"(x), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(wa[7]), "v"(wb[0]),
          "v"(wb[1]), "v"(wb[2]), "v"(wb[3]), "v"(wb[4]), "v"(wb[5]), "v"(wb[6]), "v"(wb[7]));
  } else {
    asm("s_nop 1\n\t" R1FM(8, 0, 3) R1FM(8, 1, 11) R1FM(9, 0, 4) R1FM(9, 1, 12) R1FM(10, 0, 5) R1FM(10, 1, 13) R1FM(11, 0, 6)
            R1FM(11, 1, 14) R1FM(12, 0, 7) R1FM(12, 1, 15) R1FM(13, 0, 8) R1FM(13, 1, 16) R1FM(14, 0, 9) R1FM(14, 1, 17)
                R1FM(15, 0, 10) R1FM(15, 1, 18)
        : "+v"(aA), "+v"(aB)
        : "v"(x), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(wa[7]), "v"(wb[0]),
          "v"(wb[1]), "v"(wb[2]), "v"(wb[3]), "v"(wb[4]), "v"(wb[5]), "v"(wb[6]), "v"(wb[7]));
  }
}
// acc += x[position J of the lane's row] * w
template <int J>
__device__ __forceinline__ void r1_fmac_bc(float& acc, float x, float w) {
  asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(J));
}

// down_0 / res_0 on the state: a += sum_J x_J w[J] for J < DMAX, x_J broadcast from position J of the lane's row.  Columns past d
// are zero in the image and positions past d are zero in the state register: no run-time bound.
template <int DMAX>
__device__ __forceinline__ void r1_state_pair(float& aA, float& aB, float x, const float* wa, const float* wb) {
  if constexpr (DMAX == 3) {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(0, 1, 6) R1FM(1, 0, 4) R1FM(1, 1, 7) R1FM(2, 0, 5) R1FM(2, 1, 8)
        : "+v"(aA), "+v"(aB)
        : "v"(x), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wb[0]), "v"(wb[1]), "v"(wb[2]));
  } else {
    static_assert(DMAX == 11, "pairs: 2 DMAX + 3 asm operands");
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(0, 1, 14) R1FM(1, 0, 4) R1FM(1, 1, 15) R1FM(2, 0, 5) R1FM(2, 1, 16) R1FM(3, 0, 6) R1FM(3, 1, 17) R1FM(4, 0, 7) R1FM(4, 1, 18) R1FM(5, 0, 8) R1FM(5, 1, 19) R1FM(6, 0, 9) R1FM(6, 1, 20) R1FM(7, 0, 10) R1FM(7, 1, 21) R1FM(8, 0, 11) R1FM(8, 1, 22) R1FM(9, 0, 12) R1FM(9, 1, 23) R1FM(10, 0, 13) R1FM(10, 1, 24)
        : "+v"(aA), "+v"(aB)
        : "v"(x), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wa[4]), "v"(wa[5]), "v"(wa[6]), "v"(wa[7]), "v"(wa[8]), "v"(wa[9]), "v"(wa[10]), "v"(wb[0]), "v"(wb[1]), "v"(wb[2]), "v"(wb[3]), "v"(wb[4]), "v"(wb[5]), "v"(wb[6]), "v"(wb[7]), "v"(wb[8]), "v"(wb[9]), "v"(wb[10]));
  }
}
// one unit, two alternating accumulators
template <int DMAX>
__device__ __forceinline__ void r1_state_one(float& a0, float& a1, float x, const float* w) {
  if constexpr (DMAX == 3) {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(1, 1, 4) R1FM(2, 0, 5)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]));
  } else if constexpr (DMAX == 11) {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(1, 1, 4) R1FM(2, 0, 5) R1FM(3, 1, 6) R1FM(4, 0, 7) R1FM(5, 1, 8) R1FM(6, 0, 9) R1FM(7, 1, 10) R1FM(8, 0, 11) R1FM(9, 1, 12) R1FM(10, 0, 13)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]));
  } else {
    asm("s_nop 1\n\t" R1FM(0, 0, 3) R1FM(1, 1, 4) R1FM(2, 0, 5) R1FM(3, 1, 6) R1FM(4, 0, 7) R1FM(5, 1, 8) R1FM(6, 0, 9) R1FM(7, 1, 10) R1FM(8, 0, 11) R1FM(9, 1, 12) R1FM(10, 0, 13) R1FM(11, 1, 14) R1FM(12, 0, 15) R1FM(13, 1, 16) R1FM(14, 0, 17)
        : "+v"(a0), "+v"(a1)
        : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]));
  }
}

// sum over the four 16-lane rows of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48): every lane ends with the total
__device__ __forceinline__ float r1_rows_sum(float v) {
  float t;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %1\n\t"
      "v_add_f32 %0, %0, %1"
      : "+v"(v), "=&v"(t));
  return v;
}
// kg_reduce (socmx_unet.h) for VALU-produced accumulators: no MFMA wait states in front
__device__ __forceinline__ float r1_reduce4(float a, float b, float c, float d) {
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_permlane32_swap_b32 %2, %3\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_add_f32 %2, %2, %3\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_add_f32 %0, %0, %2"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return a;
}
// lo = the lower half's values in both halves, hi = the upper half's
__device__ __forceinline__ void r1_halves(float v, float& lo, float& hi) {
  float a = v, b;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1"
      : "+v"(a), "=&v"(b));
  lo = a;
  hi = b;
}
// rows 2, 3 <- the row rotated by eight positions; rows 0, 1 unchanged
__device__ __forceinline__ float r1_ror8_upper(float v) {
  float r = v;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xc bank_mask:0xf" : "+v"(r) : "v"(v));
  return r;
}

// rows 1, 3 <- the row rotated by eight positions; rows 0, 2 unchanged
__device__ __forceinline__ float r1_ror8_odd_rows(float v) {
  float r = v;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xa bank_mask:0xf" : "+v"(r) : "v"(v));
  return r;
}

// One streamed fragment as a raw buffer load: descriptor of the packed image (four SGPRs, once per wave) + the lane's 32-bit
// byte offset + the block's byte offset in an SGPR + the fragment as an immediate -- no 64-bit VGPR address per request.
// Compiler-visible loads on purpose: with asm requests and written-out vmcnt the compiler does not know that a register
// is still in flight, and a copy it inserts where a live range is split (the loop's exit into the terminal evaluation)
// reads the register before the data arrives -- seen as a non-deterministic nabla_V(T, X_K).  Its own wait counts are exact
// in this straight-line ring (vmcnt(6) / vmcnt(4) in front of a block's two halves).
typedef int r1_i32x4 __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ f32x4 r1_gload(__amdgpu_buffer_rsrc_t img, uint32_t lane_off, int block_bytes) {
  // (the fragment's 1 KiB rides in the scalar offset: added to the lane offset it became four loop-invariant VGPRs)
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(img, (int)lane_off, block_bytes + IMM, 0));
}

// the activation layout: lane (g, p) of an activation register of chunk C holds element 64 C + r1_perm(lane); an involution
__device__ __forceinline__ int r1_perm(int x) { return 16 * ((x >> 2) & 3) + 4 * (x >> 4) + (x & 3); }

}  // namespace socmx
