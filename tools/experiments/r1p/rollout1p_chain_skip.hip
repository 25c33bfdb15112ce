// socmx_rollout1p.hip -- the fused Euler-Maruyama rollout, ONE ROW PER WORKGROUP, on PACKED fp32 multiply-adds (gfx950).
//
// Replaces reference SOC_matching/utils.py:17-128 (stochastic_trajectories), method.py:58-80 (control) and
// models.py:233-242 (FullyConnectedUNet.forward) for training-size batches (B <= 256 rows) at the default hidden widths,
// sigma = I, d <= 15 -- BASELINE configs[1] / [2] and the README's molecular_dynamics run.  Round 5's successor of
// socmx_rollout1.hip's v_fmac_f32_dpp form for these shapes.
//
// Why another form.  Measured on this chip (tools/ubench/valu_banks.hip, profiles/r5/valu_banks.txt): v_fmac_f32_dpp issues at
// 4.5 cycles per SIMD (64 MACs) whatever the registers, v_pk_fma_f32 at 4.4 (128 MACs) with two waves on the SIMD and 5.1 with
// one -- the DPP operand costs a second pass.  So the matrix-VECTOR products run as
//     acc[lane = unit].{lo, hi} += W[unit][k, k + 1] * x[k, k + 1]                       (v_pk_fma_f32, 128 MACs)
// with the activation pair REPLICATED in every lane: a wave that owns a slice of a layer's INPUT writes its slice to LDS and
// reads it back as broadcast ds_read_b128 (every lane the same address: four values per read), multiplies it into ALL the
// units of the layers that consume that input (split-K over the waves), and leaves per-wave partial sums in LDS for whoever
// owns that unit as an input of the next layer.  One barrier per layer of the chain -- five per step.
//
// Roles.  Waves 0..3 ("chain", one per SIMD, q = input quarter) run the sequential chain
//     [sum nabla_V, Euler-Maruyama, down_0 of their quarter] -> down_1 | down_2 | up_2 | up_1 | up_0
// and nothing else; waves 4..7 ("skip", the second wave of each SIMD) run what is OFF that chain -- the skip GEMMs res_1
// (39 % of the network's MACs) and res_2, whose inputs are a barrier old when they start and whose outputs are needed two to
// three barriers later -- plus the noise (Philox words, Box-Muller), the step's scalars, res_0, the running costs and every
// global store.  The chain waves stall on LDS round trips and barriers most of the time; the skip waves' packed fmas fill
// the SIMDs' issue slots meanwhile (tools/ubench/valu_banks: a lone wave reaches 5.1 cycles per v_pk_fma_f32, two 4.4).
//
// Weights: a second image behind the fragment-ordered one (socmx_unet_pack_f32 writes both), wave-major: block b of wave w =
// 1024 floats [c (4)][lane (64)][e (4)] = W[unit(64 j + lane)][16 kg + 4 c + e] of the wave's (layer, j, kg) -- one 16-byte
// load per lane is two (k, k + 1) pairs.  Per wave a compile-time plan says where each block lives: R registers for the
// whole launch, L copied to LDS once (read one block ahead), S streamed from L2 every step through a two-block ring.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "socmx_rollout_common.h"
#include "socmx_launch.h"
#include "socmx_row1.h"
#include "socmx_rollout1p.h"

namespace socmx {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- where a block's weights come from: 'R' registers, 'L' LDS, 'S' L2 stream.  role 0 = chain waves, 1 = skip wave 4 (the
//      books: it also holds res_0 and the running costs, so fewer register blocks), 2 = skip waves 5..7 ----
//   chain: 0 down_0 | 1-8 down_1 | 9-10 down_2 | 11-12 up_2 | 13-20 up_1 | 21 up_0        (r1p_block, socmx_rollout1p.h)
//   skip:  0-3 res_1 (kg 0) | 4-7 res_2 | 8-11 res_1 (kg 1) | 12-19 res_1 (kg 2, 3) | 20 res_0 (wave 4 only)
__host__ __device__ constexpr int r1p_blocks(int role) { return role == 0 ? kR1pChainBlocks : role == 1 ? kR1pSkipBlocks : kR1pSkipBlocks - 1; }
__host__ __device__ constexpr char r1p_src(int role, int b) {
#ifdef SOCMX_R1P_PLAN0
  constexpr char plan0[kR1pChainBlocks + 1] = SOCMX_R1P_PLAN0;
#else
  constexpr char plan0[kR1pChainBlocks + 1] = "RLRSRLSRLRSLRSRLSRLRSR";
#endif
#ifdef SOCMX_R1P_PLAN1
  constexpr char plan1[kR1pSkipBlocks + 1] = SOCMX_R1P_PLAN1;
#else
  constexpr char plan1[kR1pSkipBlocks + 1] = "SRLSRLSRSRLSRSLRSLRSR";
#endif
#ifdef SOCMX_R1P_PLAN2
  constexpr char plan2[kR1pSkipBlocks] = SOCMX_R1P_PLAN2;
#else
  constexpr char plan2[kR1pSkipBlocks] = "SRSRSRSRSRLSRSRSRSRS";
#endif
  return role == 0 ? plan0[b] : role == 1 ? plan1[b] : plan2[b];
}
__host__ __device__ constexpr int r1p_count(int role, char s, int upto = -1) {
  int n = 0;
  const int e = upto < 0 ? r1p_blocks(role) : upto;
  for (int b = 0; b < e; ++b) n += r1p_src(role, b) == s;
  return n;
}
__host__ __device__ constexpr int r1p_nth(int role, char s, int i) {
  int n = 0;
  for (int b = 0; b < r1p_blocks(role); ++b)
    if (r1p_src(role, b) == s) {
      if (n == i) return b;
      ++n;
    }
  return -1;
}
static_assert(r1p_src(0, 0) == 'R' && r1p_src(0, 21) == 'R' && r1p_src(1, 20) == 'R', "the DPP-form blocks (down_0, up_0, res_0) are register blocks");
static_assert(r1p_count(0, 'S') % 2 == 0 && r1p_count(1, 'S') % 2 == 0 && r1p_count(2, 'S') % 2 == 0 && r1p_count(0, 'S') >= 2 &&
              r1p_count(1, 'S') >= 2 && r1p_count(2, 'S') >= 2, "static ring slots across steps");
// first LDS block of a wave's LDS-resident blocks
__host__ __device__ constexpr int r1p_lds_first(int wave) {
  return wave < 4 ? wave * r1p_count(0, 'L') : wave == 4 ? 4 * r1p_count(0, 'L') : 4 * r1p_count(0, 'L') + r1p_count(1, 'L') + (wave - 5) * r1p_count(2, 'L');
}

// LDS map (floats)
struct R1pLds {
  static constexpr int xr1 = 0;          // (256) r1, r2, r3, o2 by POSITION: what the owner of a slice wrote, read back as broadcasts
  static constexpr int xr2 = 256;        // (128)
  static constexpr int xr3 = 384;        // (64)
  static constexpr int xo2 = 448;        // (128)
  static constexpr int p1 = 576;         // (128, 4) per-wave partial sums [position][chain / skip wave q]: one ds_read_b128 per position
  static constexpr int p2 = 1088;        // (64, 4)
  static constexpr int pr2 = 1344;       // (128, 4)   res_2
  static constexpr int p3 = 1856;        // (128, 4)
  static constexpr int p4 = 2368;        // (256, 4)
  static constexpr int pr1 = 3392;       // (256, 4)   res_1
  static constexpr int p5 = 4416;        // (16, 4)    up_0
  static constexpr int res0 = 4480;      // (16)  res_0 [t_k, x_k] + b of the evaluation under way (skip wave 0)
  static constexpr int bk = 4496;        // (48)  the finished step's nabla_V, new state, fractional step, stop flag: chain 0 -> skip 0
  static constexpr int nz = 4544;        // (3, 16) noise of steps k - 1, k, k + 1
  static constexpr int wz = 4592;        // (2, 8, 2) Philox words of steps k + 1, k + 2
  static constexpr int sc = 4624;        // (3, 4) per-step scalars
  static constexpr int amat = 4640;      // (16, 16) A TRANSPOSED (amat[j * 16 + i] = A[i][j]: lanes along i), P row-major
  static constexpr int pmat = 4896;
  static constexpr int bias = 5152;      // the nine layers' padded biases (image order)
  static constexpr int weights = 6400;   // LDS-resident blocks, 1024 floats each: chain waves' first
};
static_assert(R1pLds::bias + 1248 <= R1pLds::weights, "bias copy");
__host__ __device__ constexpr int r1p_lds_blocks() { return r1p_lds_first(7) + r1p_count(2, 'L'); }
static_assert((R1pLds::weights + r1p_lds_blocks() * 1024) * 4 <= 160 * 1024, "LDS-resident weight blocks do not fit");

// Developer instrumentation (make PROF=1): per-wave s_memtime deltas between the marks of a step, summed over the launch, written
// by workgroup 0 to a.prof[wave * 16 + slot] (socmx_rollout_phase_cycles_f32; tools/r1p_phases.py).  Even slots: work of phase
// P0 .. P4, odd slots: the wait at the barrier behind it.
#ifdef SOCMX_R1_PROF
#define R1P_TICK(slot)                                   \
  {                                                      \
    const long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_last;                  \
    prof_last = now_;                                    \
  }
#define R1P_PROF_DECL long long prof_acc[16] = {0}, prof_last = 0;
#define R1P_PROF_START prof_last = __builtin_readcyclecounter();
#define R1P_PROF_END(wave)                                   \
  if (a.prof && blockIdx.x == 0 && lane == 0)                \
    for (int sl = 0; sl < 16; ++sl) a.prof[(wave) * 16 + sl] = prof_acc[sl];
#else
#define R1P_TICK(slot)
#define R1P_PROF_DECL
#define R1P_PROF_START
#define R1P_PROF_END(wave)
#endif

template <int N, typename F>
__device__ __forceinline__ void r1p_static_for(F&& f) {
  if constexpr (N > 0) {
    r1p_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// eight packed fmas: one half (pieces c = 2 H, 2 H + 1 = eight inputs) of TWO blocks against eight activations x -- wave-uniform
// values (SGPR pairs: a v_pk_fma_f32 takes one scalar pair at no cost, tools/ubench/valu_banks)
template <int H>
__device__ __forceinline__ void r1p_pk_half2(f32x2& aA, f32x2& aB, const float (&x)[16], const f32x4 (&wa)[2], const f32x4 (&wb)[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const f32x2 x0 = {x[8 * H + 4 * c], x[8 * H + 4 * c + 1]}, x1 = {x[8 * H + 4 * c + 2], x[8 * H + 4 * c + 3]};
    aA = __builtin_elementwise_fma(f32x2{wa[c][0], wa[c][1]}, x0, aA);
    aB = __builtin_elementwise_fma(f32x2{wb[c][0], wb[c][1]}, x0, aB);
    aA = __builtin_elementwise_fma(f32x2{wa[c][2], wa[c][3]}, x1, aA);
    aB = __builtin_elementwise_fma(f32x2{wb[c][2], wb[c][3]}, x1, aB);
  }
}
// four: half of ONE block, two accumulators (no fma reads the result of the one before it)
template <int H>
__device__ __forceinline__ void r1p_pk_half1(f32x2& a0, f32x2& a1, const float (&x)[16], const f32x4 (&w)[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    a0 = __builtin_elementwise_fma(f32x2{w[c][0], w[c][1]}, f32x2{x[8 * H + 4 * c], x[8 * H + 4 * c + 1]}, a0);
    a1 = __builtin_elementwise_fma(f32x2{w[c][2], w[c][3]}, f32x2{x[8 * H + 4 * c + 2], x[8 * H + 4 * c + 3]}, a1);
  }
}
__device__ __forceinline__ float r1p_sum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// sixteen activations out of the lanes FIRST .. FIRST + 15 of v into wave-uniform values (v_readlane_b32: one VALU
// instruction each, no LDS round trip, no vector register -- the replicated form the packed fmas read)
template <int FIRST>
__device__ __forceinline__ void r1p_gather(float (&x)[16], float v) {
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), FIRST + k));
}

// the LDS block that follows block `after` in a role's program (wraps into the next evaluation), -1 if the role has none
__host__ __device__ constexpr int r1p_next_lds(int role, int after) {
  for (int b = after + 1; b < r1p_blocks(role); ++b)
    if (r1p_src(role, b) == 'L') return b;
  for (int b = 0; b <= after; ++b)
    if (r1p_src(role, b) == 'L') return b;
  return -1;
}

// ---- what every wave of the workgroup has: its blocks (resident / LDS / stream), one LDS landing block, a two-block stream ring ----
template <int ROLE>
struct R1pWeights {
  static constexpr int NRES = r1p_count(ROLE, 'R'), NLDS = r1p_count(ROLE, 'L'), NSTR = r1p_count(ROLE, 'S');
  f32x4 wres[NRES][4];
  f32x4 ring[2][4];
  f32x4 lq[4];
  const float* LW;
  __amdgpu_buffer_rsrc_t img;
  uint32_t loff;
  int wave_bytes;     // byte offset of the wave's first block inside the image
  int lane;

  __device__ __forceinline__ void init(const RolloutArgs& a, float* lds, int wave, int lane_) {
    constexpr UnetDesc u = DefaultNet::desc();
    const float* pk = a.packed + u.total_floats;
    lane = lane_;
    loff = lane * 16;
    wave_bytes = wave * kR1pWaveBlocks * 4096;
#pragma unroll
    for (int r = 0; r < NRES; ++r) {
      const f32x4* src = reinterpret_cast<const f32x4*>(pk + (wave * kR1pWaveBlocks + r1p_nth(ROLE, 'R', r)) * 1024) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) wres[r][c] = src[c * 64];
    }
    float* lw = lds + R1pLds::weights + r1p_lds_first(wave) * 1024;
    LW = lw;
#pragma unroll
    for (int r = 0; r < NLDS; ++r) {
      const f32x4* src = reinterpret_cast<const f32x4*>(pk + (wave * kR1pWaveBlocks + r1p_nth(ROLE, 'L', r)) * 1024) + lane;
      f32x4* dst = reinterpret_cast<f32x4*>(lw + r * 1024) + lane;
#pragma unroll
      for (int c = 0; c < 4; ++c) dst[c * 64] = src[c * 64];
    }
    img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk), 0, kR1pWaves * kR1pWaveBlocks * 4096, 0x00020000);
#pragma unroll
    for (int s = 0; s < 2; ++s) request(s, r1p_nth(ROLE, 'S', s));
    pre<r1p_next_lds(ROLE, -1)>();                     // (the first LDS block of the program)
  }
  __device__ __forceinline__ void request(int slot, int b) {
#ifdef SOCMX_R1P_FAKE_STREAM      // (developer timing experiment, WRONG results: every stream request reads the wave's block 0 -- L1 hits)
    b = 0;
#endif
    const int p = wave_bytes + b * 4096;
    ring[slot][0] = r1_gload<0>(img, loff, p);
    ring[slot][1] = r1_gload<1024>(img, loff, p);
    ring[slot][2] = r1_gload<2048>(img, loff, p);
    ring[slot][3] = r1_gload<3072>(img, loff, p);
  }
  // block B (an LDS block) into the landing registers
  template <int B>
  __device__ __forceinline__ void pre() {
    if constexpr (B >= 0) {
      static_assert(r1p_src(ROLE, B) == 'L', "pre(): an LDS block");
      constexpr int r = r1p_count(ROLE, 'L', B);
#pragma unroll
      for (int c = 0; c < 4; ++c) lq[c] = *(reinterpret_cast<const f32x4*>(LW + r * 1024) + c * 64 + lane);
    }
  }
  template <int B, int H>
  __device__ __forceinline__ void fetch(f32x4 (&w)[2]) {
    constexpr char src = r1p_src(ROLE, B);
    if constexpr (src == 'R') {
      constexpr int r = r1p_count(ROLE, 'R', B);
#pragma unroll
      for (int c = 0; c < 2; ++c) w[c] = wres[r][2 * H + c];
    } else if constexpr (src == 'L') {
#pragma unroll
      for (int c = 0; c < 2; ++c) w[c] = lq[2 * H + c];
    } else {
      constexpr int s = r1p_count(ROLE, 'S', B) % 2;
#pragma unroll
      for (int c = 0; c < 2; ++c) w[c] = ring[s][2 * H + c];
    }
  }
  // Piece C (one 16-byte load per lane) of what replaces block B once its piece C has been multiplied: the same piece of the
  // stream block two further on into the ring slot (wraps into the next step), or -- B being the LDS block in the landing
  // registers -- of the program's next LDS block.  One request behind every piece's fmas, never four back to back: a burst of
  // vector-memory instructions waits at ISSUE for room in the CU's queue while the wave could be multiplying (measured: the
  // weight stream requested block-wise cost as much as if nothing overlapped it, 0.41 -> 0.62 ms per rollout).
  template <int B, int C>
  __device__ __forceinline__ void replace() {
    if constexpr (B >= 0) {
      if constexpr (r1p_src(ROLE, B) == 'S') {
#ifndef SOCMX_R1P_NO_STREAM       // (developer timing experiment, WRONG results: the ring is never refilled -- no vector-memory traffic in the loop)
        constexpr int i = r1p_count(ROLE, 'S', B);
#ifdef SOCMX_R1P_FAKE_STREAM      // (developer timing experiment, WRONG results: every stream request reads the wave's block 0 -- L1 hits)
        const int p = wave_bytes;
#else
        const int p = wave_bytes + r1p_nth(ROLE, 'S', (i + 2) % NSTR) * 4096;
#endif
        ring[i % 2][C] = r1_gload<C * 1024>(img, loff, p);
#endif
      } else if constexpr (r1p_src(ROLE, B) == 'L') {
        constexpr int nb = r1p_next_lds(ROLE, B);
        constexpr int r = r1p_count(ROLE, 'L', nb);
        lq[C] = *(reinterpret_cast<const f32x4*>(LW + r * 1024) + C * 64 + lane);
      }
    }
  }
  // One unit of the program: blocks BA, BB (BB = -1: BA alone, on two accumulators) against sixteen wave-uniform activations,
  // piece by piece; behind every piece's fmas the requests that refill what it consumed (replace()).
  // (scheduling fences: left alone, the compiler hoists every request of the coming units as far up as dependences allow, and
  //  the registers they land in no longer fit: 256 VGPRs + scratch)
  template <int BA, int BB>
  __device__ __forceinline__ void unit(f32x2& aA, f32x2& aB, const float (&x)[16]) {
    static_assert(BB < 0 || !(r1p_src(ROLE, BA) == 'L' && r1p_src(ROLE, BB) == 'L'), "one LDS block per unit: one landing block");
    __builtin_amdgcn_sched_barrier(0);
    r1p_static_for<4>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      f32x4 wa[1], wb[1];
      piece<BA, c>(wa[0]);
      const f32x2 x0 = {x[4 * c], x[4 * c + 1]}, x1 = {x[4 * c + 2], x[4 * c + 3]};
      if constexpr (BB >= 0) {
        piece<BB, c>(wb[0]);
        aA = __builtin_elementwise_fma(f32x2{wa[0][0], wa[0][1]}, x0, aA);
        aB = __builtin_elementwise_fma(f32x2{wb[0][0], wb[0][1]}, x0, aB);
        aA = __builtin_elementwise_fma(f32x2{wa[0][2], wa[0][3]}, x1, aA);
        aB = __builtin_elementwise_fma(f32x2{wb[0][2], wb[0][3]}, x1, aB);
      } else {
        aA = __builtin_elementwise_fma(f32x2{wa[0][0], wa[0][1]}, x0, aA);
        aB = __builtin_elementwise_fma(f32x2{wa[0][2], wa[0][3]}, x1, aB);
      }
      __builtin_amdgcn_sched_barrier(0);
      replace<BA, c>();
      replace<BB, c>();
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // piece C of block B
  template <int B, int C>
  __device__ __forceinline__ void piece(f32x4& w) {
    constexpr char src = r1p_src(ROLE, B);
    if constexpr (src == 'R') w = wres[r1p_count(ROLE, 'R', B)][C];
    else if constexpr (src == 'L') w = lq[C];
    else w = ring[r1p_count(ROLE, 'S', B) % 2][C];
  }
  // ---- the skip waves' form: activations as broadcast LDS reads (every lane the same address: four values per ds_read_b128),
  //      eight at a time into one of THREE landing buffers, requested two half-steps ahead of the fmas that read them (an LDS
  //      read takes ~140 cycles, a half-step of sixteen packed fmas 70-140); a half-step multiplies them into half H of up to
  //      FOUR blocks (the unit registers that share the activations) ----
  f32x4 X[3][2];
  template <int SLOT>
  __device__ __forceinline__ void xread(const float* xs) {
#pragma unroll
    for (int c = 0; c < 2; ++c) X[SLOT][c] = *reinterpret_cast<const f32x4*>(xs + 4 * c);
  }
  template <int H, int SLOT, int B0, int B1, int B2 = -1, int B3 = -1>
  __device__ __forceinline__ void halfstep(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3) {
    constexpr int nL = (r1p_src(ROLE, B0) == 'L') + (r1p_src(ROLE, B1) == 'L') + (B2 >= 0 ? (r1p_src(ROLE, B2) == 'L') + (r1p_src(ROLE, B3) == 'L') : 0);
    static_assert(nL <= 1, "one LDS block per set of blocks: one landing block");
    __builtin_amdgcn_sched_barrier(0);
    r1p_static_for<2>([&](auto cc) {
      constexpr int c = decltype(cc)::value, C = 2 * H + c;
      f32x4 w0, w1, w2, w3;
      piece<B0, C>(w0);
      piece<B1, C>(w1);
      if constexpr (B2 >= 0) { piece<B2, C>(w2); piece<B3, C>(w3); }
      const f32x2 x0 = {X[SLOT][c][0], X[SLOT][c][1]}, x1 = {X[SLOT][c][2], X[SLOT][c][3]};
      a0 = __builtin_elementwise_fma(f32x2{w0[0], w0[1]}, x0, a0);
      a1 = __builtin_elementwise_fma(f32x2{w1[0], w1[1]}, x0, a1);
      if constexpr (B2 >= 0) {
        a2 = __builtin_elementwise_fma(f32x2{w2[0], w2[1]}, x0, a2);
        a3 = __builtin_elementwise_fma(f32x2{w3[0], w3[1]}, x0, a3);
      }
      a0 = __builtin_elementwise_fma(f32x2{w0[2], w0[3]}, x1, a0);
      a1 = __builtin_elementwise_fma(f32x2{w1[2], w1[3]}, x1, a1);
      if constexpr (B2 >= 0) {
        a2 = __builtin_elementwise_fma(f32x2{w2[2], w2[3]}, x1, a2);
        a3 = __builtin_elementwise_fma(f32x2{w3[2], w3[3]}, x1, a3);
      }
      __builtin_amdgcn_sched_barrier(0);
      replace<B0, C>();
      replace<B1, C>();
      replace<B2, C>();
      replace<B3, C>();
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  template <int B0, int B1, int B2 = -1, int B3 = -1>
  __device__ __forceinline__ void done() {}            // (the refills ride behind every piece: halfstep())
  // the sixteen registers of a DPP-form block (down_0, up_0, res_0: always resident)
  template <int B>
  __device__ __forceinline__ void resident(float (&w)[16]) {
    static_assert(r1p_src(ROLE, B) == 'R', "DPP-form blocks live in registers");
    constexpr int r = r1p_count(ROLE, 'R', B);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) w[4 * c + e] = wres[r][c][e];
  }
};

// ---- chain waves (0..3) -------------------------------------------------------------------------------------------------------
// MODE: 0 elementwise drift (double_well), 1 the same with a stopping time (molecular_dynamics), 2 OU drift (A x)
template <int MODE, int DMAX>
__device__ __forceinline__ void r1p_chain(const RolloutArgs& a, float* lds, const int q, const int lane) {
  constexpr int ROLE = 0;
  constexpr UnetDesc u = DefaultNet::desc();
  typedef R1pLds LM;
  constexpr bool STOPPING = MODE == 1, is_ou = MODE == 2;
  R1pWeights<ROLE> W;
  W.init(a, lds, q, lane);
  R1P_PROF_DECL
  const int d = a.d, K = a.K;
  const int n = lane & 15, i = n;
  const bool lane_ok = i < d;
  const float* BL = lds + LM::bias;
  // the wave's register blocks of the DPP-form layers: down_0 (lane = unit 64 q + lane; register p <-> input p of [t, x]) and
  // up_0 (the fragments (0, 4 q + f) of the standard image: register 4 f + i <-> position 4 f + i of the activation register)
  float w0[16], w8[16];
  W.resident<0>(w0);
  W.resident<21>(w8);
  const float* b0p = BL + u.L[0].b_lds + 64 * q + lane;       // (read where they are used: two registers fewer in the loop)
  const float* b8p = BL + u.L[8].b_lds + n;
  // the state: every 16-lane row of the wave runs the same arithmetic, component i = lane & 15 (all four chain waves alike)
  float x = lane_ok ? a.x0[(size_t)blockIdx.x * d + i] : 0.f;
  const float kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
  float stop = 1.f, pre_b = 0.f;
  float* A_l = lds + LM::amat;
  auto drift = [&]() {                      // b(x) of the state just formed: needed one evaluation later
    if constexpr (is_ou) {                  // OU_quadratic.py:51-52, OU_linear.py:43-44
      float b_even = 0.f, b_odd = 0.f;
      r1p_static_for<DMAX>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float aj = A_l[j * 16 + i];                     // A[i][j]; columns past d are zero
        if constexpr (j & 1) r1_fmac_bc<j>(b_odd, x, aj);
        else r1_fmac_bc<j>(b_even, x, aj);
      });
      pre_b = lane_ok ? b_even + b_odd : 0.f;
    } else {
      pre_b = -2.f * kap * (x * x - 1.f) * 2.f * x;           // double_well.py:44-48
    }
  };
  // r1 of the wave's quarter: relu(down_0 [t, x] + b), lane = unit 64 q + lane; written to LDS (the skip wave's copy, and
  // this wave's own broadcast source)
  auto first_layer = [&](float t) -> float {
    float a0 = fmaf(t, w0[0], *b0p), a1 = 0.f;
    r1_state_one<DMAX>(a0, a1, x, &w0[1]);
    const float y = relu_keep_nan(a0 + a1);
    lds[LM::xr1 + 64 * q + lane] = y;                          // (the skip wave's copy: res_1)
    return y;
  };
  // Euler-Maruyama step j from the finished evaluation of x_j (utils.py:37-101): every chain wave alike; wave 0 hands the
  // step's nabla_V, new state, fractional step and stop flag to skip wave 0 (running costs and stores)
  auto sde_step = [&](int j) {
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 4);
    const float r0 = lds[LM::res0 + n];
    const f32x4 scal = *reinterpret_cast<const f32x4*>(lds + LM::sc + (j % 3) * 4);
    const float eps = lds[LM::nz + (j % 3) * 16 + i];
    const float dt = scal[0], sq_ldt = scal[1];
    const float gv = relu_keep_nan(r1p_sum4(pa) + *b8p) + r0;
    const float su = lane_ok ? -gv : 0.f;                               // sigma u = -nabla_V (sigma = I; method.py:58-80)
    const float upd = (pre_b + su) * dt + sq_ldt * eps;                 // utils.py:45-47
    const float xn = x + stop * upd;                                    // utils.py:48
    float xe = xn, step = dt, stop_new = 1.f;
    if (STOPPING) {                                                     // utils.py:42-44, 49-75; Phi = -x_0
      const float phi_b = -__shfl(x, 0, 16), phi_a = -__shfl(xn, 0, 16);
      const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
      const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
      const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
      xe = js * (x + fr * stop * upd) + (1.f - js) * xn;
      step = js * (fr * fr) * dt + ns * dt;                             // step_fraction squared (utils.py:70-72)
      stop_new = (-__shfl(xe, 0, 16) > 0.f) ? 1.f : 0.f;
    }
    x = lane_ok ? xe : 0.f;
    if (STOPPING) stop = stop_new;
    if (q == 0 && lane < 16) {
      lds[LM::bk + i] = gv;
      lds[LM::bk + 16 + i] = x;
      if (lane == 0) { lds[LM::bk + 32] = step; lds[LM::bk + 33] = stop; }
    }
  };

  // ---- one evaluation of the network on the current state: five barriers; leaves up_0's partial sums in p5 ----
  auto evaluate = [&](float t) {
    // P0: r1 quarter -> down_1 partials (blocks 1..8: four k16-groups x two unit registers)
    const float r1v = first_layer(t);
    drift();
    {
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      float x[16];
      r1p_gather<0>(x, r1v);
      W.unit<1, 2>(A0, A1, x);
      r1p_gather<16>(x, r1v);
      W.unit<3, 4>(A0, A1, x);
      r1p_gather<32>(x, r1v);
      W.unit<5, 6>(A0, A1, x);
      r1p_gather<48>(x, r1v);
      W.unit<7, 8>(A0, A1, x);
      lds[LM::p1 + lane * 4 + q] = A0.x + A0.y;
      lds[LM::p1 + (64 + lane) * 4 + q] = A1.x + A1.y;
    }
    R1P_TICK(0)
    __syncthreads();
    R1P_TICK(1)
    // P1: r2 = relu(sum of down_1's partials + b), positions 32 q .. 32 q + 31 (both halves of the wave alike) -> down_2 partials
    {
      const int pos = 32 * q + (lane & 31);
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + LM::p1 + pos * 4);
      const float y = relu_keep_nan(r1p_sum4(v) + BL[u.L[1].b_lds + pos]);
      if (lane < 32) lds[LM::xr2 + pos] = y;                   // (the skip wave's copy: res_2)
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      float x[16];
      r1p_gather<0>(x, y);
      W.unit<9, -1>(A0, A1, x);
      r1p_gather<16>(x, y);
      W.unit<10, -1>(A0, A1, x);
      lds[LM::p2 + lane * 4 + q] = (A0.x + A0.y) + (A1.x + A1.y);
    }
    R1P_TICK(2)
    __syncthreads();
    R1P_TICK(3)
    // P2: r3 = relu(sum of down_2's partials + b), positions 16 q .. 16 q + 15 (every row alike) -> up_2 partials (blocks 11, 12)
    {
      const int pos = 16 * q + (lane & 15);
      const f32x4 v = *reinterpret_cast<const f32x4*>(lds + LM::p2 + pos * 4);
      const float y = relu_keep_nan(r1p_sum4(v) + BL[u.L[2].b_lds + pos]);
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      float x[16];
      r1p_gather<0>(x, y);
      W.unit<11, 12>(A0, A1, x);
      lds[LM::p3 + lane * 4 + q] = A0.x + A0.y;
      lds[LM::p3 + (64 + lane) * 4 + q] = A1.x + A1.y;
    }
    R1P_TICK(4)
    __syncthreads();
    R1P_TICK(5)
    // P3: o2 = relu(sum of up_2's partials + b) + sum of res_2's partials + b, positions 32 q .. -> up_1 partials (13..20)
    {
      const int pos = 32 * q + (lane & 31);
      const f32x4 vu = *reinterpret_cast<const f32x4*>(lds + LM::p3 + pos * 4);
      const f32x4 vr = *reinterpret_cast<const f32x4*>(lds + LM::pr2 + pos * 4);
      const float y = relu_keep_nan(r1p_sum4(vu) + BL[u.L[6].b_lds + pos]) + (r1p_sum4(vr) + BL[u.L[5].b_lds + pos]);
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f}, A2 = {0.f, 0.f}, A3 = {0.f, 0.f};
      float x[16];
      r1p_gather<0>(x, y);
      W.unit<13, 14>(A0, A1, x);
      W.unit<15, 16>(A2, A3, x);
      r1p_gather<16>(x, y);
      W.unit<17, 18>(A0, A1, x);
      W.unit<19, 20>(A2, A3, x);
      lds[LM::p4 + lane * 4 + q] = A0.x + A0.y;
      lds[LM::p4 + (64 + lane) * 4 + q] = A1.x + A1.y;
      lds[LM::p4 + (128 + lane) * 4 + q] = A2.x + A2.y;
      lds[LM::p4 + (192 + lane) * 4 + q] = A3.x + A3.y;
    }
    R1P_TICK(6)
    __syncthreads();
    R1P_TICK(7)
    // P4: o1 = relu(sum of up_1's partials + b) + sum of res_1's partials + b at position 64 q + lane = unit 64 q + perm(lane):
    // the activation register of the DPP form -> this quarter's share of up_0 (sixteen fmacs), summed over the four rows
    {
      const int pos = 64 * q + lane, unit = 64 * q + r1_perm(lane);
      const f32x4 vu = *reinterpret_cast<const f32x4*>(lds + LM::p4 + pos * 4);
      const f32x4 vr = *reinterpret_cast<const f32x4*>(lds + LM::pr1 + pos * 4);
      const float o1 = relu_keep_nan(r1p_sum4(vu) + BL[u.L[7].b_lds + unit]) + (r1p_sum4(vr) + BL[u.L[4].b_lds + unit]);
      float p0 = 0.f, p1 = 0.f;
      r1_fmac8<0>(p0, p1, o1, &w8[0]);
      r1_fmac8<1>(p0, p1, o1, &w8[8]);
      const float y = r1_rows_sum(p0 + p1);
      if (lane < 16) lds[LM::p5 + n * 4 + q] = y;
    }
    R1P_TICK(8)
    __syncthreads();
    R1P_TICK(9)
  };

  // ---- prologue: two barriers (skip waves: noise of step 0, scalars, res_0 of the initial state) ----
  uint64_t key_seed = 0, key_offset = 0;
  if (q == 0) rollout_key(a, key_seed, key_offset);      // (thread 0 advances a device-resident key: rollout_key_advance)
  if constexpr (is_ou) {
    if (q == 0)
      for (int e = lane; e < 256; e += 64) A_l[e] = ((e & 15) < d && (e >> 4) < d) ? a.A[(e & 15) * d + (e >> 4)] : 0.f;
  }
  __syncthreads();
  if (q == 0) rollout_key_advance(a, key_offset);        // (every wave that draws noise read the key in front of this barrier)
  __syncthreads();
  __builtin_amdgcn_s_setprio(2);
  R1P_PROF_START
  for (int c = 0; c < K; ++c) {
    if (c > 0) sde_step(c - 1);
    R1P_TICK(10)
    evaluate(a.ts[c]);
  }
  R1P_PROF_END(q)
  sde_step(K - 1);
  if (a.nabla_v) {                       // nabla_V(T, X_K) (method.py:272-278 evaluates it on every grid point)
    evaluate(a.ts[K]);
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 4);
    const float gv = relu_keep_nan(r1p_sum4(pa) + *b8p) + lds[LM::res0 + n];
    if (q == 0 && lane < 16 && lane_ok) a.nabla_v[((size_t)K * a.B + blockIdx.x) * d + i] = gv;
  } else {
    __syncthreads();                     // (the last step's hand-over to skip wave 0)
  }
}

// ---- skip waves (4..7) -----------------------------------------------------------------------------------------------------------
template <int MODE, int DMAX, bool BOOKS>
__device__ __forceinline__ void r1p_skip(const RolloutArgs& a, float* lds, const int wave, const int lane) {
  constexpr int ROLE = BOOKS ? 1 : 2;
  constexpr UnetDesc u = DefaultNet::desc();
  typedef R1pLds LM;
  constexpr bool STOPPING = MODE == 1, is_ou = MODE == 2;
  const int q = wave & 3;
  R1pWeights<ROLE> W;
  W.init(a, lds, wave, lane);
  R1P_PROF_DECL
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int grow = blockIdx.x;
  const int n = lane & 15, i = n;
  const bool lane_ok = i < d;
  const int ic = min(i, d - 1);
  const float* BL = lds + LM::bias;
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);

  // ---- the books (wave 4): running costs, trajectory stores, res_0 of the coming evaluation ----
  float w3[16];
  float b3 = 0.f, lpd = 0.f, lps = 0.f, xlast = 0.f;
  const bool is_quad = is_ou && kind == SOCMX_OU_QUADRATIC;
  const bool traj = a.states != nullptr;
  const bool store = BOOKS && lane < 16 && lane_ok && traj;
  const bool store0 = BOOKS && lane == 0 && traj;
  const uint32_t rowoff = (uint32_t)(grow * d + i);
  const size_t step_floats = (size_t)B * d;
  size_t kbd = 0, kb = 0;
  float* P_l = lds + LM::pmat;
  if constexpr (BOOKS) {
    W.template resident<20>(w3);
    b3 = BL[u.L[3].b_lds + n];
    xlast = lane_ok ? a.x0[(size_t)grow * d + i] : 0.f;
    if (is_quad)
      for (int e = lane; e < 256; e += 64) P_l[e] = ((e & 15) < d && (e >> 4) < d) ? a.P[(e >> 4) * d + (e & 15)] : 0.f;
    if (store) a.states[rowoff] = xlast;
    if (store0) a.stop_ind[grow] = 1.f;
  }
  auto res0_of = [&](float t, float xv) {       // res_0 [t, x] + b: lane n = unit n (every row alike)
    float r0 = fmaf(t, w3[0], b3), r1v = 0.f;
    r1_state_one<DMAX>(r0, r1v, xv, &w3[1]);
    if (lane < 16) lds[LM::res0 + n] = r0 + r1v;
  };
  // the finished step j: costs (utils.py:92-99) from the hand-over.  The stores wait for the slack behind B4 (they share
  // the vector-memory counter with the weight stream: issued here they would sit in front of every stream block this wave
  // waits for during the skip GEMMs) and read the hand-over again there -- no register is held for them in between.
  int st_k = -1;
  auto books = [&](int j) {
    const float gv = lds[LM::bk + i], xe = lds[LM::bk + 16 + i];
    const f32x4 scal = *reinterpret_cast<const f32x4*>(lds + LM::sc + (j % 3) * 4);
    const float eps = lds[LM::nz + (j % 3) * 16 + i];
    const float uc = lane_ok ? -gv : 0.f;                               // u = -sigma^T nabla_V (method.py:58-80)
    float f = 0.f;                                                      // f at the NEW state, OLD time (utils.py:92-96)
    if (is_quad) {
      float px = 0.f;
      for (int jj = 0; jj < d; ++jj) px += P_l[ic * 16 + jj] * __shfl(xe, jj, 16);
      f = row16_sum(lane_ok ? xe * px : 0.f);
    } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
      f = 1.f;
    }
    const float uu = row16_sum(uc * uc), ue = row16_sum(uc * eps);
    const float sol = STOPPING ? lds[LM::bk + 32] / a.lmbd : scal[2];
    const float ssol = STOPPING ? sqrtf(sol) : scal[3];
    lpd = lpd + sol * (-f - 0.5f * uu);
    lps = lps + ssol * (-ue);
    st_k = j;
    xlast = xe;
  };
  auto stores = [&]() {
    if (st_k < 0) return;
    const int j = st_k;
    st_k = -1;
    // (every LDS operand first: one round trip, then the six stores back to back)
    const float gv = lds[LM::bk + i], eps = lds[LM::nz + (j % 3) * 16 + i], xe = lds[LM::bk + 16 + i];
    const float fr = STOPPING ? lds[LM::bk + 32] : lds[LM::sc + (j % 3) * 4];
    const float sp = STOPPING ? lds[LM::bk + 33] : 1.f;
    __builtin_amdgcn_sched_barrier(0);
    if (store) {
      if (a.nabla_v) (a.nabla_v + kbd)[rowoff] = gv;
      (a.controls + kbd)[rowoff] = -gv;
      (a.noises + kbd)[rowoff] = eps;
      (a.states + kbd + step_floats)[rowoff] = xe;
    }
    kbd += step_floats;
    if (store0) {
      (a.frac + kb)[grow] = fr;
      (a.stop_ind + kb + B)[grow] = sp;
    }
    kb += B;
  };

  // ---- noise one step ahead: wave 5 the Philox words of step k + 2, wave 6 Box-Muller on the words of step k + 1, wave 7 the
  //      step's scalars (dt, sqrt(lambda dt), dt / lambda and its root: utils.py:38, 47) ----
  float* NZ = lds + LM::nz;
  uint32_t* WZ = reinterpret_cast<uint32_t*>(lds + LM::wz);
  const bool w_words = wave == 5 && lane < 8 && !a.noise_in, w_draws = wave == 6 && lane < 8;
  auto words = [&](int k) {
    if (!w_words || k >= K) return;
    uint32_t wa, wb;
    philox_pair_words(key_seed, key_offset, (uint32_t)(a.row0 + grow), (uint32_t)k, lane >> 1, lane & 1, wa, wb);
    WZ[((k & 1) * 8 + lane) * 2] = wa;
    WZ[((k & 1) * 8 + lane) * 2 + 1] = wb;
  };
  auto draws = [&](int k) {
    if (!w_draws || k >= K) return;
    float z0 = 0.f, z1 = 0.f;
    const int c0 = 2 * lane;
    if (a.noise_in) {
      const float* src = a.noise_in + ((size_t)k * B + grow) * d;
      if (c0 < d) z0 = src[c0];
      if (c0 + 1 < d) z1 = src[c0 + 1];
    } else {
      box_muller_pair(WZ[((k & 1) * 8 + lane) * 2], WZ[((k & 1) * 8 + lane) * 2 + 1], z0, z1);
    }
    NZ[(k % 3) * 16 + c0] = c0 < d ? z0 : 0.f;
    NZ[(k % 3) * 16 + c0 + 1] = c0 + 1 < d ? z1 : 0.f;
  };
  auto step_scalars = [&](int k) {
    if (wave == 7 && lane == 0 && k < K) {
      const float dt = a.ts[k + 1] - a.ts[k];
      const float dol = dt / a.lmbd;
      *reinterpret_cast<f32x4*>(lds + LM::sc + (k % 3) * 4) = f32x4{dt, sqrtf(a.lmbd * dt), dol, sqrtf(dol)};
    }
  };

  // ---- the skip GEMMs of one evaluation: res_1 on the wave's quarter of r1 (blocks 0-3, 8-19), res_2 on its quarter of r2 (4-7) ----
  auto evaluate = [&](int c) {
    // in front of B1 the chain waves integrate and run down_0 / down_1 (~1k cycles) and nothing is here for these waves yet:
    // the noise and the scalars of the coming steps (three-deep rings: the step being integrated reads other slots)
    words(c + 2);
    draws(c + 1);
    step_scalars(c + 1);
    R1P_TICK(10) __syncthreads(); R1P_TICK(1)                                                     // B1: r1 is in LDS; the finished step's hand-over too
    f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f}, A2 = {0.f, 0.f}, A3 = {0.f, 0.f};
    const float* xs1 = lds + LM::xr1 + 64 * q;                          // the quarter of r1: four groups of sixteen
    const float* xs2 = lds + LM::xr2 + 32 * q;                          // the quarter of r2: two groups
    // half-steps in time order T = 0 .. 11, landing buffer T % 3, requested two half-steps ahead where the activations exist:
    //   B1 | T0, T1: res_1 group 0 | B2 | T2 .. T5: res_2 | B3 | T6 .. T11: res_1 groups 1 .. 3 | B4
    // (every phase shorter than the chain waves' own: res_1's groups 1 .. 3 are not needed before B4)
    W.template xread<0>(xs1);
    W.template xread<1>(xs1 + 8);
    if constexpr (BOOKS) {
      if (c > 0) {
        books(c - 1);
        res0_of(a.ts[c], xlast);
      }
    }
    W.template halfstep<0, 0, 0, 1, 2, 3>(A0, A1, A2, A3);
    W.template halfstep<1, 1, 0, 1, 2, 3>(A0, A1, A2, A3);
    W.template done<0, 1, 2, 3>();
    R1P_TICK(2) __syncthreads(); R1P_TICK(3)                                                     // B2: r2 is in LDS
    {
      f32x2 R0 = {0.f, 0.f}, R1 = {0.f, 0.f}, R2 = {0.f, 0.f}, R3 = {0.f, 0.f};
      W.template xread<2>(xs2);
      W.template xread<0>(xs2 + 8);
      W.template xread<1>(xs2 + 16);
      W.template halfstep<0, 2, 4, 5>(R0, R1, R2, R3);
      W.template xread<2>(xs2 + 24);
      W.template halfstep<1, 0, 4, 5>(R0, R1, R2, R3);
      W.template done<4, 5>();
      W.template xread<0>(xs1 + 16);                                    // (T6, T7: r1 is two barriers old -- they ride across B3)
      W.template halfstep<0, 1, 6, 7>(R0, R1, R2, R3);
      W.template xread<1>(xs1 + 24);
      W.template halfstep<1, 2, 6, 7>(R0, R1, R2, R3);
      W.template done<6, 7>();
      lds[LM::pr2 + lane * 4 + q] = R0.x + R0.y;
      lds[LM::pr2 + (64 + lane) * 4 + q] = R1.x + R1.y;
    }
    R1P_TICK(4) __syncthreads(); R1P_TICK(5)                                                     // B3
    W.template xread<2>(xs1 + 32);
    W.template halfstep<0, 0, 8, 9, 10, 11>(A0, A1, A2, A3);
    W.template xread<0>(xs1 + 40);
    W.template halfstep<1, 1, 8, 9, 10, 11>(A0, A1, A2, A3);
    W.template done<8, 9, 10, 11>();
    W.template xread<1>(xs1 + 48);
    W.template halfstep<0, 2, 12, 13, 14, 15>(A0, A1, A2, A3);
    W.template xread<2>(xs1 + 56);
    W.template halfstep<1, 0, 12, 13, 14, 15>(A0, A1, A2, A3);
    W.template done<12, 13, 14, 15>();
    W.template halfstep<0, 1, 16, 17, 18, 19>(A0, A1, A2, A3);
    W.template halfstep<1, 2, 16, 17, 18, 19>(A0, A1, A2, A3);
    W.template done<16, 17, 18, 19>();
    lds[LM::pr1 + lane * 4 + q] = A0.x + A0.y;
    lds[LM::pr1 + (64 + lane) * 4 + q] = A1.x + A1.y;
    lds[LM::pr1 + (128 + lane) * 4 + q] = A2.x + A2.y;
    lds[LM::pr1 + (192 + lane) * 4 + q] = A3.x + A3.y;
    R1P_TICK(6) __syncthreads(); R1P_TICK(7)                                                     // B4
    // (the stores of the finished step: nothing of the chain waits for this wave until the next evaluation's B3)
    if constexpr (BOOKS) stores();
    R1P_TICK(8) __syncthreads(); R1P_TICK(9)                                                     // B5
  };

  // ---- prologue ----
  words(0);
  if constexpr (BOOKS) res0_of(a.ts[0], xlast);
  __syncthreads();
  draws(0);
  words(1);
  step_scalars(0);
  __syncthreads();
  R1P_PROF_START
  for (int c = 0; c < K; ++c) evaluate(c);
  R1P_PROF_END(wave)
  if (a.nabla_v) {
    evaluate(K);
  } else {
    __syncthreads();
    if constexpr (BOOKS) books(K - 1);
  }
  if constexpr (BOOKS) {
    stores();
    float gval = 0.f;                                                   // terminal cost (utils.py:101)
    const float xK = xlast;
    if (kind == SOCMX_OU_QUADRATIC) {
      float qx = 0.f;
      for (int jj = 0; jj < d; ++jj) qx += a.Q[ic * d + jj] * __shfl(xK, jj, 16);
      gval = row16_sum(lane_ok ? xK * qx : 0.f);
    } else if (kind == SOCMX_OU_LINEAR) {
      gval = row16_sum(lane_ok ? a.omega[ic] * xK : 0.f);
    } else if (kind == SOCMX_DOUBLE_WELL) {
      const float qq = xK * xK - 1.f;
      gval = row16_sum(lane_ok ? a.nu[ic] * (qq * qq) : 0.f);
    }
    if (lane == 0) {
      a.lpd[grow] = lpd;
      a.lps[grow] = lps;
      a.ltw[grow] = -gval / a.lmbd;
    }
  }
}

template <int MODE, int DMAX>
__global__ __launch_bounds__(kR1pWaves * 64) void rollout1p_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < R1pLds::bias; e += kR1pWaves * 64) lds[e] = 0.f;
  unet_load_biases_at(a.packed, DefaultNet::desc(), lds + R1pLds::bias, tid, kR1pWaves * 64);
  __syncthreads();
#ifdef SOCMX_R1P_ONLY_ROLE        // (developer: the register need of ONE role, -Rpass-analysis=kernel-resource-usage; not a working kernel)
  if (SOCMX_R1P_ONLY_ROLE == 0) r1p_chain<MODE, DMAX>(a, lds, wave & 3, lane);
  else if (SOCMX_R1P_ONLY_ROLE == 1) r1p_skip<MODE, DMAX, true>(a, lds, 4, lane);
  else r1p_skip<MODE, DMAX, false>(a, lds, 5 + (wave & 1), lane);
#else
  if (wave < 4) r1p_chain<MODE, DMAX>(a, lds, wave, lane);
  else if (wave == 4) r1p_skip<MODE, DMAX, true>(a, lds, wave, lane);
  else r1p_skip<MODE, DMAX, false>(a, lds, wave, lane);
#endif
}

// ---- the second weight image ---------------------------------------------------------------------------------------------------
struct PackPkArgs {
  int fin[9], fout[9];
  const float* w[9];
  float* pk;
};
__global__ void unet_pack_pk_kernel(const PackPkArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= r1p_image_floats()) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, c = (idx >> 8) & 3, blk = idx >> 10;
  const int wave = blk / kR1pWaveBlocks, b = blk - wave * kR1pWaveBlocks;
  float v = 0.f;
  if (b < (wave < 4 ? kR1pChainBlocks : kR1pSkipBlocks)) {
    const R1pBlk B = r1p_block(wave, b);
    int unit, k;
    if (B.form == R1P_PK) {
      // lane = the unit's POSITION inside register j; up_1 / res_1 (whose outputs feed the DPP form of up_0) order their units so
      // that the sums land in the activation layout: position 64 j + lane <-> unit 64 j + perm(lane)
      unit = 64 * B.j + ((B.layer == 7 || B.layer == 4) ? r1_perm(lane) : lane);
      k = 16 * B.kg + 4 * c + e;
    } else if (B.form == R1P_STATE) {
      unit = B.layer == 3 ? (lane & 15) : 64 * B.j + lane;       // res_0: sixteen units, every row alike
      k = 4 * c + e;                                               // input p of [t, x_0 .. x_14]
    } else {
      unit = lane & 15;                                            // the standard fragment (0, kc = 4 q + c), component e
      k = 16 * (4 * B.kg + c) + 4 * (lane >> 4) + e;
    }
    if (unit < a.fout[B.layer] && k < a.fin[B.layer]) v = a.w[B.layer][(size_t)unit * a.fin[B.layer] + k];
  }
  a.pk[idx] = v;
}

bool rollout1p_available() { return r1_supported_default(); }
int rollout1p_pack(const socmx_unet* net, float* pk, void* stream) {
  PackPkArgs a;
  const int h[3] = {net->hdims[0], net->hdims[1], net->hdims[2]};
  unet_layer_dims(net->d, h, a.fin, a.fout);
  for (int l = 0; l < 9; ++l) a.w[l] = net->weight[l];
  a.pk = pk;
  const int threads = 256, blocks = (r1p_image_floats() + threads - 1) / threads;
  return launch(unet_pack_pk_kernel, dim3(blocks), dim3(threads), 0, stream, a);
}

int rollout1p_launch(const RolloutArgs& a, bool stopping, void* stream) {
  if constexpr (r1_supported_default()) {
    if (!a.sigma_identity || a.d > 15) return SOCMX_E_DIM;
    void (*k)(const RolloutArgs);
    const bool ou = a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR;
#define R1PPICK(DM) (stopping ? rollout1p_kernel<1, DM> : ou ? rollout1p_kernel<2, DM> : rollout1p_kernel<0, DM>)
    if (a.d <= 3) k = R1PPICK(3);
    else if (a.d <= 11) k = R1PPICK(11);
    else k = R1PPICK(15);
#undef R1PPICK
    if (const int err = ensure_max_lds(k)) return err;
    // (the CU's whole LDS: one workgroup per CU, nobody else's workgroups beside this latency-bound chain)
    return launch(k, dim3((unsigned)a.B), dim3(kR1pWaves * 64), (size_t)kLdsBytesPerCU, stream, a);
  } else {
    return SOCMX_E_DIM;          // (an architecture-variant build: these kernels exist for the default widths only)
  }
}

}  // namespace socmx
