// socmx_rollout1p.hip -- the fused Euler-Maruyama rollout, ONE ROW PER WORKGROUP, on PACKED fp32 multiply-adds (gfx950).
//
// Replaces reference SOC_matching/utils.py:17-128 (stochastic_trajectories), method.py:58-80 (control) and
// models.py:233-242 (FullyConnectedUNet.forward) for training-size batches (B <= 256 rows) at the default hidden widths,
// sigma = I, d <= 15 -- BASELINE configs[1] / [2] and the README's molecular_dynamics run.  Round 5's successor of
// socmx_rollout1.hip's v_fmac_f32_dpp form for these shapes.
//
// Why another form.  Measured on this chip (tools/ubench/valu_banks.hip, profiles/r5/valu_banks.txt): v_fmac_f32_dpp issues at
// 4.5 cycles per SIMD (64 MACs) whatever the registers, v_pk_fma_f32 at 4.4 (128 MACs) -- the DPP operand costs a second pass.
// So the matrix-VECTOR products run as
//     acc[lane = unit].{lo, hi} += W[unit][k, k + 1] * x[k, k + 1]                       (v_pk_fma_f32, 128 MACs)
// with the activation pair wave-uniform, in an SGPR pair (a packed fma reads one scalar pair at no cost): the wave that OWNS
// x[k] -- it just summed it, one value per lane -- pulls it out with v_readlane_b32.  Split-K over the waves: wave w owns an
// eighth of every layer's input, multiplies it into ALL the units of the layers that read that input and leaves per-wave
// partial sums in LDS ([position][wave]: two ds_read_b128 per position) for whoever owns that unit as an input of the next
// layer.  One barrier per layer of the chain, five per step; every wave runs the same program:
//     P0 [sum nabla_V, Euler-Maruyama (every wave, redundantly), down_0 of its 32 units] -> down_1 | P1 down_2 | P2 up_2 |
//     P3 up_1 | P4 up_0
// and the skip GEMMs res_1 (39 % of the network's MACs) and res_2 -- whose inputs the wave owns as well and whose outputs
// are needed phases later -- sit in the SHADOW of each phase's LDS round trip (the partial-sum reads are issued, the skip
// unit's fmas run, then the sums are formed), so that the chain's latency hides them.
// Weights: a second image behind the fragment-ordered one (socmx_unet_pack_f32 writes both), wave-major, blocks in the order
// the wave consumes them (socmx_rollout1p.h).  Per wave a compile-time plan says where each block lives: R registers for the
// whole launch, L copied to LDS once, S streamed from L2 every step through a two-block ring -- every wave streams the same
// share, spread over the whole step (the CU's vector-memory path delivers ~95 B/clk: tools/ubench/l2ring).
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

// ---- where a block's weights come from: 'R' registers, 'L' LDS, 'S' L2 stream.  role 0 = waves 0..6, role 1 = wave 7 (the books:
//      it also holds res_0 -- block 22 -- the running costs and the stores, so fewer register blocks).  Blocks in consumption
//      order: socmx_rollout1p.h ----
__host__ __device__ constexpr int r1p_blocks(int role) { return role == 0 ? kR1pBlocks : kR1pBlocks + 1; }
__host__ __device__ constexpr char r1p_src(int role, int b) {
#ifdef SOCMX_R1P_PLAN0
  constexpr char plan0[kR1pBlocks + 1] = SOCMX_R1P_PLAN0;
#else
  constexpr char plan0[kR1pBlocks + 1] = "RSRLRSSRLRSSRLRSSRLRSR";
#endif
#ifdef SOCMX_R1P_PLAN1
  constexpr char plan1[kR1pBlocks + 2] = SOCMX_R1P_PLAN1;
#else
  constexpr char plan1[kR1pBlocks + 2] = "RSRSLSRSSSRLSSRSSLRSSRR";
#endif
  return role == 0 ? plan0[b] : plan1[b];
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
static_assert(r1p_src(0, 0) == 'R' && r1p_src(0, 21) == 'R' && r1p_src(1, 0) == 'R' && r1p_src(1, 21) == 'R' && r1p_src(1, 22) == 'R',
              "the DPP-form blocks (down_0, up_0, res_0) are register blocks");
static_assert(r1p_count(0, 'S') % 2 == 0 && r1p_count(1, 'S') % 2 == 0 && r1p_count(0, 'S') >= 2 && r1p_count(1, 'S') >= 2, "static ring slots across steps");
// first LDS block of a wave's LDS-resident blocks
__host__ __device__ constexpr int r1p_lds_first(int wave) { return wave * r1p_count(0, 'L'); }
__host__ __device__ constexpr int r1p_lds_blocks() { return 7 * r1p_count(0, 'L') + r1p_count(1, 'L'); }

// LDS map (floats).  Partial sums [position][wave]: the owner of a position reads its eight partial sums as two ds_read_b128.
struct R1pLds {
  static constexpr int p1 = 0;           // (128, 8) down_1      (p3, up_2's, lives in the same memory: p1 is read in P1, p3 written in P2)
  static constexpr int p3 = 0;           // (128, 8)
  static constexpr int p2 = 1024;        // (64, 8)  down_2
  static constexpr int pr2 = 1536;       // (128, 8) res_2
  static constexpr int p4 = 2560;        // (256, 8) up_1
  static constexpr int pr1 = 4608;       // (256, 8) res_1
  static constexpr int p5 = 6656;        // (16, 8)  up_0
  static constexpr int res0 = 6784;      // (16)  res_0 [t_k, x_k] + b of the evaluation under way (wave 7)
  static constexpr int nz = 6800;        // (24, 16) noise of 24 steps: batches of eight steps, one batch ahead (wave 6)
  static constexpr int sc = 7184;        // (32, 4) per-step scalars: batches of sixteen steps (wave 5)
  static constexpr int amat = 7312;      // (16, 16) A TRANSPOSED (amat[j * 16 + i] = A[i][j]: lanes along i), P row-major
  static constexpr int pmat = 7568;
  static constexpr int bias = 7824;      // the nine layers' padded biases (image order)
  static constexpr int weights = 9216;   // LDS-resident blocks, 1024 floats each, wave by wave
};
static_assert(R1pLds::bias + 1248 <= R1pLds::weights, "bias copy");
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

template <int N>
__device__ __forceinline__ void r1p_gather_n(float (&x)[16], float v) {
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
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
    img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk), 0, r1p_image_floats() * 4, 0x00020000);
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
  // up_2 on the wave's EIGHT inputs (block B: pieces 0, 1 = unit register 0's weights of inputs 0 .. 7, pieces 2, 3 = register 1's)
  template <int B>
  __device__ __forceinline__ void unit8(f32x2& a0, f32x2& a1, const float (&x)[16]) {
    __builtin_amdgcn_sched_barrier(0);
    r1p_static_for<4>([&](auto cc) {
      constexpr int c = decltype(cc)::value, kc = c & 1;
      f32x4 w;
      piece<B, c>(w);
      const f32x2 x0 = {x[4 * kc], x[4 * kc + 1]}, x1 = {x[4 * kc + 2], x[4 * kc + 3]};
      if constexpr (c < 2) {
        a0 = __builtin_elementwise_fma(f32x2{w[0], w[1]}, x0, a0);
        a0 = __builtin_elementwise_fma(f32x2{w[2], w[3]}, x1, a0);
      } else {
        a1 = __builtin_elementwise_fma(f32x2{w[0], w[1]}, x0, a1);
        a1 = __builtin_elementwise_fma(f32x2{w[2], w[3]}, x1, a1);
      }
      __builtin_amdgcn_sched_barrier(0);
      replace<B, c>();
      __builtin_amdgcn_sched_barrier(0);
    });
  }
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


__device__ __forceinline__ float r1p_sum8(const f32x4 a, const f32x4 b) { return ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])); }

// ---- one wave of the workgroup ------------------------------------------------------------------------------------------------
// MODE: 0 elementwise drift (double_well), 1 the same with a stopping time (molecular_dynamics), 2 OU drift (A x)
// BOOKS (wave 7): also res_0 of the coming evaluation, the running costs and every global store
template <int MODE, int DMAX, bool BOOKS>
__device__ __forceinline__ void r1p_wave(const RolloutArgs& a, float* lds, const int w, const int lane) {
  constexpr int ROLE = BOOKS ? 1 : 0;
  constexpr UnetDesc u = DefaultNet::desc();
  typedef R1pLds LM;
  constexpr bool STOPPING = MODE == 1, is_ou = MODE == 2;
  R1pWeights<ROLE> W;
  W.init(a, lds, w, lane);
  R1P_PROF_DECL
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int grow = blockIdx.x;
  const int n = lane & 15, i = n;
  const bool lane_ok = i < d;
  const int ic = min(i, d - 1);
  const float* BL = lds + LM::bias;
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);
  // the wave's register blocks of the DPP-form layers: down_0 (lane = unit 32 w + (lane & 31); register p <-> input p of [t, x])
  // and up_0 (the fragments (0, 2 w), (0, 2 w + 1) of the standard image: register 4 f + i <-> position 4 f + i, f < 2)
  float w0[16], w8[16];
  W.template resident<0>(w0);
  W.template resident<21>(w8);
  const float* b0p = BL + u.L[0].b_lds + 32 * w + (lane & 31);
  const float* b8p = BL + u.L[8].b_lds + n;
  // the state: every 16-lane row of every wave runs the same arithmetic, component i = lane & 15
  float x = lane_ok ? a.x0[(size_t)grow * d + i] : 0.f;
  const float kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
  float stop = 1.f, pre_b = 0.f;
  float* A_l = lds + LM::amat;
  float* P_l = lds + LM::pmat;
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
  // r1 of the wave's 32 units: relu(down_0 [t, x] + b), lane l and l + 32 alike
  auto first_layer = [&](float t) -> float {
    float a0 = fmaf(t, w0[0], *b0p), a1 = 0.f;
    r1_state_one<DMAX>(a0, a1, x, &w0[1]);
    return relu_keep_nan(a0 + a1);
  };

  // ---- the books (wave 7) ----
  float w3[16];
  float b3 = 0.f, lpd = 0.f, lps = 0.f;
  const bool is_quad = is_ou && kind == SOCMX_OU_QUADRATIC;
  const bool traj = a.states != nullptr;
  const bool store = BOOKS && lane < 16 && lane_ok && traj;
  const bool store0 = BOOKS && lane == 0 && traj;
  const uint32_t rowoff = (uint32_t)(grow * d + i);
  const size_t step_floats = (size_t)B * d;
  size_t kbd = 0, kb = 0;
  if constexpr (BOOKS) {
    W.template resident<kR1pRes0Block>(w3);
    b3 = BL[u.L[3].b_lds + n];
    if (store) a.states[rowoff] = x;
    if (store0) a.stop_ind[grow] = 1.f;
  }
  auto res0_of = [&](float t) {             // res_0 [t, x] + b of the CURRENT state: lane n = unit n (every row alike)
    float r0 = fmaf(t, w3[0], b3), r1v = 0.f;
    r1_state_one<DMAX>(r0, r1v, x, &w3[1]);
    if (lane < 16) lds[LM::res0 + n] = r0 + r1v;
  };
  // what the finished step leaves for the books: kept in registers from the wave's own integration, closed in a later shadow
  int bk_k = -1;
  float bk_gv = 0.f, bk_eps = 0.f, bk_step = 0.f, bk_sol = 0.f, bk_ssol = 0.f;
  auto books = [&]() {                      // costs of step bk_k (utils.py:92-99); x is x_{k+1} by now
    if (bk_k < 0) return;
    const float uc = lane_ok ? -bk_gv : 0.f;                            // u = -sigma^T nabla_V (method.py:58-80)
    float f = 0.f;                                                      // f at the NEW state, OLD time (utils.py:92-96)
    if (is_quad) {
      float px = 0.f;
      for (int jj = 0; jj < d; ++jj) px += P_l[ic * 16 + jj] * __shfl(x, jj, 16);
      f = row16_sum(lane_ok ? x * px : 0.f);
    } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
      f = 1.f;
    }
    const float uu = row16_sum(uc * uc), ue = row16_sum(uc * bk_eps);
    lpd = lpd + bk_sol * (-f - 0.5f * uu);
    lps = lps + bk_ssol * (-ue);
  };
  auto stores = [&]() {
    if (bk_k < 0) return;
    bk_k = -1;
    if (store) {
      if (a.nabla_v) (a.nabla_v + kbd)[rowoff] = bk_gv;
      (a.controls + kbd)[rowoff] = -bk_gv;
      (a.noises + kbd)[rowoff] = bk_eps;
      (a.states + kbd + step_floats)[rowoff] = x;
    }
    kbd += step_floats;
    if (store0) {
      (a.frac + kb)[grow] = bk_step;
      (a.stop_ind + kb + B)[grow] = STOPPING ? stop : 1.f;
    }
    kb += B;
  };

  // Euler-Maruyama step j from the finished evaluation of x_j (utils.py:37-101): every wave alike (the state lives in all of them)
  auto sde_step = [&](int j) {
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8);
    const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8 + 4);
    const float r0 = lds[LM::res0 + n];
    const f32x4 scal = *reinterpret_cast<const f32x4*>(lds + LM::sc + (j & 31) * 4);
    const float eps = lds[LM::nz + (j % 24) * 16 + i];
    const float dt = scal[0], sq_ldt = scal[1];
    const float gv = relu_keep_nan(r1p_sum8(pa, pb) + *b8p) + r0;
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
    if constexpr (BOOKS) {
      bk_k = j; bk_gv = gv; bk_eps = eps; bk_step = step;
      bk_sol = STOPPING ? step / a.lmbd : scal[2];
      bk_ssol = STOPPING ? sqrtf(bk_sol) : scal[3];
    }
  };

  // ---- noise and scalars, in batches (one wave each, once every eight / sixteen steps: ~150 cycles per step averaged, nothing
  //      on any other step's chain).  Philox draws as documented in include/socmx.h: bit-identical to the other tile shapes. ----
  float* NZ = lds + LM::nz;
  auto noise_batch = [&](int nb) {          // steps 8 nb .. 8 nb + 7: lane = 8 s + pair
    if (w != 6) return;
    const int k = 8 * nb + (lane >> 3), pr = lane & 7, c0 = 2 * pr;
    if (k >= K) return;
    float z0 = 0.f, z1 = 0.f;
    if (a.noise_in) {
      const float* src = a.noise_in + ((size_t)k * B + grow) * d;
      if (c0 < d) z0 = src[c0];
      if (c0 + 1 < d) z1 = src[c0 + 1];
    } else {
      uint32_t wa, wb;
      philox_pair_words(key_seed, key_offset, (uint32_t)(a.row0 + grow), (uint32_t)k, pr >> 1, pr & 1, wa, wb);
      box_muller_pair(wa, wb, z0, z1);
    }
    NZ[(k % 24) * 16 + c0] = c0 < d ? z0 : 0.f;
    NZ[(k % 24) * 16 + c0 + 1] = c0 + 1 < d ? z1 : 0.f;
  };
  auto scalar_batch = [&](int nb) {         // steps 16 nb .. 16 nb + 15: dt (utils.py:38), sqrt(lambda dt) (utils.py:47), dt / lambda and its root
    if (w != 5 || lane >= 16) return;
    const int k = 16 * nb + lane;
    if (k >= K) return;
    const float dt = a.ts[k + 1] - a.ts[k];
    const float dol = dt / a.lmbd;
    *reinterpret_cast<f32x4*>(lds + LM::sc + (k & 31) * 4) = f32x4{dt, sqrtf(a.lmbd * dt), dol, sqrtf(dol)};
  };

  // ---- one evaluation of the network on the current state: five barriers; leaves up_0's partial sums in p5 ----
  auto evaluate = [&](int c) {
    const float t = a.ts[c];
    if ((c & 7) == 0) noise_batch((c >> 3) + 1);
    if ((c & 15) == 1) scalar_batch((c >> 4) + 1);     // (one evaluation later than the slots' last readers: step 16 nb - 1 is integrated at c = 16 nb)
    // P0: r1 of the wave's 32 units -> down_1 partials (blocks 1..4: two k16-groups x two unit registers)
    const float r1v = first_layer(t);
    drift();
    float xs[16];
    {
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      r1p_gather<0>(xs, r1v);
      W.template unit<1, 2>(A0, A1, xs);
      r1p_gather<16>(xs, r1v);
      W.template unit<3, 4>(A0, A1, xs);
      lds[LM::p1 + lane * 8 + w] = A0.x + A0.y;
      lds[LM::p1 + (64 + lane) * 8 + w] = A1.x + A1.y;
    }
    R1P_TICK(0)
    __syncthreads();
    R1P_TICK(1)
    // P1: r2 = relu(sum of down_1's partials + b), positions 16 w .. 16 w + 15 (every row alike) -> down_2 partials (block 7);
    //     res_1's unit registers 0, 1 (blocks 5, 6 in the shadow of the partial-sum reads; 8, 9 behind)
    float y2;
    {
      const int pos = 16 * w + n;
      const f32x4 va = *reinterpret_cast<const f32x4*>(lds + LM::p1 + pos * 8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(lds + LM::p1 + pos * 8 + 4);
      const float bias = BL[u.L[1].b_lds + pos];
      f32x2 R0 = {0.f, 0.f}, R1 = {0.f, 0.f};
      r1p_gather<0>(xs, r1v);
      W.template unit<5, 6>(R0, R1, xs);
      if constexpr (BOOKS) {                // (the finished step's costs and the coming evaluation's res_0: the same shadow)
        books();
        if (c > 0) res0_of(t);
      }
      y2 = relu_keep_nan(r1p_sum8(va, vb) + bias);
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      float x2[16];
      r1p_gather<0>(x2, y2);
      W.template unit<7, -1>(A0, A1, x2);
      lds[LM::p2 + lane * 8 + w] = (A0.x + A0.y) + (A1.x + A1.y);
      r1p_gather<16>(xs, r1v);
      W.template unit<8, 9>(R0, R1, xs);
      lds[LM::pr1 + lane * 8 + w] = R0.x + R0.y;
      lds[LM::pr1 + (64 + lane) * 8 + w] = R1.x + R1.y;
    }
    R1P_TICK(2)
    __syncthreads();
    R1P_TICK(3)
    // P2: r3 = relu(sum of down_2's partials + b), positions 8 w .. 8 w + 7 -> up_2 partials (block 12); res_2 in the shadow (10, 11)
    {
      const int pos = 8 * w + (lane & 7);
      const f32x4 va = *reinterpret_cast<const f32x4*>(lds + LM::p2 + pos * 8);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(lds + LM::p2 + pos * 8 + 4);
      const float bias = BL[u.L[2].b_lds + pos];
      f32x2 R0 = {0.f, 0.f}, R1 = {0.f, 0.f};
      r1p_gather<0>(xs, y2);
      W.template unit<10, 11>(R0, R1, xs);
      lds[LM::pr2 + lane * 8 + w] = R0.x + R0.y;
      lds[LM::pr2 + (64 + lane) * 8 + w] = R1.x + R1.y;
      const float y3 = relu_keep_nan(r1p_sum8(va, vb) + bias);
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f};
      r1p_gather_n<8>(xs, y3);
      W.template unit8<12>(A0, A1, xs);
      lds[LM::p3 + lane * 8 + w] = A0.x + A0.y;
      lds[LM::p3 + (64 + lane) * 8 + w] = A1.x + A1.y;
    }
    R1P_TICK(4)
    __syncthreads();
    R1P_TICK(5)
    // P3: o2 = relu(sum of up_2's partials + b) + sum of res_2's partials + b, positions 16 w .. -> up_1 partials (15..18);
    //     res_1's unit registers 2, 3 (13, 14 in the shadow; 19, 20 behind)
    {
      const int pos = 16 * w + n;
      const f32x4 ua = *reinterpret_cast<const f32x4*>(lds + LM::p3 + pos * 8);
      const f32x4 ub = *reinterpret_cast<const f32x4*>(lds + LM::p3 + pos * 8 + 4);
      const f32x4 ra = *reinterpret_cast<const f32x4*>(lds + LM::pr2 + pos * 8);
      const f32x4 rb = *reinterpret_cast<const f32x4*>(lds + LM::pr2 + pos * 8 + 4);
      const float bu = BL[u.L[6].b_lds + pos], br = BL[u.L[5].b_lds + pos];
      f32x2 R0 = {0.f, 0.f}, R1 = {0.f, 0.f};
      r1p_gather<0>(xs, r1v);
      W.template unit<13, 14>(R0, R1, xs);
      const float y = relu_keep_nan(r1p_sum8(ua, ub) + bu) + (r1p_sum8(ra, rb) + br);
      f32x2 A0 = {0.f, 0.f}, A1 = {0.f, 0.f}, A2 = {0.f, 0.f}, A3 = {0.f, 0.f};
      float xo[16];
      r1p_gather<0>(xo, y);
      W.template unit<15, 16>(A0, A1, xo);
      W.template unit<17, 18>(A2, A3, xo);
      lds[LM::p4 + lane * 8 + w] = A0.x + A0.y;
      lds[LM::p4 + (64 + lane) * 8 + w] = A1.x + A1.y;
      lds[LM::p4 + (128 + lane) * 8 + w] = A2.x + A2.y;
      lds[LM::p4 + (192 + lane) * 8 + w] = A3.x + A3.y;
      r1p_gather<16>(xs, r1v);
      W.template unit<19, 20>(R0, R1, xs);
      lds[LM::pr1 + (128 + lane) * 8 + w] = R0.x + R0.y;
      lds[LM::pr1 + (192 + lane) * 8 + w] = R1.x + R1.y;
    }
    R1P_TICK(6)
    __syncthreads();
    R1P_TICK(7)
    // P4: o1 = relu(sum of up_1's partials + b) + sum of res_1's partials + b for the wave's 32 units, straight into the DPP
    // form's activation layout (lane (g, p = 4 f + i), f < 2, holds unit 32 w + 16 f + 4 g + i) -> this wave's share of up_0
    {
      const int f = (lane >> 2) & 1, pos = 32 * w + 16 * f + 4 * (lane >> 4) + (lane & 3);
      const f32x4 ua = *reinterpret_cast<const f32x4*>(lds + LM::p4 + pos * 8);
      const f32x4 ub = *reinterpret_cast<const f32x4*>(lds + LM::p4 + pos * 8 + 4);
      const f32x4 ra = *reinterpret_cast<const f32x4*>(lds + LM::pr1 + pos * 8);
      const f32x4 rb = *reinterpret_cast<const f32x4*>(lds + LM::pr1 + pos * 8 + 4);
      const float bu = BL[u.L[7].b_lds + pos], br = BL[u.L[4].b_lds + pos];
      if constexpr (BOOKS) stores();        // (behind B4: the next stream block this wave waits for is a phase away)
      const float o1 = relu_keep_nan(r1p_sum8(ua, ub) + bu) + (r1p_sum8(ra, rb) + br);
      float p0 = 0.f, p1 = 0.f;
      r1_fmac8<0>(p0, p1, o1, &w8[0]);
      const float y = r1_rows_sum(p0 + p1);
      if (lane < 16) lds[LM::p5 + n * 8 + w] = y;
    }
    R1P_TICK(8)
    __syncthreads();
    R1P_TICK(9)
  };

  // ---- prologue ----
  if (w == 0) {
    for (int e = lane; e < 256; e += 64) {
      if (is_ou) A_l[e] = ((e & 15) < d && (e >> 4) < d) ? a.A[(e & 15) * d + (e >> 4)] : 0.f;
      if (is_quad) P_l[e] = ((e & 15) < d && (e >> 4) < d) ? a.P[(e >> 4) * d + (e & 15)] : 0.f;
    }
  }
  noise_batch(0);
  scalar_batch(0);
  if constexpr (BOOKS) res0_of(a.ts[0]);
  __syncthreads();
  if (w == 0) rollout_key_advance(a, key_offset);        // (every wave read the key in front of this barrier)
  __builtin_amdgcn_s_setprio(1);
  R1P_PROF_START
  for (int c = 0; c < K; ++c) {
    if (c > 0) sde_step(c - 1);
    R1P_TICK(10)
    evaluate(c);
  }
  R1P_PROF_END(w)
  sde_step(K - 1);
  if (a.nabla_v) {                       // nabla_V(T, X_K) (method.py:272-278 evaluates it on every grid point)
    evaluate(K);
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8);
    const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8 + 4);
    const float gv = relu_keep_nan(r1p_sum8(pa, pb) + *b8p) + lds[LM::res0 + n];
    if (store) a.nabla_v[(size_t)K * B * d + rowoff] = gv;
  }
  if constexpr (BOOKS) {
    books();
    stores();
    float gval = 0.f;                                                   // terminal cost (utils.py:101)
    if (kind == SOCMX_OU_QUADRATIC) {
      float qx = 0.f;
      for (int jj = 0; jj < d; ++jj) qx += a.Q[ic * d + jj] * __shfl(x, jj, 16);
      gval = row16_sum(lane_ok ? x * qx : 0.f);
    } else if (kind == SOCMX_OU_LINEAR) {
      gval = row16_sum(lane_ok ? a.omega[ic] * x : 0.f);
    } else if (kind == SOCMX_DOUBLE_WELL) {
      const float qq = x * x - 1.f;
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
  if (SOCMX_R1P_ONLY_ROLE == 0) r1p_wave<MODE, DMAX, false>(a, lds, wave & 3, lane);
  else r1p_wave<MODE, DMAX, true>(a, lds, 7, lane);
#else
  if (wave < 7) r1p_wave<MODE, DMAX, false>(a, lds, wave, lane);
  else r1p_wave<MODE, DMAX, true>(a, lds, wave, lane);
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
  const bool is_res0 = blk == kR1pWaves * kR1pWaveBlocks;
  const int wave = is_res0 ? 7 : blk / kR1pWaveBlocks, b = blk - wave * kR1pWaveBlocks;
  const R1pBlk Bk = is_res0 ? R1pBlk{3, 0, 0, R1P_STATE} : r1p_block(wave, b);
  int unit, k;
  bool ok = true;
  if (Bk.form == R1P_PK) {
    unit = 64 * Bk.j + lane;
    k = 16 * Bk.kg + 4 * c + e;
  } else if (Bk.form == R1P_PK8) {           // up_2 on inputs 8 w .. 8 w + 7: pieces 0, 1 unit register 0, pieces 2, 3 register 1
    unit = 64 * (c >> 1) + lane;
    k = 8 * wave + 4 * (c & 1) + e;
  } else if (Bk.form == R1P_STATE) {
    unit = Bk.layer == 3 ? (lane & 15) : 32 * wave + (lane & 31);     // res_0: sixteen units, every row alike
    k = 4 * c + e;                                                    // input p of [t, x_0 .. x_14]
  } else {                                   // the standard fragments (0, kc = 2 w + c), c < 2, component e
    unit = lane & 15;
    k = 16 * (2 * wave + c) + 4 * (lane >> 4) + e;
    ok = c < 2;
  }
  float v = 0.f;
  if (ok && unit < a.fout[Bk.layer] && k < a.fin[Bk.layer]) v = a.w[Bk.layer][(size_t)unit * a.fin[Bk.layer] + k];
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
