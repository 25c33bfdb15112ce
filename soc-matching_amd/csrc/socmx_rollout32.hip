// socmx_rollout32.hip -- the evaluation-burst form of the fused Euler-Maruyama rollout: TWO 16-row tiles per workgroup.
//
// Replaces the same reference code as socmx_rollout.hip (SOC_matching/utils.py:17-128 with the per-step control evaluation,
// method.py:58-80, models.py:233-242) for the launches utils.py:131-231 issue (control_objective / normalization_constant:
// 65,536 rows and more per call): d <= 15 (sigma = I, or dense without a stopping time), the constexpr-specialised default
// widths, more 16-row tiles than CUs.
//
// Why.  On this chip an fp32 MFMA does not overlap with anything else the SIMD issues: VALU instructions between the MFMAs
// of a wave cost their full time (tools/ubench/mfma_valu_mix.hip: two v_fma_f32 per v_mfma_f32_16x16x4_f32 turn 32 cycles
// into 53), and so does every KiB written into the register file by a fragment load or a ds_read_b128.  The one-tile kernel
// ran at 30,700 cycles per step against an MFMA issue floor of 21,600 (0.69 of peak) -- and removing its barriers, or putting
// two 4-wave workgroups on a CU to cover them, changed almost nothing: the loss is instructions and bytes per MFMA.
//  * Two tiles per workgroup: every weight fragment feeds EIGHT MFMAs (half the weight bytes into VGPRs per MAC), the
//    per-stage address / bias / ReLU / prefetch instructions are paid once per 32 rows, and consecutive MFMAs of a one-block
//    stage alternate between the two tiles' accumulators.                                       40.9 -> 37.3 ms (65,536 rows)
//  * Weight fragments by BUFFER loads (one 32-bit lane offset register, the fragment's position in the scalar offset)
//    instead of global loads with a 64-bit address per lane.                                      36.4 -> 34.1 ms
//  * ReLU as one integer max per value; GEMM 1's register ring IS the prefetched fragments (no copies); no activation or
//    fragment read past a GEMM's last chunk; down_0's two fragments and stage 1's ring are requested a whole SDE step
//    ahead.                                                                                        37.0 -> 36.2 ms
//  * Noise: the step's 32 x 8 Box-Muller pairs are drawn ONCE each by the four waves that have no block of down_2, instead of
//    one Philox block + one pair per component on every thread (2.9x the instructions).           37.3 -> 37.0 ms
// Tried and measured slower: stages 4 + 5 fused through the accumulators (o1 never in LDS, six barriers), requesting the next
// stage's ring slot by slot as GEMM 1 drains (B32_ROLL), two co-resident 4-wave workgroups per CU (+3 % only).
// Per tile the same MFMAs in the same order, the same split-K combine, the same SDE-step arithmetic and noise counters as
// the 16-row kernel: bit-identical 8-tuples (tests/test_gpu_parity.py::test_two_tile_burst_rollout_equals_the_16_row_kernel).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/socmx.h"
#include "socmx_unet.h"
#include "socmx_launch.h"
#include "socmx_philox.h"
#include "socmx_rollout_common.h"

namespace socmx {

constexpr int kB32Waves = 8;
#ifndef B32_ROLL
#define B32_ROLL 0
#endif
constexpr bool kB32Roll = B32_ROLL != 0;   // request the next stage's ring slot by slot as GEMM 1 drains (1) or in one go at the stage's end (0)

// Weight fragments in flight between stages.  Every request is issued at least one GEMM ahead of its first use: at full
// chip an L2 round trip is ~1,000 cycles, and the hand-over of the 16-row stages (eight fragments requested right before a
// stage's closing barrier) left most of it in front of the next stage's first MFMA -- 4 % of the step, measured by pointing
// those requests at a cache-hot line.  Here a ring slot is re-requested the moment its GEMM has no further use for it:
//   ring   the eight-fragment ring of the GEMM 1 under way; as its last chunks drain, slot by slot the first ring of the
//          stage that follows (stage 3: of stage 4 as soon as GEMM 1 ends, landing under GEMM 2)
//   first  down_0's two fragments (blocks wave, wave + 8), requested while the previous step's last stage multiplies
struct Frags2 {
  f32x4 ring[8];
  f32x4 first[4];       // (down_0: two blocks per wave x one or two 16-input chunks)
};

// The tile layout of socmx_unet.h with the split-K scratch cut to what the last stage writes here (eight waves' partial tiles of
// up_0 + up to four of res_0): two tiles of the 17 <= d <= 31 network would not fit 160 KiB with the full-size scratch.
constexpr int kB32ScratchFloats = 12 * 256;
template <class NET>
__host__ __device__ constexpr TileLayout b32_layout() {
  TileLayout t = NET::layout(kB32Waves);
  t.bias = t.scratch + kB32ScratchFloats;
  t.floats = t.bias + NET::desc().bias_floats;
  return t;
}

// The packed weight image as a buffer resource: a fragment load is buffer_load_dwordx4 with the lane's 16-byte slot in ONE
// offset VGPR (the same for every load) and the fragment's position in the scalar offset -- no 64-bit per-lane addresses.
struct B32Img {
  __amdgpu_buffer_rsrc_t rsrc;
  int lane_off;
};
__device__ __forceinline__ f32x4 b32_frag(const B32Img& im, int float_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(im.rsrc, im.lane_off, float_off * 4, 0));
}

// one wave's GEMM over both tiles: float offset of chunk 0 of each of its blocks in the image, and its lane's activation row
template <int NB>
struct Plan2 {
  int woff[NB];
  const float* xrow;
};
template <int NB>
__device__ __forceinline__ Plan2<NB> make_plan2(const LayerDesc& L, int blk0, int bstride, const float* X, int S, int lane) {
  Plan2<NB> p;
  const int KC = L.in_pad >> 4;
#pragma unroll
  for (int j = 0; j < NB; ++j) p.woff[j] = L.w_off + (blk0 + j * bstride) * KC * 256;
  p.xrow = X + (lane & 15) * S + 4 * (lane >> 4);
  return p;
}

template <class NET, int SI>
struct B32Stage {
  static constexpr int NW = kB32Waves;
  static constexpr UnetDesc u = NET::desc();
  static constexpr TileLayout t = b32_layout<NET>();
  static constexpr StageDesc sd = unet_stage_desc(u, t, SI);
  static constexpr int NBLK = sd.L1.out_pad >> 4, KC1 = sd.L1.in_pad >> 4, KC2 = sd.L2.in_pad >> 4;
  static constexpr int NB = NBLK >= NW ? NBLK / NW : 1;      // blocks per active wave
  static constexpr int NACT = NBLK >= NW ? NW : NBLK;        // waves with a block of their own
  static constexpr bool HAS2 = sd.has2 != 0;
};

// flat fragment f of stage SI's first GEMM-1 ring for this wave: chunk f / NB of block wave + (f % NB) NW
// (f is a constant once the caller's loop is unrolled; nothing is requested past the GEMM's last chunk or by a wave without a block)
template <class NET, int SI>
__device__ __forceinline__ void request_fragment(const B32Img& Wp, int wave, int lane, f32x4& dst, int f) {
  typedef B32Stage<NET, SI> S;
  const int kc = f / S::NB, j = f % S::NB;
  if (kc < S::KC1 && (S::NBLK >= S::NW || wave < S::NBLK)) dst = b32_frag(Wp, S::sd.L1.w_off + ((wave + j * S::NW) * S::KC1 + kc) * 256);
}

// request number f of what stage SI starts from: its first GEMM-1 ring (stages 1..4), down_0's two fragments (stage 0, into
// `first`), or the split stage's share of up_0 and res_0 (stage 5, ring[0 .. CPW])
template <class NET, int SI>
__device__ __forceinline__ void b32_request(const B32Img& Wp, int wave, int lane, Frags2& fr, int f) {
  typedef B32Stage<NET, SI> S;
  if constexpr (SI == 0) {
    if (f < S::NB * S::KC1) request_fragment<NET, 0>(Wp, wave, lane, fr.first[f], f);
  } else if constexpr (SI == 5) {
    // wave -> (block wave % NBLK, part wave / NBLK) of up_0: chunks part CPW .. + CPW - 1; unit `wave` of res_0 = (block
    // wave % NBLK, chunk wave / NBLK) while there are units
    constexpr int PARTS = S::NW / S::NBLK, CPW = S::KC1 / PARTS;
    const int blk = wave % S::NBLK, part = wave / S::NBLK;
    if (f < CPW) fr.ring[f] = b32_frag(Wp, S::sd.L1.w_off + (blk * S::KC1 + part * CPW + f) * 256);
    else if (f == CPW && wave < S::NBLK * S::KC2) fr.ring[f] = b32_frag(Wp, S::sd.L2.w_off + (blk * S::KC2 + part) * 256);
  } else {
    request_fragment<NET, SI>(Wp, wave, lane, fr.ring[f], f);
  }
}

// chunk kc of both tiles' GEMM: 4 k-steps x (tile, block) -- consecutive MFMAs hit different accumulators
template <int NB>
__device__ __forceinline__ void mfma_chunk2(f32x4 (&acc)[2][NB], const f32x4* a, const f32x4 bx0, const f32x4 bx1) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], bx0[i], acc[0][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], bx1[i], acc[1][j], 0, 0, 0);
  }
}

// ReLU of an accumulator quad as ONE integer instruction per value: max(bits, 0) keeps every float with a clear sign bit
// (+0, positive numbers, +inf, NaNs of positive sign) and sends every one with the sign bit set to +0.  fp32 MFMAs and VALU
// instructions do not overlap on this chip (tools/ubench/mfma_valu_mix.hip), so the compare + select pair of
// relu4_keep_nan (socmx_unet.h) is 16 more instructions per stage on the matrix pipe's time.  Differs from it only for -0.0
// (+0.0 here; never the sum of a bias and products in practice) and for NaNs whose sign bit is set.  The compiler sees the
// MFMA -> VALU dependency and pads it once per stage (the inline-asm form pays its eleven wait states per quad).
__device__ __forceinline__ void relu4_imax(f32x4& v) {
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __int_as_float(max(__float_as_int(v[e]), 0));
}

// One GEMM over both tiles (TS floats apart) from an eight-fragment register ring: slot s = chunk % PD, block j at
// ring[s * NB + j]; KC chunks, fully unrolled; activations (the MFMA's B operand) are read from LDS two chunks ahead.
// dead(s): called where slot s has been consumed for the last time (or straight away for slots the GEMM never uses) -- the
// caller requests whatever comes next into it.
template <int NB, int KC, int TS, typename Dead>
__device__ __forceinline__ void gemm2_run(const B32Img& im, f32x4 (&acc)[2][NB], f32x4 (&ring)[8], const Plan2<NB>& p, Dead dead) {
  constexpr int PD = 8 / NB;
  static_assert(PD * NB == 8 && (KC < PD || KC % PD == 0), "ring of eight fragments");
  constexpr int last = KC - 1;
#pragma unroll
  for (int s = KC; s < PD; ++s) dead(s);
  // (nothing is read past the GEMM's last chunk: every ds_read_b128 is a KiB written into the register file on the matrix
  //  pipe's time, see the header)
  auto ld = [&](int tile, int kc) { return *reinterpret_cast<const f32x4*>(p.xrow + tile * TS + kc * 16); };
  f32x4 bx[2] = {ld(0, 0), ld(1, 0)}, bxn[2] = {bx[0], bx[1]};
  if (KC > 1) { bxn[0] = ld(0, 1); bxn[1] = ld(1, 1); }
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const int s = kc % PD;
    f32x4 bxn2[2] = {bxn[0], bxn[1]};
    if (kc + 2 <= last) { bxn2[0] = ld(0, kc + 2); bxn2[1] = ld(1, kc + 2); }
    mfma_chunk2<NB>(acc, &ring[s * NB], bx[0], bx[1]);
    // (the scheduling barriers keep the loads HERE: left alone, the scheduler sinks them towards their use)
    __builtin_amdgcn_sched_barrier(0);
    if (kc + PD <= last) {
#pragma unroll
      for (int j = 0; j < NB; ++j) ring[s * NB + j] = b32_frag(im, p.woff[j] + (kc + PD) * 256);
    } else {
      dead(s);
    }
    __builtin_amdgcn_sched_barrier(0);
    bx[0] = bxn[0]; bx[1] = bxn[1];
    bxn[0] = bxn2[0]; bxn[1] = bxn2[1];
  }
}

#define B32_SETTLE(fr)                                                                                       \
  do {                                                                                                       \
    asm volatile("" : "+v"((fr).ring[0]), "+v"((fr).ring[1]), "+v"((fr).ring[2]), "+v"((fr).ring[3]));       \
    asm volatile("" : "+v"((fr).ring[4]), "+v"((fr).ring[5]), "+v"((fr).ring[6]), "+v"((fr).ring[7]));       \
  } while (0)

// stage 0: r1 = relu(down_0 [t, x]) -- one or two chunks, two blocks per wave, from the `first` fragments (chunk kc, block j at
// first[kc * 2 + j]; the ring already holds stage 1's)
template <class NET, int TS>
__device__ __forceinline__ void b32_stage0(float* lds, Frags2& fr, int wave) {
  typedef B32Stage<NET, 0> S;
  static_assert(S::NB == 2 && S::KC1 <= 2 && !S::HAS2, "down_0: at most two chunks, two blocks per wave");
  const int lane = threadIdx.x & 63, row = lane & 15, g = lane >> 4;
  asm volatile("" : "+v"(fr.first[0]), "+v"(fr.first[1]), "+v"(fr.first[2]), "+v"(fr.first[3]));
  const float* bias_lds = lds + S::t.bias;
  f32x4 acc[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    acc[0][j] = *reinterpret_cast<const f32x4*>(bias_lds + S::sd.L1.b_lds + (wave + j * S::NW) * 16 + 4 * g);
    acc[1][j] = acc[0][j];
  }
  const float* xrow = lds + S::sd.x1 + row * S::sd.s1 + 4 * g;
#pragma unroll
  for (int kc = 0; kc < S::KC1; ++kc)
    mfma_chunk2<2>(acc, &fr.first[kc * 2], *reinterpret_cast<const f32x4*>(xrow + kc * 16), *reinterpret_cast<const f32x4*>(xrow + TS + kc * 16));
#pragma unroll
  for (int tl = 0; tl < 2; ++tl)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      relu4_imax(acc[tl][j]);
      *reinterpret_cast<f32x4*>(lds + tl * TS + S::sd.y + row * S::sd.sy + (wave + j * S::NW) * 16 + 4 * g) = acc[tl][j];
    }
  __syncthreads();
}

// stages 1, 2, 3: Y = relu(L1 X1 + b1) [+ L2 X2 + b2] on both tiles, one block per active wave; the ring rolls over into
// stage NEXT's first ring.  idle(): what the waves without a block do meanwhile.  Ends with a workgroup barrier.
template <class NET, int SI, int NEXT, int TS, bool ROLL, typename Idle>
__device__ __forceinline__ void b32_stage(const B32Img& Wp, float* lds, Frags2& fr, int wave, Idle idle) {
  typedef B32Stage<NET, SI> S;
  constexpr int NB = S::NB;
  const int lane = threadIdx.x & 63, row = lane & 15, g = lane >> 4;
  B32_SETTLE(fr);
  if (wave < S::NACT) {
    const float* bias_lds = lds + S::t.bias;
    const Plan2<NB> p1 = make_plan2<NB>(S::sd.L1, wave, S::NW, lds + S::sd.x1, S::sd.s1, lane);
    const Plan2<NB> p2 = make_plan2<NB>(S::sd.L2, wave, S::NW, lds + S::sd.x2, S::sd.s2, lane);
    f32x4 r2[8];
    if (S::HAS2) {                             // the residual GEMM's first chunks fly while GEMM 1 runs
#pragma unroll
      for (int s = 0; s < 8 / NB; ++s)
#pragma unroll
        for (int j = 0; j < NB; ++j) r2[s * NB + j] = b32_frag(Wp, p2.woff[j] + (s < S::KC2 ? s : S::KC2 - 1) * 256);
    }
    f32x4 acc[2][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc[0][j] = *reinterpret_cast<const f32x4*>(bias_lds + S::sd.L1.b_lds + (wave + j * S::NW) * 16 + 4 * g);
      acc[1][j] = acc[0][j];
    }
    gemm2_run<NB, S::KC1, TS>(Wp, acc, fr.ring, p1, [&](int s) {
      if (ROLL) {
#pragma unroll
        for (int j = 0; j < NB; ++j) b32_request<NET, NEXT>(Wp, wave, lane, fr, s * NB + j);
      }
    });
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int j = 0; j < NB; ++j) relu4_imax(acc[tl][j]);
    if (S::HAS2) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(bias_lds + S::sd.L2.b_lds + (wave + j * S::NW) * 16 + 4 * g);
        acc[0][j] += b2;
        acc[1][j] += b2;
      }
      gemm2_run<NB, S::KC2, TS>(Wp, acc, r2, p2, [](int) {});
    }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        *reinterpret_cast<f32x4*>(lds + tl * TS + S::sd.y + row * S::sd.sy + (wave + j * S::NW) * 16 + 4 * g) = acc[tl][j];
    if (!ROLL) {
#pragma unroll
      for (int f = 0; f < 8; ++f) b32_request<NET, NEXT>(Wp, wave, lane, fr, f);
    }
  } else {
#pragma unroll
    for (int f = 0; f < 8; ++f) b32_request<NET, NEXT>(Wp, wave, lane, fr, f);
    idle();
  }
  __syncthreads();
}

// stage 5: nabla_V = relu(up_0 o1) + res_0 [t, x], one or two 16-wide blocks: wave w multiplies part w / NBLK of block w % NBLK of
// up_0 (CPW consecutive chunks; with one block: chunks 2w, 2w + 1 -- the 16-row kernel's split) and, while there are units, chunk
// w / NBLK of block w % NBLK of res_0, from the fragments in ring[0 .. CPW]; partial sums through the tile's scratch, and after
// the barrier thread (tile, e) adds up element e of each block of its tile in the order of unet_stage_static.
// Requests down_0's fragments and stage 1's whole first ring for the step that follows: the ring is idle until then.
template <class NET, int TS>
__device__ __forceinline__ void b32_stage5(const B32Img& Wp, float* lds, Frags2& fr, int wave, float* to_reg) {
  typedef B32Stage<NET, 5> S;
  constexpr int NW = kB32Waves, NBLK = S::NBLK, PARTS = NW / NBLK, CPW = S::KC1 / PARTS, UNITS2 = NBLK * S::KC2;
  static_assert(NBLK <= 2 && S::KC1 % PARTS == 0 && CPW + 1 <= 8 && S::HAS2 && UNITS2 <= NW &&
                (NW + UNITS2) * 256 <= kB32ScratchFloats, "one or two blocks split over the waves");
  const int lane = threadIdx.x & 63, row = lane & 15, g = lane >> 4;
  const int blk = wave % NBLK, part = wave / NBLK;
  B32_SETTLE(fr);
  {
    const float* xrow = lds + S::sd.x1 + row * S::sd.s1 + 4 * g + part * (CPW * 16);
    f32x4 acc[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
    for (int f = 0; f < CPW; ++f)
      mfma_chunk2<1>(acc, &fr.ring[f], *reinterpret_cast<const f32x4*>(xrow + f * 16), *reinterpret_cast<const f32x4*>(xrow + TS + f * 16));
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
      *reinterpret_cast<f32x4*>(lds + tl * TS + S::t.scratch + ((blk * PARTS + part) * 16 + row) * 16 + 4 * g) = acc[tl][0];
  }
  if (wave < UNITS2) {                                 // (unit = (block blk, chunk part) of res_0)
    const float* xrow = lds + S::sd.x2 + row * S::sd.s2 + 4 * g + part * 16;
    f32x4 acc[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
    mfma_chunk2<1>(acc, &fr.ring[CPW], *reinterpret_cast<const f32x4*>(xrow), *reinterpret_cast<const f32x4*>(xrow + TS));
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
      *reinterpret_cast<f32x4*>(lds + tl * TS + S::t.scratch + NW * 256 + ((blk * S::KC2 + part) * 16 + row) * 16 + 4 * g) = acc[tl][0];
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) b32_request<NET, 0>(Wp, wave, lane, fr, f);
#pragma unroll
  for (int f = 0; f < 8; ++f) b32_request<NET, 1>(Wp, wave, lane, fr, f);
  __syncthreads();
  const float* bias_lds = lds + S::t.bias;
  const int e = threadIdx.x & 255, tile = threadIdx.x >> 8;
  const float* P1 = lds + tile * TS + S::t.scratch;
  const float* P2 = P1 + NW * 256;
  const int n = e & 15;
#pragma unroll
  for (int b = 0; b < NBLK; ++b) {
    float v = bias_lds[S::sd.L1.b_lds + b * 16 + n];
#pragma unroll
    for (int p = 0; p < PARTS; ++p) v += P1[(b * PARTS + p) * 256 + e];
    v = relu_keep_nan(v);
    float v2 = bias_lds[S::sd.L2.b_lds + b * 16 + n];
#pragma unroll
    for (int p = 0; p < S::KC2; ++p) v2 += P2[(b * S::KC2 + p) * 256 + e];
    to_reg[b] = v + v2;
  }
}

// the fragments the first stages of the first step start from
template <class NET>
__device__ __forceinline__ void b32_frags_init(const B32Img& Wp, Frags2& fr, int wave) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int f = 0; f < 8; ++f) fr.ring[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 4; ++f) fr.first[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < 4; ++f) b32_request<NET, 0>(Wp, wave, lane, fr, f);
#pragma unroll
  for (int f = 0; f < 8; ++f) request_fragment<NET, 1>(Wp, wave, lane, fr.ring[f], f);
}

// the whole network on both tiles: X0 (filled, [t, x, 0-pad]) -> thread (tile, e)'s element of nabla_V in *gv_reg.
// idle2: run by the upper four waves while the lower four multiply down_2's four blocks.  hook(i): after stage i (profiling).
template <class NET, int TS, typename Hook, typename Idle>
__device__ __forceinline__ void unet_forward2(const B32Img& Wp, float* lds, Frags2& fr, float* gv_reg, Hook hook, Idle idle2) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto none = []() {};
  b32_stage0<NET, TS>(lds, fr, wave); hook(1);
  b32_stage<NET, 1, 2, TS, kB32Roll>(Wp, lds, fr, wave, none); hook(2);
  b32_stage<NET, 2, 3, TS, kB32Roll>(Wp, lds, fr, wave, idle2); hook(3);
  b32_stage<NET, 3, 4, TS, kB32Roll>(Wp, lds, fr, wave, none); hook(4);
  b32_stage<NET, 4, 5, TS, kB32Roll>(Wp, lds, fr, wave, none); hook(5);
  b32_stage5<NET, TS>(Wp, lds, fr, wave, gv_reg); hook(6);
}

template <int H>                          // H = 16-component halves of a row: 1 (d <= 15) or 2 (17 <= d <= 31)
struct Burst32Lds {                       // float offsets behind the two tiles
  static constexpr int mat = H == 1 ? 256 : 1024;   // a (d, stride) matrix: stride 17 / 33 (socmx_sde_stride)
  static constexpr int A = 0;             // OU drift
  static constexpr int P = mat;           // OU_quadratic running cost
  static constexpr int nz = 2 * mat;      // (32 rows, 16 H): the noise of the step under way
  static constexpr int sig = nz + 512 * H;          // a dense sigma (OU_linear; d <= 15 only)
  static constexpr int floats = sig + (H == 1 ? 256 : 0);
};

// Can the two-tile stages run this architecture?  The shapes the reference's default widths give: down_0 one chunk and two
// blocks per wave; down_1 and up_2 / res_2 one block per wave; down_2 four blocks (the other four waves draw the noise);
// up_1 / res_1 two blocks per wave feeding a single 16-wide block of up_0.  (Architecture-variant builds answer for their own
// widths; what does not fit keeps the 16-row kernel.)
template <class NET>
__host__ __device__ constexpr bool b32_supported() {
  constexpr int NW = kB32Waves;
  typedef B32Stage<NET, 0> S0; typedef B32Stage<NET, 1> S1; typedef B32Stage<NET, 2> S2; typedef B32Stage<NET, 3> S3;
  typedef B32Stage<NET, 4> S4; typedef B32Stage<NET, 5> S5;
  constexpr UnetDesc u = NET::desc();
  constexpr TileLayout t = b32_layout<NET>();
  auto ring_ok = [](int kc, int nb) { return kc < 8 / nb || kc % (8 / nb) == 0; };
  if (!((u.in0p == 16 && u.outp == 16) || (u.in0p == 32 && u.outp == 32))) return false;
  if (!(S0::NBLK == 2 * NW && S0::KC1 <= 2 && !S0::HAS2)) return false;
  if (!(S1::NBLK == NW && !S1::HAS2 && ring_ok(S1::KC1, 1))) return false;
  if (!(S2::NBLK * 2 == NW && !S2::HAS2 && ring_ok(S2::KC1, 1))) return false;
  if (!(S3::NBLK == NW && S3::HAS2 && ring_ok(S3::KC1, 1) && ring_ok(S3::KC2, 1))) return false;
  if (!(S4::NBLK == 2 * NW && S4::HAS2 && ring_ok(S4::KC1, 2) && ring_ok(S4::KC2, 2))) return false;
  if (!(S5::NBLK <= 2 && S5::HAS2 && S5::KC1 % (NW / S5::NBLK) == 0 && S5::KC1 / (NW / S5::NBLK) + 1 <= 8 &&
        S5::NBLK * S5::KC2 <= NW && (NW + S5::NBLK * S5::KC2) * 256 <= kB32ScratchFloats))
    return false;
  return ((((t.floats + 3) & ~3) + t.bias + Burst32Lds<(NET::outp >> 4)>::floats) * 4 <= 160 * 1024);
}

// thread (tile = tid >> 8, r = (tid >> 4) & 15, i = tid & 15): component i of row r of that tile -- the whole SDE step of a
// row lives in one 16-lane group (state, control, noise, update and costs in registers, row sums by DPP), exactly as in the
// FAST path of rollout_kernel (socmx_rollout.hip).
// Noise: fp32 MFMAs and VALU instructions do not overlap on this chip (tools/ubench/mfma_valu_mix.hip: two VALU instructions
// between the MFMAs of a wave cost 21 cycles of matrix time), so the generator's instruction count is what matters: the step's
// 32 x 8 Box-Muller pairs are drawn ONCE each -- thread p of waves 4..7 takes pair p & 7 of row p >> 3 -- while those waves
// have no block of down_2 to multiply, and reach the SDE step through LDS (one thread per component drew a whole Philox block
// and a pair for one value before: 2.9x the instructions).  Same counters, same arithmetic: the values are bit-identical.
// PROF (developer builds of the phase table, socmx_rollout_phase_cycles_f32): wave a.prof_wave's s_memtime per phase -- slot 0 the
// barrier that opens a step, 1..6 the network stages (3: with the noise draw of waves 4..7), 7 the SDE step
template <bool STOPPING, class NET, bool PROF>
__global__ __launch_bounds__(kB32Waves * 64) void rollout32_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = kB32Waves;
  constexpr TileLayout tl = b32_layout<NET>();
  constexpr UnetDesc ud = NET::desc();
  constexpr int TS = (tl.floats + 3) & ~3;
  // H: 16-component halves of a row.  17 <= d <= 31 (the 32-wide network input / output): thread (row, i) carries components i
  // and 16 + i, so that a row is still one 16-lane group (row sums by DPP) -- every per-component statement below loops over h.
  constexpr int H = NET::outp >> 4, CW = 16 * H;
  typedef Burst32Lds<H> LM;
  static_assert((NET::outp == 16 && NET::in0p == 16) || (NET::outp == 32 && NET::in0p == 32), "d <= 15 or 17 <= d <= 31");
  static_assert((TS + tl.bias + LM::floats) * 4 <= 160 * 1024, "two tiles in one CU's LDS");
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);
  const int tid = threadIdx.x;
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int ds = socmx_sde_stride(d);
  const bool is_ou = (kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR);
  const bool is_quad = kind == SOCMX_OU_QUADRATIC;
  float* SD = lds + TS + tl.bias;
  float* A_l = SD + LM::A;
  float* P_l = SD + LM::P;
  float* NZ = SD + LM::nz;
  float* SIG = SD + LM::sig;
  const bool dense = H == 1 && !a.sigma_identity;   // u = -sigma^T nabla_V, sigma u, sigma eps: three d x d products per row
  const int tile = tid >> 8, r = (tid >> 4) & 15, i = tid & 15;
  float* X0 = lds + tile * TS + tl.x0;
  for (int e = tid; e < d * d; e += NW * 64) {
    const int rr = e / d, cc = e - rr * d;
    if (is_ou) A_l[rr * ds + cc] = a.A[e];
    if (is_quad) P_l[rr * ds + cc] = a.P[e];
    if (dense) SIG[rr * ds + cc] = a.sigma[e];
  }
  unet_load_biases(a.packed, ud, tl, lds, tid, NW * 64, ud.folded != 0);
  const B32Img img = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, ud.image_floats * 4, 0x00020000), (tid & 63) * 16};
  Frags2 carry;
  b32_frags_init<NET>(img, carry, __builtin_amdgcn_readfirstlane(tid >> 6));

  const int grow = blockIdx.x * 32 + tile * 16 + r;
  const bool traj = a.states != nullptr;        // costs-only launches (evaluation bursts) pass no trajectory buffers
  const bool row_live = grow < B;
  const bool store0 = i == 0 && row_live && traj;
  const size_t rowbase = (size_t)grow * d;
  auto gsum = [](float v) { return row16_sum(v); };
  int comp[H], ic[H];
  bool lane_ok[H], store[H];
  float x[H], kap[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    comp[h] = i + 16 * h;
    ic[h] = min(comp[h], d - 1);
    lane_ok[h] = comp[h] < d;
    store[h] = lane_ok[h] && row_live && traj;
    x[h] = lane_ok[h] ? a.x0[(size_t)min(grow, B - 1) * d + comp[h]] : 0.f;
    kap[h] = (lane_ok[h] && !is_ou) ? a.kappa[comp[h]] : 0.f;
    if (store[h]) a.states[rowbase + comp[h]] = x[h];
  }
  // component j of this row's vector v (any j < d): the 16-lane group's lane j & 15, half j >> 4
  auto comp_of = [&](const float (&v)[H], int j) -> float {
    float t = __shfl(v[0], j & 15, 16);
    if constexpr (H == 2) { const float t1 = __shfl(v[1], j & 15, 16); t = (j & 16) ? t1 : t; }
    return t;
  };
  float stop = 1.f, lpd = 0.f, lps = 0.f;
  if (store0) a.stop_ind[grow] = 1.f;
  auto put_input = [&](float t) {                // the network input [t, x, 0..]: column 1 + component (lanes past d hold x = 0)
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (1 + comp[h] < NET::in0p) X0[r * tl.s0 + 1 + comp[h]] = x[h];
    if (i == 0) X0[r * tl.s0] = t;
  };
  put_input(a.ts[0]);
  __syncthreads();
  rollout_key_advance(a, key_offset);           // (every thread read the key above)
  const bool injected = a.noise_in != nullptr;
  auto produce = [&](int k) {                   // waves 4..7: thread p draws pairs p, p + 256 (H = 2) of the 32 x 8 H (pairs past d are never read)
#pragma unroll
    for (int rep = 0; rep < H; ++rep) {
      const int unit = tid - 256 + 256 * rep, prow = unit / (8 * H), q = unit % (8 * H);
      if (!injected && 2 * q < d) {
        uint32_t wa, wb;
        philox_pair_words(key_seed, key_offset, (uint32_t)(a.row0 + blockIdx.x * 32 + prow), (uint32_t)k, q >> 1, q & 1, wa, wb);
        float z0, z1;
        box_muller_pair(wa, wb, z0, z1);
        *reinterpret_cast<float2*>(NZ + prow * CW + 2 * q) = make_float2(z0, z1);
      }
    }
  };
  long long acc_prof[16];
  if (PROF) for (int sl = 0; sl < 16; ++sl) acc_prof[sl] = 0;
  long long last_tick = PROF ? clock64() : 0;
  auto hook = [&](int slot) {
    if (PROF) {
      const long long now_ = clock64();
      acc_prof[slot] += now_ - last_tick;
      last_tick = now_;
    }
  };
  for (int k = 0; k < K; ++k) {
    const float t0 = a.ts[k], t1 = a.ts[k + 1];
    const float dt = t1 - t0;                 // utils.py:38
    const float sq_ldt = sqrtf(a.lmbd * dt);  // utils.py:47
    __syncthreads();
    hook(0);
    float eps[H];
#pragma unroll
    for (int h = 0; h < H; ++h)
      eps[h] = (injected && lane_ok[h]) ? a.noise_in[((size_t)k * B + min(grow, B - 1)) * d + comp[h]] : 0.f;
    float gv[H];                              // nabla_V[r][comp] of this thread's tile
#pragma unroll
    for (int h = 0; h < H; ++h) gv[h] = 0.f;
    unet_forward2<NET, TS>(img, lds, carry, gv, hook, [&]() { produce(k); });
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (store[h] && a.nabla_v) a.nabla_v[(size_t)k * B * d + rowbase + comp[h]] = gv[h];
    {
      float u[H], su[H], se[H], bi[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (!injected) eps[h] = lane_ok[h] ? NZ[(tile * 16 + r) * CW + comp[h]] : 0.f;
        u[h] = lane_ok[h] ? -gv[h] : 0.f;                               // u = -sigma^T nabla_V (method.py:68-72)
        su[h] = u[h]; se[h] = eps[h];                                   // sigma u, sigma eps (utils.py:45-47)
      }
      if constexpr (H == 1) {
        if (dense) {                                                    // (sums in the order of the 16-row kernel: j ascending)
          float s_ = 0.f;
          for (int j = 0; j < d; ++j) s_ += SIG[j * ds + ic[0]] * __shfl(gv[0], j, 16);
          u[0] = lane_ok[0] ? -s_ : 0.f;
          float a_ = 0.f, b_ = 0.f;
          for (int j = 0; j < d; ++j) {
            a_ += SIG[ic[0] * ds + j] * __shfl(u[0], j, 16);
            b_ += SIG[ic[0] * ds + j] * __shfl(eps[0], j, 16);
          }
          su[0] = lane_ok[0] ? a_ : 0.f;
          se[0] = lane_ok[0] ? b_ : 0.f;
        }
      }
      if (is_ou) {                                                      // b = A x
#pragma unroll
        for (int h = 0; h < H; ++h) bi[h] = 0.f;
        for (int j = 0; j < d; ++j) {
          const float xj = comp_of(x, j);
#pragma unroll
          for (int h = 0; h < H; ++h) bi[h] += A_l[ic[h] * ds + j] * xj;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) bi[h] = lane_ok[h] ? bi[h] : 0.f;
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) bi[h] = -2.f * kap[h] * (x[h] * x[h] - 1.f) * 2.f * x[h];      // double_well.py:44-48
      }
      float upd[H], xn[H], xe[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        upd[h] = (bi[h] + su[h]) * dt + sq_ldt * se[h];                 // utils.py:45-47
        xn[h] = x[h] + stop * upd[h];                                   // utils.py:48
        xe[h] = xn[h];
      }
      float step = dt, stop_new = 1.f;
      if (STOPPING) {                                                   // utils.py:42-44, 49-75; Phi = -x_0
        const float phi_b = -__shfl(x[0], 0, 16), phi_a = -__shfl(xn[0], 0, 16);
        const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
        const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
        const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
#pragma unroll
        for (int h = 0; h < H; ++h) xe[h] = js * (x[h] + fr * stop * upd[h]) + (1.f - js) * xn[h];
        step = js * (fr * fr) * dt + ns * dt;                           // step_fraction squared (utils.py:70-72)
        stop_new = (-__shfl(xe[0], 0, 16) > 0.f) ? 1.f : 0.f;
      }
      float f = 0.f;                                                    // f at the NEW state, OLD time (utils.py:92-96)
      if (kind == SOCMX_OU_QUADRATIC) {
        float px[H];
#pragma unroll
        for (int h = 0; h < H; ++h) px[h] = 0.f;
        for (int j = 0; j < d; ++j) {
          const float xj = comp_of(xe, j);
#pragma unroll
          for (int h = 0; h < H; ++h) px[h] += P_l[ic[h] * ds + j] * xj;
        }
        // (H = 1: the very expressions of the 16-row kernel -- the compiler's multiply-add contraction follows the shape of the
        //  source, and the two kernels are compared bit for bit)
        float part = lane_ok[0] ? xe[0] * px[0] : 0.f;
        if constexpr (H == 2) part += lane_ok[1] ? xe[1] * px[1] : 0.f;
        f = gsum(part);
      } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
        f = 1.f;
      }
      float puu = u[0] * u[0], pue = u[0] * eps[0];
      if constexpr (H == 2) { puu += u[1] * u[1]; pue += u[1] * eps[1]; }
      const float uu = gsum(puu), ue = gsum(pue);
      lpd = lpd + step / a.lmbd * (-f - 0.5f * uu);
      lps = lps + sqrtf(step / a.lmbd) * (-ue);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (store[h]) {
          a.controls[(size_t)k * B * d + rowbase + comp[h]] = u[h];
          a.noises[(size_t)k * B * d + rowbase + comp[h]] = eps[h];
          a.states[(size_t)(k + 1) * B * d + rowbase + comp[h]] = xe[h];
        }
        x[h] = lane_ok[h] ? xe[h] : 0.f;
      }
      if (store0) {
        a.frac[(size_t)k * B + grow] = step;
        a.stop_ind[(size_t)(k + 1) * B + grow] = STOPPING ? stop_new : 1.f;
      }
      if (STOPPING) stop = stop_new;
      put_input(t1);                                                    // next step's network input [t, x, 0..]
    }
    hook(7);
  }
  if (PROF && tid == a.prof_wave * 64 && a.prof)
    for (int sl = 0; sl < 16; ++sl) a.prof[(size_t)blockIdx.x * 64 + sl] = acc_prof[sl];
  if (a.nabla_v) {                        // nabla_V(T, X_K): X0 already holds [t_K, x_K]
    __syncthreads();
    float gv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) gv[h] = 0.f;
    unet_forward2<NET, TS>(img, lds, carry, gv, [](int) {}, []() {});
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (store[h]) a.nabla_v[(size_t)K * B * d + rowbase + comp[h]] = gv[h];
  }
  float gval = 0.f;                       // terminal cost (utils.py:101)
  if (kind == SOCMX_OU_QUADRATIC) {
    float qx[H];
#pragma unroll
    for (int h = 0; h < H; ++h) qx[h] = 0.f;
    for (int j = 0; j < d; ++j) {
      const float xj = comp_of(x, j);
#pragma unroll
      for (int h = 0; h < H; ++h) qx[h] += a.Q[ic[h] * d + j] * xj;
    }
    float part = lane_ok[0] ? x[0] * qx[0] : 0.f;
    if constexpr (H == 2) part += lane_ok[1] ? x[1] * qx[1] : 0.f;
    gval = gsum(part);
  } else if (kind == SOCMX_OU_LINEAR) {
    float part = lane_ok[0] ? a.omega[ic[0]] * x[0] : 0.f;
    if constexpr (H == 2) part += lane_ok[1] ? a.omega[ic[1]] * x[1] : 0.f;
    gval = gsum(part);
  } else if (kind == SOCMX_DOUBLE_WELL) {
    const float q0 = x[0] * x[0] - 1.f;
    float part = lane_ok[0] ? a.nu[ic[0]] * (q0 * q0) : 0.f;
    if constexpr (H == 2) {
      const float q1 = x[1] * x[1] - 1.f;
      part += lane_ok[1] ? a.nu[ic[1]] * (q1 * q1) : 0.f;
    }
    gval = gsum(part);
  }
  if (i == 0 && grow < B) {
    a.lpd[grow] = lpd;
    a.lps[grow] = lps;
    a.ltw[grow] = -gval / a.lmbd;
  }
}

// (hidden, as the one-row kernel's entry points: see socmx_rollout_common.h)
template <class NET>
static int rollout32_launch_t(const RolloutArgs& a, bool stopping, void* stream) {
  if constexpr (b32_supported<NET>()) {
    constexpr TileLayout tl = b32_layout<NET>();
    constexpr int TS = (tl.floats + 3) & ~3;
    const size_t lds_bytes = (size_t)(TS + tl.bias + Burst32Lds<(NET::outp >> 4)>::floats) * sizeof(float);
    void (*k)(const RolloutArgs) = a.prof ? rollout32_kernel<false, NET, true>
                                 : stopping ? rollout32_kernel<true, NET, false> : rollout32_kernel<false, NET, false>;
    if (a.prof && stopping) return SOCMX_E_DIM;       // (the phase table is taken on the settings without a stopping time)
    if (const int err = ensure_max_lds(k)) return err;
    return launch(k, dim3((a.B + 31) / 32), dim3(kB32Waves * 64), lds_bytes, (socmx_stream_t)stream, a);
  } else {
    return SOCMX_E_DIM;
  }
}

// in0p = 16: d <= 15 (sigma = I or dense); in0p = 32: 17 <= d <= 31 with sigma = I
bool rollout32_available(int in0p) { return in0p == 16 ? b32_supported<DefaultNet>() : in0p == 32 ? b32_supported<Wide32Net>() : false; }
int rollout32_launch(const RolloutArgs& a, bool stopping, void* stream) {
  return a.u.in0p == 16 ? rollout32_launch_t<DefaultNet>(a, stopping, stream) : rollout32_launch_t<Wide32Net>(a, stopping, stream);
}

}  // namespace socmx
