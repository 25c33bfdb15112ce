// socmx_rollout1.hip -- the fused Euler-Maruyama rollout with ONE ROW PER WORKGROUP (gfx950): training-size batches.
//
// Replaces reference SOC_matching/utils.py:17-128 (stochastic_trajectories), method.py:58-80 (control) and
// models.py:233-242 (FullyConnectedUNet.forward) for B <= 256 rows, sigma = I, d <= 15 and the default hidden widths -- the
// shape of BASELINE configs[1] / [2] (B = 128) and of the README's molecular_dynamics run (B = 64).
//
// Why one row.  A batch of 128 rows is 8 tiles of 16 rows (8 of 256 CUs busy) or 32 tiles of 4 rows (32 CUs, on an MFMA
// form that runs at half rate); the K-long chain of network evaluations per tile is the whole run time.  With one row per
// workgroup 128 CUs work, and a network evaluation is a chain of MATRIX-VECTOR products: every weight is used once per
// step, no MFMA tile exists.  On the VALU the product is
//     acc[lane (g, n)] += x[k] * W[16 nb + n][k],   k = 16 kc + 4 g + i
//   = v_fmac_f32_dpp acc, xv, w  row_newbcast:(4 f + i)          (64 MACs per instruction, the matrix pipe's 4x4x1 rate)
// with `w` one component of the SAME fragment image the MFMA kernels read (socmx_unet.h: lane (g, n), component i of fragment
// (nb, kc) holds W[16 nb + n][16 kc + 4 g + i]) and `xv` an activation register whose 16-lane row g holds, at position
// 4 f + i, x[64 C + 16 f + 4 g + i] (f = kc & 3: four fragments = one 64-wide chunk C share one activation register; DPP
// row_newbcast:j broadcasts position j of each row to the row's lanes).  The four rows of a wave are four k-groups of the
// same 16 neurons; their partial sums are added across rows when a layer ends (v_permlane32/16_swap).
// What bounds the step is then where the 677 KB of weights come FROM, every step: 64 MACs per cycle and CU need 256 B per
// cycle, which only the register file delivers.  So the image is split three ways per wave (compile-time tables below):
//   RES  blocks live in VGPRs for the whole launch (a block = 16 registers = one (neuron block, chunk) = 4 KiB per wave),
//   LDS  blocks are copied to LDS once and read back every step (ds_read_b128, 128 B/clk),
//   STR  blocks stream from L2 every step through a two-block register ring, requested a ring ahead of their use.
// Wave 0 also integrates: it owns the network's first layer (all 256 units, 11 inputs), the last layer's final sum and the
// Euler-Maruyama step, runs no stream (its global stores share the vmcnt counter) and keeps its big stage in LDS instead.
// tools/ubench/valu_row1.hip measured the pieces (profiles/r4/valu_row1_ubench.txt).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "socmx_rollout_common.h"
#include "socmx_launch.h"
#include "socmx_row1.h"

namespace socmx {

constexpr int kR1Waves = 8;
#ifndef SOCMX_R1_S2DIRECT
#define SOCMX_R1_S2DIRECT 1
#endif
#ifndef SOCMX_R1_RES0_SLACK
#define SOCMX_R1_RES0_SLACK 0
#endif
// stage 2 in the direct layout (round 6; developer A/B: -DSOCMX_R1_S2DIRECT=0).  configs[2]: 0.409 -> 0.386 ms per rollout call, A/B on one box
constexpr bool R1_S2DIRECT = SOCMX_R1_S2DIRECT != 0;
// res_0 [t, x] formed in wave 0's slack behind stage 4 instead of inside first_layer, on the serial section: measured, NO gain
// (0.391 against 0.385 ms: the slack is not free -- a longer low-priority section makes wave 0 the last one at that barrier): off
constexpr bool R1_RES0_SLACK = SOCMX_R1_RES0_SLACK != 0;
#ifndef SOCMX_R1_EX_PARTS
#define SOCMX_R1_EX_PARTS 3     /* developer A/B: 1 = the activation stores, 2 = the sign records */
#endif
constexpr int kR1Blocks = 20;   // main-program blocks per wave and step
constexpr int kR1PD = 2;        // ring depth (blocks)

// Where block b of a wave's program comes from: 'R' registers, 'L' LDS, 'S' L2 stream, 'X' not part of the program.
// Class 0 = wave 0, class 1 = waves 1..7.
//   blocks in the order a wave consumes them: 0-3 down_1 (chunks 0..3), 4 down_2 over the wave's own outputs (always resident) | 5-6 res_2, 7 up_2 |
//   16-19 up_1 (U0c0 U1c0 U0c1 U1c1).  Blocks 8-15 were res_1 (256 x 256: 39 % of the network) until round 5: res_1 r1 reaches the
//   next ReLU only through the linear map up_0, so the kernel multiplies r1 by the FOLDED (outp x 256) matrix up_0 res_1 instead
//   (UnetDesc::fold, formed by the pack kernel) -- eight fmacs per wave and step where there were 128, and with 256 KB of weights
//   gone the remaining twelve blocks per wave fit registers + LDS: NOTHING streams from L2 any more.
__host__ __device__ constexpr char r1_src(int cls, int dmx, int b) {
  // wave 0 also holds 4 (1 + DMAX) + (1 + DMAX) registers of down_0 / res_0: fewer resident blocks at larger d
  // dmx = DMAX (+ 100 for a dense sigma: sigma and sigma sigma^T sit behind the LDS blocks)
  const int dm = dmx % 100;
#ifndef SOCMX_R1_PLAN0_3
#define SOCMX_R1_PLAN0_3 "RLRLRLRLXXXXXXXXLRLR"
#endif
#ifndef SOCMX_R1_PLAN0_11
#define SOCMX_R1_PLAN0_11 "RLRLRLLLXXXXXXXXLRLR"
#endif
#ifndef SOCMX_R1_PLAN0_15
#define SOCMX_R1_PLAN0_15 "RLLLRLLLXXXXXXXXLRLR"
#endif
#ifndef SOCMX_R1_PLAN0_31
#define SOCMX_R1_PLAN0_31 "RLLLRLLLXXXXXXXXLRLR"
#endif
#ifndef SOCMX_R1_PLAN1_31
#define SOCMX_R1_PLAN1_31 "RRRLRRRLXXXXXXXXRLRR"
#endif
// (all twelve blocks of waves 1..7 resident: one register of the sigma = I instantiations spills -- outside the step loop -- and the
//  ~70-cycle stall stage 3 paid for the LDS block's second half is gone: 0.414 -> 0.405 ms at configs[2], A/B on one box)
#ifndef SOCMX_R1_PLAN1
#define SOCMX_R1_PLAN1 "RRRRRRRRXXXXXXXXRRRR"
#endif
  constexpr char plan0_3[kR1Blocks + 1] = SOCMX_R1_PLAN0_3, plan0_11[kR1Blocks + 1] = SOCMX_R1_PLAN0_11, plan0_15[kR1Blocks + 1] = SOCMX_R1_PLAN0_15;
  // 17 <= d <= 31 (dm = 31; the 32-wide input / output network): down_0 runs on all eight waves (two resident fragments pairs per
  // wave) instead of from wave 0's registers; the down_0 fragments, the second up_0 block and the second fold block take registers
  constexpr char plan0_31[kR1Blocks + 1] = SOCMX_R1_PLAN0_31, plan1_31[kR1Blocks + 1] = SOCMX_R1_PLAN1_31;
  constexpr char plan1[kR1Blocks + 1] = SOCMX_R1_PLAN1;     // (developer sweeps: -DSOCMX_R1_PLAN1=...)
  if (dm > 15) return cls == 1 ? plan1_31[b] : plan0_31[b];
  if (cls == 1) return plan1[b];
  // (a dense sigma adds row i of sigma sigma^T to wave 0's registers: at d <= 11 it takes the four-resident-block plan of d <= 15 --
  //  with the five-block one 3-5 registers of the step loop spilled to scratch)
  if (dmx >= 100 && dm > 3) return plan0_15[b];
  return dm <= 3 ? plan0_3[b] : dm <= 11 ? plan0_11[b] : plan0_15[b];
}
__host__ __device__ constexpr int r1_count(int cls, int dm, char s, int upto = kR1Blocks) {
  int n = 0;
  for (int b = 0; b < upto; ++b) n += r1_src(cls, dm, b) == s;
  return n;
}
// block number of a class's i-th block of source s
__host__ __device__ constexpr int r1_nth(int cls, int dm, char s, int i) {
  int n = 0;
  for (int b = 0; b < kR1Blocks; ++b)
    if (r1_src(cls, dm, b) == s) {
      if (n == i) return b;
      ++n;
    }
  return -1;
}
__host__ __device__ constexpr int r1_lds_blocks(int dm) { return r1_count(0, dm, 'L') + 7 * r1_count(1, dm, 'L'); }

template <class NET>
__host__ __device__ constexpr bool r1_supported() {
  constexpr UnetDesc u = NET::desc();
  return ((u.in0p == 16 && u.outp == 16) || (u.in0p == 32 && u.outp == 32)) && u.hp[0] == 256 && u.hp[1] == 128 && u.hp[2] == 64;
}

// float offset of block b of wave w inside the packed image (fragments (nb, 4C .. 4C+3) of a layer are 4 KiB contiguous)
template <class NET>
__device__ __forceinline__ int r1_block_off(int b, int w) {
  constexpr UnetDesc u = NET::desc();
  if (b < 4) return u.L[1].w_off + (w * 16 + 4 * b) * 256;                                  // down_1: nb = w, chunk b
  if (b == 4) return u.L[2].w_off + w * 256;      // down_2: unit blocks 0..3, input chunk w (the wave's own stage-1 outputs): r1_frag_stride
  if (b < 7) return u.L[5].w_off + (w * 8 + 4 * (b - 5)) * 256;                             // res_2: nb = w, chunk b - 5
  if (b == 7) return u.L[6].w_off + (w * 4) * 256;                                          // up_2: nb = w
  if (b < 16) return u.L[4].w_off + ((2 * w + ((b - 8) & 1)) * 16 + 4 * ((b - 8) >> 1)) * 256;   // res_1: nb = 2w + r
  return u.L[7].w_off + ((2 * w + ((b - 16) & 1)) * 8 + 4 * ((b - 16) >> 1)) * 256;         // up_1
}

// floats between the four fragments of block b (4 KiB contiguous, except down_2's: one fragment per unit block, KC = 8 apart)
__host__ __device__ constexpr int r1_frag_stride(int b) { return b == 4 ? 8 * 256 : 256; }

// LDS map (floats)
struct R1Lds {
  static constexpr int r1 = 0;        // (64 lanes, 4 chunks)  down_0's output, lane-ordered: one ds_read_b128 per lane
  static constexpr int r2 = 256;      // (64, 2)   down_1's output
  static constexpr int o2 = 384;      // (64, 2)   stage 3's output
  static constexpr int p5 = 512;      // (16, 8)   up_0's per-wave partial sums, neuron-major
  static constexpr int tsr = 640;     // (32)      t_{k+1} of 32 steps: written with the scalars (wave 3), read by wave 0 in the step
  static constexpr int amat = 704;    // (16, 16)  A, P of the OU settings (wave 0's drift / running cost)
  static constexpr int pmat = 960;
  static constexpr int sc = 1204;     // (3, 4) per-step scalars of steps k - 1, k, k + 1 (dt, sqrt(lambda dt), dt / lambda, its root): the unused tail of pmat
  static constexpr int bias = 1216;   // the nine layers' padded biases (image order)
  static constexpr int p2 = 1216 + 1248;        // (8, 64)   down_2's per-wave partial sums: wave w's contribution of ITS 16 down_1 outputs to all 64 units
  static constexpr int nzb = p2 + 512;          // (24, 16)  the noise of 24 steps: drawn in batches of eight steps, one batch ahead (wave 2)
  static constexpr int scb = nzb + 384;         // (32, 4)   per-step scalars of 32 steps: batches of sixteen (wave 3)
  static constexpr int xk = p2 + 1024;          // (16)  x_k, written by wave 0 when it has integrated: wave 1 forms the d x d products of the
  static constexpr int pbx = xk + 16;           // (16)  OU / dense-sigma settings from it in its slack -- b = A x_k here,
  static constexpr int psx = pbx + 16;          // (16)  sigma eps_k here -- and wave 0 picks them up behind the barrier
  static constexpr int weights = p2 + 1024 + 64;     // LDS-resident blocks: wave 0's, then waves 1..7's, 1024 floats each
  static constexpr int xin = 0, res0 = 0, pb = 0, fq = 0;   // (17 <= d <= 31 only: R1LdsW; named here so that the shared code compiles)
};
// 17 <= d <= 31: 32-wide vectors, 32 x 32 matrices, the network input and res_0's output through LDS
struct R1LdsW {
  static constexpr int r1 = 0;
  static constexpr int r2 = 256;
  static constexpr int o2 = 384;
  static constexpr int p5 = 512;      // (32, 8)
  static constexpr int tsr = 768;     // (32)
  static constexpr int xin = 896;     // (32)   the network input [t, x, 0..] of the coming evaluation (wave 0 writes it)
  static constexpr int res0 = 928;    // (32)   res_0 [t, x] + b of the evaluation under way
  static constexpr int sc = 960;      // (3, 4)
  static constexpr int pb = 976;      // (32)   b = A x of the evaluation's state (OU settings: formed by wave 1 in the slack)
  static constexpr int fq = 1008;     // (1)    x' P x of the same state (OU_quadratic: wave 3)
  static constexpr int amat = 1024;   // (32, 36): row-major, rows padded to 36 floats (16-byte reads of eight row entries by the
  static constexpr int pmat = 2176;   //           16 lanes of a DPP row land on distinct banks)
  static constexpr int bias = 3328;
  static constexpr int p2 = 3328 + 1344;
  static constexpr int nzb = p2 + 512;          // (12, 32)  the noise of 12 steps: batches of four steps
  static constexpr int scb = nzb + 384;         // (32, 4)
  static constexpr int weights = p2 + 1024;
  static constexpr int xk = 0, pbx = 0, psx = 0;   // (d <= 15 only: named so that the shared code compiles)
};
static_assert((R1LdsW::weights + r1_lds_blocks(31) * 1024) * 4 <= 160 * 1024, "LDS-resident weight blocks do not fit (17 <= d <= 31)");
template <int H> struct R1LdsOf { typedef R1Lds type; };
template <> struct R1LdsOf<2> { typedef R1LdsW type; };
static_assert((R1Lds::weights + r1_lds_blocks(3) * 1024) * 4 <= 160 * 1024 && (R1Lds::weights + r1_lds_blocks(11) * 1024) * 4 <= 160 * 1024 &&
              (R1Lds::weights + r1_lds_blocks(15) * 1024) * 4 <= 160 * 1024, "LDS-resident weight blocks do not fit");
// dense sigma: sigma and S = sigma sigma^T (16 x 16 each) behind the weight blocks
static_assert((R1Lds::weights + r1_lds_blocks(103) * 1024 + 768) * 4 <= 160 * 1024 && (R1Lds::weights + r1_lds_blocks(111) * 1024 + 768) * 4 <= 160 * 1024 &&
              (R1Lds::weights + r1_lds_blocks(115) * 1024 + 768) * 4 <= 160 * 1024, "dense sigma: no room for sigma, sigma sigma^T");

// Developer instrumentation (-DSOCMX_R1_PROF): per-wave s_memtime deltas between the marks of a step, summed over the launch,
// written by workgroup 0 to a.prof[wave * 16 + slot] (socmx_rollout_phase_cycles_f32; tools/r1_phases.py).
#ifdef SOCMX_R1_PROF
#define R1_TICK(slot)                                \
  {                                                  \
    const long long now_ = __builtin_readcyclecounter(); \
    prof_acc[slot] += now_ - prof_last;              \
    prof_last = now_;                                \
  }
#else
#define R1_TICK(slot)
#endif

// ---- one wave of the workgroup --------------------------------------------------------------------------------------------
// DMAX: the state dimensions this instantiation takes (d <= DMAX): wave 0 holds 1 + DMAX input columns of down_0 / res_0
// MODE: 0 elementwise drift (double_well), 1 the same with a stopping time (molecular_dynamics), 2 OU drift (A x; x'Px for
// OU_quadratic); + 4: a dense sigma (u = -sigma^T nabla_V, sigma u = -(sigma sigma^T) nabla_V, sigma eps) -- MODE 4, 6
template <int CLS, int MODE, class NET, int DMAX0, bool EXPORT>
__device__ __forceinline__ void r1_wave(const RolloutArgs& a, float* lds, const int wave, const int lane) {
  constexpr bool DENSE = (MODE & 4) != 0;
  constexpr int DMAX = DMAX0;                       // state dimensions this instantiation takes
  constexpr int DMX = DMAX0 + (DENSE ? 100 : 0);    // ... as the key of the source plans
  constexpr UnetDesc u = NET::desc();
  // H: 16-component halves of the row's vectors.  H = 2 (17 <= d <= 31, the 32-wide network input / output): lane n of a row
  // carries components n and 16 + n; down_0 and res_0 run as a stage of their own on all waves (network(): stage 0)
  constexpr int H = NET::outp >> 4;
  static_assert(H == 1 || (H == 2 && DMAX0 == 31 && !DENSE), "17 <= d <= 31: sigma = I, one instantiation");
  typedef typename R1LdsOf<H>::type LM;
  // EXPORT (socmx_rollout_ex_f32's act_workspace / act_records): every evaluation's activations leave as the slabs the control-network
  // backward would otherwise re-compute, their ReLU signs as a 128-byte record per row (socmx_unet.h).  Stores only: a buffer store per
  // tensor and wave (scalar base: tile and row of the evaluation; lane offset: the unit), one record store per wave.
  constexpr bool EX = EXPORT;
  static_assert(!EXPORT || H == 1, "the activation export is built for d <= 15");
  constexpr int MS = 16 * H;                        // row stride of the noise ring (and of A, P at d <= 15)
  constexpr int MSA = H == 2 ? 36 : 16;             // row stride of A, P in LDS
  constexpr int NBS = 8 / H;                        // steps per noise batch (noise_batch below); its ring holds three batches: slot = step % (3 NBS)
  constexpr int NRES = r1_count(CLS, DMX, 'R'), NLDS = r1_count(CLS, DMX, 'L'), NSTR = r1_count(CLS, DMX, 'S');
  const float* __restrict__ Wp = a.packed;
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  // Which row this workgroup integrates.  Workgroup b runs on XCD b % 8 (observed placement, used for speed only: any mapping is
  // correct), and a row's per-step stores are 4 d bytes -- 40 at d = 10, a third of a 128-byte line of the (K, B, d) tensors.
  // With row = b, the rows sharing a line belong to workgroups on DIFFERENT XCDs: each XCD's L2 holds its fragment of the line
  // and writes its sectors back on its own (rocprofv3 WRITE_SIZE 2.3x the bytes stored, profiles/r4/pmc_summary.json).  Rows
  // are therefore dealt out XCD by XCD -- XCD x takes the contiguous rows [row0(x), row0(x + 1)) -- so that a line's fragments
  // meet in ONE L2 before they leave it.
  const int xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3;
  const int rows_base = a.B >> 3, rows_rem = a.B & 7;
  const int grow = xcd * rows_base + min(xcd, rows_rem) + in_xcd;       // (blocks b = x mod 8: ceil((B - x) / 8) = base + (x < rem) of them)
  const int g = lane >> 4, n = lane & 15;
  const uint32_t loff = lane * 16;
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);

#ifdef SOCMX_R1_PROF
  long long prof_acc[16] = {0}, prof_last = 0;
#endif
  // ---- resident weights ----
  float wres[NRES][16];
#pragma unroll
  for (int r = 0; r < NRES; ++r) {
    const int rb = r1_nth(CLS, DMX, 'R', r);            // (a constant once the loop is unrolled)
    if (R1_S2DIRECT && rb == 4) {
      // down_2's block in the DIRECT layout (round 6): lane (g, n), register j = W[unit 16 g + n][input 16 w + j] -- the row index
      // picks the unit block, the broadcast position the input.  The operand is then the wave's r2 outputs as they are (y: position
      // n of EVERY row = r2[16 w + n]) and lane (g, n) ends with the whole sum of unit 16 g + n: no cross-lane gather in front of
      // the fmacs (it was a ds_bpermute round trip on the step's chain), no cross-row reduction behind them.  In the fragment
      // image that element sits in fragment (nb = g, kc = w) at lane ((j >> 2), n), component j & 3: sixteen 4-byte reads, once.
#pragma unroll
      for (int j = 0; j < 16; ++j)
        wres[r][j] = Wp[u.L[2].w_off + (((g * (u.L[2].in_pad >> 4) + wave) * 64) + (j >> 2) * 16 + n) * 4 + (j & 3)];
      continue;
    }
    const f32x4* src = reinterpret_cast<const f32x4*>(Wp + r1_block_off<NET>(rb, wave)) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = src[c * (r1_frag_stride(rb) / 4)];
#pragma unroll
      for (int e = 0; e < 4; ++e) wres[r][c * 4 + e] = v[e];
    }
  }
  static_assert(r1_src(CLS, DMX, 4) == 'R', "down_2's strided fragments are only loaded by the resident path");
  // up_0's share of this wave: k = 32 w + 16 (g & 1) + 8 (g >> 1) + j  (rows 2, 3 read the rotated copy of the wave's outputs)
  float w5[H][8];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * wave + 16 * (g & 1) + 8 * (g >> 1) + j;           // (output block h: KC = 16 fragments further on)
      w5[h][j] = Wp[u.L[8].w_off + (((h * 16 + (k >> 4)) * 64) + ((k & 15) >> 2) * 16 + n) * 4 + (k & 3)];
    }
  // the FOLD's share of this wave (UnetDesc::fold = up_0 res_1, fragment-ordered): r1 inputs 32 w .. 32 w + 31 = positions
  // 8 (w & 1) .. 8 (w & 1) + 7 of the activation register of chunk w >> 1 -- the fragments (h, 4 (w >> 1) + 2 (w & 1) + f), f < 2
  float wf[H][8];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kc = 4 * (wave >> 1) + 2 * (wave & 1) + (j >> 2);
      wf[h][j] = Wp[u.fold.w_off + ((h * 16 + kc) * 64 + lane) * 4 + (j & 3)];
    }
  // H = 2: this wave's two unit blocks of down_0 (blocks 2 w, 2 w + 1; two 16-input fragments each: register 4 kc + i <->
  // broadcast position 4 kc + i of the input register) and, on wave 0, res_0's two blocks
  float wd0[H == 2 ? 2 : 1][8], w3f[H == 2 ? 2 : 1][8];
  if constexpr (H == 2) {
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const f32x4 v = *(reinterpret_cast<const f32x4*>(Wp + u.L[0].w_off) + ((2 * wave + bb) * 2 + kc) * 64 + lane);
        const f32x4 v3 = *(reinterpret_cast<const f32x4*>(Wp + u.L[3].w_off) + (bb * 2 + kc) * 64 + lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wd0[bb][kc * 4 + e] = v[e]; w3f[bb][kc * 4 + e] = v3[e]; }
      }
  }
  // biases of the units whose totals land in this lane
  static_assert(u.bias_floats <= LM::p2 - LM::bias, "bias copy");
  const float* BL = lds + LM::bias;          // (copied by the kernel's prologue; read where a layer ends)
  // LDS-resident blocks: copied once, read back with ds_read_b128 at lane * 16 + fragment * 1 KiB
  constexpr int lds_first = CLS == 0 ? 0 : r1_count(0, DMX, 'L');
  static_assert(r1_count(CLS, DMX, 'S') == 0 || (r1_count(CLS, DMX, 'S') % kR1PD == 0 && r1_count(CLS, DMX, 'S') >= kR1PD), "static ring slots across steps");
  static_assert(r1_count(CLS, DMX, 'X') == 8 && r1_src(CLS, DMX, 8) == 'X' && r1_src(CLS, DMX, 15) == 'X', "blocks 8-15 (res_1) are folded into up_0");
  float* LW = lds + LM::weights + (lds_first + (CLS == 0 ? 0 : (wave - 1) * NLDS)) * 1024;
#pragma unroll
  for (int r = 0; r < NLDS; ++r) {
    const f32x4* src = reinterpret_cast<const f32x4*>(Wp + r1_block_off<NET>(r1_nth(CLS, DMX, 'L', r), wave)) + lane;
    f32x4* dst = reinterpret_cast<f32x4*>(LW + r * 1024) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c * 64] = src[c * 64];
  }
  const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.packed), 0, u.total_floats * 4, 0x00020000);
  // the stream ring: the first kR1PD stream blocks are requested here, every consumed block requests the block kR1PD later
  f32x4 ring[kR1PD][4];
  if constexpr (NSTR > 0) {
#pragma unroll
    for (int s = 0; s < kR1PD; ++s) {
      const int p = r1_block_off<NET>(r1_nth(CLS, DMX, 'S', s), wave) * 4;
      ring[s][0] = r1_gload<0>(img, loff, p);
      ring[s][1] = r1_gload<1024>(img, loff, p);
      ring[s][2] = r1_gload<2048>(img, loff, p);
      ring[s][3] = r1_gload<3072>(img, loff, p);
    }
  }

  // The first half of an LDS block is read one unit AHEAD of its fmacs (lq: eight landing registers; pre(b) sits in front
  // of the fmacs of the unit before b), the second half in front of the first half's fmacs.
  float lq[8];
  auto pre = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    if constexpr (b < kR1Blocks) {
      if constexpr (r1_src(CLS, DMX, b) == 'L') {
        constexpr int r = r1_count(CLS, DMX, 'L', b);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f32x4 v = *(reinterpret_cast<const f32x4*>(LW + r * 1024) + c * 64 + lane);
#pragma unroll
          for (int e = 0; e < 4; ++e) lq[c * 4 + e] = v[e];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // half h of block b into w[8]
  auto fetch = [&](auto bc, auto hc, float (&w)[8]) {
    constexpr int b = decltype(bc)::value, h = decltype(hc)::value;
    constexpr char src = r1_src(CLS, DMX, b);
    if constexpr (src == 'R') {
      constexpr int r = r1_count(CLS, DMX, 'R', b);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = wres[r][8 * h + e];
    } else if constexpr (src == 'L') {
      if constexpr (h == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = lq[e];
      } else {
        constexpr int r = r1_count(CLS, DMX, 'L', b);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const f32x4 v = *(reinterpret_cast<const f32x4*>(LW + r * 1024) + (2 + c) * 64 + lane);
#pragma unroll
          for (int e = 0; e < 4; ++e) w[c * 4 + e] = v[e];
        }
      }
    } else {
      constexpr int s = r1_count(CLS, DMX, 'S', b) % kR1PD;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[c * 4 + e] = ring[s][2 * h + c][e];
    }
  };
  auto refill = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    if constexpr (r1_src(CLS, DMX, b) == 'S') {
      constexpr int i = r1_count(CLS, DMX, 'S', b), s = i % kR1PD;
      const int p = r1_block_off<NET>(r1_nth(CLS, DMX, 'S', (i + kR1PD) % NSTR), wave) * 4;   // (wraps into the next step)
      ring[s][0] = r1_gload<0>(img, loff, p);
      ring[s][1] = r1_gload<1024>(img, loff, p);
      ring[s][2] = r1_gload<2048>(img, loff, p);
      ring[s][3] = r1_gload<3072>(img, loff, p);
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // one block: a0 / a1 (alternating) += x . W_b; `nx` = the block whose LDS reads are issued in front of this one's fmacs
  auto blk = [&](auto bc, auto nx, float& a0, float& a1, float x) {
    float w0h[8], w1h[8];
    fetch(bc, H0{}, w0h);
    fetch(bc, H1{}, w1h);
    pre(nx);
    r1_fmac8<0>(a0, a1, x, w0h);
    r1_fmac8<1>(a0, a1, x, w1h);
    refill(bc);
  };
  // two blocks of one chunk: aA += x . W_bA, aB += x . W_bB (at most one of them an LDS block); nx as above
  auto blk2 = [&](auto ba, auto bb, auto nx, float& aA, float& aB, float x) {
    static_assert(!(r1_src(CLS, DMX, decltype(ba)::value) == 'L' && r1_src(CLS, DMX, decltype(bb)::value) == 'L'), "one landing buffer");
    float wa0[8], wb0[8], wa1[8], wb1[8];
    fetch(ba, H0{}, wa0);
    fetch(bb, H0{}, wb0);
    fetch(ba, H1{}, wa1);
    fetch(bb, H1{}, wb1);
    pre(nx);
    r1_fmac8x2<0>(aA, aB, x, wa0, wb0);
    r1_fmac8x2<1>(aA, aB, x, wa1, wb1);
    refill(ba);
    refill(bb);
  };
  static_assert(r1_src(CLS, DMX, 0) != 'L', "a step's first block has nobody in front of it to issue its LDS reads");
#define R1B(b) std::integral_constant<int, b>{}

  // ---- wave 0's own state: the row (every 16-lane row of the wave runs the same arithmetic: component i = lane & 15) ----
  const int i = n;
  int icv[H];
  bool okv[H];
#pragma unroll
  for (int h = 0; h < H; ++h) { icv[h] = min(i + 16 * h, d - 1); okv[h] = i + 16 * h < d; }
  const int ic = icv[0];
  const bool lane_ok = okv[0];
  constexpr bool STOPPING = MODE == 1, is_ou = (MODE & 2) != 0;
  constexpr bool OFF = H == 1 && (is_ou || DENSE);      // d <= 15: b = A x and sigma eps are formed by wave 1 (h1_products)
  static_assert(!(STOPPING && DENSE), "the stopping-time step is built for sigma = I");
  const bool is_quad = is_ou && kind == SOCMX_OU_QUADRATIC;
  const bool traj = a.states != nullptr;
  const bool store = CLS == 0 && lane < 16 && lane_ok && traj;
  const bool store_h = H == 2 && CLS == 0 && lane < 16 && okv[H - 1] && traj;
  const bool store0 = CLS == 0 && lane == 0 && traj;
  const uint32_t rowoff = (uint32_t)(grow * d + i);     // (32-bit lane offset against wave-uniform step bases: SGPR-base stores)
  const size_t step_floats = (size_t)B * d;
  size_t kbd = 0, kb = 0;                                // k * B * d, k * B
  float x = 0.f, kap = 0.f, stop = 1.f, lpd = 0.f, lps = 0.f, res0 = 0.f;
  float xh = 0.f, kaph = 0.f, pre_bh = 0.f, pre_seh = 0.f, b8h = 0.f;   // H = 2: the upper halves (components 16 + n)
  float w0[4][H == 1 ? DMAX + 1 : 1], b0[4], w3[H == 1 ? DMAX + 1 : 1], b8 = 0.f, b3 = 0.f;   // H = 1: down_0 (units 64 m + lane), res_0 (unit n): wave 0 only
  float* A_l = lds + LM::amat;                         // OU: A, P with row stride 16 (d <= 15)
  float* P_l = lds + LM::pmat;
  float* S_l = lds + LM::weights + r1_lds_blocks(DMX) * 1024;   // dense sigma: sigma, then S = sigma sigma^T (stride 16)
  float* SS_l = S_l + 256;
  float* ST_l = S_l + 512;                              // ... and sigma^T (the control u = -sigma^T nabla_V reads its rows)
  float srow[DENSE ? DMAX : 1];                           // row i of S: sigma u = -S nabla_V is on the step's serial chain
  float pre_b = 0.f, pre_se = 0.f;                        // drift b(x_k) and (sigma eps_k)_i of the coming step, formed in the slack
  if constexpr (CLS == 0) {
    x = lane_ok ? a.x0[(size_t)grow * d + i] : 0.f;
    kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
    if constexpr (H == 2) {
      xh = okv[1] ? a.x0[(size_t)grow * d + 16 + i] : 0.f;
      kaph = (okv[1] && !is_ou) ? a.kappa[16 + i] : 0.f;
      b8h = Wp[u.L[8].b_off + 16 + n] + Wp[u.fold.b_off + 16 + n];     // (+ up_0 b4: the fold's constant term)
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int unit = 64 * m + lane;
#pragma unroll
        for (int kk = 0; kk <= DMAX; ++kk)
          w0[m][kk] = Wp[u.L[0].w_off + (((unit >> 4) * 64) + (kk >> 2) * 16 + (unit & 15)) * 4 + (kk & 3)];
        b0[m] = Wp[u.L[0].b_off + unit];
      }
#pragma unroll
      for (int kk = 0; kk <= DMAX; ++kk) w3[kk] = Wp[u.L[3].w_off + ((kk >> 2) * 16 + n) * 4 + (kk & 3)];
    }
    b8 = Wp[u.L[8].b_off + n] + Wp[u.fold.b_off + n];                    // (+ up_0 b4: the fold's constant term)
    b3 = Wp[u.L[3].b_off + n];
    for (int e = lane; e < d * d; e += 64) {
      const int r = e / d, c = e - r * d;
      if (is_ou) A_l[r * MSA + c] = a.A[e];
      if (is_quad) P_l[r * MSA + c] = a.P[e];
    }
    if constexpr (DENSE) {
      for (int e = lane; e < 256; e += 64) {
        S_l[e] = ((e >> 4) < d && (e & 15) < d) ? a.sigma[(e >> 4) * d + (e & 15)] : 0.f;
        ST_l[e] = ((e >> 4) < d && (e & 15) < d) ? a.sigma[(e & 15) * d + (e >> 4)] : 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (wave-private LDS traffic: no barrier needed)
      for (int e = lane; e < 256; e += 64) {
        float acc = 0.f;
        for (int c = 0; c < 16; ++c) acc += S_l[(e >> 4) * 16 + c] * S_l[(e & 15) * 16 + c];
        SS_l[e] = acc;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < DMAX; ++j) srow[j] = SS_l[n * 16 + j];
    }
    if (store) a.states[rowoff] = x;
    if constexpr (OFF) {
      if (lane < 16) lds[LM::xk + i] = x;
    }
    if (H == 2 && store_h) a.states[rowoff + 16] = xh;
    if (store0) a.stop_ind[grow] = 1.f;
  }
  // component j (any j < d) of a vector held as (lo, hi) halves in the 16-lane rows
  auto comp = [&](float lo, float hi, int j) -> float {
    const float t0 = __shfl(lo, j & 15, 16);
    if constexpr (H == 2) { const float t1 = __shfl(hi, j & 15, 16); return (j & 16) ? t1 : t0; }
    return t0;
  };
  // y_i = sum_j M[i][j] v_j for the component i this lane carries: row i of a 16 x 16 LDS matrix (zero past d) as four 16-byte
  // reads, v_j by DPP row broadcast inside the multiply-adds (H = 1: v holds component j at position j of every row, zero past
  // d).  A `for (j < d) .. __shfl(v, j, 16)` loop was d ds_bpermute round trips: at OU_linear d = 10 (dense sigma) the three such
  // products per step outlasted the slack wave 0 has for them -- 3.5 us per step against 2.0 for the elementwise drift.
  auto row_dot = [&](const float* M16, float v) -> float {
    float w[16];
    const f32x4* r4 = reinterpret_cast<const f32x4*>(M16 + ic * 16);
#pragma unroll
    for (int q = 0; q < (DMAX + 3) / 4; ++q) {
      const f32x4 t = r4[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[4 * q + e] = t[e];
    }
    float a0 = 0.f, a1 = 0.f;
    r1_state_one<DMAX>(a0, a1, v, w);
    return a0 + a1;
  };
  // what the coming step's chain needs besides nabla_V, from values known a stage earlier: b(x_k) and sigma eps_k
  auto prepare_step = [&](int k) {
    if (k >= K) return;
    const float eps = lds[LM::nzb + (k % (3 * NBS)) * MS + i];
    if constexpr (H == 2) pre_seh = lds[LM::nzb + (k % (3 * NBS)) * MS + 16 + i];
    if (is_ou) {                                                        // b = A x   (OU_quadratic.py:51-52, OU_linear.py:43-44)
      if constexpr (H == 1 && !OFF) {
        pre_b = lane_ok ? row_dot(A_l, x) : 0.f;
      }                                                                 // (H = 2: wave 1 forms it, ou_products(); read behind the barrier)
    } else {
      pre_b = -2.f * kap * (x * x - 1.f) * 2.f * x;                     // double_well.py:44-48
      if constexpr (H == 2) pre_bh = -2.f * kaph * (xh * xh - 1.f) * 2.f * xh;
    }
    if constexpr (DENSE) {
      if constexpr (!OFF) pre_se = lane_ok ? row_dot(S_l, lane_ok ? eps : 0.f) : 0.f;
    } else {
      pre_se = eps;
    }
  };
  // res_0 [t_k, x_k] + b, the skip term of the evaluation UNDER WAY (added behind the network's last ReLU): eleven multiply-adds that
  // sat in first_layer, on the serial section every other wave waits for; wave 0 now forms them in its slack behind stage 4
  // (x is still x_k there, t_k was kept when first_layer ran)
  float res_t = 0.f;
  auto res0_update = [&]() {
    if constexpr (R1_RES0_SLACK && H == 1 && CLS == 0) {
      float r0 = b3 + res_t * w3[0], r1v = 0.f;
      r1_state_one<DMAX>(r0, r1v, x, &w3[1]);
      res0 = r0 + r1v;
    }
  };
  // y[unit] = relu(down_0 [t, x] + b) for the 256 units, lane-ordered into LDS (wave 0); res_0 [t, x] + b for the step's end
  auto first_layer = [&](float t) {
    if constexpr (H == 2) {
      // the network input [t, x_0 .. x_30, 0]: stage 0 of the evaluation that follows reads it (all waves, behind the barrier)
      if (lane < 16) {
        lds[LM::xin + 1 + i] = x;
        if (i < 15) lds[LM::xin + 17 + i] = xh;
        if (i == 0) lds[LM::xin] = t;
      }
      return;
    }
    float acc[4];
    [[maybe_unused]] float r0 = b3 + t * w3[0], r1v = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = b0[m] + t * w0[m][0];
    if constexpr (DMAX <= 11) {
      r1_state_pair<DMAX>(acc[0], acc[1], x, &w0[0][1], &w0[1][1]);
      r1_state_pair<DMAX>(acc[2], acc[3], x, &w0[2][1], &w0[3][1]);
    } else {
      float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
      r1_state_one<DMAX>(acc[0], e0, x, &w0[0][1]);
      r1_state_one<DMAX>(acc[1], e1, x, &w0[1][1]);
      r1_state_one<DMAX>(acc[2], e2, x, &w0[2][1]);
      r1_state_one<DMAX>(acc[3], e3, x, &w0[3][1]);
      acc[0] += e0; acc[1] += e1; acc[2] += e2; acc[3] += e3;
    }
    if constexpr (!R1_RES0_SLACK) {
      r1_state_one<DMAX>(r0, r1v, x, &w3[1]);
      res0 = r0 + r1v;
    } else {
      res_t = t;                               // (res_0 [t, x] + b is formed in the slack of the evaluation that uses it: res0_update)
    }
    f32x4 y;
#pragma unroll
    for (int m = 0; m < 4; ++m) y[m] = relu_keep_nan(acc[m]);
    *reinterpret_cast<f32x4*>(lds + LM::r1 + r1_perm(lane) * 4) = y;
  };

  // ---- noise and scalars in BATCHES: wave 2 draws eight steps' noise (four at 17 <= d <= 31) with all 64 lanes once every eight
  // steps, one batch ahead; wave 3 forms sixteen steps' scalars once every sixteen steps.  Per step that is ~40 instructions
  // instead of ~300 (Philox4x32-10 + Box-Muller per step ran ~1.2k cycles on the drawing wave: hidden behind the long stage 4
  // until round 5, on the step's critical path once res_1 had left it).  Draws as documented in include/socmx.h: lane = (step,
  // pair), the same words, the same pair arithmetic -- bit-identical to the other tile shapes.
  float* NZ = lds + LM::nzb;
  auto noise_batch = [&](int nb) {
    if (!(CLS == 1 && wave == 2)) return;
    const int k = NBS * nb + lane / (8 * H), pr = lane % (8 * H), c0 = 2 * pr;
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
    NZ[(k % (3 * NBS)) * MS + c0] = c0 < d ? z0 : 0.f;
    NZ[(k % (3 * NBS)) * MS + c0 + 1] = c0 + 1 < d ? z1 : 0.f;
  };
  // The step's scalars -- dt (utils.py:38), sqrt(lambda dt) (utils.py:47), dt / lambda and its root -- are an IEEE division and
  // two square roots, ~40 dependent instructions per step: lane l of wave 3 forms those of step 16 nb + l
  float* SC = lds + LM::scb;
  auto scalar_batch = [&](int nb) {
    if (!(CLS == 1 && wave == 3 && lane < 16)) return;
    const int k = 16 * nb + lane;
    if (k >= K) return;
    const float dt = a.ts[k + 1] - a.ts[k];
    const float dol = dt / a.lmbd;
    *reinterpret_cast<f32x4*>(SC + (k & 31) * 4) = f32x4{dt, sqrtf(a.lmbd * dt), dol, sqrtf(dol)};
    lds[LM::tsr + (k & 31)] = a.ts[k + 1];
  };
  // (a batch is drawn while the step 8 nb is integrated -- its slots were last read three batches ago -- and first read eight
  //  steps later; the scalars of steps 16 nb + 16 .. while step 16 nb + 1 is integrated: the other half of their ring)
  auto batches = [&](int k) {
    if (k % NBS == 0) noise_batch(k / NBS + 1);
    if ((k & 15) == 1) scalar_batch((k >> 4) + 1);
  };
  // ---- what a step leaves behind for later (wave 0): only x_{k+1} -> down_0 -> r1 is on the path to the next evaluation;
  // the running costs (two row sums), the trajectory stores and nabla_V's hand-over are issued at the end of the NEXT
  // evaluation's stage 4, where wave 0 waits for the others anyway, from a few registers (the step's scalars sit in a
  // three-deep LDS ring: wave 3 writes those of step k + 2 in the same slack)
  int cur_k = 0;                       // the step network() runs for (the noise waves prepare steps cur_k + 1, cur_k + 2 inside)
  // ---- the activation export (EX) ----
  const __amdgpu_buffer_rsrc_t ex_rs = __builtin_amdgcn_make_buffer_rsrc(EX ? (void*)a.act_ws : (void*)lds, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t ex_rr = __builtin_amdgcn_make_buffer_rsrc(EX ? (void*)a.act_rec : (void*)lds, 0, 0x7FFFFFFF, 0x00020000);
  const int ex_pk = r1_perm(lane);
  int ex_tile = 0, ex_r4 = 0, ex_row = 0;          // tile, 4 (row & 15), row of the evaluation under way (wave-uniform)
  uint32_t ex_d0 = 0, ex_d1 = 0;
  uint64_t ex_b64 = 0;
  float bk_pre = 0.f;                               // wave 0: the output's pre-activation of the step whose books are open
  float ex_o1 = 0.f;                                // waves 1 .. 7: this evaluation's A1 units, kept for ex_flush()
  auto ex_begin = [&]() {
    ex_row = cur_k * B + grow;
    ex_tile = ex_row >> 4;
    ex_r4 = (ex_row & 15) * 4;
  };
  // this lane's value of unit `unit` of slab tensor T (socmx_unet.h): [tile][unit][16 rows]
  auto ex_store = [&](auto tc, int unit, float v) {
    constexpr int T = decltype(tc)::value;
    constexpr int W = tensor_width(u, T), P = tensor_prefix(u, T);
    const int soff = (a.act_tile_rows * P + ex_tile * (W * 16)) * 4 + ex_r4;
    if constexpr ((SOCMX_R1_EX_PARTS & 1) != 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ex_rs, unit * 64, soff, 0);
  };
  // a wave's own record: dwords 0, 1; dwords 2, 3 only where the wave itself formed them (waves 1 .. 3: their chunk of R1) -- slots 4 and 5
  // get theirs from waves 0 and 1 (ex_aux), wave 0's dword 2 follows with its books
  auto ex_record = [&]() {
    if ((SOCMX_R1_EX_PARTS & 2) != 0 && lane == 0) {
      const int soff = (ex_row * kActRecordDwords + wave * 4) * 4;
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      if (CLS == 1 && wave <= 3) __builtin_amdgcn_raw_buffer_store_b128(u32x4{ex_d0, ex_d1, (uint32_t)ex_b64, (uint32_t)(ex_b64 >> 32)}, ex_rr, 0, soff, 0);
      else __builtin_amdgcn_raw_buffer_store_b64(u32x2{ex_d0, ex_d1}, ex_rr, 0, soff, 0);
    }
  };
  auto ex_aux = [&](int slot, uint64_t b) {         // dwords 2, 3 of another wave's slot (64 signs in lane order)
    if ((SOCMX_R1_EX_PARTS & 2) != 0 && lane == 0) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)b, (uint32_t)(b >> 32)}, ex_rr, 0, (ex_row * kActRecordDwords + slot * 4 + 2) * 4, 0);
    }
  };
  auto ex_output_sign = [&](int row, float pre) {   // wave 0: the sign of the output's pre-activation -> dword 2 of its record
    const uint32_t m = (uint32_t)__builtin_amdgcn_ballot_w64(pre > 0.f) & 0xFFFFu;
    if ((SOCMX_R1_EX_PARTS & 2) != 0 && lane == 0) __builtin_amdgcn_raw_buffer_store_b32(m, ex_rr, 0, (row * kActRecordDwords + 2) * 4, 0);
  };
  using TR1 = std::integral_constant<int, T_R1>;
  using TR2 = std::integral_constant<int, T_R2>;
  using TR3 = std::integral_constant<int, T_R3>;
  using TO2 = std::integral_constant<int, T_O2>;
  using TO1 = std::integral_constant<int, T_O1>;
  int bk_k = -1;                       // the step whose bookkeeping is outstanding
  float bk_gv = 0.f, bk_step = 0.f, bk_eps = 0.f, bk_gvh = 0.f, bk_epsh = 0.f;
  float bk_sol = 0.f;                  // H = 2, OU_quadratic: step / lambda of the step whose -f x'Px term is outstanding
  // H = 2, OU settings: the d x d products of the state leave wave 0.  Waves 1 and 3 read the evaluation's input vector from LDS
  // (xin = [t_k, x_k]) in the slack behind stage 4, form y = M x_k as DPP multiply-adds (below) and leave b = A x_k (wave 1) and
  // x_k' P x_k (wave 3) in LDS; wave 0 picks them up behind the barrier.  (On wave 0 they were forty ds_bpermute + eighty
  // multiply-adds per step at d = 20: 6.5 us per step against 3.7 for the elementwise drift.)
  auto ou_products = [&]() {
    if constexpr (H == 2 && is_ou && CLS == 1) {
      // (wave 1 for A x: wave 2 draws the noise batches and wave 3 the scalars in the same slack)
      if (wave == 1 || (wave == 3 && is_quad)) {
        // y = M x for the 32-wide vector as DPP multiply-adds: row g of the wave takes the eight terms j = 8 g .. 8 g + 7 of both
        // outputs of its lane (i = n and 16 + n) -- operand register: rows 0, 1 hold x_0 .. x_15 (row 1 rotated by eight), rows 2, 3
        // x_16 .. x_31 likewise, so that position e of row g is x_{8 g + e}; the eight matrix entries M[i][8 g ..] are two 16-byte
        // LDS reads -- and the four rows' partial sums meet in r1_rows_sum.  (The first form -- one LDS pass per term in a rolled
        // loop, lanes along i -- took ~2,000 cycles of these waves' ~800-cycle slack: everybody waited for them, 7.0k cycles per
        // step at d = 20 against 6.0k for the elementwise drift.)
        const float* M = wave == 1 ? A_l : P_l;
        const float xlo = lds[LM::xin + 1 + n], xhi = n < 15 ? lds[LM::xin + 17 + n] : 0.f;
        const float xs = r1_ror8_odd_rows(g < 2 ? xlo : xhi);
        float wl[8], wh[8];
        {
          const f32x4* rl = reinterpret_cast<const f32x4*>(M + n * MSA + 8 * g);
          const f32x4* rh = reinterpret_cast<const f32x4*>(M + (16 + n) * MSA + 8 * g);
          const f32x4 l0 = rl[0], l1 = rl[1], h0 = rh[0], h1 = rh[1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { wl[e] = l0[e]; wl[4 + e] = l1[e]; wh[e] = h0[e]; wh[4 + e] = h1[e]; }
        }
        float al0 = 0.f, al1 = 0.f, ah0 = 0.f, ah1 = 0.f;
        r1_fmac8<0>(al0, al1, xs, wl);
        r1_fmac8<0>(ah0, ah1, xs, wh);
        const float ylo = r1_rows_sum(al0 + al1), yhi = r1_rows_sum(ah0 + ah1);
        if (wave == 1) {
          if (lane < 16) {
            lds[LM::pb + n] = n < d ? ylo : 0.f;
            lds[LM::pb + 16 + n] = 16 + n < d ? yhi : 0.f;
          }
        } else {
          float t = lane < 16 ? fmaf(xhi, yhi, __fmul_rn(xlo, ylo)) : 0.f;    // (x is zero past d)
          t = row16_sum(t);
          if (lane == 0) lds[LM::fq] = t;
        }
      }
    }
  };
  // d <= 15, OU drift and / or dense sigma: wave 1 forms b = A x_k and sigma eps_k in its slack behind stage 4 (x_k from LDS, where
  // wave 0 left it when it integrated; eps_k from the noise ring) -- on wave 0 the two products stood between its books and the
  // barrier everybody waits at (1.5k cycles of "slack" work against 0.5k on the other waves)
  auto h1_products = [&]() {
    if constexpr (OFF && CLS == 1) {
      if (wave == 1 && cur_k < K) {
        if constexpr (is_ou) {
          const float pbv = row_dot(A_l, lds[LM::xk + i]);
          if (lane < 16) lds[LM::pbx + i] = lane_ok ? pbv : 0.f;
        }
        if constexpr (DENSE) {
          const float e = lds[LM::nzb + (cur_k % (3 * NBS)) * MS + i];
          const float psv = row_dot(S_l, lane_ok ? e : 0.f);
          if (lane < 16) lds[LM::psx + i] = lane_ok ? psv : 0.f;
        }
      }
    }
  };
  auto apply_quad_cost = [&]() {       // wave 0, behind the barrier that follows ou_products()
    if constexpr (H == 2 && CLS == 0) {
      if (is_quad) lpd = fmaf(-bk_sol, lds[LM::fq], lpd);
      bk_sol = 0.f;
    }
  };
  auto bookkeeping = [&]() {
    if (bk_k < 0) return;
    const int k = bk_k;
    bk_k = -1;
    const f32x4 scal = *reinterpret_cast<const f32x4*>(lds + LM::scb + (k & 31) * 4);
    const float eps = bk_eps;
    float uc = lane_ok ? -bk_gv : 0.f;                                  // u = -sigma^T nabla_V (method.py:58-80)
    if constexpr (DENSE) {
      uc = lane_ok ? -row_dot(ST_l, lane_ok ? bk_gv : 0.f) : 0.f;
    }
    const float xe = x;                                                 // x_{k+1}
    const float uch = (H == 2 && okv[H - 1]) ? -bk_gvh : 0.f;
    float f = 0.f;                                                      // f at the NEW state, OLD time (utils.py:92-96)
    if (is_quad) {
      if constexpr (H == 1) {
        f = row16_sum(lane_ok ? xe * row_dot(P_l, xe) : 0.f);
      }                                       // (H = 2: x'Px comes from wave 3 behind the barrier -- apply_quad_cost())
    } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
      f = 1.f;
    }
    float puu = uc * uc, pue = uc * eps;
    if constexpr (H == 2) { puu = fmaf(uch, uch, __fmul_rn(uc, uc)); pue = fmaf(uch, bk_epsh, __fmul_rn(uc, eps)); }
    const float uu = row16_sum(puu), ue = row16_sum(pue);
    const float sol = STOPPING ? bk_step / a.lmbd : scal[2];
    const float ssol = STOPPING ? sqrtf(sol) : scal[3];
    if constexpr (H == 2) {
      lpd = fmaf(sol, fmaf(-0.5f, uu, -f), lpd);
      lps = fmaf(ssol, -ue, lps);
      bk_sol = sol;
    } else {
      lpd = lpd + sol * (-f - 0.5f * uu);
      lps = lps + ssol * (-ue);
    }
    if (store) {
      if (a.nabla_v) (a.nabla_v + kbd)[rowoff] = bk_gv;
      (a.controls + kbd)[rowoff] = uc;
      (a.noises + kbd)[rowoff] = eps;
      (a.states + kbd + step_floats)[rowoff] = xe;
    }
    if (H == 2 && store_h) {
      if (a.nabla_v) (a.nabla_v + kbd)[rowoff + 16] = bk_gvh;
      (a.controls + kbd)[rowoff + 16] = uch;
      (a.noises + kbd)[rowoff + 16] = bk_epsh;
      (a.states + kbd + step_floats)[rowoff + 16] = xh;
    }
    if constexpr (EX) ex_output_sign(k * B + grow, bk_pre);
    kbd += step_floats;
    if (store0) {
      (a.frac + kb)[grow] = STOPPING ? bk_step : scal[0];
      (a.stop_ind + kb + B)[grow] = STOPPING ? stop : 1.f;
    }
    kb += B;
  };

  // ---- stages 1..5 of the network on the row; leaves up_0's per-wave partial sums in LDS (p5) ----
  // the LDS block among blocks [first, first + count), kR1Blocks if none: what the unit in front of them prefetches
#define R1NX(first, count)                                                                                         \
  std::integral_constant<int, (r1_src(CLS, DMX, first) == 'L'                         ? (first)                     \
                               : ((count) > 1 && r1_src(CLS, DMX, (first) + 1) == 'L') ? (first) + 1                 \
                                                                                        : kR1Blocks)>{}
  // (The two waves of a SIMD share its VALU; at equal priority the older one wins most slots, finishes a multi-block stage
  //  early and leaves the younger one to run the rest alone.  Alternating s_setprio block by block to keep them abreast was
  //  measured and is slower -- 0.626 ms against 0.559: both waves then stall more often than they gain.)
  auto network = [&]() {
    if constexpr (H == 2) {
      // stage 0 (17 <= d <= 31): r1 = relu(down_0 [t, x] + b), wave w: units 32 w .. 32 w + 31 from its sixteen resident
      // registers; wave 0 also res_0 [t, x] + b.  The input register: row g holds input 16 f + 4 g + i at position 4 f + i.
      const float xin = (lane & 15) < 8 ? lds[LM::xin + 16 * ((lane & 15) >> 2) + 4 * g + (lane & 3)] : 0.f;
      float aA = 0.f, aB = 0.f;
      r1_fmac8x2<0>(aA, aB, xin, wd0[0], wd0[1]);
      const float t = r1_reduce4(aA, 0.f, aB, 0.f);                       // rows 0: block 2 w's totals, 1: block 2 w + 1's
      if (lane < 32) {
        const int unit = 32 * wave + lane;
        lds[LM::r1 + r1_perm(unit & 63) * 4 + (unit >> 6)] = relu_keep_nan(t + BL[u.L[0].b_lds + unit]);
      }
      if constexpr (CLS == 0) {
        float rA = 0.f, rB = 0.f;
        r1_fmac8x2<0>(rA, rB, xin, w3f[0], w3f[1]);
        const float tr = r1_reduce4(rA, 0.f, rB, 0.f);
        if (lane < 32) lds[LM::res0 + lane] = tr + BL[u.L[3].b_lds + lane];
      }
      __syncthreads();
    }
    // stage 1: r2 = relu(down_1 r1 + b)            wave w: units 16 w .. 16 w + 15
    const f32x4 xr1 = *reinterpret_cast<const f32x4*>(lds + LM::r1 + lane * 4);
    const float bias1 = BL[u.L[1].b_lds + 16 * wave + n];
    if constexpr (EX) ex_begin();
    float s0 = 0.f, s1 = 0.f;
    blk(R1B(0), R1NX(1, 1), s0, s1, xr1[0]);
    blk(R1B(1), R1NX(2, 1), s0, s1, xr1[1]);
    blk(R1B(2), R1NX(3, 1), s0, s1, xr1[2]);
    blk(R1B(3), R1NX(4, 1), s0, s1, xr1[3]);
    {
      const float y = relu_keep_nan(r1_rows_sum(s0 + s1) + bias1);         // r2[16 w + n], in every row of the wave
      if (lane < 16) lds[LM::r2 + (16 * (n >> 2) + 4 * (wave & 3) + (n & 3)) * 2 + (wave >> 2)] = y;
      // stage 2 without a barrier of its own: down_2 is linear in r2, so the wave multiplies ITS 16 outputs into all 64
      // units right away (block 4: unit blocks 0..3 x input chunk w) and stage 3 adds the eight waves' partial sums.
      // Row g of the operand register holds r2[16 w + 4 g + (p & 3)] at every position p: one cross-lane gather.
      if constexpr (EX) {
        ex_d0 = (uint32_t)__builtin_amdgcn_ballot_w64(y > 0.f) & 0xFFFFu;
      }
      if constexpr (R1_S2DIRECT) {
        float c0 = 0.f, c1 = 0.f;
        pre(R1NX(5, 1));                        // (stage 3's first block, if it is an LDS block: read behind these fmacs)
        r1_fmac8<0>(c0, c1, y, wres[r1_count(CLS, DMX, 'R', 4)]);
        r1_fmac8<1>(c0, c1, y, wres[r1_count(CLS, DMX, 'R', 4)] + 8);
        lds[LM::p2 + wave * 64 + lane] = c0 + c1;                           // lane (g, n): unit 16 g + n
      } else {
      const float x4 = __shfl(y, (lane & 48) + 4 * g + (lane & 3));
      float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
      pre(R1NX(5, 1));                          // (stage 3's first block, if it is an LDS block: read behind these fmacs)
      r1_fmac16_4acc(c0, c1, c2, c3, x4, wres[r1_count(CLS, DMX, 'R', 4)]);
      const float tsum = r1_reduce4(c0, c1, c2, c3);                         // lane (g, n): unit 16 {0, 2, 1, 3}[g] + n
      lds[LM::p2 + wave * 64 + 16 * (((g & 1) << 1) | (g >> 1)) + n] = tsum;
      }
    }
    if constexpr (EX) {
      // R1 (every wave holds all of it): chunk c by the older wave of SIMD c -- waves 1, 2, 3 chunks 0, 1, 2 (their own slots' dwords 2, 3),
      // wave 0 chunk 3 (slot 4's).  The older waves reach this stage's barrier hundreds of cycles before the younger ones:
      // behind their last LDS write of the stage, in front of the barrier.
      if constexpr (CLS == 0) {
        ex_store(TR1{}, 64 * 3 + ex_pk, xr1[3]);
        ex_aux(4, __builtin_amdgcn_ballot_w64(xr1[3] > 0.f));
      } else if (wave <= 3) {
        const float v = wave == 1 ? xr1[0] : wave == 2 ? xr1[1] : xr1[2];
        ex_store(TR1{}, 64 * (wave - 1) + ex_pk, v);
        ex_b64 = __builtin_amdgcn_ballot_w64(v > 0.f);
      }
    }
    R1_TICK(1)
    __syncthreads();
    R1_TICK(2)
    const float2 xr2 = *reinterpret_cast<const float2*>(lds + LM::r2 + lane * 2);
    // stage 3: o2 = relu(up_2 r3 + b) + res_2 r2 + b   wave w: units 16 w ..   (res_2 first: r2 is in registers)
    const int pk = r1_perm(lane);
    // (down_2's eight partial sums per unit: four are requested in front of each res_2 block and folded behind it -- all
    //  eight at once held eight more registers live across both blocks)
    float p2a[4], p2b[4];
#pragma unroll
    for (int w8 = 0; w8 < 4; ++w8) p2a[w8] = lds[LM::p2 + w8 * 64 + pk];
    const float bias2 = BL[u.L[2].b_lds + pk];
    const float bu3 = BL[u.L[6].b_lds + 16 * wave + n], br3 = BL[u.L[5].b_lds + 16 * wave + n];
    float u0 = 0.f, u1 = 0.f, q0 = 0.f, q1 = 0.f;
    blk(R1B(5), R1NX(6, 1), q0, q1, xr2.x);
    const float p2lo = (p2a[0] + p2a[1]) + (p2a[2] + p2a[3]);
#pragma unroll
    for (int w8 = 0; w8 < 4; ++w8) p2b[w8] = lds[LM::p2 + (4 + w8) * 64 + pk];
    blk(R1B(6), R1NX(7, 1), q0, q1, xr2.y);
    const float xr3 = relu_keep_nan((p2lo + ((p2b[0] + p2b[1]) + (p2b[2] + p2b[3]))) + bias2);
    if constexpr (EX && CLS == 1) {
      if (wave == 1) {                            // R3 is formed by every wave: an older one writes it (slot 5's dwords 2, 3)
        ex_store(TR3{}, ex_pk, xr3);
        ex_aux(5, __builtin_amdgcn_ballot_w64(xr3 > 0.f));
      }
    }
    blk(R1B(7), R1NX(16, 2), u0, u1, xr3);
    {
      const float t = r1_reduce4(u0 + u1, q0 + q1, 0.f, 0.f);     // rows 0: up_2's totals, rows 2: res_2's
      float up, rs;
      r1_halves(t, up, rs);
      const float y = relu_keep_nan(up + bu3) + (rs + br3);
      if (lane < 16) lds[LM::o2 + (16 * (n >> 2) + 4 * (wave & 3) + (n & 3)) * 2 + (wave >> 2)] = y;
      if constexpr (EX) {
        ex_d0 |= ((uint32_t)__builtin_amdgcn_ballot_w64(up + bu3 > 0.f) & 0xFFFFu) << 16;
      }
    }
    R1_TICK(5)
    __syncthreads();
    R1_TICK(6)
    // stage 4: o1' = relu(up_1 o2 + b)   wave w: units 32 w .. 32 w + 31 (two neuron blocks).  The skip term res_1 r1 + b of
    // models.py:239 is not formed: it reaches the output only through up_0, as (up_0 res_1) r1 + up_0 b -- the FOLD, whose share of
    // this wave (r1 inputs 32 w .. 32 w + 31, in registers since stage 1) runs first and covers part of o2's LDS round trip
    const float2 xo2 = *reinterpret_cast<const float2*>(lds + LM::o2 + lane * 2);
    const float bu4 = BL[u.L[7].b_lds + 32 * wave + 16 * (g & 1) + n];
    float p0 = 0.f, p1 = 0.f, q0_ = 0.f, q1_ = 0.f;
    {
      const int C = wave >> 1;
      const float xc = C == 0 ? xr1[0] : C == 1 ? xr1[1] : C == 2 ? xr1[2] : xr1[3];
      if (wave & 1) {
        r1_fmac8<1>(p0, p1, xc, wf[0]);
        if constexpr (H == 2) r1_fmac8<1>(q0_, q1_, xc, wf[H - 1]);
      } else {
        r1_fmac8<0>(p0, p1, xc, wf[0]);
        if constexpr (H == 2) r1_fmac8<0>(q0_, q1_, xc, wf[H - 1]);
      }
    }
    // (the two accumulators of a block pair are the two neuron blocks': consecutive fmacs never depend on each other)
    float ua = 0.f, ub = 0.f;
    blk2(R1B(16), R1B(17), R1NX(18, 2), ua, ub, xo2.x);
    R1_TICK(14)
    blk2(R1B(18), R1B(19), std::integral_constant<int, kR1Blocks>{}, ua, ub, xo2.y);
    R1_TICK(15)
    float o1;
    {
      // rows 0: up_1 block 2w, 1: up_1 block 2w + 1
      const float t = r1_reduce4(ua, 0.f, ub, 0.f);
      float up, rs;
      r1_halves(t, up, rs);
      R1_TICK(7)
      o1 = relu_keep_nan(up + bu4);                       // lane (g, n): unit 32 w + 16 (g & 1) + n, both halves alike
    }
    if constexpr (EX) {
      ex_d1 = (uint32_t)__builtin_amdgcn_ballot_w64(o1 > 0.f);
      if constexpr (CLS == 0) {                   // (wave 0 waits for the younger waves at the coming barrier: here; waves 1 .. 7: ex_flush)
        if (lane < 32) ex_store(TO1{}, 32 * wave + lane, o1);
        ex_record();
      } else {
        ex_o1 = o1;
      }
    }
    // stage 5, this wave's share: up_0 over the wave's own 32 outputs (no barrier in between), on top of the fold's share
    {
      const float x5 = r1_ror8_upper(o1);
      r1_fmac8<0>(p0, p1, x5, w5[0]);
      const float y = r1_rows_sum(p0 + p1);
      if (lane < 16) lds[LM::p5 + n * 8 + wave] = y;
      if constexpr (H == 2) {                                              // the second 16-unit block of up_0's outputs
        r1_fmac8<0>(q0_, q1_, x5, w5[1]);
        const float yh = r1_rows_sum(q0_ + q1_);
        if (lane < 16) lds[LM::p5 + (16 + n) * 8 + wave] = yh;
      }
    }
    // The first wave of every SIMD reaches this barrier ~900 cycles before the second one (the older wave has issue
    // priority through the long stage 4): wave 0 closes the previous step's books in that slack, waves 1 .. 3 prepare the
    // next steps' noise and scalars -- none of it is left for the serial section behind the barrier.
    // (at the lowest issue priority: the SIMD's other wave is still inside stage 4 and everybody waits for IT)
    __builtin_amdgcn_s_setprio(0);
    // (closing the books in stage 3's slack instead was measured: 0.454 ms against 0.437 -- at the lowest priority the
    //  ~450-cycle chain of LDS reads and row sums outlasts that slack, and wave 0 becomes the wave stage 3 waits for)
    if constexpr (CLS == 0) {
      res0_update();
      bookkeeping();
      prepare_step(cur_k);
    } else {
      batches(cur_k);
      ou_products();
      h1_products();
    }
    __builtin_amdgcn_s_setprio(2);
    R1_TICK(8)
    __syncthreads();
    R1_TICK(9)
  };
  // Waves 1 .. 7 between the barrier that ends the network and the one behind wave 0's serial section (~700 idle cycles): their A1 units
  // and records, and -- from LDS, where every wave's share still stands -- R2 and O2 (their halves by waves 1, 2 and 3, 5: lane x of half j
  // holds unit 64 j + r1_perm(x)).  Wave 4 shares its SIMD with the integrating wave: its own two stores only, at the lowest priority.
  auto ex_flush = [&]() {
    if constexpr (EX && CLS == 1) {
      __builtin_amdgcn_s_setprio(0);
      if (lane < 32) ex_store(TO1{}, 32 * wave + lane, ex_o1);
      ex_record();
      if (wave == 1 || wave == 2) ex_store(TR2{}, 64 * (wave - 1) + ex_pk, lds[LM::r2 + lane * 2 + (wave - 1)]);
      if (wave == 3 || wave == 5) ex_store(TO2{}, 64 * (wave == 3 ? 0 : 1) + ex_pk, lds[LM::o2 + lane * 2 + (wave == 3 ? 0 : 1)]);
      __builtin_amdgcn_s_setprio(2);
    }
  };
  // nabla_V[i] in every row of wave 0 (the other waves: not used)
  auto network_output = [&]() -> float {
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8);
    const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8 + 4);
    const float s = ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]));
    if constexpr (H == 2) return relu_keep_nan(s + b8) + lds[LM::res0 + n];
    if constexpr (EX) bk_pre = s + b8;
    return relu_keep_nan(s + b8) + res0;
  };
  auto network_output_hi = [&]() -> float {                               // H = 2: unit 16 + n
    const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + (16 + n) * 8);
    const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + LM::p5 + (16 + n) * 8 + 4);
    const float s = ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]));
    return relu_keep_nan(s + b8h) + lds[LM::res0 + 16 + n];
  };

  // ---- prologue ----
  __builtin_amdgcn_s_setprio(2);
  noise_batch(0);
  scalar_batch(0);
  if constexpr (CLS == 0) first_layer(a.ts[0]);
  __syncthreads();
  if constexpr (CLS == 0) rollout_key_advance(a, key_offset);     // (every wave read the key in front of this barrier)
  __syncthreads();
  if constexpr (CLS == 0) prepare_step(0);
  // EXPORT: every load of the prologue (the resident weights, the key) has landed before the first step -- said HERE, once, in a
  // form the compiler's wait-count pass reads: without it the loop keeps conservative s_waitcnt vmcnt(0) in front of fmac blocks (a
  // weight register might still be in flight on the first trip), and with stores inside the loop those waits become real every step:
  // each one then waits for the export stores issued just before it (0.433 instead of 0.383 ms at configs[2])
  if constexpr (EX) __builtin_amdgcn_s_waitcnt(0x0F70);          // s_waitcnt vmcnt(0)
  for (int k = 0; k < K; ++k) {
#ifdef SOCMX_R1_PROF
    if (k == 0) prof_last = __builtin_readcyclecounter();
#endif
    cur_k = k;
    network();
    ex_flush();
    if constexpr (CLS == 0) {
      // every LDS operand of the chain is requested at once (left to the scheduler the three round trips ran one after the other)
      const f32x4 pa = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8);
      const f32x4 pb = *reinterpret_cast<const f32x4*>(lds + LM::p5 + n * 8 + 4);
      const f32x4 scal = *reinterpret_cast<const f32x4*>(SC + (k & 31) * 4);
      const float eps = NZ[(k % (3 * NBS)) * MS + i];                    // drawn a batch ago
      // t_{k+1} from the scalars' ring, not from memory: the kernel stores, so the compiler loads a.ts[] through the vector path,
      // and the one vmcnt counter made this read wait for the books' stores of the stage before (s_waitcnt vmcnt(0) on the chain)
      const float t1 = lds[LM::tsr + (k & 31)];
      float epsh = 0.f, gvh = 0.f;
      if constexpr (H == 2) {
        epsh = NZ[(k % (3 * NBS)) * MS + 16 + i];
        gvh = network_output_hi();
        if (is_ou) { pre_b = lds[LM::pb + i]; pre_bh = lds[LM::pb + 16 + i]; }
        apply_quad_cost();
      }
      const float res0v = H == 2 ? lds[LM::res0 + n] : res0;
      if constexpr (OFF) {
        if constexpr (is_ou) pre_b = lds[LM::pbx + i];
        if constexpr (DENSE) pre_se = lds[LM::psx + i];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float dt = scal[0], sq_ldt = scal[1];
      const float gpre = (((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]))) + b8;
      const float gv = relu_keep_nan(gpre) + res0v;
      if constexpr (EX) bk_pre = gpre;
      R1_TICK(12)
      // sigma u = -sigma sigma^T nabla_V: the one product that waits for the network (sigma = I: u = -nabla_V itself); the drift
      // and sigma eps were formed a stage ago (prepare_step)
      float su;
      if constexpr (DENSE) {
        float s0 = 0.f, s1 = 0.f;
        r1_state_one<DMAX>(s0, s1, gv, srow);
        su = lane_ok ? -(s0 + s1) : 0.f;
      } else {
        su = lane_ok ? -gv : 0.f;
      }
      const float upd = (pre_b + su) * dt + sq_ldt * pre_se;            // utils.py:45-47
      const float xn = x + stop * upd;                                  // utils.py:48
      const float updh = H == 2 ? (pre_bh + (okv[H - 1] ? -gvh : 0.f)) * dt + sq_ldt * pre_seh : 0.f;
      const float xnh = xh + stop * updh;
      float xeh = xnh;
      float xe = xn, step = dt, stop_new = 1.f;
      if (STOPPING) {                                                   // utils.py:42-44, 49-75; Phi = -x_0
        const float phi_b = -__shfl(x, 0, 16), phi_a = -__shfl(xn, 0, 16);
        const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
        const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
        const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
        xe = js * (x + fr * stop * upd) + (1.f - js) * xn;
        if constexpr (H == 2) xeh = js * (xh + fr * stop * updh) + (1.f - js) * xnh;
        step = js * (fr * fr) * dt + ns * dt;                           // step_fraction squared (utils.py:70-72)
        stop_new = (-__shfl(xe, 0, 16) > 0.f) ? 1.f : 0.f;
      }
      x = lane_ok ? xe : 0.f;
      if constexpr (OFF) {
        if (lane < 16) lds[LM::xk + i] = x;
      }
      if constexpr (H == 2) { xh = okv[1] ? xeh : 0.f; bk_gvh = gvh; bk_epsh = epsh; }
      if (STOPPING) stop = stop_new;
      bk_k = k; bk_gv = gv; bk_step = step; bk_eps = eps;               // costs + stores: see bookkeeping()
      R1_TICK(13)
      first_layer(t1);                                                  // the next evaluation's first layer: [t_{k+1}, x_{k+1}]
    }
    R1_TICK(10)
    __syncthreads();
    R1_TICK(11)
  }
#ifdef SOCMX_R1_PROF
  if (a.prof && blockIdx.x == 0 && lane == 0)
    for (int sl = 0; sl < 16; ++sl) a.prof[wave * 16 + sl] = prof_acc[sl];
#endif
  cur_k = K;                              // (nothing left to prepare)
  if (a.nabla_v) {                        // nabla_V(T, X_K): r1 already holds down_0 [t_K, x_K]
    network();                            // (wave 0 closes the last step's books inside)
    ex_flush();
    apply_quad_cost();
    if constexpr (CLS == 0) {
      const float gv = network_output();
      if (store) a.nabla_v[(size_t)K * B * d + rowoff] = gv;
      if constexpr (EX) ex_output_sign(K * B + grow, bk_pre);
      if constexpr (H == 2) {
        const float gvh = network_output_hi();
        if (store_h) a.nabla_v[(size_t)K * B * d + rowoff + 16] = gvh;
      }
    }
  }
  if constexpr (H == 2) {
    if (!a.nabla_v && is_quad) {          // (no terminal evaluation: x_K' P x_K still has to come from wave 3 -- the same code path)
      if constexpr (CLS == 0) bookkeeping();
      ou_products();
      __syncthreads();
      apply_quad_cost();
    }
  }
  if constexpr (CLS == 0) bookkeeping();  // (no terminal evaluation: the last step's costs and stores)
  if constexpr (CLS == 0) {                                             // terminal cost (utils.py:101)
    float gval = 0.f;
    if (kind == SOCMX_OU_QUADRATIC) {
      float qx = 0.f, qxh = 0.f;
      for (int j = 0; j < d; ++j) {
        const float xj = comp(x, xh, j);
        qx += a.Q[ic * d + j] * xj;
        if constexpr (H == 2) qxh += a.Q[icv[H - 1] * d + j] * xj;
      }
      float part = lane_ok ? x * qx : 0.f;
      if constexpr (H == 2) part += okv[1] ? xh * qxh : 0.f;
      gval = row16_sum(part);
    } else if (kind == SOCMX_OU_LINEAR) {
      float part = lane_ok ? a.omega[ic] * x : 0.f;
      if constexpr (H == 2) part += okv[1] ? a.omega[icv[1]] * xh : 0.f;
      gval = row16_sum(part);
    } else if (kind == SOCMX_DOUBLE_WELL) {
      const float q = x * x - 1.f;
      float part = lane_ok ? a.nu[ic] * (q * q) : 0.f;
      if constexpr (H == 2) {
        const float qh = xh * xh - 1.f;
        part += okv[1] ? a.nu[icv[1]] * (qh * qh) : 0.f;
      }
      gval = row16_sum(part);
    }
    if (lane == 0) {
      a.lpd[grow] = lpd;
      a.lps[grow] = lps;
      a.ltw[grow] = -gval / a.lmbd;
    }
  }
#undef R1B
}

template <int MODE, class NET, int DMAX, bool EXPORT = false>
__global__ __launch_bounds__(kR1Waves * 64) void rollout1_kernel(const RolloutArgs a) {  // MODE: see r1_wave
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  typedef typename R1LdsOf<(NET::outp >> 4)>::type LM;
  for (int e = tid; e < LM::bias; e += kR1Waves * 64) lds[e] = 0.f;
  unet_load_biases_at(a.packed, NET::desc(), lds + LM::bias, tid, kR1Waves * 64);
  __syncthreads();
  if (wave == 0) r1_wave<0, MODE, NET, DMAX, EXPORT>(a, lds, 0, lane);
  else r1_wave<1, MODE, NET, DMAX, EXPORT>(a, lds, wave, lane);
}

bool rollout1_available() { return r1_supported<DefaultNet>(); }
// 17 <= d <= 31 with sigma = I: the 32-wide input / output network (soc.yaml's default d = 20)
bool rollout1_wide_available() { return r1_supported<Wide32Net>(); }
int rollout1_wide_launch(const RolloutArgs& a, bool stopping, void* stream) {
  if constexpr (r1_supported<Wide32Net>()) {
    const bool ou = a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR;
    if (!a.sigma_identity || a.act_ws) return SOCMX_E_DIM;
    void (*k)(const RolloutArgs) = stopping ? rollout1_kernel<1, Wide32Net, 31> : ou ? rollout1_kernel<2, Wide32Net, 31> : rollout1_kernel<0, Wide32Net, 31>;
    if (const int err = ensure_max_lds(k)) return err;
    return launch(k, dim3((unsigned)a.B), dim3(kR1Waves * 64), (size_t)kLdsBytesPerCU, stream, a);
  } else {
    return SOCMX_E_DIM;
  }
}

int rollout1_launch(const RolloutArgs& a, bool stopping, void* stream) {
  if constexpr (r1_supported<DefaultNet>()) {
    void (*k)(const RolloutArgs);
    const bool ou = a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR;
    const bool dense = !a.sigma_identity;
    if (dense && stopping) return SOCMX_E_DIM;          // (the launcher does not send this combination here)
#define R1PICK(DM)                                                                                       \
  (stopping ? rollout1_kernel<1, DefaultNet, DM>                                                         \
            : dense ? (ou ? rollout1_kernel<6, DefaultNet, DM> : rollout1_kernel<4, DefaultNet, DM>)     \
                    : (ou ? rollout1_kernel<2, DefaultNet, DM> : rollout1_kernel<0, DefaultNet, DM>))
    if (a.d <= 3) k = R1PICK(3);
    else if (a.d <= 11) k = R1PICK(11);
    else k = R1PICK(15);
#undef R1PICK
    if (a.act_ws) {                                        // the activation export (the terminal evaluation included)
      if (!a.act_rec || !a.nabla_v) return SOCMX_E_DIM;
#define R1PICKX(DM)                                                                                                  \
  (stopping ? rollout1_kernel<1, DefaultNet, DM, true>                                                               \
            : dense ? (ou ? rollout1_kernel<6, DefaultNet, DM, true> : rollout1_kernel<4, DefaultNet, DM, true>)     \
                    : (ou ? rollout1_kernel<2, DefaultNet, DM, true> : rollout1_kernel<0, DefaultNet, DM, true>))
      if (a.d <= 3) k = R1PICKX(3);
      else if (a.d <= 11) k = R1PICKX(11);
      else k = R1PICKX(15);
#undef R1PICKX
    }
    if (const int err = ensure_max_lds(k)) return err;
    // (the CU's whole LDS: one workgroup per CU, nobody else's workgroups beside this latency-bound chain)
    return launch(k, dim3((unsigned)a.B), dim3(kR1Waves * 64), (size_t)kLdsBytesPerCU, stream, a);
  } else {
    return SOCMX_E_DIM;
  }
}

}  // namespace socmx
