// socmx_unet.h -- the control network (FullyConnectedUNet, reference models.py:202-242)
// as a chain of fp32 MFMA stages over ONE 16-row tile that lives in LDS.
//
// gfx950 only.  Every dense layer is D = A.B with v_mfma_f32_16x16x4_f32:
//   A (16 x 4)  = weights   : 16 output neurons x 4 inputs      (streamed from L2, fragment-ordered)
//   B (4 x 16)  = activations: 4 inputs x 16 batch rows          (ds_read_b128 from the LDS tile)
//   D (16 x 16) : lane l holds neurons 4*(l>>4)..+3 of batch row (l&15)  -> one ds_write_b128
// A wave keeps up to four independent 16-neuron accumulators so one activation fragment feeds
// four MFMAs and the 40-cycle dependent-accumulator latency is covered.
//
// Fragment order ("packed" image, built by socmx_unet_pack_f32): for layer L, neuron block nb,
// input chunk kc (16 inputs), lane l, component i:
//     packed[L.w_off + ((nb*KC + kc)*64 + l)*4 + i] = W[nb*16 + (l&15)][kc*16 + 4*(l>>4) + i]
// so MFMA k-step i of a chunk multiplies inputs {kc*16 + 4g + i : g=0..3}; the activation
// fragment X[row][kc*16 + 4g .. +3] (one 16-byte LDS read) has the same (g,i) indexing.
// Layer widths are zero-padded to multiples of 16 (zero weights/bias => padded units stay 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Padded hidden widths of the constexpr-specialised kernel instantiations (StaticNet<..., H0P, H1P, H2P, ...>).  The shipped
// library is built for the reference's default arch.hdims = [256, 128, 64] (configs/soc.yaml:33-35); a VARIANT library for
// another architecture is the same sources compiled with -DSOCMX_H0P=.. -DSOCMX_H1P=.. -DSOCMX_H2P=.. (csrc/Makefile
// VARIANT=h0_h1_h2; socmx/_lib.py builds and loads it at first use when backend.specialize_arch is on).
#ifndef SOCMX_H0P
#define SOCMX_H0P 256
#define SOCMX_H1P 128
#define SOCMX_H2P 64
#endif

namespace socmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LayerDesc {
  int w_off;   // float offset of the fragment-ordered weights inside the packed image
  int b_off;   // float offset of the (padded) bias
  int in_pad;  // padded fan-in  (multiple of 16)
  int out_pad; // padded fan-out (multiple of 16)
  int b_lds;   // float offset of this layer's bias inside the LDS bias copy (TileLayout::bias)
};

struct UnetDesc {
  int d;      // state dimension
  int in0;    // d + 1
  int in0p;   // pad16(d+1)
  int outp;   // pad16(d)
  int h[3];   // hdims
  int hp[3];  // padded hdims
  LayerDesc L[9];
  int total_floats; // the nine layers' fragments and biases (the part every kernel reads)
  int bias_floats;  // sum of padded fan-outs
  // Behind them, the FOLD: o1 = relu(up_1 o2 + b) + res_1 r1 + b reaches the next ReLU only through the LINEAR map up_0
  // (models.py:239-242), so   up_0 o1 = up_0 relu(up_1 o2 + b7) + (up_0 res_1) r1 + up_0 b4   exactly -- a forward-only kernel
  // that multiplies r1 by the (outp x h0) matrix  up_0 res_1  never needs res_1's h0 x h0 weights (39 % of the network's MACs and
  // bytes at the default widths).  The pack kernel forms it (fp64 accumulation, rounded once) whenever it re-lays the weights.
  LayerDesc fold;   // fragment-ordered like a layer: in_pad = hp[0], out_pad = outp; b_off: up_0 b4 (outp floats); b_lds unused
  // ... and for the MFMA tile kernels the CONCATENATED last layer  cat = [up_0 res_1 | up_0]  (outp x 2 hp[0]): with the R1 and O1
  // tiles side by side in LDS (TileLayout: one tile of row stride 2 hp[0] + 4) the last stage is ONE split-K GEMM over [r1 | o1']
  // (o1' = relu(up_1 o2 + b): stage 4 loses its 256 x 256 second GEMM) -- unet_stage_desc.  b_off = the fold's, b_lds = up_0's:
  // the kernels that run the folded program load b_up0 + up_0 b4 into that slot (unet_load_biases(..., fold_bias = true)).
  LayerDesc cat;
  int folded;       // 1: unet_stage_desc returns the folded program (16-wide outputs, or >= 64: the last stage's split / direct
                    // forms take any K; 32- and 48-wide outputs keep the plain program -- their slim split holds eight fragments)
  int image_floats; // total_floats + the fold + cat: the size of the packed image
};

__host__ __device__ constexpr int pad16(int x) { return (x + 15) & ~15; }

// layer (fan_in, fan_out) in SOCMX_L_* order
// socmx_rollout.hip: F = up_0 res_1 (and f = up_0 b4 when out_b is given) in MFMA fragment order; transposed: the image of F^T
// as a layer of h0 outputs and outp inputs (the backward chain, socmx_unet_bwd.hip).  Raw torch-layout weights in.
__attribute__((visibility("hidden"))) int unet_fold_launch(const float* up0, const float* res1, const float* b4, int h0, int dout,
                                                           int in_pad, int out_pad, float* out_w, float* out_b, int transposed,
                                                           void* stream, float* out_cat = nullptr);

inline void unet_layer_dims(int d, const int h[3], int fin[9], int fout[9]) {
  const int i0 = d + 1;
  fin[0] = i0;   fout[0] = h[0];  // down_0
  fin[1] = h[0]; fout[1] = h[1];  // down_1
  fin[2] = h[1]; fout[2] = h[2];  // down_2
  fin[3] = i0;   fout[3] = d;     // res_0
  fin[4] = h[0]; fout[4] = h[0];  // res_1
  fin[5] = h[1]; fout[5] = h[1];  // res_2
  fin[6] = h[2]; fout[6] = h[1];  // up_2
  fin[7] = h[1]; fout[7] = h[0];  // up_1
  fin[8] = h[0]; fout[8] = d;     // up_0
}

// Everything the kernels need depends only on the PADDED widths; this form is constexpr so that the
// specialised kernels (socmx_rollout.hip, StaticNet) fold every offset into an immediate.
__host__ __device__ constexpr UnetDesc make_unet_desc_padded(int d, int in0p, int h0p, int h1p, int h2p, int outp) {
  UnetDesc u{};
  u.d = d; u.in0 = d + 1; u.in0p = in0p; u.outp = outp;
  u.h[0] = h0p; u.h[1] = h1p; u.h[2] = h2p;
  u.hp[0] = h0p; u.hp[1] = h1p; u.hp[2] = h2p;
  const int fin[9] = {in0p, h0p, h1p, in0p, h0p, h1p, h2p, h1p, h0p};   // SOCMX_L_* order
  const int fout[9] = {h0p, h1p, h2p, outp, h0p, h1p, h1p, h0p, outp};
  int off = 0, boff = 0;
  for (int l = 0; l < 9; ++l) {
    u.L[l].in_pad = fin[l];
    u.L[l].out_pad = fout[l];
    u.L[l].w_off = off; off += fin[l] * fout[l];
    u.L[l].b_off = off; off += fout[l];
    u.L[l].b_lds = boff; boff += fout[l];
  }
  u.total_floats = off;
  u.bias_floats = boff;
  u.fold.in_pad = h0p; u.fold.out_pad = outp; u.fold.b_lds = 0;
  u.fold.w_off = off; off += h0p * outp;
  u.fold.b_off = off; off += outp;
  u.cat.in_pad = 2 * h0p; u.cat.out_pad = outp; u.cat.b_lds = u.L[8].b_lds; u.cat.b_off = u.fold.b_off;
  u.cat.w_off = off; off += 2 * h0p * outp;
  u.folded = (outp == 16 || outp >= 64) ? 1 : 0;
  u.image_floats = off;
  return u;
}

inline UnetDesc make_unet_desc(int d, const int h[3]) {
  UnetDesc u = make_unet_desc_padded(d, pad16(d + 1), pad16(h[0]), pad16(h[1]), pad16(h[2]), pad16(d));
  for (int i = 0; i < 3; ++i) u.h[i] = h[i];
  return u;
}

// LDS tile of one 16-row block.  Row strides are width+4 floats (16-byte aligned rows,
// conflict-free 16-byte writes, <=2-way on the fragment reads).
struct TileLayout {
  int s0, s1, s2, s3, sg;                    // strides of X0, R1/O1, R2/O2, R3, GV
  int x0, r1, r2, r3, o2, o1, gv, scratch;   // float offsets
  int bias;                                  // float offset of the LDS copy of all (padded) biases
  int floats;                                // total
};

__host__ __device__ constexpr int r4_stride(int width) { return width + ((16 - width % 64) + 64) % 64; }   // = 16 (mod 64)
// (rows = 4: the small-batch tile of the 4x4x1 kernels below)
__host__ __device__ constexpr TileLayout make_tile_layout(const UnetDesc& u, int nwaves, int rows = 16) {
  TileLayout t{};
  t.s0 = u.in0p + 4; t.s1 = u.hp[0] + 4; t.s2 = u.hp[1] + 4; t.s3 = u.hp[2] + 4; t.sg = u.outp + 4;
  if (rows == 4) {
    // 4-row tile: a ds_read_b128 of the activation operand touches (row l & 3, k-group l >> 4), 16 bytes each -- row strides
    // of 16 (mod 64) dwords put the 4 x 4 chunks on 64 distinct banks (width + 4 left rows one chunk apart: row j + 1,
    // k-group g on the banks of row j, k-group g + 1 -- 2.0k conflict cycles per step, SQ_LDS_BANK_CONFLICT)
    t.s0 = r4_stride(u.in0p); t.s1 = r4_stride(u.hp[0]); t.s2 = r4_stride(u.hp[1]); t.s3 = r4_stride(u.hp[2]);
    t.sg = r4_stride(u.outp);
  }
  // R1 and O1 share ONE tile: row = [r1 (hp0) | o1 (hp0) | 4] -- the folded last stage reads the row as a 2 hp0-wide operand
  t.s1 = rows == 4 ? r4_stride(2 * u.hp[0]) : 2 * u.hp[0] + 4;
  int off = 0;
  t.x0 = off; off += rows * t.s0;
  t.r1 = off; t.o1 = off + u.hp[0]; off += rows * t.s1;
  t.r2 = off; off += rows * t.s2;
  t.r3 = off; off += rows * t.s3;
  t.o2 = off; off += rows * t.s2;
  t.gv = off; off += rows * t.sg;
  t.scratch = off; off += 2 * rows * 16 * nwaves;  // split-K partials: 2 GEMMs x (parts*out_pad <= 16*nwaves) x rows
  t.bias = off; off += u.bias_floats;
  t.floats = off;
  return t;
}

// One network stage: Y = relu(L1 . X1 + b1) [+ L2 . X2 + b2]; Ln = GEMM 1 of the stage that follows (its first
// weight fragments are requested before this stage's closing barrier).  LDS operands as float offsets.
struct StageDesc {
  LayerDesc L1, L2, Ln;
  int x1, s1, x2, s2, y, sy, has2;
};

// The six stages of FullyConnectedUNet.forward (models.py:233-242) as a table, so that the kernels run ONE
// copy of the stage code in a loop (the fully inlined form is ~130 KB of ISA and thrashes the 64 KB I-cache).
//
// `ww[stage][wave]` is the work split, precomputed on the host so that the kernel does no integer division per
// stage (a 32-bit division is ~25 dependent instructions on this ISA; eight of them per stage cost more than
// the small layers' MFMAs):
//   direct stages (neuron blocks >= waves): wave owns blocks blk0, blk0+NW, ... (cnt of them), full K range;
//   split-K stages (fewer blocks than waves): wave owns block blk0, chunk range [kc0a,kc1a) / [kc0b,kc1b).
//   pf[4]: fragment numbers (block*KC + chunk) of the NEXT stage's first four GEMM-1 fragments for this wave.
constexpr int kMaxWaves = 8;
constexpr int kSimds = 4;      // SIMDs (MFMA pipes) per CU; wave w runs on SIMD w % 4
struct WaveWork {
  unsigned short split, blk0, cnt, active;
  unsigned short kc0a, kc1a, kc0b, kc1b;
  unsigned short part, parts, use_pre, pad;
  unsigned short pf[8];
};

struct UnetProgram {
  StageDesc st[6];
  WaveWork ww[6][kMaxWaves];
};

__host__ __device__ constexpr StageDesc make_stage(const UnetDesc& u, int l1, int x1, int s1, int has2, int l2, int x2,
                                                   int s2, int y, int sy, int ln) {
  StageDesc d{};
  d.L1 = u.L[l1]; d.L2 = u.L[l2]; d.Ln = u.L[ln];
  d.x1 = x1; d.s1 = s1; d.x2 = x2; d.s2 = s2; d.y = y; d.sy = sy; d.has2 = has2;
  return d;
}

// stage i of the network (0..5); constexpr so the specialised kernels see immediates
__host__ __device__ constexpr StageDesc unet_stage_desc(const UnetDesc& u, const TileLayout& t, int i) {
  if (u.folded && i >= 4) {
    // THE FOLD (UnetDesc::cat): stage 4 is o1' = relu(up_1 o2 + b) alone; stage 5 multiplies [r1 | o1'] by [up_0 res_1 | up_0]
    if (i == 4) {
      StageDesc d = make_stage(u, 7, t.o2, t.s2, 0, 7, t.o2, t.s2, t.o1, t.s1, 8);
      d.Ln = u.cat;
      return d;
    }
    StageDesc d = make_stage(u, 8, t.r1, t.s1, 1, 3, t.x0, t.s0, t.gv, t.sg, 0);
    d.L1 = u.cat;
    return d;
  }
  switch (i) {
    case 0: return make_stage(u, 0, t.x0, t.s0, 0, 0, t.x0, t.s0, t.r1, t.s1, 1);   // r1 = relu(down_0 x)
    case 1: return make_stage(u, 1, t.r1, t.s1, 0, 1, t.r1, t.s1, t.r2, t.s2, 2);   // r2 = relu(down_1 r1)
    case 2: return make_stage(u, 2, t.r2, t.s2, 0, 2, t.r2, t.s2, t.r3, t.s3, 6);   // r3 = relu(down_2 r2)
    case 3: return make_stage(u, 6, t.r3, t.s3, 1, 5, t.r2, t.s2, t.o2, t.s2, 7);   // o2 = relu(up_2 r3) + res_2 r2
    case 4: return make_stage(u, 7, t.o2, t.s2, 1, 4, t.r1, t.s1, t.o1, t.s1, 8);   // o1 = relu(up_1 o2) + res_1 r1
    default: return make_stage(u, 8, t.o1, t.s1, 1, 3, t.x0, t.s0, t.gv, t.sg, 0);  // o0 = relu(up_0 o1) + res_0 x; then down_0
  }
}

__host__ __device__ inline UnetProgram make_unet_program(const UnetDesc& u, const TileLayout& t) {
  UnetProgram p;
  for (int i = 0; i < 6; ++i) p.st[i] = unet_stage_desc(u, t, i);
  return p;
}

// first eight GEMM-1 fragments wave `w` consumes in a stage whose GEMM 1 is layer Lg
__host__ __device__ __attribute__((always_inline)) constexpr bool first_fragment_numbers(const LayerDesc& Lg, int NW, int w, unsigned short (&pf)[8]) {
  const int NBLK = Lg.out_pad >> 4, KC = Lg.in_pad >> 4;
  int blk = 0, bstride = 0, nb = 1, kc0 = 0, kc1 = 1;
  bool ok = false;
  if (NBLK >= NW || NBLK >= kSimds) {   // one block per SIMD already saturates the four MFMA pipes: no split-K
    int cnt = w < NBLK ? (NBLK - w + NW - 1) / NW : 0;
    // (blocks divide evenly over the waves: the count does not depend on the wave -- said explicitly, so that with constant
    //  NBLK / NW the divisions by `nb` below fold away: evaluated with a run-time wave id they were ~100 instructions of
    //  float-reciprocal division per stage in front of the barrier of the control-network backward)
    if (NBLK >= NW && NBLK % NW == 0) cnt = NBLK / NW;
    nb = cnt > 4 ? 4 : cnt;
    ok = cnt > 0 && nb != 3;                 // a 3-block group ignores the prefetch
    if (nb < 1) nb = 1;
    blk = w < NBLK ? w : 0; bstride = NW; kc0 = 0; kc1 = KC;
  } else {
    const int parts = NW / NBLK, part = w / NBLK;
    blk = w % NBLK; bstride = 0; nb = 1;
    kc0 = (part * KC) / parts; kc1 = ((part + 1) * KC) / parts;
    ok = part < parts && kc1 > kc0;
    if (!ok) { kc0 = 0; kc1 = 1; }
  }
#pragma unroll
  for (int f = 0; f < 8; ++f) {
    int kc = kc0 + f / nb; if (kc > kc1 - 1) kc = kc1 - 1;
    int b = blk + (f % nb) * bstride; if (b > NBLK - 1) b = NBLK - 1;
    pf[f] = (unsigned short)(b * KC + kc);
  }
  return ok;
}

// work split of wave w in stage d (constexpr: the specialised kernels evaluate it with constant d and NW, so
// only shifts/masks of the wave id remain)
__host__ __device__ constexpr WaveWork wave_work_of(const StageDesc& d, int NW, int w) {
  WaveWork x{};
  const int NBLK = d.L1.out_pad >> 4, KC1 = d.L1.in_pad >> 4, KC2 = d.L2.in_pad >> 4;
  if (NBLK >= NW || NBLK >= kSimds) {   // one block per SIMD already saturates the four MFMA pipes: no split-K
    x.split = 0; x.blk0 = (unsigned short)w;
    x.cnt = (unsigned short)((NBLK >= NW && NBLK % NW == 0) ? NBLK / NW : (w < NBLK ? (NBLK - w + NW - 1) / NW : 0));
    x.active = x.cnt > 0;
    x.kc0a = 0; x.kc1a = (unsigned short)KC1; x.kc0b = 0; x.kc1b = (unsigned short)KC2;
  } else {
    const int parts = NW / NBLK, part = w / NBLK;
    x.split = 1; x.blk0 = (unsigned short)(w % NBLK); x.cnt = 1;
    x.parts = (unsigned short)parts; x.part = (unsigned short)part; x.active = part < parts;
    if (x.active) {
      x.kc0a = (unsigned short)((part * KC1) / parts); x.kc1a = (unsigned short)(((part + 1) * KC1) / parts);
      x.kc0b = (unsigned short)((part * KC2) / parts); x.kc1b = (unsigned short)(((part + 1) * KC2) / parts);
    }
  }
  unsigned short tmp[8] = {};
  x.use_pre = first_fragment_numbers(d.L1, NW, w, tmp);   // does THIS stage consume a prefetch?
  first_fragment_numbers(d.Ln, NW, w, x.pf);              // what to request for the NEXT stage
  return x;
}

inline void fill_wave_work(UnetProgram& p, int NW) {
  for (int si = 0; si < 6; ++si)
    for (int w = 0; w < NW; ++w) p.ww[si][w] = wave_work_of(p.st[si], NW, w);
}

// ---- tensors that travel from kernel A to kernel B, each [tile][width][16 rows] ---------------------------------------
enum { T_X = 0, T_R1, T_R2, T_R3, T_O2, T_O1, T_G0, T_ZU0, T_GO1, T_ZU1, T_GO2, T_ZU2, T_ZD2, T_ZD1, T_ZD0, T_N };

__host__ __device__ constexpr int tensor_width(const UnetDesc& u, int t) {
  switch (t) {
    case T_X: return u.in0p;
    case T_R1: case T_O1: case T_GO1: case T_ZU1: case T_ZD0: return u.hp[0];
    case T_R2: case T_O2: case T_GO2: case T_ZU2: case T_ZD1: return u.hp[1];
    case T_R3: case T_ZD2: return u.hp[2];
    default: return u.outp;   // T_G0, T_ZU0
  }
}
__host__ __device__ constexpr int tensor_prefix(const UnetDesc& u, int t) {   // sum of the widths before tensor t
  int s = 0;
  for (int i = 0; i < t; ++i) s += tensor_width(u, i);
  return s;
}
// ---- what the ONE-ROW rollout saves for the control-network backward (socmx_rollout_ex_f32: act_workspace / act_records) ----------
// The rollout evaluates the network on every trajectory row anyway: with the activation slabs X.. O1 written by it (the layout above,
// straight into the backward's workspace) and the ReLU signs of a row in a 128-byte RECORD, kernel A runs its five backward stages only.
// Record of row r: eight waves x four dwords, rec[(r * 8 + w) * 4 + j]:
//   j = 0: bits 0..15  sign of R2[16 w + n],  bits 16..31  sign of the up-path pre-activation of O2[16 w + n]   (n = bit & 15)
//   j = 1: sign of A1[32 w + bit] = relu(up_1 O2 + b)
//   j = 2, 3 (64 bits, LANE order: bit x <-> element r1_perm(x)): waves 1..4: R1[64 (w - 1) + r1_perm(x)];  wave 5: R3[r1_perm(x)]
//   wave 0, j = 2: bits 0..15 sign of the output's pre-activation (up_0 A1 + F R1 + f + b)[bit]
enum { MK_NONE = 0, MK_R1, MK_R2, MK_R3, MK_U2, MK_U1 };
constexpr int kActRecordDwords = 32;
// the nibble (units n0 .. n0 + 3, n0 a multiple of 4) of mask `mk` in the record of one row
__host__ __device__ inline unsigned act_record_nibble(const uint32_t* rec, int mk, int n0) {
  if (mk == MK_R2) return (rec[(n0 >> 4) * 4] >> (n0 & 12)) & 0xFu;
  if (mk == MK_U2) return (rec[(n0 >> 4) * 4] >> (16 + (n0 & 12))) & 0xFu;
  if (mk == MK_U1) return (rec[(n0 >> 5) * 4 + 1] >> (n0 & 31)) & 0xFu;
  const int pos = 16 * ((n0 >> 2) & 3) + 4 * ((n0 >> 4) & 3);           // element 16 a + 4 b + i sits in lane 16 b + 4 a + i
  const int w = mk == MK_R1 ? 1 + (n0 >> 6) : 5;
  return (rec[w * 4 + 2 + (pos >> 5)] >> (pos & 31)) & 0xFu;
}

#if defined(__HIPCC__)

__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

// the same for an accumulator quad: four compares into four SGPR pairs, then four selects.  The compiler's form goes
// through VCC (v_cmp / s_nop / v_cndmask per element: a wait state after every compare), and this sits on the serial
// path between a stage's last MFMA and its ds_write.  NaN stays NaN (not (x < 0) selects x).
__device__ __forceinline__ void relu4_keep_nan(f32x4& v) {
  unsigned long long m0, m1, m2, m3;
  float a = v[0], b = v[1], c = v[2], d = v[3];
  // (the compiler does not look inside inline asm when it inserts the wait states an MFMA result needs before a
  //  VALU instruction may read it -- 11 for the 8-pass 16x16x4 -- so they are spelled out here)
  asm volatile(
      "s_nop 7\n\t"
      "s_nop 2\n\t"
      "v_cmp_ngt_f32_e64 %4, 0, %0\n\t"
      "v_cmp_ngt_f32_e64 %5, 0, %1\n\t"
      "v_cmp_ngt_f32_e64 %6, 0, %2\n\t"
      "v_cmp_ngt_f32_e64 %7, 0, %3\n\t"
      "v_cndmask_b32_e64 %0, 0, %0, %4\n\t"
      "v_cndmask_b32_e64 %1, 0, %1, %5\n\t"
      "v_cndmask_b32_e64 %2, 0, %2, %6\n\t"
      "v_cndmask_b32_e64 %3, 0, %3, %7"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3));
  v = f32x4{a, b, c, d};
}

// all threads: copy the padded biases of the nine layers from the packed image into LDS (once per kernel)
// fold_bias: up_0's slot gets b_up0 + up_0 b_res1 (the kernels that run the folded program of unet_stage_desc)
__device__ __forceinline__ void unet_load_biases(const float* __restrict__ Wp, const UnetDesc& u, const TileLayout& t,
                                                 float* lds, int tid, int nthr, bool fold_bias = false) {
  // (unrolled: a runtime index into u.L[] would put the whole descriptor into scratch memory)
#pragma unroll
  for (int l = 0; l < 9; ++l)
    for (int e = tid; e < u.L[l].out_pad; e += nthr) {
      float v = Wp[u.L[l].b_off + e];
      if (l == 8 && fold_bias) v += Wp[u.fold.b_off + e];
      lds[t.bias + u.L[l].b_lds + e] = v;
    }
}

// the same copy to an explicit LDS address (bias of layer l at dst + u.L[l].b_lds)
__device__ __forceinline__ void unet_load_biases_at(const float* __restrict__ Wp, const UnetDesc& u, float* dst, int tid,
                                                    int nthr) {
#pragma unroll
  for (int l = 0; l < 9; ++l)
    for (int e = tid; e < u.L[l].out_pad; e += nthr) dst[u.L[l].b_lds + e] = Wp[u.L[l].b_off + e];
}

// R4: the 4-row tile.  The SAME weight fragment feeds v_mfma_f32_4x4x1_16b_f32: its 16 blocks are lane quads, block
// (l >> 2) = (k-group kg = l >> 4, neuron group ng = (l >> 2) & 3); lane l's A value is W[16 nb + (l & 15)][16 kc + 4 kg + i]
// as before, its B value x[row l & 3][16 kc + 4 kg + i], and D = neurons 4 ng .. 4 ng + 3 of row (l & 3), summed over
// k-group kg's inputs only -- the four k-groups' partial sums are added across lanes when a GEMM ends (kg_reduce).
// 16 cycles per instruction (half the 16x16x4 rate per MAC), a quarter of the rows: half the time per tile.
template <int NB, bool R4 = false>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[NB], const f32x4 (&a)[NB], const f32x4 bx) {
  // k-step outer, block inner: consecutive MFMAs hit different accumulators (the dependent-accumulator latency
  // of v_mfma_f32_16x16x4_f32 is 40 cycles against a 32-cycle issue interval)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (R4) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j][i], bx[i], acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], bx[i], acc[j], 0, 0, 0);
    }
}

// Sum over the four k-groups (lanes l, l ^ 16, l ^ 32, l ^ 48) of a 4-row accumulator as a butterfly that also SCATTERS:
// gfx950's v_permlane32_swap (upper half of the first register <-> lower half of the second) pairs components (0, 1) and
// (2, 3), one add each folds k-groups g and g + 2; v_permlane16_swap (odd rows of the first <-> even rows of the second)
// and one more add fold the rest -- six VALU instructions for the four registers, and lane l is left with the TOTAL of
// ONE component: kg_comp(l) = {0, 2, 1, 3}[l >> 4], i.e. neuron 4 ng + kg_comp of row l & 3 (every total exactly once
// over the wave: the stage writes one float per lane).  Inline asm: with both operands of the swap builtins the same value
// this compiler folds the result pair into one register; the leading s_nop covers the MFMA -> VALU wait states of the
// accumulator (not inserted in front of inline asm).  Checked against the CPU by tools/ubench/r4_check.hip.
__device__ __forceinline__ float kg_reduce(const f32x4 v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  asm volatile(
      "s_nop 7\n\t"
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
__device__ __forceinline__ int kg_comp(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }   // {0, 2, 1, 3}[lane >> 4]

// Weight fragments come straight from L2 into VGPRs (each is used by exactly one MFMA group of one
// wave, so an LDS round trip would be pure overhead).  L2 latency is ~500-1000 cycles while one chunk is
// only NB x 4 MFMAs, so PD chunks are kept in flight in a statically indexed register ring: slot s is
// refilled with chunk kc+PD right after chunk kc used it.  The first fragments of a GEMM can arrive
// through `pre` (4 fragments requested by the PREVIOUS stage before its barrier), which takes the L2
// latency off the stage-to-stage critical path.
// the four prefetched fragments handed from one stage to the next (by value: stays in VGPRs)
struct Pre {
  f32x4 f[8];   // PD*NB = 8 for NB in {1,2,4}: the whole first ring of the next stage's GEMM 1
};

template <int NB>
struct Ring {
  static constexpr int PD = (NB >= 4) ? 2 : (NB >= 2 ? 4 : 8);   // chunks in flight (PD*NB*4 VGPRs)
  static constexpr int CP = (NB == 3) ? 0 : 8 / NB;              // chunks covered by the 8 `pre` fragments (= PD)
  f32x4 slot[PD][NB];
};

// A lane's view of a run of 1-KiB weight fragments in a packed image, read with BUFFER loads: the lane's 16-byte slot sits in one
// 32-bit offset register (the same for every load of the wave), the fragment's position in the instruction's scalar offset.
// Written as `base[(size_t)n * 64]` like the f32x4 pointer it replaces (n fragments on): with global_load_dwordx4 every
// fragment load carried a 64-bit address per lane, and issuing those is time the matrix pipe does not get back -- the
// two-tile burst kernel ran 6.5 % faster for this change alone (socmx_rollout32.hip), the same streams everywhere else.
struct FragBase {
  __amdgpu_buffer_rsrc_t rsrc;
  int lane_off;   // bytes
  int fo;         // float offset of fragment 0 (wave-uniform)
  __device__ __forceinline__ f32x4 operator[](size_t i) const {      // i in f32x4 units (whole fragments: a multiple of 64)
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, fo * 4 + (int)i * 16, 0));
  }
  __device__ __forceinline__ FragBase operator+(size_t i) const {
    FragBase r = *this;
    r.fo += (int)i * 4;
    return r;
  }
};
// (num_records = 2 GiB: the image's extent is the caller's contract, as it was for the pointer)
__device__ __forceinline__ FragBase frag_base(const float* __restrict__ Wp, int float_off, int lane) {
  FragBase b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wp), 0, 0x7FFFFFFF, 0x00020000);
  b.lane_off = lane * 16;
  b.fo = float_off;
  return b;
}

template <int NB>
struct GemmPlan {
  FragBase wb[NB];  // per block: fragment base, indexable by chunk*64
  const float* xrow;     // this lane's activation row (+4g)
  int kc0, kc1;
};

template <int NB, bool R4 = false>
__device__ __forceinline__ GemmPlan<NB> make_plan(const float* __restrict__ Wp, const LayerDesc& L, int blk0, int bstride,
                                                  const float* X, int S, int lane, int kc0, int kc1) {
  GemmPlan<NB> p;
  const int KC = L.in_pad >> 4;
  const FragBase wl = frag_base(Wp, L.w_off, lane);
#pragma unroll
  for (int j = 0; j < NB; ++j) p.wb[j] = wl + (size_t)((blk0 + j * bstride) * KC) * 64;
  p.xrow = X + (R4 ? (lane & 3) : (lane & 15)) * S + 4 * (lane >> 4);
  p.kc0 = kc0; p.kc1 = kc1;
  return p;
}

// issue the loads of ring slots [from, PD); slots [0, from) are taken from `pre` (from == CP) or loaded too (from == 0)
template <int NB, bool USE_PRE, class RingT = Ring<NB>>
__device__ __forceinline__ void ring_fill(RingT& r, const GemmPlan<NB>& p, const Pre& pre) {
  constexpr int PD = RingT::PD, CP = USE_PRE ? RingT::CP : 0;
  const int last = p.kc1 - 1;
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    if (s < CP) {
#pragma unroll
      for (int j = 0; j < NB; ++j) r.slot[s][j] = pre.f[s * NB + j];
    } else {
      const int kc = min(p.kc0 + s, last);
#pragma unroll
      for (int j = 0; j < NB; ++j) r.slot[s][j] = p.wb[j][(size_t)kc * 64];
    }
  }
}

// PINNED (the constexpr-specialised stages, fully unrolled): refill loads fenced by scheduling barriers and skipped
// past the end of the GEMM.  The table-driven stages keep rolled loops with runtime bounds, where the fences and the
// extra branch cost more than they give (3.78 -> 4.00 ms on cfg3): they re-request the last chunk instead.
template <int NB, bool PINNED = false, bool R4 = false, class RingT = Ring<NB>>
__device__ __forceinline__ void gemm_run(f32x4 (&acc)[NB], RingT& r, const GemmPlan<NB>& p) {
  constexpr int PD = RingT::PD;
  const int last = p.kc1 - 1;
  int kc = p.kc0;
  // activations (B operand) are read from LDS two chunks ahead of their MFMAs
  f32x4 bx = *reinterpret_cast<const f32x4*>(p.xrow + min(kc, last) * 16);
  f32x4 bx_next = *reinterpret_cast<const f32x4*>(p.xrow + min(kc + 1, last) * 16);
  for (; kc + PD <= p.kc1; kc += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const f32x4 bx_next2 = *reinterpret_cast<const f32x4*>(p.xrow + min(kc + s + 2, last) * 16);
      mfma_chunk<NB, R4>(acc, r.slot[s], bx);
      // Refill the slot just consumed with chunk kc+s+PD (nothing to fetch past the end of the GEMM).  The scheduling
      // barriers keep the loads HERE: left alone, the scheduler sinks them towards their use and the ring that
      // should hold PD chunks in flight ends up ~2 deep (s_waitcnt vmcnt(2..3) in front of the MFMAs).
      if constexpr (PINNED) {
        __builtin_amdgcn_sched_barrier(0);
        if (kc + s + PD <= last) {
#pragma unroll
          for (int j = 0; j < NB; ++j) r.slot[s][j] = p.wb[j][(size_t)(kc + s + PD) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const int nk = min(kc + s + PD, last);
#pragma unroll
        for (int j = 0; j < NB; ++j) r.slot[s][j] = p.wb[j][(size_t)nk * 64];
      }
      bx = bx_next;
      bx_next = bx_next2;
    }
  }
  // tail: fewer than PD chunks left; they already sit in slots 0..rem-1
#pragma unroll
  for (int s = 0; s < PD - 1; ++s) {
    if (kc + s < p.kc1) {
      const f32x4 bx_next2 = *reinterpret_cast<const f32x4*>(p.xrow + min(kc + s + 2, last) * 16);
      mfma_chunk<NB, R4>(acc, r.slot[s], bx);
      bx = bx_next;
      bx_next = bx_next2;
    }
  }
}

// ---- descriptor residency ---------------------------------------------------------------------------
// The stage loop reads its StageDesc / WaveWork from the kernarg segment.  Left alone, the compiler loads
// each field lazily in the basic block that first needs it: five to eight serial `s_load` -> `s_waitcnt`
// round trips (~200 cycles each) per stage.  Pinning every field in an SGPR at the top of the stage makes
// it fetch the whole descriptor in one batch.
#define SOCMX_PIN(x) asm volatile("" : "+s"(x))

struct WaveWorkS {  // WaveWork widened to SGPR-resident ints
  int split, blk0, cnt, active, kc0a, kc1a, kc0b, kc1b, part, parts, use_pre;
  int pf[8];
};

__device__ __forceinline__ void pin_layer(LayerDesc& L) {
  SOCMX_PIN(L.w_off); SOCMX_PIN(L.b_off); SOCMX_PIN(L.in_pad); SOCMX_PIN(L.out_pad); SOCMX_PIN(L.b_lds);
}

__device__ __forceinline__ StageDesc load_stage(const StageDesc& src) {
  StageDesc d = src;
  pin_layer(d.L1); pin_layer(d.L2); pin_layer(d.Ln);
  SOCMX_PIN(d.x1); SOCMX_PIN(d.s1); SOCMX_PIN(d.x2); SOCMX_PIN(d.s2); SOCMX_PIN(d.y); SOCMX_PIN(d.sy); SOCMX_PIN(d.has2);
  return d;
}

__device__ __forceinline__ WaveWorkS load_work(const WaveWork& src) {
  WaveWorkS w;
  w.split = src.split; w.blk0 = src.blk0; w.cnt = src.cnt; w.active = src.active;
  w.kc0a = src.kc0a; w.kc1a = src.kc1a; w.kc0b = src.kc0b; w.kc1b = src.kc1b;
  w.part = src.part; w.parts = src.parts; w.use_pre = src.use_pre;
#pragma unroll
  for (int f = 0; f < 8; ++f) w.pf[f] = src.pf[f];
  SOCMX_PIN(w.split); SOCMX_PIN(w.blk0); SOCMX_PIN(w.cnt); SOCMX_PIN(w.active);
  SOCMX_PIN(w.kc0a); SOCMX_PIN(w.kc1a); SOCMX_PIN(w.kc0b); SOCMX_PIN(w.kc1b);
  SOCMX_PIN(w.part); SOCMX_PIN(w.parts); SOCMX_PIN(w.use_pre);
  SOCMX_PIN(w.pf[0]); SOCMX_PIN(w.pf[1]); SOCMX_PIN(w.pf[2]); SOCMX_PIN(w.pf[3]);
  SOCMX_PIN(w.pf[4]); SOCMX_PIN(w.pf[5]); SOCMX_PIN(w.pf[6]); SOCMX_PIN(w.pf[7]);
  return w;
}

// request the eight fragments numbered pf[0..7] of layer Lg (GEMM 1 of the stage that follows)
__device__ __forceinline__ Pre prefetch_fragments(const float* __restrict__ Wp, const LayerDesc& Lg, const WaveWorkS& w,
                                                  int lane) {
  const FragBase wl = frag_base(Wp, Lg.w_off, lane);
  Pre pre;
#pragma unroll
  for (int f = 0; f < 8; ++f) pre.f[f] = wl[(size_t)w.pf[f] * 64];
  return pre;
}

template <int NB, int NW, typename Hook, bool PINNED = false>
__device__ __forceinline__ void stage_direct(const float* __restrict__ Wp, const float* bias_lds,
                                             const LayerDesc& L1, const float* X1, int S1, bool has2,
                                             const LayerDesc& L2, const float* X2, int S2, float* Y, int SY,
                                             int blk0, int lane, const Pre& pre, bool use_pre, Hook hook) {
  const int row = lane & 15, g = lane >> 4;
  const GemmPlan<NB> p1 = make_plan<NB>(Wp, L1, blk0, NW, X1, S1, lane, 0, L1.in_pad >> 4);
  const GemmPlan<NB> p2 = make_plan<NB>(Wp, L2, blk0, NW, X2, S2, lane, 0, L2.in_pad >> 4);
  Ring<NB> r1, r2;
  if (use_pre) ring_fill<NB, true>(r1, p1, pre); else ring_fill<NB, false>(r1, p1, pre);
  if (has2) ring_fill<NB, false>(r2, p2, pre);   // the residual GEMM's first chunks fly while GEMM 1 runs
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    acc[j] = *reinterpret_cast<const f32x4*>(bias_lds + L1.b_lds + (blk0 + j * NW) * 16 + 4 * g);
  }
  hook(0);
  gemm_run<NB, PINNED>(acc, r1, p1);
  hook(1);
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (r == 0) relu4_keep_nan(acc[j]); }
  if (has2) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc[j] += *reinterpret_cast<const f32x4*>(bias_lds + L2.b_lds + (blk0 + j * NW) * 16 + 4 * g);
    }
    gemm_run<NB, PINNED>(acc, r2, p2);
  }
  hook(2);
#pragma unroll
  for (int j = 0; j < NB; ++j)
    *reinterpret_cast<f32x4*>(Y + row * SY + (blk0 + j * NW) * 16 + 4 * g) = acc[j];
}

// Y = relu(W1.X1 + b1) [+ W2.X2 + b2]   for the 16-row tile; all NW waves of the workgroup call it.
// `pre` holds this wave's first four GEMM-1 fragments on entry (if w.use_pre) and, on exit, the first four
// fragments of the following stage's GEMM 1, requested BEFORE this stage's closing barrier.
// Ends with a workgroup barrier (Y visible, inputs free to overwrite).
// `to_reg` (split-K stages with a 16-wide output only): thread e < 256 keeps Y[e / 16][e % 16] in *to_reg instead of
// writing the tile to LDS, and the closing barrier is dropped -- the caller consumes the value in the same thread
// mapping (the fused SDE step: row = tid >> 4, component = tid & 15) and synchronises before LDS is reused.
template <int NW, typename Hook>
__device__ __forceinline__ void unet_stage(const float* __restrict__ Wp, const float* bias_lds, const StageDesc& sd,
                                           const WaveWorkS& w, float* lds, float* scratch, Pre& pre, Hook hook,
                                           float* to_reg = nullptr) {
  const int lane = threadIdx.x & 63;
  // Settle the fragments the previous stage requested BEFORE this stage issues any load of its own: the wait
  // the compiler puts in front of this statement then only covers those (long since landed) loads; placed
  // later it would also drain this stage's fresh ring loads -- a full L2 round trip per stage.
  hook(5);
  asm volatile("" : "+v"(pre.f[0]), "+v"(pre.f[1]), "+v"(pre.f[2]), "+v"(pre.f[3]));
  asm volatile("" : "+v"(pre.f[4]), "+v"(pre.f[5]), "+v"(pre.f[6]), "+v"(pre.f[7]));
  hook(6);
  const LayerDesc& L1 = sd.L1;
  const LayerDesc& L2 = sd.L2;
  const float* X1 = lds + sd.x1;
  const float* X2 = lds + sd.x2;
  float* Y = lds + sd.y;
  const int S1 = sd.s1, S2 = sd.s2, SY = sd.sy;
  const bool has2 = sd.has2 != 0;
  if (!w.split) {
    bool use_pre = w.use_pre != 0;  // the first group starts from the fragments the previous stage requested
    int blk0 = w.blk0;
    for (int cnt = w.cnt; cnt > 0; cnt -= 4, blk0 += 4 * NW) {
      if (cnt >= 4)      stage_direct<4, NW>(Wp, bias_lds, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane, pre, use_pre, hook);
      else if (cnt == 3) stage_direct<3, NW>(Wp, bias_lds, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane, pre, false, hook);
      else if (cnt == 2) stage_direct<2, NW>(Wp, bias_lds, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane, pre, use_pre, hook);
      else               stage_direct<1, NW>(Wp, bias_lds, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane, pre, use_pre, hook);
      use_pre = false;
    }
    pre = prefetch_fragments(Wp, sd.Ln, w, lane);
    hook(3);
    __syncthreads();
    hook(4);
  } else {
    // fewer neuron blocks than waves: split the reduction (K) dimension across waves,
    // partial sums through LDS, bias + ReLU applied after the combine.
    const int parts = w.parts, blk = w.blk0, part = w.part;
    const int outp = L1.out_pad;
    const int row = lane & 15, g = lane >> 4;
    float* P1 = scratch;
    float* P2 = scratch + parts * 16 * outp;
    if (w.active) {
      const GemmPlan<1> p1 = make_plan<1>(Wp, L1, blk, 0, X1, S1, lane, w.kc0a, w.kc1a);
      const GemmPlan<1> p2 = make_plan<1>(Wp, L2, blk, 0, X2, S2, lane, w.kc0b, w.kc1b);
      Ring<1> r1, r2;
      const bool w1 = p1.kc1 > p1.kc0, w2 = has2 && p2.kc1 > p2.kc0;
      if (w1) ring_fill<1, true>(r1, p1, pre);
      if (w2) ring_fill<1, false>(r2, p2, pre);
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      if (w1) gemm_run<1>(acc, r1, p1);
      *reinterpret_cast<f32x4*>(P1 + (part * 16 + row) * outp + blk * 16 + 4 * g) = acc[0];
      if (has2) {
        f32x4 acc2[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        if (w2) gemm_run<1>(acc2, r2, p2);
        *reinterpret_cast<f32x4*>(P2 + (part * 16 + row) * outp + blk * 16 + 4 * g) = acc2[0];
      }
    }
    pre = prefetch_fragments(Wp, sd.Ln, w, lane);
    __syncthreads();
    if (to_reg) {                                   // outp == 16: element e = threadIdx.x, no LDS round trip
      const int e = threadIdx.x;
      if (e < 256) {
        const int r = e >> 4, n = e & 15;
        float v = bias_lds[L1.b_lds + n];
        for (int p = 0; p < parts; ++p) v += P1[(p * 16 + r) * 16 + n];
        v = relu_keep_nan(v);
        if (has2) {
          float v2 = bias_lds[L2.b_lds + n];
          for (int p = 0; p < parts; ++p) v2 += P2[(p * 16 + r) * 16 + n];
          v += v2;
        }
        *to_reg = v;
      }
      return;
    }
    const float inv_outp = __builtin_amdgcn_rcpf((float)outp);
    for (int e = threadIdx.x; e < 16 * outp; e += NW * 64) {
      const int r = (int)(((float)e + 0.5f) * inv_outp), n = e - r * outp;  // e / outp without an integer divide
      float v = bias_lds[L1.b_lds + n];
      for (int p = 0; p < parts; ++p) v += P1[(p * 16 + r) * outp + n];
      v = relu_keep_nan(v);
      if (has2) {
        float v2 = bias_lds[L2.b_lds + n];
        for (int p = 0; p < parts; ++p) v2 += P2[(p * 16 + r) * outp + n];
        v += v2;
      }
      Y[r * SY + n] = v;
    }
    __syncthreads();
  }
}

// `Pre` is carried across stages and across time steps (the last stage prefetches down_0 again).
__device__ __forceinline__ Pre unet_carry_init(const float* __restrict__ Wp, const UnetProgram& prog) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return prefetch_fragments(Wp, prog.st[5].Ln, load_work(prog.ww[5][wave]), lane);
}

// The whole network on the tile: X0 (already filled, [t, x, 0-pad]) -> GV (nabla_V, first d columns valid).
// hook(i) is called after stage i = 1..6 (diagnostics).
template <int NW, typename Hook>
__device__ __forceinline__ void unet_tile_forward(const float* __restrict__ Wp, const UnetProgram& prog,
                                                  const TileLayout& t, float* lds, Pre& c, Hook hook) {
  static_assert(NW <= kMaxWaves, "work table too small");
  float* SC = lds + t.scratch;
  const float* BL = lds + t.bias;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 1
  for (int si = 0; si < 6; ++si) {
    const StageDesc sd = load_stage(prog.st[si]);
    const WaveWorkS ww = load_work(prog.ww[si][wave]);
    unet_stage<NW>(Wp, BL, sd, ww, lds, SC, c, [&](int sub) { hook(16 + si * 8 + sub); });
    hook(si + 1);
  }
}

// ---- fully specialised form -----------------------------------------------------------------------------
// For a network whose padded widths are template constants every descriptor above is a compile-time value:
// no descriptor loads, no runtime NB dispatch, offsets folded into immediates.  The per-stage bookkeeping of
// the table-driven loop (~100 scalar instructions per wave per stage) costs more than the MFMAs of the small
// layers, so the reference's default architecture gets this instantiation (socmx_rollout.hip picks it).
template <int IN0P, int H0P, int H1P, int H2P, int OUTP>
struct StaticNet {
  static constexpr int in0p = IN0P, outp = OUTP;
  __host__ __device__ static constexpr UnetDesc desc() { return make_unet_desc_padded(0, IN0P, H0P, H1P, H2P, OUTP); }
  __host__ __device__ static constexpr TileLayout layout(int nw) { return make_tile_layout(desc(), nw); }
  __host__ __device__ static constexpr TileLayout layout4(int nw) { return make_tile_layout(desc(), nw, 4); }   // 4-row tile
};

template <int NW, class NET, int SI, typename Hook>
__device__ __forceinline__ void unet_stage_static(const float* __restrict__ Wp, float* lds, Pre& c, int wave, Hook hook,
                                                  float* to_reg = nullptr) {
  constexpr UnetDesc u = NET::desc();
  constexpr TileLayout t = NET::layout(NW);
  constexpr StageDesc sd = unet_stage_desc(u, t, SI);
  constexpr int NBLK = sd.L1.out_pad >> 4;
  constexpr WaveWork wref = wave_work_of(sd, NW, 0);
  // the stage that follows (its GEMM 1 is this stage's Ln): is its work split the uniform direct one as well?
  constexpr StageDesc sdn = unet_stage_desc(u, t, (SI + 1) % 6);
  constexpr int NBLKn = sdn.L1.out_pad >> 4, KCn = sdn.L1.in_pad >> 4;
  constexpr WaveWork wrefn = wave_work_of(sdn, NW, 0);
  constexpr bool uniform_next = !wrefn.split && ((NBLKn >= NW && NBLKn % NW == 0 && NBLKn / NW <= 4 && NBLKn / NW != 3) ||
                                                 (NBLKn < NW));
  // one 16-wide output block whose K range divides evenly over the waves (the last stage of the default networks):
  // the wave's whole share of GEMM 1 (and GEMM 2's few chunks) arrives through the prefetch, the stage itself issues
  // no load and has no runtime loop bounds
  constexpr int KC1 = sd.L1.in_pad >> 4, KC2 = sd.L2.in_pad >> 4;
  // (one or two blocks: wave w takes block w % NBLK and part w / NBLK of the K range, NW / NBLK parts per block)
  constexpr int PARTS = (NBLK <= 2 && NW % NBLK == 0) ? NW / NBLK : 1;
  constexpr bool slim_split = wref.split && NBLK <= 2 && NW % NBLK == 0 && KC1 % PARTS == 0 && (KC1 / PARTS) + 1 <= 8 &&
                              (!sd.has2 || KC2 <= PARTS);
  constexpr int KC2n = sdn.L2.in_pad >> 4;
  constexpr int PARTSn = (NBLKn <= 2 && NW % NBLKn == 0) ? NW / NBLKn : 1;
  constexpr bool slim_split_next = wrefn.split && NBLKn <= 2 && NW % NBLKn == 0 && KCn % PARTSn == 0 &&
                                   (KCn / PARTSn) + 1 <= 8 && (!sdn.has2 || KC2n <= PARTSn);
  // every active wave owns the same number of neuron blocks: instantiate exactly that stage_direct<NB> (the generic
  // stage keeps all four NB variants alive behind a runtime switch on w.cnt -- three quarters of its code is dead)
  constexpr bool uniform = !wref.split && ((NBLK >= NW && NBLK % NW == 0 && NBLK / NW <= 4 && NBLK / NW != 3) ||
                                           (NBLK < NW));
  if constexpr (uniform) {
    constexpr int NBc = NBLK >= NW ? NBLK / NW : 1;
    constexpr int nact = NBLK >= NW ? NW : NBLK;
    const int lane = threadIdx.x & 63;
    auto sub = [&](int x) { hook(16 + SI * 8 + x); };
    sub(5);
    asm volatile("" : "+v"(c.f[0]), "+v"(c.f[1]), "+v"(c.f[2]), "+v"(c.f[3]));
    asm volatile("" : "+v"(c.f[4]), "+v"(c.f[5]), "+v"(c.f[6]), "+v"(c.f[7]));
    sub(6);
    if (wave < nact)
      stage_direct<NBc, NW, decltype(sub), true>(Wp, lds + t.bias, sd.L1, lds + sd.x1, sd.s1, sd.has2 != 0, sd.L2,
                                                 lds + sd.x2, sd.s2, lds + sd.y, sd.sy, wave, lane, c, true, sub);
    if constexpr (uniform_next) {
      // first ring of the next stage's GEMM 1 for this wave: fragment (block wave + (f % nb) NW, chunk f / nb) -- the
      // numbering of first_fragment_numbers() with every term but the wave id folded into an immediate
      constexpr int nbn = NBLKn >= NW ? NBLKn / NW : 1;
      const FragBase wl = frag_base(Wp, sdn.L1.w_off, lane);
      const int wb = min(wave, NBLKn - 1) * KCn;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int kc = (f / nbn) < KCn ? (f / nbn) : KCn - 1;
        const int boff = ((f % nbn) * NW) * KCn;
        c.f[f] = wl[(size_t)(wb + boff + kc) * 64];
      }
    } else if constexpr (slim_split_next) {
      // wave w of the next stage (block b = w % NBLKn, part p = w / NBLKn) multiplies chunks [p CPW, (p+1) CPW) of
      // GEMM 1 and chunk p of GEMM 2 (if it has one); fragment (block, chunk) sits at index block KC + chunk
      constexpr int CPWn = KCn / PARTSn;
      const int bn = wave % NBLKn, pn = wave / NBLKn;
      const FragBase w1 = frag_base(Wp, sdn.L1.w_off, lane);
#pragma unroll
      for (int f = 0; f < CPWn; ++f) c.f[f] = w1[(size_t)(bn * KCn + pn * CPWn + f) * 64];
      if (sdn.has2 && pn < KC2n)
        c.f[CPWn] = frag_base(Wp, sdn.L2.w_off, lane)[(size_t)(bn * KC2n + pn) * 64];
    } else {
      const WaveWork w0 = wave_work_of(sd, NW, wave);
      WaveWorkS w{};
#pragma unroll
      for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
      c = prefetch_fragments(Wp, sd.Ln, w, lane);
    }
    sub(3);
    __syncthreads();
    sub(4);
  } else if constexpr (slim_split) {
    constexpr int CPW = KC1 / PARTS;
    static_assert(NBLK * 256 <= NW * 64, "the combine needs one thread per output element");
    const int lane = threadIdx.x & 63, row = lane & 15, g = lane >> 4;
    const int blk = wave % NBLK, part = wave / NBLK;
    float* P1 = lds + t.scratch;                       // [NBLK][PARTS][16 rows][16]  (NBLK PARTS = NW)
    float* P2 = P1 + NW * 256;                         // [NBLK][KC2][16 rows][16]
    asm volatile("" : "+v"(c.f[0]), "+v"(c.f[1]), "+v"(c.f[2]), "+v"(c.f[3]));
    asm volatile("" : "+v"(c.f[4]), "+v"(c.f[5]), "+v"(c.f[6]), "+v"(c.f[7]));
    {
      const float* xrow = lds + sd.x1 + row * sd.s1 + 4 * g + part * (CPW * 16);
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int f = 0; f < CPW; ++f) {
        const f32x4 a1[1] = {c.f[f]};
        mfma_chunk<1>(acc, a1, *reinterpret_cast<const f32x4*>(xrow + f * 16));
      }
      *reinterpret_cast<f32x4*>(P1 + ((blk * PARTS + part) * 16 + row) * 16 + 4 * g) = acc[0];
    }
    if (sd.has2 && part < KC2) {
      const float* xrow = lds + sd.x2 + row * sd.s2 + 4 * g + part * 16;
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      const f32x4 a2[1] = {c.f[CPW]};
      mfma_chunk<1>(acc, a2, *reinterpret_cast<const f32x4*>(xrow));
      *reinterpret_cast<f32x4*>(P2 + ((blk * KC2 + part) * 16 + row) * 16 + 4 * g) = acc[0];
    }
    // prefetch for the stage that follows (stage 0 of the next time step)
    if constexpr (uniform_next) {
      constexpr int nbn = NBLKn >= NW ? NBLKn / NW : 1;
      const FragBase wl = frag_base(Wp, sdn.L1.w_off, lane);
      const int wb = min(wave, NBLKn - 1) * KCn;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const int kc = (f / nbn) < KCn ? (f / nbn) : KCn - 1;
        const int boff = ((f % nbn) * NW) * KCn;
        c.f[f] = wl[(size_t)(wb + boff + kc) * 64];
      }
    } else {
      const WaveWork w0 = wave_work_of(sd, NW, wave);
      WaveWorkS w{};
#pragma unroll
      for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
      c = prefetch_fragments(Wp, sd.Ln, w, lane);
    }
    __syncthreads();
    const float* bias_lds = lds + t.bias;
    const int e = threadIdx.x & 255, eb = threadIdx.x >> 8;   // element (r, n) = (e >> 4, e & 15) of block eb
    float v = 0.f;
    if (eb < NBLK) {                                   // bias, ReLU, residual
      const int n = eb * 16 + (e & 15);
      v = bias_lds[sd.L1.b_lds + n];
#pragma unroll
      for (int p = 0; p < PARTS; ++p) v += P1[(eb * PARTS + p) * 256 + e];
      v = relu_keep_nan(v);
      if (sd.has2) {
        float v2 = bias_lds[sd.L2.b_lds + n];
#pragma unroll
        for (int p = 0; p < KC2; ++p) v2 += P2[(eb * KC2 + p) * 256 + e];
        v += v2;
      }
    }
    if (to_reg) {                                      // (16-wide outputs only: thread tid < 256 holds element tid)
      *to_reg = v;                                     // the caller synchronises before LDS is reused (see unet_stage)
    } else {
      if (eb < NBLK) (lds + sd.y)[(e >> 4) * sd.sy + eb * 16 + (e & 15)] = v;
      __syncthreads();
    }
  } else {
    const WaveWork w0 = wave_work_of(sd, NW, wave);
    WaveWorkS w;
    w.split = w0.split; w.blk0 = w0.blk0; w.cnt = w0.cnt; w.active = w0.active;
    w.kc0a = w0.kc0a; w.kc1a = w0.kc1a; w.kc0b = w0.kc0b; w.kc1b = w0.kc1b;
    w.part = w0.part; w.parts = w0.parts; w.use_pre = w0.use_pre;
#pragma unroll
    for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
    unet_stage<NW>(Wp, lds + t.bias, sd, w, lds, lds + t.scratch, c, [&](int sub) { hook(16 + SI * 8 + sub); }, to_reg);
  }
  hook(SI + 1);
}

// gv_reg != nullptr (output width 16 only): the last stage hands nabla_V[tid >> 4][tid & 15] to thread tid < 256 in a
// register instead of the GV tile (see unet_stage)
template <int NW, class NET, typename Hook>
__device__ __forceinline__ void unet_tile_forward_static(const float* __restrict__ Wp, float* lds, Pre& c, Hook hook,
                                                         float* gv_reg = nullptr) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unet_stage_static<NW, NET, 0>(Wp, lds, c, wave, hook);
  unet_stage_static<NW, NET, 1>(Wp, lds, c, wave, hook);
  unet_stage_static<NW, NET, 2>(Wp, lds, c, wave, hook);
  unet_stage_static<NW, NET, 3>(Wp, lds, c, wave, hook);
  unet_stage_static<NW, NET, 4>(Wp, lds, c, wave, hook);
  unet_stage_static<NW, NET, 5>(Wp, lds, c, wave, hook, NET::outp == 16 ? gv_reg : nullptr);
}

template <int NW, class NET>
__device__ __forceinline__ Pre unet_carry_init_static(const float* __restrict__ Wp) {
  constexpr UnetDesc u = NET::desc();
  constexpr TileLayout t = NET::layout(NW);
  constexpr StageDesc sd = unet_stage_desc(u, t, 5);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const WaveWork w0 = wave_work_of(sd, NW, wave);
  WaveWorkS w{};
#pragma unroll
  for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
  return prefetch_fragments(Wp, sd.Ln, w, lane);
}

// ---- the 4-row tile (small batches) ---------------------------------------------------------------------------------
// A training batch of 128 rows is 8 tiles of 16 rows: 8 of 256 CUs work, each through a chain of 2,700 dependent-stage
// MFMAs per step.  With v_mfma_f32_4x4x1_16b_f32 (see mfma_chunk) the same weight image multiplies a 4-row tile: four
// times the workgroups, each with half the MFMA time per step.  Constexpr-specialised networks only, every stage either
// `uniform` (each active wave owns NBc whole neuron blocks) or `slim_split` (one or two blocks, K split over the waves) --
// which is what the default architecture's six stages are.
// can the 4-row stages run this architecture with NW waves?  (every stage uniform or slim_split, see unet_stage_static4)
__host__ __device__ constexpr bool r4_stage_ok(const StageDesc& sd, int NW) {
  const int NBLK = sd.L1.out_pad >> 4, KC1 = sd.L1.in_pad >> 4, KC2 = sd.L2.in_pad >> 4;
  const bool split = !(NBLK >= NW || NBLK >= kSimds);
  const int PARTS = (NBLK <= 2 && NW % NBLK == 0) ? NW / NBLK : 1;
  const bool slim = split && NBLK <= 2 && NW % NBLK == 0 && KC1 % PARTS == 0 && (KC1 / PARTS) + 1 <= 8 && (!sd.has2 || KC2 <= PARTS);
  const bool uniform = !split && ((NBLK >= NW && NBLK % NW == 0 && NBLK / NW <= 4 && NBLK / NW != 3) || (NBLK < NW));
  return uniform || slim;
}
template <int NW, class NET>
__host__ __device__ constexpr bool r4_supported() {
  for (int si = 0; si < 6; ++si)
    if (!r4_stage_ok(unet_stage_desc(NET::desc(), NET::layout4(NW), si), NW)) return false;
  return true;
}
// the stage whose GEMM-1 layer the 4-row FAST kernel keeps resident in LDS (-1: none -- it must be a uniform two-GEMM stage
// and leave room for the tiles: 132 KiB at most)
template <int NW, class NET>
__host__ __device__ constexpr int r4_resident_stage() {
  const StageDesc sd = unet_stage_desc(NET::desc(), NET::layout4(NW), 4);
  const int NBLK = sd.L1.out_pad >> 4;
  const bool uniform = (NBLK >= NW && NBLK % NW == 0 && NBLK / NW <= 4 && NBLK / NW != 3) || (NBLK < NW && NBLK >= kSimds);
  return (uniform && sd.L1.in_pad * sd.L1.out_pad <= 33 * 1024) ? 4 : -1;     // (with or without a second GEMM: the folded stage 4 has none)
}
template <int NW, class NET>
__host__ __device__ constexpr int r4_resident_floats() {
  const StageDesc sd = unet_stage_desc(NET::desc(), NET::layout4(NW), 4);
  return r4_resident_stage<NW, NET>() < 0 ? 0 : sd.L1.in_pad * sd.L1.out_pad;
}

// the 4-row stages' ring: PF fragments in flight per wave (8; 4 in a 16-wave workgroup, whose budget is 128 VGPRs)
template <int NB, int PF>
struct Ring4 {
  static constexpr int PD = PF / NB > 0 ? PF / NB : 1;
  static constexpr int CP = PD;                                  // chunks covered by the first PF `pre` fragments
  f32x4 slot[PD][NB];
};
#ifndef SOCMX_R4_FRAGS
#define SOCMX_R4_FRAGS 4
#endif
template <int NW> struct R4Frags { static constexpr int value = NW > 8 ? 4 : SOCMX_R4_FRAGS; };

// W1LDS: GEMM 1's weights are RESIDENT in LDS (w1lds: the layer's fragment image, copied once per launch) -- its fragments
// are read where they are used and `pre` carries the first ring of GEMM 2 instead (see unet_stage_static4).
template <int NB, int NW, bool W1LDS = false>
__device__ __forceinline__ void stage_direct4(const float* __restrict__ Wp, const float* bias_lds, const LayerDesc& L1,
                                              const float* X1, int S1, bool has2, const LayerDesc& L2, const float* X2,
                                              int S2, float* Y, int SY, int blk0, int lane, const Pre& pre,
                                              const float* w1lds = nullptr) {
  const int j = lane & 3, ng = (lane >> 2) & 3;
  const GemmPlan<NB> p1 = make_plan<NB, true>(Wp, L1, blk0, NW, X1, S1, lane, 0, L1.in_pad >> 4);
  const GemmPlan<NB> p2 = make_plan<NB, true>(Wp, L2, blk0, NW, X2, S2, lane, 0, L2.in_pad >> 4);
  typedef Ring4<NB, R4Frags<NW>::value> RingT;
  RingT r1, r2;
  if constexpr (W1LDS) {
    if (has2) ring_fill<NB, true, RingT>(r2, p2, pre);
  } else {
    ring_fill<NB, true, RingT>(r1, p1, pre);
    if (has2) ring_fill<NB, false, RingT>(r2, p2, pre);   // the residual GEMM's first chunks fly while GEMM 1 runs
  }
  // accumulators start from zero; a lane's ONE bias value (of the neuron whose total kg_reduce leaves it with) is read here
  // and added behind the reduction -- nothing between the barrier and the first MFMA waits for it
  f32x4 acc[NB], acc2[NB];
  float bias1[NB], bias2[NB];
  const f32x4 zero{0.f, 0.f, 0.f, 0.f};
  const int comp = kg_comp(lane);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    acc[b] = zero;
    acc2[b] = zero;
    bias1[b] = bias_lds[L1.b_lds + (blk0 + b * NW) * 16 + 4 * ng + comp];
    bias2[b] = has2 ? bias_lds[L2.b_lds + (blk0 + b * NW) * 16 + 4 * ng + comp] : 0.f;
  }
  if constexpr (W1LDS) {
    const int KC = L1.in_pad >> 4;
    const f32x4* wl = reinterpret_cast<const f32x4*>(w1lds) + lane;
#pragma unroll 4
    for (int kc = 0; kc < KC; ++kc) {
      f32x4 a[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) a[b] = wl[((blk0 + b * NW) * KC + kc) * 64];
      mfma_chunk<NB, true>(acc, a, *reinterpret_cast<const f32x4*>(p1.xrow + kc * 16));
    }
  } else {
    gemm_run<NB, true, true, RingT>(acc, r1, p1);
  }
  if (has2) gemm_run<NB, true, true, RingT>(acc2, r2, p2);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float v = relu_keep_nan(kg_reduce(acc[b]) + bias1[b]);
    if (has2) v += kg_reduce(acc2[b]) + bias2[b];
    Y[j * SY + (blk0 + b * NW) * 16 + 4 * ng + comp] = v;                  // one float per lane, 64 banks
  }
}

// RES_SI: the stage (or -1) whose GEMM-1 layer is resident in LDS at `res_w` (a uniform two-GEMM stage)
template <int NW, class NET, int SI, int RES_SI = -1>
__device__ __forceinline__ void unet_stage_static4(const float* __restrict__ Wp, float* lds, Pre& c, int wave,
                                                   float* to_reg = nullptr, const float* res_w = nullptr) {
  constexpr UnetDesc u = NET::desc();
  constexpr TileLayout t = NET::layout4(NW);
  constexpr StageDesc sd = unet_stage_desc(u, t, SI);
  constexpr int NBLK = sd.L1.out_pad >> 4;
  constexpr WaveWork wref = wave_work_of(sd, NW, 0);
  constexpr StageDesc sdn = unet_stage_desc(u, t, (SI + 1) % 6);
  constexpr int NBLKn = sdn.L1.out_pad >> 4, KCn = sdn.L1.in_pad >> 4;
  constexpr WaveWork wrefn = wave_work_of(sdn, NW, 0);
  constexpr bool uniform_next = !wrefn.split && ((NBLKn >= NW && NBLKn % NW == 0 && NBLKn / NW <= 4 && NBLKn / NW != 3) ||
                                                 (NBLKn < NW));
  constexpr int KC1 = sd.L1.in_pad >> 4, KC2 = sd.L2.in_pad >> 4;
  constexpr int PARTS = (NBLK <= 2 && NW % NBLK == 0) ? NW / NBLK : 1;
  constexpr bool slim_split = wref.split && NBLK <= 2 && NW % NBLK == 0 && KC1 % PARTS == 0 && (KC1 / PARTS) + 1 <= 8 &&
                              (!sd.has2 || KC2 <= PARTS);
  constexpr int KC2n = sdn.L2.in_pad >> 4;
  constexpr int PARTSn = (NBLKn <= 2 && NW % NBLKn == 0) ? NW / NBLKn : 1;
  constexpr bool slim_split_next = wrefn.split && NBLKn <= 2 && NW % NBLKn == 0 && KCn % PARTSn == 0 &&
                                   (KCn / PARTSn) + 1 <= 8 && (!sdn.has2 || KC2n <= PARTSn);
  constexpr bool uniform = !wref.split && ((NBLK >= NW && NBLK % NW == 0 && NBLK / NW <= 4 && NBLK / NW != 3) ||
                                           (NBLK < NW));
  static_assert(uniform || slim_split, "4-row tile: this architecture needs the 16-row kernels");
  static_assert(uniform_next || slim_split_next, "4-row tile: this architecture needs the 16-row kernels");
  const int lane = threadIdx.x & 63;
  // (settle the fragments the previous stage requested before this stage issues loads of its own, see unet_stage)
  asm volatile("" : "+v"(c.f[0]), "+v"(c.f[1]), "+v"(c.f[2]), "+v"(c.f[3]));
  asm volatile("" : "+v"(c.f[4]), "+v"(c.f[5]), "+v"(c.f[6]), "+v"(c.f[7]));
  // first ring of the next stage's GEMM 1 (requested before this stage's closing barrier)
  constexpr bool res_here = SI == RES_SI, res_next = (SI + 1) % 6 == RES_SI;
  static_assert(!res_here || uniform, "the resident layer is GEMM 1 of a uniform stage");
  auto prefetch_next = [&]() {
    if constexpr (res_next && !sdn.has2) {
      // (next stage's only GEMM is resident in LDS: nothing to request)
    } else if constexpr (uniform_next) {
      // (next stage's GEMM 1 resident in LDS: the first ring of its GEMM 2 instead)
      constexpr int nbn = NBLKn >= NW ? NBLKn / NW : 1;
      constexpr int KCx = res_next ? KC2n : KCn;
      const FragBase wl = frag_base(Wp, res_next ? sdn.L2.w_off : sdn.L1.w_off, lane);
      const int wb = min(wave, NBLKn - 1) * KCx;
#pragma unroll
      for (int f = 0; f < R4Frags<NW>::value; ++f) {
        const int kc = (f / nbn) < KCx ? (f / nbn) : KCx - 1;
        const int boff = ((f % nbn) * NW) * KCx;
        c.f[f] = wl[(size_t)(wb + boff + kc) * 64];
      }
    } else {
      constexpr int CPWn = KCn / PARTSn;
      const int bn = wave % NBLKn, pn = wave / NBLKn;
      const FragBase w1 = frag_base(Wp, sdn.L1.w_off, lane);
#pragma unroll
      for (int f = 0; f < CPWn; ++f) c.f[f] = w1[(size_t)(bn * KCn + pn * CPWn + f) * 64];
      if (sdn.has2 && pn < KC2n)
        c.f[CPWn] = frag_base(Wp, sdn.L2.w_off, lane)[(size_t)(bn * KC2n + pn) * 64];
    }
  };
  if constexpr (uniform) {
    constexpr int NBc = NBLK >= NW ? NBLK / NW : 1;
    constexpr int nact = NBLK >= NW ? NW : NBLK;
    if (wave < nact)
      stage_direct4<NBc, NW, res_here>(Wp, lds + t.bias, sd.L1, lds + sd.x1, sd.s1, sd.has2 != 0, sd.L2, lds + sd.x2, sd.s2,
                                       lds + sd.y, sd.sy, wave, lane, c, res_w);
    prefetch_next();
    __syncthreads();
  } else {
    constexpr int CPW = KC1 / PARTS;
    static_assert(NBLK * 64 <= NW * 64, "the combine needs one thread per output element");
    const int j = lane & 3, ng = (lane >> 2) & 3, kg = lane >> 4;
    const int blk = wave % NBLK, part = wave / NBLK;
    float* P1 = lds + t.scratch;                       // [NBLK][PARTS][4 rows][16]  (NBLK PARTS = NW)
    float* P2 = P1 + NW * 64;                          // [NBLK][KC2][4 rows][16]
    {
      const float* xrow = lds + sd.x1 + j * sd.s1 + 4 * kg + part * (CPW * 16);
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int f = 0; f < CPW; ++f) {
        const f32x4 a1[1] = {c.f[f]};
        mfma_chunk<1, true>(acc, a1, *reinterpret_cast<const f32x4*>(xrow + f * 16));
      }
      P1[((blk * PARTS + part) * 4 + j) * 16 + 4 * ng + kg_comp(lane)] = kg_reduce(acc[0]);
    }
    if (sd.has2 && part < KC2) {
      const float* xrow = lds + sd.x2 + j * sd.s2 + 4 * kg + part * 16;
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      const f32x4 a2[1] = {c.f[CPW]};
      mfma_chunk<1, true>(acc, a2, *reinterpret_cast<const f32x4*>(xrow));
      P2[((blk * KC2 + part) * 4 + j) * 16 + 4 * ng + kg_comp(lane)] = kg_reduce(acc[0]);
    }
    prefetch_next();
    __syncthreads();
    const float* bias_lds = lds + t.bias;
    const int e = threadIdx.x & 63, eb = threadIdx.x >> 6;    // element (r, n) = (e >> 4, e & 15) of block eb
    float v = 0.f;
    if (eb < NBLK) {                                   // bias, ReLU, residual
      const int n = eb * 16 + (e & 15);
      // every operand is requested before the first add (one LDS round trip on the chain that ends the step, not one per
      // pair of partials), then a fixed-order tree
      float q1[PARTS], q2[KC2 > 0 ? KC2 : 1];
      const float b1 = bias_lds[sd.L1.b_lds + n], b2 = sd.has2 ? bias_lds[sd.L2.b_lds + n] : 0.f;
#pragma unroll
      for (int p = 0; p < PARTS; ++p) q1[p] = P1[(eb * PARTS + p) * 64 + e];
#pragma unroll
      for (int p = 0; p < KC2; ++p) q2[p] = sd.has2 ? P2[(eb * KC2 + p) * 64 + e] : 0.f;
#pragma unroll
      for (int w = 1; w < PARTS; w *= 2)
#pragma unroll
        for (int p = 0; p + w < PARTS; p += 2 * w) q1[p] += q1[p + w];
      v = relu_keep_nan(q1[0] + b1);
      if (sd.has2) {
        float v2 = b2;
#pragma unroll
        for (int p = 0; p < KC2; ++p) v2 += q2[p];
        v += v2;
      }
    }
    if (to_reg) {                                      // (16-wide outputs only: thread tid < 64 holds element tid)
      *to_reg = v;                                     // the caller synchronises before LDS is reused
    } else {
      if (eb < NBLK) (lds + sd.y)[(e >> 4) * sd.sy + eb * 16 + (e & 15)] = v;
      __syncthreads();
    }
  }
}

// X0 (4 rows of [t, x, 0-pad]) -> nabla_V; gv_reg (output width 16): thread tid < 64 gets nabla_V[tid >> 4][tid & 15]
template <int NW, class NET, int RES_SI = -1>
__device__ __forceinline__ void unet_tile_forward_static4(const float* __restrict__ Wp, float* lds, Pre& c,
                                                          float* gv_reg = nullptr, const float* res_w = nullptr) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unet_stage_static4<NW, NET, 0, RES_SI>(Wp, lds, c, wave, nullptr, res_w);
  unet_stage_static4<NW, NET, 1, RES_SI>(Wp, lds, c, wave, nullptr, res_w);
  unet_stage_static4<NW, NET, 2, RES_SI>(Wp, lds, c, wave, nullptr, res_w);
  unet_stage_static4<NW, NET, 3, RES_SI>(Wp, lds, c, wave, nullptr, res_w);
  unet_stage_static4<NW, NET, 4, RES_SI>(Wp, lds, c, wave, nullptr, res_w);
  unet_stage_static4<NW, NET, 5, RES_SI>(Wp, lds, c, wave, NET::outp == 16 ? gv_reg : nullptr, res_w);
}

#endif  // __HIPCC__
}  // namespace socmx
