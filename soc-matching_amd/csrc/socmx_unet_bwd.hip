// socmx_unet_bwd.hip -- parameter gradients of the control network over the trajectory rows, gfx950 (MI355X).
//
// Replaces autograd through reference SOC_matching/models.py:233-242 (FullyConnectedUNet.forward) on the (K+1)*B rows
// of method.py:272-278: given G = d objective / d nabla_V (N, d) it returns d objective / d (every weight and bias).
// The inputs of the network are detached states (utils.py:103-115), so no input gradient exists.
//
// Three kernels, no library GEMM:
//   A  unet_bwd_tile_kernel    one workgroup per 16-row tile: RE-COMPUTES the forward in LDS (same fp32 MFMA tile code
//                              as the rollout: socmx_unet.h), then walks the backward chain through the transposed
//                              weights; every activation tile and every pre-activation gradient tile leaves ONCE, as a
//                              [tile][unit][16 rows] slab (the MFMA operand order of kernel B).
//   B  unet_wgrad_kernel       dW_l = sum_rows gz_l (x) act_l for the nine layers in one launch: a wave owns a <= 4x4 group
//                              of 16x16 blocks of one layer and a slab of row tiles (split-K), operands straight from
//                              global memory as 16-byte loads, bias gradients as row sums of the same A fragments.
//   C  unet_wgrad_finish_kernel  adds the slabs in a fixed order (deterministic) and scatters into torch layout.
// Bound: fp32 MFMA (A: forward + activation-gradient GEMMs = 2 x 2 x MACs per row; B: 2 x MACs per row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/socmx.h"
#include "socmx_unet.h"
#include "socmx_launch.h"

namespace socmx {

// ---- transposed layers, in the order the backward chain uses them ------------------------------------------------
enum { KT_U0 = 0, KT_U1, KT_U2, KT_D2, KT_R2, KT_D1, KT_R1, KT_N };
// forward layer (SOCMX_L_* index) whose transpose each one is
__host__ __device__ constexpr int kt_source(int t) {
  constexpr int src[KT_N] = {8, 7, 6, 2, 5, 1, 4};
  return src[t];
}

struct BwdDesc {
  LayerDesc L[KT_N];
  int total_floats;
};

__host__ __device__ constexpr BwdDesc make_bwd_desc(const UnetDesc& u) {
  BwdDesc b{};
  int off = 0;
  for (int t = 0; t < KT_N; ++t) {
    const LayerDesc& f = u.L[kt_source(t)];
    b.L[t].in_pad = f.out_pad;
    b.L[t].out_pad = f.in_pad;
    b.L[t].w_off = off;
    b.L[t].b_off = 0;
    b.L[t].b_lds = 0;
    off += f.in_pad * f.out_pad;
  }
  b.total_floats = off;
  return b;
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
// (gradient tensor, activation tensor) of forward layer l: dW_l = grad^T . act
__host__ __device__ constexpr int layer_grad_tensor(int l) {
  constexpr int g[9] = {T_ZD0, T_ZD1, T_ZD2, T_G0, T_GO1, T_GO2, T_ZU2, T_ZU1, T_ZU0};
  return g[l];
}
__host__ __device__ constexpr int layer_act_tensor(int l) {
  constexpr int a[9] = {T_X, T_R1, T_R2, T_X, T_R1, T_R2, T_R3, T_O2, T_O1};
  return a[l];
}

// ---- LDS of kernel A: the forward tile layout of socmx_unet.h, the backward tiles behind it, nibble masks last ------
struct BwdLayout {
  TileLayout t;                                   // x0, r1, r2, r3, o2, o1, gv (= the G tile here), scratch, bias
  int zu0, go1, zu1, go2, zu2, zd2, zd1;          // float offsets
  int mu2, mu1, mu0;                              // float offsets of the byte arrays [16][width/4]
  int floats;
};

__host__ __device__ constexpr BwdLayout make_bwd_layout(const UnetDesc& u, int nwaves) {
  BwdLayout b{};
  b.t = make_tile_layout(u, nwaves);
  int off = b.t.floats;
  b.zu0 = off; off += 16 * b.t.sg;
  b.go1 = off; off += 16 * b.t.s1;
  b.zu1 = off; off += 16 * b.t.s1;
  b.go2 = off; off += 16 * b.t.s2;
  b.zu2 = off; off += 16 * b.t.s2;
  b.zd2 = off; off += 16 * b.t.s3;
  b.zd1 = off; off += 16 * b.t.s2;
  b.mu2 = off; off += u.hp[1];                    // 16 rows x hp/4 bytes = hp floats
  b.mu1 = off; off += u.hp[0];
  b.mu0 = off; off += u.outp;
  b.floats = off;
  return b;
}

// ---- the eleven stages of kernel A ------------------------------------------------------------------------------------
//  forward   0: R1 = relu(d0 X + b)      1: R2 = relu(d1 R1 + b)     2: R3 = relu(d2 R2 + b)
//            3: O2 = relu(u2 R3 + b) + r2 R2 + b   [mask MU2]        4: O1 = relu(u1 O2 + b) + r1 R1 + b   [mask MU1]
//            5: MU0 = (u0 O1 + b > 0);  ZU0 = G (.) MU0              (the output itself is not needed: res_0 is skipped)
//  backward  6: GO1 = u0^T ZU0;  ZU1 = GO1 (.) MU1                   7: GO2 = u1^T ZU1;  ZU2 = GO2 (.) MU2
//            8: ZD2 = (u2^T ZU2) (.) [R3 > 0]                         9: ZD1 = (d2^T ZD2 + r2^T GO2) (.) [R2 > 0]
//           10: ZD0 = (d1^T ZD1 + r1^T GO1) (.) [R1 > 0]
constexpr int kBwdStages = 11;
enum { EPI_RELU = 0, EPI_RES, EPI_MASK0, EPI_DUAL, EPI_ACTMASK };

struct K2Stage {
  StageDesc sd;          // L1, L2, Ln (GEMM 1 of the stage that follows), x1/s1, x2/s2, y/sy (primary output tile), has2
  int img1, img2, imgn;  // weight image of L1 / L2 / Ln: 0 = forward image, 1 = transposed image
  int epi;
  int y2, sy2;           // second tile (EPI_DUAL: the masked copy; EPI_MASK0: ZU0) -- float offset / stride
  int mask;              // float offset of the nibble-mask byte array written (EPI_RES, EPI_MASK0) or read (EPI_DUAL)
  int aux, saux;         // EPI_ACTMASK: activation tile whose sign masks the result; EPI_MASK0: the G tile
  int ex1, ex2;          // exported tensor ids (T_*) of the primary / second result, -1 = none
  int w1, w2;            // their widths
  int p1, p2;            // ... and the summed widths of the tensors before them (workspace offsets / (16 ntiles))
};

__host__ __device__ constexpr K2Stage make_k2_stage(const UnetDesc& u, const BwdDesc& bd, const BwdLayout& b, int si) {
  const TileLayout& t = b.t;
  K2Stage s{};
  s.y2 = -1; s.mask = -1; s.aux = -1; s.ex1 = -1; s.ex2 = -1;
  auto fwd = [&](int l1, int x1, int s1, int has2, int l2, int x2, int s2, int y, int sy) {
    s.sd.L1 = u.L[l1]; s.sd.L2 = u.L[l2]; s.sd.x1 = x1; s.sd.s1 = s1; s.sd.x2 = x2; s.sd.s2 = s2; s.sd.y = y; s.sd.sy = sy;
    s.sd.has2 = has2; s.img1 = 0; s.img2 = 0;
  };
  auto bwd = [&](int l1, int x1, int s1, int has2, int l2, int x2, int s2, int y, int sy) {
    s.sd.L1 = bd.L[l1]; s.sd.L2 = bd.L[l2]; s.sd.x1 = x1; s.sd.s1 = s1; s.sd.x2 = x2; s.sd.s2 = s2; s.sd.y = y; s.sd.sy = sy;
    s.sd.has2 = has2; s.img1 = 1; s.img2 = 1;
  };
  switch (si) {
    case 0: fwd(0, t.x0, t.s0, 0, 0, t.x0, t.s0, t.r1, t.s1); s.epi = EPI_RELU; s.ex1 = T_R1; break;
    case 1: fwd(1, t.r1, t.s1, 0, 1, t.r1, t.s1, t.r2, t.s2); s.epi = EPI_RELU; s.ex1 = T_R2; break;
    case 2: fwd(2, t.r2, t.s2, 0, 2, t.r2, t.s2, t.r3, t.s3); s.epi = EPI_RELU; s.ex1 = T_R3; break;
    case 3: fwd(6, t.r3, t.s3, 1, 5, t.r2, t.s2, t.o2, t.s2); s.epi = EPI_RES; s.mask = b.mu2; s.ex1 = T_O2; break;
    case 4: fwd(7, t.o2, t.s2, 1, 4, t.r1, t.s1, t.o1, t.s1); s.epi = EPI_RES; s.mask = b.mu1; s.ex1 = T_O1; break;
    case 5: fwd(8, t.o1, t.s1, 0, 8, t.o1, t.s1, -1, 0); s.epi = EPI_MASK0; s.mask = b.mu0; s.aux = t.gv; s.saux = t.sg;
            s.y2 = b.zu0; s.sy2 = t.sg; s.ex2 = T_ZU0; break;
    case 6: bwd(KT_U0, b.zu0, t.sg, 0, KT_U0, b.zu0, t.sg, b.go1, t.s1); s.epi = EPI_DUAL; s.mask = b.mu1;
            s.y2 = b.zu1; s.sy2 = t.s1; s.ex1 = T_GO1; s.ex2 = T_ZU1; break;
    case 7: bwd(KT_U1, b.zu1, t.s1, 0, KT_U1, b.zu1, t.s1, b.go2, t.s2); s.epi = EPI_DUAL; s.mask = b.mu2;
            s.y2 = b.zu2; s.sy2 = t.s2; s.ex1 = T_GO2; s.ex2 = T_ZU2; break;
    case 8: bwd(KT_U2, b.zu2, t.s2, 0, KT_U2, b.zu2, t.s2, b.zd2, t.s3); s.epi = EPI_ACTMASK; s.aux = t.r3; s.saux = t.s3;
            s.ex1 = T_ZD2; break;
    case 9: bwd(KT_D2, b.zd2, t.s3, 1, KT_R2, b.go2, t.s2, b.zd1, t.s2); s.epi = EPI_ACTMASK; s.aux = t.r2; s.saux = t.s2;
            s.ex1 = T_ZD1; break;
    default: bwd(KT_D1, b.zd1, t.s2, 1, KT_R1, b.go1, t.s1, -1, 0); s.epi = EPI_ACTMASK; s.aux = t.r1; s.saux = t.s1;
            s.ex1 = T_ZD0; break;
  }
  s.w1 = s.ex1 >= 0 ? tensor_width(u, s.ex1) : 0;
  s.w2 = s.ex2 >= 0 ? tensor_width(u, s.ex2) : 0;
  s.p1 = s.ex1 >= 0 ? tensor_prefix(u, s.ex1) : 0;
  s.p2 = s.ex2 >= 0 ? tensor_prefix(u, s.ex2) : 0;
  return s;
}

// GEMM 1 of stage si (what the stage before it prefetches)
__host__ __device__ constexpr void k2_first_layer(const UnetDesc& u, const BwdDesc& bd, int si, LayerDesc& L, int& img) {
  constexpr int f1[6] = {0, 1, 2, 6, 7, 8};
  constexpr int b1[5] = {KT_U0, KT_U1, KT_U2, KT_D2, KT_D1};
  if (si < 6) { L = u.L[f1[si]]; img = 0; }
  else { L = bd.L[b1[si - 6]]; img = 1; }
}

struct K2Program {      // (the per-wave work split is evaluated on the device: the table would not fit the kernel arguments)
  K2Stage st[kBwdStages];
};

__host__ __device__ constexpr K2Stage k2_stage_desc(const UnetDesc& u, const BwdDesc& bd, const BwdLayout& b, int si) {
  K2Stage s = make_k2_stage(u, bd, b, si);
  k2_first_layer(u, bd, (si + 1) % kBwdStages, s.sd.Ln, s.imgn);   // (the last stage prefetches stage 0 of the next tile: unused)
  return s;
}

#if defined(__HIPCC__)

struct TileArgs {
  UnetDesc u;
  BwdDesc bd;
  BwdLayout lay;
  K2Program prog;
  const float* packed;    // forward image (socmx_unet_pack_f32)
  const float* packedT;   // transposed image (socmx_unet_pack_bwd_f32)
  const float* x;         // (N, d) rows
  const float* ts;        // time of row r = ts[r / rows_per_t]
  const float* gout;      // (N, d)  d objective / d nabla_V
  int dbg;                // developer switch (SOCMX_K2_DBG): bit 0 = no exports
  float* ws;              // workspace: T_N tensors, tensor t at ws + 16 * ntiles * prefix(t), each [tile][width][16]
  int64_t N;
  int rows_per_t;
  int ntiles;
};

// one 16-byte LDS read
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// result quad (row r, units n0 .. n0+3) -> [unit][16 rows] slab of its tile: four 4-byte stores, each a 64-byte run over the
// 16 lanes of a row group
__device__ __forceinline__ void export4(float* slab, int r, int n0, const f32x4 v) {
  if (!slab) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) slab[(size_t)(n0 + i) * 16 + r] = v[i];
}

struct EpiCtx {
  float* lds;
  float* ws;               // workspace base (nullptr: developer switch SOCMX_K2_DBG=1, no exports -- timing only)
  const float* bias_lds;
  int64_t tile_rows;       // 16 * ntiles
  int tile;
};

template <int EPI>
struct Epi {
  const K2Stage& s;
  const EpiCtx& c;
  __device__ __forceinline__ float* slab(int prefix, int width) const {
    return c.ws ? c.ws + (size_t)c.tile_rows * prefix + (size_t)c.tile * width * 16 : nullptr;
  }
  __device__ __forceinline__ f32x4 init(int n0) const {
    if constexpr (EPI == EPI_RELU || EPI == EPI_RES || EPI == EPI_MASK0) return lds4(c.bias_lds + s.sd.L1.b_lds + n0);
    return f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // between GEMM 1 and GEMM 2
  __device__ __forceinline__ void mid(f32x4& v, int r, int n0) const {
    if constexpr (EPI == EPI_RES) {
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { m |= (v[i] > 0.f ? 1u : 0u) << i; v[i] = relu_keep_nan(v[i]); }
      reinterpret_cast<unsigned char*>(c.lds + s.mask)[r * (s.sd.L1.out_pad >> 2) + (n0 >> 2)] = (unsigned char)m;
      v += lds4(c.bias_lds + s.sd.L2.b_lds + n0);
    }
  }
  __device__ __forceinline__ void fin(f32x4 v, int r, int n0, const UnetDesc& u) const {
    if constexpr (EPI == EPI_RELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(v[i]);
      *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;
      export4(slab(s.p1, s.w1), r, n0, v);
    } else if constexpr (EPI == EPI_RES) {
      *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;
      export4(slab(s.p1, s.w1), r, n0, v);
    } else if constexpr (EPI == EPI_MASK0) {
      const f32x4 g = lds4(c.lds + s.aux + r * s.saux + n0);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = v[i] > 0.f ? g[i] : 0.f;
      *reinterpret_cast<f32x4*>(c.lds + s.y2 + r * s.sy2 + n0) = z;
      export4(slab(s.p2, s.w2), r, n0, z);
    } else if constexpr (EPI == EPI_DUAL) {
      const unsigned m = reinterpret_cast<const unsigned char*>(c.lds + s.mask)[r * (s.sd.L1.out_pad >> 2) + (n0 >> 2)];
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = ((m >> i) & 1u) ? v[i] : 0.f;
      *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;
      *reinterpret_cast<f32x4*>(c.lds + s.y2 + r * s.sy2 + n0) = z;
      export4(slab(s.p1, s.w1), r, n0, v);
      export4(slab(s.p2, s.w2), r, n0, z);
    } else {   // EPI_ACTMASK
      const f32x4 a = lds4(c.lds + s.aux + r * s.saux + n0);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = a[i] > 0.f ? v[i] : 0.f;
      if (s.sd.y >= 0) *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = z;
      export4(slab(s.p1, s.w1), r, n0, z);
    }
  }
};

template <int NB, int NW, bool PINNED, class EPI>
__device__ __forceinline__ void k2_direct(const float* __restrict__ W1, const float* __restrict__ W2, const StageDesc& sd,
                                          float* lds, int blk0, int lane, const Pre& pre, bool use_pre, const EPI& epi,
                                          const UnetDesc& u) {
  const int row = lane & 15, g = lane >> 4;
  const bool has2 = sd.has2 != 0;
  const GemmPlan<NB> p1 = make_plan<NB>(W1, sd.L1, blk0, NW, lds + sd.x1, sd.s1, lane, 0, sd.L1.in_pad >> 4);
  const GemmPlan<NB> p2 = make_plan<NB>(W2, sd.L2, blk0, NW, lds + sd.x2, sd.s2, lane, 0, sd.L2.in_pad >> 4);
  Ring<NB> r1, r2;
  if (use_pre) ring_fill<NB, true>(r1, p1, pre); else ring_fill<NB, false>(r1, p1, pre);
  if (has2) ring_fill<NB, false>(r2, p2, pre);
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = epi.init((blk0 + j * NW) * 16 + 4 * g);
  gemm_run<NB, PINNED>(acc, r1, p1);
#pragma unroll
  for (int j = 0; j < NB; ++j) epi.mid(acc[j], row, (blk0 + j * NW) * 16 + 4 * g);
  if (has2) gemm_run<NB, PINNED>(acc, r2, p2);
#pragma unroll
  for (int j = 0; j < NB; ++j) epi.fin(acc[j], row, (blk0 + j * NW) * 16 + 4 * g, u);
}

// One stage on the 16-row tile; same work split as socmx_unet.h's unet_stage (direct: neuron blocks dealt to the waves;
// fewer than four blocks: the reduction dimension is split over the waves and combined through LDS), generalised
// epilogue.  Ends with a workgroup barrier.
template <int NW, bool PINNED, class EPI>
__device__ __forceinline__ void k2_stage(const float* __restrict__ W1, const float* __restrict__ W2,
                                         const float* __restrict__ Wn, const StageDesc& sd, const WaveWorkS& w, float* lds,
                                         float* scratch, Pre& pre, const EPI& epi, const UnetDesc& u) {
  const int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(pre.f[0]), "+v"(pre.f[1]), "+v"(pre.f[2]), "+v"(pre.f[3]));
  asm volatile("" : "+v"(pre.f[4]), "+v"(pre.f[5]), "+v"(pre.f[6]), "+v"(pre.f[7]));
  const bool has2 = sd.has2 != 0;
  if (!w.split) {
    bool use_pre = w.use_pre != 0;
    int blk0 = w.blk0;
    for (int cnt = w.cnt; cnt > 0; cnt -= 4, blk0 += 4 * NW) {
      if (cnt >= 4)      k2_direct<4, NW, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      else if (cnt == 3) k2_direct<3, NW, PINNED>(W1, W2, sd, lds, blk0, lane, pre, false, epi, u);
      else if (cnt == 2) k2_direct<2, NW, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      else               k2_direct<1, NW, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      use_pre = false;
    }
    pre = prefetch_fragments(Wn, sd.Ln, w, lane);
    __syncthreads();
  } else {
    const int parts = w.parts, blk = w.blk0, part = w.part;
    const int outp = sd.L1.out_pad;
    const int row = lane & 15, g = lane >> 4;
    float* P1 = scratch;
    float* P2 = scratch + parts * 16 * outp;
    if (w.active) {
      const GemmPlan<1> p1 = make_plan<1>(W1, sd.L1, blk, 0, lds + sd.x1, sd.s1, lane, w.kc0a, w.kc1a);
      const GemmPlan<1> p2 = make_plan<1>(W2, sd.L2, blk, 0, lds + sd.x2, sd.s2, lane, w.kc0b, w.kc1b);
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
    pre = prefetch_fragments(Wn, sd.Ln, w, lane);
    __syncthreads();
    // combine: thread e owns the quad (row e / Q, units 4 (e % Q) ..), Q = outp / 4 <= 12
    const int Q = outp >> 2;
    const int e = threadIdx.x;
    if (e < 16 * Q) {
      const int r = (int)(((float)e + 0.5f) * __builtin_amdgcn_rcpf((float)Q)), n0 = 4 * (e - r * Q);
      f32x4 v = epi.init(n0);
      for (int p = 0; p < parts; ++p) v += lds4(P1 + (p * 16 + r) * outp + n0);
      epi.mid(v, r, n0);
      if (has2)
        for (int p = 0; p < parts; ++p) v += lds4(P2 + (p * 16 + r) * outp + n0);
      epi.fin(v, r, n0, u);
    }
    __syncthreads();
  }
}

template <int NW, class NET, int SI>
__device__ __forceinline__ void k2_run_stage(const TileArgs& a, float* lds, Pre& carry, const EpiCtx& ctx, int wave) {
  constexpr bool kStatic = !std::is_same<NET, void>::value;
  auto body = [&](const K2Stage& s, const WaveWorkS& w, const UnetDesc& u, const BwdLayout& lay) {
    const float* W1 = s.img1 ? a.packedT : a.packed;
    const float* W2 = s.img2 ? a.packedT : a.packed;
    const float* Wn = s.imgn ? a.packedT : a.packed;
    float* scratch = lds + lay.t.scratch;
    if (s.epi == EPI_RELU)       k2_stage<NW, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_RELU>{s, ctx}, u);
    else if (s.epi == EPI_RES)   k2_stage<NW, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_RES>{s, ctx}, u);
    else if (s.epi == EPI_MASK0) k2_stage<NW, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_MASK0>{s, ctx}, u);
    else if (s.epi == EPI_DUAL)  k2_stage<NW, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_DUAL>{s, ctx}, u);
    else                         k2_stage<NW, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_ACTMASK>{s, ctx}, u);
  };
  if constexpr (kStatic) {
    // every descriptor is a compile-time value: offsets fold into immediates, one NB variant and one epilogue survive
    constexpr UnetDesc u = NET::desc();
    constexpr BwdDesc bd = make_bwd_desc(u);
    constexpr BwdLayout lay = make_bwd_layout(u, NW);
    constexpr K2Stage s = k2_stage_desc(u, bd, lay, SI);
    const WaveWork w0 = wave_work_of(s.sd, NW, wave);
    WaveWorkS w;
    w.split = w0.split; w.blk0 = w0.blk0; w.cnt = w0.cnt; w.active = w0.active;
    w.kc0a = w0.kc0a; w.kc1a = w0.kc1a; w.kc0b = w0.kc0b; w.kc1b = w0.kc1b;
    w.part = w0.part; w.parts = w0.parts; w.use_pre = w0.use_pre;
#pragma unroll
    for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
    body(s, w, u, lay);
  } else {
    K2Stage s = a.prog.st[SI];
    s.sd = load_stage(a.prog.st[SI].sd);
    const WaveWork w0 = wave_work_of(s.sd, NW, wave);
    WaveWorkS w;
    w.split = w0.split; w.blk0 = w0.blk0; w.cnt = w0.cnt; w.active = w0.active;
    w.kc0a = w0.kc0a; w.kc1a = w0.kc1a; w.kc0b = w0.kc0b; w.kc1b = w0.kc1b;
    w.part = w0.part; w.parts = w0.parts; w.use_pre = w0.use_pre;
#pragma unroll
    for (int f = 0; f < 8; ++f) w.pf[f] = w0.pf[f];
    body(s, w, a.u, a.lay);
  }
}

// NET = StaticNet<...> (constexpr descriptors) or void (descriptors from the kernel arguments: any architecture)
template <int NW, class NET>
__global__ __launch_bounds__(NW * 64) void unet_bwd_tile_kernel(const TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool kStatic = !std::is_same<NET, void>::value;
  UnetDesc u;
  BwdLayout lay;
  if constexpr (kStatic) {
    constexpr UnetDesc uc = NET::desc();
    constexpr BwdLayout lc = make_bwd_layout(uc, NW);
    u = uc; lay = lc;
  } else {
    u = a.u; lay = a.lay;
  }
  const TileLayout& t = lay.t;
  const int tid = threadIdx.x, nthr = NW * 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int d = a.u.d, in0p = u.in0p, outp = u.outp;
  const int tile = blockIdx.x;
  const int64_t row0 = (int64_t)tile * 16;
  const int64_t tile_rows = (int64_t)a.ntiles * 16;
  float* X0 = lds + t.x0;
  float* G0 = lds + t.gv;
  // ---- input tile [t, x, 0..] and gradient tile (zero rows past N: they then contribute nothing to any gradient) ------
  float* slabX = a.ws + (size_t)tile_rows * tensor_prefix(u, T_X) + (size_t)tile * in0p * 16;
  float* slabG = a.ws + (size_t)tile_rows * tensor_prefix(u, T_G0) + (size_t)tile * outp * 16;
  for (int e = tid; e < 16 * in0p; e += nthr) {
    const int c = e >> 4, r = e & 15;                         // unit-major like the slab: coalesced slab stores
    const int64_t grow = min(row0 + r, a.N - 1);
    float v = 0.f;
    if (c == 0) v = a.ts[grow / a.rows_per_t];
    else if (c <= d) v = a.x[grow * d + c - 1];
    X0[r * t.s0 + c] = v;
    slabX[e] = v;
  }
  for (int e = tid; e < 16 * outp; e += nthr) {
    const int c = e >> 4, r = e & 15;
    const float v = (row0 + r < a.N && c < d) ? a.gout[(row0 + r) * d + c] : 0.f;
    G0[r * t.sg + c] = v;
    slabG[e] = v;
  }
  unet_load_biases(a.packed, u, t, lds, tid, nthr);
  EpiCtx ctx{lds, (a.dbg & 1) ? nullptr : a.ws, lds + t.bias, tile_rows, tile};
  // first ring of stage 0 (GEMM 1 = down_0 of the forward image)
  Pre carry;
  {
    LayerDesc L0; int img0 = 0;
    if constexpr (kStatic) { constexpr UnetDesc uc = NET::desc(); L0 = uc.L[0]; } else { L0 = a.u.L[0]; }
    (void)img0;
    unsigned short pf[8] = {};
    first_fragment_numbers(L0, NW, wave, pf);
    WaveWorkS w{};
#pragma unroll
    for (int f = 0; f < 8; ++f) w.pf[f] = pf[f];
    carry = prefetch_fragments(a.packed, L0, w, lane);
  }
  __syncthreads();
  k2_run_stage<NW, NET, 0>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 1>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 2>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 3>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 4>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 5>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 6>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 7>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 8>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 9>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, 10>(a, lds, carry, ctx, wave);
}

// ---- kernel B: weight / bias gradient partials -----------------------------------------------------------------------
struct WgItem {
  int gt_off, at_off;      // float offsets (without the tile term) of the gradient / activation tensors in the workspace
  int gW, aW;              // their widths
  int ob0, nob, ib0, nib;  // block group
  int part_off;            // float offset of block (ob0, ib0)'s 256-float cell inside a slab of partials
  int IB;                  // in-blocks of the layer (cells are ordered (ob, ib))
  int bias_off;            // float offset of the layer's bias partial inside a slab, or -1 when this group is not ib0 == 0
  int variant;             // index into the (NOB, NIB) instantiations
};

struct WgradArgs {
  // per forward layer: tensors, block counts, first item; block groups (<= 4 x 4 blocks) are numbered ig-fastest
  int gt_off[9], at_off[9], gW[9], aW[9], OB[9], IB[9], w_cell_off[9], b_part_off[9], item0[10];
  int n_items;
  int S;                   // slabs
  int ntiles;
  int64_t slab_floats;     // floats per slab of partials
  const float* ws;
  float* part;             // (S, slab_floats)
};

template <int NOB, int NIB>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const WgItem& it, int slab, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const int t0 = (int)(((int64_t)slab * a.ntiles) / a.S), t1 = (int)(((int64_t)(slab + 1) * a.ntiles) / a.S);
  const int64_t tile_rows = (int64_t)a.ntiles * 16;
  // lane (c, g) reads rows 4g .. 4g+3 of unit (block * 16 + c): one 16-byte load; MFMA number s of a tile takes component s
  // from every lane, i.e. k-slot g carries row 4g + s -- the same permutation for both operands
  const float* gbase = a.ws + (size_t)tile_rows * it.gt_off + (size_t)(it.ob0 * 16 + c) * 16 + 4 * g;
  const float* abase = a.ws + (size_t)tile_rows * it.at_off + (size_t)(it.ib0 * 16 + c) * 16 + 4 * g;
  const size_t gstep = (size_t)it.gW * 16, astep = (size_t)it.aW * 16;
  // block indices past the group (variants wider than the group) are clamped: computed twice, stored once
  int oo[NOB], io[NIB];
#pragma unroll
  for (int j = 0; j < NOB; ++j) oo[j] = min(j, it.nob - 1) * 256;
#pragma unroll
  for (int j = 0; j < NIB; ++j) io[j] = min(j, it.nib - 1) * 256;
  f32x4 acc[NOB][NIB];
  float bsum[NOB];
#pragma unroll
  for (int j = 0; j < NOB; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int k = 0; k < NIB; ++k) acc[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int PD = 3;                         // tiles in flight
  f32x4 ga[PD][NOB], ab[PD][NIB];
  auto load = [&](int t, int s) {
#pragma unroll
    for (int j = 0; j < NOB; ++j) ga[s][j] = *reinterpret_cast<const f32x4*>(gbase + (size_t)t * gstep + oo[j]);
#pragma unroll
    for (int k = 0; k < NIB; ++k) ab[s][k] = *reinterpret_cast<const f32x4*>(abase + (size_t)t * astep + io[k]);
  };
#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (t0 + s < t1) load(t0 + s, s);
  for (int t = t0; t < t1; t += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      if (t + s < t1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NOB; ++j)
#pragma unroll
            for (int k = 0; k < NIB; ++k)
              acc[j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s][j][i], ab[s][k][i], acc[j][k], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NOB; ++j) bsum[j] += (ga[s][j][0] + ga[s][j][1]) + (ga[s][j][2] + ga[s][j][3]);
        if (t + s + PD < t1) load(t + s + PD, s);
      }
    }
  }
  float* out = a.part + (size_t)slab * a.slab_floats;
  // D: lane (c, g) holds out-units 4g .. 4g+3 of in-unit c -> the cell's 256 floats in (lane, component) order
#pragma unroll
  for (int j = 0; j < NOB; ++j)
#pragma unroll
    for (int k = 0; k < NIB; ++k)
      if (j < it.nob && k < it.nib)
        *reinterpret_cast<f32x4*>(out + it.part_off + (size_t)(j * it.IB + k) * 256 + lane * 4) = acc[j][k];
  if (it.bias_off >= 0) {
#pragma unroll
    for (int j = 0; j < NOB; ++j) {
      float b = bsum[j];
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if (j < it.nob && lane < 16) out[it.bias_off + (it.ob0 + j) * 16 + lane] = b;
    }
  }
}

__global__ __launch_bounds__(256) void unet_wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slab = blockIdx.y * 4 + wave;
  if (slab >= a.S) return;
  int l = 8;
  while (l > 0 && (int)blockIdx.x < a.item0[l]) --l;
  WgItem it;
  {
    const int rel = (int)blockIdx.x - a.item0[l];
    const int nig = (a.IB[l] + 3) >> 2;
    const int og = rel / nig, ig = rel - og * nig;
    it.gt_off = a.gt_off[l]; it.at_off = a.at_off[l]; it.gW = a.gW[l]; it.aW = a.aW[l];
    it.ob0 = 4 * og; it.nob = min(4, a.OB[l] - it.ob0);
    it.ib0 = 4 * ig; it.nib = min(4, a.IB[l] - it.ib0);
    it.IB = a.IB[l];
    it.part_off = a.w_cell_off[l] + (it.ob0 * a.IB[l] + it.ib0) * 256;
    it.bias_off = it.ib0 == 0 ? a.b_part_off[l] : -1;
    const int vo = it.nob > 2 ? 4 : (it.nob > 1 ? 2 : 1), vi = it.nib > 2 ? 4 : (it.nib > 1 ? 2 : 1);
    // instantiated: (4,4) (4,2) (2,4) (2,2) (4,1) (1,4) (1,1); the two remaining shapes run the next larger one
    it.variant = (vo == 4 && vi == 4) ? 0 : (vo == 4 && vi == 2) ? 1 : (vo == 2 && vi == 4) ? 2 : (vo == 2 && vi == 2) ? 3 :
                 (vo == 4 && vi == 1) ? 4 : (vo == 1 && vi == 4) ? 5 : (vo == 1 && vi == 1) ? 6 : (vo == 2) ? 1 : 2;
  }
  switch (it.variant) {
    case 0: wgrad_body<4, 4>(a, it, slab, lane); break;
    case 1: wgrad_body<4, 2>(a, it, slab, lane); break;
    case 2: wgrad_body<2, 4>(a, it, slab, lane); break;
    case 3: wgrad_body<2, 2>(a, it, slab, lane); break;
    case 4: wgrad_body<4, 1>(a, it, slab, lane); break;
    case 5: wgrad_body<1, 4>(a, it, slab, lane); break;
    default: wgrad_body<1, 1>(a, it, slab, lane); break;
  }
}

// ---- kernel C: add the slabs (fixed order), scatter into torch layout ----------------------------------------------------
struct FinishArgs {
  int w_cell_off[9];       // float offset of layer l's first cell inside a slab
  int b_part_off[9];       // ... of its bias partial
  int OB[9], IB[9];
  int fin[9], fout[9];
  int64_t gw_off[9], gb_off[9];   // offsets into the flat gradient buffer (torch layout: weight (out, in), bias (out,))
  int total_cells_floats;  // sum over layers of OB * IB * 256
  int total_bias;          // sum of padded fan-outs
  int S;
  int64_t slab_floats;
  const float* part;
  float* grads;
};

__global__ __launch_bounds__(256) void unet_wgrad_finish_kernel(const FinishArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < a.total_cells_floats) {
    int l = 8;
    while (l > 0 && idx < a.w_cell_off[l]) --l;
    const int rel = idx - a.w_cell_off[l];
    const int cell = rel >> 8, lane = (rel >> 2) & 63, rr = rel & 3;
    const int ob = cell / a.IB[l], ib = cell - ob * a.IB[l];
    const int o = ob * 16 + 4 * (lane >> 4) + rr, i = ib * 16 + (lane & 15);
    if (o < a.fout[l] && i < a.fin[l]) {
      float s = 0.f;
      for (int p = 0; p < a.S; ++p) s += a.part[(size_t)p * a.slab_floats + idx];
      a.grads[a.gw_off[l] + (int64_t)o * a.fin[l] + i] = s;
    }
    return;
  }
  const int b = idx - a.total_cells_floats;
  if (b < a.total_bias) {
    int l = 8;
    while (l > 0 && a.total_cells_floats + b < a.b_part_off[l]) --l;
    const int o = a.total_cells_floats + b - a.b_part_off[l];
    if (o < a.fout[l]) {
      float s = 0.f;
      for (int p = 0; p < a.S; ++p) s += a.part[(size_t)p * a.slab_floats + a.b_part_off[l] + o];
      a.grads[a.gb_off[l] + o] = s;
    }
  }
}

// ---- transposed weight image ------------------------------------------------------------------------------------------
struct PackTArgs {
  BwdDesc bd;
  int fin[KT_N], fout[KT_N];      // fan-in / fan-out of the FORWARD layer each transposed layer comes from
  const float* w[KT_N];
  float* packedT;
};

__global__ void unet_pack_bwd_kernel(const PackTArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.bd.total_floats) return;
  int t = KT_N - 1;
  while (t > 0 && idx < a.bd.L[t].w_off) --t;
  const LayerDesc L = a.bd.L[t];
  const int rel = idx - L.w_off;
  const int i = rel & 3, lane = (rel >> 2) & 63, chunk = rel >> 8;
  const int KC = L.in_pad >> 4;
  const int nb = chunk / KC, kc = chunk - nb * KC;
  const int n = nb * 16 + (lane & 15), kk = kc * 16 + 4 * (lane >> 4) + i;   // W^T[n][kk] = W[kk][n]
  float v = 0.f;
  if (n < a.fin[t] && kk < a.fout[t]) v = a.w[t][(size_t)kk * a.fin[t] + n];
  a.packedT[idx] = v;
}

#endif  // __HIPCC__
}  // namespace socmx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace socmx;

static bool k2_dims_ok(int d, const int32_t h[3]) {
  if (d < 1 || d > 1024) return false;
  for (int i = 0; i < 3; ++i)
    if (h[i] < 1 || h[i] > 4096) return false;
  return true;
}

struct K2Plan {
  UnetDesc u;
  BwdDesc bd;
  BwdLayout lay;
  int ntiles, S;
  int n_items;
  int64_t ws_floats, slab_floats, part_floats;
  int w_cell_off[9], b_part_off[9], OB[9], IB[9];
  int total_cells_floats, total_bias;
  int fin[9], fout[9];
  int64_t gw_off[9], gb_off[9], grad_floats;
};

static const int kK2Waves = 8;

static int k2_plan(int32_t d, const int32_t hdims[3], int64_t N, K2Plan& p) {
  if (!hdims) return SOCMX_E_NULL;
  if (!k2_dims_ok(d, hdims) || N < 1) return SOCMX_E_DIM;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  p.u = make_unet_desc(d, h);
  p.bd = make_bwd_desc(p.u);
  p.lay = make_bwd_layout(p.u, kK2Waves);
  if ((size_t)p.lay.floats * sizeof(float) > (size_t)kLdsBytesPerCU) return SOCMX_E_LDS;
  if ((N + 15) / 16 > (int64_t)1 << 27) return SOCMX_E_DIM;
  p.ntiles = (int)((N + 15) / 16);
  int wsum = 0;
  for (int t = 0; t < T_N; ++t) wsum += tensor_width(p.u, t);
  p.ws_floats = (int64_t)p.ntiles * 16 * wsum;
  unet_layer_dims(d, h, p.fin, p.fout);
  int off = 0, items = 0;
  int64_t goff = 0;
  for (int l = 0; l < 9; ++l) {
    p.OB[l] = p.u.L[l].out_pad >> 4;
    p.IB[l] = p.u.L[l].in_pad >> 4;
    p.w_cell_off[l] = off;
    off += p.OB[l] * p.IB[l] * 256;
    items += ((p.OB[l] + 3) / 4) * ((p.IB[l] + 3) / 4);
    p.gw_off[l] = goff; goff += (int64_t)p.fin[l] * p.fout[l];
    p.gb_off[l] = goff; goff += p.fout[l];
  }
  p.grad_floats = goff;
  p.total_cells_floats = off;
  for (int l = 0; l < 9; ++l) { p.b_part_off[l] = off; off += p.u.L[l].out_pad; }
  p.total_bias = off - p.total_cells_floats;
  p.slab_floats = off;
  p.n_items = items;
  // Slabs of row tiles for kernel B (one wave per (block group, slab)): all waves should be resident at once -- a few
  // waves past the chip's capacity would run a second round alone and double the kernel's time.  142 VGPRs -> 3 waves
  // per SIMD = 12 per CU; at least 4 tiles per slab.
  static const int n_cus = [] { int v = 256, dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256; return v > 0 ? v : 256; }();
  int S = (int)(0.97 * n_cus * 12) / items;
  if (S > p.ntiles / 4) S = p.ntiles / 4;
  p.S = S < 1 ? 1 : (S > 128 ? 128 : S);
  p.part_floats = (int64_t)p.S * p.slab_floats;
  return 0;
}

extern "C" size_t socmx_unet_packed_bwd_floats(int32_t d, const int32_t hdims[3]) {
  if (!hdims || !k2_dims_ok(d, hdims)) return 0;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  return (size_t)make_bwd_desc(make_unet_desc(d, h)).total_floats;
}

extern "C" int socmx_unet_pack_bwd_f32(const socmx_unet* net, float* packedT, socmx_stream_t stream) {
  if (!net || !packedT) return SOCMX_E_NULL;
  if (!k2_dims_ok(net->d, net->hdims)) return SOCMX_E_DIM;
  PackTArgs a;
  const int h[3] = {net->hdims[0], net->hdims[1], net->hdims[2]};
  const UnetDesc u = make_unet_desc(net->d, h);
  a.bd = make_bwd_desc(u);
  int fin[9], fout[9];
  unet_layer_dims(net->d, h, fin, fout);
  for (int t = 0; t < KT_N; ++t) {
    const int l = kt_source(t);
    if (!net->weight[l]) return SOCMX_E_NULL;
    a.w[t] = net->weight[l];
    a.fin[t] = fin[l];
    a.fout[t] = fout[l];
  }
  a.packedT = packedT;
  const int threads = 256, blocks = (a.bd.total_floats + threads - 1) / threads;
  return launch(unet_pack_bwd_kernel, dim3(blocks), dim3(threads), 0, stream, a);
}

extern "C" int socmx_unet_backward_sizes(int32_t d, const int32_t hdims[3], int64_t N, int64_t* workspace_floats,
                                         int64_t* grad_floats) {
  K2Plan p;
  const int rc = k2_plan(d, hdims, N, p);
  if (rc) return rc;
  if (workspace_floats) *workspace_floats = p.ws_floats + p.part_floats;
  if (grad_floats) *grad_floats = p.grad_floats;
  return 0;
}

extern "C" int socmx_unet_backward_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                       const float* x, const float* ts, int32_t rows_per_t, int64_t N,
                                       const float* gout, float* workspace, float* grads, socmx_stream_t stream) {
  if (!packed || !packedT || !x || !ts || !gout || !workspace || !grads) return SOCMX_E_NULL;
  if (rows_per_t < 1) return SOCMX_E_DIM;
  K2Plan p;
  if (const int rc = k2_plan(d, hdims, N, p)) return rc;
  // ---- kernel A ----
  TileArgs ta;
  ta.u = p.u; ta.bd = p.bd; ta.lay = p.lay;
  for (int si = 0; si < kBwdStages; ++si) ta.prog.st[si] = k2_stage_desc(p.u, p.bd, p.lay, si);
  ta.packed = packed; ta.packedT = packedT; ta.x = x; ta.ts = ts; ta.gout = gout; ta.ws = workspace;
  ta.N = N; ta.rows_per_t = rows_per_t; ta.ntiles = p.ntiles;
  { const char* e = getenv("SOCMX_K2_DBG"); ta.dbg = e ? atoi(e) : 0; }
  const size_t lds_bytes = (size_t)p.lay.floats * sizeof(float);
  static const bool force_generic = getenv("SOCMX_GENERIC") != nullptr;
  const UnetDesc& u = p.u;
  const bool widths_default = !force_generic && u.hp[0] == 256 && u.hp[1] == 128 && u.hp[2] == 64;
  void (*kern)(const TileArgs) = unet_bwd_tile_kernel<kK2Waves, void>;
  if (widths_default && u.in0p == 16 && u.outp == 16) kern = unet_bwd_tile_kernel<kK2Waves, StaticNet<16, 256, 128, 64, 16>>;
  else if (widths_default && u.in0p == 32 && u.outp == 32) kern = unet_bwd_tile_kernel<kK2Waves, StaticNet<32, 256, 128, 64, 32>>;
  else if (widths_default && u.in0p == 80 && u.outp == 64) kern = unet_bwd_tile_kernel<kK2Waves, StaticNet<80, 256, 128, 64, 64>>;
  if (const int err = ensure_max_lds(kern)) return err;
  if (const int err = launch(kern, dim3(p.ntiles), dim3(kK2Waves * 64), lds_bytes, stream, ta)) return err;
  // ---- kernel B ----
  WgradArgs wa;
  wa.n_items = 0; wa.S = p.S; wa.ntiles = p.ntiles; wa.slab_floats = p.slab_floats;
  wa.ws = workspace; wa.part = workspace + p.ws_floats;
  for (int l = 0; l < 9; ++l) {
    const int gt = layer_grad_tensor(l), at = layer_act_tensor(l);
    wa.gt_off[l] = tensor_prefix(p.u, gt); wa.at_off[l] = tensor_prefix(p.u, at);
    wa.gW[l] = tensor_width(p.u, gt); wa.aW[l] = tensor_width(p.u, at);
    wa.OB[l] = p.OB[l]; wa.IB[l] = p.IB[l]; wa.w_cell_off[l] = p.w_cell_off[l]; wa.b_part_off[l] = p.b_part_off[l];
    wa.item0[l] = wa.n_items;
    wa.n_items += ((p.OB[l] + 3) / 4) * ((p.IB[l] + 3) / 4);
  }
  wa.item0[9] = wa.n_items;
  if (const int err = launch(unet_wgrad_kernel, dim3(wa.n_items, (p.S + 3) / 4), dim3(256), 0, stream, wa)) return err;
  // ---- kernel C ----
  FinishArgs fa;
  for (int l = 0; l < 9; ++l) {
    fa.w_cell_off[l] = p.w_cell_off[l]; fa.b_part_off[l] = p.b_part_off[l]; fa.OB[l] = p.OB[l]; fa.IB[l] = p.IB[l];
    fa.fin[l] = p.fin[l]; fa.fout[l] = p.fout[l]; fa.gw_off[l] = p.gw_off[l]; fa.gb_off[l] = p.gb_off[l];
  }
  fa.total_cells_floats = p.total_cells_floats; fa.total_bias = p.total_bias; fa.S = p.S; fa.slab_floats = p.slab_floats;
  fa.part = wa.part; fa.grads = grads;
  const int nthreads = p.total_cells_floats + p.total_bias;
  return launch(unet_wgrad_finish_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, stream, fa);
}
