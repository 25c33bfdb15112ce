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
//   D  unet_unfold_kernel      the skip res_1 is FOLDED through up_0 in all of the above (see "the fold" below): this kernel
//                              turns G' = ZU0^T R1 (d x h0) into dW_res_1, db_res_1 and the skip's share of dW_up_0.
// Bound: fp32 MFMA (A: forward + activation-gradient GEMMs = 2 x 2 x MACs per row; B: 2 x MACs per row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cstring>

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
    // (KT_R1: not res_1^T but F^T = (up_0 res_1)^T -- the fold: outp inputs instead of h0)
    const LayerDesc& f = t == KT_R1 ? u.fold : u.L[kt_source(t)];
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

// (the slab tensors T_* that travel from kernel A to kernel B, their widths and prefixes: socmx_unet.h -- the one-row rollout exports the
//  activation slabs itself when asked to, csrc/socmx_rollout1.hip)
// (gradient tensor, activation tensor) of forward layer l: dW_l = grad^T . act
__host__ __device__ constexpr int layer_grad_tensor(int l) {
  constexpr int g[9] = {T_ZD0, T_ZD1, T_ZD2, T_G0, T_ZU0 /* the fold: G' = ZU0^T R1, see unet_unfold_kernel */, T_GO2, T_ZU2, T_ZU1, T_ZU0};
  return g[l];
}
__host__ __device__ constexpr int layer_act_tensor(int l) {
  constexpr int a[9] = {T_X, T_R1, T_R2, T_X, T_R1, T_R2, T_R3, T_O2, T_O1};
  return a[l];
}

// ---- LDS of kernel A ----------------------------------------------------------------------------------------------------
// Forward tiles X0 R1 R2 R3 O2 O1 G (row stride width + 4, as in socmx_unet.h).  Every forward tile is dead once the
// forward part has consumed and exported it -- only the SIGN of R1/R2/R3 and of the up-path pre-activations is needed
// later, kept as nibble masks (4 units per byte) -- so the backward tiles reuse the same memory:
//     ZU0 -> X0,  GO1 -> R1,  ZU1 -> O1,  GO2 -> R2,  ZU2 -> O2,  ZD2 -> R3;   ZD1 has its own tile, which also holds the
// biases (read by the forward stages only) until stage 9 writes it.  The split-K scratch is sized by the stages that
// split.  Default widths: 75-77 KiB per workgroup = two workgroups per CU (the barriers of one overlap the MFMAs of
// the other).
struct BwdLayout {
  TileLayout t;                                   // x0, r1, r2, r3, o2, o1, gv (= the G tile here), scratch, bias
  int zu0, go1, zu1, go2, zu2, zd2, zd1;          // float offsets
  int mu2, mu1, mr1, mr2, mr3;                    // float offsets of the byte arrays [16][width / 4]
  int floats;
};

__host__ __device__ constexpr int k2_scratch_floats(const UnetDesc& u, int nwaves, int rt);

// rt = 16-row tiles per workgroup (the tiles hold 16 rt rows): with rt = 2 every weight fragment that arrives from L2
// feeds two MFMA column tiles -- half the weight traffic and half the barriers per row
#ifndef SOCMX_K2S_ALIAS
#define SOCMX_K2S_ALIAS 1        /* SAVED form: ZD2 / ZD1 in ZU1's place (developer A/B: 0) */
#endif
#ifndef SOCMX_K2S_WGS
#define SOCMX_K2S_WGS 4          /* SAVED form: workgroups per CU the launch bounds ask for */
#endif
__host__ __device__ constexpr BwdLayout make_bwd_layout(const UnetDesc& u, int nwaves, int rt, bool saved = false) {
  BwdLayout b{};
  TileLayout& t = b.t;
  const int rows = 16 * rt;
  t.s0 = u.in0p + 4; t.s1 = u.hp[0] + 4; t.s2 = u.hp[1] + 4; t.s3 = u.hp[2] + 4; t.sg = u.outp + 4;
  int off = 0;
  if (saved) {
    // The SAVED form (no forward stages): only the tiles the backward chain READS -- ZU0, ZU1, GO2, ZU2, ZD2, ZD1, the G tile and the
    // sixteen sign records; GO1 has no reader (its skip is folded: stage 10 multiplies F^T by ZU0) and is not stored.  51 KiB at the
    // default widths: THREE workgroups per CU where the re-computing form's 77 KiB admit two.
    b.zu0 = off; t.x0 = off; off += rows * t.sg;
    t.gv = off; off += rows * t.sg;
    b.go1 = -1; t.r1 = -1;
    b.zu1 = off; t.o1 = off; off += rows * t.s1;
    b.go2 = off; t.r2 = off; off += rows * t.s2;
    b.zu2 = off; t.o2 = off; off += rows * t.s2;
    if (SOCMX_K2S_ALIAS && rows * (t.s3 + t.s2) <= rows * t.s1) {
      // ZU1 is read by stage 7 only: ZD2 (written by stage 8) and ZD1 (stage 9) live in its place -- 38 KiB at the default widths, FOUR workgroups per CU
      b.zd2 = b.zu1; t.r3 = b.zd2;
      b.zd1 = b.zu1 + rows * t.s3; t.bias = b.zd1;
    } else {
      b.zd2 = off; t.r3 = off; off += rows * t.s3;
      b.zd1 = off; t.bias = off; off += rows * t.s2;
    }
    b.mu2 = off; b.mu1 = off; b.mr1 = off; b.mr2 = off; b.mr3 = off; off += rows * kActRecordDwords;
    t.scratch = off;          // (no backward stage splits its reduction)
    t.floats = off;
    b.floats = off;
    return b;
  }
  t.x0 = off; off += rows * t.s0;
  t.r1 = off; off += rows * t.s1;
  t.r2 = off; off += rows * t.s2;
  t.r3 = off; off += rows * t.s3;
  t.o2 = off; off += rows * t.s2;
  t.o1 = off; off += rows * t.s1;
  t.gv = off; off += rows * t.sg;
  b.zu0 = t.x0; b.go1 = t.r1; b.zu1 = t.o1; b.go2 = t.r2; b.zu2 = t.o2; b.zd2 = t.r3;
  b.zd1 = off;
  t.bias = off;
  const int zd1_floats = rows * t.s2;
  off += zd1_floats > u.bias_floats ? zd1_floats : u.bias_floats;
  b.mu2 = off; off += rt * u.hp[1];               // rows x (width / 4) bytes = rt x width floats
  b.mu1 = off; off += rt * u.hp[0];
  b.mr1 = off; off += rt * u.hp[0];
  b.mr2 = off; off += rt * u.hp[1];
  b.mr3 = off; off += rt * u.hp[2];
  t.scratch = off; off += k2_scratch_floats(u, nwaves, rt);
  t.floats = off;
  b.floats = off;
  return b;
}

// ---- the eleven stages of kernel A ------------------------------------------------------------------------------------
//  forward   0: R1 = relu(d0 X + b)      1: R2 = relu(d1 R1 + b)     2: R3 = relu(d2 R2 + b)
//            3: O2 = relu(u2 R3 + b) + r2 R2 + b   [mask MU2]        4: A1 = relu(u1 O2 + b)               [mask MU1]
//            5: MU0 = (u0 A1 + F R1 + f + b > 0);  ZU0 = G (.) MU0   (the output itself is not needed: res_0 is skipped)
//  backward  6: GO1 = u0^T ZU0;  ZU1 = GO1 (.) MU1                   7: GO2 = u1^T ZU1;  ZU2 = GO2 (.) MU2
//            8: ZD2 = (u2^T ZU2) (.) [R3 > 0]                         9: ZD1 = (d2^T ZD2 + r2^T GO2) (.) [R2 > 0]
//           10: ZD0 = (d1^T ZD1 + F^T ZU0) (.) [R1 > 0]
// THE FOLD (round 5).  models.py:239-240: o1 = relu(u1 o2 + b) + r1 R1 + b4 enters the output's ReLU only through the linear u0, so
// with F = u0 r1 (outp x h0) and f = u0 b4:  u0 o1 = u0 A1 + F R1 + f.  The 256 x 256 skip product disappears from the forward
// (stage 4), from the backward chain (r1^T GO1 = r1^T u0^T ZU0 = F^T ZU0: stage 10 multiplies a 16-wide tile) and from the weight
// gradients: dW_r1 = GO1^T R1 = u0^T (ZU0^T R1) and the skip's share of dW_u0 = ZU0^T o1 is (ZU0^T R1) r1^T + (sum ZU0) b4^T --
// kernel B forms G' = ZU0^T R1 (outp x h0) in layer 4's slot, kernel D the three small products.  39 % of the network's
// multiply-adds at the default widths, and the GO1 slab (h0 floats per row) is not exported any more.
constexpr int kBwdStages = 11;
enum { EPI_RELU = 0, EPI_RES, EPI_MASK0, EPI_DUAL, EPI_ACTMASK };

struct K2Stage {
  StageDesc sd;          // L1, L2, Ln (GEMM 1 of the stage that follows), x1/s1, x2/s2, y/sy (primary output tile), has2
  int img1, img2, imgn;  // weight image of L1 / L2 / Ln: 0 = forward image, 1 = transposed image
  int epi;
  int y2, sy2;           // second tile (EPI_DUAL: the masked copy; EPI_MASK0: ZU0) -- float offset / stride
  int mask;              // float offset of the nibble-mask byte array written (EPI_RES, EPI_MASK0) or read (EPI_DUAL)
  int mkind;             // MK_* of the mask a backward stage reads (the SAVED form looks it up in the rows' records, socmx_unet.h)
  int aux, saux;         // EPI_ACTMASK: activation tile whose sign masks the result; EPI_MASK0: the G tile
  int ex1, ex2;          // exported tensor ids (T_*) of the primary / second result, -1 = none
  int w1, w2;            // their widths
  int p1, p2;            // ... and the summed widths of the tensors before them (workspace offsets / (16 ntiles))
};

__host__ __device__ constexpr K2Stage make_k2_stage(const UnetDesc& u, const BwdDesc& bd, const BwdLayout& b, int si) {
  const TileLayout& t = b.t;
  K2Stage s{};
  s.y2 = -1; s.mask = -1; s.aux = -1; s.ex1 = -1; s.ex2 = -1; s.mkind = MK_NONE;
  auto fwd = [&](int l1, int x1, int s1, int has2, int l2, int x2, int s2, int y, int sy) {
    s.sd.L1 = u.L[l1]; s.sd.L2 = u.L[l2]; s.sd.x1 = x1; s.sd.s1 = s1; s.sd.x2 = x2; s.sd.s2 = s2; s.sd.y = y; s.sd.sy = sy;
    s.sd.has2 = has2; s.img1 = 0; s.img2 = 0;
  };
  auto bwd = [&](int l1, int x1, int s1, int has2, int l2, int x2, int s2, int y, int sy) {
    s.sd.L1 = bd.L[l1]; s.sd.L2 = bd.L[l2]; s.sd.x1 = x1; s.sd.s1 = s1; s.sd.x2 = x2; s.sd.s2 = s2; s.sd.y = y; s.sd.sy = sy;
    s.sd.has2 = has2; s.img1 = 1; s.img2 = 1;
  };
  switch (si) {
    case 0: fwd(0, t.x0, t.s0, 0, 0, t.x0, t.s0, t.r1, t.s1); s.epi = EPI_RELU; s.mask = b.mr1; s.ex1 = T_R1; break;
    case 1: fwd(1, t.r1, t.s1, 0, 1, t.r1, t.s1, t.r2, t.s2); s.epi = EPI_RELU; s.mask = b.mr2; s.ex1 = T_R2; break;
    case 2: fwd(2, t.r2, t.s2, 0, 2, t.r2, t.s2, t.r3, t.s3); s.epi = EPI_RELU; s.mask = b.mr3; s.ex1 = T_R3; break;
    case 3: fwd(6, t.r3, t.s3, 1, 5, t.r2, t.s2, t.o2, t.s2); s.epi = EPI_RES; s.mask = b.mu2; s.ex1 = T_O2; break;
    case 4: fwd(7, t.o2, t.s2, 0, 7, t.o2, t.s2, t.o1, t.s1); s.epi = EPI_RELU; s.mask = b.mu1; s.ex1 = T_O1; break;   // (T_O1 carries A1)
    case 5: fwd(8, t.o1, t.s1, 1, 8, t.r1, t.s1, -1, 0); s.sd.L2 = u.fold; s.epi = EPI_MASK0; s.aux = t.gv; s.saux = t.sg;
            s.y2 = b.zu0; s.sy2 = t.sg; s.ex2 = T_ZU0; break;
    case 6: bwd(KT_U0, b.zu0, t.sg, 0, KT_U0, b.zu0, t.sg, b.go1, t.s1); s.epi = EPI_DUAL; s.mask = b.mu1; s.mkind = MK_U1;
            s.y2 = b.zu1; s.sy2 = t.s1; s.ex1 = -1 /* GO1 stays in LDS: nobody downstream reads its slab */; s.ex2 = T_ZU1; break;
    case 7: bwd(KT_U1, b.zu1, t.s1, 0, KT_U1, b.zu1, t.s1, b.go2, t.s2); s.epi = EPI_DUAL; s.mask = b.mu2; s.mkind = MK_U2;
            s.y2 = b.zu2; s.sy2 = t.s2; s.ex1 = T_GO2; s.ex2 = T_ZU2; break;
    case 8: bwd(KT_U2, b.zu2, t.s2, 0, KT_U2, b.zu2, t.s2, b.zd2, t.s3); s.epi = EPI_ACTMASK; s.mask = b.mr3; s.mkind = MK_R3;
            s.ex1 = T_ZD2; break;
    case 9: bwd(KT_D2, b.zd2, t.s3, 1, KT_R2, b.go2, t.s2, b.zd1, t.s2); s.epi = EPI_ACTMASK; s.mask = b.mr2; s.mkind = MK_R2;
            s.ex1 = T_ZD1; break;
    default: bwd(KT_D1, b.zd1, t.s2, 1, KT_R1 /* F^T */, b.zu0, t.sg, -1, 0); s.epi = EPI_ACTMASK; s.mask = b.mr1; s.mkind = MK_R1;
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

// split-K scratch of kernel A: the largest (1 + has2) * parts * 16 * out_pad over the stages whose GEMM 1 has fewer than four
// 16-wide output blocks (the work-split rule of wave_work_of)
__host__ __device__ constexpr int k2_scratch_floats(const UnetDesc& u, int nwaves, int rt) {
  const BwdDesc bd = make_bwd_desc(u);
  // (out_pad of GEMM 1, has2) per stage, in stage order
  const int outs[kBwdStages] = {u.hp[0], u.hp[1], u.hp[2], u.hp[1], u.hp[0], u.outp,
                                bd.L[KT_U0].out_pad, bd.L[KT_U1].out_pad, bd.L[KT_U2].out_pad, bd.L[KT_D2].out_pad,
                                bd.L[KT_D1].out_pad};
  const int two[kBwdStages] = {0, 0, 0, 1, 0, 0 /* stage 5: GEMM 2 adds into GEMM 1's partial sums */, 0, 0, 0, 1, 1};
  int need = 0;
  for (int si = 0; si < kBwdStages; ++si) {
    const int nblk = outs[si] >> 4;
    if (nblk >= kSimds || nblk >= nwaves) continue;
    const int parts = nwaves / nblk;
    const int n = (1 + two[si]) * parts * 16 * rt * outs[si];
    need = n > need ? n : need;
  }
  return need;
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
  const float* gscale;    // (1,) device scalar that multiplies gout, or nullptr
  float* ws;              // workspace: T_N tensors, tensor t at ws + 16 * ntiles * prefix(t), each [tile][width][16]
  const uint32_t* rec;    // SAVED form: the rows' sign records (socmx_unet.h), written by the rollout beside the activation slabs
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
  for (int i = 0; i < 4; ++i) slab[(size_t)(n0 + i) * 16 + (r & 15)] = v[i];
}
// the same from the direct stages, where the tile's slab and the 16-unit block are wave-uniform: scalar base + ONE lane
// offset, the four units as immediates (no per-store 64-bit address arithmetic: 3 vector instructions per store before)
// (buffer stores: the block's base in a scalar resource, the lane offset in one 32-bit register -- a global store carries a
//  64-bit address per lane, and issuing those is time the SIMD's matrix pipe does not get back, see socmx_rollout32.hip)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* ubase) {       // ubase: wave-uniform
  const uint64_t u = reinterpret_cast<uint64_t>(ubase);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void export4_block(float* slab_block, uint32_t lane_off, const f32x4 v) {
  if (!slab_block) return;
  const __amdgpu_buffer_rsrc_t r = uniform_rsrc(slab_block);
#pragma unroll
  for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), r, (int)lane_off, 64 * i, 0);
}
// export4 with a wave-uniform slab (the pair network's tiles: one per workgroup)
__device__ __forceinline__ void export4_u(float* slab, int r, int n0, const f32x4 v) {
  const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(slab);
  const int off = (n0 * 16 + (r & 15)) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), rs, off, 64 * i, 0);
}

struct EpiCtx {
  float* lds;
  float* ws;               // workspace base
  const float* bias_lds;
  int64_t tile_rows;       // 16 * ntiles
  int tile;                // first 16-row tile of this workgroup (row r of the LDS tiles belongs to tile + r / 16)
  const float* fold_bias;  // f = up_0 b_res_1 (global memory, the forward image): added to stage 5's pre-activation
  const uint32_t* rec_lds; // SAVED form: the tile's sixteen records in LDS
};

template <int EPI, bool SAVED = false>
struct Epi {
  const K2Stage& s;
  const EpiCtx& c;
  // the nibble (units n0 .. n0 + 3 of row r) of the mask this backward stage reads
  __device__ __forceinline__ unsigned nibble(int r, int n0) const {
    if constexpr (SAVED) return act_record_nibble(c.rec_lds + r * kActRecordDwords, s.mkind, n0);
    return reinterpret_cast<const unsigned char*>(c.lds + s.mask)[r * (s.sd.L1.out_pad >> 2) + (n0 >> 2)];
  }
  __device__ __forceinline__ float* slab(int prefix, int width, int r) const {
    return c.ws ? c.ws + (size_t)c.tile_rows * prefix + (size_t)(c.tile + (r >> 4)) * width * 16 : nullptr;
  }
  // h >= 0: called from a direct stage for row tile h and 16-unit block nblk (both wave-uniform)
  __device__ __forceinline__ void put(int prefix, int width, int r, int n0, const f32x4 v, int h, int nblk) const {
    if (h >= 0) {
      float* sb = c.ws ? c.ws + (size_t)c.tile_rows * prefix + ((size_t)(c.tile + h) * width + (size_t)nblk * 16) * 16 : nullptr;
      export4_block(sb, (uint32_t)(((n0 & 15) * 16 + (r & 15)) * 4), v);
    } else {
      export4(slab(prefix, width, r), r, n0, v);
    }
  }
  // EPI_MASK0: only the sign of L1 X1 + L2 X2 + biases matters -- a split stage adds GEMM 2's partial sums to GEMM 1's
  static constexpr bool kFuse2 = EPI == EPI_MASK0;
  __device__ __forceinline__ f32x4 init(int n0) const {
    if constexpr (EPI == EPI_MASK0) return lds4(c.bias_lds + s.sd.L1.b_lds + n0) + *reinterpret_cast<const f32x4*>(c.fold_bias + n0);
    if constexpr (EPI == EPI_RELU || EPI == EPI_RES) return lds4(c.bias_lds + s.sd.L1.b_lds + n0);
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
  __device__ __forceinline__ void fin(f32x4 v, int r, int n0, const UnetDesc& u, int h = -1, int nblk = 0) const {
    if constexpr (EPI == EPI_RELU) {
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { m |= (v[i] > 0.f ? 1u : 0u) << i; v[i] = relu_keep_nan(v[i]); }
      reinterpret_cast<unsigned char*>(c.lds + s.mask)[r * (s.sd.L1.out_pad >> 2) + (n0 >> 2)] = (unsigned char)m;
      *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;
      put(s.p1, s.w1, r, n0, v, h, nblk);
    } else if constexpr (EPI == EPI_RES) {
      *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;
      put(s.p1, s.w1, r, n0, v, h, nblk);
    } else if constexpr (EPI == EPI_MASK0) {
      const f32x4 g = lds4(c.lds + s.aux + r * s.saux + n0);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = v[i] > 0.f ? g[i] : 0.f;
      *reinterpret_cast<f32x4*>(c.lds + s.y2 + r * s.sy2 + n0) = z;
      put(s.p2, s.w2, r, n0, z, h, nblk);
    } else if constexpr (EPI == EPI_DUAL) {
      const unsigned m = nibble(r, n0);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = ((m >> i) & 1u) ? v[i] : 0.f;
      if (s.sd.y >= 0) *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = v;      // (SAVED: GO1 has no reader and no tile)
      *reinterpret_cast<f32x4*>(c.lds + s.y2 + r * s.sy2 + n0) = z;
      if (s.ex1 >= 0) put(s.p1, s.w1, r, n0, v, h, nblk);
      put(s.p2, s.w2, r, n0, z, h, nblk);
    } else {   // EPI_ACTMASK: the sign of the forward activation, saved as a nibble by that stage's EPI_RELU
      const unsigned m = nibble(r, n0);
      f32x4 z;
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] = ((m >> i) & 1u) ? v[i] : 0.f;
      if (s.sd.y >= 0) *reinterpret_cast<f32x4*>(c.lds + s.sd.y + r * s.sd.sy + n0) = z;
      put(s.p1, s.w1, r, n0, z, h, nblk);
    }
  }
};

// gemm_run of socmx_unet.h with RT batch-column tiles per weight fragment: acc[h][j] += W-block j . X rows [16 h, 16 h + 16)
template <int NB, int RT>
__device__ __forceinline__ void mfma_chunk_rt(f32x4 (&acc)[RT][NB], const f32x4 (&a)[NB], const f32x4 (&bx)[RT]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int h = 0; h < RT; ++h) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], bx[h][i], acc[h][j], 0, 0, 0);
}

template <int NB, int RT, bool PINNED>
__device__ __forceinline__ void k2_gemm_run(f32x4 (&acc)[RT][NB], Ring<NB>& r, const GemmPlan<NB>& p, int half) {
  constexpr int PD = Ring<NB>::PD;
  const int last = p.kc1 - 1;
  int kc = p.kc0;
  f32x4 bx[RT], bx_next[RT];
#pragma unroll
  for (int h = 0; h < RT; ++h) {
    bx[h] = lds4(p.xrow + h * half + min(kc, last) * 16);
    bx_next[h] = lds4(p.xrow + h * half + min(kc + 1, last) * 16);
  }
  for (; kc + PD <= p.kc1; kc += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      f32x4 bx_next2[RT];
#pragma unroll
      for (int h = 0; h < RT; ++h) bx_next2[h] = lds4(p.xrow + h * half + min(kc + s + 2, last) * 16);
      mfma_chunk_rt<NB, RT>(acc, r.slot[s], bx);
      if constexpr (PINNED) {           // see gemm_run: the refill stays where it is written
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
#pragma unroll
      for (int h = 0; h < RT; ++h) { bx[h] = bx_next[h]; bx_next[h] = bx_next2[h]; }
    }
  }
#pragma unroll
  for (int s = 0; s < PD - 1; ++s) {
    if (kc + s < p.kc1) {
      f32x4 bx_next2[RT];
#pragma unroll
      for (int h = 0; h < RT; ++h) bx_next2[h] = lds4(p.xrow + h * half + min(kc + s + 2, last) * 16);
      mfma_chunk_rt<NB, RT>(acc, r.slot[s], bx);
#pragma unroll
      for (int h = 0; h < RT; ++h) { bx[h] = bx_next[h]; bx_next[h] = bx_next2[h]; }
    }
  }
}

template <int NB, int NW, int RT, bool PINNED, class EPI>
__device__ __forceinline__ void k2_direct(const float* __restrict__ W1, const float* __restrict__ W2, const StageDesc& sd,
                                          float* lds, int blk0, int lane, const Pre& pre, bool use_pre, const EPI& epi,
                                          const UnetDesc& u) {
  const int row = lane & 15, g = lane >> 4;
  const bool has2 = sd.has2 != 0;
  const GemmPlan<NB> p1 = make_plan<NB>(W1, sd.L1, blk0, NW, lds + sd.x1, sd.s1, lane, 0, sd.L1.in_pad >> 4);
  const GemmPlan<NB> p2 = make_plan<NB>(W2, sd.L2, blk0, NW, lds + sd.x2, sd.s2, lane, 0, sd.L2.in_pad >> 4);
  f32x4 acc[RT][NB];
#pragma unroll
  for (int h = 0; h < RT; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[h][j] = epi.init((blk0 + j * NW) * 16 + 4 * g);
  Ring<NB> r1, r2;
  if (use_pre) ring_fill<NB, true>(r1, p1, pre); else ring_fill<NB, false>(r1, p1, pre);
  if (has2) ring_fill<NB, false>(r2, p2, pre);     // the residual GEMM's first chunks fly while GEMM 1 runs
  k2_gemm_run<NB, RT, PINNED>(acc, r1, p1, 16 * sd.s1);
#pragma unroll
  for (int h = 0; h < RT; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j) epi.mid(acc[h][j], h * 16 + row, (blk0 + j * NW) * 16 + 4 * g);
  if (has2) k2_gemm_run<NB, RT, PINNED>(acc, r2, p2, 16 * sd.s2);
#pragma unroll
  for (int h = 0; h < RT; ++h)
#pragma unroll
    for (int j = 0; j < NB; ++j) epi.fin(acc[h][j], h * 16 + row, (blk0 + j * NW) * 16 + 4 * g, u, h, blk0 + j * NW);
}

// One stage on the 16-row tile; same work split as socmx_unet.h's unet_stage (direct: neuron blocks dealt to the waves;
// fewer than four blocks: the reduction dimension is split over the waves and combined through LDS), generalised
// epilogue.  Ends with a workgroup barrier.
template <int NW, int RT, bool PINNED, class EPI>
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
      if (cnt >= 4)      k2_direct<4, NW, RT, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      else if (cnt == 3) k2_direct<3, NW, RT, PINNED>(W1, W2, sd, lds, blk0, lane, pre, false, epi, u);
      else if (cnt == 2) k2_direct<2, NW, RT, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      else               k2_direct<1, NW, RT, PINNED>(W1, W2, sd, lds, blk0, lane, pre, use_pre, epi, u);
      use_pre = false;
    }
    pre = prefetch_fragments(Wn, sd.Ln, w, lane);
    __syncthreads();
  } else {
    const int parts = w.parts, blk = w.blk0, part = w.part;
    const int outp = sd.L1.out_pad;
    const int row = lane & 15, g = lane >> 4;
    constexpr int ROWS = 16 * RT;
    float* P1 = scratch;
    float* P2 = scratch + parts * ROWS * outp;
    if (w.active) {
      const GemmPlan<1> p1 = make_plan<1>(W1, sd.L1, blk, 0, lds + sd.x1, sd.s1, lane, w.kc0a, w.kc1a);
      const GemmPlan<1> p2 = make_plan<1>(W2, sd.L2, blk, 0, lds + sd.x2, sd.s2, lane, w.kc0b, w.kc1b);
      Ring<1> r1, r2;
      const bool w1 = p1.kc1 > p1.kc0, w2 = has2 && p2.kc1 > p2.kc0;
      if (w1) ring_fill<1, true>(r1, p1, pre);
      if (w2) ring_fill<1, false>(r2, p2, pre);
      f32x4 acc[RT][1];
#pragma unroll
      for (int h = 0; h < RT; ++h) acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (w1) k2_gemm_run<1, RT, false>(acc, r1, p1, 16 * sd.s1);
      if constexpr (EPI::kFuse2) {
        if (w2) k2_gemm_run<1, RT, false>(acc, r2, p2, 16 * sd.s2);
      }
#pragma unroll
      for (int h = 0; h < RT; ++h)
        *reinterpret_cast<f32x4*>(P1 + (part * ROWS + h * 16 + row) * outp + blk * 16 + 4 * g) = acc[h][0];
      if (has2 && !EPI::kFuse2) {
        f32x4 acc2[RT][1];
#pragma unroll
        for (int h = 0; h < RT; ++h) acc2[h][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (w2) k2_gemm_run<1, RT, false>(acc2, r2, p2, 16 * sd.s2);
#pragma unroll
        for (int h = 0; h < RT; ++h)
          *reinterpret_cast<f32x4*>(P2 + (part * ROWS + h * 16 + row) * outp + blk * 16 + 4 * g) = acc2[h][0];
      }
    }
    pre = prefetch_fragments(Wn, sd.Ln, w, lane);
    __syncthreads();
    // combine: thread e owns the quad (row e / Q, units 4 (e % Q) ..), Q = outp / 4 <= 12
    const int Q = outp >> 2;
    const int e = threadIdx.x;
    if (e < ROWS * Q) {                                    // (Q <= 12, ROWS <= 32: at most 384 of the 512 threads)
      const int r = (int)(((float)e + 0.5f) * __builtin_amdgcn_rcpf((float)Q)), n0 = 4 * (e - r * Q);
      f32x4 v = epi.init(n0);
      for (int p = 0; p < parts; ++p) v += lds4(P1 + (p * ROWS + r) * outp + n0);
      epi.mid(v, r, n0);
      if (has2 && !EPI::kFuse2)
        for (int p = 0; p < parts; ++p) v += lds4(P2 + (p * ROWS + r) * outp + n0);
      epi.fin(v, r, n0, u);
    }
    __syncthreads();
  }
}

template <int NW, class NET, int RT, int SI, bool SAVED = false>
__device__ __forceinline__ void k2_run_stage(const TileArgs& a, float* lds, Pre& carry, const EpiCtx& ctx, int wave) {
  constexpr bool kStatic = !std::is_same<NET, void>::value;
  auto body = [&](const K2Stage& s, const WaveWorkS& w, const UnetDesc& u, const BwdLayout& lay) {
    const float* W1 = s.img1 ? a.packedT : a.packed;
    const float* W2 = s.img2 ? a.packedT : a.packed;
    const float* Wn = s.imgn ? a.packedT : a.packed;
    float* scratch = lds + lay.t.scratch;
    if (s.epi == EPI_RELU)       k2_stage<NW, RT, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_RELU>{s, ctx}, u);
    else if (s.epi == EPI_RES)   k2_stage<NW, RT, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_RES>{s, ctx}, u);
    else if (s.epi == EPI_MASK0) k2_stage<NW, RT, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_MASK0>{s, ctx}, u);
    else if (s.epi == EPI_DUAL)  k2_stage<NW, RT, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_DUAL, SAVED>{s, ctx}, u);
    else                         k2_stage<NW, RT, kStatic>(W1, W2, Wn, s.sd, w, lds, scratch, carry, Epi<EPI_ACTMASK, SAVED>{s, ctx}, u);
  };
  if constexpr (kStatic) {
    // every descriptor is a compile-time value: offsets fold into immediates, one NB variant and one epilogue survive
    constexpr UnetDesc u = NET::desc();
    constexpr BwdDesc bd = make_bwd_desc(u);
    constexpr BwdLayout lay = make_bwd_layout(u, NW, RT, SAVED);
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
// constexpr instantiations: two 16-row tiles per workgroup; the descriptor-driven one: one (smaller LDS: more
// architectures fit; it is the fallback for non-default widths)
// Kernel A's constexpr instantiations: FOUR waves (one per SIMD) on ONE 16-row tile, two workgroups per CU (77 KiB of LDS, 170
// VGPRs each).  Round 2 ran eight waves on two tiles, one workgroup per CU -- every weight fragment fed two column tiles, but
// nothing covered a workgroup's eleven barriers and ring restarts; two independent four-wave workgroups do (their stalls
// interleave), at twice the fragment traffic from L2 (32 B/clk/CU: still half of what the L1 sustains): 260 -> 220 us at
// cfg3, 1.88 -> 1.73 ms at the configs[4] slice.  (Two EIGHT-wave workgroups on one tile each spilled at 128 VGPRs.)
#ifndef SOCMX_K2A_WAVES
#define SOCMX_K2A_WAVES 4        /* waves per workgroup ... */
#define SOCMX_K2A_RT 1           /* ... and 16-row tiles per workgroup */
#endif
template <class NET> struct K2RowTiles { static constexpr int value = SOCMX_K2A_RT; };
template <> struct K2RowTiles<void> { static constexpr int value = 1; };

template <int NW, class NET, bool SAVED = false> struct K2Const {
  static constexpr UnetDesc u = NET::desc();
  static constexpr BwdLayout lay = make_bwd_layout(NET::desc(), NW, K2RowTiles<NET>::value, SAVED);
  __device__ static const UnetDesc& desc(const UnetDesc&) { return u; }
  __device__ static const BwdLayout& layout(const BwdLayout&) { return lay; }
};
template <int NW> struct K2Const<NW, void, false> {
  __device__ static const UnetDesc& desc(const UnetDesc& x) { return x; }
  __device__ static const BwdLayout& layout(const BwdLayout& x) { return x; }
};

// SAVED (socmx_unet_backward_saved_f32; constexpr instantiations, one tile per workgroup): the rollout that produced the rows wrote
// the activation slabs and the rows' sign records -- no forward stages: ZU0 = G (.) [output pre-activation > 0] from the record, then
// stages 6 .. 10 with their masks looked up in the sixteen records of the tile.
template <int NW, class NET, bool SAVED = false>
__global__ __launch_bounds__(NW * 64, SAVED ? SOCMX_K2S_WGS : 2) void unet_bwd_tile_kernel(const TileArgs a) {   // (two waves per SIMD is what the LDS admits: no AGPR copies to stay under 128 VGPRs)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool kStatic = !std::is_same<NET, void>::value;
  // constexpr instantiations: the descriptors are constants in the code object (a local copy whose address is handed on
  // lives in scratch memory); the descriptor-driven one copies them from the kernel arguments
  UnetDesc u_arg;
  BwdLayout lay_arg;
  if constexpr (!kStatic) { u_arg = a.u; lay_arg = a.lay; }
  const UnetDesc& u = K2Const<NW, NET, SAVED>::desc(u_arg);
  const BwdLayout& lay = K2Const<NW, NET, SAVED>::layout(lay_arg);
  const TileLayout& t = lay.t;
  const int tid = threadIdx.x, nthr = NW * 64;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int d = a.u.d, in0p = u.in0p, outp = u.outp;
  constexpr int RT = K2RowTiles<NET>::value, ROWS = 16 * RT;
  const int tile = blockIdx.x * RT;                       // first 16-row tile of this workgroup
  const int64_t row0 = (int64_t)tile * 16;
  const int64_t tile_rows = (int64_t)a.ntiles * 16;
  float* X0 = lds + t.x0;
  float* G0 = lds + t.gv;
  // ---- input tile [t, x, 0..] and gradient tile (zero rows past N: they then contribute nothing to any gradient) ------
  float* slabX = a.ws + (size_t)tile_rows * tensor_prefix(u, T_X) + (size_t)tile * in0p * 16;
  float* slabG = a.ws + (size_t)tile_rows * tensor_prefix(u, T_G0) + (size_t)tile * outp * 16;
  for (int e = tid; e < ROWS * in0p; e += nthr) {
    // e = ((h in0p + c) 16 + r16): unit-major inside each 16-row tile like the slab -> coalesced slab stores
    const int r16 = e & 15, hc = e >> 4;
    const int h = hc >= in0p ? 1 : 0, c = hc - h * in0p, r = h * 16 + r16;
    const int64_t grow = min(row0 + r, a.N - 1);
    float v = 0.f;
    if (c == 0) v = a.ts[grow / a.rows_per_t];
    else if (c <= d) v = a.x[grow * d + c - 1];
    if constexpr (!SAVED) X0[r * t.s0 + c] = v;     // (SAVED: no forward stage reads it, and ZU0 -- formed below -- lives in its place)
    slabX[e] = v;
  }
  const float gsc = a.gscale ? a.gscale[0] : 1.f;
  float* slabZ = a.ws + (size_t)tile_rows * tensor_prefix(u, T_ZU0) + (size_t)tile * outp * 16;
  uint32_t* recl = reinterpret_cast<uint32_t*>(lds + lay.mu2);      // SAVED: the tile's records (the mask arrays' place: 512 of >= 832 dwords)
  for (int e = tid; e < ROWS * outp; e += nthr) {
    const int r16 = e & 15, hc = e >> 4;
    const int h = hc >= outp ? 1 : 0, c = hc - h * outp, r = h * 16 + r16;
    const float v = (row0 + r < a.N && c < d) ? a.gout[(row0 + r) * d + c] * gsc : 0.f;
    G0[r * t.sg + c] = v;
    slabG[e] = v;
    if constexpr (SAVED) {     // stage 5's result without its GEMMs: ZU0 = G (.) [up_0 A1 + F R1 + f + b > 0] (wave 0's record, dword 2)
      const float z = ((a.rec[(size_t)(row0 + r) * kActRecordDwords + 2] >> c) & 1u) ? v : 0.f;
      lds[lay.zu0 + r * t.sg + c] = z;
      slabZ[e] = z;
    }
  }
  if constexpr (SAVED) {
    static_assert(RT == 1, "SAVED: one 16-row tile per workgroup");
    for (int e = tid; e < ROWS * kActRecordDwords; e += nthr) recl[e] = a.rec[(size_t)row0 * kActRecordDwords + e];
  } else {
    unet_load_biases(a.packed, u, t, lds, tid, nthr);
  }
  EpiCtx ctx{lds, a.ws, lds + t.bias, tile_rows, tile, a.packed + u.fold.b_off, recl};
  // first ring of stage 0 (GEMM 1 = down_0 of the forward image)
  Pre carry;
  {
    // (SAVED: the first stage that runs is 6, GEMM 1 = up_0^T of the transposed image)
    LayerDesc L0; int img0 = 0;
    if constexpr (SAVED) { constexpr BwdDesc bc = make_bwd_desc(NET::desc()); L0 = bc.L[KT_U0]; }
    else if constexpr (kStatic) { constexpr UnetDesc uc = NET::desc(); L0 = uc.L[0]; } else { L0 = a.u.L[0]; }
    (void)img0;
    unsigned short pf[8] = {};
    first_fragment_numbers(L0, NW, wave, pf);
    WaveWorkS w{};
#pragma unroll
    for (int f = 0; f < 8; ++f) w.pf[f] = pf[f];
    carry = prefetch_fragments(SAVED ? a.packedT : a.packed, L0, w, lane);
  }
  __syncthreads();
  if constexpr (!SAVED) {
    k2_run_stage<NW, NET, RT, 0>(a, lds, carry, ctx, wave);
    k2_run_stage<NW, NET, RT, 1>(a, lds, carry, ctx, wave);
    k2_run_stage<NW, NET, RT, 2>(a, lds, carry, ctx, wave);
    k2_run_stage<NW, NET, RT, 3>(a, lds, carry, ctx, wave);
    k2_run_stage<NW, NET, RT, 4>(a, lds, carry, ctx, wave);
    k2_run_stage<NW, NET, RT, 5>(a, lds, carry, ctx, wave);
  }
  k2_run_stage<NW, NET, RT, 6, SAVED>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, RT, 7, SAVED>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, RT, 8, SAVED>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, RT, 9, SAVED>(a, lds, carry, ctx, wave);
  k2_run_stage<NW, NET, RT, 10, SAVED>(a, lds, carry, ctx, wave);
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

// Cost-balanced placement of kernel B's waves (control network).  A wave is one (block group, slab of row tiles); the chip
// holds 3 waves per SIMD of this kernel (144 VGPRs: scalar bases + immediate block offsets, two tiles in flight for the
// (4, 4) groups), i.e. 384 per XCD, and a uniform "every group x S slabs" grid neither
// fills a whole number of such rounds nor gives a (4, 4) group (64 MFMAs per tile) more waves than a (2, 2) one (16).
// Instead every XCD x owns the x-th eighth of the row tiles (its L2 then holds what the groups re-read) and runs
// `nslots` waves on it: group i gets n[i] of them, n[i] proportional to its cost per tile (largest-remainder by a
// greedy pass on the host, wgrad_make_schedule), so that ONE full round of equal-length waves covers the work.
// Group i is therefore cut into S_i = 8 n[i] slabs, and kernel C adds S_i partials for its cells.
constexpr int kWgSlots = 384;            // wave slots per XCD: one full round of the chip at three waves per SIMD
constexpr int kWgMaxItems = 255;         // block groups the table can describe (more: the uniform grid)
struct WgSched {
  int nslots;                            // 0: uniform grid (S slabs for every group); else kWgSlots
  uint8_t item[kWgSlots];                // wave slot -> block group (255: idle)
  uint8_t k[kWgSlots];                   // ... -> which of the group's n slabs inside the XCD's eighth
  uint8_t n[kWgMaxItems + 1];            // per block group: slabs per XCD
};

struct WgradArgs {
  // per forward layer: tensors, block counts, first item; block groups (<= 4 x 4 blocks) are numbered ig-fastest
  int gt_off[9], at_off[9], gW[9], aW[9], OB[9], IB[9], w_cell_off[9], b_part_off[9], item0[10];
  int n_items;
  int S;                   // slabs (uniform grid; with a schedule: the largest S_i, which sizes `part`)
  int ntiles;
  int64_t slab_floats;     // floats per slab of partials
  const float* ws;
  float* part;             // (S, slab_floats)
  int bias_even_tiles;     // 1: only even 16-row tiles count towards the bias gradients (pair network: the odd tiles hold
                           //    the forward TANGENT rows, which pass through the weights but not the biases)
  WgSched sched;
};

#ifndef SOCMX_K2B_PD44
#define SOCMX_K2B_PD44 2     // tiles in flight per wave of a (4, 4) block group (three waves per SIMD)
#endif
// N asm loads of consecutive 1-KiB blocks: scalar base, 32-bit lane offset, the block as an immediate offset
template <int J, int N>
struct WgLoad {
  __device__ static __forceinline__ void run(f32x4 (&dst)[N], unsigned voff, const float* base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst[J]) : "v"(voff), "s"(base), "n"(J * 1024) : "memory");
    if constexpr (J + 1 < N) WgLoad<J + 1, N>::run(dst, voff, base);
  }
};

template <int NOB, int NIB>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const WgItem& it, int slab, int S, int lane) {
  const int c = lane & 15, g = lane >> 4;
  const int t0 = (int)(((int64_t)slab * a.ntiles) / S), t1 = (int)(((int64_t)(slab + 1) * a.ntiles) / S);
  const int64_t tile_rows = (int64_t)a.ntiles * 16;
  // lane (c, g) reads rows 4g .. 4g+3 of unit (block * 16 + c): one 16-byte load; MFMA number s of a tile takes component s
  // from every lane, i.e. k-slot g carries row 4g + s -- the same permutation for both operands.
  // Addresses: a wave-uniform base per operand and tile (SGPR pair) + one 32-bit lane offset + the block as an IMMEDIATE
  // (blocks are 1 KiB apart): two address VGPRs instead of a 64-bit pointer per load -- 173 -> 16x VGPRs, i.e. three waves
  // per SIMD.  Blocks past the group (instantiations wider than the group) read the neighbouring blocks of the workspace --
  // in bounds: the partials follow the tiles in the same allocation -- and their products are never stored.
  const float* gbase = a.ws + (size_t)tile_rows * it.gt_off + (size_t)it.ob0 * 256;
  const float* abase = a.ws + (size_t)tile_rows * it.at_off + (size_t)it.ib0 * 256;
  const unsigned lane_off = (unsigned)(c * 16 + 4 * g) * 4u;
  const size_t gstep = (size_t)it.gW * 16, astep = (size_t)it.aW * 16;
  f32x4 acc[NOB][NIB];
  float bsum[NOB];
#pragma unroll
  for (int j = 0; j < NOB; ++j) {
    bsum[j] = 0.f;
#pragma unroll
    for (int k = 0; k < NIB; ++k) acc[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // PD (three or more) tiles of operands in flight.  The loads are asm statements and the waits are written out: with `if (t < t1) load`
  // in the loop the compiler's wait-count insertion falls back to (nearly) vmcnt(0) in front of every tile's MFMAs, i.e. one
  // tile in flight (the socm_target_lds4_kernel story, socmx_loss.hip).  Requests past the slab's last tile re-read it, so
  // that "PD - 1 younger tiles may stay in flight" is the same number on every trip.
  // (narrow block groups have few MFMAs per tile to hide the load latency behind: more tiles in flight for them)
  constexpr int PD = NOB * NIB >= 16 ? SOCMX_K2B_PD44 : NOB * NIB >= 8 ? 4 : NOB * NIB >= 4 ? 6 : 8, NL = NOB + NIB;
  f32x4 ga[PD][NOB], ab[PD][NIB];
  auto load = [&](int t, int s) {
    const int tc = min(t, t1 - 1);
    WgLoad<0, NOB>::run(ga[s], lane_off, gbase + (size_t)tc * gstep);
    WgLoad<0, NIB>::run(ab[s], lane_off, abase + (size_t)tc * astep);
  };
  auto wait_slot = [&](int s) {                 // the PD - 1 younger tiles may stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NL) : "memory");
#pragma unroll
    for (int j = 0; j < NOB; ++j) asm volatile("" : "+v"(ga[s][j]));
#pragma unroll
    for (int k = 0; k < NIB; ++k) asm volatile("" : "+v"(ab[s][k]));
  };
  if (t0 < t1) {
#pragma unroll
    for (int s = 0; s < PD; ++s) load(t0 + s, s);
    for (int t = t0; t < t1; t += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
        wait_slot(s);
        if (t + s < t1) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NOB; ++j)
#pragma unroll
              for (int k = 0; k < NIB; ++k)
                acc[j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s][j][i], ab[s][k][i], acc[j][k], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < NOB; ++j)
            if (!a.bias_even_tiles || !((t + s) & 1)) bsum[j] += (ga[s][j][0] + ga[s][j][1]) + (ga[s][j][2] + ga[s][j][3]);
        }
        load(t + s + PD, s);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

__global__ __launch_bounds__(256, 3) void unet_wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with its own L2): all block groups
  // that read one slab of row tiles are numbered onto ONE XCD, so a tile's operands cross the fabric once and the
  // other groups' re-reads hit that L2.  (Speed only: any placement is correct.)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  int item, slab, S;
  if (a.sched.nslots) {
    const int ws = q * 4 + wave;
    item = a.sched.item[ws];
    if (item == 255) return;
    const int n = a.sched.n[item];
    S = 8 * n;
    slab = xcd * n + a.sched.k[ws];
  } else {
    item = q % a.n_items;
    slab = ((q / a.n_items) * 8 + xcd) * 4 + wave;
    S = a.S;
    if (slab >= S) return;
  }
  int l = 8;
  while (l > 0 && item < a.item0[l]) --l;
  WgItem it;
  {
    const int rel = item - a.item0[l];
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
    case 0: wgrad_body<4, 4>(a, it, slab, S, lane); break;
    case 1: wgrad_body<4, 2>(a, it, slab, S, lane); break;
    case 2: wgrad_body<2, 4>(a, it, slab, S, lane); break;
    case 3: wgrad_body<2, 2>(a, it, slab, S, lane); break;
    case 4: wgrad_body<4, 1>(a, it, slab, S, lane); break;
    case 5: wgrad_body<1, 4>(a, it, slab, S, lane); break;
    default: wgrad_body<1, 1>(a, it, slab, S, lane); break;
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
  int scheduled;           // 1: block group i of kernel B's schedule wrote 8 n[i] partials (else S for every cell)
  int item0[10];
  uint8_t n[kWgMaxItems + 1];
  float* fold_g;           // control network: layer 4's slot holds G' = ZU0^T R1 (fout[4] x fin[4]) and s = sum ZU0 -- they go
                           // here ([fout][fin], then fout sums) for unet_unfold_kernel, not into `grads`; null: no such layer
  int fold_fout;           // ... G' rows (= d); fout[4] stays res_1's fan-out for nobody: the finish kernel reads this one
};

__device__ __forceinline__ int finish_partials(const FinishArgs& a, int l, int ob, int ib) {
  if (!a.scheduled) return a.S;
  return 8 * a.n[a.item0[l] + (ob >> 2) * ((a.IB[l] + 3) >> 2) + (ib >> 2)];
}

// 64 x 4 threads: thread (x, y) adds quarter y of the partials of four consecutive floats (one lane's 16 bytes of a cell, or four
// bias units) in four independent chains; the quarters meet in LDS and are added in a fixed order.  (One thread per float
// over all S partials was a chain of S / 4 dependent round trips: 20 us for 38 MB.)
__global__ __launch_bounds__(256) void unet_wgrad_finish_kernel(const FinishArgs a) {
  __shared__ f32x4 quarter[4][64];
  const int xl = threadIdx.x & 63, y = threadIdx.x >> 6;
  const int idx = (blockIdx.x * 64 + xl) * 4;
  const int total = a.total_cells_floats + a.total_bias;           // both multiples of 16
  int l = 8, S = 0, o = 0, i = 0;
  bool cells = false;
  if (idx < a.total_cells_floats) {
    cells = true;
    while (l > 0 && idx < a.w_cell_off[l]) --l;
    const int rel = idx - a.w_cell_off[l];
    const int cell = rel >> 8, lane = (rel >> 2) & 63;
    const int ob = cell / a.IB[l], ib = cell - ob * a.IB[l];
    o = ob * 16 + 4 * (lane >> 4); i = ib * 16 + (lane & 15);
    S = finish_partials(a, l, ob, ib);
  } else if (idx < total) {
    while (l > 0 && idx < a.b_part_off[l]) --l;
    o = idx - a.b_part_off[l];
    S = finish_partials(a, l, o >> 4, 0);
  }
  f32x4 s0{0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  {
    const int p1 = ((y + 1) * S) >> 2;
    int p = (y * S) >> 2;
    const float* src = a.part + idx;
    for (; p + 3 < p1; p += 4) {
      s0 += *reinterpret_cast<const f32x4*>(src + (size_t)p * a.slab_floats);
      s1 += *reinterpret_cast<const f32x4*>(src + (size_t)(p + 1) * a.slab_floats);
      s2 += *reinterpret_cast<const f32x4*>(src + (size_t)(p + 2) * a.slab_floats);
      s3 += *reinterpret_cast<const f32x4*>(src + (size_t)(p + 3) * a.slab_floats);
    }
    for (; p < p1; ++p) s0 += *reinterpret_cast<const f32x4*>(src + (size_t)p * a.slab_floats);
  }
  quarter[y][xl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (y != 0 || S == 0) return;
  const f32x4 v = (quarter[0][xl] + quarter[1][xl]) + (quarter[2][xl] + quarter[3][xl]);
  const bool folded = a.fold_g && l == 4;
  const int fo = folded ? a.fold_fout : a.fout[l];
  float* gw = folded ? a.fold_g : a.grads + a.gw_off[l];
  float* gb = folded ? a.fold_g + (size_t)a.fold_fout * a.fin[l] : a.grads + a.gb_off[l];
  if (cells) {
    if (i < a.fin[l])
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        if (o + rr < fo) gw[(int64_t)(o + rr) * a.fin[l] + i] = v[rr];
  } else {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (o + rr < fo) gb[o + rr] = v[rr];
  }
}

// ---- kernel D: the fold undone on the gradients --------------------------------------------------------------------------
// In: G' = ZU0^T R1 (d x h0) and s = sum_rows ZU0 (d) from kernel C (fold_g); u0 (d x h0), r1 (h0 x h0), b4 from the forward image.
//   dW_r1[m][k]  = sum_n u0[n][m] G'[n][k]              (= GO1^T R1 with GO1 = ZU0 u0)
//   db_r1[m]     = sum_n u0[n][m] s[n]
//   dW_u0[n][m] += sum_k G'[n][k] r1[m][k] + s[n] b4[m]  (kernel C wrote ZU0^T A1 there: the skip's share of ZU0^T o1 is added)
// Latency layout like unet_fold_kernel: block (y = 0: 64 columns k of dW_r1 x 4 rows m per thread group | y = 1: dW_u0).
struct UnfoldArgs {
  UnetDesc u;
  int d, h0;
  const float* packed;     // forward image
  const float* fold_g;     // [d][h0] G', then d sums
  float* grads;
  int64_t gw4, gb4, gw8;   // offsets of dW_r1, db_r1, dW_u0 in `grads`
};
// element [n][k] of layer L of the forward image (socmx_unet.h fragment order)
__device__ __forceinline__ float image_w(const float* packed, const LayerDesc& L, int n, int k) {
  const int KC = L.in_pad >> 4;
  return packed[L.w_off + (((n >> 4) * KC + (k >> 4)) * 64 + (n & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3)];
}
__global__ __launch_bounds__(256) void unet_unfold_kernel(const UnfoldArgs a) {
  __shared__ float red[16][17];
  const int d = a.d, h0 = a.h0;
  const float* G = a.fold_g;
  const float* sv = a.fold_g + (size_t)d * h0;
  const int nb1 = (h0 + 3) / 4 * ((h0 + 63) / 64);          // dW_r1: blocks of (4 rows m, 64 columns k)
  if ((int)blockIdx.x < nb1) {
    const int kb = blockIdx.x % ((h0 + 63) / 64), mb = blockIdx.x / ((h0 + 63) / 64);
    const int k = kb * 64 + (threadIdx.x & 63), m = mb * 4 + (threadIdx.x >> 6);
    if (k < h0 && m < h0) {
      // (all 2 d operands requested before the first multiply-add: a rolled loop ran d dependent round trips -- 10 us at d = 10)
      float acc = 0.f;
      for (int n0 = 0; n0 < d; n0 += 8) {
        float wv[8], gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = min(n0 + j, d - 1);
          wv[j] = image_w(a.packed, a.u.L[8], n, m);
          gv[j] = G[(size_t)n * h0 + k];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf(n0 + j < d ? wv[j] : 0.f, gv[j], acc);
      }
      a.grads[a.gw4 + (int64_t)m * h0 + k] = acc;
    }
    if (kb == 0 && (threadIdx.x & 63) == 0 && m < h0) {
      float acc = 0.f;
      for (int n = 0; n < d; ++n) acc = fmaf(image_w(a.packed, a.u.L[8], n, m), sv[n], acc);
      a.grads[a.gb4 + m] = acc;
    }
    return;
  }
  // dW_u0: block = (n, 16 units m), thread (m = tid & 15, part = tid >> 4) adds the terms k = part, part + 16, ...
  const int b = blockIdx.x - nb1, mblocks = (h0 + 15) / 16;
  const int n = b / mblocks, m = (b % mblocks) * 16 + (threadIdx.x & 15), part = threadIdx.x >> 4;
  float acc = 0.f;
  if (m < h0) {
    for (int k0 = part; k0 < h0; k0 += 16 * 8) {          // (eight terms' operands in flight at a time)
      float gv[8], wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = min(k0 + 16 * j, h0 - 1);
        gv[j] = G[(size_t)n * h0 + k];
        wv[j] = image_w(a.packed, a.u.L[4], m, k);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(k0 + 16 * j < h0 ? gv[j] : 0.f, wv[j], acc);
    }
  }
  red[part][threadIdx.x & 15] = acc;
  __syncthreads();
  if (part == 0 && m < h0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x & 15];
    t = fmaf(sv[n], a.packed[a.u.L[4].b_off + m], t);
    a.grads[a.gw8 + (int64_t)n * h0 + m] += t;
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
  if (t == KT_R1) return;                 // F^T: unet_fold_launch(transposed) writes this slot
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

// =====================================================================================================================
// K3: the pair-grid network M (SigmoidMLP, reference models.py:245-275) and its s-tangent (method.py:510-515 obtains
// d/ds by functorch.jacrev) as tile kernels.  net = L2 relu(L1 relu(L0 [t,s] + b0) + b1) + b2 on the Np pairs, and the
// forward tangent  dnet = L2 (m2 (.) L1 (m1 (.) L0 e_s))  shares every weight fragment with it: a workgroup holds 16 pairs
// as 32 tile rows -- rows 0..15 the values, rows 16..31 the tangents -- so each fragment from L2 feeds two MFMA column
// tiles (the RT = 2 machinery of kernel A), and a lane that holds a value quad also holds the tangent quad of the same
// (pair, units): the ReLU sign is applied to both in registers.
//   forward kernel : F1, F2, F3 -> net, dnet (Np, d*d) row-major (what the contraction kernels consume)
//   backward kernel: F1, F2 recomputed, then  (g_h2, g_t2) = L2^T (g_net, g_dnet),  (g_h1, g_t1) = L1^T (m2 (.) ...);
//                    activations and pre-activation gradients leave as slabs, value and tangent rows as CONSECUTIVE 16-row
//                    tiles, so kernel B / C above produce dW_l = sum over value AND tangent rows (biases: value tiles only).
// Supported while the widest tile (d*d + 4 floats per row, 32 rows) fits LDS: d <= 22 with 128-wide hidden layers.
struct MDesc {
  int d2, d2p, h0, h0p, h1, h1p;
  LayerDesc L[3];      // forward image: L0 (16 -> h0p), L1 (h0p -> h1p), L2 (h1p -> d2p), weights then bias per layer
  LayerDesc LT[2];     // transposed image: L2^T (d2p -> h1p), L1^T (h1p -> h0p)
  int total_floats, totalT_floats, bias_floats;
  // LDS (floats): tiles of 32 rows
  int sx, s1, s2, so;
  int x, hh1, hh2, gout, gz2, m1, m2, bias, scratch, lds_floats;
  // slab tensors [tile][unit][16]: X, H1, H2, GOUT, GZ2, GZ1 -- prefix = sum of widths before
  int pre[6], wid[6];
  // WIDE form (the (d*d)-wide tile does not fit LDS -- BASELINE configs[4]: d = 64, 4096 outputs): the last layer's outputs
  // leave the forward kernel straight from the accumulators, the backward kernel reads g_net / g_dnet rows from HBM as
  // the MFMA B operand (split-K over the waves) and the last layer's weight gradient has its own kernel
  // (mnet_wgrad_wide_kernel); no GOUT tile, no GOUT slab.  Needs h1p <= 256.  The kernels move 16-byte pieces of a row of
  // d*d floats; where d*d is not a multiple of 16 (d % 4 != 0) the row's last piece is read shifted back into the row and
  // stored element by element (rows start at any 4-byte address then: global accesses need no more).
  int wide;
  int wscratch;            // wide backward kernel: float offset of the split-K combine scratch (aliases the forward tiles)
  int lds_fwd_floats;      // LDS of the forward kernel (wide: without the backward kernel's tiles and combine scratch)
};
enum { MT_X = 0, MT_H1, MT_H2, MT_GOUT, MT_GZ2, MT_GZ1, MT_N };

inline MDesc make_mdesc(int d, int h0, int h1, int nwaves) {
  MDesc m{};
  m.d2 = d * d; m.d2p = pad16(d * d); m.h0 = h0; m.h0p = pad16(h0); m.h1 = h1; m.h1p = pad16(h1);
  const int fin[3] = {16, m.h0p, m.h1p}, fout[3] = {m.h0p, m.h1p, m.d2p};
  int off = 0, boff = 0;
  for (int l = 0; l < 3; ++l) {
    m.L[l].in_pad = fin[l]; m.L[l].out_pad = fout[l];
    m.L[l].w_off = off; off += fin[l] * fout[l];
    m.L[l].b_off = off; off += fout[l];
    m.L[l].b_lds = boff; boff += fout[l];
  }
  m.total_floats = off; m.bias_floats = boff;
  m.LT[0].in_pad = m.d2p; m.LT[0].out_pad = m.h1p; m.LT[0].w_off = 0;
  m.LT[1].in_pad = m.h1p; m.LT[1].out_pad = m.h0p; m.LT[1].w_off = m.d2p * m.h1p;
  m.totalT_floats = m.d2p * m.h1p + m.h1p * m.h0p;
  m.sx = 20; m.s1 = m.h0p + 4; m.s2 = m.h1p + 4; m.so = m.d2p + 4;
  int o = 0;
  m.x = o; o += 32 * m.sx;
  m.hh1 = o; o += 32 * m.s1;
  m.hh2 = o; o += 32 * m.s2;
  m.gout = o; o += 32 * m.so;          // backward only (the forward kernel's F3 writes straight to HBM)
  m.gz2 = o; o += 32 * m.s2;
  m.m1 = o; o += m.h0p;                // 16 rows x (width / 4) bytes
  m.m2 = o; o += m.h1p;
  m.bias = o; o += boff;
  // split-K scratch: stages whose GEMM has fewer than four 16-wide output blocks (widths < 64)
  int need = 0;
  const int outs[5] = {m.h0p, m.h1p, m.d2p, m.h1p, m.h0p};
  for (int i = 0; i < 5; ++i) {
    const int nblk = outs[i] >> 4;
    if (nblk >= kSimds || nblk >= nwaves) continue;
    const int n = (nwaves / nblk) * 32 * outs[i];
    need = n > need ? n : need;
  }
  m.scratch = o; o += need;
  m.lds_floats = o;
  m.lds_fwd_floats = m.lds_floats;
  m.wide = 0;
  if ((size_t)m.lds_floats * sizeof(float) > (size_t)160 * 1024 && m.h1p <= 256) {
    // wide layout: [x | hh1 | hh2] | gz2 | m1 | m2 | bias (L0, L1, then L2: forward only) | small scratch.  The backward
    // kernel's split-K combine of the wide stage ALIASES the bracketed forward tiles (dead by then): wscratch = 0.
    m.wide = 1;
    o = 0;
    m.x = o; o += 32 * m.sx;
    m.hh1 = o; o += 32 * m.s1;
    m.hh2 = o; o += 32 * m.s2;
    m.gout = -1;
    // one half of the combine: four waves' quads of half the output blocks, value and tangent
    const int nob_t = (m.h1p >> 4) <= 2 ? 2 : ((m.h1p >> 4) <= 4 ? 4 : 8);   // the kernel's NOB instantiation (129 .. 256 units: two passes of 8)
    const int combine_half = 4 * 2 * (nob_t / 2) * 256;
    m.wscratch = 0;
    if (combine_half > o) o = combine_half;
    m.gz2 = o; o += 32 * m.s2;
    m.m1 = o; o += m.h0p;
    m.m2 = o; o += m.h1p;
    m.bias = o;
    // small-stage split-K scratch as above, without the (d*d)-wide stage
    int need_small = 0;
    const int outs_w[4] = {m.h0p, m.h1p, m.h1p, m.h0p};
    for (int i = 0; i < 4; ++i) {
      const int nblk = outs_w[i] >> 4;
      if (nblk >= kSimds || nblk >= nwaves) continue;
      const int n = (nwaves / nblk) * 32 * outs_w[i];
      need_small = n > need_small ? n : need_small;
    }
    // forward: all three biases + small scratch; backward: two biases + small scratch
    m.lds_fwd_floats = m.bias + boff + need_small;
    m.scratch = m.bias + m.L[0].out_pad + m.L[1].out_pad;      // (backward; the forward kernel places it behind L2's bias)
    m.lds_floats = m.scratch + need_small;
  }
  const int wid[6] = {16, m.h0p, m.h1p, m.wide ? 0 : m.d2p, m.h1p, m.h0p};
  int pre = 0;
  for (int t = 0; t < 6; ++t) { m.wid[t] = wid[t]; m.pre[t] = pre; pre += wid[t]; }
  return m;
}

#if defined(__HIPCC__)
struct MArgs {
  MDesc m;
  const float* packed;     // forward image
  const float* packedT;    // transposed image (backward kernel)
  const float *t, *s;      // (Np,) pair times
  const float* z;          // (Np,) third network input (TwoBoundarySigmoidMLP's stopped / running flag), or nullptr
  int64_t Np;
  int ntiles;              // 16-pair tiles (= workgroups); slab tiles = 2 ntiles (value, tangent alternating)
  float *net, *dnet;       // forward kernel outputs (Np, d2)
  const float *gnet, *gdnet;   // backward kernel inputs (Np, d2)
  float* ws;               // backward kernel: slabs
};

enum { ME_RELU = 0, ME_OUT, ME_MASK };

struct MEpi {
  int kind;
  float* lds;
  const float* bias;       // LDS bias of the GEMM's layer (value rows), or nullptr
  int y, sy;               // output tile (float offset, stride) or -1
  int mask, mw;            // nibble-mask array (float offset) written (ME_RELU) / read (ME_MASK), its width in units
  float* slab; int sw;     // export slab base of this workgroup's VALUE tile for the output tensor (nullptr: none), width
  float *net, *dnet; int d2; int64_t p0, Np;   // ME_OUT
  __device__ __forceinline__ f32x4 init(int n0) const { return bias ? lds4(bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f}; }
  // value quad vq and tangent quad tq of pair-row r (0..15), units n0 .. n0+3
  __device__ __forceinline__ void fin(f32x4 vq, f32x4 tq, int r, int n0) const {
    if (kind == ME_OUT) {
      if (p0 + r < Np) {
        if ((d2 & 3) == 0) {           // rows are whole 16-byte pieces: one (buffer) store per quad
          if (n0 < d2) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int off = (r * d2 + n0) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vq), uniform_rsrc(net + (size_t)p0 * d2), off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tq), uniform_rsrc(dnet + (size_t)p0 * d2), off, 0, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (n0 + i < d2) { net[(size_t)(p0 + r) * d2 + n0 + i] = vq[i]; dnet[(size_t)(p0 + r) * d2 + n0 + i] = tq[i]; }
        }
      }
      return;
    }
    unsigned msk;
    unsigned char* mp = reinterpret_cast<unsigned char*>(lds + mask) + r * (mw >> 2) + (n0 >> 2);
    if (kind == ME_RELU) {
      msk = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { msk |= (vq[i] > 0.f ? 1u : 0u) << i; vq[i] = relu_keep_nan(vq[i]); }
      *mp = (unsigned char)msk;
    } else {
      msk = *mp;
#pragma unroll
      for (int i = 0; i < 4; ++i) vq[i] = ((msk >> i) & 1u) ? vq[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) tq[i] = ((msk >> i) & 1u) ? tq[i] : 0.f;
    if (y >= 0) {
      *reinterpret_cast<f32x4*>(lds + y + r * sy + n0) = vq;
      *reinterpret_cast<f32x4*>(lds + y + (16 + r) * sy + n0) = tq;
    }
    if (slab) {
      export4_u(slab, r, n0, vq);                       // value tile
      export4_u(slab + (size_t)sw * 16, r, n0, tq);     // the tangent tile follows it
    }
  }
};

// one GEMM stage on the 32-row (value | tangent) tile: Y = L . X, paired epilogue; work split as in socmx_unet.h
template <int NB, int NW>
__device__ __forceinline__ void m_direct(const float* __restrict__ Wp, const LayerDesc& L, const float* X, int S, int blk0,
                                         int lane, const MEpi& epi) {
  const int row = lane & 15, g = lane >> 4;
  const GemmPlan<NB> p = make_plan<NB>(Wp, L, blk0, NW, X, S, lane, 0, L.in_pad >> 4);
  f32x4 acc[2][NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) { acc[0][j] = epi.init((blk0 + j * NW) * 16 + 4 * g); acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  Ring<NB> r;
  Pre none;
  ring_fill<NB, false>(r, p, none);
  k2_gemm_run<NB, 2, false>(acc, r, p, 16 * S);
#pragma unroll
  for (int j = 0; j < NB; ++j) epi.fin(acc[0][j], acc[1][j], row, (blk0 + j * NW) * 16 + 4 * g);
}

template <int NW>
__device__ __forceinline__ void m_stage(const float* __restrict__ Wp, const LayerDesc& L, float* lds, int x, int S,
                                        float* scratch, const MEpi& epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NBLK = L.out_pad >> 4, KC = L.in_pad >> 4;
  if (NBLK >= kSimds || NBLK >= NW) {
    int cnt = wave < NBLK ? (NBLK - wave + NW - 1) / NW : 0;
    for (int blk0 = wave; cnt > 0; cnt -= 4, blk0 += 4 * NW) {
      if (cnt >= 4)      m_direct<4, NW>(Wp, L, lds + x, S, blk0, lane, epi);
      else if (cnt == 3) m_direct<3, NW>(Wp, L, lds + x, S, blk0, lane, epi);
      else if (cnt == 2) m_direct<2, NW>(Wp, L, lds + x, S, blk0, lane, epi);
      else               m_direct<1, NW>(Wp, L, lds + x, S, blk0, lane, epi);
    }
    __syncthreads();
    return;
  }
  // fewer than four output blocks: split K over the waves, combine through LDS
  const int parts = NW / NBLK, blk = wave % NBLK, part = wave / NBLK, outp = L.out_pad;
  const int row = lane & 15, g = lane >> 4;
  if (part < parts) {
    const int kc0 = (part * KC) / parts, kc1 = ((part + 1) * KC) / parts;
    f32x4 acc[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
    if (kc1 > kc0) {
      const GemmPlan<1> p = make_plan<1>(Wp, L, blk, 0, lds + x, S, lane, kc0, kc1);
      Ring<1> r;
      Pre none;
      ring_fill<1, false>(r, p, none);
      k2_gemm_run<1, 2, false>(acc, r, p, 16 * S);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      *reinterpret_cast<f32x4*>(scratch + (part * 32 + h * 16 + row) * outp + blk * 16 + 4 * g) = acc[h][0];
  }
  __syncthreads();
  const int Q = outp >> 2, e = threadIdx.x;
  if (e < 16 * Q) {
    const int r = (int)(((float)e + 0.5f) * __builtin_amdgcn_rcpf((float)Q)), n0 = 4 * (e - r * Q);
    f32x4 v = epi.init(n0), t = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < parts; ++p) {
      v += lds4(scratch + (p * 32 + r) * outp + n0);
      t += lds4(scratch + (p * 32 + 16 + r) * outp + n0);
    }
    epi.fin(v, t, r, n0);
  }
  __syncthreads();
}

// F1, F2 (both kernels): H1 = relu-pair(L0 X), H2 = relu-pair(L1 H1); `ws` non-null: export X, H1, H2 slabs
template <int NW>
__device__ __forceinline__ void m_forward_hidden(const MArgs& a, float* lds, int tile, int64_t tile_rows, int nbias = 3,
                                                 int scratch_off = -1) {
  const MDesc& m = a.m;
  const int tid = threadIdx.x, nthr = NW * 64;
  const int64_t p0 = (int64_t)tile * 16;
  float* X = lds + m.x;
  for (int e = tid; e < 32 * 16; e += nthr) {
    const int r = e >> 4, c = e & 15;
    float v = 0.f;
    if (r < 16) {
      const int64_t p = min(p0 + r, a.Np - 1);
      v = c == 0 ? a.t[p] : (c == 1 ? a.s[p] : ((c == 2 && a.z) ? a.z[p] : 0.f));
    } else {
      v = c == 1 ? 1.f : 0.f;                          // d [t, s] / d s
    }
    X[r * m.sx + c] = v;
  }
  for (int l = 0; l < nbias; ++l)
    for (int e = tid; e < m.L[l].out_pad; e += nthr) lds[m.bias + m.L[l].b_lds + e] = a.packed[m.L[l].b_off + e];
  __syncthreads();
  if (a.ws) {
    float* sl = a.ws + (size_t)tile_rows * m.pre[MT_X] + (size_t)(2 * tile) * 16 * 16;
    for (int e = tid; e < 2 * 16 * 16; e += nthr) {     // [h][unit][row16]
      const int r16 = e & 15, c = (e >> 4) & 15, h = e >> 8;
      sl[e] = X[(h * 16 + r16) * m.sx + c];
    }
  }
  float* scratch = lds + (scratch_off >= 0 ? scratch_off : m.scratch);
  MEpi e1{ME_RELU, lds, lds + m.bias + m.L[0].b_lds, m.hh1, m.s1, m.m1, m.h0p,
          a.ws ? a.ws + (size_t)tile_rows * m.pre[MT_H1] + (size_t)(2 * tile) * m.h0p * 16 : nullptr, m.h0p,
          nullptr, nullptr, 0, 0, 0};
  m_stage<NW>(a.packed, m.L[0], lds, m.x, m.sx, scratch, e1);
  MEpi e2{ME_RELU, lds, lds + m.bias + m.L[1].b_lds, m.hh2, m.s2, m.m2, m.h1p,
          a.ws ? a.ws + (size_t)tile_rows * m.pre[MT_H2] + (size_t)(2 * tile) * m.h1p * 16 : nullptr, m.h1p,
          nullptr, nullptr, 0, 0, 0};
  m_stage<NW>(a.packed, m.L[1], lds, m.hh1, m.s1, scratch, e2);
}

// F3 of the WIDE form: net / dnet = L2 (h2 | t2) for the 16 pairs of the tile, d2p / 16 output blocks of 16 units.  Wave w
// owns blocks w, w + NW, ... in groups of four; a group is eight chunks (h1p = 128; fewer for narrower layers) of four weight
// fragments and 32 MFMAs (4 blocks x value / tangent tile x 4 k-steps).  The fragments stream from L2 through a ring of
// three chunks whose requests are asm statements with written-out waits and run on ACROSS the group boundaries (the generic
// stage restarts its ring for every group of blocks: a full L2 round trip per 256 MFMAs); the activation quads come from
// LDS; every accumulator quad leaves with one 16-byte store.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a quad at any 4-byte address

template <int NW>
__device__ __forceinline__ void m_wide_out_stage(const MArgs& a, const float* lds, int tile) {
  // (h1p = 128, i.e. eight chunks per block -- the caller checks: narrower last layers take the generic stage)
  constexpr int KC = 8, PD = 4;
  const MDesc& m = a.m;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, g = lane >> 4;
  const LayerDesc& L = m.L[2];
  const int NBLK = L.out_pad >> 4;
  const int cnt = wave < NBLK ? (NBLK - wave + NW - 1) / NW : 0;       // blocks of this wave
  const int ngroups = (cnt + 3) >> 2;
  if (ngroups == 0) return;
  const float* xv = lds + m.hh2 + row * m.s2 + 4 * g;                   // value rows; tangent rows 16 s2 further
  const float* xt = xv + 16 * m.s2;
  const float* bias = lds + m.bias + L.b_lds + 4 * g;
  const int64_t p = (int64_t)tile * 16 + row;
  const bool rowok = p < a.Np;
  float* orow_n = a.net + (size_t)min(p, a.Np - 1) * m.d2 + 4 * g;
  float* orow_d = a.dnet + (size_t)min(p, a.Np - 1) * m.d2 + 4 * g;
  auto blk_of = [&](int grp, int j) { return min(wave + NW * (4 * grp + j), NBLK - 1); };
  // ring of four chunks (4 fragments each); chunk c of a group sits in slot c & 3.  Requests carry their chunk as an
  // immediate offset from a per-(group, block) base: no address arithmetic inside the group.
  f32x4 fr[PD][4];
  // (scalar base + 32-bit lane offset: eight offset registers instead of sixteen pointer registers keep the kernel at two
  //  workgroups per CU)
  const float* wbase = a.packed + L.w_off;
#define SOCMX_GLD(dst, voff, off) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #off : "=v"(dst) : "v"(voff), "s"(wbase) : "memory")
  unsigned cur[4];                                       // byte offsets of (block, chunk 0, this lane)
#pragma unroll
  for (int j = 0; j < 4; ++j) cur[j] = (unsigned)(blk_of(0, j) * KC) * 1024u + (unsigned)lane * 16u;
  // (chunk after chunk: the waits below count whole chunks)
#pragma unroll
  for (int j = 0; j < 4; ++j) SOCMX_GLD(fr[0][j], cur[j], 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) SOCMX_GLD(fr[1][j], cur[j], 1024);
#pragma unroll
  for (int j = 0; j < 4; ++j) SOCMX_GLD(fr[2][j], cur[j], 2048);
#pragma unroll
  for (int j = 0; j < 4; ++j) SOCMX_GLD(fr[3][j], cur[j], 3072);
  f32x4 acc[2][4];
  for (int grp = 0; grp < ngroups; ++grp) {
    const int gn = min(grp + 1, ngroups - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0][j] = lds4(bias + blk_of(grp, j) * 16);
      acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto chunk = [&](const int c) {                       // c = 0..7, constant after unrolling
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the three younger chunks (12 requests) may stay in flight
#pragma unroll
      for (int q = 0; q < PD; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fr[q][j]));
      const f32x4 bv = lds4(xv + c * 16), bt = lds4(xt + c * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w = fr[c & 3][j][i];
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bv[i], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bt[i], acc[1][j], 0, 0, 0);
        }
      // refill the slot: chunk c + 4 of this group (c < 4) or chunk c - 4 of the next one
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (c == 0) SOCMX_GLD(fr[0][j], cur[j] + 4096u, 0);           // (+ 4 KiB: immediate offsets stay below 4096)
        else if (c == 1) SOCMX_GLD(fr[1][j], cur[j] + 4096u, 1024);
        else if (c == 2) SOCMX_GLD(fr[2][j], cur[j] + 4096u, 2048);
        else if (c == 3) SOCMX_GLD(fr[3][j], cur[j] + 4096u, 3072);
        else if (c == 4) SOCMX_GLD(fr[0][j], cur[j], 0);              // (cur = the NEXT group's offsets from here on)
        else if (c == 5) SOCMX_GLD(fr[1][j], cur[j], 1024);
        else if (c == 6) SOCMX_GLD(fr[2][j], cur[j], 2048);
        else SOCMX_GLD(fr[3][j], cur[j], 3072);
      }
    };
    chunk(0); chunk(1); chunk(2); chunk(3);
    // this group's last refill from `cur` is out: the registers take the next group's offsets (two workgroups per CU need
    // the kernel under 128 VGPRs)
#pragma unroll
    for (int j = 0; j < 4; ++j) cur[j] = (unsigned)(blk_of(gn, j) * KC) * 1024u + (unsigned)lane * 16u;
    chunk(4); chunk(5); chunk(6); chunk(7);
    if (rowok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = wave + NW * (4 * grp + j);
        const int left = m.d2 - (b * 16 + 4 * g);           // floats of the row from this quad on
        if (b < NBLK && left >= 4) {
          *reinterpret_cast<f32x4u*>(orow_n + b * 16) = acc[0][j];
          *reinterpret_cast<f32x4u*>(orow_d + b * 16) = acc[1][j];
        } else if (b < NBLK && left > 0) {                  // (d % 2 == 1: the row ends inside the quad)
#pragma unroll
          for (int i = 0; i < 3; ++i)
            if (i < left) { orow_n[b * 16 + i] = acc[0][j][i]; orow_d[b * 16 + i] = acc[1][j][i]; }
        }
      }
    }
  }
#undef SOCMX_GLD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 2 : 1) void mnet_forward_kernel(const MArgs a) {   // (one workgroup per CU measured faster than two at 128 VGPRs: 1.72 vs 1.89 ms)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tile = blockIdx.x;
  const MDesc& m = a.m;
  // (wide form: the forward kernel's split-K scratch sits behind all three biases)
  const int sc = m.wide ? m.bias + m.bias_floats : m.scratch;
  m_forward_hidden<NW>(a, lds, tile, 0, 3, sc);
  if (m.wide && m.h1p == 128) {                   // (eight chunks per block)
    m_wide_out_stage<NW>(a, lds, tile);
    return;
  }
  MEpi e3{ME_OUT, lds, lds + m.bias + m.L[2].b_lds, -1, 0, 0, 0, nullptr, 0, a.net, a.dnet, m.d2, (int64_t)tile * 16, a.Np};
  m_stage<NW>(a.packed, m.L[2], lds, m.hh2, m.s2, lds + sc, e3);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void mnet_backward_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MDesc& m = a.m;
  const int tile = blockIdx.x, tid = threadIdx.x, nthr = NW * 64;
  const int64_t tile_rows = (int64_t)a.ntiles * 32;      // slab tiles are 16 rows: two per workgroup
  const int64_t p0 = (int64_t)tile * 16;
  // upstream gradients: rows 0..15 d obj / d net, rows 16..31 d obj / d dnet (zero past Np and in the padding units)
  float* G = lds + m.gout;
  float* slG = a.ws + (size_t)tile_rows * m.pre[MT_GOUT] + (size_t)(2 * tile) * m.d2p * 16;
  for (int e = tid; e < 32 * m.d2p; e += nthr) {
    const int r16 = e & 15, hc = e >> 4;
    const int h = hc >= m.d2p ? 1 : 0, c = hc - h * m.d2p;
    const float* src = h ? a.gdnet : a.gnet;
    const float v = (p0 + r16 < a.Np && c < m.d2) ? src[(size_t)(p0 + r16) * m.d2 + c] : 0.f;
    G[(h * 16 + r16) * m.so + c] = v;
    slG[e] = v;
  }
  m_forward_hidden<NW>(a, lds, tile, tile_rows);          // (its first barrier also covers the G tile)
  float* scratch = lds + m.scratch;
  MEpi b1{ME_MASK, lds, nullptr, m.gz2, m.s2, m.m2, m.h1p,
          a.ws + (size_t)tile_rows * m.pre[MT_GZ2] + (size_t)(2 * tile) * m.h1p * 16, m.h1p, nullptr, nullptr, 0, 0, 0};
  m_stage<NW>(a.packedT, m.LT[0], lds, m.gout, m.so, scratch, b1);
  MEpi b2{ME_MASK, lds, nullptr, -1, 0, m.m1, m.h0p,
          a.ws + (size_t)tile_rows * m.pre[MT_GZ1] + (size_t)(2 * tile) * m.h0p * 16, m.h0p, nullptr, nullptr, 0, 0, 0};
  m_stage<NW>(a.packedT, m.LT[1], lds, m.gz2, m.s2, scratch, b2);
}

// ---- RESIDENT form (hidden widths <= 128 and d*d <= 128: the reference's hdims_M = [128, 128] at d <= 11 -- BASELINE
// configs[1], [2]) ---------------------------------------------------------------------------------------------------------
// The tile kernels above launch ONE 16-pair tile per workgroup and stream the whole weight image (118 KB at d = 10) from L2
// through a register ring for 32 rows of MFMA work: 150 MB of L2 traffic per launch at configs[2], and every stage starts with
// an L2 round trip (0.27 / 0.36 of the fp32 MFMA peak stand-alone, 0.16-0.20 beside the rollout -- profiles/r5).  With at most
// eight 16-wide output blocks per layer, wave w of eight OWNS output block w of every layer, and that block's fragments are
// 4 x (fan-in / 16) <= 32 registers per layer: the whole image is REGISTER-RESIDENT across the workgroup (68 VGPRs per lane
// forward, 100 backward), loaded once; a persistent workgroup then walks tiles b, b + grid, ... with nothing but LDS reads of
// the activation tile (the MFMA B operand, shared by the eight waves) and MFMAs in its loop.  ReLU masks stay in the
// registers of the wave that made them (the backward stage that needs a layer's mask produces the same units in the same
// lanes).  Two barriers per tile.  Same slabs / outputs as the kernels above, so kernel B and the finish kernel are unchanged.
struct MResFrag {
  f32x4 k[8];
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// fragments (nb = wave, kc = 0 .. KC-1) of one layer of a fragment-ordered image (zero beyond KC / for waves without a block)
__device__ __forceinline__ void mres_load(MResFrag& f, const float* img, const LayerDesc& L, int wave, int lane) {
  const int KC = L.in_pad >> 4, NB = L.out_pad >> 4;
#pragma unroll
  for (int kc = 0; kc < 8; ++kc)
    f.k[kc] = (wave < NB && kc < KC) ? ldg4(img + L.w_off + ((size_t)(wave * KC + kc) * 64 + lane) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// (av | at) += W_block . (X value rows | X tangent rows): xv = tile + row * S + 4 g (tangent rows 16 S further)
__device__ __forceinline__ void mres_gemm(const MResFrag& f, int KC, const float* xv, int S, f32x4& av, f32x4& at) {
  const float* xt = xv + 16 * S;
#pragma unroll
  for (int kc = 0; kc < 8; ++kc)
    if (kc < KC) {
      const f32x4 bv = lds4(xv + 16 * kc), bt = lds4(xt + 16 * kc);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av = __builtin_amdgcn_mfma_f32_16x16x4f32(f.k[kc][i], bv[i], av, 0, 0, 0);
        at = __builtin_amdgcn_mfma_f32_16x16x4f32(f.k[kc][i], bt[i], at, 0, 0, 0);
      }
    }
}

// ReLU pair: value -> relu (NaN kept), tangent -> tangent where value > 0; returns the four mask bits
__device__ __forceinline__ unsigned mres_relu(f32x4& v, f32x4& t) {
  unsigned msk = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = v[i] > 0.f;
    msk |= (on ? 1u : 0u) << i;
    v[i] = relu_keep_nan(v[i]);
    t[i] = on ? t[i] : 0.f;
  }
  return msk;
}
__device__ __forceinline__ void mres_mask(unsigned msk, f32x4& v, f32x4& t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = (msk >> i) & 1u;
    v[i] = on ? v[i] : 0.f;
    t[i] = on ? t[i] : 0.f;
  }
}

// the (value | tangent) input tile of a 16-pair tile: value rows [t, s, z?, 0 ...], tangent rows d/ds = [0, 1, 0 ...]
__device__ __forceinline__ void mres_write_x(const MArgs& a, float* X, int sx, int tile, float* slab) {
  const int64_t p0 = (int64_t)tile * 16;
  for (int e = threadIdx.x; e < 32 * 16; e += blockDim.x) {
    const int r = e >> 4, c = e & 15;
    float v = 0.f;
    if (r < 16) {
      const int64_t p = min(p0 + r, a.Np - 1);
      v = c == 0 ? a.t[p] : (c == 1 ? a.s[p] : ((c == 2 && a.z) ? a.z[p] : 0.f));
    } else {
      v = c == 1 ? 1.f : 0.f;
    }
    X[r * sx + c] = v;
    if (slab) slab[(r >> 4) * 256 + c * 16 + (r & 15)] = v;      // [h][unit][row16]
  }
}

__host__ __device__ inline bool mres_ok(const MDesc& m) { return !m.wide && m.h0p <= 128 && m.h1p <= 128 && m.d2p <= 128; }
__host__ __device__ inline int mres_fwd_lds_floats(const MDesc& m) { return 32 * (m.sx + m.s1 + m.s2); }
__host__ __device__ inline int mres_bwd_lds_floats(const MDesc& m) { return 32 * (m.sx + m.s1 + m.s2 + 2 * m.so); }

__global__ __launch_bounds__(512, 2) void mnet_forward_resident_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MDesc& m = a.m;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, g = lane >> 4;
  const int NB0 = m.h0p >> 4, NB1 = m.h1p >> 4, NB2 = m.d2p >> 4;
  float* X = lds;
  float* H1 = X + 32 * m.sx;
  float* H2 = H1 + 32 * m.s1;
  MResFrag w0, w1, w2;
  mres_load(w0, a.packed, m.L[0], wave, lane);
  mres_load(w1, a.packed, m.L[1], wave, lane);
  mres_load(w2, a.packed, m.L[2], wave, lane);
  const int n0 = 16 * wave + 4 * g;                         // this lane's four units of the wave's block
  const f32x4 b0 = wave < NB0 ? ldg4(a.packed + m.L[0].b_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 b1 = wave < NB1 ? ldg4(a.packed + m.L[1].b_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 b2 = wave < NB2 ? ldg4(a.packed + m.L[2].b_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  int tile = blockIdx.x;
  if (tile < a.ntiles) mres_write_x(a, X, m.sx, tile, nullptr);
  __syncthreads();
  for (; tile < a.ntiles; tile += gridDim.x) {
    if (wave < NB0) {                                       // F1: H1 = relu-pair(L0 X + b)
      f32x4 v = b0, t = zero;
      mres_gemm(w0, 1, X + row * m.sx + 4 * g, m.sx, v, t);
      mres_relu(v, t);
      *reinterpret_cast<f32x4*>(H1 + row * m.s1 + n0) = v;
      *reinterpret_cast<f32x4*>(H1 + (16 + row) * m.s1 + n0) = t;
    }
    __syncthreads();                                        // (A)
    if (wave < NB1) {                                       // F2: H2 = relu-pair(L1 H1 + b)
      f32x4 v = b1, t = zero;
      mres_gemm(w1, NB0, H1 + row * m.s1 + 4 * g, m.s1, v, t);
      mres_relu(v, t);
      *reinterpret_cast<f32x4*>(H2 + row * m.s2 + n0) = v;
      *reinterpret_cast<f32x4*>(H2 + (16 + row) * m.s2 + n0) = t;
    }
    // the next tile's inputs (X was last read in F1, in front of barrier A)
    if (tile + (int)gridDim.x < a.ntiles) mres_write_x(a, X, m.sx, tile + gridDim.x, nullptr);
    __syncthreads();                                        // (B)
    if (wave < NB2) {                                       // F3: net | dnet = L2 (h2 | t2) + b, straight to HBM
      f32x4 v = b2, t = zero;
      mres_gemm(w2, NB1, H2 + row * m.s2 + 4 * g, m.s2, v, t);
      MEpi e3{ME_OUT, lds, nullptr, -1, 0, 0, 0, nullptr, 0, a.net, a.dnet, m.d2, (int64_t)tile * 16, a.Np};
      e3.fin(v, t, row, n0);
    }
    // (no barrier: F1 of the next tile writes H1, last read in front of B; F2 writes H2 behind the next A, which every wave
    //  reaches only after its F3 reads)
  }
}

__global__ __launch_bounds__(512, 1) void mnet_backward_resident_kernel(const MArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MDesc& m = a.m;
  const int lane = threadIdx.x & 63, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane & 15, g = lane >> 4;
  const int NB0 = m.h0p >> 4, NB1 = m.h1p >> 4, NB2 = m.d2p >> 4;
  float* X = lds;
  float* H1 = X + 32 * m.sx;
  float* GZ2 = H1 + 32 * m.s1;
  float* Gb[2] = {GZ2 + 32 * m.s2, GZ2 + 32 * m.s2 + 32 * m.so};
  const int64_t tile_rows = (int64_t)a.ntiles * 32;
  MResFrag w0, w1, t2, t1;
  mres_load(w0, a.packed, m.L[0], wave, lane);
  mres_load(w1, a.packed, m.L[1], wave, lane);
  mres_load(t2, a.packedT, m.LT[0], wave, lane);            // L2^T: d2p -> h1p
  mres_load(t1, a.packedT, m.LT[1], wave, lane);            // L1^T: h1p -> h0p
  const int n0 = 16 * wave + 4 * g;
  const f32x4 b0 = wave < NB0 ? ldg4(a.packed + m.L[0].b_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 b1 = wave < NB1 ? ldg4(a.packed + m.L[1].b_off + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // upstream gradients of a tile: rows 0..15 d obj / d net, rows 16..31 d obj / d dnet (zero past Np and in the padding units)
  auto load_g = [&](int tile, float* G) {
    const int64_t p0 = (int64_t)tile * 16;
    float* slG = a.ws + (size_t)tile_rows * m.pre[MT_GOUT] + (size_t)(2 * tile) * m.d2p * 16;
    for (int e = tid; e < 32 * m.d2p; e += 512) {
      const int r16 = e & 15, hc = e >> 4;
      const int h = hc >= m.d2p ? 1 : 0, c = hc - h * m.d2p;
      const float* src = h ? a.gdnet : a.gnet;
      const float v = (p0 + r16 < a.Np && c < m.d2) ? src[(size_t)(p0 + r16) * m.d2 + c] : 0.f;
      G[(h * 16 + r16) * m.so + c] = v;
      slG[e] = v;
    }
  };
  int tile = blockIdx.x, it = 0;
  if (tile < a.ntiles) {
    mres_write_x(a, X, m.sx, tile, a.ws + (size_t)tile_rows * m.pre[MT_X] + (size_t)(2 * tile) * 256);
    load_g(tile, Gb[0]);
  }
  __syncthreads();
  for (; tile < a.ntiles; tile += gridDim.x, ++it) {
    float* G = Gb[it & 1];
    unsigned m1 = 0, m2 = 0;
    if (wave < NB0) {                                       // F1
      f32x4 v = b0, t = zero;
      mres_gemm(w0, 1, X + row * m.sx + 4 * g, m.sx, v, t);
      m1 = mres_relu(v, t);
      *reinterpret_cast<f32x4*>(H1 + row * m.s1 + n0) = v;
      *reinterpret_cast<f32x4*>(H1 + (16 + row) * m.s1 + n0) = t;
      float* sl = a.ws + (size_t)tile_rows * m.pre[MT_H1] + (size_t)(2 * tile) * m.h0p * 16;
      export4_u(sl, row, n0, v);
      export4_u(sl + (size_t)m.h0p * 16, row, n0, t);
    }
    __syncthreads();                                        // (A)
    if (wave < NB1) {
      {                                                     // F2: H2 leaves as a slab only (nothing downstream reads its tile)
        f32x4 v = b1, t = zero;
        mres_gemm(w1, NB0, H1 + row * m.s1 + 4 * g, m.s1, v, t);
        m2 = mres_relu(v, t);
        float* sl = a.ws + (size_t)tile_rows * m.pre[MT_H2] + (size_t)(2 * tile) * m.h1p * 16;
        export4_u(sl, row, n0, v);
        export4_u(sl + (size_t)m.h1p * 16, row, n0, t);
      }
      {                                                     // B1: (g_h2 | g_t2) = L2^T (g_net | g_dnet), masked by F2's sign
        f32x4 v = zero, t = zero;
        mres_gemm(t2, NB2, G + row * m.so + 4 * g, m.so, v, t);
        mres_mask(m2, v, t);
        *reinterpret_cast<f32x4*>(GZ2 + row * m.s2 + n0) = v;
        *reinterpret_cast<f32x4*>(GZ2 + (16 + row) * m.s2 + n0) = t;
        float* sl = a.ws + (size_t)tile_rows * m.pre[MT_GZ2] + (size_t)(2 * tile) * m.h1p * 16;
        export4_u(sl, row, n0, v);
        export4_u(sl + (size_t)m.h1p * 16, row, n0, t);
      }
    }
    // the next tile's inputs: X was last read in F1 (in front of A), the other G buffer in B1 of the previous tile
    const int nxt = tile + (int)gridDim.x;
    if (nxt < a.ntiles) {
      mres_write_x(a, X, m.sx, nxt, a.ws + (size_t)tile_rows * m.pre[MT_X] + (size_t)(2 * nxt) * 256);
      load_g(nxt, Gb[(it + 1) & 1]);
    }
    __syncthreads();                                        // (B)
    if (wave < NB0) {                                       // B2: (g_h1 | g_t1) = L1^T (gz2), masked by F1's sign: slab only
      f32x4 v = zero, t = zero;
      mres_gemm(t1, NB1, GZ2 + row * m.s2 + 4 * g, m.s2, v, t);
      mres_mask(m1, v, t);
      float* sl = a.ws + (size_t)tile_rows * m.pre[MT_GZ1] + (size_t)(2 * tile) * m.h0p * 16;
      export4_u(sl, row, n0, v);
      export4_u(sl + (size_t)m.h0p * 16, row, n0, t);
    }
  }
}

// ---- WIDE form (d*d outputs do not fit an LDS tile: BASELINE configs[4], d = 64) -----------------------------------------
// Backward tile kernel: F1, F2 recomputed as above, then  (g_h2, g_t2) = L2^T (g_net, g_dnet)  with the reduction over the
// d*d outputs SPLIT OVER THE (four) WAVES: wave w multiplies chunks [w KC / NW, (w+1) KC / NW) of all NOB = h1p / 16 output
// blocks for the value and the tangent rows (2 NOB accumulators).  The B operand is not an LDS tile: lane (row, g) reads
// g_net[p0 + row][16 kc + 4 g .. + 3] -- one 16-byte piece of the row-major gradient, the same (g, i) indexing the LDS
// fragments have -- so every byte of g_net / g_dnet is read from HBM once, by one wave.  The A fragments (L2^T, fragment
// order) stream from L2.  Three chunks (8 A + 2 B requests each) in flight, asm loads with written-out wait counts (see
// wgrad_body).  Four waves per workgroup, two workgroups per CU (their prologues, combines and barriers interleave); the
// combine and the masked epilogue are described in the kernel.  Then L1^T as above.
// q read `sh` floats in front of where it belongs: element i := element i + sh of the read, zero past its end (sh <= 0: unchanged)
__device__ __forceinline__ f32x4 quad_shift(f32x4 q, int sh) {
  if (sh <= 0) return q;
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = i + sh;
    r[i] = k == 1 ? q[1] : (k == 2 ? q[2] : (k == 3 ? q[3] : 0.f));
  }
  return r;
}

// GEN: rows of d*d floats that are not whole 16-wide blocks and / or more than NOB output blocks (the d % 4 = 0, <= 128-unit
// instantiation -- BASELINE configs[4] -- carries none of that code: it cost 0.36 ms of 1.57 at the slice)
template <int NW, int NOB, bool GEN>
__global__ __launch_bounds__(NW * 64, 2) void mnet_backward_wide_kernel(const MArgs a) {   // (two workgroups per CU = two waves per SIMD: the full register budget, no AGPR copies)
  static_assert(NW == 4, "the combine below is written for four waves");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const MDesc& m = a.m;
  const int tile = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile_rows = (int64_t)a.ntiles * 32;
  const int64_t p0 = (int64_t)tile * 16;
  m_forward_hidden<NW>(a, lds, tile, tile_rows, 2);
  const int row = lane & 15, g = lane >> 4;
  const int KC = m.d2p >> 4;                              // chunks of the reduction (the last one may end past the row)
  const int kc0 = (wave * KC) / NW, kc1 = ((wave + 1) * KC) / NW;
  // A last hidden layer beyond NOB = 8 blocks (129 .. 256 units) takes two passes of eight blocks over the gradient rows:
  // sixteen blocks' accumulators and fragment rings do not fit the registers (and a spilled ring register is copied BEFORE
  // its asm-issued load has landed).
  const int nob = m.h1p >> 4;
  float* sc = lds + m.wscratch;
  constexpr int HB = NOB / 2 > 0 ? NOB / 2 : 1;           // blocks per half
  MEpi b1{ME_MASK, lds, nullptr, m.gz2, m.s2, m.m2, m.h1p,
          a.ws + (size_t)tile_rows * m.pre[MT_GZ2] + (size_t)(2 * tile) * m.h1p * 16, m.h1p, nullptr, nullptr, 0, 0, 0};
  for (int jb = 0; jb < (GEN ? nob : 1); jb += NOB) {
  f32x4 acc[2][NOB];
#pragma unroll
  for (int j = 0; j < NOB; ++j) { acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (kc1 > kc0) {
    const bool rowok = p0 + row < a.Np;
    const int64_t prow = min(p0 + row, a.Np - 1);
    const float* bv = a.gnet + (size_t)prow * m.d2 + 4 * g;
    const float* bt = a.gdnet + (size_t)prow * m.d2 + 4 * g;
    // the row's last quads: read from at most d2 - 4 on (inside the row, hence inside the buffer) and shifted into place below
    const int klim = m.d2 - 4 - 4 * g;
    const bool ragged = GEN && (m.d2 & 15) != 0;
    const f32x4* wl = reinterpret_cast<const f32x4*>(a.packedT + m.LT[0].w_off) + lane;   // fragment (block j, chunk kc): (j KC + kc) 64
    constexpr int PD = 3, NL = NOB + 2;
    const int nobm1 = nob - 1;
    f32x4 fa[PD][NOB], fb[PD][2];
    auto load = [&](int kc, int sl) {
      const int k = min(kc, kc1 - 1);
#pragma unroll
      for (int j = 0; j < NOB; ++j)      // (blocks past h1p / 16 re-read the last one: their accumulators are never used)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fa[sl][j]) : "v"(wl + (size_t)(min(jb + j, nobm1) * KC + k) * 64) : "memory");
      const int ko = GEN ? min(k * 16, klim) : k * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fb[sl][0]) : "v"(bv + ko) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fb[sl][1]) : "v"(bt + ko) : "memory");
    };
    auto wait_slot = [&](int sl) {                 // the PD - 1 younger chunks may stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PD - 1) * NL) : "memory");
#pragma unroll
      for (int j = 0; j < NOB; ++j) asm volatile("" : "+v"(fa[sl][j]));
      asm volatile("" : "+v"(fb[sl][0]), "+v"(fb[sl][1]));
    };
#pragma unroll
    for (int sl = 0; sl < PD; ++sl) load(kc0 + sl, sl);
    for (int kc = kc0; kc < kc1; kc += PD) {
#pragma unroll
      for (int sl = 0; sl < PD; ++sl) {
        wait_slot(sl);
        if (kc + sl < kc1) {
          f32x4 bq[2] = {fb[sl][0], fb[sl][1]};
          if (!rowok) { bq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; bq[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          if (ragged && kc + sl == KC - 1) {      // (wave-uniform) quads read `sh` floats early: element i is element i + sh of the read
            const int sh = (kc + sl) * 16 - klim;
#pragma unroll
            for (int x = 0; x < 2; ++x) bq[x] = quad_shift(bq[x], sh);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NOB; ++j) {
              acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][j][i], bq[0][i], acc[0][j], 0, 0, 0);
              acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[sl][j][i], bq[1][i], acc[1][j], 0, 0, 0);
            }
        }
        load(kc + sl + PD, sl);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // Combine the four waves' partial sums in two halves of the output blocks, through an LDS scratch that ALIASES the dead
  // forward tiles (X, H1, H2: exported as slabs, only their ReLU masks are needed from here on) -- 2 x (h1p / 32) x 4 KiB per
  // half: the whole kernel stays under 80 KiB of LDS = two workgroups per CU.  Per half: every wave parks its quads of the
  // half's blocks, barrier, wave w reduces block(s) w, w + 4 of the half (four partials per tile) and runs the masked
  // epilogue (gz2 tile + GZ2 slab), barrier.
  __syncthreads();                                        // (everyone is done with the forward tiles)
#pragma unroll
  for (int half = 0; half < (NOB + HB - 1) / HB; ++half) {
#pragma unroll
    for (int jj = 0; jj < HB; ++jj) {
      const int j = half * HB + jj;
      if (j < NOB) {
        *reinterpret_cast<f32x4*>(sc + ((size_t)((wave * 2 + 0) * HB + jj) * 64 + lane) * 4) = acc[0][j];
        *reinterpret_cast<f32x4*>(sc + ((size_t)((wave * 2 + 1) * HB + jj) * 64 + lane) * 4) = acc[1][j];
      }
    }
    __syncthreads();
    for (int jj = wave; jj < HB; jj += NW) {
      const int j = jb + half * HB + jj;
      if (j < nob) {
        f32x4 vq = f32x4{0.f, 0.f, 0.f, 0.f}, tq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w4 = 0; w4 < NW; ++w4) {
          vq += lds4(sc + ((size_t)((w4 * 2 + 0) * HB + jj) * 64 + lane) * 4);
          tq += lds4(sc + ((size_t)((w4 * 2 + 1) * HB + jj) * 64 + lane) * 4);
        }
        b1.fin(vq, tq, row, j * 16 + 4 * g);
      }
    }
    __syncthreads();
  }
  }   // (pass)
  MEpi b2{ME_MASK, lds, nullptr, -1, 0, m.m1, m.h0p,
          a.ws + (size_t)tile_rows * m.pre[MT_GZ1] + (size_t)(2 * tile) * m.h0p * 16, m.h0p, nullptr, nullptr, 0, 0, 0};
  m_stage<NW>(a.packedT, m.LT[1], lds, m.gz2, m.s2, lds + m.scratch, b2);
}

// Weight / bias gradient of the WIDE last layer:  dW2[n][c] = sum_p g_net[p][n] h2[p][c] + g_dnet[p][n] t2[p][c],
// db2[n] = sum_p g_net[p][n]  (4096 x 128 outputs, 2 x 80,601 rows at BASELINE configs[4]: 169 GFLOP).
//   D (16 n x 16 c) += A (16 n x 4 p) . B (4 p x 16 c):  B = the H2 slab the tile kernel exported ([16-row tile][unit c][16 rows]:
//   lane (c, g) reads rows 4g..4g+3 with one 16-byte load -- value tile 2t, tangent tile 2t+1), A = g^T, which in the
//   row-major (Np, d*d) gradient is 4 floats from 4 DIFFERENT rows per lane: the 16 x 64 tile of g_net (and of g_dnet) is
//   loaded in 4 x 4 register blocks (16-byte pieces of four consecutive rows), transposed in registers and parked in LDS as
//   T[n][16 rows] (20-float stride: conflict-free 16-byte column writes and fragment reads -- the layout of
//   socm_target_bwd_lds2_kernel), double-buffered, one barrier per 16 pairs.
// Workgroup (4 waves) = 64 outputs n x all h1p inputs c x one slab of pair tiles; wave w owns c-blocks {NCB w .. NCB w + NCB - 1}
// x 4 n-blocks; waves 0 / 1 load the g_net / g_dnet tiles.  Workgroups reading one slab are numbered onto one XCD (their
// H2 tiles hit that L2).  Partials go to the cells kernel C adds up.
struct MWgradWideArgs {
  const float *gnet, *gdnet;
  const float* h2slab;       // [2 ntiles][h1p][16]
  float* part;               // (S, slab_floats)
  int64_t Np, slab_floats;
  int d2, h1p, ntiles, S, NG, IB;
  int cell_off, bias_off;    // float offsets of layer 2's cells / bias partial inside a slab of partials
};

template <int NCB, bool RAG>   // RAG: d*d is odd -- a row may end inside a 16-byte piece
__global__ __launch_bounds__(256, 2) void mnet_wgrad_wide_kernel(const MWgradWideArgs a) {
  __shared__ __attribute__((aligned(16))) float T[2][2][64][20];   // [stage][g_net, g_dnet][n][pair row]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int ng = q % a.NG;
  const int slab = (q / a.NG) * 8 + xcd;
  if (slab >= a.S) return;
  const int t0 = (int)(((int64_t)slab * a.ntiles) / a.S), t1 = (int)(((int64_t)(slab + 1) * a.ntiles) / a.S);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int n0 = ng * 64;
  // Loader duty, the same for every wave (equal instruction counts: nobody waits at the barrier for a loader): wave (x = w & 1:
  // g_net | g_dnet, h = w >> 1), lane (rg, pc) owns rows 4 rg + 2 h, + 1 and columns n0 + 4 pc .. + 3 of tensor x's 16 x 64 tile:
  // two 16-byte loads, transposed in registers, four 8-byte LDS writes T[x][4 pc + e][4 rg + 2 h .. + 1].
  const int lx = wave & 1, lh = wave >> 1;
  const int rg = lane & 3, pc = lane >> 2;
  const float* src = lx == 0 ? a.gnet : a.gdnet;
  const bool colok = n0 + 4 * pc < a.d2;
  const int ncol = colok ? min(n0 + 4 * pc, a.d2 - 4) : 0;        // (RAG: the row's last piece is read shifted back into the row)
  const int csh = RAG ? n0 + 4 * pc - ncol : 0;
  // B operand: H2 slab, value tile 2t (with g_net), tangent tile 2t + 1 (with g_dnet)
  const int cb0 = wave * NCB;
  f32x4 acc[4][NCB];
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t0 < t1) {
    // Requests are asm statements and the one wait per iteration is written out (vmcnt(6): the six requests of the previous
    // iteration may stay in flight): written as plain loads the compiler waited for every request right behind its issue
    // (vmcnt(0) in front of the MFMAs) -- the kernel ran at 58 % MFMA-busy.  Tile x is requested in iteration x - 3, parked
    // in LDS in iteration x - 1, multiplied in iteration x; the H2 fragments are requested two iterations ahead.
    f32x4 tl[2][2];                  // [tile slot][row]
    f32x4 bq[3][2][NCB];             // [slot][value | tangent][c-block]
    // (slots are plain ints, constant at every call site: after inlining each switch leaves one asm statement with a fixed
    //  register -- a generic lambda may not name captured variables in asm operands)
// (scalar base + 32-bit lane offset: a 64-bit address per lane is issue time the matrix pipe does not get back; the s_nop
    //  covers the five wait states a VALU-written SGPR -- v_readfirstlane -- needs before a VMEM instruction reads it)
#define SOCMX_GLD(dst, voff, sbase) \
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
    auto spin = [](const float* q) -> const float* {             // a wave-uniform pointer, pinned to scalar registers
      const uint64_t uq = reinterpret_cast<uint64_t>(q);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)uq), hi = __builtin_amdgcn_readfirstlane((uint32_t)(uq >> 32));
      return reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
    };
    auto tile_load = [&](int t, int sl) {
      const int tc = min(t, t1 - 1);
      const float* base = spin(src + (size_t)tc * 16 * a.d2);
      const int64_t p0 = (int64_t)tc * 16 + 4 * rg + 2 * lh;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t voff = (uint32_t)((int)(min(p0 + i, a.Np - 1) - (int64_t)tc * 16) * a.d2 + ncol) * 4u;
        if (sl == 0) SOCMX_GLD(tl[0][i], voff, base); else SOCMX_GLD(tl[1][i], voff, base);
      }
    };
    auto tile_park = [&](int t, int sl, int st) {
      const int64_t p0 = (int64_t)t * 16 + 4 * rg + 2 * lh;
      const bool ok0 = colok && p0 < a.Np, ok1 = colok && p0 + 1 < a.Np;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float2 col;
        if (RAG) {
          const f32x4 q0 = quad_shift(sl == 0 ? tl[0][0] : tl[1][0], csh), q1 = quad_shift(sl == 0 ? tl[0][1] : tl[1][1], csh);
          col.x = ok0 ? q0[e] : 0.f;
          col.y = ok1 ? q1[e] : 0.f;
        } else {
          col.x = ok0 ? (sl == 0 ? tl[0][0][e] : tl[1][0][e]) : 0.f;
          col.y = ok1 ? (sl == 0 ? tl[0][1][e] : tl[1][1][e]) : 0.f;
        }
        *reinterpret_cast<float2*>(&T[st][lx][4 * pc + e][4 * rg + 2 * lh]) = col;
      }
    };
    const uint32_t hoff = (uint32_t)(c16 * 16 + 4 * g4) * 4u;
    auto bload = [&](int t, int sl) {
      const int tc = min(t, t1 - 1);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int k = 0; k < NCB; ++k) {
          const int cb = min(cb0 + k, a.IB - 1);
          const float* base = spin(a.h2slab + ((size_t)(2 * tc + x) * a.h1p + cb * 16) * 16);
          if (sl == 0) SOCMX_GLD(bq[0][x][k], hoff, base); else if (sl == 1) SOCMX_GLD(bq[1][x][k], hoff, base); else SOCMX_GLD(bq[2][x][k], hoff, base);
        }
    };
#undef SOCMX_GLD
    // prologue: tile t0 straight into T[0]; then the two request groups the loop expects to find in flight
    tile_load(t0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(tl[0][0]), "+v"(tl[0][1]));
    tile_park(t0, 0, 0);
    tile_load(t0 + 1, 1); bload(t0, 0);
    tile_load(t0 + 2, 0); bload(t0 + 1, 1);
    auto iter = [&](int t, const int TS /* slot of tile t + 1 (= of tile t + 3) */, const int BS /* slot of B(t) */,
                    const int bnx /* of B(t + 2) */) {
      const int st = (t - t0) & 1;
      __syncthreads();                                   // T[st] holds tile t; everyone is done with T[st ^ 1]
      f32x4 af[2][4];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 4; ++j) af[x][j] = lds4(&T[st][x][j * 16 + c16][4 * g4]);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + 2 * NCB) : "memory");    // the group requested two iterations ago is here
      // (the wait covers every register of the older groups: tell the compiler all of them changed)
      asm volatile("" : "+v"(tl[0][0]), "+v"(tl[0][1]), "+v"(tl[1][0]), "+v"(tl[1][1]));
#pragma unroll
      for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int k = 0; k < NCB; ++k) asm volatile("" : "+v"(bq[sl][x][k]));
      f32x4 bcur[2][NCB];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int k = 0; k < NCB; ++k) bcur[x][k] = BS == 0 ? bq[0][x][k] : (BS == 1 ? bq[1][x][k] : bq[2][x][k]);
      if (t + 1 < t1) tile_park(t + 1, TS, st ^ 1);
      tile_load(t + 3, TS);
      bload(t + 2, bnx);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < NCB; ++k)
              acc[j][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[x][j][i], bcur[x][k][i], acc[j][k], 0, 0, 0);
      if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum[j] += (af[0][j][0] + af[0][j][1]) + (af[0][j][2] + af[0][j][3]);
      }
    };
    // relative iteration r = t - t0: tile t + 1 sits in slot (r + 1) & 1, B(t) in slot r % 3 -- six static cases
    for (int t = t0; t < t1; t += 6) {
      iter(t, 1, 0, 2);
      if (t + 1 < t1) iter(t + 1, 0, 1, 0);
      if (t + 2 < t1) iter(t + 2, 1, 2, 1);
      if (t + 3 < t1) iter(t + 3, 0, 0, 2);
      if (t + 4 < t1) iter(t + 4, 1, 1, 0);
      if (t + 5 < t1) iter(t + 5, 0, 2, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float* out = a.part + (size_t)slab * a.slab_floats;
  const int OB = (a.d2 + 15) >> 4;                         // (the last block's cells past d*d are zero: zero operand columns)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
      const int ob = ng * 4 + j, ib = cb0 + k;
      if (ob < OB && ib < a.IB)
        *reinterpret_cast<f32x4*>(out + a.cell_off + ((size_t)ob * a.IB + ib) * 256 + lane * 4) = acc[j][k];
    }
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float b = bsum[j];
      b += __shfl_xor(b, 16, 64);
      b += __shfl_xor(b, 32, 64);
      if (ng * 4 + j < OB && lane < 16) out[a.bias_off + (ng * 4 + j) * 16 + lane] = b;
    }
  }
}

struct MPackArgs {
  MDesc m;
  int d, h0, h1, n_in;
  const float *w0, *b0, *w1, *b1, *w2, *b2;   // torch layouts: (h0,2) (h0,) (h1,h0) (h1,) (d2,h1) (d2,)
  float *packed, *packedT;
};

__global__ void mnet_pack_kernel(const MPackArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int fin[3] = {a.n_in, a.h0, a.h1}, fout[3] = {a.h0, a.h1, a.d * a.d};
  const float* W[3] = {a.w0, a.w1, a.w2};
  const float* Bv[3] = {a.b0, a.b1, a.b2};
  if (idx < a.m.total_floats) {
    int l = 2;
    while (l > 0 && idx < a.m.L[l].w_off) --l;
    const LayerDesc L = a.m.L[l];
    float v = 0.f;
    if (idx >= L.b_off) {
      const int n = idx - L.b_off;
      if (n < fout[l]) v = Bv[l][n];
    } else {
      const int rel = idx - L.w_off, i = rel & 3, lane = (rel >> 2) & 63, chunk = rel >> 8, KC = L.in_pad >> 4;
      const int nb = chunk / KC, kc = chunk - nb * KC;
      const int n = nb * 16 + (lane & 15), kk = kc * 16 + 4 * (lane >> 4) + i;
      if (n < fout[l] && kk < fin[l]) v = W[l][(size_t)n * fin[l] + kk];
    }
    a.packed[idx] = v;
    return;
  }
  const int j = idx - a.m.total_floats;
  if (j < a.m.totalT_floats) {
    const int t = j >= a.m.LT[1].w_off ? 1 : 0;      // LT[0] = L2^T, LT[1] = L1^T
    const int l = t == 0 ? 2 : 1;
    const LayerDesc L = a.m.LT[t];
    const int rel = j - L.w_off, i = rel & 3, lane = (rel >> 2) & 63, chunk = rel >> 8, KC = L.in_pad >> 4;
    const int nb = chunk / KC, kc = chunk - nb * KC;
    const int n = nb * 16 + (lane & 15), kk = kc * 16 + 4 * (lane >> 4) + i;   // W^T[n][kk] = W[kk][n]
    float v = 0.f;
    if (n < fin[l] && kk < fout[l]) v = W[l][(size_t)kk * fin[l] + n];
    a.packedT[j] = v;
  }
}
#endif  // __HIPCC__ (K3)
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

// which instantiation of kernel A: 0 = descriptor-driven, 1..3 = the constexpr ones (default hidden widths)
static int k2_variant(const UnetDesc& u) {
  static const bool force_generic = getenv("SOCMX_GENERIC") != nullptr;
  if (force_generic || u.hp[0] != SOCMX_H0P || u.hp[1] != SOCMX_H1P || u.hp[2] != SOCMX_H2P) return 0;
  if (u.in0p == 16 && u.outp == 16) return 1;
  if (u.in0p == 32 && u.outp == 32) return 2;
  if (u.in0p == 80 && u.outp == 64) return 3;
  return 0;
}

// Fills `sc` for the block groups of `OB x IB` (9 layers, numbered like kernel B numbers them) and returns the largest
// S_i; nslots = 0 (uniform grid) when the table cannot describe the net.  Cost of a group per row tile = the MFMAs of the
// instantiation that runs it (clamped blocks are computed twice) + its loads' share of the issue slots.
static int wgrad_make_schedule(const int OB[9], const int IB[9], int ntiles, WgSched& sc) {
  sc.nslots = 0;
  float cost[kWgMaxItems];
  int ni = 0;
  for (int l = 0; l < 9; ++l)
    for (int og = 0; og < (OB[l] + 3) / 4; ++og)
      for (int ig = 0; ig < (IB[l] + 3) / 4; ++ig) {
        if (ni >= kWgMaxItems) return 0;
        const int nob = std::min(4, OB[l] - 4 * og), nib = std::min(4, IB[l] - 4 * ig);
        int vo = nob > 2 ? 4 : (nob > 1 ? 2 : 1), vi = nib > 2 ? 4 : (nib > 1 ? 2 : 1);
        if (vo == 2 && vi == 1) vi = 2;
        if (vo == 1 && vi == 2) vo = 2;
        cost[ni++] = (float)(vo * vi) + 0.125f * (float)(vo + vi);
      }
  if (ntiles < 64 * 8) return 0;               // (small inputs: the uniform grid with few slabs)
  // n[i] by largest remainder: the next wave goes to the group whose waves are longest
  int n[kWgMaxItems];
  for (int i = 0; i < ni; ++i) n[i] = 1;
  for (int left = kWgSlots - ni; left > 0; --left) {
    int arg = -1;
    float worst = 0.f;
    for (int i = 0; i < ni; ++i)
      if (n[i] < 31 && cost[i] / (float)n[i] > worst) { worst = cost[i] / (float)n[i]; arg = i; }
    if (arg < 0) break;
    ++n[arg];
  }
  // Slot order: slab index first, groups inside it by falling wave length -- the four waves of a workgroup are then
  // different groups on (nearly) the same row tiles, which they re-read from L2 while they walk in step.  (Measured, cfg3 /
  // d = 64 slice: 96.8 / 763 us; group-major order 98.7 / 761; a second round of half-length waves 97.5 / 768 and kernel C
  // twice as long; the uniform grid 112.7 / 943.)
  int order[kWgMaxItems];
  for (int i = 0; i < ni; ++i) order[i] = i;
  std::sort(order, order + ni, [&](int x, int y) {
    const float cx = cost[x] / (float)n[x], cy = cost[y] / (float)n[y];
    return cx != cy ? cx > cy : x < y;
  });
  int ws = 0, smax = 0;
  for (int i = 0; i < ni; ++i) { sc.n[i] = (uint8_t)n[i]; smax = std::max(smax, 8 * n[i]); }
  for (int k = 0; k < 32; ++k)
    for (int oi = 0; oi < ni; ++oi)
      if (k < n[order[oi]]) { sc.item[ws] = (uint8_t)order[oi]; sc.k[ws] = (uint8_t)k; ++ws; }
  for (; ws < kWgSlots; ++ws) { sc.item[ws] = 255; sc.k[ws] = 0; }
  sc.nslots = kWgSlots;
  return smax;
}

struct K2Plan {
  WgSched sched;
  int variant, rt;
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
  int64_t fold_floats;     // G' and s behind the partials (kernel C -> kernel D)
};

static const int kK2Waves = 8;

static int k2_plan(int32_t d, const int32_t hdims[3], int64_t N, K2Plan& p) {
  if (!hdims) return SOCMX_E_NULL;
  if (!k2_dims_ok(d, hdims) || N < 1) return SOCMX_E_DIM;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  p.u = make_unet_desc(d, h);
  p.bd = make_bwd_desc(p.u);
  p.variant = k2_variant(p.u);
  p.rt = p.variant ? SOCMX_K2A_RT : 1;
  p.lay = make_bwd_layout(p.u, p.variant ? SOCMX_K2A_WAVES : kK2Waves, p.rt);
  if (p.variant && (size_t)p.lay.floats * sizeof(float) > (size_t)kLdsBytesPerCU) {
    // (a variant build with wide layers: two row tiles per workgroup do not fit -- the descriptor-driven form with one)
    p.variant = 0; p.rt = 1;
    p.lay = make_bwd_layout(p.u, kK2Waves, p.rt);
  }
  if ((size_t)p.lay.floats * sizeof(float) > (size_t)kLdsBytesPerCU) return SOCMX_E_LDS;
  if ((N + 15) / 16 > (int64_t)1 << 27) return SOCMX_E_DIM;
  p.ntiles = (int)((N + 16 * p.rt - 1) / (16 * p.rt)) * p.rt;     // 16-row tiles, whole workgroups (padding tiles: zero gradient)
  int wsum = 0;
  for (int t = 0; t < T_N; ++t) wsum += tensor_width(p.u, t);
  p.ws_floats = (int64_t)p.ntiles * 16 * wsum;
  unet_layer_dims(d, h, p.fin, p.fout);
  int off = 0, items = 0;
  int64_t goff = 0;
  for (int l = 0; l < 9; ++l) {
    // (layer 4's slot: not dW_res_1 = GO1^T R1 but G' = ZU0^T R1 -- outp x h0, the fold; kernel D makes dW_res_1 of it)
    p.OB[l] = (l == 4 ? p.u.fold.out_pad : p.u.L[l].out_pad) >> 4;
    p.IB[l] = p.u.L[l].in_pad >> 4;
    p.w_cell_off[l] = off;
    off += p.OB[l] * p.IB[l] * 256;
    items += ((p.OB[l] + 3) / 4) * ((p.IB[l] + 3) / 4);
    p.gw_off[l] = goff; goff += (int64_t)p.fin[l] * p.fout[l];
    p.gb_off[l] = goff; goff += p.fout[l];
  }
  p.grad_floats = goff;
  p.total_cells_floats = off;
  for (int l = 0; l < 9; ++l) { p.b_part_off[l] = off; off += l == 4 ? p.u.fold.out_pad : p.u.L[l].out_pad; }
  p.total_bias = off - p.total_cells_floats;
  p.slab_floats = off;
  p.n_items = items;
  p.fold_floats = ((int64_t)d * h[0] + d + 63) & ~(int64_t)63;
  // Slabs of row tiles for kernel B's UNIFORM grid (small inputs, and nets the wave schedule's table cannot describe): one
  // wave per (block group, slab), four slabs per workgroup; the slab GROUPS must divide evenly over the 8 XCDs -- kernel B
  // numbers them onto XCDs -- so S is a multiple of 32; at least 4 tiles per slab.
  (void)items;
#ifndef SOCMX_K2_SLABS
#define SOCMX_K2_SLABS 64
#endif
  int S = SOCMX_K2_SLABS;
  if (S > p.ntiles / 4) S = p.ntiles / 4 >= 32 ? 32 : p.ntiles / 4;
  p.S = S < 1 ? 1 : (S > 128 ? 128 : S);
  // (large inputs: the cost-balanced schedule, WgSched; its largest per-group slab count sizes the partials)
  if (const int smax = wgrad_make_schedule(p.OB, p.IB, p.ntiles, p.sched)) p.S = smax;
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
  if (const int err = launch(unet_pack_bwd_kernel, dim3(blocks), dim3(threads), 0, stream, a)) return err;
  // the KT_R1 slot: F^T = (up_0 res_1)^T as a layer of h0 outputs and outp inputs (the fold: see the stage table)
  return unet_fold_launch(net->weight[8], net->weight[4], nullptr, h[0], net->d, u.fold.in_pad, u.fold.out_pad,
                          packedT + a.bd.L[KT_R1].w_off, nullptr, 1, stream);
}

extern "C" int socmx_unet_backward_sizes(int32_t d, const int32_t hdims[3], int64_t N, int64_t* workspace_floats,
                                         int64_t* grad_floats) {
  K2Plan p;
  const int rc = k2_plan(d, hdims, N, p);
  if (rc) return rc;
  if (workspace_floats) *workspace_floats = p.ws_floats + p.part_floats + p.fold_floats;
  if (grad_floats) *grad_floats = p.grad_floats;
  return 0;
}

extern "C" int socmx_unet_backward_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                       const float* x, const float* ts, int32_t rows_per_t, int64_t N,
                                       const float* gout, float* workspace, float* grads, socmx_stream_t stream) {
  return socmx_unet_backward_scaled_f32(packed, packedT, d, hdims, x, ts, rows_per_t, N, gout, nullptr, workspace, grads, stream);
}

static int unet_backward_impl(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3], const float* x,
                              const float* ts, int32_t rows_per_t, int64_t N, const float* gout, const float* gout_scale,
                              const uint32_t* records, float* workspace, float* grads, socmx_stream_t stream);

extern "C" int socmx_unet_backward_scaled_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                              const float* x, const float* ts, int32_t rows_per_t, int64_t N,
                                              const float* gout, const float* gout_scale, float* workspace, float* grads,
                                              socmx_stream_t stream) {
  return unet_backward_impl(packed, packedT, d, hdims, x, ts, rows_per_t, N, gout, gout_scale, nullptr, workspace, grads, stream);
}

extern "C" int socmx_unet_backward_saved_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                             const float* x, const float* ts, int32_t rows_per_t, int64_t N,
                                             const float* gout, const float* gout_scale, const uint32_t* records,
                                             float* workspace, float* grads, socmx_stream_t stream) {
  if (!records) return SOCMX_E_NULL;
  return unet_backward_impl(packed, packedT, d, hdims, x, ts, rows_per_t, N, gout, gout_scale, records, workspace, grads, stream);
}

static int unet_backward_impl(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3], const float* x,
                              const float* ts, int32_t rows_per_t, int64_t N, const float* gout, const float* gout_scale,
                              const uint32_t* records, float* workspace, float* grads, socmx_stream_t stream) {
  if (!packed || !packedT || !x || !ts || !gout || !workspace || !grads) return SOCMX_E_NULL;
  if (rows_per_t < 1) return SOCMX_E_DIM;
  K2Plan p;
  if (const int rc = k2_plan(d, hdims, N, p)) return rc;
  // (the SAVED form: what the one-row rollout exports -- the 16-wide default-width network, whole tiles)
  if (records && (p.variant != 1 || p.rt != 1 || N % 16 != 0)) return SOCMX_E_DIM;
  // ---- kernel A ----
  TileArgs ta;
  ta.u = p.u; ta.bd = p.bd; ta.lay = p.lay;
  for (int si = 0; si < kBwdStages; ++si) ta.prog.st[si] = k2_stage_desc(p.u, p.bd, p.lay, si);
  ta.packed = packed; ta.packedT = packedT; ta.x = x; ta.ts = ts; ta.gout = gout; ta.gscale = gout_scale; ta.ws = workspace;
  ta.N = N; ta.rows_per_t = rows_per_t; ta.ntiles = p.ntiles; ta.rec = records;
  const size_t lds_bytes = (size_t)(records ? make_bwd_layout(p.u, SOCMX_K2A_WAVES, 1, true).floats : p.lay.floats) * sizeof(float);
  void (*kern)(const TileArgs) = unet_bwd_tile_kernel<kK2Waves, void>;
  if (p.variant == 1 && records) kern = unet_bwd_tile_kernel<SOCMX_K2A_WAVES, StaticNet<16, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 16>, true>;
  else if (p.variant == 1) kern = unet_bwd_tile_kernel<SOCMX_K2A_WAVES, StaticNet<16, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 16>>;
  else if (p.variant == 2) kern = unet_bwd_tile_kernel<SOCMX_K2A_WAVES, StaticNet<32, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 32>>;
  else if (p.variant == 3) kern = unet_bwd_tile_kernel<SOCMX_K2A_WAVES, StaticNet<80, SOCMX_H0P, SOCMX_H1P, SOCMX_H2P, 64>>;
  if (const int err = ensure_max_lds(kern)) return err;
  const int nwa = p.variant ? SOCMX_K2A_WAVES : kK2Waves;
  if (const int err = launch(kern, dim3(p.ntiles / p.rt), dim3(nwa * 64), lds_bytes, stream, ta)) return err;
  // ---- kernel B ----
  WgradArgs wa;
  wa.n_items = 0; wa.S = p.S; wa.ntiles = p.ntiles; wa.slab_floats = p.slab_floats; wa.bias_even_tiles = 0;
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
  wa.sched = p.sched;
  const int slab_groups = (((p.S + 3) / 4) + 7) & ~7;       // padded to a multiple of the XCD count
  const unsigned wgrid = p.sched.nslots ? 8u * (unsigned)(p.sched.nslots / 4) : (unsigned)(wa.n_items * slab_groups);
  if (const int err = launch(unet_wgrad_kernel, dim3(wgrid), dim3(256), 0, stream, wa)) return err;
  // ---- kernel C ----
  FinishArgs fa;
  fa.scheduled = p.sched.nslots != 0;
  std::memcpy(fa.item0, wa.item0, sizeof(fa.item0));
  std::memcpy(fa.n, p.sched.n, sizeof(fa.n));
  for (int l = 0; l < 9; ++l) {
    fa.w_cell_off[l] = p.w_cell_off[l]; fa.b_part_off[l] = p.b_part_off[l]; fa.OB[l] = p.OB[l]; fa.IB[l] = p.IB[l];
    fa.fin[l] = p.fin[l]; fa.fout[l] = p.fout[l]; fa.gw_off[l] = p.gw_off[l]; fa.gb_off[l] = p.gb_off[l];
  }
  fa.total_cells_floats = p.total_cells_floats; fa.total_bias = p.total_bias; fa.S = p.S; fa.slab_floats = p.slab_floats;
  fa.part = wa.part; fa.grads = grads;
  fa.fold_g = workspace + p.ws_floats + p.part_floats; fa.fold_fout = d;
  const int nthreads = p.total_cells_floats + p.total_bias;
  if (const int err = launch(unet_wgrad_finish_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, stream, fa)) return err;
  // ---- kernel D ----
  UnfoldArgs ua;
  ua.u = p.u; ua.d = d; ua.h0 = hdims[0]; ua.packed = packed; ua.fold_g = fa.fold_g; ua.grads = grads;
  ua.gw4 = p.gw_off[4]; ua.gb4 = p.gb_off[4]; ua.gw8 = p.gw_off[8];
  const int h0 = hdims[0];
  const unsigned nb1 = (unsigned)((h0 + 3) / 4 * ((h0 + 63) / 64)), nb2 = (unsigned)(d * ((h0 + 15) / 16));
  return launch(unet_unfold_kernel, dim3(nb1 + nb2), dim3(256), 0, stream, ua);
}

// ---- K3: the pair-grid network ---------------------------------------------------------------------------------------
// developer A/B switch, read once: SOCMX_K3_TILE=1 keeps the one-tile-per-workgroup kernels
static int mnet_tile_env() { static const int v = [] { const char* e = getenv("SOCMX_K3_TILE"); return e ? atoi(e) : 0; }(); return v; }

static const int kK3WideWaves = 4;   // the WIDE pair-net kernels: four waves per workgroup, two workgroups per CU

static int mnet_plan(int32_t d, const int32_t hdims[2], MDesc& m) {
  if (!hdims) return SOCMX_E_NULL;
  if (d < 1 || d > 1024 || hdims[0] < 1 || hdims[0] > 4096 || hdims[1] < 1 || hdims[1] > 4096) return SOCMX_E_DIM;
  m = make_mdesc(d, hdims[0], hdims[1], kK2Waves);
  if (m.wide) m = make_mdesc(d, hdims[0], hdims[1], kK3WideWaves);      // (the small-stage scratch depends on the wave count)
  if ((size_t)m.lds_floats * sizeof(float) > (size_t)kLdsBytesPerCU) return SOCMX_E_LDS;
  return 0;
}

extern "C" size_t socmx_mnet_packed_floats(int32_t d, const int32_t hdims[2]) {
  MDesc m;
  if (mnet_plan(d, hdims, m)) return 0;
  return (size_t)m.total_floats + (size_t)m.totalT_floats;
}

extern "C" int socmx_mnet_pack_f32(int32_t d, const int32_t hdims[2], int32_t n_in, const float* w0, const float* b0,
                                   const float* w1, const float* b1, const float* w2, const float* b2, float* packed,
                                   socmx_stream_t stream) {
  if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !packed) return SOCMX_E_NULL;
  if (n_in != 2 && n_in != 3) return SOCMX_E_DIM;
  MPackArgs a;
  if (const int rc = mnet_plan(d, hdims, a.m)) return rc;
  a.d = d; a.h0 = hdims[0]; a.h1 = hdims[1]; a.n_in = n_in;
  a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
  a.packed = packed; a.packedT = packed + a.m.total_floats;
  const int n = a.m.total_floats + a.m.totalT_floats;
  return launch(mnet_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
}

extern "C" int socmx_mnet_forward_f32(const float* packed, int32_t d, const int32_t hdims[2], const float* t,
                                      const float* s, const float* z, int64_t Np, float* net, float* dnet,
                                      socmx_stream_t stream) {
  if (!packed || !t || !s || !net || !dnet) return SOCMX_E_NULL;
  if (Np < 1 || Np > ((int64_t)1 << 30)) return SOCMX_E_DIM;
  MArgs a{};
  if (const int rc = mnet_plan(d, hdims, a.m)) return rc;
  a.packed = packed; a.packedT = packed + a.m.total_floats; a.t = t; a.s = s; a.z = z; a.Np = Np;
  a.ntiles = (int)((Np + 15) / 16); a.net = net; a.dnet = dnet; a.ws = nullptr;
  const size_t lds_bytes = (size_t)a.m.lds_fwd_floats * sizeof(float);
  if (a.m.wide) {
    if (const int err = ensure_max_lds(mnet_forward_kernel<kK3WideWaves>)) return err;
    return launch(mnet_forward_kernel<kK3WideWaves>, dim3(a.ntiles), dim3(kK3WideWaves * 64), lds_bytes, stream, a);
  }
  if (mres_ok(a.m) && !mnet_tile_env()) {
    // persistent workgroups with register-resident weights: two per CU, tiles b, b + grid, ...
    if (const int err = ensure_max_lds(mnet_forward_resident_kernel)) return err;
    // (developer sweep, read once: SOCMX_K3_GRID = persistent workgroups of the forward kernel)
    static const int grid_env = [] { const char* e = getenv("SOCMX_K3_GRID"); return e ? atoi(e) : 0; }();
    return launch(mnet_forward_resident_kernel, dim3(std::min(a.ntiles, grid_env > 0 ? grid_env : 512)), dim3(512),
                  (size_t)mres_fwd_lds_floats(a.m) * sizeof(float), stream, a);
  }
  if (const int err = ensure_max_lds(mnet_forward_kernel<kK2Waves>)) return err;
  return launch(mnet_forward_kernel<kK2Waves>, dim3(a.ntiles), dim3(kK2Waves * 64), lds_bytes, stream, a);
}

struct MnetBwdPlan {
  MDesc m;
  int ntiles, S;
  int OB[3], IB[3], w_cell_off[3], b_part_off[3], fin[3], fout[3];
  int total_cells_floats, total_bias;
  int64_t ws_floats, slab_floats, part_floats, gw_off[3], gb_off[3], grad_floats;
};

static int mnet_bwd_plan(int32_t d, const int32_t hdims[2], int32_t n_in, int64_t Np, MnetBwdPlan& p) {
  if (const int rc = mnet_plan(d, hdims, p.m)) return rc;
  if (Np < 1 || Np > ((int64_t)1 << 30) || (n_in != 2 && n_in != 3)) return SOCMX_E_DIM;
  p.ntiles = (int)((Np + 15) / 16);
  int wsum = 0;
  for (int t = 0; t < MT_N; ++t) wsum += p.m.wid[t];
  p.ws_floats = (int64_t)p.ntiles * 32 * wsum;
  const int fin[3] = {n_in, hdims[0], hdims[1]}, fout[3] = {hdims[0], hdims[1], d * d};
  int off = 0;
  int64_t goff = 0;
  for (int l = 0; l < 3; ++l) {
    p.fin[l] = fin[l]; p.fout[l] = fout[l];
    p.OB[l] = p.m.L[l].out_pad >> 4; p.IB[l] = p.m.L[l].in_pad >> 4;
    p.w_cell_off[l] = off; off += p.OB[l] * p.IB[l] * 256;
    p.gw_off[l] = goff; goff += (int64_t)fin[l] * fout[l];
    p.gb_off[l] = goff; goff += fout[l];
  }
  p.grad_floats = goff;
  p.total_cells_floats = off;
  for (int l = 0; l < 3; ++l) { p.b_part_off[l] = off; off += p.m.L[l].out_pad; }
  p.total_bias = off - p.total_cells_floats;
  p.slab_floats = off;
  const int nt16 = 2 * p.ntiles;
  // (wide form: the last layer's partials are 2 MB per slab at d = 64 -- 32 slabs keep kernel C's reduction at 70 MB)
  // (developer A/B switch, read once: SOCMX_K3_SLABS)
  static const int env_slabs = [] { const char* e = getenv("SOCMX_K3_SLABS"); return e ? atoi(e) : 0; }();
  // (96 slabs once there are enough tiles: this kernel runs on the second stream beside the rollout, i.e. on the half of the chip
  //  the one-row workgroups leave free -- 240 workgroups of shorter waves fill it in one round: 80 -> 61 us at configs[2])
  int S = p.m.wide ? 32 : (env_slabs > 0 ? env_slabs : (nt16 >= 1536 ? 96 : 64));
  if (S > nt16 / 4) S = nt16 / 4 >= 32 ? 32 : nt16 / 4;
  p.S = S < 1 ? 1 : S;
  p.part_floats = (int64_t)p.S * p.slab_floats;
  return 0;
}

extern "C" int socmx_mnet_backward_sizes(int32_t d, const int32_t hdims[2], int32_t n_in, int64_t Np,
                                         int64_t* workspace_floats, int64_t* grad_floats) {
  MnetBwdPlan p;
  if (const int rc = mnet_bwd_plan(d, hdims, n_in, Np, p)) return rc;
  if (workspace_floats) *workspace_floats = p.ws_floats + p.part_floats;
  if (grad_floats) *grad_floats = p.grad_floats;
  return 0;
}

extern "C" int socmx_mnet_backward_f32(const float* packed, int32_t d, const int32_t hdims[2], int32_t n_in,
                                       const float* t, const float* s, const float* z, int64_t Np, const float* gnet,
                                       const float* gdnet, float* workspace, float* grads, socmx_stream_t stream) {
  if (!packed || !t || !s || !gnet || !gdnet || !workspace || !grads || (n_in == 3 && !z)) return SOCMX_E_NULL;
  MnetBwdPlan p;
  if (const int rc = mnet_bwd_plan(d, hdims, n_in, Np, p)) return rc;
  MArgs a{};
  a.m = p.m; a.packed = packed; a.packedT = packed + p.m.total_floats; a.t = t; a.s = s; a.z = z; a.Np = Np;
  a.ntiles = p.ntiles;
  a.gnet = gnet; a.gdnet = gdnet; a.ws = workspace;
  const size_t lds_bytes = (size_t)p.m.lds_floats * sizeof(float);
  const int nlay_b = p.m.wide ? 2 : 3;                    // layers whose weight gradient kernel B forms from the slabs
  if (p.m.wide) {
    const int nob = p.m.h1p >> 4;
    const bool gen = (p.m.d2 & 15) != 0 || nob > 8;
    void (*kern)(const MArgs) = gen ? mnet_backward_wide_kernel<kK3WideWaves, 8, true> : mnet_backward_wide_kernel<kK3WideWaves, 8, false>;
    if (nob <= 2) kern = gen ? mnet_backward_wide_kernel<kK3WideWaves, 2, true> : mnet_backward_wide_kernel<kK3WideWaves, 2, false>;
    else if (nob <= 4) kern = gen ? mnet_backward_wide_kernel<kK3WideWaves, 4, true> : mnet_backward_wide_kernel<kK3WideWaves, 4, false>;
    if (const int err = ensure_max_lds(kern)) return err;
    if (const int err = launch(kern, dim3(p.ntiles), dim3(kK3WideWaves * 64), lds_bytes, stream, a)) return err;
    // the last layer's weight / bias gradient partials (its own kernel: A operand transposed through LDS)
    MWgradWideArgs ga;
    ga.gnet = gnet; ga.gdnet = gdnet;
    ga.h2slab = workspace + (size_t)p.ntiles * 32 * p.m.pre[MT_H2];
    ga.part = workspace + p.ws_floats;
    ga.Np = Np; ga.slab_floats = p.slab_floats; ga.d2 = p.m.d2; ga.h1p = p.m.h1p; ga.ntiles = p.ntiles; ga.S = p.S;
    ga.NG = (p.m.d2 + 63) / 64; ga.IB = p.IB[2]; ga.cell_off = p.w_cell_off[2]; ga.bias_off = p.b_part_off[2];
    const unsigned wgrid = (unsigned)(ga.NG * 8 * ((p.S + 7) / 8));
    const bool rag = (p.m.d2 & 3) != 0;
    const int err = ga.IB > 8 ? (rag ? launch(mnet_wgrad_wide_kernel<4, true>, dim3(wgrid), dim3(256), 0, stream, ga)
                                     : launch(mnet_wgrad_wide_kernel<4, false>, dim3(wgrid), dim3(256), 0, stream, ga))
                  : ga.IB > 4 ? (rag ? launch(mnet_wgrad_wide_kernel<2, true>, dim3(wgrid), dim3(256), 0, stream, ga)
                                     : launch(mnet_wgrad_wide_kernel<2, false>, dim3(wgrid), dim3(256), 0, stream, ga))
                              : (rag ? launch(mnet_wgrad_wide_kernel<1, true>, dim3(wgrid), dim3(256), 0, stream, ga)
                                     : launch(mnet_wgrad_wide_kernel<1, false>, dim3(wgrid), dim3(256), 0, stream, ga));
    if (err) return err;
  } else {
    const int tile_env = mnet_tile_env();
    if (mres_ok(p.m) && !tile_env) {
      if (const int err = ensure_max_lds(mnet_backward_resident_kernel)) return err;
      static const int bgrid_env = [] { const char* e = getenv("SOCMX_K3_BGRID"); return e ? atoi(e) : 0; }();
      if (const int err = launch(mnet_backward_resident_kernel, dim3(std::min(p.ntiles, bgrid_env > 0 ? bgrid_env : 256)), dim3(512),
                                 (size_t)mres_bwd_lds_floats(p.m) * sizeof(float), stream, a)) return err;
    } else {
      if (const int err = ensure_max_lds(mnet_backward_kernel<kK2Waves>)) return err;
      if (const int err = launch(mnet_backward_kernel<kK2Waves>, dim3(p.ntiles), dim3(kK2Waves * 64), lds_bytes, stream, a)) return err;
    }
  }
  // weight / bias gradient partials: kernel B over the 2 ntiles slab tiles (value, tangent alternating)
  WgradArgs wa{};
  wa.S = p.S; wa.ntiles = 2 * p.ntiles; wa.slab_floats = p.slab_floats; wa.bias_even_tiles = 1;
  wa.ws = workspace; wa.part = workspace + p.ws_floats;
  const int gt[3] = {MT_GZ1, MT_GZ2, MT_GOUT}, at[3] = {MT_X, MT_H1, MT_H2};
  wa.n_items = 0;
  for (int l = 0; l < 9; ++l) {
    if (l < nlay_b) {
      wa.gt_off[l] = p.m.pre[gt[l]]; wa.at_off[l] = p.m.pre[at[l]]; wa.gW[l] = p.m.wid[gt[l]]; wa.aW[l] = p.m.wid[at[l]];
      wa.OB[l] = p.OB[l]; wa.IB[l] = p.IB[l]; wa.w_cell_off[l] = p.w_cell_off[l]; wa.b_part_off[l] = p.b_part_off[l];
      wa.item0[l] = wa.n_items;
      wa.n_items += ((p.OB[l] + 3) / 4) * ((p.IB[l] + 3) / 4);
    } else {
      wa.gt_off[l] = wa.at_off[l] = 0; wa.gW[l] = wa.aW[l] = 16; wa.OB[l] = wa.IB[l] = 1;
      wa.w_cell_off[l] = p.total_cells_floats; wa.b_part_off[l] = (int)p.slab_floats; wa.item0[l] = 1 << 30;
    }
  }
  wa.item0[9] = wa.n_items;
  wa.sched.nslots = 0;                                      // (uniform grid: the wide form's last layer has its own kernel)
  const int slab_groups = (((p.S + 3) / 4) + 7) & ~7;
  if (const int err = launch(unet_wgrad_kernel, dim3(wa.n_items * slab_groups), dim3(256), 0, stream, wa)) return err;
  FinishArgs fa{};
  for (int l = 0; l < 9; ++l) {
    if (l < 3) {
      fa.w_cell_off[l] = p.w_cell_off[l]; fa.b_part_off[l] = p.b_part_off[l]; fa.OB[l] = p.OB[l]; fa.IB[l] = p.IB[l];
      fa.fin[l] = p.fin[l]; fa.fout[l] = p.fout[l]; fa.gw_off[l] = p.gw_off[l]; fa.gb_off[l] = p.gb_off[l];
    } else {
      fa.w_cell_off[l] = 1 << 30; fa.b_part_off[l] = 1 << 30; fa.OB[l] = fa.IB[l] = 1; fa.fin[l] = fa.fout[l] = 0;
      fa.gw_off[l] = fa.gb_off[l] = 0;
    }
  }
  fa.total_cells_floats = p.total_cells_floats; fa.total_bias = p.total_bias; fa.S = p.S; fa.slab_floats = p.slab_floats;
  fa.part = wa.part; fa.grads = grads;
  const int nthreads = p.total_cells_floats + p.total_bias;
  return launch(unet_wgrad_finish_kernel, dim3((nthreads + 255) / 256), dim3(256), 0, stream, fa);
}
