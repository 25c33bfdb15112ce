// socmx_mlp.h -- the control network (FullyConnectedUNet, reference models.py:202-242) as a weight-STREAMING
// engine on gfx950: fp32 MFMA stages over one batch tile that lives in LDS, fed by a continuous per-wave
// stream of 1-KiB weight fragments from L2.
//
// Two tile shapes share all the code below:
//   Tile16: 16 batch rows per workgroup, v_mfma_f32_16x16x4_f32.  Fragment = 16 neurons x 16 inputs.
//           MFMA-bound (32 cycles/SIMD per 1024 MACs).  For large batches (throughput).
//   Tile4 :  4 batch rows per workgroup, v_mfma_f32_4x4x1_16b_f32.  Fragment = 64 neurons x 4 inputs.
//           4x more workgroups for the same batch: for B ~ 128 the chip is latency-bound on a K-long
//           dependency chain and only B/rows CUs can work at all (small-batch / latency path).
// In both, one fragment is one contiguous 1-KiB wave load (16 B per lane) that feeds four MFMAs, the activation
// operand is one 16-byte LDS read per fragment row-group, and the result lands as 4 consecutive neurons per
// lane (one 16-byte LDS write).
//
// Streaming: per time step every wave consumes a FIXED sequence of fragments (its share of the 9 GEMMs).  The
// sequence is described by a per-wave segment list (built once per launch, in LDS).  A 4-slot register ring
// holds the next four fragments; slot q is refilled with the fragment four positions ahead right after its
// MFMAs issue -- across GEMM, stage, barrier and time-step boundaries alike (segment lengths are padded to
// multiples of the ring depth, the pad fragments are loaded but not multiplied).  The L2 round trip therefore never sits
// on the stage-to-stage critical path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace socmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Tile16 {
  static constexpr int RT = 16;   // batch rows per tile
  static constexpr int NPB = 16;  // neurons per block (per fragment)
  static constexpr int IPC = 16;  // inputs per chunk (per fragment)
  static constexpr int RD = 8;    // ring depth: fragments in flight per wave (multiple of 4)
};
struct Tile4 {
  static constexpr int RT = 4;
  static constexpr int NPB = 64;
  static constexpr int IPC = 4;
  static constexpr int RD = 4;
};

__host__ __device__ inline int pad_to(int x, int m) { return (x + m - 1) / m * m; }

struct LayerDesc {
  int w_off;    // float offset of the fragment-ordered weights inside this tile shape's packed image
  int b_off;    // float offset of the (padded) bias
  int in_pad;   // fan-in padded to IPC
  int out_pad;  // fan-out padded to NPB
  int b_lds;    // float offset of this layer's bias inside the LDS bias copy
};

struct UnetDesc {
  int d, in0, in0p, outp;  // state dim, d+1, padded input width, padded output width
  int h[3], hp_in[3], hp_out[3];
  LayerDesc L[9];
  int total_floats;  // size of this tile shape's image
  int bias_floats;
};

// layer (fan_in, fan_out) in SOCMX_L_* order (models.py:212-228)
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

// Activation widths must satisfy both roles (a layer's output feeds the next layer's input), so every width
// is padded to lcm(NPB, IPC) = NPB.
template <class T>
inline UnetDesc make_unet_desc(int d, const int h[3]) {
  UnetDesc u;
  u.d = d; u.in0 = d + 1; u.in0p = pad_to(d + 1, T::IPC); u.outp = pad_to(d, T::NPB);
  for (int i = 0; i < 3; ++i) { u.h[i] = h[i]; u.hp_out[i] = pad_to(h[i], T::NPB); u.hp_in[i] = u.hp_out[i]; }
  int fin[9], fout[9];
  unet_layer_dims(d, h, fin, fout);
  int off = 0, boff = 0;
  for (int l = 0; l < 9; ++l) {
    // fan-in of a layer = padded width of the activation it reads
    const int inw = (l == 0 || l == 3) ? u.in0p : pad_to(fin[l], T::NPB);   // down_0 / res_0 read [t, x]
    u.L[l].in_pad = inw;
    u.L[l].out_pad = pad_to(fout[l], T::NPB);
    u.L[l].w_off = off; off += u.L[l].in_pad * u.L[l].out_pad;
    u.L[l].b_off = off; off += u.L[l].out_pad;
    u.L[l].b_lds = boff; boff += u.L[l].out_pad;
  }
  u.total_floats = off;
  u.bias_floats = boff;
  return u;
}

// Fragment order of a layer: packed[w_off + ((nb*KC + kc)*64 + lane)*4 + i] = W[n][k] with
//   Tile16: n = nb*16 + (lane & 15), k = kc*16 + 4*(lane >> 4) + i
//   Tile4 : n = nb*64 + lane,        k = kc*4 + i
template <class T>
__host__ __device__ inline void fragment_coords(int nb, int kc, int lane, int i, int* n, int* k) {
  if (T::RT == 16) { *n = nb * 16 + (lane & 15); *k = kc * 16 + 4 * (lane >> 4) + i; }
  else { *n = nb * 64 + lane; *k = kc * 4 + i; }
}

// LDS tile: RT rows; row strides are width+4 floats (16-byte aligned rows, bank spread).
struct TileLayout {
  int s0, s1, s2, s3, sg;                    // strides of X0, R1/O1, R2/O2, R3, GV
  int x0, r1, r2, r3, o2, o1, gv, scratch;   // float offsets
  int bias;                                  // LDS copy of all (padded) biases
  int segs;                                  // per-wave segment lists (ints)
  int floats;                                // total
};

constexpr int kSegInts = 16;     // ints per segment descriptor
constexpr int kSegsPerWave = 12; // 6 stages x 2 GEMMs

template <class T>
__host__ __device__ inline TileLayout make_tile_layout(const UnetDesc& u, int nwaves) {
  TileLayout t;
  const int RT = T::RT;
  t.s0 = u.in0p + 4; t.s1 = u.hp_out[0] + 4; t.s2 = u.hp_out[1] + 4; t.s3 = u.hp_out[2] + 4; t.sg = u.outp + 4;
  int off = 0;
  t.x0 = off; off += RT * t.s0;
  t.r1 = off; off += RT * t.s1;
  t.r2 = off; off += RT * t.s2;
  t.r3 = off; off += RT * t.s3;
  t.o2 = off; off += RT * t.s2;
  t.o1 = off; off += RT * t.s1;
  t.gv = off; off += RT * t.sg;
  // split-K partials: 2 GEMMs x parts x RT rows x out_pad, with parts*NBLK <= nwaves  =>  parts*out_pad <= NPB*nwaves
  t.scratch = off; off += 2 * RT * T::NPB * nwaves;
  t.bias = off; off += u.bias_floats;
  t.segs = off; off += nwaves * kSegsPerWave * kSegInts;
  t.floats = off;
  return t;
}

// One network stage: Y = relu(L1 . X1 + b1) [+ L2 . X2 + b2].  LDS operands as float offsets.
struct StageDesc {
  LayerDesc L1, L2;
  int x1, s1, x2, s2, y, sy, has2;
};

struct UnetProgram {
  StageDesc st[6];
};

inline UnetProgram make_unet_program(const UnetDesc& u, const TileLayout& t) {
  UnetProgram p;
  const LayerDesc* L = u.L;
  auto set = [&](int i, int l1, int x1, int s1, int has2, int l2, int x2, int s2, int y, int sy) {
    StageDesc& d = p.st[i];
    d.L1 = L[l1]; d.L2 = L[l2];
    d.x1 = x1; d.s1 = s1; d.x2 = x2; d.s2 = s2; d.y = y; d.sy = sy; d.has2 = has2;
  };
  set(0, 0, t.x0, t.s0, 0, 0, t.x0, t.s0, t.r1, t.s1);   // r1 = relu(down_0 x)
  set(1, 1, t.r1, t.s1, 0, 1, t.r1, t.s1, t.r2, t.s2);   // r2 = relu(down_1 r1)
  set(2, 2, t.r2, t.s2, 0, 2, t.r2, t.s2, t.r3, t.s3);   // r3 = relu(down_2 r2)
  set(3, 6, t.r3, t.s3, 1, 5, t.r2, t.s2, t.o2, t.s2);   // o2 = relu(up_2 r3) + res_2 r2
  set(4, 7, t.o2, t.s2, 1, 4, t.r1, t.s1, t.o1, t.s1);   // o1 = relu(up_1 o2) + res_1 r1
  set(5, 8, t.o1, t.s1, 1, 3, t.x0, t.s0, t.gv, t.sg);   // o0 = relu(up_0 o1) + res_0 x
  return p;
}

// Largest neuron-block count a wave may own in a direct stage (accumulators acc[4]).
constexpr int kMaxBlocksPerWave = 4;

template <class T>
inline bool unet_fits(const UnetDesc& u, int nwaves) {
  for (int l = 0; l < 9; ++l) {
    const int nblk = u.L[l].out_pad / T::NPB;
    if (nblk > kMaxBlocksPerWave * nwaves) return false;
  }
  return true;
}

#if defined(__HIPCC__)

__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

#define SOCMX_PIN(x) asm volatile("" : "+s"(x))

// ---- segment descriptors (LDS, ints) --------------------------------------------------------------------
// One per (wave, stage, GEMM).  Fragment f of a segment (consumption order): kc = kc0 + (f >> nbl2),
// block = blk0 + (f & (NB-1)) * bstride; address = wbase4 + (block*KC + kc)*64 (+ lane) float4s.
enum {
  SEG_WBASE4 = 0, SEG_KC, SEG_BLK0, SEG_BSTRIDE, SEG_NBL2, SEG_KC0, SEG_KC1, SEG_NBODY, SEG_NREAL, SEG_NEXT,
  SEG_CNT,   // direct: blocks owned (<= NB); split: 1
  SEG_SPLIT, // 1 if this stage runs split-K
  SEG_PART, SEG_PARTS, SEG_ACTIVE, SEG_PAD
};

struct Seg {  // SGPR-resident copy
  int wbase4, KC, blk0, bstride, nbl2, kc0, kc1, nbody, nreal, next, cnt, split, part, parts, active;
};

__device__ __forceinline__ Seg load_seg(const int* lds_seg) {
  // every lane reads the same words (LDS broadcast); readfirstlane moves them to SGPRs
  const int4 a = *reinterpret_cast<const int4*>(lds_seg);
  const int4 b = *reinterpret_cast<const int4*>(lds_seg + 4);
  const int4 c = *reinterpret_cast<const int4*>(lds_seg + 8);
  const int4 d = *reinterpret_cast<const int4*>(lds_seg + 12);
  Seg s;
#define RFL(x) __builtin_amdgcn_readfirstlane(x)
  s.wbase4 = RFL(a.x); s.KC = RFL(a.y); s.blk0 = RFL(a.z); s.bstride = RFL(a.w);
  s.nbl2 = RFL(b.x); s.kc0 = RFL(b.y); s.kc1 = RFL(b.z); s.nbody = RFL(b.w);
  s.nreal = RFL(c.x); s.next = RFL(c.y); s.cnt = RFL(c.z); s.split = RFL(c.w);
  s.part = RFL(d.x); s.parts = RFL(d.y); s.active = RFL(d.z);
#undef RFL
  return s;
}

// Built once per launch: lane s (< 12) of every wave fills segment s of its own list; lane 0 then links the
// non-empty segments into a ring (SEG_NEXT).  Integer divisions happen here and nowhere in the time loop.
template <class T, int NW>
__device__ __forceinline__ void build_segments(const UnetProgram& prog, int* segs_lds) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int* mine = segs_lds + wave * kSegsPerWave * kSegInts;
  if (lane < kSegsPerWave) {
    const int si = lane >> 1, g = lane & 1;
    const StageDesc& sd = prog.st[si];
    const LayerDesc& L = g ? sd.L2 : sd.L1;
    const int NBLK = sd.L1.out_pad / T::NPB;   // both GEMMs of a stage share the output width
    const int KC = L.in_pad / T::IPC;
    int blk0 = 0, bstride = 0, nbl2 = 0, kc0 = 0, kc1 = 0, cnt = 0, split = 0, part = 0, parts = 1, active = 0;
    if (NBLK >= NW) {
      cnt = wave < NBLK ? (NBLK - wave + NW - 1) / NW : 0;
      blk0 = wave; bstride = NW;
      nbl2 = cnt > 2 ? 2 : (cnt > 1 ? 1 : 0);
      kc0 = 0; kc1 = cnt > 0 ? KC : 0;
      active = cnt > 0;
    } else {
      split = 1; parts = NW / NBLK; part = wave / NBLK; blk0 = wave % NBLK; cnt = 1;
      active = part < parts;
      if (active) { kc0 = (part * KC) / parts; kc1 = ((part + 1) * KC) / parts; }
    }
    if (g && !sd.has2) { kc0 = kc1 = 0; }
    const int nreal = (kc1 - kc0) << nbl2;
    int* s = mine + lane * kSegInts;
    s[SEG_WBASE4] = L.w_off >> 2; s[SEG_KC] = KC; s[SEG_BLK0] = blk0; s[SEG_BSTRIDE] = bstride; s[SEG_NBL2] = nbl2;
    s[SEG_KC0] = kc0; s[SEG_KC1] = kc1; s[SEG_NBODY] = (nreal + T::RD - 1) / T::RD; s[SEG_NREAL] = nreal; s[SEG_NEXT] = -1;
    s[SEG_CNT] = cnt; s[SEG_SPLIT] = split; s[SEG_PART] = part; s[SEG_PARTS] = parts; s[SEG_ACTIVE] = active;
    s[SEG_PAD] = 0;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  if (lane == 0) {
    int first = -1;
    for (int i = kSegsPerWave - 1; i >= 0; --i)
      if (mine[i * kSegInts + SEG_NBODY] > 0) first = i;
    int nxt = first;  // the stream wraps around: the last non-empty segment is followed by the first one
    for (int i = kSegsPerWave - 1; i >= 0; --i) {
      mine[i * kSegInts + SEG_NEXT] = nxt;
      if (mine[i * kSegInts + SEG_NBODY] > 0) nxt = i;
    }
  }
}

// ---- MFMA + fragment geometry ---------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
  if (T::RT == 16) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], b[3], c, 0, 0, 0);
  }
  return c;
}

// this lane's offset inside an activation tile for chunk 0 (float index), and inside an output tile for block 0
template <class T>
__device__ __forceinline__ int act_lane_off(int lane, int S) {
  return T::RT == 16 ? (lane & 15) * S + 4 * (lane >> 4) : (lane & 3) * S;
}
template <class T>
__device__ __forceinline__ int out_lane_off(int lane, int S) {
  return T::RT == 16 ? (lane & 15) * S + 4 * (lane >> 4) : (lane & 3) * S + 4 * (lane >> 2);
}

template <int RD>
struct Ring {
  f32x4 slot[RD];
};

// address (float4 index, without the lane term) of fragment f of segment s; pad fragments repeat the last one
__device__ __forceinline__ int frag_index(const Seg& s, int f) {
  const int nb_mask = (1 << s.nbl2) - 1;
  int kc = s.kc0 + (f >> s.nbl2);
  kc = kc < s.kc1 - 1 ? kc : s.kc1 - 1;
  int j = f & nb_mask;
  j = j < s.cnt - 1 ? j : s.cnt - 1;          // a 3-block group runs as NB=4 with the last block repeated
  return s.wbase4 + ((s.blk0 + j * s.bstride) * s.KC + kc) * 64;
}

// Consume one segment (NB = 1 << NBL2 interleaved blocks) from the ring, refilling it with the fragments of the
// next body iteration -- of this segment, or of segment `nxt` (which may belong to the next stage / time step).
template <class T, int NBL2>
__device__ __forceinline__ void run_segment(f32x4 (&acc)[4], Ring<T::RD>& ring, const Seg& cur, const Seg& nxt,
                                            const f32x4* __restrict__ wl /* packed image + lane */, const float* X,
                                            int S, int lane) {
  constexpr int NB = 1 << NBL2;
  const float* xl = X + act_lane_off<T>(lane, S);
  for (int it = 0; it < cur.nbody; ++it) {
    const bool last = (it + 1 == cur.nbody);
    const int fit = last ? 0 : T::RD * (it + 1);
#pragma unroll
    for (int q = 0; q < T::RD; ++q) {
      const int f = T::RD * it + q;
      if (f < cur.nreal) {
        const int kc = cur.kc0 + (f >> NBL2);
        const f32x4 bx = *reinterpret_cast<const f32x4*>(xl + kc * T::IPC);
        acc[q & (NB - 1)] = mfma4<T>(ring.slot[q], bx, acc[q & (NB - 1)]);
      }
      const int idx = last ? frag_index(nxt, fit + q) : frag_index(cur, fit + q);
      ring.slot[q] = wl[idx];
    }
  }
}

// Y = relu(W1.X1 + b1) [+ W2.X2 + b2] for the tile; all NW waves call it.  Ends with a workgroup barrier.
// `cur1`/`cur2` are this wave's segments of the stage; `nxt1` follows cur1 in the wave's stream, `nxt2` follows cur2.
template <class T, int NW, typename Hook>
__device__ __forceinline__ void unet_stage(const f32x4* __restrict__ wl, const float* bias_lds, const StageDesc& sd,
                                           const Seg& cur1, const Seg& cur2, const Seg& nxt1, const Seg& nxt2,
                                           float* lds, float* scratch, Ring<T::RD>& ring, Hook hook) {
  const int lane = threadIdx.x & 63;
  const float* X1 = lds + sd.x1;
  const float* X2 = lds + sd.x2;
  float* Y = lds + sd.y;
  f32x4 acc[4];
  if (!cur1.split) {
    if (cur1.cnt > 0) {
      const int ob = out_lane_off<T>(lane, 0);   // neuron offset of this lane inside a block
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int jj = j < cur1.cnt ? j : cur1.cnt - 1;
        acc[j] = *reinterpret_cast<const f32x4*>(bias_lds + sd.L1.b_lds + (cur1.blk0 + jj * cur1.bstride) * T::NPB + ob);
      }
      hook(0);
      if (cur1.nbl2 == 0)      run_segment<T, 0>(acc, ring, cur1, nxt1, wl, X1, sd.s1, lane);
      else if (cur1.nbl2 == 1) run_segment<T, 1>(acc, ring, cur1, nxt1, wl, X1, sd.s1, lane);
      else                     run_segment<T, 2>(acc, ring, cur1, nxt1, wl, X1, sd.s1, lane);
      hook(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = relu_keep_nan(acc[j][r]);
      if (sd.has2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int jj = j < cur1.cnt ? j : cur1.cnt - 1;
          acc[j] += *reinterpret_cast<const f32x4*>(bias_lds + sd.L2.b_lds + (cur1.blk0 + jj * cur1.bstride) * T::NPB + ob);
        }
        if (cur2.nbl2 == 0)      run_segment<T, 0>(acc, ring, cur2, nxt2, wl, X2, sd.s2, lane);
        else if (cur2.nbl2 == 1) run_segment<T, 1>(acc, ring, cur2, nxt2, wl, X2, sd.s2, lane);
        else                     run_segment<T, 2>(acc, ring, cur2, nxt2, wl, X2, sd.s2, lane);
      }
      hook(2);
      float* yl = Y + out_lane_off<T>(lane, sd.sy);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < cur1.cnt) *reinterpret_cast<f32x4*>(yl + (cur1.blk0 + j * cur1.bstride) * T::NPB) = acc[j];
    }
    hook(3);
    __syncthreads();
    hook(4);
  } else {
    // fewer neuron blocks than waves: split the reduction (K) dimension across waves, partial sums through
    // LDS, bias + ReLU applied after the combine.
    const int outp = sd.L1.out_pad;
    float* P1 = scratch;
    float* P2 = scratch + cur1.parts * T::RT * outp;
    if (cur1.active) {
      const int po = cur1.part * T::RT * outp + out_lane_off<T>(lane, outp) + cur1.blk0 * T::NPB;
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = acc[0]; acc[2] = acc[0]; acc[3] = acc[0];
      hook(0);
      run_segment<T, 0>(acc, ring, cur1, nxt1, wl, X1, sd.s1, lane);
      *reinterpret_cast<f32x4*>(P1 + po) = acc[0];
      hook(1);
      if (sd.has2) {
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        run_segment<T, 0>(acc, ring, cur2, nxt2, wl, X2, sd.s2, lane);
        *reinterpret_cast<f32x4*>(P2 + po) = acc[0];
      }
      hook(2);
    }
    __syncthreads();
    hook(3);
    const float inv_outp = __builtin_amdgcn_rcpf((float)outp);
    for (int e = threadIdx.x; e < T::RT * outp; e += NW * 64) {
      const int r = (int)(((float)e + 0.5f) * inv_outp), n = e - r * outp;  // e / outp without an integer divide
      float v = bias_lds[sd.L1.b_lds + n];
      for (int p = 0; p < cur1.parts; ++p) v += P1[(p * T::RT + r) * outp + n];
      v = relu_keep_nan(v);
      if (sd.has2) {
        float v2 = bias_lds[sd.L2.b_lds + n];
        for (int p = 0; p < cur1.parts; ++p) v2 += P2[(p * T::RT + r) * outp + n];
        v += v2;
      }
      Y[r * sd.sy + n] = v;
    }
    __syncthreads();
    hook(4);
  }
}

// all threads: copy the padded biases of the nine layers from the packed image into LDS (once per kernel)
__device__ __forceinline__ void unet_load_biases(const float* __restrict__ Wp, const UnetDesc& u, const TileLayout& t,
                                                 float* lds, int tid, int nthr) {
  for (int l = 0; l < 9; ++l)
    for (int e = tid; e < u.L[l].out_pad; e += nthr) lds[t.bias + u.L[l].b_lds + e] = Wp[u.L[l].b_off + e];
}

// State carried across stages and time steps: the ring and the stream position.
template <class T>
struct UnetStream {
  Ring<T::RD> ring;
  int first;  // index of this wave's first non-empty segment, -1 if it has no work at all
};

// After build_segments + a workgroup barrier: start the stream (load the first four fragments).
template <class T>
__device__ __forceinline__ void unet_stream_init(UnetStream<T>& st, const f32x4* __restrict__ wl, const int* segs_lds) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int* mine = segs_lds + wave * kSegsPerWave * kSegInts;
  const Seg last = load_seg(mine + (kSegsPerWave - 1) * kSegInts);
  // SEG_NEXT of the last slot points at the first non-empty segment (wrap-around); if that slot itself is
  // non-empty its NEXT still names the first one.
  st.first = last.next;
  if (st.first >= 0) {
    const Seg s0 = load_seg(mine + st.first * kSegInts);
#pragma unroll
    for (int q = 0; q < T::RD; ++q) st.ring.slot[q] = wl[frag_index(s0, q)];
  }
}

// The whole network on the tile: X0 (already filled, [t, x, 0-pad]) -> GV (nabla_V, first d columns valid).
// hook(i) is called after stage i = 1..6; hook(16 + 8*s + k) inside stage s (diagnostics).
template <class T, int NW, typename Hook>
__device__ __forceinline__ void unet_tile_forward(const f32x4* __restrict__ wl, const UnetProgram& prog,
                                                  const TileLayout& t, float* lds, UnetStream<T>& st, Hook hook) {
  float* SC = lds + t.scratch;
  const float* BL = lds + t.bias;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int* mine = reinterpret_cast<const int*>(lds + t.segs) + wave * kSegsPerWave * kSegInts;
#pragma unroll 1
  for (int si = 0; si < 6; ++si) {
    StageDesc sd = prog.st[si];
    SOCMX_PIN(sd.L1.b_lds); SOCMX_PIN(sd.L1.out_pad); SOCMX_PIN(sd.L2.b_lds);
    SOCMX_PIN(sd.x1); SOCMX_PIN(sd.s1); SOCMX_PIN(sd.x2); SOCMX_PIN(sd.s2); SOCMX_PIN(sd.y); SOCMX_PIN(sd.sy);
    SOCMX_PIN(sd.has2);
    const Seg c1 = load_seg(mine + (2 * si) * kSegInts);
    const Seg c2 = load_seg(mine + (2 * si + 1) * kSegInts);
    // successor of c1 in the stream: c2 if it has work, else whatever follows; successor of c2: its NEXT
    const Seg n1 = load_seg(mine + (c1.next >= 0 ? c1.next : 0) * kSegInts);
    const Seg n2 = load_seg(mine + (c2.next >= 0 ? c2.next : 0) * kSegInts);
    unet_stage<T, NW>(wl, BL, sd, c1, c2, n1, n2, lds, SC, st.ring, [&](int sub) { hook(16 + si * 8 + sub); });
    hook(si + 1);
  }
}

#endif  // __HIPCC__
}  // namespace socmx
