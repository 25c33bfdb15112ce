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

namespace socmx {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LayerDesc {
  int w_off;   // float offset of the fragment-ordered weights inside the packed image
  int b_off;   // float offset of the (padded) bias
  int in_pad;  // padded fan-in  (multiple of 16)
  int out_pad; // padded fan-out (multiple of 16)
};

struct UnetDesc {
  int d;      // state dimension
  int in0;    // d + 1
  int in0p;   // pad16(d+1)
  int outp;   // pad16(d)
  int h[3];   // hdims
  int hp[3];  // padded hdims
  LayerDesc L[9];
  int total_floats;
};

__host__ __device__ inline int pad16(int x) { return (x + 15) & ~15; }

// layer (fan_in, fan_out) in SOCMX_L_* order
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

inline UnetDesc make_unet_desc(int d, const int h[3]) {
  UnetDesc u;
  u.d = d; u.in0 = d + 1; u.in0p = pad16(d + 1); u.outp = pad16(d);
  for (int i = 0; i < 3; ++i) { u.h[i] = h[i]; u.hp[i] = pad16(h[i]); }
  int fin[9], fout[9];
  unet_layer_dims(d, h, fin, fout);
  int off = 0;
  for (int l = 0; l < 9; ++l) {
    u.L[l].in_pad = pad16(fin[l]);
    u.L[l].out_pad = pad16(fout[l]);
    u.L[l].w_off = off; off += u.L[l].in_pad * u.L[l].out_pad;
    u.L[l].b_off = off; off += u.L[l].out_pad;
  }
  u.total_floats = off;
  return u;
}

// LDS tile of one 16-row block.  Row strides are width+4 floats (16-byte aligned rows,
// conflict-free 16-byte writes, <=2-way on the fragment reads).
struct TileLayout {
  int s0, s1, s2, s3, sg;                    // strides of X0, R1/O1, R2/O2, R3, GV
  int x0, r1, r2, r3, o2, o1, gv, scratch;   // float offsets
  int floats;                                // total
};

__host__ __device__ inline TileLayout make_tile_layout(const UnetDesc& u, int nwaves) {
  TileLayout t;
  t.s0 = u.in0p + 4; t.s1 = u.hp[0] + 4; t.s2 = u.hp[1] + 4; t.s3 = u.hp[2] + 4; t.sg = u.outp + 4;
  int off = 0;
  t.x0 = off; off += 16 * t.s0;
  t.r1 = off; off += 16 * t.s1;
  t.r2 = off; off += 16 * t.s2;
  t.r3 = off; off += 16 * t.s3;
  t.o2 = off; off += 16 * t.s2;
  t.o1 = off; off += 16 * t.s1;
  t.gv = off; off += 16 * t.sg;
  t.scratch = off; off += 2 * 16 * 16 * nwaves;  // split-K partials: 2 GEMMs x (parts*out_pad <= 16*nwaves) x 16 rows
  t.floats = off;
  return t;
}

#if defined(__HIPCC__)

__device__ __forceinline__ float relu_keep_nan(float x) { return x < 0.f ? 0.f : x; }

// acc[j] += W[block blk0 + j*bstride] . X  over input chunks [kc0, kc1)
template <int NB>
__device__ __forceinline__ void gemm_acc(f32x4 (&acc)[NB], const float4* __restrict__ wl, int blk0, int bstride,
                                         int KC, const float* X, int S, int row, int g, int kc0, int kc1) {
  const float* xrow = X + row * S + 4 * g;
#pragma unroll 4
  for (int kc = kc0; kc < kc1; ++kc) {
    const float4 bx = *reinterpret_cast<const float4*>(xrow + kc * 16);
    float4 a[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] = wl[(size_t)((blk0 + j * bstride) * KC + kc) * 64];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, bx.x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, bx.y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, bx.z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, bx.w, acc[j], 0, 0, 0);
    }
  }
}

template <int NB, int NW>
__device__ __forceinline__ void stage_direct(const float* __restrict__ Wp, const LayerDesc& L1, const float* X1, int S1,
                                             bool has2, const LayerDesc& L2, const float* X2, int S2, float* Y,
                                             int SY, int blk0, int lane) {
  const int row = lane & 15, g = lane >> 4;
  f32x4 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float4 b = *reinterpret_cast<const float4*>(Wp + L1.b_off + (blk0 + j * NW) * 16 + 4 * g);
    acc[j] = f32x4{b.x, b.y, b.z, b.w};
  }
  gemm_acc<NB>(acc, reinterpret_cast<const float4*>(Wp + L1.w_off) + lane, blk0, NW, L1.in_pad >> 4, X1, S1, row, g,
               0, L1.in_pad >> 4);
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = relu_keep_nan(acc[j][r]);
  if (has2) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(Wp + L2.b_off + (blk0 + j * NW) * 16 + 4 * g);
      acc[j] += f32x4{b.x, b.y, b.z, b.w};
    }
    gemm_acc<NB>(acc, reinterpret_cast<const float4*>(Wp + L2.w_off) + lane, blk0, NW, L2.in_pad >> 4, X2, S2, row,
                 g, 0, L2.in_pad >> 4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    *reinterpret_cast<float4*>(Y + row * SY + (blk0 + j * NW) * 16 + 4 * g) =
        make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// Y = relu(W1.X1 + b1) [+ W2.X2 + b2]   for the 16-row tile; all NW waves of the workgroup call it.
// Ends with a workgroup barrier (Y visible, inputs free to overwrite).
template <int NW>
__device__ __forceinline__ void unet_stage(const float* __restrict__ Wp, const LayerDesc& L1, const float* X1, int S1,
                                           bool has2, const LayerDesc& L2, const float* X2, int S2, float* Y, int SY,
                                           float* scratch) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NBLK = L1.out_pad >> 4;
  if (NBLK >= NW) {
    for (int blk0 = wave; blk0 < NBLK; blk0 += 4 * NW) {
      const int cnt = (NBLK - blk0 + NW - 1) / NW;  // blocks this wave still owns (wave-uniform)
      if (cnt >= 4)      stage_direct<4, NW>(Wp, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane);
      else if (cnt == 3) stage_direct<3, NW>(Wp, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane);
      else if (cnt == 2) stage_direct<2, NW>(Wp, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane);
      else               stage_direct<1, NW>(Wp, L1, X1, S1, has2, L2, X2, S2, Y, SY, blk0, lane);
    }
    __syncthreads();
  } else {
    // fewer neuron blocks than waves: split the reduction (K) dimension across waves,
    // partial sums through LDS, bias + ReLU applied after the combine.
    const int parts = NW / NBLK;
    const int blk = wave % NBLK, part = wave / NBLK;
    const int outp = L1.out_pad;
    const int row = lane & 15, g = lane >> 4;
    float* P1 = scratch;
    float* P2 = scratch + parts * 16 * outp;
    if (part < parts) {
      {
        const int KC = L1.in_pad >> 4;
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        gemm_acc<1>(acc, reinterpret_cast<const float4*>(Wp + L1.w_off) + lane, blk, 0, KC, X1, S1, row, g,
                    (part * KC) / parts, ((part + 1) * KC) / parts);
        *reinterpret_cast<float4*>(P1 + (part * 16 + row) * outp + blk * 16 + 4 * g) =
            make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
      }
      if (has2) {
        const int KC = L2.in_pad >> 4;
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        gemm_acc<1>(acc, reinterpret_cast<const float4*>(Wp + L2.w_off) + lane, blk, 0, KC, X2, S2, row, g,
                    (part * KC) / parts, ((part + 1) * KC) / parts);
        *reinterpret_cast<float4*>(P2 + (part * 16 + row) * outp + blk * 16 + 4 * g) =
            make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * outp; e += NW * 64) {
      const int r = e / outp, n = e - r * outp;
      float v = Wp[L1.b_off + n];
      for (int p = 0; p < parts; ++p) v += P1[(p * 16 + r) * outp + n];
      v = relu_keep_nan(v);
      if (has2) {
        float v2 = Wp[L2.b_off + n];
        for (int p = 0; p < parts; ++p) v2 += P2[(p * 16 + r) * outp + n];
        v += v2;
      }
      Y[r * SY + n] = v;
    }
    __syncthreads();
  }
}

// The whole network on the tile: X0 (already filled, [t, x, 0-pad]) -> GV (nabla_V, first d columns valid).
template <int NW>
__device__ __forceinline__ void unet_tile_forward(const float* __restrict__ Wp, const UnetDesc& u, const TileLayout& t,
                                                  float* lds) {
  const LayerDesc* L = u.L;
  float* X0 = lds + t.x0; float* R1 = lds + t.r1; float* R2 = lds + t.r2; float* R3 = lds + t.r3;
  float* O2 = lds + t.o2; float* O1 = lds + t.o1; float* GV = lds + t.gv; float* SC = lds + t.scratch;
  unet_stage<NW>(Wp, L[0], X0, t.s0, false, L[0], X0, t.s0, R1, t.s1, SC);        // r1 = relu(down_0 x)
  unet_stage<NW>(Wp, L[1], R1, t.s1, false, L[1], R1, t.s1, R2, t.s2, SC);        // r2 = relu(down_1 r1)
  unet_stage<NW>(Wp, L[2], R2, t.s2, false, L[2], R2, t.s2, R3, t.s3, SC);        // r3 = relu(down_2 r2)
  unet_stage<NW>(Wp, L[6], R3, t.s3, true, L[5], R2, t.s2, O2, t.s2, SC);         // o2 = relu(up_2 r3) + res_2 r2
  unet_stage<NW>(Wp, L[7], O2, t.s2, true, L[4], R1, t.s1, O1, t.s1, SC);         // o1 = relu(up_1 o2) + res_1 r1
  unet_stage<NW>(Wp, L[8], O1, t.s1, true, L[3], X0, t.s0, GV, t.sg, SC);         // o0 = relu(up_0 o1) + res_0 x
}

#endif  // __HIPCC__
}  // namespace socmx
