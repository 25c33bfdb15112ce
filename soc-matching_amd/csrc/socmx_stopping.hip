// socmx_stopping.hip -- the SOCM least-squares target with PER-SAMPLE pair matrices (stopping times), gfx950.
//
// Replaces reference SOC_matching/method.py:484-507, 548-564, 597-613, 649-660, 692-701 for the molecular_dynamics
// setting, where M = TwoBoundarySigmoidMLP(t, s; tau_m) (models.py:278-393) depends on the sample's stopping time:
//     M[p,m]      =              w I +  c0 N0[p] +  c1 N1[p]
//     dM/ds[p,m]  = nan_to_num( dw I + dc0 N0[p] + c0 dN0[p] + dc1 N1[p] + c1 dN1[p] )          (method.py:553-555)
// with N0 / N1 the two network evaluations (third input 0 / 1: socmx_mnet_forward_f32 with n_in = 3), dN their
// s-tangents, and (w, c0, c1) the scalar gates of models.py:341-392 -- functions of (t_p, s_p, tau_m, gamma, gamma2, gamma3):
//     st = s - t,  ratio = (1 - e^{-gamma st}) / (1 - e^{-gamma |tau - t|} + 1e-7),
//     factor1 = [tau - 1e-3 > s] nan_to_num(1 - min(ratio, 1)),   running = [tau > T - 1e-3],   e3 = e^{-gamma3 st},
//     fun2(x) = (1 - e^{-gamma2 x})(e^{-gamma2 x} - e^{-gamma2}),
//     w = running ? e3 : factor1,   c0 = running ? 0 : fun2(factor1),   c1 = running ? 1 - e3 : 0,
// and (dw, dc0, dc1) their s-derivatives (the reference differentiates with functorch.jacrev, method.py:510-515; here the
// closed forms, with torch.minimum's tie rule).  The reference materialises M and dM/ds as (Np, B, d, d) tensors and round 2
// passed eight (Np, B) gate fields computed by torch; now the gates, their s-derivatives and -- in the backward kernel --
// their derivatives in gamma, gamma2, gamma3 are formed per (pair, sample) inside the kernels: nothing of size Np x B exists.
//   target[i,m] = sum_{j>=i} ( M[p,m] qx[j,m] - dM/ds[p,m] vx[j,m] ),   qx = q (j < K) | nabla_g (j = K),  vx = v | 0.
// d <= 16 (the setting uses d = 1 or 2).  Bound: latency / HBM at these sizes (VALU kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/socmx.h"
#include "socmx_launch.h"

namespace socmx {

struct StopArgs {
  int d, K, B;
  int64_t Np;
  const float *pt, *ps;              // (Np,) pair times t_p, s_p (method.py:533-547)
  const float* tau;                  // (B,) stopping times (method.py:525-530)
  const float* gam;                  // (3,) gamma, gamma2, gamma3 in device memory
  float Tm;                          // the model's own T (models.py:287: stays 1.0 whatever cfg.method.T is)
  const float *N0, *N1, *dN0, *dN1;  // (Np, d, d)
  const float *q, *v, *gT;           // (K,B,d), (K,B,d), (B,d)
  float* target;                     // (K+1, B, d)                         forward
  const float* gtarget;              // (K+1, B, d)  d obj / d target        backward
  float *gN0, *gN1, *gdN0, *gdN1;    // (Np, d, d)
  float* ggam_part;                  // (Np, 3) per-pair partial sums of d obj / d (gamma, gamma2, gamma3)
};

__host__ __device__ inline int64_t stop_pair_row(int i, int K) { return (int64_t)i * (K + 1) - (int64_t)i * (i - 1) / 2; }

struct Gates {
  float w, c0, c1, dw, dc0, dc1;
};

// value gates and their s-derivatives; WITH_GRAD: also the partial derivatives in (gamma, gamma2, gamma3)
struct GateGrads {
  float w_g, dw_g, c0_g, dc0_g, c0_g2, dc0_g2;     // stopped branch: d / d gamma, d / d gamma2
  float w_g3, dw_g3, c1_g3, dc1_g3;                // running branch: d / d gamma3
};

template <bool WITH_GRAD>
__device__ __forceinline__ Gates stop_gates(float t, float s, float tau, float g, float g2, float g3, float Tm,
                                            GateGrads* gg) {
  Gates o;
  const float st = s - t;
  const bool running = tau > Tm - 1e-3f;
  const float e3 = expf(-g3 * st);
  if (WITH_GRAD) *gg = GateGrads{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (running) {
    o.w = e3; o.c0 = 0.f; o.c1 = 1.f - e3;
    o.dw = -g3 * e3; o.dc0 = 0.f; o.dc1 = g3 * e3;
    if (WITH_GRAD) {
      gg->w_g3 = -st * e3; gg->c1_g3 = st * e3;
      gg->dw_g3 = -e3 + g3 * st * e3; gg->dc1_g3 = e3 - g3 * st * e3;
    }
    return o;
  }
  const float a = fabsf(tau - t);
  const float E = expf(-g * st), F = expf(-g * a);
  const float num = 1.f - E, den = 1.f - F + 1e-7f;
  const float ratio = num / den;
  const bool nz = tau - 1e-3f > s;
  // torch.minimum(ratio, 1): value, and its derivative in ratio (1 below, 1/2 at the tie, 0 above; NaN propagates to 0 through
  // nan_to_num(.., nan=0))
  const float kap = (nz && ratio < 1.f) ? 1.f : ((nz && ratio == 1.f) ? 0.5f : 0.f);
  float f1 = 1.f - fminf(ratio, 1.f);
  if (!(f1 == f1) || !nz) f1 = 0.f;
  const float dratio = g * E / den;
  const float df1 = -kap * dratio;
  const float u = expf(-g2 * f1), k = expf(-g2);
  const float G1 = -g2 * u * (1.f + k - 2.f * u);                 // fun2'(f1)
  o.w = f1; o.c0 = (1.f - u) * (u - k); o.c1 = 0.f;
  o.dw = df1; o.dc0 = G1 * df1; o.dc1 = 0.f;
  if (WITH_GRAD) {
    const float ratio_g = (st * E * den - num * a * F) / (den * den);
    const float f1_g = -kap * ratio_g;
    const float dratio_g = E * (1.f - g * st) / den - g * E * a * F / (den * den);
    const float df1_g = -kap * dratio_g;
    const float G2 = g2 * g2 * u * (1.f + k - 4.f * u);          // fun2''(f1)
    const float G_g2 = -f1 * u * (1.f + k - 2.f * u) + k * (1.f - u);
    const float G1_g2 = -u * (1.f + k - 2.f * u) - g2 * (-f1 * u * (1.f + k - 2.f * u) + u * (-k + 2.f * f1 * u));
    gg->w_g = f1_g; gg->dw_g = df1_g;
    gg->c0_g = G1 * f1_g; gg->dc0_g = G2 * f1_g * df1 + G1 * df1_g;
    gg->c0_g2 = G_g2; gg->dc0_g2 = G1_g2 * df1;
  }
  return o;
}

// torch.nan_to_num on one entry of dM/ds (method.py:553-555): NaN -> 0, +-inf -> +-FLT_MAX; `fin`: the entry was finite
// (nan_to_num passes gradients through finite entries only)
__device__ __forceinline__ float nan_to_num1(float x, bool& fin) {
  fin = fabsf(x) <= 3.402823466e+38f;             // false for NaN and +-inf
  if (x != x) return 0.f;
  if (!fin) return x > 0.f ? 3.402823466e+38f : -3.402823466e+38f;
  return x;
}

// forward: workgroup = (row i, 64 samples); lane = sample; the pair matrices of a row are read at uniform addresses
template <int D>
__global__ __launch_bounds__(64) void stopping_target_kernel(const StopArgs a) {
  const int K = a.K, B = a.B, d = a.d, i = blockIdx.x;
  const int m = blockIdx.y * 64 + threadIdx.x;
  const bool live = m < B;
  const int mc = live ? m : B - 1;
  const float g = a.gam[0], g2 = a.gam[1], g3 = a.gam[2];
  const float tau = a.tau[mc];
  float tgt[D];
#pragma unroll
  for (int k = 0; k < D; ++k) tgt[k] = 0.f;
  const int64_t prow = stop_pair_row(i, K);
  float qn[D], vn[D];                 // operands of the NEXT pair: requested one iteration ahead
  auto fetch = [&](int j) {
    const bool last = j >= K;
    const int jc = last ? K - 1 : j;
#pragma unroll
    for (int l = 0; l < D; ++l) {
      const bool in = l < d;
      qn[l] = in ? (last ? a.gT[(size_t)mc * d + l] : a.q[((size_t)jc * B + mc) * d + l]) : 0.f;
      vn[l] = (in && !last) ? a.v[((size_t)jc * B + mc) * d + l] : 0.f;
    }
  };
  fetch(i);
  for (int j = i; j <= K; ++j) {
    const int64_t p = prow + (j - i);
    float qx[D], vx[D];
#pragma unroll
    for (int l = 0; l < D; ++l) { qx[l] = qn[l]; vx[l] = vn[l]; }
    if (j < K) fetch(j + 1);
    const Gates gt = stop_gates<false>(a.pt[p], a.ps[p], tau, g, g2, g3, a.Tm, nullptr);
    const float* n0 = a.N0 + p * d * d;     // (uniform addresses: one pair per iteration)
    const float* n1 = a.N1 + p * d * d;
    const float* d0 = a.dN0 + p * d * d;
    const float* d1 = a.dN1 + p * d * d;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      if (k < d) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < D; ++l) {
          if (l < d) {
            const float mkl = gt.c0 * n0[k * d + l] + gt.c1 * n1[k * d + l] + (k == l ? gt.w : 0.f);
            bool fin;
            const float ekl = nan_to_num1(gt.dc0 * n0[k * d + l] + gt.c0 * d0[k * d + l] + gt.dc1 * n1[k * d + l] +
                                          gt.c1 * d1[k * d + l] + (k == l ? gt.dw : 0.f), fin);
            s += mkl * qx[l] - ekl * vx[l];
          }
        }
        tgt[k] += s;
      }
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < d) a.target[((size_t)i * B + m) * d + k] = tgt[k];
  }
}

// backward: workgroup (256 threads) = pair p.  Per chunk of 256 samples: (1) the chunk's vectors gt_i, qx_j, vx_j go to LDS
// (contiguous loads), every thread < chunk forms its sample's gates; (2) thread (entry e = (k, l), slice sl) walks the samples
// sl, sl + nsl, ... and accumulates d obj / d (N0, N1, dN0, dN1)[k][l]; (3) thread = sample again: the gate gradients
// (sums over the entries) chained to gamma, gamma2, gamma3.  Slices are combined in a fixed order (deterministic).
constexpr int kStopThreads = 256;

template <int D>
__global__ __launch_bounds__(kStopThreads) void stopping_target_bwd_kernel(const StopArgs a) {
  constexpr int E = D * D, NSL = kStopThreads / E;
  __shared__ float Nm[4][E];                         // N0, N1, dN0, dN1 of this pair (zero-padded to D x D)
  __shared__ float Vg[kStopThreads][D], Vq[kStopThreads][D], Vv[kStopThreads][D];
  __shared__ float Gs[6][kStopThreads];              // w, c0, c1, dw, dc0, dc1 per sample of the chunk
  __shared__ float Red[4][kStopThreads];
  __shared__ float Rg[3][kStopThreads / 64];
  const int K = a.K, B = a.B, d = a.d, tid = threadIdx.x;
  const int64_t p = blockIdx.x;
  // invert the i-major triangular numbering (rows counted from the end have 1, 2, 3, ... pairs)
  const int64_t pe = a.Np - 1 - p;
  int r = (int)((sqrtf(8.f * (float)pe + 1.f) - 1.f) * 0.5f);
  while ((int64_t)(r + 1) * (r + 2) / 2 <= pe) ++r;
  while ((int64_t)r * (r + 1) / 2 > pe) --r;
  const int i = K - r;
  const int j = i + (int)(p - stop_pair_row(i, K));
  const bool last = j == K;
  const float g = a.gam[0], g2 = a.gam[1], g3 = a.gam[2];
  const float tp = a.pt[p], sp = a.ps[p];
  for (int e = tid; e < 4 * E; e += kStopThreads) {
    const int which = e / E, kl = e - which * E, k = kl / D, l = kl - k * D;
    const float* src = which == 0 ? a.N0 : (which == 1 ? a.N1 : (which == 2 ? a.dN0 : a.dN1));
    Nm[which][kl] = (k < d && l < d) ? src[p * d * d + k * d + l] : 0.f;
  }
  const int e = tid % E, sl = tid / E, ek = e / D, el = e - ek * D;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;
  for (int c0 = 0; c0 < B; c0 += kStopThreads) {
    const int nb = min(kStopThreads, B - c0);
    __syncthreads();                                  // (previous chunk consumed; Nm visible on the first trip)
    for (int idx = tid; idx < nb * d; idx += kStopThreads) {
      const int mm = idx / d, l = idx - mm * d;
      Vg[mm][l] = a.gtarget[((size_t)i * B + c0) * d + idx];
      Vq[mm][l] = last ? a.gT[(size_t)c0 * d + idx] : a.q[((size_t)j * B + c0) * d + idx];
      Vv[mm][l] = last ? 0.f : a.v[((size_t)j * B + c0) * d + idx];
    }
    if (tid < nb) {
      const Gates gt = stop_gates<false>(tp, sp, a.tau[c0 + tid], g, g2, g3, a.Tm, nullptr);
      Gs[0][tid] = gt.w; Gs[1][tid] = gt.c0; Gs[2][tid] = gt.c1; Gs[3][tid] = gt.dw; Gs[4][tid] = gt.dc0; Gs[5][tid] = gt.dc1;
    }
    __syncthreads();
    // (2) matrix gradients
    if (sl < NSL && ek < d && el < d) {
      const float n0 = Nm[0][e], n1 = Nm[1][e], dd0 = Nm[2][e], dd1 = Nm[3][e];
      for (int mm = sl; mm < nb; mm += NSL) {
        const float c0v = Gs[1][mm], c1v = Gs[2][mm], dwv = Gs[3][mm], dc0v = Gs[4][mm], dc1v = Gs[5][mm];
        bool fin;
        nan_to_num1(dc0v * n0 + c0v * dd0 + dc1v * n1 + c1v * dd1 + (ek == el ? dwv : 0.f), fin);
        const float gtk = Vg[mm][ek];
        const float gm = gtk * Vq[mm][el];
        const float ge = fin ? -gtk * Vv[mm][el] : 0.f;
        acc0 += gm * c0v + ge * dc0v;
        acc1 += gm * c1v + ge * dc1v;
        acc2 += ge * c0v;
        acc3 += ge * c1v;
      }
    }
    // (3) gate gradients of sample tid, chained to the three gammas
    if (tid < nb) {
      GateGrads gg;
      const Gates gt = stop_gates<true>(tp, sp, a.tau[c0 + tid], g, g2, g3, a.Tm, &gg);
      float gw = 0.f, gdw = 0.f, gc0 = 0.f, gdc0 = 0.f, gc1 = 0.f, gdc1 = 0.f;
      for (int k = 0; k < d; ++k) {
        const float gtk = Vg[tid][k];
        for (int l = 0; l < d; ++l) {
          const int kl = k * D + l;
          bool fin;
          nan_to_num1(gt.dc0 * Nm[0][kl] + gt.c0 * Nm[2][kl] + gt.dc1 * Nm[1][kl] + gt.c1 * Nm[3][kl] + (k == l ? gt.dw : 0.f),
                      fin);
          const float gm = gtk * Vq[tid][l];
          const float ge = fin ? -gtk * Vv[tid][l] : 0.f;
          gc0 += gm * Nm[0][kl] + ge * Nm[2][kl];
          gc1 += gm * Nm[1][kl] + ge * Nm[3][kl];
          gdc0 += ge * Nm[0][kl];
          gdc1 += ge * Nm[1][kl];
          if (k == l) { gw += gm; gdw += ge; }
        }
      }
      gs0 += gw * gg.w_g + gdw * gg.dw_g + gc0 * gg.c0_g + gdc0 * gg.dc0_g;
      gs1 += gc0 * gg.c0_g2 + gdc0 * gg.dc0_g2;
      gs2 += gw * gg.w_g3 + gdw * gg.dw_g3 + gc1 * gg.c1_g3 + gdc1 * gg.dc1_g3;
    }
  }
  // combine the slices of every entry in a fixed order
  __syncthreads();
  Red[0][tid] = acc0; Red[1][tid] = acc1; Red[2][tid] = acc2; Red[3][tid] = acc3;
  __syncthreads();
  if (tid < E) {
    const int k = tid / D, l = tid - k * D;
    if (k < d && l < d) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int q = 0; q < NSL; ++q) {
        s0 += Red[0][q * E + tid]; s1 += Red[1][q * E + tid]; s2 += Red[2][q * E + tid]; s3 += Red[3][q * E + tid];
      }
      const size_t o = (size_t)p * d * d + k * d + l;
      a.gN0[o] = s0; a.gN1[o] = s1; a.gdN0[o] = s2; a.gdN1[o] = s3;
    }
  }
  // block sums of the gamma partials (wave shuffle, then the four waves in order)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    gs0 += __shfl_down(gs0, off, 64); gs1 += __shfl_down(gs1, off, 64); gs2 += __shfl_down(gs2, off, 64);
  }
  if ((tid & 63) == 0) { Rg[0][tid >> 6] = gs0; Rg[1][tid >> 6] = gs1; Rg[2][tid >> 6] = gs2; }
  __syncthreads();
  if (tid < 3) a.ggam_part[p * 3 + tid] = (Rg[tid][0] + Rg[tid][1]) + (Rg[tid][2] + Rg[tid][3]);
}

}  // namespace socmx

using namespace socmx;

static int stop_check(int32_t d, int32_t K, int32_t B) {
  if (K < 1 || B < 1 || d < 1) return SOCMX_E_DIM;
  if (d > 16) return SOCMX_E_DIM;      // per-sample matrices are formed entry by entry in registers / LDS: d <= 16
  return 0;
}

#define SOCMX_STOP_DISPATCH(KERN, GRID, BLK)                                    \
  if (d == 1) return launch(KERN<1>, GRID, BLK, 0, stream, a);                  \
  if (d == 2) return launch(KERN<2>, GRID, BLK, 0, stream, a);                  \
  if (d <= 4) return launch(KERN<4>, GRID, BLK, 0, stream, a);                  \
  if (d <= 8) return launch(KERN<8>, GRID, BLK, 0, stream, a);                  \
  return launch(KERN<16>, GRID, BLK, 0, stream, a)

extern "C" int socmx_socm_stopping_target_fwd_f32(int32_t d, int32_t K, int32_t B, const float* pair_t, const float* pair_s,
                                                  const float* tau, const float* gammas, float T_model, const float* N0,
                                                  const float* N1, const float* dN0, const float* dN1, const float* q,
                                                  const float* v, const float* gT, float* target,
                                                  socmx_stream_t stream) {
  if (!pair_t || !pair_s || !tau || !gammas || !N0 || !N1 || !dN0 || !dN1 || !q || !v || !gT || !target) return SOCMX_E_NULL;
  if (const int rc = stop_check(d, K, B)) return rc;
  StopArgs a{};
  a.d = d; a.K = K; a.B = B; a.Np = (int64_t)(K + 1) * (K + 2) / 2;
  a.pt = pair_t; a.ps = pair_s; a.tau = tau; a.gam = gammas; a.Tm = T_model;
  a.N0 = N0; a.N1 = N1; a.dN0 = dN0; a.dN1 = dN1; a.q = q; a.v = v; a.gT = gT; a.target = target;
  const dim3 grid(K + 1, (B + 63) / 64), blk(64);
  SOCMX_STOP_DISPATCH(stopping_target_kernel, grid, blk);
}

extern "C" int socmx_socm_stopping_target_bwd_f32(int32_t d, int32_t K, int32_t B, const float* pair_t, const float* pair_s,
                                                  const float* tau, const float* gammas, float T_model, const float* N0,
                                                  const float* N1, const float* dN0, const float* dN1, const float* q,
                                                  const float* v, const float* gT, const float* gtarget, float* gN0,
                                                  float* gN1, float* gdN0, float* gdN1, float* ggamma_part,
                                                  socmx_stream_t stream) {
  if (!pair_t || !pair_s || !tau || !gammas || !N0 || !N1 || !dN0 || !dN1 || !q || !v || !gT || !gtarget || !gN0 || !gN1 ||
      !gdN0 || !gdN1 || !ggamma_part)
    return SOCMX_E_NULL;
  if (const int rc = stop_check(d, K, B)) return rc;
  StopArgs a{};
  a.d = d; a.K = K; a.B = B; a.Np = (int64_t)(K + 1) * (K + 2) / 2;
  a.pt = pair_t; a.ps = pair_s; a.tau = tau; a.gam = gammas; a.Tm = T_model;
  a.N0 = N0; a.N1 = N1; a.dN0 = dN0; a.dN1 = dN1; a.q = q; a.v = v; a.gT = gT;
  a.gtarget = gtarget; a.gN0 = gN0; a.gN1 = gN1; a.gdN0 = gdN0; a.gdN1 = gdN1; a.ggam_part = ggamma_part;
  const dim3 grid((unsigned)a.Np), blk(kStopThreads);
  SOCMX_STOP_DISPATCH(stopping_target_bwd_kernel, grid, blk);
}
