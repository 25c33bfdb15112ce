// socmx_loss.hip -- SOCM matching-loss kernels for gfx950 (MI355X).
//
// Replaces reference SOC_matching/method.py:258-262 (importance weights), 591-690 (least-squares
// target) and 702-720 (weighted residual reduction) in the restated form of SURVEY.md section 8 a6:
//
//   v[j,m]  = -( sqrt(lmbd) sqrt(dt_j) S^-T eps[j,m] + dt_j S^-T u[j,m] )
//   q[j,m]  = dt_j nabla_f(X[j,m]) + nabla_b(X[j,m])^T v[j,m]
//   target[i,m] = sum_{j=i}^{K-1} ( M_ij q[j,m] - dM_ij v[j,m] ) + M_iK nabla_g(X[K,m])
//   objective   = inv_norm * sum_{i,m} w[m] | sigma^T (nablaV[i,m] - target[i,m]) |^2
//
// Nothing of size (Kp,Kp,B,d,d) is ever formed (the reference does: method.py:614-618).
// These are HBM/L2-bound streaming reductions at d <= 16: lanes run along the batch index so every
// operand load is a contiguous 256-byte wave access, the d x d pair matrices are shared through LDS
// (transposed, 16-byte broadcast reads), and block/wave reductions feed one atomic per workgroup.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/socmx.h"

namespace socmx {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// block-wide sum, result broadcast to all threads; `red` = 32 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// ---- importance weights: method.py:258-262, 903-904 -------------------------------------------
__global__ __launch_bounds__(256) void weights_stats_kernel(const float* __restrict__ lpd, const float* __restrict__ lps,
                                                            const float* __restrict__ ltw, int B, float* __restrict__ w,
                                                            float* __restrict__ stats) {
  __shared__ float red[32];
  float s = 0.f;
  for (int m = threadIdx.x; m < B; m += blockDim.x) {
    const float x = expf(lpd[m] + lps[m] + ltw[m]);
    w[m] = x;
    s += x;
  }
  const float total = block_sum(s, red);
  const float mean = total / (float)B;
  float q = 0.f;
  for (int m = threadIdx.x; m < B; m += blockDim.x) {
    const float c = w[m] - mean;  // own writes: visible to the same thread
    q += c * c;
  }
  const float m2 = block_sum(q, red);
  if (threadIdx.x == 0) {
    stats[0] = total;
    stats[1] = m2;
    stats[2] = (float)B;
  }
}

// ---- operand preparation: method.py:591-646 -------------------------------------------------------
struct PrepArgs {
  int kind, d, K, B;
  float sqrt_lmbd;
  const float *sit, *A, *P, *Q, *omega, *kappa, *nu;
  const float *ts, *states, *noises, *controls, *frac;
  float *v, *q, *gT;      // (K,B,d), (K,B,d), (B,d)      batch-major  (backward kernel)
  float *vT, *qT, *gTT;   // (K,d,B), (K,d,B), (d,B)      batch-fastest (forward kernel)
};

// one thread per (j, m); j == K handles the terminal row (nabla_g)
__global__ __launch_bounds__(256) void socm_prep_kernel(const PrepArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int d = a.d, B = a.B, K = a.K;
  if (idx >= (int64_t)(K + 1) * B) return;
  const int j = (int)(idx / B), m = (int)(idx - (int64_t)j * B);
  const float* x = a.states + ((size_t)j * B + m) * d;
  if (j == K) {
    for (int l = 0; l < d; ++l) {
      float g = 0.f;
      if (a.kind == SOCMX_OU_QUADRATIC) {          // 2 Q x          OU_quadratic.py:82-83
        for (int c = 0; c < d; ++c) g += a.Q[l * d + c] * x[c];
        g *= 2.f;
      } else if (a.kind == SOCMX_OU_LINEAR) {      // omega          OU_linear.py:87-96
        g = a.omega[l];
      } else if (a.kind == SOCMX_DOUBLE_WELL) {    // 4 nu x (x^2-1)  double_well.py:87-97
        g = 2.f * a.nu[l] * (x[l] * x[l] - 1.f) * 2.f * x[l];
      }
      a.gT[(size_t)m * d + l] = g;
      a.gTT[(size_t)l * B + m] = g;
    }
    return;
  }
  const float dt = a.frac ? a.frac[(size_t)j * B + m] : (a.ts[j + 1] - a.ts[j]);
  const float sdt = sqrtf(dt);
  const float* eps = a.noises + ((size_t)j * B + m) * d;
  const float* u = a.controls + ((size_t)j * B + m) * d;
  const bool is_ou = (a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR);
  // v = -( sqrt(lmbd) sqrt(dt) S^-T eps + dt S^-T u )
  for (int l = 0; l < d; ++l) {
    float se = 0.f, su = 0.f;
    for (int c = 0; c < d; ++c) {
      se += a.sit[l * d + c] * eps[c];
      su += a.sit[l * d + c] * u[c];
    }
    const float vl = -(a.sqrt_lmbd * sdt * se + dt * su);
    a.v[((size_t)j * B + m) * d + l] = vl;
    a.vT[((size_t)j * d + l) * B + m] = vl;
  }
  // q = dt nabla_f + nabla_b^T v   (second pass reads this thread's own v back)
  const float* vv = a.v + ((size_t)j * B + m) * d;
  for (int l = 0; l < d; ++l) {
    float ql;
    if (is_ou) {
      float s = 0.f;
      for (int n = 0; n < d; ++n) s += a.A[n * d + l] * vv[n];  // (A^T v)_l
      ql = s;
      if (a.kind == SOCMX_OU_QUADRATIC) {                          // nabla_f = 2 P x
        float px = 0.f;
        for (int c = 0; c < d; ++c) px += a.P[l * d + c] * x[c];
        ql += dt * 2.f * px;
      }
    } else {  // diagonal Jacobian: -(12 kappa x^2 - 4 kappa)
      const float kap = a.kappa[l];
      ql = -(8.f * kap * x[l] * x[l] + 4.f * kap * (x[l] * x[l] - 1.f)) * vv[l];
    }
    a.q[((size_t)j * B + m) * d + l] = ql;
    a.qT[((size_t)j * d + l) * B + m] = ql;
  }
}

// ---- target + residual (forward) --------------------------------------------------------------------
struct TargetArgs {
  int d, K, B, KG;        // KG = k-groups (waves) per workgroup; each thread owns KO outputs
  float inv_norm;
  const float *sigma;
  const float *M_all, *dM_all;   // (Np,d,d)
  const float *qT, *vT, *gTT;    // (K,d,B), (K,d,B), (d,B)
  const float *nablaV, *w;       // (Kp,B,d), (B,)
  float *target, *G, *objective; // (Kp,B,d) or NULL, (Kp,B,d), (1,)
};

__host__ __device__ inline int64_t pair_row_offset(int i, int K) {
  return (int64_t)i * (K + 1) - (int64_t)i * (i - 1) / 2;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// target[i,m,:] = sum_{j>=i} ( M_ij q_j[m] - dM_ij v_j[m] ) + M_iK gT[m]  as per-row GEMMs on the fp32 MFMA:
//   D (16 k-rows x 16 batch columns) += A (M_ij[k][l], 16 x 4) . B (q_j[l][m], 4 x 16)      v_mfma_f32_16x16x4_f32
// One wave owns (row pair (i, K-i), 16-column batch tile, 16-row k-block): every wave runs K+2 pair matrices
// (balanced triangular work), KP2 x B/16 x ceil(d/16) waves fill the chip, and the next (pair, l-block)'s four
// operand fragments are loaded while the current one is multiplied.  Lanes outside d x d are fed zeros.
constexpr int kTargetWaves = 8;  // waves per workgroup: the (pair, l-block) iterations of a row are dealt round-robin

__global__ __launch_bounds__(64 * kTargetWaves) void socm_target_mfma_kernel(const TargetArgs a) {
  __shared__ f32x4 part[kTargetWaves][64];
  const int d = a.d, K = a.K, B = a.B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int m = blockIdx.y * 16 + c16;
  const int mc = m < B ? m : B - 1;
  const int kb = blockIdx.z * 16;          // first k-row of this wave's block
  const int krow = kb + c16;               // A-fragment row of this lane
  const int nlb = (d + 15) >> 4;           // 16-wide l-blocks
  const int dd = d * d;
  for (int rep = 0; rep < 2; ++rep) {
    const int i = rep == 0 ? (int)blockIdx.x : K - (int)blockIdx.x;
    if (rep == 1 && i <= (int)blockIdx.x) break;
    const float* Mrow = a.M_all + (size_t)pair_row_offset(i, K) * dd;
    const float* dMrow = a.dM_all + (size_t)pair_row_offset(i, K) * dd;
    const int niter = (K - i + 1) * nlb;   // flattened (pair, l-block) iterations
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float am[4], ad[4], bq[4], bv[4];
    auto load = [&](int t, float (&xm)[4], float (&xd)[4], float (&xq)[4], float (&xv)[4]) {
      const int jr = t / nlb, lb = (t - jr * nlb) * 16;
      const int j = i + jr;
      const float* Mp = Mrow + (size_t)jr * dd;
      const float* dMp = dMrow + (size_t)jr * dd;
      const float* qs = (j < K) ? a.qT + (size_t)j * d * B : a.gTT;
      const float* vs = a.vT + (size_t)(j < K ? j : 0) * d * B;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int l = lb + 4 * s + g4;
        const bool okl = l < d, oka = okl && krow < d;
        xm[s] = oka ? Mp[krow * d + l] : 0.f;
        xd[s] = (oka && j < K) ? -dMp[krow * d + l] : 0.f;
        xq[s] = okl ? qs[(size_t)l * B + mc] : 0.f;
        xv[s] = (okl && j < K) ? vs[(size_t)l * B + mc] : 0.f;
      }
    };
    if (wave < niter) load(wave, am, ad, bq, bv);
    for (int t = wave; t < niter; t += kTargetWaves) {
      float nm[4], nd[4], nq[4], nv[4];
      if (t + kTargetWaves < niter) load(t + kTargetWaves, nm, nd, nq, nv);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(am[s], bq[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ad[s], bv[s], acc, 0, 0, 0);
      }
      if (t + kTargetWaves < niter) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { am[s] = nm[s]; ad[s] = nd[s]; bq[s] = nq[s]; bv[s] = nv[s]; }
      }
    }
    // combine the waves' partial tiles (fixed order: deterministic)
    __syncthreads();
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
      acc = part[0][lane];
#pragma unroll
      for (int w = 1; w < kTargetWaves; ++w) acc += part[w][lane];
    }
    // D: lane holds k = kb + 4*g4 + r (r = 0..3) of batch column m
    if (wave == 0 && m < B) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kb + 4 * g4 + r;
        if (k < d) a.target[((size_t)i * B + m) * d + k] = acc[r];
      }
    }
  }
}

// r = sigma^T (nablaV - target), objective += inv_norm * sum w |r|^2, G = 2 w inv_norm sigma r.
// Workgroup = one row i x 64 batch lanes; wave shuffle -> LDS -> one atomic per workgroup.
__global__ __launch_bounds__(64) void socm_residual_kernel(const TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int d = a.d, B = a.B;
  float* diff = lds;                 // [64][d+1]
  float* R = diff + 64 * (d + 1);    // [64][d+1]
  const int ml = threadIdx.x;
  const int i = blockIdx.x;
  const int m = blockIdx.y * 64 + ml;
  const bool valid = m < B;
  const int mc = valid ? m : B - 1;
  const float wm = a.w[mc];
  const size_t row = ((size_t)i * B + mc) * d;
  for (int k = 0; k < d; ++k) diff[ml * (d + 1) + k] = a.nablaV[row + k] - a.target[row + k];
  float obj = 0.f;
  for (int c = 0; c < d; ++c) {
    float r = 0.f;
    for (int k = 0; k < d; ++k) r += a.sigma[k * d + c] * diff[ml * (d + 1) + k];
    R[ml * (d + 1) + c] = r;
    if (valid) obj += wm * r * r;
  }
  if (valid) {
    for (int k = 0; k < d; ++k) {
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += a.sigma[k * d + c] * R[ml * (d + 1) + c];
      a.G[row + k] = 2.f * wm * a.inv_norm * s;
    }
  }
  obj = wave_sum(obj);
  if (ml == 0) atomicAdd(a.objective, obj * a.inv_norm);
}

// ---- backward: gradients w.r.t. the pair matrices ---------------------------------------------------
struct TargetBwdArgs {
  int d, K, B;
  const float *G, *q, *v, *gT;   // (Kp,B,d), (K,B,d), (K,B,d), (B,d)
  float *gM, *gdM;               // (Np,d,d)
};

// thread <-> one column (j,l) of row i; accumulates over the batch the KC rows k0..k0+KC-1:
//   gM[i,j][k][l] = -sum_m G[i,m,k] q[j,m,l]     gdM[i,j][k][l] = +sum_m G[i,m,k] v[j,m,l]
template <int KC>
__global__ __launch_bounds__(256) void socm_target_bwd_kernel(const TargetBwdArgs a) {
  const int d = a.d, K = a.K, B = a.B;
  const int i = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;  // column inside row i: (j-i)*d + l
  const int ncols = (K + 1 - i) * d;
  if (blockIdx.x * blockDim.x >= ncols) return;
  const bool valid = c < ncols;
  const int cc = valid ? c : ncols - 1;
  const int jr = cc / d, l = cc - jr * d;
  const int j = i + jr;
  const bool last = (j == K);
  const float* qcol = last ? a.gT + l : a.q + (size_t)j * B * d + l;
  const float* vcol = a.v + (size_t)(last ? 0 : j) * B * d + l;
  const float* Grow = a.G + (size_t)i * B * d;
  float* outM = a.gM + (size_t)(pair_row_offset(i, K) + jr) * d * d + l;
  float* outD = a.gdM + (size_t)(pair_row_offset(i, K) + jr) * d * d + l;
  for (int k0 = 0; k0 < d; k0 += KC) {
    float aq[KC], av[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) { aq[kk] = 0.f; av[kk] = 0.f; }
    for (int m = 0; m < B; ++m) {
      const float ql = qcol[(size_t)m * d];
      const float vl = last ? 0.f : vcol[(size_t)m * d];
      const float* g = Grow + (size_t)m * d + k0;  // wave-uniform address -> scalar loads
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        const float gk = (k0 + kk < d) ? g[kk] : 0.f;
        aq[kk] += gk * ql;
        av[kk] += gk * vl;
      }
    }
    if (valid) {
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        if (k0 + kk < d) {
          outM[(size_t)(k0 + kk) * d] = -aq[kk];
          outD[(size_t)(k0 + kk) * d] = av[kk];
        }
      }
    }
  }
}

}  // namespace socmx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace socmx;

extern "C" int socmx_weights_stats_f32(const float* lpd, const float* lps, const float* ltw, int32_t B, float* w,
                                       float* stats, socmx_stream_t stream) {
  if (!lpd || !lps || !ltw || !w || !stats) return SOCMX_E_NULL;
  if (B < 1) return SOCMX_E_DIM;
  hipLaunchKernelGGL(weights_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lpd, lps, ltw, B, w, stats);
  return (int)hipGetLastError();
}

extern "C" int64_t socmx_num_pairs(int32_t K) { return K < 0 ? 0 : (int64_t)(K + 1) * (K + 2) / 2; }

extern "C" int socmx_socm_prep_f32(const socmx_problem* pb, const float* ts, int32_t K, int32_t B, float lmbd,
                                   const float* states, const float* noises, const float* controls,
                                   const float* frac, float* v, float* q, float* gT, float* vT, float* qT,
                                   float* gTT, socmx_stream_t stream) {
  if (!pb || !ts || !states || !noises || !controls || !v || !q || !gT || !vT || !qT || !gTT || !pb->sigma_inv_t)
    return SOCMX_E_NULL;
  if (pb->d < 1 || K < 1 || B < 1) return SOCMX_E_DIM;
  switch (pb->kind) {
    case SOCMX_OU_QUADRATIC: if (!pb->A || !pb->P || !pb->Q) return SOCMX_E_NULL; break;
    case SOCMX_OU_LINEAR: if (!pb->A || !pb->omega) return SOCMX_E_NULL; break;
    case SOCMX_DOUBLE_WELL: if (!pb->kappa || !pb->nu) return SOCMX_E_NULL; break;
    case SOCMX_MOLECULAR_DYNAMICS: if (!pb->kappa) return SOCMX_E_NULL; break;
    default: return SOCMX_E_KIND;
  }
  PrepArgs a;
  a.kind = pb->kind; a.d = pb->d; a.K = K; a.B = B; a.sqrt_lmbd = sqrtf(lmbd);
  a.sit = pb->sigma_inv_t; a.A = pb->A; a.P = pb->P; a.Q = pb->Q; a.omega = pb->omega; a.kappa = pb->kappa;
  a.nu = pb->nu;
  a.ts = ts; a.states = states; a.noises = noises; a.controls = controls; a.frac = frac;
  a.v = v; a.q = q; a.gT = gT; a.vT = vT; a.qT = qT; a.gTT = gTT;
  const int64_t n = (int64_t)(K + 1) * B;
  hipLaunchKernelGGL(socm_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int socmx_socm_target_fwd_f32(const socmx_problem* pb, int32_t K, int32_t B, const float* M_all,
                                         const float* dM_all, const float* qT, const float* vT, const float* gTT,
                                         const float* nablaV, const float* w, float inv_norm, float* target,
                                         float* G, float* objective, socmx_stream_t stream) {
  if (!pb || !M_all || !dM_all || !qT || !vT || !gTT || !nablaV || !w || !G || !objective || !target || !pb->sigma)
    return SOCMX_E_NULL;
  const int d = pb->d;
  if (d < 1 || d > 1024 || K < 1 || B < 1) return SOCMX_E_DIM;
  TargetArgs a;
  a.d = d; a.K = K; a.B = B; a.KG = 0; a.inv_norm = inv_norm;
  a.sigma = pb->sigma; a.M_all = M_all; a.dM_all = dM_all; a.qT = qT; a.vT = vT; a.gTT = gTT;
  a.nablaV = nablaV; a.w = w; a.target = target; a.G = G; a.objective = objective;
  dim3 grid((K + 2) / 2, (B + 15) / 16, (d + 15) / 16);
  hipLaunchKernelGGL(socm_target_mfma_kernel, grid, dim3(64 * kTargetWaves), 0, (hipStream_t)stream, a);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return (int)err;
  const size_t lds = (size_t)2 * 64 * (d + 1) * sizeof(float);
  if (lds > 160 * 1024) return SOCMX_E_LDS;
  auto kern = socm_residual_kernel;
  err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(kern, dim3(K + 1, (B + 63) / 64), dim3(64), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int socmx_socm_target_bwd_f32(int32_t d, int32_t K, int32_t B, const float* G, const float* q,
                                         const float* v, const float* gT, float* gM, float* gdM,
                                         socmx_stream_t stream) {
  if (!G || !q || !v || !gT || !gM || !gdM) return SOCMX_E_NULL;
  if (d < 1 || K < 1 || B < 1) return SOCMX_E_DIM;
  TargetBwdArgs a;
  a.d = d; a.K = K; a.B = B; a.G = G; a.q = q; a.v = v; a.gT = gT; a.gM = gM; a.gdM = gdM;
  dim3 grid(((K + 1) * d + 255) / 256, K + 1);
  hipLaunchKernelGGL(socm_target_bwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
