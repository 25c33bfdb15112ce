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

template <int KO>
__global__ void socm_target_fwd_kernel(const TargetArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int d = a.d, K = a.K, B = a.B;
  const int dpad = a.KG * KO;
  float* Mt = lds;                    // [d][dpad]  Mt[l][k] = M_ij[k][l]
  float* dMt = Mt + d * dpad;         // [d][dpad]
  float* diff = dMt + d * dpad;       // [64][d+1]
  float* R = diff + 64 * (d + 1);     // [64][d+1]
  float* red = R + 64 * (d + 1);      // [32]
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int ml = tid & 63;
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = kg * KO;
  const int m = blockIdx.y * 64 + ml;
  const bool valid = m < B;
  const int mc = valid ? m : B - 1;
  const float wm = a.w[mc];
  float obj_part = 0.f;

  for (int rep = 0; rep < 2; ++rep) {
    // rows are processed in pairs (i, K-i): every workgroup does K+2 pair matrices in total
    const int i = rep == 0 ? (int)blockIdx.x : K - (int)blockIdx.x;
    if (rep == 1 && i <= (int)blockIdx.x) break;
    float acc[KO];
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) acc[kk] = 0.f;
    const int64_t p0 = pair_row_offset(i, K);
    for (int j = i; j <= K; ++j) {
      const float* Mp = a.M_all + (size_t)(p0 + (j - i)) * d * d;
      const float* dMp = a.dM_all + (size_t)(p0 + (j - i)) * d * d;
      __syncthreads();
      for (int e = tid; e < d * d; e += nthr) {
        const int k = e / d, l = e - k * d;
        Mt[l * dpad + k] = Mp[e];
        dMt[l * dpad + k] = (j < K) ? dMp[e] : 0.f;
      }
      __syncthreads();
      const float* qs = (j < K) ? a.qT + (size_t)j * d * B : a.gTT;
      const float* vs = a.vT + (size_t)(j < K ? j : 0) * d * B;
      for (int l = 0; l < d; ++l) {
        const float ql = qs[(size_t)l * B + mc];
        const float vl = (j < K) ? vs[(size_t)l * B + mc] : 0.f;
        const float* mrow = Mt + l * dpad + k0;
        const float* drow = dMt + l * dpad + k0;
#pragma unroll
        for (int kk = 0; kk < KO; kk += 4) {
          const float4 mv = *reinterpret_cast<const float4*>(mrow + kk);
          const float4 dv = *reinterpret_cast<const float4*>(drow + kk);
          acc[kk + 0] += mv.x * ql - dv.x * vl;
          acc[kk + 1] += mv.y * ql - dv.y * vl;
          acc[kk + 2] += mv.z * ql - dv.z * vl;
          acc[kk + 3] += mv.w * ql - dv.w * vl;
        }
      }
    }
    // ---- residual r = sigma^T (nablaV - target), objective, G ---------------------------------
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
      const int k = k0 + kk;
      if (k < d) {
        const float nv = a.nablaV[((size_t)i * B + mc) * d + k];
        diff[ml * (d + 1) + k] = nv - acc[kk];
        if (a.target && valid) a.target[((size_t)i * B + m) * d + k] = acc[kk];
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
      const int c = k0 + kk;
      if (c < d) {
        float r = 0.f;
        for (int k = 0; k < d; ++k) r += a.sigma[k * d + c] * diff[ml * (d + 1) + k];
        R[ml * (d + 1) + c] = r;
        if (valid) obj_part += wm * r * r;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KO; ++kk) {
      const int k = k0 + kk;
      if (k < d && valid) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += a.sigma[k * d + c] * R[ml * (d + 1) + c];
        a.G[((size_t)i * B + m) * d + k] = 2.f * wm * a.inv_norm * s;
      }
    }
  }
  const float tot = block_sum(obj_part, red);
  if (tid == 0) atomicAdd(a.objective, tot * a.inv_norm);
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
  if (!pb || !M_all || !dM_all || !qT || !vT || !gTT || !nablaV || !w || !G || !objective || !pb->sigma)
    return SOCMX_E_NULL;
  const int d = pb->d;
  if (d < 1 || d > 128 || K < 1 || B < 1) return SOCMX_E_DIM;
  constexpr int KO = 8;
  TargetArgs a;
  a.d = d; a.K = K; a.B = B; a.KG = (d + KO - 1) / KO; a.inv_norm = inv_norm;
  a.sigma = pb->sigma; a.M_all = M_all; a.dM_all = dM_all; a.qT = qT; a.vT = vT; a.gTT = gTT;
  a.nablaV = nablaV; a.w = w; a.target = target; a.G = G; a.objective = objective;
  const int dpad = a.KG * KO;
  const size_t lds = ((size_t)2 * d * dpad + 2 * 64 * (d + 1) + 32) * sizeof(float);
  if (lds > 160 * 1024) return SOCMX_E_LDS;
  auto kern = socm_target_fwd_kernel<KO>;
  hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return (int)err;
  dim3 grid((K + 2) / 2, (B + 63) / 64);
  hipLaunchKernelGGL(kern, grid, dim3(64 * a.KG), lds, (hipStream_t)stream, a);
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
