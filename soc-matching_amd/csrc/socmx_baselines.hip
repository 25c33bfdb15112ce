// socmx_baselines.hip -- the reference's OTHER losses on the rollout buffers, gfx950.
//
// Two kernel families over the buffers the fused rollout leaves in HBM (no reference counterpart beyond the eager op lists):
//   matching family (SOCM_const_M, SOCM_exp, SOCM_adjoint: method.py:289-478, 722-749): the least-squares TARGET of
//     objective = sum w |sigma^T (nabla_V - target)|^2 / ((K+1) B)  as one launch --
//       const_M : target[i] = sum_{j>=i}^{K-1} q_j + nabla_g(X_K)                      (a reverse running sum per column)
//       exp     : target[i] = sum_{j>=i} e^{-gamma (t_j - t_i)} (q_j + gamma v_j) + e^{-gamma (T - t_i)} nabla_g(X_K),
//                 together with d target / d gamma (the same recurrence, differentiated: method.py:371-478 trains gamma)
//       adjoint : a_K = nabla_g(X_K),  a_i = a_{i+1} + dt ( (nabla_f_i + nabla_f_{i+1}) / 2 + ((J_i + J_{i+1}) / 2)^T a_{i+1} )
//                 -- the K-step costate recursion of method.py:722-749 INSIDE one kernel (the reference, and round 2's torch
//                 form, launch ~10 kernels per step)
//     with q, v, nabla_g from socmx_socm_prep_f32; the residual / objective / d obj / d nabla_V come from
//     socmx_socm_residual_f32 (the residual kernels of the SOCM loss).
//   Girsanov family (cross_entropy, variance, log-variance, moment: method.py:751-856): per (step, sample)
//       c[i,m] = dt ( -<l,u>/lmbd + |l|^2/(2 lmbd) [- f(X_i)/lmbd] ) [stop] - sqrt(dt/lmbd) <l,eps> [stop],   l = -sigma^T nabla_V
//     in one launch (the sample sums and the B-long loss formulas stay in torch), and its backward
//       d obj / d nabla_V[i,m] = gtotal[m] * ( -sigma dl ),   dl = dt (l - u)/lmbd - sqrt(dt/lmbd) eps   [stop]
//     in one launch.
// Bound: HBM / latency (VALU kernels; these losses are not on a BASELINE metric).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/socmx.h"
#include "socmx_launch.h"

namespace socmx {

// ---- matching family: running sums --------------------------------------------------------------------------------
struct ScanArgs {
  int kind, K, B, d;            // kind 0 = const_M, 1 = exp
  float T;
  const float* ts;              // (K+1,)
  const float* gamma;           // (1,) device (exp)
  const float *q, *v, *gT;      // (K,B,d), (K,B,d), (B,d)
  float* target;                // (K+1,B,d)
  float* dtarget;               // (K+1,B,d) d target / d gamma (exp) or nullptr
};

__global__ __launch_bounds__(256) void matching_scan_kernel(const ScanArgs a) {
  const int64_t n = (int64_t)a.B * a.d;
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  const int K = a.K;
  const float gterm = a.gT[c];
  if (a.kind == 0) {
    float acc = 0.f;
    a.target[(size_t)K * n + c] = gterm;
    for (int i = K - 1; i >= 0; --i) {
      acc += a.q[(size_t)i * n + c];
      a.target[(size_t)i * n + c] = acc + gterm;
    }
    return;
  }
  const float gam = a.gamma[0];
  float S = 0.f, dS = 0.f;                       // sum_{j>=i} e^{-gamma (t_j - t_i)} x_j and its derivative in gamma
  {
    const float e = expf(-gam * (a.T - a.ts[K]));
    a.target[(size_t)K * n + c] = e * gterm;
    if (a.dtarget) a.dtarget[(size_t)K * n + c] = -(a.T - a.ts[K]) * e * gterm;
  }
  for (int i = K - 1; i >= 0; --i) {
    const float dt = a.ts[i + 1] - a.ts[i];
    const float e = expf(-gam * dt);
    const float qi = a.q[(size_t)i * n + c], vi = a.v[(size_t)i * n + c];
    dS = vi + e * (dS - dt * S);
    S = qi + gam * vi + e * S;
    const float rem = a.T - a.ts[i], et = expf(-gam * rem);
    a.target[(size_t)i * n + c] = S + et * gterm;
    if (a.dtarget) a.dtarget[(size_t)i * n + c] = dS - rem * et * gterm;
  }
}

// ---- matching family: the adjoint recursion --------------------------------------------------------------------------
struct AdjArgs {
  int kind, K, B, d;
  float dt;                     // the constant step T / K the reference uses here (method.py:736)
  const float *A, *P, *kappa;
  const float* states;          // (K+1,B,d)
  const float* gT;              // (B,d)  nabla_g(X_K)
  float* target;                // (K+1,B,d) the costates a_i
};

// workgroup = RPB samples x dp lanes (dp = d rounded up to a power of two <= 64); the costate of a sample lives in LDS so
// that the dense OU Jacobian-transpose product can read all its components
__global__ __launch_bounds__(256) void adjoint_kernel(const AdjArgs a, int dp, int rpb) {
  extern __shared__ float sh[];
  const int d = a.d, B = a.B, K = a.K;
  float* av = sh;                                 // [rpb][dp] costate
  float* Am = sh + rpb * dp;                      // [d][d] A (OU settings)
  const int r = threadIdx.x / dp, k = threadIdx.x - r * dp;
  const int m = blockIdx.x * rpb + r;
  const bool live = r < rpb && m < B && k < d;
  const bool is_ou = a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR;
  if (is_ou)
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) Am[e] = a.A[e];
  float ak = live ? a.gT[(size_t)m * d + k] : 0.f;
  if (live) a.target[((size_t)K * B + m) * d + k] = ak;
  const float kap = (live && !is_ou) ? a.kappa[k] : 0.f;
  // nabla_f (OU_quadratic only): 2 P x -- needs the whole state row of the sample: staged next to the costate
  float* xs = Am + (is_ou ? d * d : 0);           // [rpb][2][dp] states X_{i+1}, X_i
  auto nabla_f_k = [&](const float* xrow) {
    float s = 0.f;
    for (int c = 0; c < d; ++c) s += a.P[k * d + c] * xrow[c];
    return 2.f * s;
  };
  float x_hi = live ? a.states[((size_t)K * B + m) * d + k] : 0.f;          // X_{i+1}[k]
  for (int i = K - 1; i >= 0; --i) {
    const float x_lo = live ? a.states[((size_t)i * B + m) * d + k] : 0.f; // X_i[k]
    if (r < rpb && k < dp) {
      av[r * dp + k] = ak;
      if (a.kind == SOCMX_OU_QUADRATIC) { xs[(r * 2 + 0) * dp + k] = x_hi; xs[(r * 2 + 1) * dp + k] = x_lo; }
    }
    __syncthreads();
    if (live) {
      float jb, nf = 0.f;
      if (is_ou) {                                   // (J^T a)_k = sum_n A[n][k] a_n, the same at both ends
        float s = 0.f;
        for (int n = 0; n < d; ++n) s += Am[n * d + k] * av[r * dp + n];
        jb = s;
        if (a.kind == SOCMX_OU_QUADRATIC)
          nf = 0.5f * (nabla_f_k(xs + (r * 2 + 0) * dp) + nabla_f_k(xs + (r * 2 + 1) * dp));
      } else {                                       // diagonal: -(12 kappa x^2 - 4 kappa), averaged over the two ends
        const float j_hi = -(12.f * kap * x_hi * x_hi - 4.f * kap), j_lo = -(12.f * kap * x_lo * x_lo - 4.f * kap);
        jb = 0.5f * (j_lo + j_hi) * ak;
      }
      ak = ak + a.dt * (nf + jb);
      a.target[((size_t)i * B + m) * d + k] = ak;
    }
    x_hi = x_lo;
    __syncthreads();
  }
}

// ---- Girsanov family -------------------------------------------------------------------------------------------------------
struct GirArgs {
  int kind, K, B, d, with_f, sigma_identity;
  float lmbd;
  const float *sigma, *P;
  const float* ts;
  const float *nablaV, *noises, *controls, *states;   // (K+1,B,d), (K,B,d), (K,B,d), (K+1,B,d)
  const float* frac;            // (K,B) or nullptr
  const float* stop;            // (K+1,B) or nullptr
  float* c;                     // forward: (K,B)
  const float* gtotal;          // backward: (B,) d obj / d total_m
  float* G;                     // backward: (K+1,B,d) d obj / d nabla_V
};

// thread = (step i, sample m); the row's vectors are read d (or d^2, dense sigma) times from L1/L2
template <bool BWD>
__global__ __launch_bounds__(256) void girsanov_kernel(const GirArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int K = a.K, B = a.B, d = a.d;
  const int64_t rows = (int64_t)(BWD ? K + 1 : K) * B;
  if (idx >= rows) return;
  const int i = (int)(idx / B), m = (int)(idx - (int64_t)i * B);
  const size_t off = (size_t)idx * d;
  if (BWD && i == K) {                               // the last grid point enters no integrand
    for (int j = 0; j < d; ++j) a.G[off + j] = 0.f;
    return;
  }
  const float dt = a.frac ? a.frac[idx] : (a.ts[i + 1] - a.ts[i]);
  const float st = a.stop ? a.stop[idx] : 1.f;
  const float il = 1.f / a.lmbd, sq = sqrtf(dt * il);
  const float* nv = a.nablaV + off;
  const float* u = a.controls + off;
  const float* eps = a.noises + off;
  if (!BWD) {
    float lu = 0.f, ll = 0.f, le = 0.f;
    for (int k = 0; k < d; ++k) {
      float l;
      if (a.sigma_identity) l = -nv[k];
      else { float s = 0.f; for (int j = 0; j < d; ++j) s += nv[j] * a.sigma[j * d + k]; l = -s; }
      lu += l * u[k]; ll += l * l; le += l * eps[k];
    }
    float det = -il * lu + 0.5f * il * ll;
    if (a.with_f) {
      float f = 0.f;
      if (a.kind == SOCMX_OU_QUADRATIC) {
        const float* x = a.states + off;
        for (int k = 0; k < d; ++k) { float s = 0.f; for (int c2 = 0; c2 < d; ++c2) s += a.P[k * d + c2] * x[c2]; f += x[k] * s; }
      } else if (a.kind == SOCMX_MOLECULAR_DYNAMICS) {
        f = 1.f;
      }
      det -= il * f;
    }
    a.c[idx] = st * (det * dt - sqrtf(il) * le * sqrtf(dt));
  } else {
    const float gt = a.gtotal[m] * st;
    // dl_k = dt (l_k - u_k) / lmbd - sqrt(dt / lmbd) eps_k;   G_j = -gt sum_k sigma[j][k] dl_k
    if (a.sigma_identity) {
      for (int j = 0; j < d; ++j) a.G[off + j] = -gt * (dt * il * (-nv[j] - u[j]) - sq * eps[j]);
    } else {
      for (int j = 0; j < d; ++j) {
        float s = 0.f;
        for (int k = 0; k < d; ++k) {
          float l = 0.f;
          for (int j2 = 0; j2 < d; ++j2) l += nv[j2] * a.sigma[j2 * d + k];
          s += a.sigma[j * d + k] * (dt * il * (-l - u[k]) - sq * eps[k]);
        }
        a.G[off + j] = -gt * s;
      }
    }
  }
}

}  // namespace socmx

using namespace socmx;

extern "C" int socmx_matching_target_f32(int32_t kind, const socmx_problem* pb, int32_t K, int32_t B, const float* ts, float T,
                                         float dt, const float* gamma, const float* q, const float* v, const float* gT,
                                         const float* states, float* target, float* dtarget, socmx_stream_t stream) {
  if (!pb || !ts || !gT || !target) return SOCMX_E_NULL;
  const int d = pb->d;
  if (d < 1 || K < 1 || B < 1 || kind < 0 || kind > 2) return SOCMX_E_DIM;
  if (kind <= 1) {
    if (!q || (kind == 1 && (!v || !gamma))) return SOCMX_E_NULL;
    ScanArgs a;
    a.kind = kind; a.K = K; a.B = B; a.d = d; a.T = T; a.ts = ts; a.gamma = gamma; a.q = q; a.v = v; a.gT = gT;
    a.target = target; a.dtarget = kind == 1 ? dtarget : nullptr;
    const int64_t n = (int64_t)B * d;
    return launch(matching_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
  }
  if (!states) return SOCMX_E_NULL;
  if (d > 64) return SOCMX_E_DIM;
  AdjArgs a;
  a.kind = pb->kind; a.K = K; a.B = B; a.d = d; a.dt = dt; a.A = pb->A; a.P = pb->P; a.kappa = pb->kappa;
  a.states = states; a.gT = gT; a.target = target;
  const bool is_ou = pb->kind == SOCMX_OU_QUADRATIC || pb->kind == SOCMX_OU_LINEAR;
  if (is_ou ? !pb->A : !pb->kappa) return SOCMX_E_NULL;
  if (pb->kind == SOCMX_OU_QUADRATIC && !pb->P) return SOCMX_E_NULL;
  int dp = 1;
  while (dp < d) dp <<= 1;
  const int rpb = 256 / dp;
  const size_t lds = ((size_t)rpb * dp * 3 + (is_ou ? (size_t)d * d : 0)) * sizeof(float);
  return launch(adjoint_kernel, dim3((B + rpb - 1) / rpb), dim3(256), lds, stream, a, dp, rpb);
}

static int gir_args(GirArgs& a, const socmx_problem* pb, int32_t K, int32_t B, float lmbd, int32_t with_f, const float* ts,
                    const float* nablaV, const float* noises, const float* controls, const float* states, const float* frac,
                    const float* stop) {
  if (!pb || !ts || !nablaV || !noises || !controls || !pb->sigma) return SOCMX_E_NULL;
  if (pb->d < 1 || K < 1 || B < 1 || !(lmbd > 0.f)) return SOCMX_E_DIM;
  if (with_f && pb->kind == SOCMX_OU_QUADRATIC && (!states || !pb->P)) return SOCMX_E_NULL;
  a.kind = pb->kind; a.K = K; a.B = B; a.d = pb->d; a.with_f = with_f; a.sigma_identity = (pb->flags & SOCMX_SIGMA_IDENTITY) ? 1 : 0;
  a.lmbd = lmbd; a.sigma = pb->sigma; a.P = pb->P; a.ts = ts; a.nablaV = nablaV; a.noises = noises; a.controls = controls;
  a.states = states; a.frac = frac; a.stop = stop; a.c = nullptr; a.gtotal = nullptr; a.G = nullptr;
  return 0;
}

extern "C" int socmx_girsanov_fwd_f32(const socmx_problem* pb, int32_t K, int32_t B, float lmbd, int32_t with_f,
                                      const float* ts, const float* nablaV, const float* noises, const float* controls,
                                      const float* states, const float* frac, const float* stop, float* c,
                                      socmx_stream_t stream) {
  GirArgs a;
  if (const int rc = gir_args(a, pb, K, B, lmbd, with_f, ts, nablaV, noises, controls, states, frac, stop)) return rc;
  if (!c) return SOCMX_E_NULL;
  a.c = c;
  const int64_t rows = (int64_t)K * B;
  return launch(girsanov_kernel<false>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, a);
}

extern "C" int socmx_girsanov_bwd_f32(const socmx_problem* pb, int32_t K, int32_t B, float lmbd, const float* ts,
                                      const float* nablaV, const float* noises, const float* controls, const float* frac,
                                      const float* stop, const float* gtotal, float* G, socmx_stream_t stream) {
  GirArgs a;
  if (const int rc = gir_args(a, pb, K, B, lmbd, 0, ts, nablaV, noises, controls, nullptr, frac, stop)) return rc;
  if (!gtotal || !G) return SOCMX_E_NULL;
  a.gtotal = gtotal; a.G = G;
  const int64_t rows = (int64_t)(K + 1) * B;
  return launch(girsanov_kernel<true>, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, a);
}
