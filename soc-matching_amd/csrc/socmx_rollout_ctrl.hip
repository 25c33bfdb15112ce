// socmx_rollout_ctrl.hip -- Euler-Maruyama rollout under a TABULATED control (no network), gfx950.
//
// Replaces reference SOC_matching/utils.py:17-128 when `sde.u` is one of the ground-truth controls of
// SOC_matching/models.py:10-150 (method.py:103-107 routes `control()` to them):
//   LINEAR    u(t,x) = U[idx(t)] x        LinearControl          (models.py:10-41;  LQ Riccati solution, utils.py:234-254)
//   CONSTANT  u(t,x) = c[idx(t)]          ConstantControlLinear  (models.py:61-83;  OU_linear closed form)
//   TABLE     u_j(t,x) = T[idx(t), clamp(floor((x_j + xb) / dx)), j]   LowDimControl (models.py:98-150; double_well PDE)
// i.e. the optimal-SDE evaluation bursts of main.py:137-150 (512 sequential eager rollouts there: ~75 launches per step).
// The time index of every step is evaluated by the HOST with the reference's own fp32 formula (floor((n-1) t / T),
// floor(n t / T), ceil(t / dt)) and passed as a table: the kernel does no index arithmetic that could round
// differently.  One workgroup = 16 rows x 16 lanes (4 waves); the state tile lives in LDS, per-row sums by DPP.
// Bound: latency/instruction (d^2 multiply-adds per row and step); the win is one launch instead of K x 75.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/socmx.h"
#include "socmx_launch.h"
#include "socmx_philox.h"

namespace socmx {

struct CtrlArgs {
  int kind, d, B, K, ckind, n_x;
  int sigma_identity;
  float lmbd, xb, delta_x;
  uint64_t seed, offset;
  int64_t row0;
  const float *sigma, *A, *P, *Q, *omega, *kappa, *nu;
  const float* table;
  const int32_t* tidx;
  const float *x0, *ts, *noise_in;
  float *states, *noises, *controls, *stop_ind, *frac, *lpd, *lps, *ltw;
};

__global__ __launch_bounds__(256) void rollout_ctrl_kernel(const CtrlArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int ds = d + 1;
  float* XS = lds;                 // (16, ds) state
  float* U = XS + 16 * ds;         // control
  float* E = U + 16 * ds;          // noise
  float* XN = E + 16 * ds;         // new state
  const int tid = threadIdx.x, r = tid >> 4, l = tid & 15;
  const int grow = blockIdx.x * 16 + r;
  const bool live = grow < B;
  const bool traj = a.states != nullptr;
  const int growc = min(grow, B - 1);
  const bool is_ou = kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR;
  for (int i = l; i < d; i += 16) {
    const float x = a.x0[(size_t)growc * d + i];
    XS[r * ds + i] = x;
    if (live && traj) a.states[(size_t)grow * d + i] = x;
  }
  if (l == 0 && live && traj) a.stop_ind[grow] = 1.f;
  float lpd = 0.f, lps = 0.f;
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    const float t0 = a.ts[k], t1 = a.ts[k + 1];
    const float dt = t1 - t0;                        // utils.py:38
    const float sq_ldt = sqrtf(a.lmbd * dt);         // utils.py:47
    const int tk = a.tidx[k];
    const float* x = XS + r * ds;
    // noise first (utils.py:39-41), then the control at the old state
    for (int i = l; i < d; i += 16) {
      const float eps = a.noise_in ? a.noise_in[((size_t)k * B + growc) * d + i]
                                   : philox_normal(a.seed, a.offset, (uint32_t)(a.row0 + grow), (uint32_t)k, i);
      float u;
      if (a.ckind == SOCMX_CTRL_LINEAR) {
        const float* Uk = a.table + ((size_t)tk * d + i) * d;
        u = 0.f;
        for (int j = 0; j < d; ++j) u += Uk[j] * x[j];
      } else if (a.ckind == SOCMX_CTRL_CONSTANT) {
        u = a.table[(size_t)tk * d + i];
      } else {
        int ix = (int)floorf((x[i] + a.xb) / a.delta_x);           // models.py:113 (fp32 add, fp32 divide, floor)
        ix = min(max(ix, 0), a.n_x - 1);
        u = a.table[((size_t)tk * a.n_x + ix) * d + i];
      }
      U[r * ds + i] = u;
      E[r * ds + i] = eps;
      if (live && traj) {
        a.controls[((size_t)k * B + grow) * d + i] = u;
        a.noises[((size_t)k * B + grow) * d + i] = eps;
      }
    }
    __syncthreads();
    float uu = 0.f, ue = 0.f;
    for (int i = l; i < d; i += 16) {
      float bi;
      if (is_ou) {
        bi = 0.f;
        for (int j = 0; j < d; ++j) bi += a.A[i * d + j] * x[j];
      } else {
        const float xi = x[i];
        bi = -2.f * a.kappa[i] * (xi * xi - 1.f) * 2.f * xi;     // double_well.py:44-48
      }
      float su, se;
      if (a.sigma_identity) {
        su = U[r * ds + i];
        se = E[r * ds + i];
      } else {
        su = 0.f; se = 0.f;
        for (int j = 0; j < d; ++j) {
          su += a.sigma[i * d + j] * U[r * ds + j];
          se += a.sigma[i * d + j] * E[r * ds + j];
        }
      }
      XN[r * ds + i] = x[i] + ((bi + su) * dt + sq_ldt * se);       // utils.py:45-48 (stop_inds = 1)
      const float u = U[r * ds + i];
      uu += u * u;
      ue += u * E[r * ds + i];
    }
    uu = row16_sum(uu);
    ue = row16_sum(ue);
    __syncthreads();
    float f = 0.f;                                                  // f at the NEW state, OLD time (utils.py:92-96)
    if (kind == SOCMX_OU_QUADRATIC) {
      const float* xn = XN + r * ds;
      float part = 0.f;
      for (int i = l; i < d; i += 16) {
        float px = 0.f;
        for (int j = 0; j < d; ++j) px += a.P[i * d + j] * xn[j];
        part += xn[i] * px;
      }
      f = row16_sum(part);
    }
    lpd = lpd + dt / a.lmbd * (-f - 0.5f * uu);
    lps = lps + sqrtf(dt / a.lmbd) * (-ue);
    for (int i = l; i < d; i += 16) {
      const float xn = XN[r * ds + i];
      XS[r * ds + i] = xn;
      if (live && traj) a.states[((size_t)(k + 1) * B + grow) * d + i] = xn;
    }
    if (l == 0 && live && traj) {
      a.frac[(size_t)k * B + grow] = dt;
      a.stop_ind[(size_t)(k + 1) * B + grow] = 1.f;
    }
    __syncthreads();
  }
  // terminal cost (utils.py:101)
  const float* x = XS + r * ds;
  float part = 0.f;
  if (kind == SOCMX_OU_QUADRATIC) {
    for (int i = l; i < d; i += 16) {
      float qx = 0.f;
      for (int j = 0; j < d; ++j) qx += a.Q[i * d + j] * x[j];
      part += x[i] * qx;
    }
  } else if (kind == SOCMX_OU_LINEAR) {
    for (int i = l; i < d; i += 16) part += a.omega[i] * x[i];
  } else if (kind == SOCMX_DOUBLE_WELL) {
    for (int i = l; i < d; i += 16) {
      const float q = x[i] * x[i] - 1.f;
      part += a.nu[i] * (q * q);
    }
  }
  const float gval = row16_sum(part);
  if (l == 0 && live) {
    a.lpd[grow] = lpd;
    a.lps[grow] = lps;
    a.ltw[grow] = -gval / a.lmbd;
  }
}

}  // namespace socmx

using namespace socmx;

extern "C" int socmx_rollout_control_f32(const socmx_problem* pb, const socmx_control* ctrl, const float* x0, const float* ts,
                                         int32_t B, int32_t K, float lmbd, uint64_t seed, uint64_t offset, int64_t row0,
                                         const float* noise_in, float* states, float* noises, float* controls,
                                         float* stop_indicators, float* fractional_timesteps, float* lpd, float* lps,
                                         float* ltw, socmx_stream_t stream) {
  if (!pb || !ctrl || !x0 || !ts || !lpd || !lps || !ltw || !pb->sigma || !ctrl->table || !ctrl->tidx) return SOCMX_E_NULL;
  const int n_traj = !!states + !!noises + !!controls + !!stop_indicators + !!fractional_timesteps;
  if (n_traj != 0 && n_traj != 5) return SOCMX_E_NULL;
  const int d = pb->d;
  if (d < 1 || d > 1024 || B < 1 || K < 1) return SOCMX_E_DIM;
  switch (pb->kind) {
    case SOCMX_OU_QUADRATIC: if (!pb->A || !pb->P || !pb->Q) return SOCMX_E_NULL; break;
    case SOCMX_OU_LINEAR: if (!pb->A || !pb->omega) return SOCMX_E_NULL; break;
    case SOCMX_DOUBLE_WELL: if (!pb->kappa || !pb->nu) return SOCMX_E_NULL; break;
    default: return SOCMX_E_KIND;            // molecular_dynamics has no ground-truth control (settings.py:112-114)
  }
  if (ctrl->kind != SOCMX_CTRL_LINEAR && ctrl->kind != SOCMX_CTRL_CONSTANT && ctrl->kind != SOCMX_CTRL_TABLE) return SOCMX_E_KIND;
  if (ctrl->kind == SOCMX_CTRL_TABLE && (ctrl->n_x < 1 || !(ctrl->delta_x > 0.f))) return SOCMX_E_DIM;
  CtrlArgs a;
  a.kind = pb->kind; a.d = d; a.B = B; a.K = K; a.ckind = ctrl->kind; a.n_x = ctrl->n_x;
  a.sigma_identity = (pb->flags & SOCMX_SIGMA_IDENTITY) ? 1 : 0;
  a.lmbd = lmbd; a.xb = ctrl->xb; a.delta_x = ctrl->delta_x;
  a.seed = seed; a.offset = offset; a.row0 = row0;
  a.sigma = pb->sigma; a.A = pb->A; a.P = pb->P; a.Q = pb->Q; a.omega = pb->omega; a.kappa = pb->kappa; a.nu = pb->nu;
  a.table = ctrl->table; a.tidx = ctrl->tidx;
  a.x0 = x0; a.ts = ts; a.noise_in = noise_in;
  a.states = states; a.noises = noises; a.controls = controls; a.stop_ind = stop_indicators; a.frac = fractional_timesteps;
  a.lpd = lpd; a.lps = lps; a.ltw = ltw;
  const size_t lds_bytes = (size_t)4 * 16 * (d + 1) * sizeof(float);
  if (lds_bytes > (size_t)kLdsBytesPerCU) return SOCMX_E_LDS;
  if (const int err = ensure_max_lds(rollout_ctrl_kernel)) return err;
  return launch(rollout_ctrl_kernel, dim3((B + 15) / 16), dim3(256), lds_bytes, stream, a);
}
