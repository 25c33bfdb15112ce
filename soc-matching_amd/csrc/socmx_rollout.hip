// socmx_rollout.hip -- fused Euler-Maruyama rollout for gfx950 (MI355X).
//
// Replaces reference SOC_matching/utils.py:17-128 (stochastic_trajectories) together with the
// per-step control evaluation (method.py:58-80, models.py:233-242) and the per-setting closed
// forms (experiment_settings/*.py).  One workgroup owns one 16-row tile of the batch for ALL
// K steps: the state, the seven activation tiles of the control network and the per-row cost
// accumulators never leave the CU; per step the only HBM traffic is the coalesced write of the
// tile's slab of states / noises / controls (the algorithmic bytes).  Weights stream from L2
// in MFMA fragment order (socmx_unet.h).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/socmx.h"
#include "socmx_unet.h"
#include "socmx_launch.h"
#include "socmx_philox.h"
#include "socmx_rollout_common.h"

namespace socmx {

// drift b_i(x) -- OU_quadratic.py:51-52, OU_linear.py:43-44, double_well.py:44-48, molecular_dynamics.py:49-53
__device__ __forceinline__ float drift_i(int kind, int d, int i, const float* x, const float* A_l,
                                         const float* __restrict__ kappa, int ds) {
  if (kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR) {
    float s = 0.f;
    for (int j = 0; j < d; ++j) s += A_l[i * ds + j] * x[j];
    return s;
  }
  const float xi = x[i];
  return -2.f * kappa[i] * (xi * xi - 1.f) * 2.f * xi;
}

#define SOCMX_TICK(slot)                                   \
  if (PROF) {                                              \
    const long long now_ = clock64();                      \
    acc_prof[slot] += now_ - last_tick;                    \
    last_tick = now_;                                      \
  }

// the four draws of one Philox block (dims 4b .. 4b+3 of a row): same values as four philox_normal calls, a quarter of
// the generator work
__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t offset, uint32_t grow, uint32_t step, int block) {
  uint32_t w[4];
  philox4x32_10(grow, step, (uint32_t)block, (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  f32x4 out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float ua = ((float)w[2 * h] + 0.5f) * 2.3283064365386963e-10f;
    const float ub = ((float)w[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
    const float r = sqrtf(-2.0f * logf(ua));
    float sn, cs;
    sincospif(2.0f * ub, &sn, &cs);
    out[2 * h] = r * cs;
    out[2 * h + 1] = r * sn;
  }
  return out;
}

// one Box-Muller pair of that block (draws 2h, 2h+1): lets two threads share a block's latency instead of one thread
// walking both pairs (the generator is instruction-bound for a lone wave; the redundant block costs idle VALU slots)
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint64_t offset, uint32_t grow, uint32_t step, int block,
                                               int h, float& z0, float& z1) {
  uint32_t w[4];
  philox4x32_10(grow, step, (uint32_t)block, (uint32_t)offset, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const uint32_t wa = h ? w[2] : w[0], wb = h ? w[3] : w[1];
  const float ua = ((float)wa + 0.5f) * 2.3283064365386963e-10f;
  const float ub = ((float)wb + 0.5f) * 2.3283064365386963e-10f;
  const float r = sqrtf(-2.0f * logf(ua));
  float sn, cs;
  sincospif(2.0f * ub, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

// LDS operand of a masked MFMA step: the load is unconditional (the address is clamped in range by the caller) and the
// mask is a select -- written as `c ? p[..] : 0` the compiler branches around every load and waits for each in turn
__device__ __forceinline__ float lds_sel(const float* p, bool c) {
  const float t = *p;
  return c ? t : 0.f;
}

template <int NW, bool STOPPING, bool PROF, class NET, bool FAST>
__global__ __launch_bounds__(NW * 64) void rollout_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // Philox key: by value, or read from device memory (uniform scalar loads, once per workgroup)
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);
  constexpr bool kStatic = !std::is_same<NET, DynamicNet>::value;
  TileLayout tl;
  UnetDesc ud;
  if constexpr (kStatic) {
    constexpr TileLayout tc = NET::layout(NW);
    constexpr UnetDesc uc = NET::desc();
    tl = tc; ud = uc;
  } else {
    tl = a.t; ud = a.u;
  }
  const int tid = threadIdx.x;
  const int nthr = NW * 64;
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int tile_row0 = blockIdx.x * 16;
  // e / d and e / in0p for e < 16*1024 without integer divides: floor((e + 0.5) * (1/n)) is exact in that range
  const float inv_d = __builtin_amdgcn_rcpf((float)d), inv_in0p = __builtin_amdgcn_rcpf((float)ud.in0p);
#define SOCMX_DIV_D(e) ((int)(((float)(e) + 0.5f) * inv_d))
  float* X0 = lds + tl.x0;
  float* GV = lds + tl.gv;
  // small per-tile state (behind the network tiles)
  // row stride of the LDS matrix copies and (16, d) tiles: d rounded up to the MFMA's k-block, plus one for the bank
  // spread.  The padding columns are zeroed once and never written, so the MFMA products read whole 16-wide k-blocks
  // with no per-element clamp or mask.
  const int ds = socmx_sde_stride(d);
  // which constant matrices and tiles exist depends on the setting (d = 64 with all of them would not fit 160 KiB):
  // the launcher sizes the allocation with the same rule (rollout_sde_floats)
  const bool is_ou = (kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR);
  const bool is_quad = kind == SOCMX_OU_QUADRATIC;
  float* sig = lds + a.lds_mats;          // (d,ds)
  float* A_l = sig + d * ds;              // (d,ds)  OU only
  float* KAP = A_l;                       // (d,)    the other settings: kappa (a global load per step would queue
                                          //         behind the step's stores)
  float* P_l = A_l + (is_ou ? d * ds : d);    // (d,ds)  OU_quadratic only
  // (16, d) tiles with row stride ds = d + 1: the MFMA products of the general SDE step read them with the 16
  // rows across the lanes, and a stride of d = 64 floats put all 16 rows on one bank (16-way conflicts)
  float* XS = P_l + (is_quad ? d * ds : 0);   // current state
  float* XN = XS + 16 * ds;               // proposed state
  float* U = XN + 16 * ds;                // control
  float* E0 = U + 16 * ds;                // noise of the even steps ...
  float* E1 = E0 + 16 * ds;               // ... and of the odd ones (general path: drawn one step ahead)
  float* XF = E1 + 16 * ds;               // stopping time only: state after the re-interpolation
  float* UP = XF + (STOPPING ? 16 * ds : 0);  // stopping time only: the update
  float* ST = UP + (STOPPING ? 16 * ds : 0);  // (16,)  stop_inds (1 = still running)
  float* SN = ST + 16;                    // (16,)  next stop_inds
  float* FD = SN + 16;                    // (16,)  fractional time step
  float* NZ = FD + 16;                    // (2,16,16) FAST path: double-buffered noise of steps k, k+1
  float* FQ = NZ;                         // general path: (d/16 blocks, 4, 16) partial sums of x'Px

  for (float* z = sig + tid; z < NZ + 512; z += nthr) *z = 0.f;   // (the padding columns stay zero for good)
  __syncthreads();
  rollout_key_advance(a, key_offset);
  for (int e = tid; e < d * d; e += nthr) {
    const int r = e / d, c = e - r * d;
    sig[r * ds + c] = a.sigma[e];
    if (is_ou) A_l[r * ds + c] = a.A[e];
    if (is_quad) P_l[r * ds + c] = a.P[e];
  }
  if (!is_ou) for (int e = tid; e < d; e += nthr) KAP[e] = a.kappa[e];
  unet_load_biases(a.packed, ud, tl, lds, tid, nthr, ud.folded != 0);
  Pre carry;
  if constexpr (kStatic) carry = unet_carry_init_static<NW, NET>(a.packed);
  else carry = unet_carry_init(a.packed, a.prog);

  if constexpr (FAST) {
    // ---- sigma = I, d <= 15: the whole SDE step of a row lives in one 16-lane group --------------------
    // thread (r = tid>>4, i = tid&15), tid < 256: state x_i, control, noise, update, costs in registers; row
    // sums by 16-lane butterflies; ONE barrier per step besides the network's (method.py:64-80, utils.py:37-101).
    const bool act = tid < 256;
    const int r = (tid >> 4) & 15, i = tid & 15;
    const int ic = min(i, d - 1);
    const bool lane_ok = act && i < d;
    const int grow = tile_row0 + r;
    const bool traj = a.states != nullptr;        // costs-only launches (evaluation bursts) pass no trajectory buffers
    const bool store = lane_ok && grow < B && traj;
    const bool store0 = act && i == 0 && grow < B && traj;
    const size_t rowoff = (size_t)grow * d + i;
    auto gsum = [](float v) { return row16_sum(v); };
    float x = lane_ok ? a.x0[(size_t)min(grow, B - 1) * d + i] : 0.f;
    const float kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
    float stop = 1.f, lpd = 0.f, lps = 0.f;
    if (store) a.states[rowoff] = x;
    if (store0) a.stop_ind[grow] = 1.f;
    if (act) {
      if (i < 15) X0[r * tl.s0 + 1 + i] = x;   // columns 1..15; lanes i >= d hold x = 0
      if (i == 0) X0[r * tl.s0] = a.ts[0];
    }
    // The step's noise does not depend on the network: while waves 0..3 integrate step k, waves 4..7 (idle in
    // this phase) draw / fetch the noise of step k+1 into the other half of NZ.
    constexpr bool kSplitNoise = (NW >= 8);
    const bool producer = kSplitNoise ? (tid >= 256 && tid < 512) : act;
    const int pr = (tid >> 4) & 15, pi = tid & 15, pgrow = tile_row0 + pr;
    auto draw = [&](int k) -> float {
      if (pi >= d || k >= K) return 0.f;
      if (a.noise_in) return a.noise_in[((size_t)k * B + min(pgrow, B - 1)) * d + pi];
      return philox_normal(key_seed, key_offset, (uint32_t)(a.row0 + pgrow), (uint32_t)k, pi);
    };
    if (producer) NZ[pr * 16 + pi] = draw(0);
    long long acc_prof[64];
    if (PROF) for (int s = 0; s < 64; ++s) acc_prof[s] = 0;
    long long last_tick = PROF ? clock64() : 0, last_sub = last_tick;
    for (int k = 0; k < K; ++k) {
      const float t0 = a.ts[k], t1 = a.ts[k + 1];
      const float dt = t1 - t0;                 // utils.py:38
      const float sq_ldt = sqrtf(a.lmbd * dt);  // utils.py:47
      __syncthreads();
      SOCMX_TICK(0)
      auto hook = [&](int slot) {
        if (PROF) {
          const long long now_ = clock64();
          if (slot < 16) { acc_prof[slot] += now_ - last_tick; last_tick = now_; last_sub = now_; }
          else { acc_prof[slot] += now_ - last_sub; last_sub = now_; }
        }
      };
      float gv = 0.f;                                                     // nabla_V[r][i] of this thread
      if constexpr (kStatic && NET::outp == 16) {
        unet_tile_forward_static<NW, NET>(a.packed, lds, carry, hook, &gv);      // last stage -> register, no barrier
      } else {
        if constexpr (kStatic) unet_tile_forward_static<NW, NET>(a.packed, lds, carry, hook);   // GV = nabla_V(t, x)
        else unet_tile_forward<NW>(a.packed, a.prog, a.t, lds, carry, hook);
        if (act) gv = GV[r * tl.sg + i];
      }
      if (store && a.nabla_v) a.nabla_v[(size_t)k * B * d + rowoff] = gv;
      if (act) {
        const float u = lane_ok ? -gv : 0.f;                             // u = -sigma^T nabla_V, sigma = I
        const float eps = NZ[(k & 1) * 256 + r * 16 + i];                 // drawn during the previous step
        float bi;
        if (is_ou) {                                                      // b = A x
          bi = 0.f;
          for (int j = 0; j < d; ++j) bi += A_l[ic * ds + j] * __shfl(x, j, 16);
          if (!lane_ok) bi = 0.f;
        } else {
          bi = -2.f * kap * (x * x - 1.f) * 2.f * x;                      // double_well.py:44-48
        }
        const float upd = (bi + u) * dt + sq_ldt * eps;                   // utils.py:45-47
        const float xn = x + stop * upd;                                  // utils.py:48
        float xe = xn, step = dt, stop_new = 1.f;
        if (STOPPING) {                                                   // utils.py:42-44, 49-75; Phi = -x_0
          const float phi_b = -__shfl(x, 0, 16), phi_a = -__shfl(xn, 0, 16);
          const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
          const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
          const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
          xe = js * (x + fr * stop * upd) + (1.f - js) * xn;
          step = js * (fr * fr) * dt + ns * dt;                           // step_fraction squared (utils.py:70-72)
          stop_new = (-__shfl(xe, 0, 16) > 0.f) ? 1.f : 0.f;
        }
        float f = 0.f;                                                    // f at the NEW state, OLD time (utils.py:92-96)
        if (kind == SOCMX_OU_QUADRATIC) {
          float px = 0.f;
          for (int j = 0; j < d; ++j) px += P_l[ic * ds + j] * __shfl(xe, j, 16);
          f = gsum(lane_ok ? xe * px : 0.f);
        } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
          f = 1.f;
        }
        const float uu = gsum(u * u), ue = gsum(u * eps);
        lpd = lpd + step / a.lmbd * (-f - 0.5f * uu);
        lps = lps + sqrtf(step / a.lmbd) * (-ue);
        if (store) {
          a.controls[(size_t)k * B * d + rowoff] = u;
          a.noises[(size_t)k * B * d + rowoff] = eps;
          a.states[(size_t)(k + 1) * B * d + rowoff] = xe;
        }
        if (store0) {
          a.frac[(size_t)k * B + grow] = step;
          a.stop_ind[(size_t)(k + 1) * B + grow] = STOPPING ? stop_new : 1.f;
        }
        x = lane_ok ? xe : 0.f;
        if (STOPPING) stop = stop_new;
        if (i < 15) X0[r * tl.s0 + 1 + i] = x;                            // next step's network input [t, x, 0..]
        if (i == 0) X0[r * tl.s0] = t1;
      }
      if (producer) NZ[((k + 1) & 1) * 256 + pr * 16 + pi] = draw(k + 1);
      SOCMX_TICK(9)
    }
    if (PROF && tid == a.prof_wave * 64 && a.prof)
      for (int s = 0; s < 64; ++s) a.prof[(size_t)blockIdx.x * 64 + s] = acc_prof[s];
    if (a.nabla_v) {                        // nabla_V(T, X_K): X0 already holds [t_K, x_K] (written at the end of the last step)
      __syncthreads();
      auto nohook = [](int) {};
      float gv = 0.f;
      if constexpr (kStatic && NET::outp == 16) {
        unet_tile_forward_static<NW, NET>(a.packed, lds, carry, nohook, &gv);
      } else {
        if constexpr (kStatic) unet_tile_forward_static<NW, NET>(a.packed, lds, carry, nohook);
        else unet_tile_forward<NW>(a.packed, a.prog, a.t, lds, carry, nohook);
        if (act) gv = GV[r * tl.sg + i];
      }
      if (store) a.nabla_v[(size_t)K * B * d + rowoff] = gv;
    }
    if (act) {                                                            // terminal cost (utils.py:101)
      float gval = 0.f;
      if (kind == SOCMX_OU_QUADRATIC) {
        float qx = 0.f;
        for (int j = 0; j < d; ++j) qx += a.Q[ic * d + j] * __shfl(x, j, 16);
        gval = gsum(lane_ok ? x * qx : 0.f);
      } else if (kind == SOCMX_OU_LINEAR) {
        gval = gsum(lane_ok ? a.omega[ic] * x : 0.f);
      } else if (kind == SOCMX_DOUBLE_WELL) {
        const float q = x * x - 1.f;
        gval = gsum(lane_ok ? a.nu[ic] * (q * q) : 0.f);
      }
      if (i == 0 && grow < B) {
        a.lpd[grow] = lpd;
        a.lps[grow] = lps;
        a.ltw[grow] = -gval / a.lmbd;
      }
    }
    return;
  }
  const int Bs = a.states ? B : 0;   // rows whose trajectory is stored: none in a costs-only launch (evaluation bursts)
  for (int e = tid; e < 16 * d; e += nthr) {
    const int r = SOCMX_DIV_D(e), i = e - r * d;
    const int grow = min(tile_row0 + r, B - 1);  // ragged tail: replicate the last row, never stored
    const float x = a.x0[(size_t)grow * d + i];
    XS[r * ds + i] = x;
    if (tile_row0 + r < Bs) a.states[(size_t)(tile_row0 + r) * d + i] = x;  // states[0]
  }
  if (tid < 16) {
    ST[tid] = 1.f;
    if (tile_row0 + tid < Bs) a.stop_ind[tile_row0 + tid] = 1.f;  // stop_indicators[0] = ones (utils.py:28)
  }
  float lpd = 0.f, lps = 0.f;  // per-row accumulators, live in lane 0 of each 16-lane row group (threads 0, 16, ..., 240)
  const bool mm = d >= 16;                                  // matrix products of the SDE step on the MFMA
  const bool sid = a.sigma_identity != 0;                   // sigma = I: the products with sigma drop out (mm path)
  const int mwave = __builtin_amdgcn_readfirstlane(tid >> 6), mc16 = tid & 15, mg4 = (tid & 63) >> 4;
  const int mblocks = (d + 15) >> 4;
  __syncthreads();
  long long acc_prof[64];
  for (int s = 0; s < 64; ++s) acc_prof[s] = 0;
  long long last_sub = 0;
  long long last_tick = PROF ? clock64() : 0;

  // ---- network input [t, x, 0...]  (method.py:65-67): built once; afterwards the end of every step writes the new
  //      state and time straight into it (the padding columns stay zero) ------------------------------------------
  for (int e = tid; e < 16 * ud.in0p; e += nthr) {
    const int r = (int)(((float)e + 0.5f) * inv_in0p), c = e - r * ud.in0p;
    X0[r * tl.s0 + c] = (c == 0) ? a.ts[0] : (c <= d ? XS[r * ds + c - 1] : 0.f);
  }
  // noise of step kk into tile Eb (and the noises output), by threads t_ = 0 .. nt_-1: injected, or Philox with one
  // Box-Muller pair (two consecutive components) per thread
  auto draw_noise = [&](int kk, float* Eb, int t_, int nt_) {
    if (a.noise_in) {
      for (int e = t_; e < 16 * d; e += nt_) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        const int grow = tile_row0 + r;
        const float eps = a.noise_in[((size_t)kk * B + min(grow, B - 1)) * d + i];
        Eb[r * ds + i] = eps;
        if (grow < Bs) a.noises[((size_t)kk * B + grow) * d + i] = eps;
      }
    } else {
      const int np = (d + 1) >> 1;
      const float inv_np = __builtin_amdgcn_rcpf((float)np);
      for (int q = t_; q < 16 * np; q += nt_) {
        const int r = (int)(((float)q + 0.5f) * inv_np), pr = q - r * np;
        const int grow = tile_row0 + r;
        float z0, z1;
        philox_normal2(key_seed, key_offset, (uint32_t)(a.row0 + grow), (uint32_t)kk, pr >> 1, pr & 1, z0, z1);
        const int i = 2 * pr;
        Eb[r * ds + i] = z0;
        if (grow < Bs) a.noises[((size_t)kk * B + grow) * d + i] = z0;
        if (i + 1 < d) {
          Eb[r * ds + i + 1] = z1;
          if (grow < Bs) a.noises[((size_t)kk * B + grow) * d + i + 1] = z1;
        }
      }
    }
  };
  // Philox in two halves: the counter-mode block during the Euler-Maruyama phase of step kk - 1, Box-Muller and the
  // stores during its cost phase (the words wait in registers across the barrier between them).  Each half is about
  // as long as what the other waves do in that phase; in one piece the generator was the Euler-Maruyama phase's
  // critical path.  (Tiles too wide for the rule below take the one-piece draw_noise.)
  const int nz_t = (nthr >= 512) ? tid - 256 : tid;          // noise duty: the upper four waves when there are eight
  const int nz_n = (nthr >= 512) ? nthr - 256 : nthr;
  // Unit of work: one Box-Muller pair per thread when that covers the tile in one pass (d <= 32 on 256 threads: two
  // threads share a block's latency), otherwise whole blocks (four components), up to two per thread.
  const int nz_pairs = 16 * ((d + 1) >> 1), nz_quads = 16 * ((d + 3) >> 2);
  const bool nz_by_pair = nz_pairs <= nz_n;
  const bool nz_split = !a.noise_in && (nz_by_pair || nz_quads <= 2 * nz_n);
  uint32_t pw[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  auto noise_words = [&](int kk) {
    if (nz_by_pair) {
      const int np = (d + 1) >> 1;
      if (nz_t < nz_pairs) {
        const int r = (int)(((float)nz_t + 0.5f) * __builtin_amdgcn_rcpf((float)np)), pr = nz_t - r * np;
        philox_pair_words(key_seed, key_offset, (uint32_t)(a.row0 + tile_row0 + r), (uint32_t)kk, pr >> 1, pr & 1,
                          pw[0][0], pw[0][1]);
      }
    } else {
      const int nq = (d + 3) >> 2;
      const float inv_nq = __builtin_amdgcn_rcpf((float)nq);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = nz_t + it * nz_n;
        if (q < nz_quads) {
          const int r = (int)(((float)q + 0.5f) * inv_nq), b = q - r * nq;
          philox4x32_10((uint32_t)(a.row0 + tile_row0 + r), (uint32_t)kk, (uint32_t)b, (uint32_t)key_offset,
                        (uint32_t)key_seed, (uint32_t)(key_seed >> 32), pw[it]);
        }
      }
    }
  };
  auto noise_put = [&](int kk, float* Eb, int r, int i, float z) {
    if (i < d) {
      Eb[r * ds + i] = z;
      if (tile_row0 + r < Bs) a.noises[((size_t)kk * B + tile_row0 + r) * d + i] = z;
    }
  };
  // which: 0 = the first Box-Muller pair of each block, 1 = the second, 2 = both (by-pair mode: all in call 0 or 2)
  auto noise_finish = [&](int kk, float* Eb, int which) {
    if (nz_by_pair) {
      if (which == 1) return;
      const int np = (d + 1) >> 1;
      if (nz_t < nz_pairs) {
        const int r = (int)(((float)nz_t + 0.5f) * __builtin_amdgcn_rcpf((float)np)), pr = nz_t - r * np;
        float z0, z1;
        box_muller_pair(pw[0][0], pw[0][1], z0, z1);
        noise_put(kk, Eb, r, 2 * pr, z0);
        noise_put(kk, Eb, r, 2 * pr + 1, z1);
      }
    } else {
      const int nq = (d + 3) >> 2;
      const float inv_nq = __builtin_amdgcn_rcpf((float)nq);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = nz_t + it * nz_n;
        if (q < nz_quads) {
          const int r = (int)(((float)q + 0.5f) * inv_nq), b = q - r * nq;
          float z0, z1;
          if (which != 1) {
            box_muller_pair(pw[it][0], pw[it][1], z0, z1);
            noise_put(kk, Eb, r, 4 * b, z0);
            noise_put(kk, Eb, r, 4 * b + 1, z1);
          }
          if (which != 0) {
            box_muller_pair(pw[it][2], pw[it][3], z0, z1);
            noise_put(kk, Eb, r, 4 * b + 2, z0);
            noise_put(kk, Eb, r, 4 * b + 3, z1);
          }
        }
      }
    }
  };
  draw_noise(0, E0, tid, nthr);
  __syncthreads();

  for (int k = 0; k < K; ++k) {
    const float t0 = a.ts[k], t1 = a.ts[k + 1];
    const float dt = t1 - t0;                 // utils.py:38
    const float sq_ldt = sqrtf(a.lmbd * dt);  // utils.py:47
    SOCMX_TICK(0)
    last_sub = last_tick;
    auto hook = [&](int slot) {
      if (PROF) {
        const long long now_ = clock64();
        if (slot < 16) { acc_prof[slot] += now_ - last_tick; last_tick = now_; last_sub = now_; }
        else { acc_prof[slot] += now_ - last_sub; last_sub = now_; }   // sub-stage split (wave 0's view)
      }
    };
    if constexpr (kStatic) unet_tile_forward_static<NW, NET>(a.packed, lds, carry, hook);   // GV = nabla_V(t, x)
    else unet_tile_forward<NW>(a.packed, a.prog, a.t, lds, carry, hook);  // GV = nabla_V(t,x)

    if (a.nabla_v)
      for (int e = tid; e < 16 * d; e += nthr) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        if (tile_row0 + r < Bs) a.nabla_v[((size_t)k * B + tile_row0 + r) * d + i] = GV[r * tl.sg + i];
      }
    // ---- control u = -sigma^T nabla_V (method.py:68-72); the step's noise (utils.py:39) is already in E ---
    float* E = (k & 1) ? E1 : E0;
    float* En = (k & 1) ? E0 : E1;
    const bool ctrl_in_em = mm && sid;               // sigma = I: u = -nabla_V is formed where the update needs it
    if (ctrl_in_em) {
    } else if (mm) {
      if (k > 0 && nz_split && nz_t >= 0) noise_finish(k, E, 1);     // second half of this step's noise (see below)
      // d >= 16: the (16 rows x d) . (d x d) products run on the MFMA with both operands read from LDS
      // (one ds_read_b32 pair per MFMA instead of two LDS reads per multiply-add)
      for (int ib = mwave; ib < mblocks; ib += NW) {
        const int i = ib * 16 + mc16;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < d; j0 += 16) {         // four k-steps per trip: eight LDS reads in flight, then four MFMAs
          float av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u + mg4;
            const bool okj = j < d;
            av[u] = lds_sel(sig + min(j, d - 1) * ds + min(i, d - 1), okj);   // (sigma^T)[i][j]; rows j >= d: none
            bv[u] = GV[mc16 * tl.sg + j];               // the network's padded outputs are exact zeros
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        }
        const int grow = tile_row0 + mc16;            // D: lane holds rows i = ib*16 + 4*g4 + rr of batch column c16
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int io = ib * 16 + 4 * mg4 + rr;
          if (io < d) {
            const float u = -acc[rr];
            U[mc16 * ds + io] = u;
            if (grow < Bs) a.controls[((size_t)k * B + grow) * d + io] = u;
          }
        }
      }
    } else {
      for (int e = tid; e < 16 * d; e += nthr) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        float s = 0.f;
        for (int j = 0; j < d; ++j) s += sig[j * ds + i] * GV[r * tl.sg + j];
        const float u = -s;
        const int grow = tile_row0 + r;
        U[r * ds + i] = u;
        if (grow < Bs) a.controls[((size_t)k * B + grow) * d + i] = u;
      }
    }
    if (!ctrl_in_em) __syncthreads();
    SOCMX_TICK(7)

    // ---- Euler-Maruyama update (utils.py:45-48) ------------------------------------------
    // the noise of step k + 1 is started here, by the upper four waves when there are eight: up to d = 64 they have
    // no part in the MFMA products below
    if (k + 1 < K && nz_t >= 0) {
      if (nz_split) noise_words(k + 1);
      else draw_noise(k + 1, En, nz_t, nz_n);
    }
    if (mm && sid && !is_ou) {
      // sigma = I and an elementwise drift (double_well, molecular_dynamics): nothing to multiply
      for (int e = tid; e < 16 * d; e += nthr) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        const float bi = drift_i(kind, d, i, XS + r * ds, A_l, KAP, ds);
        const float u = -GV[r * tl.sg + i];
        U[r * ds + i] = u;
        if (tile_row0 + r < Bs) a.controls[((size_t)k * B + tile_row0 + r) * d + i] = u;
        const float upd = (bi + u) * dt + sq_ldt * E[r * ds + i];
        if (STOPPING) UP[r * ds + i] = upd;
        XN[r * ds + i] = XS[r * ds + i] + ST[r] * upd;
      }
    } else if (mm) {
      // (the two uniform flags are compile-time inside the loop: a run-time test per operand splits the loads into
      //  basic blocks that each wait for their own LDS round trip)
      auto em_mfma = [&](auto sid_c, auto ou_c) {
        constexpr bool SID = decltype(sid_c)::value, OU = decltype(ou_c)::value;
        for (int ib = mwave; ib < mblocks; ib += NW) {
          const int i = ib * 16 + mc16, ic = min(i, d - 1);
          f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};   // A x + sigma u ; sigma eps
          for (int j0 = 0; j0 < d; j0 += 16) {         // four k-steps per trip (LDS reads first, MFMAs after)
            float sv[4], aa[4], bu[4], be[4], bx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int j = j0 + 4 * u + mg4;            // < stride - 1: zero columns past d (rows i >= d of the
              if constexpr (!SID) {                      //  result repeat row d-1 and are not stored)
                sv[u] = sig[ic * ds + j];
                bu[u] = U[mc16 * ds + j];
                be[u] = E[mc16 * ds + j];
              }
              if constexpr (OU) {
                aa[u] = A_l[ic * ds + j];
                bx[u] = XS[mc16 * ds + j];
              }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if constexpr (OU) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[u], bx[u], acc1, 0, 0, 0);
              if constexpr (!SID) {                      // sigma = I: sigma u = u and sigma eps = eps, added below
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[u], be[u], acc2, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(sv[u], bu[u], acc1, 0, 0, 0);
              }
            }
          }
          SOCMX_TICK(10)
          const float st = ST[mc16];
          float xs[4], uo[4], eo[4], bi[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {               // every read first, unconditionally (clamped), then the stores
            const int ioc = min(ib * 16 + 4 * mg4 + rr, d - 1), e = mc16 * ds + ioc;
            xs[rr] = XS[e];
            if constexpr (SID) { uo[rr] = -GV[mc16 * tl.sg + ioc]; eo[rr] = E[e]; }
            bi[rr] = OU ? 0.f : drift_i(kind, d, ioc, XS + mc16 * ds, A_l, KAP, ds);
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int io = ib * 16 + 4 * mg4 + rr;
            const float a1 = SID ? acc1[rr] + uo[rr] : acc1[rr], a2 = SID ? eo[rr] : acc2[rr];
            const float upd = (bi[rr] + a1) * dt + sq_ldt * a2;
            if (io < d) {
              if (STOPPING) UP[mc16 * ds + io] = upd;
              XN[mc16 * ds + io] = xs[rr] + st * upd;
              if constexpr (SID) {
                U[mc16 * ds + io] = uo[rr];
                if (tile_row0 + mc16 < Bs) a.controls[((size_t)k * B + tile_row0 + mc16) * d + io] = uo[rr];
              }
            }
          }
        }
      };
      if (sid) em_mfma(std::true_type{}, std::true_type{});          // (sigma = I without A x took the branch above)
      else if (is_ou) em_mfma(std::false_type{}, std::true_type{});
      else em_mfma(std::false_type{}, std::false_type{});
    } else {
      for (int e = tid; e < 16 * d; e += nthr) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        const float bi = drift_i(kind, d, i, XS + r * ds, A_l, KAP, ds);
        float su = 0.f, se = 0.f;
        for (int j = 0; j < d; ++j) {
          su += sig[i * ds + j] * U[r * ds + j];
          se += sig[i * ds + j] * E[r * ds + j];
        }
        const float upd = (bi + su) * dt + sq_ldt * se;
        if (STOPPING) UP[r * ds + i] = upd;
        XN[r * ds + i] = XS[r * ds + i] + ST[r] * upd;
      }
    }
    __syncthreads();
    SOCMX_TICK(8)

    const float* XE = XN;  // state at the end of the step
    if (STOPPING) {
      // utils.py:42-44, 49-75 with Phi(x) = -x_0 (molecular_dynamics.py:94-99)
      for (int e = tid; e < 16 * d; e += nthr) {
        const int r = SOCMX_DIV_D(e), i = e - r * d;
        const float phi_b = -XS[r * ds], phi_a = -XN[r * ds];
        const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
        const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
        const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
        const float xi = js * (XS[r * ds + i] + fr * ST[r] * UP[r * ds + i]) + (1.f - js) * XN[r * ds + i];
        XF[r * ds + i] = xi;
        if (i == 0) {
          FD[r] = js * (fr * fr) * dt + ns * dt;   // utils.py:70-72 (step_fraction squared)
          SN[r] = (-xi > 0.f) ? 1.f : 0.f;         // utils.py:74
        }
      }
      XE = XF;
      __syncthreads();
    }

    // ---- running cost / log path weights (utils.py:82-99), f at the NEW state, OLD time ----
    // u'u and u'eps: 16 lanes per row (threads 0..255), each lane takes components l, l+16, ...; row totals by DPP;
    // lane 0 of the row group owns the row's accumulators.
    // x'Px (OU_quadratic.py:66-69): P x on the MFMA like the products above, block ib of P's rows on wave NW-1-ib
    // (the upper waves have nothing else here); each lane folds its four (P x)_i x_i into one partial in FQ and the
    // owners add the 4 d/16 partials of their row after the barrier that ends the step.
    // (with a separate control phase -- dense sigma -- the second pair of each block waits for it: the upper waves
    //  idle there as well, and two Box-Muller evaluations in a row made them the last to reach this phase's barrier)
    if (k + 1 < K && nz_split && nz_t >= 0) noise_finish(k + 1, En, (mm && !sid) ? 0 : 2);
    const bool quad = kind == SOCMX_OU_QUADRATIC;
    if (quad) {
      for (int ib = NW - 1 - mwave; ib < mblocks; ib += NW) {
        const int i = ib * 16 + mc16, ic = min(i, d - 1);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < d; j0 += 16) {
          float pv[4], xv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + 4 * u + mg4;
            pv[u] = P_l[ic * ds + j];
            xv[u] = XE[mc16 * ds + j];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[u], xv[u], acc, 0, 0, 0);
        }
        float part = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int io = ib * 16 + 4 * mg4 + rr;
          part += lds_sel(XE + mc16 * ds + min(io, d - 1), io < d) * acc[rr];
        }
        FQ[(ib * 4 + mg4) * 16 + mc16] = part;
      }
    }
    SOCMX_TICK(11)
    float step_over_lmbd = 0.f;                        // owners: kept for the x'Px term added behind the barrier
    if (tid < 256) {
      const int r = tid >> 4, l = tid & 15;
      float uu = 0.f, ue = 0.f;
      for (int i = l; i < d; i += 16) {
        uu += U[r * ds + i] * U[r * ds + i];
        ue += U[r * ds + i] * E[r * ds + i];
      }
      uu = row16_sum(uu);
      ue = row16_sum(ue);
      if (l == 0) {
        const float step = STOPPING ? FD[r] : dt;
        const float f = kind == SOCMX_MOLECULAR_DYNAMICS ? 1.f : 0.f;      // molecular_dynamics.py:87; x'Px: below
        step_over_lmbd = step / a.lmbd;
        lpd = lpd + step_over_lmbd * (-f - 0.5f * uu);
        lps = lps + sqrtf(step_over_lmbd) * (-ue);
        const int grow = tile_row0 + r;
        if (grow < Bs) {
          a.frac[(size_t)k * B + grow] = step;
          a.stop_ind[(size_t)(k + 1) * B + grow] = STOPPING ? SN[r] : 1.f;
        }
      }
    }
    SOCMX_TICK(12)
    if (STOPPING && tid < 16) ST[tid] = SN[tid];   // (nobody reads ST in this phase; its readers sit behind barriers)
    for (int e = tid; e < 16 * d; e += nthr) {
      const int r = SOCMX_DIV_D(e), i = e - r * d;
      const float x = XE[r * ds + i];
      XS[r * ds + i] = x;
      X0[r * tl.s0 + 1 + i] = x;                      // next step's network input (the last stage is done with X0)
      if (tile_row0 + r < Bs) a.states[((size_t)(k + 1) * B + tile_row0 + r) * d + i] = x;
    }
    if (tid < 16) X0[tid * tl.s0] = t1;       // (a fresh load here would queue behind this step's stores)
    SOCMX_TICK(14)
    __syncthreads();
    SOCMX_TICK(15)
    if (quad && tid < 256 && (tid & 15) == 0) {
      const int r = tid >> 4;
      float f = 0.f;
      for (int q = 0; q < 4 * mblocks; q += 4) {      // (FQ is next written behind the barriers of the next step)
        const float f0 = FQ[q * 16 + r], f1 = FQ[(q + 1) * 16 + r], f2 = FQ[(q + 2) * 16 + r], f3 = FQ[(q + 3) * 16 + r];
        f += (f0 + f1) + (f2 + f3);
      }
      lpd = lpd - step_over_lmbd * f;
    }
    SOCMX_TICK(9)
  }
  if (PROF && tid == a.prof_wave * 64 && a.prof)
    for (int s = 0; s < 64; ++s) a.prof[(size_t)blockIdx.x * 64 + s] = acc_prof[s];
  if (a.nabla_v) {                          // nabla_V(T, X_K): X0 holds [t_K, x_K] (the loop ended behind a barrier)
    auto nohook = [](int) {};
    if constexpr (kStatic) unet_tile_forward_static<NW, NET>(a.packed, lds, carry, nohook);
    else unet_tile_forward<NW>(a.packed, a.prog, a.t, lds, carry, nohook);
    for (int e = tid; e < 16 * d; e += nthr) {
      const int r = SOCMX_DIV_D(e), i = e - r * d;
      if (tile_row0 + r < Bs) a.nabla_v[((size_t)K * B + tile_row0 + r) * d + i] = GV[r * tl.sg + i];
    }
  }

  // ---- terminal cost (utils.py:101): same 16-lanes-per-row mapping -----------------------------
  if (tid < 256) {
    const int r = tid >> 4, l = tid & 15;
    const float* x = XS + r * ds;
    float part = 0.f;
    if (kind == SOCMX_OU_QUADRATIC) {          // OU_quadratic.py:76-79
      for (int i = l; i < d; i += 16) {
        float qx = 0.f;
        for (int j = 0; j < d; ++j) qx += a.Q[i * d + j] * x[j];
        part += x[i] * qx;
      }
    } else if (kind == SOCMX_OU_LINEAR) {      // OU_linear.py:83-84
      for (int i = l; i < d; i += 16) part += a.omega[i] * x[i];
    } else if (kind == SOCMX_DOUBLE_WELL) {    // double_well.py:75-84
      for (int i = l; i < d; i += 16) {
        const float q = x[i] * x[i] - 1.f;
        part += a.nu[i] * (q * q);
      }
    }
    const float gval = row16_sum(part);
    if (l == 0 && tile_row0 + r < B) {
      a.lpd[tile_row0 + r] = lpd;
      a.lps[tile_row0 + r] = lps;
      a.ltw[tile_row0 + r] = -gval / a.lmbd;
    }
  }
}

// ---- nabla_V on arbitrary rows (method.py:272-278): same tile code, rows from HBM ----------
// ---- small batches: 4-row tiles ----------------------------------------------------------------------------------------
// The same fused step (sigma = I, d <= 15: the FAST form above) on a 4-row tile: the control network runs on
// v_mfma_f32_4x4x1_16b_f32 (socmx_unet.h, unet_tile_forward_static4) from the same packed image.  B / 4 workgroups instead
// of B / 16 -- a training batch of 128 rows works on 32 CUs instead of 8 -- and half the MFMA time per step and tile.
// Wave 0 integrates (thread = (row r, component i), 64 threads), waves 1 and 2 prepare the next steps' noise.
// The 4-row tiles leave ~135 KiB of the CU's LDS free: stage 4's GEMM-1 layer (up_1, 128 KiB at the default widths, a fifth of
// the 689 KB a step streams) is copied there once and read from there every step (r4_resident_stage, socmx_unet.h).
template <int NW, bool STOPPING, class NET>
__global__ __launch_bounds__(NW * 64) void rollout4_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(NET::outp == 16 && NW >= 3, "4-row tile: d <= 15, at least three waves");
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);
  constexpr TileLayout tl = NET::layout4(NW);
  constexpr UnetDesc ud = NET::desc();
  const int tid = threadIdx.x;
  constexpr int nthr = NW * 64;
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int tile_row0 = blockIdx.x * 4;
  float* X0 = lds + tl.x0;
  const int ds = socmx_sde_stride(d);
  const bool is_ou = (kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR);
  const bool is_quad = kind == SOCMX_OU_QUADRATIC;
  float* A_l = lds + a.lds_mats;                        // (d, ds)   OU only
  float* P_l = A_l + (is_ou ? d * ds : 0);              // (d, ds)   OU_quadratic only
  float* NZ = P_l + (is_quad ? d * ds : 0);             // (2, 4, 16) double-buffered noise of steps k, k + 1
  uint32_t* WZ = reinterpret_cast<uint32_t*>(NZ + 128); // (2, 32, 2) Philox words of steps k + 1, k + 2 (one pair of draws each)
  for (float* z = lds + tid; z < NZ + 256; z += nthr) *z = 0.f;       // (tiles' padding columns stay zero for good)
  // the resident layer's fragment image, 16-byte aligned behind the small state
  constexpr int kResidentStage = r4_resident_stage<NW, NET>();
  constexpr StageDesc sd_res = unet_stage_desc(ud, tl, 4);
  constexpr int res_floats = r4_resident_floats<NW, NET>();
  float* RES = lds + ((a.lds_mats + (is_ou ? d * ds : 0) + (is_quad ? d * ds : 0) + 256 + 3) & ~3);
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.packed + sd_res.L1.w_off);
    f32x4* dst = reinterpret_cast<f32x4*>(RES);
    for (int e = tid; e < res_floats / 4; e += nthr) dst[e] = src[e];
  }
  __syncthreads();
  rollout_key_advance(a, key_offset);
  for (int e = tid; e < d * d; e += nthr) {
    const int r = e / d, c = e - r * d;
    if (is_ou) A_l[r * ds + c] = a.A[e];
    if (is_quad) P_l[r * ds + c] = a.P[e];
  }
  unet_load_biases(a.packed, ud, tl, lds, tid, nthr, ud.folded != 0);
  Pre carry = unet_carry_init_static<NW, NET>(a.packed);

  const bool act = tid < 64;
  const int r = (tid >> 4) & 3, i = tid & 15;
  const int ic = min(i, d - 1);
  const bool lane_ok = act && i < d;
  const int grow = tile_row0 + r;
  const bool traj = a.states != nullptr;          // costs-only launches (evaluation bursts) pass no trajectory buffers
  const bool store = lane_ok && grow < B && traj;
  const bool store0 = act && i == 0 && grow < B && traj;
  const size_t rowoff = (size_t)grow * d + i;
  auto gsum = [](float v) { return row16_sum(v); };
  float x = lane_ok ? a.x0[(size_t)min(grow, B - 1) * d + i] : 0.f;
  const float kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
  float stop = 1.f, lpd = 0.f, lps = 0.f;
  if (store) a.states[rowoff] = x;
  if (store0) a.stop_ind[grow] = 1.f;
  if (act) {
    if (i < 15) X0[r * tl.s0 + 1 + i] = x;   // columns 1..15; lanes i >= d hold x = 0
    if (i == 0) X0[r * tl.s0] = a.ts[0];
  }
  // Noise, one step ahead and in two halves on two otherwise idle waves: Philox4x32-10 and Box-Muller are ~1.1k cycles each
  // for a thread, a chain of dependent integer multiplies and a log / sqrt / sincos -- in one piece on one wave they outlast
  // the integrating wave's step (1.4k cycles) and every wave waits for them at the barrier.  Wave 1 computes the counter
  // block's words of step k + 2, wave 2 turns the words of step k + 1 (from the previous step's phase) into draws; thread
  // p < 32 of either owns the pair of draws (2q, 2q + 1) of row p >> 3, q = p & 7.  (Injected noise: wave 2 fetches it.)
  const int pp = tid & 63, prow = (pp >> 3) & 3, pq = pp & 7;
  const bool w_words = tid >= 64 && tid < 96 && !a.noise_in, w_draws = tid >= 128 && tid < 160;
  const int pgrow = tile_row0 + prow;
  auto words = [&](int k) {
    if (!w_words || k >= K) return;
    uint32_t wa, wb;
    philox_pair_words(key_seed, key_offset, (uint32_t)(a.row0 + pgrow), (uint32_t)k, pq >> 1, pq & 1, wa, wb);
    WZ[((k & 1) * 32 + pp) * 2] = wa;
    WZ[((k & 1) * 32 + pp) * 2 + 1] = wb;
  };
  auto draws = [&](int k) {
    if (!w_draws || k >= K) return;
    float z0 = 0.f, z1 = 0.f;
    const int c0 = 2 * pq;
    if (a.noise_in) {
      const float* src = a.noise_in + ((size_t)k * B + min(pgrow, B - 1)) * d;
      if (c0 < d) z0 = src[c0];
      if (c0 + 1 < d) z1 = src[c0 + 1];
    } else {
      box_muller_pair(WZ[((k & 1) * 32 + pp) * 2], WZ[((k & 1) * 32 + pp) * 2 + 1], z0, z1);
    }
    NZ[(k & 1) * 64 + prow * 16 + c0] = c0 < d ? z0 : 0.f;
    NZ[(k & 1) * 64 + prow * 16 + c0 + 1] = c0 + 1 < d ? z1 : 0.f;
  };
  words(0);
  __syncthreads();
  draws(0);
  words(1);
  // (the time grid one step ahead: a scalar load at the top of a step would be waited for at the step's first LDS wait)
  float t_cur = a.ts[0], t_nxt = a.ts[1];
  for (int k = 0; k < K; ++k) {
    const float t0 = t_cur, t1 = t_nxt;
    t_cur = t_nxt;
    t_nxt = a.ts[min(k + 2, K)];
    const float dt = t1 - t0;                 // utils.py:38
    const float sq_ldt = sqrtf(a.lmbd * dt);  // utils.py:47
    // (without a stopping time the step is dt for every row: its quotient and root are formed here, off the chain that follows
    //  the network -- an IEEE division and a square root are ~25 dependent instructions)
    const float dt_over_lmbd = dt / a.lmbd, sqrt_dt_over_lmbd = sqrtf(dt_over_lmbd);
    __syncthreads();
    float gv = 0.f;                           // nabla_V[r][i] of this thread
    unet_tile_forward_static4<NW, NET, kResidentStage>(a.packed, lds, carry, &gv, RES);   // last stage -> register, no barrier
    if (store && a.nabla_v) a.nabla_v[(size_t)k * B * d + rowoff] = gv;
    if (act) {
      const float u = lane_ok ? -gv : 0.f;                             // u = -sigma^T nabla_V, sigma = I
      const float eps = NZ[(k & 1) * 64 + r * 16 + i];                  // drawn during the previous step
      float bi;
      if (is_ou) {                                                      // b = A x
        bi = 0.f;
        for (int j = 0; j < d; ++j) bi += A_l[ic * ds + j] * __shfl(x, j, 16);
        if (!lane_ok) bi = 0.f;
      } else {
        bi = -2.f * kap * (x * x - 1.f) * 2.f * x;                      // double_well.py:44-48
      }
      const float upd = (bi + u) * dt + sq_ldt * eps;                   // utils.py:45-47
      const float xn = x + stop * upd;                                  // utils.py:48
      float xe = xn, step = dt, stop_new = 1.f;
      if (STOPPING) {                                                   // utils.py:42-44, 49-75; Phi = -x_0
        const float phi_b = -__shfl(x, 0, 16), phi_a = -__shfl(xn, 0, 16);
        const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
        const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
        const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
        xe = js * (x + fr * stop * upd) + (1.f - js) * xn;
        step = js * (fr * fr) * dt + ns * dt;                           // step_fraction squared (utils.py:70-72)
        stop_new = (-__shfl(xe, 0, 16) > 0.f) ? 1.f : 0.f;
      }
      float f = 0.f;                                                    // f at the NEW state, OLD time (utils.py:92-96)
      if (kind == SOCMX_OU_QUADRATIC) {
        float px = 0.f;
        for (int j = 0; j < d; ++j) px += P_l[ic * ds + j] * __shfl(xe, j, 16);
        f = gsum(lane_ok ? xe * px : 0.f);
      } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
        f = 1.f;
      }
      const float uu = gsum(u * u), ue = gsum(u * eps);
      const float sol = STOPPING ? step / a.lmbd : dt_over_lmbd;
      const float ssol = STOPPING ? sqrtf(sol) : sqrt_dt_over_lmbd;
      lpd = lpd + sol * (-f - 0.5f * uu);
      lps = lps + ssol * (-ue);
      if (store) {
        a.controls[(size_t)k * B * d + rowoff] = u;
        a.noises[(size_t)k * B * d + rowoff] = eps;
        a.states[(size_t)(k + 1) * B * d + rowoff] = xe;
      }
      if (store0) {
        a.frac[(size_t)k * B + grow] = step;
        a.stop_ind[(size_t)(k + 1) * B + grow] = STOPPING ? stop_new : 1.f;
      }
      x = lane_ok ? xe : 0.f;
      if (STOPPING) stop = stop_new;
      if (i < 15) X0[r * tl.s0 + 1 + i] = x;                            // next step's network input [t, x, 0..]
      if (i == 0) X0[r * tl.s0] = t1;
    }
    words(k + 2);
    draws(k + 1);
  }
  if (a.nabla_v) {                        // nabla_V(T, X_K): X0 already holds [t_K, x_K] (written at the end of the last step)
    __syncthreads();
    float gv = 0.f;
    unet_tile_forward_static4<NW, NET, kResidentStage>(a.packed, lds, carry, &gv, RES);
    if (store) a.nabla_v[(size_t)K * B * d + rowoff] = gv;
  }
  if (act) {                                                            // terminal cost (utils.py:101)
    float gval = 0.f;
    if (kind == SOCMX_OU_QUADRATIC) {
      float qx = 0.f;
      for (int j = 0; j < d; ++j) qx += a.Q[ic * d + j] * __shfl(x, j, 16);
      gval = gsum(lane_ok ? x * qx : 0.f);
    } else if (kind == SOCMX_OU_LINEAR) {
      gval = gsum(lane_ok ? a.omega[ic] * x : 0.f);
    } else if (kind == SOCMX_DOUBLE_WELL) {
      const float q = x * x - 1.f;
      gval = gsum(lane_ok ? a.nu[ic] * (q * q) : 0.f);
    }
    if (i == 0 && grow < B) {
      a.lpd[grow] = lpd;
      a.lps[grow] = lps;
      a.ltw[grow] = -gval / a.lmbd;
    }
  }
}

// sum over the 64 lanes of a wave; every lane ends with the total (row sums by DPP, rows by the gfx950 row / half swaps, cf. kg_reduce)
__device__ __forceinline__ float wave64_sum(float v) {
  float a = row16_sum(v), t;
  asm volatile(
      "s_nop 1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %1\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_mov_b32 %1, %0\n\t"
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_add_f32 %0, %0, %1"
      : "+v"(a), "=&v"(t));
  return a;
}

constexpr int kRow4Stride = 68;   // floats per row vector of the general 4-row step (d <= 64, 16-byte aligned rows)

// y_i = sum_j M[i][j] v_j for lane i (row-major M with row stride ds in LDS, v a zero-padded row vector in LDS)
__device__ __forceinline__ float matvec_rows(const float* M, int ds, int ic, const float* v, int d) {
  float s0 = 0.f, s1 = 0.f;
  const float* m = M + ic * ds;
  for (int j = 0; j < d; j += 4) {                      // (columns d .. d+3 of M's rows: the next row or the zero pad, times 0)
    const f32x4 x = *reinterpret_cast<const f32x4*>(v + j);
    s0 += m[j] * x[0] + m[j + 2] * x[2];
    s1 += m[j + 1] * x[1] + m[j + 3] * x[3];
  }
  return s0 + s1;
}
// y_i = sum_j M[j][i] v_j (the transpose's rows: lane i walks column i)
__device__ __forceinline__ float matvec_cols(const float* M, int ds, int ic, const float* v, int d) {
  float s0 = 0.f, s1 = 0.f;
  const float* m = M + ic;
  for (int j = 0; j < d; j += 4) {                      // (rows d .. d+3 are zero: the matrices are stored with padded rows)
    const f32x4 x = *reinterpret_cast<const f32x4*>(v + j);
    s0 += m[j * ds] * x[0] + m[(j + 2) * ds] * x[2];
    s1 += m[(j + 1) * ds] * x[1] + m[(j + 3) * ds] * x[3];
  }
  return s0 + s1;
}

// The general step (any sigma, d <= 64) on a 4-row tile: wave r < 4 owns row r, lane i component i -- the matrix
// products are 64-term dot products per lane with the row vectors broadcast from LDS (4 rows are no MFMA tile), row sums
// are wave sums, and nothing inside the step needs a workgroup barrier.  Waves 4..7 draw the next step's noise.
template <int NW, bool STOPPING, class NET>
__global__ __launch_bounds__(NW * 64) void rollout4g_kernel(const RolloutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(NW == 8, "four integrating waves, four noise waves");
  uint64_t key_seed, key_offset;
  rollout_key(a, key_seed, key_offset);
  constexpr TileLayout tl = NET::layout4(NW);
  constexpr UnetDesc ud = NET::desc();
  constexpr int RS = kRow4Stride;
  const int tid = threadIdx.x;
  constexpr int nthr = NW * 64;
  const int d = a.d, B = a.B, K = a.K, kind = a.kind;
  const int tile_row0 = blockIdx.x * 4;
  float* X0 = lds + tl.x0;
  const float* GV = lds + tl.gv;
  const int ds = socmx_sde_stride(d), dp = (d + 3) & ~3;
  const bool is_ou = (kind == SOCMX_OU_QUADRATIC || kind == SOCMX_OU_LINEAR);
  const bool is_quad = kind == SOCMX_OU_QUADRATIC;
  const bool sid = a.sigma_identity != 0;
  float* sig = lds + a.lds_mats;                        // (dp, ds)  dense sigma only
  float* A_l = sig + (sid ? 0 : dp * ds);               // (dp, ds)  OU only
  float* P_l = A_l + (is_ou ? dp * ds : 0);             // (dp, ds)  OU_quadratic only
  const int mats4 = (sid ? 0 : 1) + (is_ou ? 1 : 0) + (is_quad ? 1 : 0);
  float* XS = sig + ((mats4 * dp * ds + 3) & ~3);       // (4, RS) row vectors, 16-byte aligned: state ...
  float* XN = XS + 4 * RS;                              // ... state at the end of the step (x' P x)
  float* U = XN + 4 * RS;                               // ... control
  float* E0 = U + 4 * RS;                               // ... noise of the even steps
  float* E1 = E0 + 4 * RS;                              // ... and of the odd ones (drawn one step ahead)
  for (float* z = lds + tid; z < E1 + 4 * RS; z += nthr) *z = 0.f;
  __syncthreads();
  rollout_key_advance(a, key_offset);
  for (int e = tid; e < d * d; e += nthr) {
    const int r = e / d, c = e - r * d;
    if (!sid) sig[r * ds + c] = a.sigma[e];
    if (is_ou) A_l[r * ds + c] = a.A[e];
    if (is_quad) P_l[r * ds + c] = a.P[e];
  }
  unet_load_biases(a.packed, ud, tl, lds, tid, nthr, ud.folded != 0);
  Pre carry = unet_carry_init_static<NW, NET>(a.packed);

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = tid & 63;
  const bool act = wave < 4;
  const int r = wave & 3;
  const int ic = min(i, d - 1);
  const bool lane_ok = act && i < d;
  const int grow = tile_row0 + r;
  const bool traj = a.states != nullptr;          // costs-only launches (evaluation bursts) pass no trajectory buffers
  const bool store = lane_ok && grow < B && traj;
  const bool store0 = act && i == 0 && grow < B && traj;
  const size_t rowoff = (size_t)grow * d + i;
  float x = lane_ok ? a.x0[(size_t)min(grow, B - 1) * d + i] : 0.f;
  const float kap = (lane_ok && !is_ou) ? a.kappa[i] : 0.f;
  float stop = 1.f, lpd = 0.f, lps = 0.f;
  if (store) a.states[rowoff] = x;
  if (store0) a.stop_ind[grow] = 1.f;
  if (lane_ok) { X0[r * tl.s0 + 1 + i] = x; XS[r * RS + i] = x; }
  if (act && i == 0) X0[r * tl.s0] = a.ts[0];
  // noise: lane p of wave 4 + r draws the Box-Muller pair (2p, 2p + 1) of row r (the 16-row kernel's draws)
  const bool producer = !act && i < ((d + 1) >> 1);
  auto draw = [&](int k, float* Eb) {
    if (!producer || k >= K) return;
    float z0, z1;
    const int c0 = 2 * i;
    if (a.noise_in) {
      const float* src = a.noise_in + ((size_t)k * B + min(grow, B - 1)) * d;
      z0 = src[c0];
      z1 = c0 + 1 < d ? src[c0 + 1] : 0.f;
    } else {
      philox_normal2(key_seed, key_offset, (uint32_t)(a.row0 + grow), (uint32_t)k, i >> 1, i & 1, z0, z1);
    }
    Eb[r * RS + c0] = z0;
    if (c0 + 1 < d) Eb[r * RS + c0 + 1] = z1;
    if (grow < B && traj) {
      float* dst = a.noises + ((size_t)k * B + grow) * d;
      dst[c0] = z0;
      if (c0 + 1 < d) dst[c0 + 1] = z1;
    }
  };
  draw(0, E0);
  float t_cur = a.ts[0], t_nxt = a.ts[1];     // (the time grid one step ahead, see rollout4_kernel)
  for (int k = 0; k < K; ++k) {
    const float t0 = t_cur, t1 = t_nxt;
    t_cur = t_nxt;
    t_nxt = a.ts[min(k + 2, K)];
    const float dt = t1 - t0;                 // utils.py:38
    const float sq_ldt = sqrtf(a.lmbd * dt);  // utils.py:47
    const float dt_over_lmbd = dt / a.lmbd, sqrt_dt_over_lmbd = sqrtf(dt_over_lmbd);   // (off the post-network chain, see rollout4_kernel)
    __syncthreads();
    unet_tile_forward_static4<NW, NET>(a.packed, lds, carry);                // GV = nabla_V(t, x); ends behind a barrier
    const float* E = (k & 1) ? E1 : E0;
    float* En = (k & 1) ? E0 : E1;
    if (act) {
      const float gv = GV[r * tl.sg + ic];
      if (store && a.nabla_v) a.nabla_v[(size_t)k * B * d + rowoff] = gv;
      // u = -sigma^T nabla_V (method.py:68-72)
      float u = sid ? -gv : -matvec_cols(sig, ds, ic, GV + r * tl.sg, d);
      if (!lane_ok) u = 0.f;
      const float eps = E[r * RS + ic];                                  // drawn during the previous step (0 past d)
      float su = u, se = lane_ok ? eps : 0.f;
      if (!sid) {
        U[r * RS + i] = u;                                               // (lanes >= d write the zero pad)
        __builtin_amdgcn_wave_barrier();
        su = matvec_rows(sig, ds, ic, U + r * RS, d);
        se = matvec_rows(sig, ds, ic, E + r * RS, d);
      }
      float bi;
      if (is_ou) bi = matvec_rows(A_l, ds, ic, XS + r * RS, d);          // b = A x
      else bi = -2.f * kap * (x * x - 1.f) * 2.f * x;                    // double_well.py:44-48
      const float upd = lane_ok ? (bi + su) * dt + sq_ldt * se : 0.f;    // utils.py:45-47
      const float xn = x + stop * upd;                                   // utils.py:48
      float xe = xn, step = dt, stop_new = 1.f;
      if (STOPPING) {                                                    // utils.py:42-44, 49-75; Phi = -x_0
        const float phi_b = -__shfl(x, 0, 64), phi_a = -__shfl(xn, 0, 64);
        const float ns = (phi_b > 0.f && phi_a > 0.f) ? 1.f : 0.f;
        const float js = (phi_b > 0.f && phi_a < 0.f) ? 1.f : 0.f;
        const float fr = js * (phi_b / (phi_b - phi_a + 1e-6f) + 1e-6f);
        xe = js * (x + fr * stop * upd) + (1.f - js) * xn;
        step = js * (fr * fr) * dt + ns * dt;                            // step_fraction squared (utils.py:70-72)
        stop_new = (-__shfl(xe, 0, 64) > 0.f) ? 1.f : 0.f;
      }
      float f = 0.f;                                                     // f at the NEW state, OLD time (utils.py:92-96)
      if (is_quad) {
        XN[r * RS + i] = lane_ok ? xe : 0.f;
        __builtin_amdgcn_wave_barrier();
        const float px = matvec_rows(P_l, ds, ic, XN + r * RS, d);
        f = wave64_sum(lane_ok ? xe * px : 0.f);
      } else if (kind == SOCMX_MOLECULAR_DYNAMICS) {
        f = 1.f;
      }
      const float uu = wave64_sum(u * u), ue = wave64_sum(lane_ok ? u * eps : 0.f);
      const float sol = STOPPING ? step / a.lmbd : dt_over_lmbd;
      const float ssol = STOPPING ? sqrtf(sol) : sqrt_dt_over_lmbd;
      lpd = lpd + sol * (-f - 0.5f * uu);
      lps = lps + ssol * (-ue);
      if (store) {
        a.controls[(size_t)k * B * d + rowoff] = u;
        a.states[(size_t)(k + 1) * B * d + rowoff] = xe;
      }
      if (store0) {
        a.frac[(size_t)k * B + grow] = step;
        a.stop_ind[(size_t)(k + 1) * B + grow] = STOPPING ? stop_new : 1.f;
      }
      x = lane_ok ? xe : 0.f;
      if (STOPPING) stop = stop_new;
      if (lane_ok) { X0[r * tl.s0 + 1 + i] = x; XS[r * RS + i] = x; }    // next step's network input [t, x, 0..]
      if (i == 0) X0[r * tl.s0] = t1;
    } else {
      draw(k + 1, En);
    }
  }
  if (a.nabla_v) {                        // nabla_V(T, X_K): X0 already holds [t_K, x_K] (written at the end of the last step)
    __syncthreads();
    unet_tile_forward_static4<NW, NET>(a.packed, lds, carry);
    if (store) a.nabla_v[(size_t)K * B * d + rowoff] = GV[r * tl.sg + ic];
  }
  if (act) {                                                            // terminal cost (utils.py:101)
    float part = 0.f;
    if (kind == SOCMX_OU_QUADRATIC) {
      float qx = 0.f;
      for (int j = 0; j < d; ++j) qx += a.Q[ic * d + j] * XS[r * RS + j];
      part = lane_ok ? x * qx : 0.f;
    } else if (kind == SOCMX_OU_LINEAR) {
      part = lane_ok ? a.omega[ic] * x : 0.f;
    } else if (kind == SOCMX_DOUBLE_WELL) {
      const float q = x * x - 1.f;
      part = lane_ok ? a.nu[ic] * (q * q) : 0.f;
    }
    const float gval = wave64_sum(part);
    if (i == 0 && grow < B) {
      a.lpd[grow] = lpd;
      a.lps[grow] = lps;
      a.ltw[grow] = -gval / a.lmbd;
    }
  }
}

struct ForwardArgs {
  UnetDesc u;
  TileLayout t;
  UnetProgram prog;
  const float* packed;
  const float* tx;  // (N, d+1)
  float* out;       // (N, d)
  int64_t N;
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void unet_forward_kernel(const ForwardArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, nthr = NW * 64;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  float* X0 = lds + a.t.x0;
  float* GV = lds + a.t.gv;
  const int in0 = a.u.in0, in0p = a.u.in0p, d = a.u.d;
  for (int e = tid; e < 16 * in0p; e += nthr) {
    const int r = e / in0p, c = e - r * in0p;
    const int64_t grow = min(row0 + r, a.N - 1);
    X0[r * a.t.s0 + c] = (c < in0) ? a.tx[grow * in0 + c] : 0.f;
  }
  unet_load_biases(a.packed, a.u, a.t, lds, tid, nthr, a.u.folded != 0);
  Pre carry = unet_carry_init(a.packed, a.prog);
  __syncthreads();
  unet_tile_forward<NW>(a.packed, a.prog, a.t, lds, carry, [](int) {});
  for (int e = tid; e < 16 * d; e += nthr) {
    const int r = e / d, i = e - r * d;
    if (row0 + r < a.N) a.out[(row0 + r) * d + i] = GV[r * a.t.sg + i];
  }
}

// ---- weight re-layout ---------------------------------------------------------------------------
struct PackArgs {
  UnetDesc u;
  int fin[9], fout[9];
  const float* w[9];
  const float* b[9];
  float* packed;
};

__device__ __forceinline__ void unet_pack_element(const PackArgs& a, const int idx) {
  if (idx >= a.u.total_floats) {
    // behind the nine layers: up_0 once more, as the second half of cat = [up_0 res_1 | up_0] (UnetDesc::cat; the fold kernel
    // writes the first half) -- fragment (nb, kc) of up_0 is fragment (nb, KC + kc) of cat
    const int rel = idx - a.u.total_floats;
    if (rel >= a.u.L[8].in_pad * a.u.L[8].out_pad) return;
    const int i = rel & 3, lane = (rel >> 2) & 63, chunk = rel >> 8;
    const int KC = a.u.L[8].in_pad >> 4;
    const int nb = chunk / KC, kc = chunk - nb * KC;
    const int n = nb * 16 + (lane & 15), kk = kc * 16 + 4 * (lane >> 4) + i;
    float v = 0.f;
    if (n < a.fout[8] && kk < a.fin[8]) v = a.w[8][(size_t)n * a.fin[8] + kk];
    a.packed[a.u.cat.w_off + ((nb * 2 * KC + KC + kc) * 64 + lane) * 4 + i] = v;
    return;
  }
  int l = 8;
  while (l > 0 && idx < a.u.L[l].w_off) --l;
  const LayerDesc L = a.u.L[l];
  float v = 0.f;
  if (idx >= L.b_off) {
    const int n = idx - L.b_off;
    if (n < a.fout[l]) v = a.b[l][n];
  } else {
    const int rel = idx - L.w_off;
    const int i = rel & 3, lane = (rel >> 2) & 63, chunk = rel >> 8;  // chunk = nb*KC + kc
    const int KC = L.in_pad >> 4;
    const int nb = chunk / KC, kc = chunk - nb * KC;
    const int n = nb * 16 + (lane & 15), kk = kc * 16 + 4 * (lane >> 4) + i;
    if (n < a.fout[l] && kk < a.fin[l]) v = a.w[l][(size_t)n * a.fin[l] + kk];
  }
  a.packed[idx] = v;
}
__global__ void unet_pack_kernel(const PackArgs a) { unet_pack_element(a, blockIdx.x * blockDim.x + threadIdx.x); }

// The fold behind the nine layers (UnetDesc::fold): F = up_0 res_1 (outp x h0, fragment-ordered like a layer) and f = up_0 b4.
// It sits in front of every rollout, so it is laid out for latency, not for work: one workgroup per (output unit n, 64 input
// columns), 16 waves that split res_1's rows sixteen ways -- wave p: rows p, p + 16, ... (a coalesced 256-byte read each, all of
// a thread's loads independent), up_0[n][m] a broadcast -- and one LDS pass adds the sixteen partial sums in a fixed order.
// fp64 accumulation: the products are exact, the h0-term sum is rounded once to fp32.  (One thread per element with an
// h0-iteration loop -- the first form -- took 96 us at the default widths: as long as the backward kernel's tail.)
// transposed: the image of F^T as a layer (h0 outputs, outp inputs) -- what the backward chain multiplies dL/d(pre-activation of
// up_0) by in place of res_1^T up_0^T (socmx_unet_bwd.hip); no bias.
constexpr int kFoldCols = 64, kFoldParts = 16;
struct FoldArgs {
  const float* up0;        // (dout, h0) row-major (torch layout)
  const float* res1;       // (h0, h0)
  const float* b4;         // (h0,) res_1's bias; unused when out_b is null
  int h0, dout, in_pad, out_pad;     // F: out_pad x in_pad after padding
  float* out_w;
  float* out_b;            // out_pad floats, or null
  int transposed;
  float* out_cat;          // null, or the image of [F | up_0] (UnetDesc::cat): F's fragment (nb, kc) goes to (nb, kc) of 2 KC
};
__device__ __forceinline__ void unet_fold_block(const FoldArgs& a, const int bx, const int by, double (*red)[kFoldCols]) {
  const int h0 = a.h0, dout = a.dout;
  const int n = bx, col = threadIdx.x & (kFoldCols - 1), part = threadIdx.x / kFoldCols;
  const int kk = by * kFoldCols + col;
  double acc = 0.0;
  if (n < dout && kk < h0) {
    const float* u = a.up0 + (size_t)n * h0;
#pragma unroll 8
    for (int m = part; m < h0; m += kFoldParts) acc = fma((double)u[m], (double)a.res1[(size_t)m * h0 + kk], acc);
  }
  red[part][col] = acc;
  __syncthreads();
  if (part == 0 && kk < a.in_pad) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < kFoldParts; ++q) t += red[q][col];
    if (a.transposed) {          // layer F^T: unit kk, input n -- fragment ((kk >> 4) KC + (n >> 4)), KC = out_pad / 16
      const int KC = a.out_pad >> 4;
      const int chunk = (kk >> 4) * KC + (n >> 4), lane = (kk & 15) + 16 * ((n & 15) >> 2);
      a.out_w[(chunk * 64 + lane) * 4 + (n & 3)] = (float)t;
    } else {
      const int KC = a.in_pad >> 4;
      const int chunk = (n >> 4) * KC + (kk >> 4), lane = (n & 15) + 16 * ((kk & 15) >> 2);
      a.out_w[(chunk * 64 + lane) * 4 + (kk & 3)] = (float)t;
      if (a.out_cat) a.out_cat[((chunk + (n >> 4) * KC) * 64 + lane) * 4 + (kk & 3)] = (float)t;
    }
  }
  if (a.out_b && by == 0 && part == 1) {                // f[n] = sum_m up_0[n][m] b4[m] (one wave, lanes along m)
    double t = 0.0;
    if (n < dout)
      for (int m = col; m < h0; m += kFoldCols) t = fma((double)a.up0[(size_t)n * h0 + m], (double)a.b4[m], t);
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) t += __shfl_xor(t, sh);
    if (col == 0) a.out_b[n] = (float)t;
  }
}
__global__ __launch_bounds__(kFoldCols * kFoldParts) void unet_fold_kernel(const FoldArgs a) {
  __shared__ double red[kFoldParts][kFoldCols];
  unet_fold_block(a, blockIdx.x, blockIdx.y, red);
}
// The re-lay and the fold in ONE launch (they read the raw weights only and write disjoint parts of the image): blocks
// [0, pack_blocks) re-lay 1,024 elements each, the rest are the fold's (output unit, 64 columns) blocks.  One launch less in front
// of every rollout (~6 us of its ~0.4 ms at configs[2]).
__global__ __launch_bounds__(kFoldCols * kFoldParts) void unet_pack_fold_kernel(const PackArgs a, const FoldArgs f, const int pack_blocks) {
  __shared__ double red[kFoldParts][kFoldCols];
  if ((int)blockIdx.x < pack_blocks) {
    unet_pack_element(a, blockIdx.x * (kFoldCols * kFoldParts) + threadIdx.x);
    return;
  }
  const int b = blockIdx.x - pack_blocks;
  unet_fold_block(f, b % f.out_pad, b / f.out_pad, red);
}

// (shared with socmx_unet_bwd.hip: socmx_rollout_common.h declares it)
int unet_fold_launch(const float* up0, const float* res1, const float* b4, int h0, int dout, int in_pad, int out_pad,
                     float* out_w, float* out_b, int transposed, void* stream, float* out_cat) {
  FoldArgs f{up0, res1, b4, h0, dout, in_pad, out_pad, out_w, out_b, transposed, out_cat};
  return launch(unet_fold_kernel, dim3(out_pad, (in_pad + kFoldCols - 1) / kFoldCols), dim3(kFoldCols * kFoldParts), 0, stream, f);
}

// key[1] += inc: the Philox offset of the next keyed rollout (its own tiny node so that every workgroup of the rollout
// before it has read the old value: same-stream order)
__global__ void philox_advance_kernel(uint64_t* key, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) key[1] += inc;
}

}  // namespace socmx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace socmx;

static const int kMaxLdsBytes = 160 * 1024;

static bool dims_ok(int d, const int32_t h[3]) {
  if (d < 1 || d > 1024) return false;
  for (int i = 0; i < 3; ++i)
    if (h[i] < 1 || h[i] > 4096) return false;
  return true;
}

extern "C" int socmx_version(void) { return SOCMX_VERSION; }

extern "C" int socmx_capabilities(char* buf, int cap) {
#define SOCMX_STR2(x) #x
#define SOCMX_STR(x) SOCMX_STR2(x)
  static const char msg[] =
      "socmx 0.1.8; target gfx950 (MI355X, CDNA4); wave64; fp32 v_mfma_f32_16x16x4_f32 control network with the skip res_1 folded through up_0 "
      "(v_mfma_f32_4x4x1_16b_f32 on 4-row tiles for small batches; one row per workgroup on v_fmac_f32_dpp for B <= 256 at d <= 15 and, with sigma = I, 17 <= d <= 31; two 16-row tiles per workgroup for bursts beyond 4096 rows at d <= 31); "
      "Philox4x32-10 noise; static_hdims=" SOCMX_STR(SOCMX_H0P) "," SOCMX_STR(SOCMX_H1P) "," SOCMX_STR(SOCMX_H2P) "; "
      "kernels: rollout, unet_forward, unet_pack, weights_stats, mpairs, socm_target; "
      // what the hand-written kernels take (outside these ranges the entry points return SOCMX_E_LDS / SOCMX_E_DIM and the
      // Python layer routes to torch autograd + library GEMMs on the GPU, with a warning -- never to the CPU)
      "ranges: rollout any arch.hdims, d <= 64 (OU_quadratic) .. 96 (double_well) at the default widths; "
      "control-network backward: architectures whose 16-row tiles fit 160 KiB of LDS (h0 <= 256 at the default depth); "
      "pair-grid network: every d with hdims_M up to [256,256] (d*d beyond an LDS tile: wide form); "
      "stopping-time SOCM kernels: d <= 16; costate (SOCM_adjoint) kernel: d <= 64";
  const int need = (int)sizeof(msg);
  if (buf && cap > 0) {
    const int n = need < cap ? need : cap;
    memcpy(buf, msg, n);
    buf[n - 1] = 0;
  }
  return need;
}

extern "C" size_t socmx_unet_packed_floats(int32_t d, const int32_t hdims[3]) {
  if (!hdims || !dims_ok(d, hdims)) return 0;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  return (size_t)make_unet_desc(d, h).image_floats;
}

extern "C" int socmx_unet_pack_f32(const socmx_unet* net, float* packed, socmx_stream_t stream) {
  if (!net || !packed) return SOCMX_E_NULL;
  if (!dims_ok(net->d, net->hdims)) return SOCMX_E_DIM;
  PackArgs a;
  const int h[3] = {net->hdims[0], net->hdims[1], net->hdims[2]};
  a.u = make_unet_desc(net->d, h);
  unet_layer_dims(net->d, h, a.fin, a.fout);
  for (int l = 0; l < 9; ++l) {
    if (!net->weight[l] || !net->bias[l]) return SOCMX_E_NULL;
    a.w[l] = net->weight[l];
    a.b[l] = net->bias[l];
  }
  a.packed = packed;
  const int threads = kFoldCols * kFoldParts;
  const int pack_blocks = (a.u.total_floats + a.u.L[8].in_pad * a.u.L[8].out_pad + threads - 1) / threads;
  const FoldArgs f{a.w[8], a.w[4], a.b[4], a.fout[4], a.fout[8], a.u.fold.in_pad, a.u.fold.out_pad,
                   packed + a.u.fold.w_off, packed + a.u.fold.b_off, 0, packed + a.u.cat.w_off};
  const int fold_blocks = f.out_pad * ((f.in_pad + kFoldCols - 1) / kFoldCols);
  return launch(unet_pack_fold_kernel, dim3(pack_blocks + fold_blocks), dim3(threads), 0, stream, a, f, pack_blocks);
}

static const int kWaves = 8;  // waves per 16-row tile workgroup (2 per SIMD)

// developer A/B switch: SOCMX_WAVES=4 runs the tile with one wave per SIMD (read once)
static int waves_per_tile() {
  static int w = 0;
  if (!w) {
    const char* e = getenv("SOCMX_WAVES");
    w = (e && e[0] == '4') ? 4 : kWaves;
  }
  return w;
}

extern "C" int socmx_unet_forward_f32(const float* packed, int32_t d, const int32_t hdims[3], const float* tx,
                                      int64_t N, float* out, socmx_stream_t stream) {
  if (!packed || !hdims || !tx || !out) return SOCMX_E_NULL;
  if (!dims_ok(d, hdims) || N < 0) return SOCMX_E_DIM;
  if (N == 0) return 0;
  ForwardArgs a;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  a.u = make_unet_desc(d, h);
  const int nw = waves_per_tile();
  a.t = make_tile_layout(a.u, nw);
  a.prog = make_unet_program(a.u, a.t);
  fill_wave_work(a.prog, nw);
  a.packed = packed; a.tx = tx; a.out = out; a.N = N;
  const size_t lds_bytes = (size_t)a.t.floats * sizeof(float);
  if (lds_bytes > (size_t)kMaxLdsBytes) return SOCMX_E_LDS;
  void (*kern)(const ForwardArgs) = nw == 4 ? unet_forward_kernel<4> : unet_forward_kernel<8>;
  if (const int err = ensure_max_lds(kern)) return err;
  const int64_t blocks = (N + 15) / 16;
  return launch(kern, dim3((unsigned)blocks), dim3(nw * 64), lds_bytes, stream, a);
}

// the 4-row kernel of a constexpr-specialised architecture, if its stages fit the 4-row forms (variant builds of other
// hidden widths may not: then the 16-row kernels run)
#ifndef SOCMX_R4F_WAVES
#define SOCMX_R4F_WAVES 8
#endif
constexpr int kR4FastWaves = SOCMX_R4F_WAVES;   // waves of the 4-row FAST kernel's workgroup
template <class NET>
static bool r4_pick(bool fast_form, bool stopping, void (**k)(const RolloutArgs)) {
  if constexpr (r4_supported<8, NET>()) {
    if constexpr (NET::outp == 16 && r4_supported<kR4FastWaves, NET>()) {
      if (fast_form) {
        *k = stopping ? rollout4_kernel<kR4FastWaves, true, NET> : rollout4_kernel<kR4FastWaves, false, NET>;
        return true;
      }
    }
    *k = stopping ? rollout4g_kernel<8, true, NET> : rollout4g_kernel<8, false, NET>;
    return true;
  } else {
    return false;
  }
}

static int rollout_launch(const socmx_problem* pb, const float* packed_unet, const int32_t hdims[3], const float* x0,
                          const float* ts, int32_t B, int32_t K, float lmbd, uint64_t seed, uint64_t offset,
                          const uint64_t* key_dev, float* nabla_v, uint32_t flags, int64_t row0, const float* noise_in, float* states,
                          float* noises, float* controls,
                          float* stop_indicators, float* fractional_timesteps, float* lpd, float* lps, float* ltw,
                          long long* prof, socmx_stream_t stream, float* act_ws = nullptr, uint32_t* act_rec = nullptr,
                          bool query_saves = false) {
  if (!pb || !packed_unet || !hdims || !x0 || !ts || !lpd || !lps || !ltw || !pb->sigma) return SOCMX_E_NULL;
  // the five trajectory buffers come together or not at all (costs-only launch: the evaluation bursts of
  // utils.py:131-231 read nothing but lpd / lps / ltw)
  const int n_traj = !!states + !!noises + !!controls + !!stop_indicators + !!fractional_timesteps;
  if (n_traj != 0 && n_traj != 5) return SOCMX_E_NULL;
  if (nabla_v && n_traj == 0) return SOCMX_E_NULL;     // (nabla_v travels with the trajectory buffers)
  const int d = pb->d;
  if (!dims_ok(d, hdims) || B < 1 || K < 1) return SOCMX_E_DIM;
  switch (pb->kind) {
    case SOCMX_OU_QUADRATIC: if (!pb->A || !pb->P || !pb->Q) return SOCMX_E_NULL; break;
    case SOCMX_OU_LINEAR: if (!pb->A || !pb->omega) return SOCMX_E_NULL; break;
    case SOCMX_DOUBLE_WELL: if (!pb->kappa || !pb->nu) return SOCMX_E_NULL; break;
    case SOCMX_MOLECULAR_DYNAMICS: if (!pb->kappa) return SOCMX_E_NULL; break;
    default: return SOCMX_E_KIND;
  }
  RolloutArgs a;
  const int h[3] = {hdims[0], hdims[1], hdims[2]};
  a.u = make_unet_desc(d, h);
  const int nw = waves_per_tile();
  a.t = make_tile_layout(a.u, nw);
  a.prog = make_unet_program(a.u, a.t);
  fill_wave_work(a.prog, nw);
  a.kind = pb->kind; a.d = d; a.B = B; a.K = K; a.lmbd = lmbd;
  a.seed = seed; a.offset = offset; a.key_dev = key_dev; a.row0 = row0;
  a.advance_key = (key_dev && (flags & SOCMX_ROLLOUT_ADVANCES_KEY)) ? 1 : 0;
  a.sigma_identity = (pb->flags & SOCMX_SIGMA_IDENTITY) ? 1 : 0;
  a.packed = packed_unet;
  a.sigma = pb->sigma; a.A = pb->A; a.P = pb->P; a.Q = pb->Q; a.omega = pb->omega; a.kappa = pb->kappa; a.nu = pb->nu;
  a.x0 = x0; a.ts = ts; a.noise_in = noise_in;
  a.states = states; a.noises = noises; a.controls = controls; a.stop_ind = stop_indicators;
  a.frac = fractional_timesteps; a.lpd = lpd; a.lps = lps; a.ltw = ltw;
  a.nabla_v = nabla_v;
  a.act_ws = act_ws; a.act_rec = act_rec; a.act_tile_rows = (int)(((int64_t)K + 1) * B);
  a.prof = prof;
  static const int prof_wave = [] { const char* e = getenv("SOCMX_PROF_WAVE"); return (e && e[0] >= '0' && e[0] <= '7') ? e[0] - '0' : 0; }();
  a.prof_wave = prof_wave;
  a.lds_mats = (a.t.floats + 3) & ~3;
  // sigma (+ A for the OU settings, kappa for the others, + P for OU_quadratic); five (16, stride) tiles (+ two with a stopping time);
  // ST/SN/FD; the 512-float noise / partial-sum scratch -- the rule of the kernel's pointer layout
  const bool ou = pb->kind == SOCMX_OU_QUADRATIC || pb->kind == SOCMX_OU_LINEAR;
  const int n_mats = 1 + (ou ? 1 : 0) + (pb->kind == SOCMX_OU_QUADRATIC ? 1 : 0);
  const int n_tiles = 5 + (pb->kind == SOCMX_MOLECULAR_DYNAMICS ? 2 : 0);
  const size_t sds = (size_t)socmx_sde_stride(d);
  const size_t lds_floats = (size_t)a.lds_mats + n_mats * (size_t)d * sds + (ou ? 0 : d) + n_tiles * 16 * sds + 48 + 512;
  const size_t lds_bytes = lds_floats * sizeof(float);
  if (lds_bytes > (size_t)kMaxLdsBytes) return SOCMX_E_LDS;
  const int blocks = (B + 15) / 16;
  const bool stopping = pb->kind == SOCMX_MOLECULAR_DYNAMICS;
  // the reference's default architecture runs the fully specialised instantiation (SOCMX_GENERIC=1 forces the
  // table-driven one, for A/B runs and tests)
  static const bool force_generic = getenv("SOCMX_GENERIC") != nullptr;
  const bool is_default = !force_generic && nw == 8 && a.u.in0p == 16 && a.u.hp[0] == SOCMX_H0P && a.u.hp[1] == SOCMX_H1P &&
                          a.u.hp[2] == SOCMX_H2P && a.u.outp == 16;
  // sigma = I and d <= 15: the SDE step runs in 16-lane groups with one barrier per step (FAST)
  static const bool force_slow = getenv("SOCMX_NOFAST") != nullptr;
  const bool fast = !force_slow && a.sigma_identity && d <= 15;
  void (*kern)(const RolloutArgs);
#define SOCMX_PICK(NWV, NETV)                                                                                   \
  do {                                                                                                          \
    if (fast) {                                                                                                 \
      if (prof) kern = stopping ? rollout_kernel<NWV, true, true, NETV, true> : rollout_kernel<NWV, false, true, NETV, true>;   \
      else kern = stopping ? rollout_kernel<NWV, true, false, NETV, true> : rollout_kernel<NWV, false, false, NETV, true>; \
    } else {                                                                                                    \
      if (prof) kern = stopping ? rollout_kernel<NWV, true, true, NETV, false> : rollout_kernel<NWV, false, true, NETV, false>; \
      else kern = stopping ? rollout_kernel<NWV, true, false, NETV, false> : rollout_kernel<NWV, false, false, NETV, false>; \
    }                                                                                                           \
  } while (0)
  const bool is_wide64 = !force_generic && nw == 8 && a.u.in0p == 80 && a.u.hp[0] == SOCMX_H0P && a.u.hp[1] == SOCMX_H1P &&
                         a.u.hp[2] == SOCMX_H2P && a.u.outp == 64;
  const bool is_wide32 = !force_generic && nw == 8 && a.u.in0p == 32 && a.u.hp[0] == SOCMX_H0P && a.u.hp[1] == SOCMX_H1P &&
                         a.u.hp[2] == SOCMX_H2P && a.u.outp == 32;
  if (is_default) SOCMX_PICK(8, DefaultNet);
  else if (is_wide64) SOCMX_PICK(8, Wide64Net);
  else if (is_wide32) SOCMX_PICK(8, Wide32Net);
  else if (nw == 4) SOCMX_PICK(4, DynamicNet);
  else SOCMX_PICK(8, DynamicNet);
#undef SOCMX_PICK
  // Small batches (at most 64 tiles of 16 rows: a quarter of the CUs; 16 at d >= 32), the constexpr-specialised widths,
  // d <= 64: 4-row tiles, B / 4 workgroups -- rollout4_kernel for sigma = I and d <= 15, rollout4g_kernel for everything else.
  // SOCMX_TILE_ROWS=16 / 4 (developer A/B switch, read once) forces one form.
  static const int force_rows = [] { const char* e = getenv("SOCMX_TILE_ROWS"); return e ? atoi(e) : 0; }();
  // (d >= 32 with more than 256 rows -- one GPU's slice of BASELINE configs[4] -- keeps the 16-row tiles: there the loss side of
  //  an iteration fills the chip for longer than the rollout runs beside it, and the 4-row form's 4x CUs at half the MFMA rate
  //  per MAC cost more than its shorter latency gives: slice iteration 21.2 ms with 4-row tiles, 19.9 with 16-row ones)
  // Training-size batches of the sigma = I, d <= 15 settings at the default widths (BASELINE configs[1], [2]; the README's
  // molecular_dynamics run): ONE ROW per workgroup, the network as matrix-vector products on the VALU with most of the weight
  // image resident in registers / LDS (socmx_rollout1.hip) -- B workgroups instead of B / 4.  SOCMX_TILE_ROWS=1 forces it
  // for any B <= 1024 (tests), =4 / =16 exclude it.
#ifdef SOCMX_R1_PROF
  const bool r1_prof_ok = true;     // (developer build: socmx_rollout_phase_cycles_f32 reaches the one-row kernel's marks)
#else
  const bool r1_prof_ok = !prof;
#endif
  // (a dense sigma at d <= 15 -- the README's Linear OU, d = 10 -- takes the same kernel: sigma u = -(sigma sigma^T) nabla_V is the
  //  one extra product on its serial chain; not with a stopping time, which is built for sigma = I)
  const bool one_row_form = fast || (!force_slow && !a.sigma_identity && d <= 15 && !stopping);
  const bool one_row = is_default && one_row_form && r1_prof_ok && rollout1_available() && (force_rows == 0 || force_rows == 1) &&
                       (B <= 256 || (force_rows == 1 && B <= 1024));
  // The activation export (act_workspace / act_records): the one-row kernel at d <= 15 (with or without a stopping time), whole 16-row tiles of
  // trajectory rows (the slabs of a ragged last tile would hold rows nobody wrote), offsets within 31 bits.  Anything else: refused,
  // and socmx_rollout_saves_activations says so beforehand.
  const bool saves = one_row && (((int64_t)K + 1) * B) % 16 == 0 && ((int64_t)K + 1) * B * 1024 * 4 < ((int64_t)1 << 31);
  if (query_saves) return saves ? 1 : 0;
  if (act_ws && (!saves || !act_rec || !nabla_v)) return SOCMX_E_DIM;
  if (one_row) return rollout1_launch(a, stopping, stream);
  // ... and 17 <= d <= 31 with sigma = I at the default widths (soc.yaml's default d = 20): the same kernel with two components
  // per lane and down_0 / res_0 as a stage of their own on all eight waves
  if (is_wide32 && !force_slow && a.sigma_identity && d >= 16 && d <= 31 && r1_prof_ok && rollout1_wide_available() &&
      (force_rows == 0 || force_rows == 1) && (B <= 256 || (force_rows == 1 && B <= 1024)))
    return rollout1_wide_launch(a, stopping, stream);
  // (... unless the launch is stand-alone -- no SOCMX_ROLLOUT_SHARES_CHIP: nothing beside it to starve -- then the shorter
  //  latency of the 4-row tiles is simply taken: 7.3 -> 4.9 ms at that slice)
  const bool small4 = blocks <= 16 || (blocks <= 64 && (d <= 31 || !(flags & SOCMX_ROLLOUT_SHARES_CHIP)));
  if ((is_default || is_wide32 || is_wide64) && !prof && d <= 64 && force_rows != 16 && (small4 || force_rows == 4)) {
    void (*k4)(const RolloutArgs) = nullptr;
    const bool fast4 = is_default && fast;
    const bool have4 = is_default ? r4_pick<DefaultNet>(fast4, stopping, &k4)
                     : is_wide32 ? r4_pick<Wide32Net>(false, stopping, &k4) : r4_pick<Wide64Net>(false, stopping, &k4);
    if (have4) {
      const int nw4 = fast4 ? kR4FastWaves : 8;
      const TileLayout t4 = is_default ? DefaultNet::layout4(nw4) : is_wide32 ? Wide32Net::layout4(8) : Wide64Net::layout4(8);
      a.lds_mats = (t4.floats + 3) & ~3;
      size_t floats4;
      if (fast4) {
        floats4 = (((size_t)a.lds_mats + (ou ? d * sds : 0) + (pb->kind == SOCMX_OU_QUADRATIC ? d * sds : 0) + 256 + 3) & ~(size_t)3) +
                  (size_t)r4_resident_floats<kR4FastWaves, DefaultNet>();
      } else {
        const size_t dp = (size_t)((d + 3) & ~3);
        const int mats4 = (a.sigma_identity ? 0 : 1) + (ou ? 1 : 0) + (pb->kind == SOCMX_OU_QUADRATIC ? 1 : 0);
        floats4 = (size_t)a.lds_mats + ((mats4 * dp * sds + 3) & ~(size_t)3) + 5 * 4 * kRow4Stride;
      }
      if (floats4 * sizeof(float) > (size_t)kMaxLdsBytes) return SOCMX_E_LDS;
      if (const int err = ensure_max_lds(k4)) return err;
      const int blocks4 = (B + 3) / 4;
      // (one workgroup per CU while there are CUs left: the whole LDS)
      const size_t lds4 = blocks4 <= 256 ? (size_t)kMaxLdsBytes : floats4 * sizeof(float);
      return launch(k4, dim3(blocks4), dim3(nw4 * 64), lds4, stream, a);
    }
  }
  // Evaluation bursts (utils.py:131-231: more 16-row tiles than CUs) at the default widths, sigma = I, d <= 15: TWO tiles per
  // workgroup (socmx_rollout32.hip) -- every weight fragment feeds both tiles' MFMAs and the per-stage fixed costs (barrier,
  // operand latency, prefetch hand-over) are paid once per 32 rows.  Bit-identical results; SOCMX_BURST_ROWS=16 keeps the
  // one-tile form (developer A/B switch, tests).
  static const int burst_rows = [] { const char* e = getenv("SOCMX_BURST_ROWS"); return e ? atoi(e) : 32; }();
  const bool burst32 = (is_default && one_row_form) || (is_wide32 && a.sigma_identity && d >= 16 && d <= 31);
  if (burst32 && !force_slow && !(prof && stopping) && blocks > 256 && burst_rows == 32 && force_rows == 0 && rollout32_available(a.u.in0p))
    return rollout32_launch(a, stopping, stream);
  if (const int err = ensure_max_lds(kern)) return err;
  // Few row tiles (a training batch: B / 16 workgroups on a 256-CU chip): claim the CU's whole LDS, so that no workgroup
  // of a kernel running beside the rollout on another stream (the pair-grid network's GEMMs) is placed on the same CU
  // and takes MFMA issue slots from this latency-bound chain.
  const size_t launch_lds = blocks <= 128 ? (size_t)kMaxLdsBytes : lds_bytes;
  return launch(kern, dim3(blocks), dim3(nw * 64), launch_lds, stream, a);
}

extern "C" int socmx_rollout_f32(const socmx_problem* pb, const float* packed_unet, const int32_t hdims[3],
                                 const float* x0, const float* ts, int32_t B, int32_t K, float lmbd, uint64_t seed,
                                 uint64_t offset, int64_t row0, const float* noise_in, float* states, float* noises,
                                 float* controls, float* stop_indicators, float* fractional_timesteps, float* lpd,
                                 float* lps, float* ltw, socmx_stream_t stream) {
  return rollout_launch(pb, packed_unet, hdims, x0, ts, B, K, lmbd, seed, offset, nullptr, nullptr, 0u, row0, noise_in, states,
                        noises, controls, stop_indicators, fractional_timesteps, lpd, lps, ltw, nullptr, stream);
}

extern "C" int socmx_rollout_ex_f32(const socmx_problem* pb, const float* packed_unet, const int32_t hdims[3],
                                    const float* x0, const float* ts, int32_t B, int32_t K, float lmbd, uint64_t seed,
                                    uint64_t offset, int64_t row0, const float* noise_in, float* states, float* noises,
                                    float* controls, float* stop_indicators, float* fractional_timesteps, float* lpd,
                                    float* lps, float* ltw, const socmx_rollout_extra* extra, socmx_stream_t stream) {
  const uint64_t* key = extra ? extra->key : nullptr;
  float* nabla_v = extra ? extra->nabla_v : nullptr;
  return rollout_launch(pb, packed_unet, hdims, x0, ts, B, K, lmbd, seed, offset, key, nabla_v, extra ? extra->flags : 0u, row0, noise_in, states,
                        noises, controls, stop_indicators, fractional_timesteps, lpd, lps, ltw, nullptr, stream,
                        extra ? extra->act_workspace : nullptr, extra ? extra->act_records : nullptr);
}

// 1 when socmx_rollout_ex_f32 with these arguments (trajectory buffers, nabla_v) honours act_workspace / act_records, 0 when it would refuse
// them, negative: invalid arguments.  Launches nothing.
extern "C" int socmx_rollout_saves_activations(const socmx_problem* pb, const int32_t hdims[3], int32_t B, int32_t K) {
  if (!pb || !hdims) return SOCMX_E_NULL;
  static float dummy[4];
  // (the selection is the launcher's own: every pointer it only checks for presence is `dummy`, nothing is dereferenced on the query path)
  socmx_problem q = *pb;
  if (!q.sigma) q.sigma = dummy;
  return rollout_launch(&q, dummy, hdims, dummy, dummy, B, K, 1.f, 0, 0, nullptr, dummy, 0u, 0, nullptr, dummy, dummy, dummy, dummy, dummy, dummy,
                        dummy, dummy, nullptr, nullptr, nullptr, nullptr, true);
}

extern "C" int socmx_philox_advance(uint64_t* key, uint64_t inc, socmx_stream_t stream) {
  if (!key) return SOCMX_E_NULL;
  return launch(philox_advance_kernel, dim3(1), dim3(64), 0, stream, key, inc);
}

extern "C" int socmx_rollout_phase_cycles_f32(const socmx_problem* pb, const float* packed_unet,
                                              const int32_t hdims[3], const float* x0, const float* ts, int32_t B,
                                              int32_t K, float lmbd, uint64_t seed, uint64_t offset, int64_t row0,
                                              const float* noise_in, float* states, float* noises, float* controls,
                                              float* stop_indicators, float* fractional_timesteps, float* lpd,
                                              float* lps, float* ltw, int64_t* cycles, socmx_stream_t stream) {
  if (!cycles) return SOCMX_E_NULL;
  return rollout_launch(pb, packed_unet, hdims, x0, ts, B, K, lmbd, seed, offset, nullptr, nullptr, SOCMX_ROLLOUT_SHARES_CHIP, row0,
                        noise_in, states, noises, controls, stop_indicators, fractional_timesteps, lpd, lps, ltw, (long long*)cycles, stream);
}
