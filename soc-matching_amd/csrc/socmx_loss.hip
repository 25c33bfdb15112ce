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
#include <algorithm>
#include <cstdlib>
#include <math.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/socmx.h"
#include "socmx_launch.h"

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
struct StatScalars {       // socmx_weights_stats_scalars_f32: all optional
  const float* gamma; float* gam_out; const float* norm; float* gout_out; float* obj_zero;
};
__global__ __launch_bounds__(256) void weights_stats_kernel(const float* __restrict__ lpd, const float* __restrict__ lps,
                                                            const float* __restrict__ ltw, int B, float* __restrict__ w,
                                                            float* __restrict__ stats, const StatScalars sc) {
  __shared__ float red[32];
  if (threadIdx.x == 0) {
    if (sc.gam_out) sc.gam_out[0] = sc.gamma[0];
    if (sc.gout_out) sc.gout_out[0] = 1.f / sc.norm[0];
    if (sc.obj_zero) sc.obj_zero[0] = 0.f;
  }
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
    stats[3] = mean;                                   // torch.mean(weight)            method.py:903
    stats[4] = sqrtf(m2 / (float)(B - 1));             // torch.std(weight), unbiased   method.py:904 (NaN for B = 1)
  }
}

// ---- importance-weight statistics of a batch SHARD, in a form ONE all_reduce(SUM) combines EXACTLY ----------------
// phase 0: tail[0] = objective; rank r's slot tail[1 + 3 r ..] = (n_r, mean_r, M2_r = sum (w - mean_r)^2) of ITS rows (two
//          passes: well conditioned whatever the weights' scale), every other rank's slot = 0.  Summing zeros is exact, so the
//          all-reduce hands every rank all world triples bit for bit (an all-gather riding in the iteration's one collective);
// phase 1: Chan's pooling rule over the slots in rank order, in fp64: mean = sum n_r mean_r / N,
//          M2 = sum M2_r + sum n_r (mean_r - mean)^2, unbiased std = sqrt(M2 / (N - 1))                 (method.py:903-904).
// (Rounds 4-5 summed (w - c), (w - c)^2 against the running normaliser c: with c far from mean(w) -- bench.py's 1.0 against
//  0.03 -- the fp32 sums cancelled to 1e-4 of the std.)
__global__ __launch_bounds__(256) void shard_stats_kernel(int phase, const float* __restrict__ w, int B, int rank, int world,
                                                          const float* __restrict__ obj, float* __restrict__ tail,
                                                          float* __restrict__ mean_std) {
  __shared__ float red[32];
  if (phase == 0) {
    float s1 = 0.f;
    for (int m = threadIdx.x; m < B; m += blockDim.x) s1 += w[m];
    const float mean = block_sum(s1, red) / (float)B;
    float m2 = 0.f;
    for (int m = threadIdx.x; m < B; m += blockDim.x) {
      const float x = w[m] - mean;
      m2 += x * x;
    }
    m2 = block_sum(m2, red);
    for (int e = threadIdx.x; e < 3 * world; e += blockDim.x) {
      const int r = e / 3, c = e - 3 * r;
      tail[1 + e] = r != rank ? 0.f : (c == 0 ? (float)B : (c == 1 ? mean : m2));
    }
    if (threadIdx.x == 0) tail[0] = obj ? obj[0] : 0.f;
  } else if (threadIdx.x == 0) {
    double n = 0.0, s = 0.0;
    for (int r = 0; r < world; ++r) {
      n += (double)tail[1 + 3 * r];
      s += (double)tail[1 + 3 * r] * (double)tail[2 + 3 * r];
    }
    const double mean = s / n;
    double m2 = 0.0;
    for (int r = 0; r < world; ++r) {
      const double dm = (double)tail[2 + 3 * r] - mean;
      m2 += (double)tail[3 + 3 * r] + (double)tail[1 + 3 * r] * dm * dm;
    }
    mean_std[0] = (float)mean;
    mean_std[1] = (float)sqrt(m2 / (n - 1.0));
  }
}

// ---- operand preparation: method.py:591-646 -------------------------------------------------------
struct PrepArgs {
  int kind, d, K, B;
  float sqrt_lmbd;
  const float *sit, *A, *P, *Q, *omega, *kappa, *nu;
  const float *ts, *states, *noises, *controls, *frac;
  float *v, *q, *gT;      // (K,B,d), (K,B,d), (B,d)      batch-major  (backward kernel)
  float *vT, *qT, *gTT;   // (K,d,B), (K,d,B), (d,B)      batch-fastest copies, optional (NULL = not written)
  int n_tiled_blocks;     // socm_prep_tiled_kernel: workgroups with row tiles (the ones behind them take the terminal rows)
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
      if (a.gTT) a.gTT[(size_t)l * B + m] = g;
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
    if (a.vT) a.vT[((size_t)j * d + l) * B + m] = vl;
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
    if (a.qT) a.qT[((size_t)j * d + l) * B + m] = ql;
  }
}

// Tiled form for d <= 128: a workgroup owns 64 consecutive (j,m) rows -- one contiguous 64*d run of states, noises
// and controls -- staged through LDS (row stride d+1: a wave reads 64 different rows of one column without bank
// conflicts); thread = (row, quarter of the output columns), so sigma^-T / A / P are wave-uniform scalar loads.
// Outputs go back through LDS and leave as one coalesced run.  (The per-thread form above walks d*d strided
// global reads per row: 27 ms at d = 64, K = 400, B = 512.)
__device__ __forceinline__ void socm_prep_terminal_rows(const PrepArgs& a, int e);
__global__ __launch_bounds__(256) void socm_prep_tiled_kernel(const PrepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // (the workgroups behind the row tiles form the terminal rows' nabla_g: one launch less on the iteration's critical path)
  if ((int)blockIdx.x >= a.n_tiled_blocks) {
    socm_prep_terminal_rows(a, ((int)blockIdx.x - a.n_tiled_blocks) * 256 + (int)threadIdx.x);
    return;
  }
  const int d = a.d, B = a.B, K = a.K, S = d + 1;
  float* E = lds;             // noise tile, later the q tile
  float* U = E + 64 * S;      // control tile
  float* X = U + 64 * S;      // state tile
  float* V = X + 64 * S;      // v
  const int64_t nrows = (int64_t)K * B;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int nr = (int)min((int64_t)64, nrows - row0);
  for (int e = threadIdx.x; e < nr * d; e += 256) {
    const int r = e / d, c = e - r * d;
    E[r * S + c] = a.noises[row0 * d + e];
    U[r * S + c] = a.controls[row0 * d + e];
    X[r * S + c] = a.states[row0 * d + e];
  }
  __syncthreads();
  const int r = threadIdx.x & 63, lq = threadIdx.x >> 6;
  const int lbeg = (d * lq) >> 2, lend = (d * (lq + 1)) >> 2;
  const bool live = r < nr;
  const int64_t idx = row0 + (live ? r : 0);
  const int j = (int)(idx / B);
  const float dt = a.frac ? a.frac[idx] : (a.ts[j + 1] - a.ts[j]);
  const float sdt = sqrtf(dt);
  const bool is_ou = (a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR);
  for (int l = lbeg; l < lend; ++l) {
    float se = 0.f, su = 0.f;
    for (int c = 0; c < d; ++c) {
      const float sc = a.sit[l * d + c];
      se += sc * E[r * S + c];
      su += sc * U[r * S + c];
    }
    V[r * S + l] = -(a.sqrt_lmbd * sdt * se + dt * su);
  }
  __syncthreads();
  for (int l = lbeg; l < lend; ++l) {
    float ql;
    if (is_ou) {
      float sacc = 0.f;
      for (int n = 0; n < d; ++n) sacc += a.A[n * d + l] * V[r * S + n];   // (A^T v)_l
      ql = sacc;
      if (a.kind == SOCMX_OU_QUADRATIC) {
        float px = 0.f;
        for (int c = 0; c < d; ++c) px += a.P[l * d + c] * X[r * S + c];
        ql += dt * 2.f * px;
      }
    } else {
      const float kap = a.kappa[l], xl = X[r * S + l];
      ql = -(8.f * kap * xl * xl + 4.f * kap * (xl * xl - 1.f)) * V[r * S + l];
    }
    E[r * S + l] = ql;        // the noise tile is consumed (barrier above): q is staged there
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nr * d; e += 256) {
    const int rr = e / d, c = e - rr * d;
    a.v[row0 * d + e] = V[rr * S + c];
    a.q[row0 * d + e] = E[rr * S + c];
    if (a.vT || a.qT) {
      const int64_t id2 = row0 + rr;
      const int jj = (int)(id2 / B), mm = (int)(id2 - (int64_t)jj * B);
      if (a.vT) a.vT[((size_t)jj * d + c) * B + mm] = V[rr * S + c];
      if (a.qT) a.qT[((size_t)jj * d + c) * B + mm] = E[rr * S + c];
    }
  }
}

// terminal rows only (j == K): gT = nabla_g(X_K); thread per batch row
// nabla_g of the terminal rows: thread e = (row m, component l) -- one thread per ROW walked d * d strided reads in a chain
// (33 us at d = 20, B = 128 on the critical path of every iteration; 128 threads busy)
__device__ __forceinline__ void socm_prep_terminal_rows(const PrepArgs& a, int e) {
  const int d = a.d, B = a.B, K = a.K;
  const int m = e / d, l = e - m * d;
  if (m >= B) return;
  const float* x = a.states + ((size_t)K * B + m) * d;
  float g = 0.f;
  if (a.kind == SOCMX_OU_QUADRATIC) {
    for (int c = 0; c < d; ++c) g += a.Q[l * d + c] * x[c];
    g *= 2.f;
  } else if (a.kind == SOCMX_OU_LINEAR) {
    g = a.omega[l];
  } else if (a.kind == SOCMX_DOUBLE_WELL) {
    g = 2.f * a.nu[l] * (x[l] * x[l] - 1.f) * 2.f * x[l];
  }
  a.gT[(size_t)m * d + l] = g;
  if (a.gTT) a.gTT[(size_t)l * B + m] = g;
}
__global__ __launch_bounds__(256) void socm_prep_terminal_kernel(const PrepArgs a) {
  socm_prep_terminal_rows(a, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- target + residual (forward) --------------------------------------------------------------------
struct TargetArgs {
  int d, K, B, KG;        // KG = k-groups (waves) per workgroup; each thread owns KO outputs
  float inv_norm;
  const float *sigma;
  const float *M_all, *dM_all;   // (Np,d,d)
  const float *q, *v, *gT;       // (K,B,d), (K,B,d), (B,d)   batch-major
  const float *nablaV, *w;       // (Kp,B,d), (B,)
  float *target, *G, *objective; // (Kp,B,d) or NULL, (Kp,B,d), (1,)
  float *ws;                     // objective workspace: [0] ticket (as unsigned; zero between launches), [2 ..] one slot per contributor
  // NET variant: M_all / dM_all hold the raw network outputs net, d(net)/ds and the pair matrices
  //   M = e I + (1-e) net,  dM = gamma e (net - I) + (1-e) dnet,  e = exp(-gamma (s-t))       (models.py:263-275)
  // are formed in registers while the A fragments are loaded.
  const float *delta, *gamma;    // (Np,) s-t per pair, (1,) on the device
  int fuse_residual;             // socm_target_mfma_kernel at d <= 16: the residual (objective, G) leaves with the target rows
};

// The objective of a launch without float atomics (method.py:717-720 is one torch.sum: the reference's loss value is reproducible
// run to run, and so is this one).  Every contributor -- a (workgroup, row) of the fused contraction, a workgroup or a wave of the
// residual kernels -- stores its partial sum into ITS slot of the caller's workspace and draws a ticket; the wave that draws the
// last one adds the slots up in a fixed order (lane-strided partial sums, then the wave butterfly), adds the total to
// objective[0] and re-arms the ticket.  Called by a WHOLE wave; `v` is read from lane 0.
__device__ __forceinline__ void objective_commit(const TargetArgs& a, unsigned slot, unsigned nslots, float v) {
  // No fences: a __threadfence() here is an L2 write-back of everything the XCD holds dirty (the G rows this kernel has just
  // written), once per contributor -- 8 us at configs[2].  Device-scope read-modify-write atomics execute at the point all
  // XCDs share, so the slot travels as an atomic exchange whose RETURN is waited for (complete = visible to everybody) before
  // the ticket is drawn, and the last wave reads the slots the same way (an atomic add of zero).
  const int lane = threadIdx.x & 63;
  unsigned* const ticket = reinterpret_cast<unsigned*>(a.ws);
  float* const slots = a.ws + 2;
  unsigned last = 0u;
  if (lane == 0) {
    const float old = atomicExch(&slots[slot], v);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    last = atomicAdd(ticket, 1u) == nslots - 1 ? 1u : 0u;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  float s = 0.f;
  for (unsigned k = lane; k < nslots; k += 64) s += atomicAdd(&slots[k], 0.f);
  s = wave_sum(s);
  if (lane == 0) {
    a.objective[0] += s;
    atomicExch(ticket, 0u);
  }
}

__host__ __device__ inline int64_t pair_row_offset(int i, int K) {
  return (int64_t)i * (K + 1) - (int64_t)i * (i - 1) / 2;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// target[i,m,:] = sum_{j>=i} ( M_ij q_j[m] - dM_ij v_j[m] ) + M_iK gT[m]  as per-row GEMMs on the fp32 MFMA:
//   D (16 k-rows x 16 batch columns) += A (M_ij[k][l], 16 x 4) . B (q_j[m][l], 4 x 16)      v_mfma_f32_16x16x4_f32
// One wave owns (row pair (i, K-i), 16-column batch tile, 16-row k-block): every wave runs K+2 pair matrices
// (balanced triangular work), KP2 x B/16 x ceil(d/16) waves fill the chip, and the next (pair, l-block)'s four
// operand fragments are loaded while the current one is multiplied.
// Operand slots: MFMA number s of an l-block takes l = lb + 4*g4 + s from lane group g4 -- for A and for B alike,
// so the reduction is unchanged while every lane reads FOUR CONSECUTIVE floats of a row of M (row-major k,l) and of
// q (batch-major m,l): one 16-byte load per operand instead of four strided 4-byte loads.  Lanes outside d x d are
// fed zeros; a 16-byte read that would cross the end of a buffer falls back to guarded scalar reads.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// four consecutive floats at base[off..off+3]; SAFE = false clamps every index below `limit` (end of the buffer)
template <bool SAFE>
__device__ __forceinline__ f32x4 load4(const float* base, int off, int limit) {
  if (SAFE) {
    const f32x4u t = *reinterpret_cast<const f32x4u*>(base + off);
    return f32x4{t[0], t[1], t[2], t[3]};
  }
  f32x4 r;
#pragma unroll
  for (int s = 0; s < 4; ++s) r[s] = base[min(off + s, limit - 1)];
  return r;
}

__device__ __forceinline__ const void* scalar_ptr(const void* p) {     // a wave-uniform pointer, pinned to scalar registers
  const uint64_t u = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}

// four consecutive floats at ubase[float_off ..] with ubase WAVE-UNIFORM: a buffer load -- the base in a scalar resource, one 32-bit
// offset register per lane -- instead of a global load with a 64-bit address per lane (whose address arithmetic and issue are
// time the SIMD's matrix pipe does not get back, see csrc/socmx_rollout32.hip).  Alignment as load4<true>: 4 bytes.
__device__ __forceinline__ f32x4 uload4(const float* ubase, int float_off) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(scalar_ptr(ubase))), 0, 0x7FFFFFFF, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 0));
}

// the first N of them (N = 1, 3: a 4- / 12-byte load -- the vector memory pipe's time follows the bytes per lane)
template <int N>
__device__ __forceinline__ f32x4 uloadn(const float* ubase, int float_off) {
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(reinterpret_cast<const float*>(scalar_ptr(ubase))), 0, 0x7FFFFFFF, 0x00020000);
  if constexpr (N == 4) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 0));
  } else if constexpr (N == 3) {
    // (the whole vector is cast at once: element-wise casts of the integer vector make this compiler narrow the load to one dword)
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x3 t = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, float_off * 4, 0, 0));
    return f32x4{t[0], t[1], t[2], 0.f};
  } else {
    static_assert(N == 1, "1, 3 or 4 floats");
    const unsigned t = __builtin_amdgcn_raw_buffer_load_b32(r, float_off * 4, 0, 0);
    return f32x4{__builtin_bit_cast(float, t), 0.f, 0.f, 0.f};
  }
}

constexpr int kTargetWaves = 8;  // waves per workgroup: the (pair, l-block) iterations of a row are dealt round-robin

// CT = 16-column batch tiles per wave: the A fragments (and, for NET, the blend that forms them) are built once
// per (pair, l-block) and multiplied into CT accumulators, so the per-iteration bookkeeping (~300 issue cycles)
// is amortised over 2*NS*CT MFMAs.  NS != 0 (d <= 16): a single l-block of NS MFMAs, no division in the iteration -> pair map.
template <bool NET, int CT, int NS>
__global__ __launch_bounds__(64 * kTargetWaves) void socm_target_mfma_kernel(const TargetArgs a) {
  __shared__ f32x4 part[kTargetWaves][CT][64];
  __shared__ float sg[NS != 0 ? 256 : 1];        // sigma (d x d, d <= 16): the fused residual's two products
  __shared__ float objw[CT];
  constexpr bool NLB1 = NS != 0;
  const int d = a.d, K = a.K, B = a.B;
  const bool fuse = NLB1 && a.fuse_residual;
  if (fuse) {                                    // (visible to the epilogue through the barriers of the combine)
    for (int e = threadIdx.x; e < d * d; e += 64 * kTargetWaves) sg[e] = a.sigma[e];
  }
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  int mcol[CT], boff0[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    mcol[c] = (blockIdx.y * CT + c) * 16 + c16;
    boff0[c] = min(mcol[c], B - 1) * d;
  }
  const int kb = blockIdx.z * 16;          // first k-row of this wave's block
  const int krow = kb + c16;               // A-fragment row of this lane
  const int nlb = NLB1 ? 1 : (d + 15) >> 4;  // 16-wide l-blocks
  const int nlb_shift = (nlb & (nlb - 1)) == 0 ? __builtin_ctz(nlb) : -1;
  const int dd = d * d;
  // d <= 12 (one l-block): lane group g4 takes l = ns*g4 .. ns*g4 + ns-1 with ns = ceil(d / 4), so that the l-block is ns MFMAs
  // instead of four (d = 10: three; d <= 4: one) -- the 16-byte reads stay, their tail is masked at consumption
  // (NS = 1: d <= 4; 3: d <= 12; 4: d <= 16; 0: several l-blocks of four MFMAs)
  constexpr int ns = NLB1 ? NS : 4;
  const float gam = NET ? a.gamma[0] : 0.f;
  float ob_rows = 0.f;                           // (wave 0) the objective's share of this workgroup's rows
  for (int rep = 0; rep < 2; ++rep) {
    const int i = rep == 0 ? (int)blockIdx.x : K - (int)blockIdx.x;
    if (rep == 1 && i <= (int)blockIdx.x) break;
    const float* drow = NET ? a.delta + pair_row_offset(i, K) : nullptr;
    const int niter = (K - i + 1) * nlb;   // flattened (pair, l-block) iterations
    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t prow_dd = pair_row_offset(i, K) * dd;
    // Raw operands travel through the prefetch ring; anything computed FROM them (zero padding, the NET blend)
    // happens at consumption time, so that issuing a load never makes the wave wait for it.
    struct Slot { f32x4 nt, dn, q[CT], v[CT]; float dl; };
    auto geom = [&](int t, int& jr, int& j, int& l0, int& nl) {
      jr = NLB1 ? t : (nlb_shift >= 0 ? t >> nlb_shift : t / nlb);   // a runtime integer division is ~30 instructions
      const int lb = NLB1 ? 0 : (t - jr * nlb) * 16;
      j = i + jr;
      l0 = min(lb + ns * g4, d - 1);                // this lane's l: l0 .. l0+ns-1 (clamped into the row)
      nl = max(0, min(ns, d - (lb + ns * g4)));     // how many of them exist
    };
    auto load = [&](int t, Slot& sl, auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      int jr, j, l0, nl;
      geom(t, jr, j, l0, nl);
      const float* Ap = a.M_all + prow_dd + (int64_t)jr * dd;         // wave-uniform bases + small lane offsets
      const float* Dp = a.dM_all + prow_dd + (int64_t)jr * dd;
      const int aoff = min(krow, d - 1) * d + l0;
      const float* qs = (j < K) ? a.q + (size_t)j * B * d : a.gT;
      const float* vs = a.v + (size_t)(j < K ? j : 0) * B * d;
      // a 16-byte read can cross the end of a buffer only in the last pair matrix / the last operand rows:
      // those iterations (j + 1 >= K: the row's last two pairs) take clamped scalar reads -- in a loop of their own (below)
      if (FAST) {
        sl.nt = uloadn<ns>(Ap, aoff);
        sl.dn = uloadn<ns>(Dp, aoff);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          sl.q[c] = uloadn<ns>(qs, boff0[c] + l0);
          sl.v[c] = uloadn<ns>(vs, boff0[c] + l0);
        }
      } else {
        sl.nt = load4<false>(Ap, aoff, dd);
        sl.dn = load4<false>(Dp, aoff, dd);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          sl.q[c] = load4<false>(qs, boff0[c] + l0, B * d);
          sl.v[c] = load4<false>(vs, boff0[c] + l0, B * d);
        }
      }
      sl.dl = NET ? drow[jr] : 0.f;
    };
    auto consume = [&](int t, const Slot& sl) {
      int jr, j, l0, nl;
      geom(t, jr, j, l0, nl);
      const int na = krow < d ? nl : 0;
      const float e = NET ? expf(-gam * sl.dl) : 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s >= ns) break;
        float xm, xd;
        if (NET) {
          const float eye = (krow == l0 + s) ? 1.f : 0.f;
          xm = e * eye + (1.f - e) * sl.nt[s];
          xd = -(gam * e * (sl.nt[s] - eye) + (1.f - e) * sl.dn[s]);
        } else {
          xm = sl.nt[s];
          xd = -sl.dn[s];
        }
        xm = (s < na) ? xm : 0.f;
        xd = (s < na && j < K) ? xd : 0.f;
        // consecutive MFMAs go to different accumulators (a dependent pair costs 40 cycles instead of 32)
#pragma unroll
        // (no masks on the B side: xm / xd are zero wherever the B element lies past the row or belongs to the terminal
        //  pair's absent v operand, and the clamped reads return finite values there)
        for (int c = 0; c < CT; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xm, sl.q[c][s], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd, sl.v[c][s], acc[c], 0, 0, 0);
      }
    };
    // PF iterations of this wave are in flight: one (pair, l-block) is 8*CT MFMAs, shorter than a trip to L2/HBM
    // The pipelined loop covers the iterations whose 16-byte reads are always in bounds (pairs j + 1 < K) and requests
    // UNCONDITIONALLY (past the wave's last such iteration it re-reads that one): with two load forms or an `if (more) load`
    // inside the loop the compiler's wait counts assume the worst path and leave about one slot in flight instead of PF.
    // The row's last two pairs follow in a plain loop with the guarded scalar reads.
    constexpr int PF = CT >= 4 ? 3 : (CT >= 2 ? 3 : 4);
    const int nfast = max(0, K - 1 - i) * nlb;           // iterations t with j + 1 < K  (t = jr * nlb + l-block, j = i + jr)
    if (wave < nfast) {
      const int tmax = wave + ((nfast - 1 - wave) / kTargetWaves) * kTargetWaves;     // this wave's last fast iteration
      Slot ring[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) load(min(wave + p * kTargetWaves, tmax), ring[p], std::true_type{});
      for (int t = wave; t < nfast; t += PF * kTargetWaves) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
          const int tt = t + p * kTargetWaves;
          if (tt < nfast) consume(tt, ring[p]);
          load(min(tt + PF * kTargetWaves, tmax), ring[p], std::true_type{});
        }
      }
    }
    {
      const int t0 = nfast + ((wave - nfast % kTargetWaves) + kTargetWaves) % kTargetWaves;   // first t >= nfast with t = wave (mod 8)
      for (int t = t0; t < niter; t += kTargetWaves) {
        Slot sl;
        load(t, sl, std::false_type{});
        consume(t, sl);
      }
    }
    // combine the waves' partial tiles (fixed order: deterministic)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CT; ++c) part[wave][c][lane] = acc[c];
    __syncthreads();
    // D: lane holds k = kb + 4*g4 + r (r = 0..3) of batch column m; wave c finishes column tile c
    if (wave < CT) {
      f32x4 tot = part[0][wave][lane];
#pragma unroll
      for (int w = 1; w < kTargetWaves; ++w) tot += part[w][wave][lane];
      const int m = (blockIdx.y * CT + wave) * 16 + c16;
      if (m < B) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kb + 4 * g4 + r;
          if (k < d) a.target[((size_t)i * B + m) * d + k] = tot[r];
        }
      }
      if (fuse) {
        // The residual of method.py:702-720 on the rows this wave just finished (d <= 16: kb = 0, the four lanes m, m + 16,
        // m + 32, m + 48 hold the row's d components, four each):  R = sigma^T (nabla_V - target),  objective += w |R|^2 inv_norm,
        // G = 2 w inv_norm sigma R  -- what socm_residual_kernel computes from the stored target, one launch later
        const int mc = min(m, B - 1);
        const size_t row = ((size_t)i * B + mc) * d;
        const float wm = a.w[mc];
        float df[4], sk[4] = {0.f, 0.f, 0.f, 0.f};
        int kq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 4 * g4 + r;
          kq[r] = min(k, d - 1) * d;
          df[r] = k < d ? a.nablaV[row + k] - tot[r] : 0.f;
        }
        float ob = 0.f;
        for (int c = 0; c < d; ++c) {
          float rc = (sg[kq[0] + c] * df[0] + sg[kq[1] + c] * df[1]) + (sg[kq[2] + c] * df[2] + sg[kq[3] + c] * df[3]);
          rc += __shfl_xor(rc, 16, 64);
          rc += __shfl_xor(rc, 32, 64);                    // every lane group now holds R[c]
          ob += rc * rc;
#pragma unroll
          for (int r = 0; r < 4; ++r) sk[r] += sg[kq[r] + c] * rc;
        }
        if (m < B) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * g4 + r < d) a.G[row + 4 * g4 + r] = 2.f * wm * a.inv_norm * sk[r];
        }
        ob = wave_sum((m < B && g4 == 0) ? wm * ob : 0.f);
        if (lane == 0) objw[wave] = ob;
      }
    }
    if (fuse) {                                           // this row's share of the objective, kept by wave 0 until both rows are done
      __syncthreads();
      if (wave == 0) {
        float ob = 0.f;
#pragma unroll
        for (int c = 0; c < CT; ++c) ob += objw[c];
        ob_rows += ob * a.inv_norm;
      }
    }
  }
  // ONE slot per workgroup (its two rows added in a fixed order): objective_commit is two dependent device-scope atomics with their
  // returns waited for -- once per row it stood between the workgroup's two rows and again at its end (~3 us each at configs[2])
  if (fuse && wave == 0)
    objective_commit(a, blockIdx.x * gridDim.y + blockIdx.y, gridDim.x * gridDim.y, ob_rows);
}

// Wide form for 16 < d <= 64 and B >= 256 (the MFMA-bound regime, e.g. d = 64, K = 400, B = 512: 676 GFLOP), A operand
// staged through LDS: a workgroup owns a row and 8 x CT x 16 batch columns; its 8 waves split the COLUMNS, each wave keeps
// all KB k-blocks x CT=2 column tiles of the output in registers and walks the (pair, l-block) iterations in order.  The
// pair matrices are fetched ONCE per workgroup: threads 0..255 each
// own one 16-byte piece (row k, four l) of net and dnet, keep three iterations of them in flight in registers,
// form M / -dM/ds (zero padding, NET blend) once and write them to a three-stage LDS ring; the eight waves read their
// MFMA A fragments from there (ds_read_b128).  One barrier per iteration.  The B fragments (q, v: different for every
// wave) stay on a three-deep register ring.
constexpr int kAStride = 20;   // floats per staged row (16 + 4 pad: 80-byte stride spreads the 16 rows over the banks)

template <bool NET, int KB>
__global__ __launch_bounds__(64 * kTargetWaves) void socm_target_lds_kernel(const TargetArgs a) {
  constexpr int CT = 2, ROWS = KB * 16;
  __shared__ __attribute__((aligned(16))) float As[3][2][ROWS][kAStride];
  const int d = a.d, K = a.K, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  int mcol[CT], boff0[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    mcol[c] = ((blockIdx.y * kTargetWaves + wave) * CT + c) * 16 + c16;
    boff0[c] = min(mcol[c], B - 1) * d;
  }
  const int nlb = (d + 15) >> 4;
  const int nlb_shift = (nlb & (nlb - 1)) == 0 ? __builtin_ctz(nlb) : -1;
  const int dd = d * d;
  const float gam = NET ? a.gamma[0] : 0.f;
  // loader role: tid < 4*ROWS -> (row ak, 16-byte piece aq)
  const bool loader = tid < 4 * ROWS;
  const int ak = tid >> 2, aq = tid & 3;
  {
    // one row per workgroup, longest rows first (blockIdx.x = i): the dispatcher hands out rows as CUs free up, which
    // balances the triangular work better than fixed (i, K-i) pairs when there are fewer pairs than 2 x CUs
    const int i = blockIdx.x;
    const float* drow = NET ? a.delta + pair_row_offset(i, K) : nullptr;
    const int nit = (K - i + 1) * nlb;
    const int64_t prow_dd = pair_row_offset(i, K) * dd;
    f32x4 acc[KB][CT];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[kb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto geom = [&](int it, int& jr, int& lb) {
      jr = nlb_shift >= 0 ? it >> nlb_shift : it / nlb;
      lb = (it - jr * nlb) * 16;
    };
    struct APend { f32x4 nt, dn; float dl; };
    struct BSlot { f32x4 q[CT], v[CT]; };
    auto loadA = [&](int it, APend& p) {          // loader threads only
      int jr, lb;
      geom(it, jr, lb);
      const int j = i + jr;
      const int l0 = min(lb + 4 * aq, d - 1);
      const float* Ap = a.M_all + prow_dd + (int64_t)jr * dd;
      const float* Dp = a.dM_all + prow_dd + (int64_t)jr * dd;
      const int aoff = min(ak, d - 1) * d + l0;
      if (j + 1 < K) {
        p.nt = uload4(Ap, aoff);
        p.dn = uload4(Dp, aoff);
      } else {
        p.nt = load4<false>(Ap, aoff, dd);
        p.dn = load4<false>(Dp, aoff, dd);
      }
      p.dl = NET ? drow[jr] : 0.f;
    };
    auto stageA = [&](int it, const APend& p, int st) {   // blend + zero padding, then one 16-byte LDS write each
      int jr, lb;
      geom(it, jr, lb);
      const int j = i + jr;
      const int lq = lb + 4 * aq;
      const int l0 = min(lq, d - 1);
      const int nl = max(0, min(4, d - lq));
      const int na = ak < d ? nl : 0;
      const float e = NET ? expf(-gam * p.dl) : 0.f;
      f32x4 xm, xd;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        float m_, d_;
        if (NET) {
          const float eye = (ak == l0 + s2) ? 1.f : 0.f;
          m_ = e * eye + (1.f - e) * p.nt[s2];
          d_ = -(gam * e * (p.nt[s2] - eye) + (1.f - e) * p.dn[s2]);
        } else {
          m_ = p.nt[s2];
          d_ = -p.dn[s2];
        }
        xm[s2] = (s2 < na) ? m_ : 0.f;
        xd[s2] = (s2 < na && j < K) ? d_ : 0.f;
      }
      *reinterpret_cast<f32x4*>(&As[st][0][ak][4 * aq]) = xm;
      *reinterpret_cast<f32x4*>(&As[st][1][ak][4 * aq]) = xd;
    };
    auto loadB = [&](int it, BSlot& b) {
      int jr, lb;
      geom(it, jr, lb);
      const int j = i + jr;
      const int l0 = min(lb + 4 * g4, d - 1);
      const float* qs = (j < K) ? a.q + (size_t)j * B * d : a.gT;
      const float* vs = a.v + (size_t)(j < K ? j : 0) * B * d;
      if (j + 1 < K) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          b.q[c] = uload4(qs, boff0[c] + l0);
          b.v[c] = uload4(vs, boff0[c] + l0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          b.q[c] = load4<false>(qs, boff0[c] + l0, B * d);
          b.v[c] = load4<false>(vs, boff0[c] + l0, B * d);
        }
      }
    };
    auto consume = [&](int it, const BSlot& b, int st) {
      int jr, lb;
      geom(it, jr, lb);
      const int j = i + jr;
      const int nl = max(0, min(4, d - (lb + 4 * g4)));
      f32x4 xm[KB], xd[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        xm[kb] = *reinterpret_cast<const f32x4*>(&As[st][0][kb * 16 + c16][4 * g4]);
        xd[kb] = *reinterpret_cast<const f32x4*>(&As[st][1][kb * 16 + c16][4 * g4]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        float xq[CT], xv[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          xq[c] = (s2 < nl) ? b.q[c][s2] : 0.f;
          xv[c] = (s2 < nl && j < K) ? b.v[c][s2] : 0.f;
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            acc[kb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xm[kb][s2], xq[c], acc[kb][c], 0, 0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int c = 0; c < CT; ++c)
            acc[kb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd[kb][s2], xv[c], acc[kb][c], 0, 0, 0);
      }
    };
    // prologue: stages 0, 1 filled; A loads of iterations 2, 3, 4 and B loads of 0, 1, 2 in flight
    APend pend[3];
    BSlot bs[3];
    __syncthreads();                               // previous row's readers are done with the ring
    if (loader) {
      loadA(0, pend[0]);
      if (1 < nit) loadA(1, pend[1]);
      stageA(0, pend[0], 0);
      if (1 < nit) stageA(1, pend[1], 1);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int it = 2 + u;                     // pend slot = it % 3
        if (it < nit) loadA(it, pend[it % 3]);
      }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < nit) loadB(u, bs[u]);
    for (int it0 = 0; it0 < nit; it0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int it = it0 + u;                   // it % 3 == u
        if (it < nit) {
          __syncthreads();                        // stage of `it` (and it+1) written; readers of it-1 finished
          if (loader && it + 2 < nit) {
            stageA(it + 2, pend[(u + 2) % 3], (u + 2) % 3);
            if (it + 5 < nit) loadA(it + 5, pend[(u + 2) % 3]);
          }
          consume(it, bs[u], u);
          if (it + 3 < nit) loadB(it + 3, bs[u]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int c = 0; c < CT; ++c)
        if (mcol[c] < B) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = kb * 16 + 4 * g4 + r;
            if (k < d) a.target[((size_t)i * B + mcol[c]) * d + k] = acc[kb][c][r];
          }
        }
  }
}

// ---- d % 4 == 0: the LDS-staged form, scheduled by hand around what the SIMD issues beside an MFMA stream ------------
// Same decomposition as socm_target_lds_kernel (row i per workgroup; 8 multiplying waves x 2 column tiles of 16 x all KB
// k-blocks; the pair matrices reach the waves through LDS).  Two measured facts (in-kernel cycle counters,
// tools/ubench/contraction_bench.hip -DSOCMX_CONTRACTION_PROF; tools/ubench/mfma_valu_overlap.hip) shape everything else:
//  (1) the compiler's wait-count insertion cannot see through a loop with conditional prefetches and waits for ALL loads
//      (vmcnt(0)) in front of every use: operands requested one iteration earlier were waited for at full latency.  Here
//      every global load is an asm statement the compiler does not track and every wait is written out with the exact
//      number of younger requests that may stay in flight; to keep that number constant the trip count is padded to a
//      multiple of four (padded trips multiply a zero A tile) and requests past the row's end re-read its last pieces.
//  (2) while one wave of a SIMD streams fp32 MFMAs, every other instruction on that SIMD -- vector, scalar, LDS, from any
//      other wave -- gets about one issue slot per MFMA (~40 cycles each), whereas a wave's OWN instructions placed
//      between its MFMAs are free.  Staging inside the multiplying waves cost 2,700 of 6,000 cycles per iteration (the
//      stagers crawled behind their SIMD neighbour's MFMAs, then multiplied while the neighbour idled at the barrier).
//      So: four extra waves (one per SIMD) do nothing but stage, with as few instructions as possible (~75 per
//      iteration, under the 128 slots two multiplying waves leave), and the multiplying waves issue their own operand
//      requests, cursor arithmetic and LDS reads BETWEEN their MFMAs.  Per iteration: 6,000 -> 4,700 cycles (4,096 is
//      the MFMA time), 7.6 -> 6.3 ms at the configs[4] slice (107 TFLOP/s, 0.68 of the fp32 MFMA peak).
// NOTE on the asm loads of this section: the compiler believes an asm statement's outputs are valid when the statement
// retires.  That is harmless only as long as it never touches those registers before the matching SOCMX_WAIT_* statement
// (whose "+v" operands tie them to the wait) -- in particular the register allocator must keep each request slot in the same
// registers around the loop (no copies on the back edge).  The loops are unrolled by the ring length for exactly that reason,
// and the GPU parity tests over many shapes (test_contraction_kernels_multi_block_shapes, the full-size configs[4] slice
// tests) are what guards it: a toolchain that inserted such a copy would fail them, not corrupt results silently.
__device__ __forceinline__ f32x4 async_load16(const void* sbase, uint32_t voff) {
  sbase = scalar_ptr(sbase);
  f32x4 r;
  // (s_nop: the base may have just been written by v_readfirstlane -- VALU-written SGPR read by VMEM needs 5 wait states,
  //  and the hazard recogniser does not look inside asm statements)
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ float async_load4(const void* sbase, uint32_t voff) {
  sbase = scalar_ptr(sbase);
  float r;
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}

constexpr int kStageWaves = 4;
// LDS ring of the staged pair-matrix tiles and the barrier period: ONE barrier per kPeriod iterations, ring = 2 kPeriod stages.
// Round 6 measured periods 2 / 4 / 6 (-DSOCMX_TARGET_PERIOD) and the three wave classes' issue priorities (-DSOCMX_TARGET_PRIO_*) at
// the configs[4] slice: 5.98 / 5.98 / 7.25 ms, and every priority assignment within 1 % -- an iteration costs the SIMD ~4,600 cycles
// for 4,096 of MFMA issue whoever waits for whom (profiles/r6/contraction_period.txt, contraction_prio.txt; tools/experiments/contraction)
#ifndef SOCMX_TARGET_PERIOD
#define SOCMX_TARGET_PERIOD 2
#endif
constexpr int kPeriod = SOCMX_TARGET_PERIOD, kRing = 2 * kPeriod;
// issue priorities (s_setprio) of the three wave classes: staging waves / first multiplying wave of a SIMD / second one
#ifndef SOCMX_TARGET_PRIO_STAGE
#define SOCMX_TARGET_PRIO_STAGE 0
#endif
#ifndef SOCMX_TARGET_PRIO_MUL0
#define SOCMX_TARGET_PRIO_MUL0 0
#endif
#ifndef SOCMX_TARGET_PRIO_MUL1
#define SOCMX_TARGET_PRIO_MUL1 0
#endif
#ifdef SOCMX_CONTRACTION_PROF
__device__ long long g_contraction_prof[kTargetWaves][4];
#endif

template <bool NET, int KB>
__global__ __launch_bounds__(64 * (kTargetWaves + kStageWaves)) void socm_target_lds4_kernel(const TargetArgs a) {
  constexpr int CT = 2, ROWS = KB * 16;
  constexpr int NA = NET ? 3 : 2;            // requests per A prefetch: net piece, dnet piece [, delta]
  __shared__ __attribute__((aligned(16))) float As[kRing][2][ROWS][kAStride];
  const int d = a.d, K = a.K, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lbl = ((d - 1) >> 4) << 4;       // the last l-block: pieces past the row re-read its last piece (zero on the A side)
  const int nlb = (d + 15) >> 4;
  const int dd = d * d;
  const int i = blockIdx.x;                  // one row per workgroup, longest rows first
  const int nit = (K - i + 1) * nlb;
  const int nitp = (nit + kRing - 1) / kRing * kRing;   // trips are unrolled by the ring length (request slots and LDS stages)
  const int lb_end = 16 * nlb;
  if (wave >= kTargetWaves) {
    __builtin_amdgcn_s_setprio(SOCMX_TARGET_PRIO_STAGE);
    // ---- staging waves: thread = (row ak, 16-byte piece aq) of the 16-column tile, both tensors ----
    // (every instruction here costs about one MFMA time: scalar bookkeeping is kept to one cursor whose per-request state
    //  travels with the request slot, 32-bit offsets against per-row base pointers, one hazard nop per request group)
    constexpr bool ALL_ON = 4 * ROWS >= 64 * kStageWaves;
    const int piece = tid - 64 * kTargetWaves;
    const bool on = ALL_ON || piece < 4 * ROWS;   // (KB = 2: two of the four waves have no piece)
    const int ak = (piece >> 2) & (ROWS - 1), aq = piece & 3;
    const uint32_t alane = (uint32_t)(min(ak, d - 1) * d + 4 * aq) * 4u;
    const uint32_t alane_last = (uint32_t)(min(ak, d - 1) * d + min(lbl + 4 * aq, d - 4) - lbl) * 4u;
    const float gam = NET ? a.gamma[0] : 0.f;
    const int64_t prow = pair_row_offset(i, K);
    const float* rowN = reinterpret_cast<const float*>(scalar_ptr(a.M_all + prow * dd));    // this row's first pair matrix
    const float* rowD = reinterpret_cast<const float*>(scalar_ptr(a.dM_all + prow * dd));
    const float* rowL = NET ? reinterpret_cast<const float*>(scalar_ptr(a.delta + prow)) : nullptr;
    const float rowmask = (on && ak < d) ? 1.f : 0.f;
    const int d_minus_aq = d - 4 * aq, diag0 = ak - 4 * aq;   // the tile's diagonal element of this row is piece element diag0 - lb
    // blend coefficients of the pair being staged (renewed at its first l-block), times this thread's row mask:
    //   M = e I + f net,   -dM/ds = ge I - ge net - f dnet      (e = exp(-gamma (s_j - t_i)), f = 1 - e, ge = gamma e)
    float ce = 0.f, cf = 0.f, cged = 0.f, cfd = 0.f;
    struct APend { f32x4 nt, dn; float dl; int lb, flags; };   // flags: 1 = first l-block of a pair, 2 = partial / padded, 4 = dead, 8 = terminal pair
    // request cursor: iteration t of the row = (pair jr, l-block lb); off = element offset of (jr, lb) inside the row's matrices
    // The pairs of a row are taken from j = K DOWN to j = i: every workgroup then starts at the same operand tile (q_K = nabla g)
    // and the workgroups running side by side sweep the q_j / v_j tiles in step -- the 20 GB of B-operand traffic of a launch
    // at the configs[4] slice hit in L2 instead of going to the memory side.
    int ct = 0, cjr = K - i, clb = 0;
    uint32_t coff = (uint32_t)(K - i) * (uint32_t)dd;
    const uint32_t wrap_step = 0u - (uint32_t)(dd + 16 * (nlb - 1));       // (mod 2^32) back to the previous pair's first l-block
    auto issueA = [&](APend& p) {             // NA requests for iteration ct (the row's last one once ct runs past it)
      const uint32_t lane_off = clb + 16 <= d ? alane : alane_last;
      const void* pn = scalar_ptr(rowN + coff);
      const void* pd = scalar_ptr(rowD + coff);
      const void* pl = scalar_ptr(NET ? rowL + cjr : rowN);
      if (NET)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %3, %4\n\tglobal_load_dwordx4 %1, %3, %5\n\tglobal_load_dword %2, %6, %7"
                     : "=&v"(p.nt), "=&v"(p.dn), "=&v"(p.dl) : "v"(lane_off), "s"(pn), "s"(pd), "v"(0u), "s"(pl) : "memory");
      else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %4"
                     : "=&v"(p.nt), "=&v"(p.dn) : "v"(lane_off), "s"(pn), "s"(pd) : "memory");
      p.lb = clb;
      p.flags = (clb == 0 ? 1 : 0) | ((clb + 16 > d || ct >= nit) ? 2 : 0) | (ct >= nit ? 4 : 0) | (i + cjr >= K ? 8 : 0);
      ++ct;
      if (ct < nit) {
        const bool wrap = clb + 16 == lb_end;
        coff += wrap ? wrap_step : 16u;
        clb = wrap ? 0 : clb + 16;
        cjr -= wrap ? 1 : 0;
      }
    };
    auto stageA = [&](const APend& p, int st) {   // blend + zero padding, then two 16-byte LDS writes
      if (p.flags & 1) {
        const float e = NET ? expf(-gam * p.dl) : 0.f;
        const float dm = (p.flags & 8) ? 0.f : rowmask;                   // terminal pair: -dM/ds = 0
        ce = rowmask * e; cf = NET ? rowmask * (1.f - e) : rowmask;
        cfd = NET ? dm * (1.f - e) : dm; cged = dm * gam * e;
      }
      f32x4 xm, xd;
      const int es = diag0 - p.lb;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        xm[s2] = NET ? fmaf(cf, p.nt[s2], es == s2 ? ce : 0.f) : cf * p.nt[s2];
        xd[s2] = NET ? fmaf(-cfd, p.dn[s2], fmaf(-cged, p.nt[s2], es == s2 ? cged : 0.f)) : -cfd * p.dn[s2];
      }
      if (p.flags & 2) {                        // the row's partial last l-block / a padded iteration: zero the tail
        const int na = (p.flags & 4) ? 0 : max(0, min(4, d_minus_aq - p.lb));
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) { xm[s2] = (s2 < na) ? xm[s2] : 0.f; xd[s2] = (s2 < na) ? xd[s2] : 0.f; }
      }
      if (on) {
        *reinterpret_cast<f32x4*>(&As[st][0][ak][4 * aq]) = xm;
        *reinterpret_cast<f32x4*>(&As[st][1][ak][4 * aq]) = xd;
      }
    };
#define SOCMX_WAIT_A(N, P) asm volatile("s_waitcnt vmcnt(%3)" : "+v"((P).nt), "+v"((P).dn), "+v"((P).dl) : "n"(N) : "memory")
    APend pend[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) pend[u].dl = 0.f;
    // ONE barrier per kPeriod iterations (period p = iterations P p .. P p + P - 1).  During period p the multiplying waves read
    // the fragments of P p + 1 .. P p + P (those of P p are in registers since the end of P p - 1), so those stages were written
    // during period p - 1, and this period writes the stages of P p + P + 1 .. P p + 2 P -- the ring of 2 P holds exactly these
    // (stage of P p + 2 P = stage of P p, whose reads were issued before this period's barrier).  Request slot of iteration t:
    // t % (2 P); a slot's request is issued 2 P iterations ahead of its staging (2 P - 1 younger requests in flight at the wait).
    // prologue: stages 0 .. P staged synchronously, then A(P + 1) .. A(3 P) in flight
#pragma unroll
    for (int u = 0; u <= kPeriod; ++u) issueA(pend[u]);
#pragma unroll
    for (int u = 0; u <= kPeriod; ++u) SOCMX_WAIT_A(0, pend[u]);
#pragma unroll
    for (int u = 0; u <= kPeriod; ++u) stageA(pend[u], u);
#pragma unroll
    for (int u = kPeriod + 1; u <= 3 * kPeriod; ++u) issueA(pend[u % kRing]);
    __syncthreads();                             // stages 0 .. P visible
    for (int it0 = 0; it0 < nitp; it0 += kRing) {
#pragma unroll
      for (int u = 0; u < kRing; u += kPeriod) {   // it = it0 + u opens a period
        __syncthreads();
#pragma unroll
        for (int v = 1; v <= kPeriod; ++v) {
          constexpr int dummy = 0; (void)dummy;
          const int sl = (u + kPeriod + v) % kRing;            // iteration it + P + v
          SOCMX_WAIT_A((kRing - 1) * NA, pend[sl]);            // younger: the 2 P - 1 requests issued after it
          stageA(pend[sl], sl);
          issueA(pend[sl]);                                    // iteration it + 3 P + v
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the over-fetched requests of the last trips
#undef SOCMX_WAIT_A
    return;
  }
  // ---- multiplying waves ----
  if (wave < 4) __builtin_amdgcn_s_setprio(SOCMX_TARGET_PRIO_MUL0); else __builtin_amdgcn_s_setprio(SOCMX_TARGET_PRIO_MUL1);
  const int c16 = lane & 15, g4 = lane >> 4;
  int mcol[CT];
  uint32_t blane[CT], blane_last[CT];        // byte offset of this lane's piece relative to (operand row block + l-block)
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    mcol[c] = ((blockIdx.y * kTargetWaves + wave) * CT + c) * 16 + c16;
    const int boff = min(mcol[c], B - 1) * d;
    blane[c] = (uint32_t)(boff + 4 * g4) * 4u;
    blane_last[c] = (uint32_t)(boff + min(lbl + 4 * g4, d - 4) - lbl) * 4u;
  }
  f32x4 acc[KB][CT];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[kb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  struct BSlot { f32x4 q[CT], v[CT]; };
  // request cursor of the B side: the operand pointers of iteration bt = (pair bjr, l-block blb) move by 16 floats inside a
  // pair and to the previous (B, d) block at a pair boundary; the terminal pair reads gT (and any finite v: its A is zero)
  static_assert(CT == 2, "the request group below is written for two column tiles");
  // (pairs from j = K down to j = i, l-blocks ascending inside a pair: the same order as the staging waves' cursor)
  int bt = 0, bjr = K - i, blb = 0;
  const float* pq = a.gT;
  const float* pv = a.v;
  auto requestB = [&](BSlot& b) {              // 2 * CT requests for iteration bt (one hazard nop)
    const bool whole = blb + 16 <= d;
    const uint32_t o0 = whole ? blane[0] : blane_last[0], o1 = whole ? blane[1] : blane_last[1];
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %7\n\t"
                 "global_load_dwordx4 %2, %5, %6\n\tglobal_load_dwordx4 %3, %5, %7"
                 : "=&v"(b.q[0]), "=&v"(b.v[0]), "=&v"(b.q[1]), "=&v"(b.v[1])
                 : "v"(o0), "v"(o1), "s"(scalar_ptr(pq)), "s"(scalar_ptr(pv)) : "memory");
  };
  auto advanceB = [&]() {
    ++bt;
    if (bt < nit) {
      if (blb + 16 == lb_end) {
        blb = 0; --bjr;
        pq = a.q + (size_t)(i + bjr) * B * d;
        pv = a.v + (size_t)(i + bjr) * B * d;
      } else {
        blb += 16; pq += 16; pv += 16;
      }
    }
  };
  f32x4 xm[KB], xd[KB];
  auto readM = [&](int st) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xm[kb] = *reinterpret_cast<const f32x4*>(&As[st][0][kb * 16 + c16][4 * g4]);
  };
  auto readD = [&](int st) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) xd[kb] = *reinterpret_cast<const f32x4*>(&As[st][1][kb * 16 + c16][4 * g4]);
  };
  // One iteration of a multiplying wave: 16 * KB MFMAs with everything else it has to do placed BETWEEN them -- a wave's own
  // instructions issue in the shadow of its MFMAs for free, while anything it executes outside its MFMA stream competes
  // for the one issue slot per MFMA the SIMD grants beside the other wave's stream.  Placed: the request for iteration it+3
  // (into the ring's fourth slot), the cursor update, and the LDS reads of the next iteration's A fragments (M after the last
  // MFMA that uses the current M fragments, -dM/ds after the last MFMA; their stage was written during it-1).
  // (the staged A tile is zero in every padded row / column / iteration and, for the terminal pair, in all of -dM/ds; the
  //  clamped reads return finite operand values there: no masks on the B side)
  auto iteration = [&](const BSlot& b, BSlot& req, int st_next) {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[kb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xm[kb][s2], b.q[c][s2], acc[kb][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s2 == 0) requestB(req);
      if (s2 == 1) advanceB();
      if (s2 == 3) readM(st_next);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[kb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd[kb][s2], b.v[c][s2], acc[kb][c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    readD(st_next);
    __builtin_amdgcn_sched_barrier(0);
  };
#define SOCMX_WAIT_B(N, S) \
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"((S).q[0]), "+v"((S).q[1]), "+v"((S).v[0]), "+v"((S).v[1]) : "n"(N) : "memory")
  BSlot bs[4];
#pragma unroll
  for (int u = 0; u < 3; ++u) { requestB(bs[u]); advanceB(); }
  __syncthreads();                               // stages 0 .. P visible
  readM(0);
  readD(0);
  // developer switch (tools/ubench/contraction_bench.hip builds this file with it): per-wave cycle counters of the three
  // phases of an iteration -- waiting at the barrier, waiting for the operands, multiplying
#ifdef SOCMX_CONTRACTION_PROF
  long long tp[3] = {0, 0, 0};
  long long tlast = clock64();
#define TICK(k) { const long long tn = clock64(); tp[k] += tn - tlast; tlast = tn; }
#else
#define TICK(k)
#endif
  static_assert(kRing % 4 == 0, "the B request ring (four slots) and the stage ring advance together");
  for (int it0 = 0; it0 < nitp; it0 += kRing) {
#pragma unroll
    for (int u = 0; u < kRing; ++u) {
      if (u % kPeriod == 0) __syncthreads();     // (it = it0 + u) one barrier per period: see the staging waves
      TICK(0)
      SOCMX_WAIT_B(8, bs[u % 4]);                // younger than B(it): B(it+1) B(it+2)
      TICK(1)
      iteration(bs[u % 4], bs[(u + 3) % 4], (u + 1) % kRing);
      TICK(2)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the over-fetched requests of the last trips
#undef SOCMX_WAIT_B
#undef TICK
#ifdef SOCMX_CONTRACTION_PROF
  if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
    for (int k2 = 0; k2 < 3; ++k2) g_contraction_prof[wave][k2] = tp[k2];
    g_contraction_prof[wave][3] = nitp;
  }
#endif
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (mcol[c] < B) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = kb * 16 + 4 * g4 + r;
          if (k < d) a.target[((size_t)i * B + mcol[c]) * d + k] = acc[kb][c][r];
        }
      }
}

// r = sigma^T (nablaV - target), objective += inv_norm * sum w |r|^2, G = 2 w inv_norm sigma r.
// Workgroup = one row i x 64 batch lanes; wave shuffle -> one workspace slot per workgroup (objective_commit).
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
  objective_commit(a, blockIdx.x * gridDim.y + blockIdx.y, gridDim.x * gridDim.y, obj * a.inv_norm);
}

// The same residual for d % 4 == 0, d <= 64 on the MFMA: with S = sigma sigma^T (formed once per workgroup in LDS, its 16
// fragments then live in registers) the row needs ONE product y = S diff:  objective += w diff.y,  G = 2 w inv_norm y.
// Orientation D (16 c x 16 rows) += A (S: 16 c x 4 k) . B (diff^T: 4 k x 16 rows) with k-slot s of lane group g carrying
// k = 16 q + 4 g + s: a lane (row m, group g) then holds diff[m][16 q + 4 g ..+3] (one 16-byte load of nablaV and target per q)
// AND receives y[m][16 cb + 4 g ..+3] -- the same positions, so diff.y is a lane-local sum and G leaves as 16-byte stores.
// The (K+1) B rows are one flat row range; workgroups loop over 16-row tiles (thread-per-row version: 0.63 ms at the
// configs[4] slice for 157 MB of traffic).
__global__ __launch_bounds__(256) void socm_residual_mfma_kernel(const TargetArgs a) {
  constexpr int SS = 68;                               // LDS row stride of sigma / S (64 + 4: conflict-free 16-byte reads)
  __shared__ __attribute__((aligned(16))) float Sg[64 * SS];
  __shared__ __attribute__((aligned(16))) float Ss[64 * SS];
  const int d = a.d, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int nq = (d + 15) >> 4;
  // sigma (zero-padded to 64 x 64) -> LDS, then S = sigma sigma^T (zero beyond d)
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    Sg[r * SS + c] = (r < d && c < d) ? a.sigma[r * d + c] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    float acc = 0.f;
    if (r < d && c < d)
      for (int j = 0; j < d; ++j) acc += Sg[r * SS + j] * Sg[c * SS + j];
    Ss[r * SS + c] = acc;
  }
  __syncthreads();
  // A fragments: S[c = 16 cb + c16][k = 16 q + 4 g4 .. +3]
  f32x4 sf[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) sf[cb][q] = *reinterpret_cast<const f32x4*>(&Ss[(cb * 16 + c16) * SS + q * 16 + 4 * g4]);
  const int64_t R = (int64_t)(a.K + 1) * B;
  const int64_t ntiles = (R + 15) >> 4;
  float obj = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t row = t * 16 + c16;
    const bool valid = row < R;
    const int64_t rc = valid ? row : R - 1;
    const float* nv = a.nablaV + rc * d;
    const float* tg = a.target + rc * d;
    f32x4 df[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      df[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < nq) {
        const int col = min(q * 16 + 4 * g4, d - 4);     // pieces past the row re-read its last piece (S is zero there)
        const f32x4 x = *reinterpret_cast<const f32x4*>(nv + col), y = *reinterpret_cast<const f32x4*>(tg + col);
        df[q] = x - y;
      }
    }
    f32x4 yv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      yv[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (cb < nq) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nq) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
              yv[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(sf[cb][q][s2], df[q][s2], yv[cb], 0, 0, 0);
          }
      }
    }
    const float wm = a.w[(int)(rc % B)];
    float dot = 0.f;
    const float gs = 2.f * wm * a.inv_norm;
    float* go = a.G + rc * d;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int col = cb * 16 + 4 * g4;
      if (cb < nq && col < d) {                          // (d % 4 == 0: a piece is inside the row or outside it)
        dot += (df[cb][0] * yv[cb][0] + df[cb][1] * yv[cb][1]) + (df[cb][2] * yv[cb][2] + df[cb][3] * yv[cb][3]);
        if (valid) *reinterpret_cast<f32x4*>(go + col) = f32x4{gs * yv[cb][0], gs * yv[cb][1], gs * yv[cb][2], gs * yv[cb][3]};
      }
    }
    if (valid) obj += wm * dot;
  }
  obj = wave_sum(obj);
  objective_commit(a, blockIdx.x * 4 + wave, gridDim.x * 4, obj * a.inv_norm);
}

// The operands v, q for d % 4 == 0, d <= 64 on the MFMA (same orientation as socm_residual_mfma_kernel: a lane (row m, group g)
// holds 16-byte pieces [m][16 q + 4 g .. +3] of its inputs AND of each product):
//   z = sqrt(lambda dt) eps + dt u,   v = -sigma^{-T} z,   q = A^T v [+ 2 dt P x]   (OU settings; elementwise for the others)
// Matrices (zero-padded to 64 x 64) live in LDS, one workgroup loops over 16-row tiles of the flat (K B, d) row range.
// (thread-per-element version with LDS tiles: 0.42 ms at the configs[4] slice for 260 MB of traffic)
__global__ __launch_bounds__(256) void socm_prep_mfma_kernel(const PrepArgs a) {
  constexpr int SS = 68;
  __shared__ __attribute__((aligned(16))) float Ms[3][64 * SS];      // sigma^{-T}, A^T, P  as  [out index l][in index c]
  const int d = a.d, B = a.B, K = a.K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int nq = (d + 15) >> 4;
  const bool is_ou = (a.kind == SOCMX_OU_QUADRATIC || a.kind == SOCMX_OU_LINEAR);
  const bool quad = a.kind == SOCMX_OU_QUADRATIC;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int l = e >> 6, c = e & 63;
    const bool in = l < d && c < d;
    Ms[0][l * SS + c] = in ? a.sit[l * d + c] : 0.f;
    Ms[1][l * SS + c] = (in && is_ou) ? a.A[c * d + l] : 0.f;           // (A^T)[l][n] = A[n][l]
    Ms[2][l * SS + c] = (in && quad) ? a.P[l * d + c] : 0.f;
  }
  __syncthreads();
  // D (16 l x 16 rows) += A (M: 16 l x 4 c) . B (x^T: 4 c x 16 rows): out[lb] = sum_q M[lb][q] x[q]
  auto product = [&](const float* M, const f32x4 (&x)[4], f32x4 (&out)[4]) {
#pragma unroll
    for (int lb = 0; lb < 4; ++lb) {
      out[lb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lb < nq) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nq) {
            const f32x4 mf = *reinterpret_cast<const f32x4*>(&M[(lb * 16 + c16) * SS + q * 16 + 4 * g4]);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) out[lb] = __builtin_amdgcn_mfma_f32_16x16x4f32(mf[s2], x[q][s2], out[lb], 0, 0, 0);
          }
      }
    }
  };
  const int64_t R = (int64_t)K * B;
  const int64_t ntiles = (R + 15) >> 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < ntiles; t += (int64_t)gridDim.x * 4) {
    const int64_t row = t * 16 + c16;
    const bool valid = row < R;
    const int64_t rc = valid ? row : R - 1;
    const int j = (int)(rc / B);
    const float dt = a.frac ? a.frac[rc] : (a.ts[j + 1] - a.ts[j]);
    const float ce = a.sqrt_lmbd * sqrtf(dt);
    f32x4 z[4], xs[4], v[4], qv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      z[q] = f32x4{0.f, 0.f, 0.f, 0.f}; xs[q] = z[q];
      if (q < nq) {
        const int64_t off = rc * d + min(q * 16 + 4 * g4, d - 4);      // pieces past the row re-read its last piece (zero matrix columns)
        const f32x4 e4 = *reinterpret_cast<const f32x4*>(a.noises + off), u4 = *reinterpret_cast<const f32x4*>(a.controls + off);
        z[q] = ce * e4 + dt * u4;
        if (!is_ou || quad) xs[q] = *reinterpret_cast<const f32x4*>(a.states + off);
      }
    }
    product(Ms[0], z, v);
#pragma unroll
    for (int lb = 0; lb < 4; ++lb) v[lb] = -v[lb];
    if (is_ou) {
      product(Ms[1], v, qv);
      if (quad) {
        f32x4 px[4];
        product(Ms[2], xs, px);
#pragma unroll
        for (int lb = 0; lb < 4; ++lb) qv[lb] += (2.f * dt) * px[lb];
      }
    } else {
#pragma unroll
      for (int lb = 0; lb < 4; ++lb) {
        qv[lb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int col = lb * 16 + 4 * g4;
        if (lb < nq && col < d) {
          const f32x4 kap = *reinterpret_cast<const f32x4*>(a.kappa + col);
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const float xl = xs[lb][s2];
            qv[lb][s2] = -(8.f * kap[s2] * xl * xl + 4.f * kap[s2] * (xl * xl - 1.f)) * v[lb][s2];
          }
        }
      }
    }
    if (valid) {
#pragma unroll
      for (int lb = 0; lb < 4; ++lb) {
        const int col = lb * 16 + 4 * g4;
        if (lb < nq && col < d) {
          *reinterpret_cast<f32x4*>(a.v + rc * d + col) = v[lb];
          *reinterpret_cast<f32x4*>(a.q + rc * d + col) = qv[lb];
        }
      }
    }
  }
}

// ---- backward: gradients w.r.t. the pair matrices ---------------------------------------------------
struct TargetBwdArgs {
  int d, K, B;
  const float *G, *q, *v, *gT;   // (Kp,B,d), (K,B,d), (K,B,d), (B,d)
  const float *gout;             // (1,) upstream gradient of the objective on the device, or NULL (= 1)
  float *gM, *gdM;               // (Np,d,d): d obj/dM, d obj/d(dM)   -- NET: d obj/d net, d obj/d dnet
  const float *net, *dnet, *delta, *gamma;   // NET only
  float *ggamma_part;            // NET only: (Np * kblocks * lblocks,) partial sums of d obj/d gamma
};

// One wave per (pair (i,j), 16x16 block of the d x d matrix); the batch is the MFMA reduction dimension:
//   gM[i,j][k][l] = -sum_m G[i,m,k] q[j,m,l]     gdM[i,j][k][l] = +sum_m G[i,m,k] v[j,m,l]
//   D (16 k x 16 l) += A (G[i,m,k], 16 x 4 m) . B (q[j,m,l], 4 m x 16 l)                    v_mfma_f32_16x16x4_f32
// Consecutive waves take consecutive j of one row i, so the A operand stays in L1/L2.  NET = true chains the
// derivative of M = e I + (1-e) net, dM = gamma e (net - I) + (1-e) dnet in the epilogue.
template <bool NET>
__global__ __launch_bounds__(256, 2) void socm_target_bwd_mfma_kernel(const TargetBwdArgs a) {
  const int d = a.d, K = a.K, B = a.B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t np = (int64_t)(K + 1) * (K + 2) / 2;
  const int64_t p = (int64_t)blockIdx.x * 4 + wave;
  if (p >= np) return;
  // invert the i-major triangular numbering: rows counted from the end have 1, 2, 3, ... pairs
  const int64_t pe = np - 1 - p;
  int r = (int)((sqrtf(8.f * (float)pe + 1.f) - 1.f) * 0.5f);
  while ((int64_t)(r + 1) * (r + 2) / 2 <= pe) ++r;
  while ((int64_t)r * (r + 1) / 2 > pe) --r;
  r = __builtin_amdgcn_readfirstlane(r);     // (came through the float unit: pin it -- and every pointer derived from it -- to scalar registers)
  const int i = K - r;
  const int jr = (int)(p - pair_row_offset(i, K));
  const int j = i + jr;
  const bool last = (j == K);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = blockIdx.y * 16, lb = blockIdx.z * 16;
  const int k = kb + c16, l = lb + c16;
  const bool okl = l < d;
  // Lean instruction stream (this kernel is bound by instruction issue, not by the MFMA pipe or memory: 10 vector
  // instructions per MFMA before, `profiles/r2`): wave-uniform base pointers advanced per chunk + four loop-invariant lane
  // offsets per operand instead of 64-bit address arithmetic per load; NO operand masks -- lanes of padded rows / columns
  // read clamped (finite) addresses and feed accumulator elements that are never stored; the terminal pair skips the v
  // products; only a ragged last chunk zeroes its missing batch rows.
  const uint32_t cola = (uint32_t)min(k, d - 1) * 4u, colb = (uint32_t)min(l, d - 1) * 4u;
  uint32_t offa[4], offb[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    offa[u] = (uint32_t)((4 * u + g4) * d) * 4u + cola;
    offb[u] = (uint32_t)((4 * u + g4) * d) * 4u + colb;
  }
  const char* Ap = reinterpret_cast<const char*>(a.G + (size_t)i * B * d);
  const char* Bq = reinterpret_cast<const char*>(last ? a.gT : a.q + (size_t)j * B * d);
  const char* Bv = reinterpret_cast<const char*>(a.v + (size_t)(last ? 0 : j) * B * d);
  const size_t chunk_bytes = (size_t)16 * d * 4;
  f32x4 accq = {0.f, 0.f, 0.f, 0.f}, accv = {0.f, 0.f, 0.f, 0.f};
  struct Chunk { float af[4], qf[4], vf[4]; };
  const char *ap = Ap, *qp = Bq, *vp = Bv;     // running chunk bases (scalar registers: the loads need no vector address arithmetic)
  auto load = [&](Chunk& ch) {                 // 12 loads: the next 16 rows (4u + g4), this lane's k / l column
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ch.af[u] = *reinterpret_cast<const float*>(ap + (size_t)offa[u]);
      ch.qf[u] = *reinterpret_cast<const float*>(qp + (size_t)offb[u]);
      ch.vf[u] = *reinterpret_cast<const float*>(vp + (size_t)offb[u]);
    }
    ap += chunk_bytes; qp += chunk_bytes; vp += chunk_bytes;
  };
  auto multiply = [&](const Chunk& ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) accq = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.af[u], ch.qf[u], accq, 0, 0, 0);
    if (!last) {
#pragma unroll
      for (int u = 0; u < 4; ++u) accv = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.af[u], ch.vf[u], accv, 0, 0, 0);
    }
  };
  const int nfull = B >> 4;                    // whole 16-row chunks
  if (nfull <= 8) {
    // training batches up to 128 (BASELINE configs[1..3]): all chunks are requested up front (up to 96 loads per lane, one
    // memory round trip per wave instead of eight half-overlapped ones: 58 -> 43 us at cfg3) and multiplied as they arrive
    Chunk ch[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nfull) load(ch[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nfull) multiply(ch[c]);
  } else {
  // the next chunk is requested before the current one is multiplied
  Chunk c0, c1;
  if (nfull > 0) load(c0);
  for (int c = 0; c < nfull; c += 2) {
    if (c + 1 < nfull) load(c1);
    multiply(c0);
    if (c + 1 < nfull) {
      if (c + 2 < nfull) load(c0);
      multiply(c1);
    }
  }
  }
  if (B & 15) {                                // ragged last chunk: rows past the batch are clamped and their A values zeroed
    const int m0 = nfull * 16;
    Chunk ct;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + 4 * u + g4;
      const size_t ro = (size_t)min(m, B - 1) * d * 4;
      const float av = *reinterpret_cast<const float*>(Ap + ro + cola);
      ct.af[u] = m < B ? av : 0.f;
      ct.qf[u] = *reinterpret_cast<const float*>(Bq + ro + colb);
      ct.vf[u] = *reinterpret_cast<const float*>(Bv + ro + colb);
    }
    multiply(ct);
  }
  const float go = a.gout ? a.gout[0] : 1.f;
  float e = 0.f, gam = 0.f, dl = 0.f, part = 0.f;
  if (NET) { gam = a.gamma[0]; dl = a.delta[p]; e = expf(-gam * dl); }
  {
    // epilogue: wave-uniform block base + one lane offset per accumulator row, constants folded
    const size_t blk = (size_t)p * d * d + (size_t)kb * d;
    const char* netb = reinterpret_cast<const char*>(a.net + blk);
    const char* dnetb = reinterpret_cast<const char*>(a.dnet + blk);
    char* gMb = reinterpret_cast<char*>(a.gM + blk);
    char* gdMb = reinterpret_cast<char*>(a.gdM + blk);
    const float f = NET ? 1.f - e : 1.f, ge = gam * e;
    const float kq = -f * go, kv = ge * go, kd = f * go;                  // gM = kq accq + kv accv,  gdM = kd accv
    const float c1g = dl * e * go, c2g = e * (1.f - gam * dl) * go;       // d/dgamma: -accq c1 nmi + accv (c2 nmi + c1 dnet)
    float nt[4], dn[4];
    bool ok[4];
    uint32_t off[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int kl = 4 * g4 + rr;              // row inside the block
      ok[rr] = kb + kl < d && okl;
      off[rr] = (uint32_t)(min(kl, d - 1 - kb) * d + min(l, d - 1)) * 4u;
      if (NET) {
        nt[rr] = *reinterpret_cast<const float*>(netb + (size_t)off[rr]);
        dn[rr] = *reinterpret_cast<const float*>(dnetb + (size_t)off[rr]);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float aq_ = accq[rr], av_ = accv[rr];
      float gm_, gd_;
      if (NET) {
        const float nmi = nt[rr] - ((kb + 4 * g4 + rr == l) ? 1.f : 0.f);
        const float t = av_ * fmaf(c2g, nmi, c1g * dn[rr]) - aq_ * (c1g * nmi);
        part += ok[rr] ? t : 0.f;
        gm_ = fmaf(kq, aq_, kv * av_);
        gd_ = kd * av_;
      } else {
        gm_ = -go * aq_;
        gd_ = go * av_;
      }
      if (ok[rr]) {
        *reinterpret_cast<float*>(gMb + (size_t)off[rr]) = gm_;
        *reinterpret_cast<float*>(gdMb + (size_t)off[rr]) = gd_;
      }
    }
  }
  if (NET) {
    part = wave_sum(part);
    if (lane == 0) a.ggamma_part[((size_t)p * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z] = part;
  }
}

// d <= 16, two rows per wave: the pairs (i0, j) and (i0 + 1, j) share the q_j / v_j fragments (16 loads and one index
// inversion per 16 MFMAs instead of 12 and one per 8; same item numbering as socm_target_bwd_lds2_kernel).
template <bool NET>
__global__ __launch_bounds__(256, 2) void socm_target_bwd_mfma2_kernel(const TargetBwdArgs a, int64_t nitems) {
  const int d = a.d, K = a.K, B = a.B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t x = (int64_t)blockIdx.x * 4 + wave;
  if (x >= nitems) return;
  const float kp2 = (float)(K + 2);
  int r2 = (int)((kp2 - sqrtf(fmaxf(kp2 * kp2 - 4.f * (float)x, 0.f))) * 0.5f);
  r2 = max(0, min(r2, (K + 2) / 2 - 1));
  while ((int64_t)(r2 + 1) * (K + 2 - (r2 + 1)) <= x) ++r2;
  while ((int64_t)r2 * (K + 2 - r2) > x) --r2;
  r2 = __builtin_amdgcn_readfirstlane(r2);
  const int i0 = 2 * r2, i1 = i0 + 1;
  const int j = i0 + (int)(x - (int64_t)r2 * (K + 2 - r2));
  const bool two = i1 <= K && j >= i1;
  const bool last = (j == K);
  const int64_t p0 = pair_row_offset(i0, K) + (j - i0);
  const int64_t p1 = two ? pair_row_offset(i1, K) + (j - i1) : p0;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = blockIdx.y * 16, lb = blockIdx.z * 16;
  const int k = kb + c16, l = lb + c16;
  const bool okl = l < d;
  const uint32_t cola = (uint32_t)min(k, d - 1) * 4u, colb = (uint32_t)min(l, d - 1) * 4u;
  uint32_t offa[4], offb[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    offa[u] = (uint32_t)((4 * u + g4) * d) * 4u + cola;
    offb[u] = (uint32_t)((4 * u + g4) * d) * 4u + colb;
  }
  const char* A0 = reinterpret_cast<const char*>(a.G + (size_t)i0 * B * d);
  const char* A1 = reinterpret_cast<const char*>(a.G + (size_t)(two ? i1 : i0) * B * d);
  const char* Bq = reinterpret_cast<const char*>(last ? a.gT : a.q + (size_t)j * B * d);
  const char* Bv = reinterpret_cast<const char*>(a.v + (size_t)(last ? 0 : j) * B * d);
  const size_t chunk_bytes = (size_t)16 * d * 4;
  f32x4 accq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  struct Chunk { float a0[4], a1[4], qf[4], vf[4]; };
  const char *ap0 = A0, *ap1 = A1, *qp = Bq, *vp = Bv;
  auto load = [&](Chunk& ch) {                 // 16 loads: the next 16 rows (4u + g4), this lane's k / l column
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ch.a0[u] = *reinterpret_cast<const float*>(ap0 + (size_t)offa[u]);
      ch.a1[u] = *reinterpret_cast<const float*>(ap1 + (size_t)offa[u]);
      ch.qf[u] = *reinterpret_cast<const float*>(qp + (size_t)offb[u]);
      ch.vf[u] = *reinterpret_cast<const float*>(vp + (size_t)offb[u]);
    }
    ap0 += chunk_bytes; ap1 += chunk_bytes; qp += chunk_bytes; vp += chunk_bytes;
  };
  auto multiply = [&](const Chunk& ch) {       // (an absent second pair / the terminal pair's v products are computed and dropped)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      accq[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.a0[u], ch.qf[u], accq[0], 0, 0, 0);
      accq[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.a1[u], ch.qf[u], accq[1], 0, 0, 0);
      accv[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.a0[u], ch.vf[u], accv[0], 0, 0, 0);
      accv[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ch.a1[u], ch.vf[u], accv[1], 0, 0, 0);
    }
  };
  const int nfull = B >> 4;
  Chunk c0, c1;
  if (nfull > 0) load(c0);
  for (int c = 0; c < nfull; c += 2) {
    if (c + 1 < nfull) load(c1);
    multiply(c0);
    if (c + 1 < nfull) {
      if (c + 2 < nfull) load(c0);
      multiply(c1);
    }
  }
  if (B & 15) {                                // ragged last chunk: rows past the batch are clamped and their A values zeroed
    const int m0 = nfull * 16;
    Chunk ct;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = m0 + 4 * u + g4;
      const size_t ro = (size_t)min(m, B - 1) * d * 4;
      const float av0 = *reinterpret_cast<const float*>(A0 + ro + cola), av1 = *reinterpret_cast<const float*>(A1 + ro + cola);
      ct.a0[u] = m < B ? av0 : 0.f;
      ct.a1[u] = m < B ? av1 : 0.f;
      ct.qf[u] = *reinterpret_cast<const float*>(Bq + ro + colb);
      ct.vf[u] = *reinterpret_cast<const float*>(Bv + ro + colb);
    }
    multiply(ct);
  }
  const float go = a.gout ? a.gout[0] : 1.f;
  const float gam = NET ? a.gamma[0] : 0.f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (h == 1 && !two) break;
    const int64_t p = h ? p1 : p0;
    float e = 0.f, dl = 0.f, part = 0.f;
    if (NET) { dl = a.delta[p]; e = expf(-gam * dl); }
    const size_t blk = (size_t)p * d * d + (size_t)kb * d;
    const char* netb = reinterpret_cast<const char*>(a.net + blk);
    const char* dnetb = reinterpret_cast<const char*>(a.dnet + blk);
    char* gMb = reinterpret_cast<char*>(a.gM + blk);
    char* gdMb = reinterpret_cast<char*>(a.gdM + blk);
    const float f = NET ? 1.f - e : 1.f, ge = gam * e;
    const float kq = -f * go, kv = ge * go, kd = f * go;                  // gM = kq accq + kv accv,  gdM = kd accv
    const float c1g = dl * e * go, c2g = e * (1.f - gam * dl) * go;       // d/dgamma: -accq c1 nmi + accv (c2 nmi + c1 dnet)
    const float vmask = last ? 0.f : 1.f;                                 // the terminal pair has no v operand
    float nt[4], dn[4];
    bool ok[4];
    uint32_t off[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int kl = 4 * g4 + rr;
      ok[rr] = kb + kl < d && okl;
      off[rr] = (uint32_t)(min(kl, d - 1 - kb) * d + min(l, d - 1)) * 4u;
      if (NET) {
        nt[rr] = *reinterpret_cast<const float*>(netb + (size_t)off[rr]);
        dn[rr] = *reinterpret_cast<const float*>(dnetb + (size_t)off[rr]);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const float aq_ = accq[h][rr], av_ = vmask * accv[h][rr];
      float gm_, gd_;
      if (NET) {
        const float nmi = nt[rr] - ((kb + 4 * g4 + rr == l) ? 1.f : 0.f);
        const float t = av_ * fmaf(c2g, nmi, c1g * dn[rr]) - aq_ * (c1g * nmi);
        part += ok[rr] ? t : 0.f;
        gm_ = fmaf(kq, aq_, kv * av_);
        gd_ = kd * av_;
      } else {
        gm_ = -go * aq_;
        gd_ = go * av_;
      }
      if (ok[rr]) {
        *reinterpret_cast<float*>(gMb + (size_t)off[rr]) = gm_;
        *reinterpret_cast<float*>(gdMb + (size_t)off[rr]) = gd_;
      }
    }
    if (NET) {
      part = wave_sum(part);
      if (lane == 0) a.ggamma_part[((size_t)p * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z] = part;
    }
  }
}

// d > 16: one workgroup (4 waves) per pair; wave w owns k-block 4*blockIdx.y + w and LB l-blocks, i.e. 2*LB
// accumulators: an A fragment (G rows) is loaded once per 16 batch rows and multiplied into all of them, the B
// fragments (q, v rows) are shared by the four waves through L1.  Operands of the next 16 batch rows are requested
// before the current ones are multiplied.
template <bool NET, int LB>
__global__ __launch_bounds__(256, 2) void socm_target_bwd_wide_kernel(const TargetBwdArgs a) {
  const int d = a.d, K = a.K, B = a.B;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t np = (int64_t)(K + 1) * (K + 2) / 2;
  const int64_t p = blockIdx.x;
  const int64_t pe = np - 1 - p;
  int r = (int)((sqrtf(8.f * (float)pe + 1.f) - 1.f) * 0.5f);
  while ((int64_t)(r + 1) * (r + 2) / 2 <= pe) ++r;
  while ((int64_t)r * (r + 1) / 2 > pe) --r;
  r = __builtin_amdgcn_readfirstlane(r);     // (came through the float unit: pin it -- and every pointer derived from it -- to scalar registers)
  const int i = K - r;
  const int j = i + (int)(p - pair_row_offset(i, K));
  const bool last = (j == K);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = (blockIdx.y * 4 + wave) * 16;
  if (kb >= d) return;                                   // (no barriers in this kernel)
  const int lb0 = blockIdx.z * LB * 16;
  const int k = kb + c16;
  const bool okk = k < d;
  const float* Ap = a.G + (size_t)i * B * d + (okk ? k : d - 1);
  const float* Bq = last ? a.gT : a.q + (size_t)j * B * d;
  const float* Bv = a.v + (size_t)(last ? 0 : j) * B * d;
  int lcol[LB];
  bool okl[LB];
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int l = lb0 + b * 16 + c16;
    okl[b] = l < d;
    lcol[b] = okl[b] ? l : d - 1;
  }
  f32x4 accq[LB], accv[LB];
#pragma unroll
  for (int b = 0; b < LB; ++b) { accq[b] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  struct Chunk { float af[4], qf[LB][4], vf[LB][4]; };
  auto load = [&](int m0, Chunk& c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = min(m0 + 4 * u + g4, B - 1);
      const size_t off = (size_t)m * d;
      c.af[u] = Ap[off];
#pragma unroll
      for (int b = 0; b < LB; ++b) {
        c.qf[b][u] = Bq[off + lcol[b]];
        c.vf[b][u] = Bv[off + lcol[b]];
      }
    }
  };
  auto consume = [&](int m0, const Chunk& c) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool okm = m0 + 4 * u + g4 < B;
      const float af = (okm && okk) ? c.af[u] : 0.f;
#pragma unroll
      for (int b = 0; b < LB; ++b)
        accq[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, okl[b] ? c.qf[b][u] : 0.f, accq[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < LB; ++b)
        accv[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, (okl[b] && !last) ? c.vf[b][u] : 0.f, accv[b], 0, 0, 0);
    }
  };
  Chunk c0, c1;
  load(0, c0);
  for (int m0 = 0; m0 < B; m0 += 32) {
    if (m0 + 16 < B) load(m0 + 16, c1);
    consume(m0, c0);
    if (m0 + 16 >= B) break;
    if (m0 + 32 < B) load(m0 + 32, c0);
    consume(m0 + 16, c1);
  }
  const float go = a.gout ? a.gout[0] : 1.f;
  const size_t base = (size_t)p * d * d;
  float e = 0.f, gam = 0.f, dl = 0.f, part = 0.f;
  if (NET) { gam = a.gamma[0]; dl = a.delta[p]; e = expf(-gam * dl); }
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int l = lb0 + b * 16 + c16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int kk = kb + 4 * g4 + rr;
      if (kk < d && okl[b]) {
        const size_t idx = base + (size_t)kk * d + l;
        const float gm = -accq[b][rr] * go, gd = accv[b][rr] * go;
        if (NET) {
          const float nmi = a.net[idx] - (kk == l ? 1.f : 0.f);
          a.gM[idx] = (1.f - e) * gm + gam * e * gd;
          a.gdM[idx] = (1.f - e) * gd;
          part += gm * dl * e * nmi + gd * (e * (1.f - gam * dl) * nmi + dl * e * a.dnet[idx]);
        } else {
          a.gM[idx] = gm;
          a.gdM[idx] = gd;
        }
      }
    }
  }
  if (NET) {
    part = wave_sum(part);
    // the caller's buffer has one slot per (pair, k-block, l-block): this wave covers LB of them (sum in the first)
    const int nb = (d + 15) >> 4;
    if (lane < LB && (int)blockIdx.z * LB + lane < nb)
      a.ggamma_part[((size_t)p * nb + (kb >> 4)) * nb + blockIdx.z * LB + lane] = lane == 0 ? part : 0.f;
  }
}

// Epilogue of the d <= 64 LDS backward kernels for one pair p: this wave's 16 x (16 LB) block of d obj / d (M, dM/ds) --
// chained to (net, dnet) and the d obj / d gamma partial when NET -- from its accumulators.
template <bool NET, int LB>
__device__ __forceinline__ void bwd_lds_epilogue(const TargetBwdArgs& a, int64_t p, const f32x4 (&accq)[LB],
                                                 const f32x4 (&accv)[LB], int kb, int wave, int lane) {
  const int d = a.d;
  const int c16 = lane & 15, g4 = lane >> 4;
  const float go = a.gout ? a.gout[0] : 1.f;
  const size_t base = (size_t)p * d * d;
  float e = 0.f, gam = 0.f, dl = 0.f, part = 0.f;
  if (NET) { gam = a.gamma[0]; dl = a.delta[p]; e = expf(-gam * dl); }
  if (kb + 16 <= d && 16 * LB <= d) {
    // Whole 16 x (16 LB) block (d = 64: always).  The epilogue runs beside the other workgroups' MFMA streams, where every
    // instruction costs about one MFMA time (see socm_target_lds4_kernel): wave-uniform base pointers + four lane offsets
    // (one per accumulator row, the l-block as an immediate) instead of 64-bit address arithmetic per element, no
    // predicates, all 2 x 4 LB loads in flight before the arithmetic, constants folded (844 -> ~300 instructions).
    const size_t blk = base + (size_t)kb * d;
    const char* netb = reinterpret_cast<const char*>(a.net + blk);
    const char* dnetb = reinterpret_cast<const char*>(a.dnet + blk);
    char* gMb = reinterpret_cast<char*>(a.gM + blk);
    char* gdMb = reinterpret_cast<char*>(a.gdM + blk);
    uint32_t off[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) off[rr] = (uint32_t)((4 * g4 + rr) * d + c16) * 4u;
    const float f = NET ? 1.f - e : 1.f, ge = gam * e;
    const float kq = -f * go, kv = ge * go, kd = f * go;                  // gM = kq accq + kv accv,  gdM = kd accv
    const float c1 = dl * e * go, c2 = e * (1.f - gam * dl) * go;         // d/dgamma: -accq c1 nmi + accv (c2 nmi + c1 dnet)
    float nt[LB][4], dn[LB][4];
    if (NET) {
#pragma unroll
      for (int b = 0; b < LB; ++b)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          nt[b][rr] = *reinterpret_cast<const float*>(netb + (size_t)off[rr] + 64 * b);
          dn[b][rr] = *reinterpret_cast<const float*>(dnetb + (size_t)off[rr] + 64 * b);
        }
    }
#pragma unroll
    for (int b = 0; b < LB; ++b)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float aq_ = accq[b][rr], av_ = accv[b][rr];
        if (NET) {
          const float nmi = nt[b][rr] - ((b == wave && c16 == 4 * g4 + rr) ? 1.f : 0.f);
          part += av_ * fmaf(c2, nmi, c1 * dn[b][rr]) - aq_ * (c1 * nmi);
          *reinterpret_cast<float*>(gMb + (size_t)off[rr] + 64 * b) = fmaf(kq, aq_, kv * av_);
          *reinterpret_cast<float*>(gdMb + (size_t)off[rr] + 64 * b) = kd * av_;
        } else {
          *reinterpret_cast<float*>(gMb + (size_t)off[rr] + 64 * b) = -go * aq_;
          *reinterpret_cast<float*>(gdMb + (size_t)off[rr] + 64 * b) = go * av_;
        }
      }
  } else
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    const int l = b * 16 + c16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int kk = kb + 4 * g4 + rr;
      if (kk < d && l < d) {
        const size_t idx = base + (size_t)kk * d + l;
        const float gm = -accq[b][rr] * go, gd = accv[b][rr] * go;
        if (NET) {
          const float nmi = a.net[idx] - (kk == l ? 1.f : 0.f);
          a.gM[idx] = (1.f - e) * gm + gam * e * gd;
          a.gdM[idx] = (1.f - e) * gd;
          part += gm * dl * e * nmi + gd * (e * (1.f - gam * dl) * nmi + dl * e * a.dnet[idx]);
        } else {
          a.gM[idx] = gm;
          a.gdM[idx] = gd;
        }
      }
    }
  }
  if (NET) {
    part = wave_sum(part);
    const int nb = (d + 15) >> 4;         // the caller's buffer has one slot per (pair, k-block, l-block)
    if (lane < nb) a.ggamma_part[((size_t)p * nb + wave) * nb + lane] = lane == 0 ? part : 0.f;
  }}

// d <= 64, d % 4 == 0: the same contraction with the three operand tiles of 16 batch rows (G_i, q_j, v_j: 16 x d floats
// each) staged through a double-buffered LDS tile, TRANSPOSED: Ts[tensor][column][batch row], 20 floats per column.  The
// batch is the MFMA reduction index and its assignment to (MFMA u, lane group g4) is free, so row m = 4*g4 + u: the four
// u-values of a lane are four consecutive floats of one column -- nine ds_read_b128 per 16-row chunk and wave feed its
// 32 MFMAs (the row-major tile needed 36 ds_read_b32, each predicated and waited for right before its MFMA: 8.2 ms at the
// configs[4] slice, 57 % MFMA-busy).  Loaders: waves 0..2 take one tensor each; lane (rg = lane & 3, pc = lane >> 2) loads
// the 4 x 4 block rows 4rg.., columns 4pc.. with four 16-byte loads and writes it as four 16-byte column pieces (an
// 8-lane group covers all 32 banks exactly once on the write and on the read side).  Columns >= d and rows >= B are zero
// in LDS, so the consumer needs no predicates.
constexpr int kBwdStride = 20;

// Two rows per workgroup: the pairs (i0, j) and (i0 + 1, j) share q_j and v_j.  At d = 64 every pair of the one-row kernel
// pulls 3 x 128 KiB of operand rows out of L2 / the memory side (31 GB per launch at the configs[4] slice, 5 TB/s while it
// runs); with two G tiles against one (q, v) tile that is 2 x 128 + 256 KiB per two pairs (-33 %), 10 instead of 18 fragment
// reads and one instead of two barriers per 64 MFMAs.  Four waves = four loaders (G_i0, G_i1, q_j, v_j), four k-blocks.
// Item x of the grid = (row pair r2, j >= 2 r2); rows counted in pairs hold K - 2 r2 + 1 items, C(r2) = r2 (K + 2 - r2) before.
template <bool NET, int LB>
__global__ __launch_bounds__(256, 2) void socm_target_bwd_lds2_kernel(const TargetBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float Ts[2][4][64][kBwdStride];   // [stage][G_i0, G_i1, q, v][column][batch row]
  const int d = a.d, K = a.K, B = a.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t x = blockIdx.x;
  const float kp2 = (float)(K + 2);
  int r2 = (int)((kp2 - sqrtf(fmaxf(kp2 * kp2 - 4.f * (float)x, 0.f))) * 0.5f);
  r2 = max(0, min(r2, (K + 2) / 2 - 1));
  while ((int64_t)(r2 + 1) * (K + 2 - (r2 + 1)) <= x) ++r2;
  while ((int64_t)r2 * (K + 2 - r2) > x) --r2;
  r2 = __builtin_amdgcn_readfirstlane(r2);   // (came through the float unit: pin it -- and every pointer derived from it -- to scalar registers)
  const int i0 = 2 * r2, i1 = i0 + 1;
  const int j = i0 + (int)(x - (int64_t)r2 * (K + 2 - r2));
  const bool two = i1 <= K && j >= i1;       // the pair (i1, j) exists
  const bool last = (j == K);
  const int64_t p0 = pair_row_offset(i0, K) + (j - i0);
  const int64_t p1 = two ? pair_row_offset(i1, K) + (j - i1) : p0;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int kb = wave * 16;
  const bool wave_on = kb < d;
  // loader role: tensor `wave`, rows 4*rg .. 4*rg+3 of the chunk, columns 4*pc .. 4*pc+3
  const int rg = lane & 3, pc = lane >> 2;
  const bool piece_on = 4 * pc < d;
  const bool tensor_zero = (wave == 1 && !two) || (wave == 3 && last);      // no second pair / the terminal pair has no v operand
  const float* src = wave == 0 ? a.G + (size_t)i0 * B * d
                   : wave == 1 ? a.G + (size_t)(two ? i1 : i0) * B * d
                   : wave == 2 ? (last ? a.gT : a.q + (size_t)j * B * d)
                               : a.v + (size_t)(last ? 0 : j) * B * d;
  f32x4 pr[2][4];                         // chunks c+1 and c+2 in flight (slot = chunk & 1): pr[slot][row][column]
  auto gload = [&](int m0, int sl) {
    if (piece_on && !tensor_zero) {
#pragma unroll
      for (int s = 0; s < 4; ++s) pr[sl][s] = uload4(src, min(m0 + 4 * rg + s, B - 1) * d + 4 * pc);
    }
  };
  auto stage = [&](int m0, int st, int sl) {   // rows past the batch, columns past d and absent tensors are zeroed here
    float* col = &Ts[st][wave][4 * pc][4 * rg];
    if (!piece_on || tensor_zero) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(col + s * kBwdStride) = z;
    } else if (m0 + 16 <= B) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        *reinterpret_cast<f32x4*>(col + s * kBwdStride) = f32x4{pr[sl][0][s], pr[sl][1][s], pr[sl][2][s], pr[sl][3][s]};
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        f32x4 t;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) t[rr] = (m0 + 4 * rg + rr < B) ? pr[sl][rr][s] : 0.f;
        *reinterpret_cast<f32x4*>(col + s * kBwdStride) = t;
      }
    }
  };
  f32x4 accq0[LB], accv0[LB], accq1[LB], accv1[LB];
#pragma unroll
  for (int b = 0; b < LB; ++b) {
    accq0[b] = f32x4{0.f, 0.f, 0.f, 0.f}; accv0[b] = accq0[b]; accq1[b] = accq0[b]; accv1[b] = accq0[b];
  }
  const int nch = (B + 15) >> 4;
  gload(0, 0);
  if (nch > 1) gload(16, 1);
  stage(0, 0, 0);
  if (nch > 2) gload(32, 0);
  // chunk loop (unrolled by two for static register slots, loader work between the MFMA groups,
  // one copy for waves without a k-block of their own)
  auto chunk_loop = [&](auto with_mfma) {
    for (int c0 = 0; c0 < nch; c0 += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = c0 + h;             // c & 1 == h
        if (c < nch) {
          __syncthreads();                // stage h visible; everyone is done reading stage 1-h
          if constexpr (decltype(with_mfma)::value) {
            const f32x4 ga0 = *reinterpret_cast<const f32x4*>(&Ts[h][0][kb + c16][4 * g4]);
            const f32x4 ga1 = *reinterpret_cast<const f32x4*>(&Ts[h][1][kb + c16][4 * g4]);
            f32x4 qb[LB], vb[LB];
#pragma unroll
            for (int b = 0; b < LB; ++b) {
              qb[b] = *reinterpret_cast<const f32x4*>(&Ts[h][2][b * 16 + c16][4 * g4]);
              vb[b] = *reinterpret_cast<const f32x4*>(&Ts[h][3][b * 16 + c16][4 * g4]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
              for (int b = 0; b < LB; ++b)
                accq0[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0[u], qb[b][u], accq0[b], 0, 0, 0);
#pragma unroll
              for (int b = 0; b < LB; ++b)
                accv0[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga0[u], vb[b][u], accv0[b], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (u == 0 && c + 1 < nch) stage((c + 1) * 16, 1 - h, 1 - h);
              if (u == 1 && c + 3 < nch) gload((c + 3) * 16, 1 - h);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int b = 0; b < LB; ++b)
                accq1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1[u], qb[b][u], accq1[b], 0, 0, 0);
#pragma unroll
              for (int b = 0; b < LB; ++b)
                accv1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga1[u], vb[b][u], accv1[b], 0, 0, 0);
            }
          } else if (c + 1 < nch) {
            stage((c + 1) * 16, 1 - h, 1 - h);
            if (c + 3 < nch) gload((c + 3) * 16, 1 - h);
          }
        }
      }
    }
  };
  if (!wave_on) {
    chunk_loop(std::false_type{});
    return;
  }
  chunk_loop(std::true_type{});
  bwd_lds_epilogue<NET, LB>(a, p0, accq0, accv0, kb, wave, lane);
  if (two) bwd_lds_epilogue<NET, LB>(a, p1, accq1, accv1, kb, wave, lane);
}

// ---- column sums (bias gradients) ---------------------------------------------------------------------
// out[c] = sum_r x[r][c] for a tall (R, C) row-major matrix: HBM-bound, one pass.  Stage 1: workgroup b sums the
// rows b, b+nblk, ... (lanes along the columns: coalesced; 256/CW row lanes per workgroup) into partial[b][:];
// stage 2 adds the nblk partials in a fixed order (deterministic).
// MASKED: x = gy, y = the ReLU output saved by the forward; gz = gy * (y > 0) is written out and summed
// (ReLU backward fused with the bias-gradient reduction: one pass instead of two kernels and three passes)
template <bool MASKED>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             float* __restrict__ gz, int64_t R, int C, int CW,
                                                             float* __restrict__ partial) {
  __shared__ float red[256];
  const int c = threadIdx.x % CW, rl = threadIdx.x / CW, nrl = 256 / CW;
  const int64_t step = (int64_t)gridDim.x * nrl;
  for (int c0 = 0; c0 < C; c0 += CW) {
    const int cc = c0 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (cc < C) {
      int64_t r = (int64_t)blockIdx.x * nrl + rl;
      auto at = [&](int64_t rr) -> float {
        const int64_t idx = rr * C + cc;
        float v = x[idx];
        if (MASKED) {
          v = y[idx] > 0.f ? v : 0.f;
          gz[idx] = v;
        }
        return v;
      };
      for (; r + 3 * step < R; r += 4 * step) {
        a0 += at(r);
        a1 += at(r + step);
        a2 += at(r + 2 * step);
        a3 += at(r + 3 * step);
      }
      for (; r < R; r += step) a0 += at(r);
    }
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && cc < C) {
      float s = red[c];
      for (int k = 1; k < nrl; ++k) s += red[k * CW + c];
      partial[(size_t)blockIdx.x * C + cc] = s;
    }
    __syncthreads();
  }
}

// 16 columns x 16 partial lanes per workgroup: lane (c, b) adds partials b, b+16, ... then the 16 lanes of a column
// are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int C,
                                                           float* __restrict__ out) {
  __shared__ float red[256];
  const int c = blockIdx.x * 16 + (threadIdx.x & 15), bl = threadIdx.x >> 4;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int b = bl;
    for (; b + 16 < nblk; b += 32) {
      a0 += partial[(size_t)b * C + c];
      a1 += partial[(size_t)(b + 16) * C + c];
    }
    if (b < nblk) a0 += partial[(size_t)b * C + c];
  }
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (bl == 0 && c < C) {
    float s = red[threadIdx.x];
    for (int k = 1; k < 16; ++k) s += red[k * 16 + threadIdx.x];
    out[c] = s;
  }
}

// One launch finishes a Linear layer's backward: blocks [0, nb1) add the S split-K slabs of the weight gradient
// (+ the optional tail GEMM's contribution), the remaining blocks add the nblk column-sum partials of the bias
// gradient (same fixed order as colsum_final_kernel).
__global__ __launch_bounds__(256) void linear_bwd_finish_kernel(const float* __restrict__ parts, int S, int64_t N,
                                                                const float* __restrict__ tail, float* __restrict__ gw,
                                                                const float* __restrict__ partial, int nblk, int C,
                                                                float* __restrict__ gb, int nb1) {
  __shared__ float red[256];
  if ((int)blockIdx.x < nb1) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N) return;
    float s = tail ? tail[idx] : 0.f;
    for (int p = 0; p < S; ++p) s += parts[(size_t)p * N + idx];
    gw[idx] = s;
    return;
  }
  const int c = ((int)blockIdx.x - nb1) * 16 + (threadIdx.x & 15), bl = threadIdx.x >> 4;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int b = bl;
    for (; b + 16 < nblk; b += 32) {
      a0 += partial[(size_t)b * C + c];
      a1 += partial[(size_t)(b + 16) * C + c];
    }
    if (b < nblk) a0 += partial[(size_t)b * C + c];
  }
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (bl == 0 && c < C) {
    float s = red[threadIdx.x];
    for (int k = 1; k < 16; ++k) s += red[k * 16 + threadIdx.x];
    gb[c] = s;
  }
}

// ---- per-iteration scalar bookkeeping of main.py:313-359 on device-resident state (hipGraph mode) ------------------------
// compute_EMA (utils.py:389-396): value at itr 0, running mean while itr <= floor(1/c), then c v + (1-c) ema.
struct ScalarArgs {
  int phase;
  float *itr, *norm, *ema_gn;
  const float *w_mean, *w_std, *obj, *gn, *gne;
  float c_norm, one_minus_c_norm, warm_norm, c_grad, one_minus_c_grad, warm_grad;
  float *ab, *out;
  // history (nullable): row itr of a (hist_rows, 8) device array also receives out[0..6] and extra[0] (column 7): the values of
  // every iteration stay where they were written, so the host needs no copy of `out` between two replays of a captured iteration
  float* hist;
  int hist_rows;
  const float* extra;
};

__device__ __forceinline__ float ema_update(float v, float ema, float c, float omc, float warm, float itr) {
  if (itr == 0.f) return v;
  if (itr <= warm) return (v + itr * ema) / (itr + 1.f);
  return c * v + omc * ema;
}

// phase 1 (one thread): the iteration's scalar outputs and the state updates of main.py:313-320, 330-345, 354-359
__device__ __forceinline__ void iteration_scalars_phase1(const ScalarArgs& a, float itr, bool has_gn, float gn, float gne) {
  const float norm = a.norm[0];
  float ema_gn = 0.f;
  if (has_gn) { ema_gn = ema_update(gn, a.ema_gn[0], a.c_grad, a.one_minus_c_grad, a.warm_grad, itr); a.ema_gn[0] = ema_gn; }
  a.out[0] = a.obj[0] * (1.f / norm);                      // loss = objective / normaliser   (main.py:313-320)
  a.out[1] = a.w_mean[0];
  a.out[2] = a.w_std[0];
  a.out[3] = has_gn ? gn : 0.f;
  a.out[4] = ema_gn;
  a.out[5] = gne;
  a.out[6] = norm;                                         // the normaliser this iteration used
  if (a.hist && itr < (float)a.hist_rows) {
    float* h = a.hist + 8 * (int)itr;
#pragma unroll
    for (int c = 0; c < 7; ++c) h[c] = a.out[c];
    h[7] = a.extra ? a.extra[0] : 0.f;
  }
  a.norm[0] = ema_update(a.w_mean[0], norm, a.c_norm, a.one_minus_c_norm, a.warm_norm, itr);   // main.py:354-359
  a.itr[0] = itr + 1.f;
}

__global__ void iteration_scalars_kernel(const ScalarArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float itr = a.itr[0];
  if (a.phase == 0) {
    // ema_grad <- A ema_grad + B grad  (the tensor form of the same EMA: main.py:330-345)
    float A, B;
    if (itr == 0.f) { A = 0.f; B = 1.f; }
    else if (itr <= a.warm_grad) { A = itr / (itr + 1.f); B = 1.f / (itr + 1.f); }
    else { A = a.one_minus_c_grad; B = a.c_grad; }
    a.ab[0] = A; a.ab[1] = B;
    return;
  }
  iteration_scalars_phase1(a, itr, a.gn != nullptr, a.gn ? a.gn[0] : 0.f, a.gne ? a.gne[0] : 0.f);
}

// ---- Adam on a flat gradient + gradient telemetry (main.py:174-238, 325-349) ------------------------------------------
struct AdamArgs {
  const socmx_adam_tensor* tensors;
  int ntensors;
  int64_t total;
  const float* grad;
  float* ema;
  const float* itr;
  float c_grad, one_minus_c_grad, warm_grad;
  float lr, beta1, beta2, eps;
  float* scratch;      // [0] = finished-workgroup ticket (as unsigned); [4 + 2 b], [5 + 2 b] = workgroup b's sum g^2, sum ema^2
  float* sums_out;
  int has_post;        // the last workgroup to finish also runs the iteration's scalar bookkeeping (phase 1 of ScalarArgs post)
  ScalarArgs post;
};

constexpr int kAdamPerBlock = 1024;

__global__ __launch_bounds__(256) void adam_step_kernel(const AdamArgs a) {
  __shared__ socmx_adam_tensor tab[64];
  __shared__ float red[2][4];
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x;
  if (tid < a.ntensors) tab[tid] = a.tensors[tid];
  __syncthreads();
  // step count BEFORE this update (every workgroup reads it before the last one to finish advances it)
  const float t = tab[0].step[0] + 1.f;
  const float bc1 = 1.f - powf(a.beta1, t), bc2s = sqrtf(1.f - powf(a.beta2, t));
  const float step_size = a.lr / bc1;
  float A = 0.f, Bc = 1.f;
  if (a.ema) {
    const float itr = a.itr[0];
    if (itr == 0.f) { A = 0.f; Bc = 1.f; }
    else if (itr <= a.warm_grad) { A = itr / (itr + 1.f); Bc = 1.f / (itr + 1.f); }
    else { A = a.one_minus_c_grad; Bc = a.c_grad; }
  }
  float sg = 0.f, se = 0.f;
  const int64_t f0 = (int64_t)blockIdx.x * kAdamPerBlock;
  int ti = 0;
#pragma unroll
  for (int r = 0; r < kAdamPerBlock / 256; ++r) {
    const int64_t f = f0 + r * 256 + tid;
    if (f < a.total) {
      while (ti + 1 < a.ntensors && f >= tab[ti + 1].offset) ++ti;      // offsets ascend; f ascends with r
      const int64_t k = f - tab[ti].offset;
      const float g = a.grad[f];
      sg += g * g;
      if (a.ema) {
        const float em = A * a.ema[f] + Bc * g;
        a.ema[f] = em;
        se += em * em;
      }
      float m = tab[ti].exp_avg[k], v = tab[ti].exp_avg_sq[k];
      m = m + (g - m) * (1.f - a.beta1);
      v = a.beta2 * v + (1.f - a.beta2) * g * g;
      tab[ti].exp_avg[k] = m;
      tab[ti].exp_avg_sq[k] = v;
      const float denom = sqrtf(v) / bc2s + a.eps;
      tab[ti].p[k] -= step_size * m / denom;
    }
  }
  sg = wave_sum(sg); se = wave_sum(se);
  if ((tid & 63) == 0) { red[0][tid >> 6] = sg; red[1][tid >> 6] = se; }
  __syncthreads();
  // per-workgroup partial sums in scratch[4 + 2 b], combined in a FIXED order by the last workgroup to finish (no float
  // atomics: the telemetry is bit-reproducible like the rest of the library)
  if (tid == 0) {
    // (the partial sums travel as device-scope atomic exchanges whose returns are waited for, the last workgroup reads them with
    //  atomic add-zero: no __threadfence() -- a fence here writes back every parameter and moment this workgroup's XCD holds dirty)
    const float o0 = atomicExch(&a.scratch[4 + 2 * blockIdx.x], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    const float o1 = atomicExch(&a.scratch[5 + 2 * blockIdx.x], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(o0), "v"(o1) : "memory");
    const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(a.scratch), 1u);
    last_flag = (ticket == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (last_flag) {                           // every workgroup has finished: publish the sums, re-arm, advance the step counters
    float pg = 0.f, pe = 0.f;
    for (unsigned b = tid; b < gridDim.x; b += 256) {
      pg += atomicAdd(&a.scratch[4 + 2 * b], 0.f);
      pe += atomicAdd(&a.scratch[5 + 2 * b], 0.f);
    }
    pg = wave_sum(pg); pe = wave_sum(pe);
    __syncthreads();
    if ((tid & 63) == 0) { red[0][tid >> 6] = pg; red[1][tid >> 6] = pe; }
    __syncthreads();
    if (tid == 0) {
      const float sg_all = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      const float se_all = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      a.sums_out[0] = sg_all;
      a.sums_out[1] = se_all;
      atomicExch(reinterpret_cast<unsigned*>(a.scratch), 0u);
      // (every workgroup read the iteration counter before it took its ticket: advancing it here is ordered behind them)
      if (a.has_post) iteration_scalars_phase1(a.post, a.post.itr[0], a.ema != nullptr, sg_all, a.ema ? se_all : 0.f);
    }
    if (tid < a.ntensors) tab[tid].step[0] = t;
  }
}

}  // namespace socmx

// =================================================================================================
// C ABI
// =================================================================================================
using namespace socmx;

extern "C" int socmx_adam_step_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                                   float* ema_grad, const float* itr, double c_grad, float lr, float beta1, float beta2,
                                   float eps, float* scratch, float* sums_out, socmx_stream_t stream) {
  if (!tensors || !grad || !scratch || !sums_out || (ema_grad && !itr)) return SOCMX_E_NULL;
  if (ntensors < 1 || ntensors > 64 || total < 1 || (ema_grad && !(c_grad > 0.0))) return SOCMX_E_DIM;
  AdamArgs a;
  a.tensors = tensors; a.ntensors = ntensors; a.total = total; a.grad = grad; a.ema = ema_grad; a.itr = itr;
  a.c_grad = (float)c_grad; a.one_minus_c_grad = (float)(1.0 - c_grad);
  a.warm_grad = ema_grad ? (float)(int)floor(1.0 / c_grad) : 0.f;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.scratch = scratch; a.sums_out = sums_out;
  a.has_post = 0; a.post = ScalarArgs{};
  const unsigned blocks = (unsigned)((total + kAdamPerBlock - 1) / kAdamPerBlock);
  return launch(adam_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

extern "C" int socmx_adam_step_scalars_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                                           float* ema_grad, float* itr, double c_grad, float lr, float beta1, float beta2,
                                           float eps, float* scratch, float* sums_out, float* norm, float* ema_gn,
                                           const float* w_mean, const float* w_std, const float* obj, double c_norm, float* out,
                                           socmx_stream_t stream) {
  return socmx_adam_step_scalars_hist_f32(tensors, ntensors, total, grad, ema_grad, itr, c_grad, lr, beta1, beta2, eps, scratch,
                                          sums_out, norm, ema_gn, w_mean, w_std, obj, c_norm, out, nullptr, 0, nullptr, stream);
}

extern "C" int socmx_adam_step_scalars_hist_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                                                float* ema_grad, float* itr, double c_grad, float lr, float beta1, float beta2,
                                                float eps, float* scratch, float* sums_out, float* norm, float* ema_gn,
                                                const float* w_mean, const float* w_std, const float* obj, double c_norm,
                                                float* out, float* hist, int32_t hist_rows, const float* extra,
                                                socmx_stream_t stream) {
  if (hist && hist_rows < 1) return SOCMX_E_DIM;
  if (!tensors || !grad || !scratch || !sums_out || !itr || !norm || !w_mean || !w_std || !obj || !out || (ema_grad && !ema_gn))
    return SOCMX_E_NULL;
  if (ntensors < 1 || ntensors > 64 || total < 1 || !(c_grad > 0.0) || !(c_norm > 0.0)) return SOCMX_E_DIM;
  AdamArgs a;
  a.tensors = tensors; a.ntensors = ntensors; a.total = total; a.grad = grad; a.ema = ema_grad; a.itr = itr;
  a.c_grad = (float)c_grad; a.one_minus_c_grad = (float)(1.0 - c_grad);
  a.warm_grad = ema_grad ? (float)(int)floor(1.0 / c_grad) : 0.f;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.scratch = scratch; a.sums_out = sums_out;
  a.has_post = 1;
  ScalarArgs& q = a.post;
  q = ScalarArgs{};
  q.phase = 1; q.itr = itr; q.norm = norm; q.ema_gn = ema_gn; q.w_mean = w_mean; q.w_std = w_std; q.obj = obj; q.out = out;
  q.hist = hist; q.hist_rows = hist_rows; q.extra = extra;
  q.c_norm = (float)c_norm; q.one_minus_c_norm = (float)(1.0 - c_norm); q.warm_norm = (float)(int)floor(1.0 / c_norm);
  q.c_grad = (float)c_grad; q.one_minus_c_grad = (float)(1.0 - c_grad); q.warm_grad = (float)(int)floor(1.0 / c_grad);
  const unsigned blocks = (unsigned)((total + kAdamPerBlock - 1) / kAdamPerBlock);
  return launch(adam_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
}

extern "C" int socmx_iteration_scalars_f32(int32_t phase, float* itr, float* norm, float* ema_gn, const float* w_mean,
                                           const float* w_std, const float* obj, const float* gn, const float* gne,
                                           double c_norm, double c_grad, float* ab, float* out, socmx_stream_t stream) {
  return socmx_iteration_scalars_hist_f32(phase, itr, norm, ema_gn, w_mean, w_std, obj, gn, gne, c_norm, c_grad, ab, out, nullptr, 0,
                                          nullptr, stream);
}

extern "C" int socmx_iteration_scalars_hist_f32(int32_t phase, float* itr, float* norm, float* ema_gn, const float* w_mean,
                                                const float* w_std, const float* obj, const float* gn, const float* gne,
                                                double c_norm, double c_grad, float* ab, float* out, float* hist,
                                                int32_t hist_rows, const float* extra, socmx_stream_t stream) {
  if (!itr) return SOCMX_E_NULL;
  if (hist && hist_rows < 1) return SOCMX_E_DIM;
  if (phase == 0 ? !ab : (!norm || !w_mean || !w_std || !obj || !out || (gn && !ema_gn))) return SOCMX_E_NULL;
  if (!(c_norm > 0.0) || !(c_grad > 0.0)) return SOCMX_E_DIM;
  ScalarArgs a;
  a.phase = phase; a.itr = itr; a.norm = norm; a.ema_gn = ema_gn; a.w_mean = w_mean; a.w_std = w_std; a.obj = obj;
  a.gn = gn; a.gne = gne; a.ab = ab; a.out = out;
  a.hist = hist; a.hist_rows = hist_rows; a.extra = extra;
  a.c_norm = (float)c_norm; a.one_minus_c_norm = (float)(1.0 - c_norm); a.warm_norm = (float)(int)floor(1.0 / c_norm);
  a.c_grad = (float)c_grad; a.one_minus_c_grad = (float)(1.0 - c_grad); a.warm_grad = (float)(int)floor(1.0 / c_grad);
  return launch(iteration_scalars_kernel, dim3(1), dim3(64), 0, stream, a);
}

extern "C" int socmx_weights_stats_f32(const float* lpd, const float* lps, const float* ltw, int32_t B, float* w,
                                       float* stats, socmx_stream_t stream) {
  if (!lpd || !lps || !ltw || !w || !stats) return SOCMX_E_NULL;
  if (B < 1) return SOCMX_E_DIM;
  return launch(weights_stats_kernel, dim3(1), dim3(256), 0, stream, lpd, lps, ltw, (int)B, w, stats, StatScalars{});
}

extern "C" int socmx_weights_stats_scalars_f32(const float* lpd, const float* lps, const float* ltw, int32_t B, float* w,
                                               float* stats, const float* gamma, float* gam_out, const float* norm,
                                               float* gout_out, float* obj_zero, socmx_stream_t stream) {
  if (!lpd || !lps || !ltw || !w || !stats || (gam_out && !gamma) || (gout_out && !norm)) return SOCMX_E_NULL;
  if (B < 1) return SOCMX_E_DIM;
  return launch(weights_stats_kernel, dim3(1), dim3(256), 0, stream, lpd, lps, ltw, (int)B, w, stats,
                StatScalars{gamma, gam_out, norm, gout_out, obj_zero});
}

extern "C" int socmx_shard_stats_f32(int32_t phase, const float* w, int32_t B, int32_t rank, int32_t world, const float* obj,
                                     float* tail, float* mean_std, socmx_stream_t stream) {
  if (!tail || (phase == 0 && !w) || (phase != 0 && !mean_std)) return SOCMX_E_NULL;
  if (world < 1 || world > 4096 || rank < 0 || rank >= world || (phase == 0 && B < 1)) return SOCMX_E_DIM;
  return launch(shard_stats_kernel, dim3(1), dim3(256), 0, stream, (int)phase, w, (int)B, (int)rank, (int)world, obj, tail, mean_std);
}

extern "C" int64_t socmx_num_pairs(int32_t K) { return K < 0 ? 0 : (int64_t)(K + 1) * (K + 2) / 2; }

extern "C" int socmx_socm_prep_f32(const socmx_problem* pb, const float* ts, int32_t K, int32_t B, float lmbd,
                                   const float* states, const float* noises, const float* controls,
                                   const float* frac, float* v, float* q, float* gT, float* vT, float* qT,
                                   float* gTT, socmx_stream_t stream) {
  if (!pb || !ts || !states || !noises || !controls || !v || !q || !gT || !pb->sigma_inv_t) return SOCMX_E_NULL;
  if (pb->d < 1 || K < 1 || B < 1) return SOCMX_E_DIM;
  switch (pb->kind) {
    case SOCMX_OU_QUADRATIC: if (!pb->A || !pb->P || !pb->Q) return SOCMX_E_NULL; break;
    case SOCMX_OU_LINEAR: if (!pb->A || !pb->omega) return SOCMX_E_NULL; break;
    case SOCMX_DOUBLE_WELL: if (!pb->kappa || !pb->nu) return SOCMX_E_NULL; break;
    case SOCMX_MOLECULAR_DYNAMICS: if (!pb->kappa) return SOCMX_E_NULL; break;
    default: return SOCMX_E_KIND;
  }
  PrepArgs a;
  a.n_tiled_blocks = 0;
  a.kind = pb->kind; a.d = pb->d; a.K = K; a.B = B; a.sqrt_lmbd = sqrtf(lmbd);
  a.sit = pb->sigma_inv_t; a.A = pb->A; a.P = pb->P; a.Q = pb->Q; a.omega = pb->omega; a.kappa = pb->kappa;
  a.nu = pb->nu;
  a.ts = ts; a.states = states; a.noises = noises; a.controls = controls; a.frac = frac;
  a.v = v; a.q = q; a.gT = gT; a.vT = vT; a.qT = qT; a.gTT = gTT;
  const size_t tile_lds = (size_t)4 * 64 * (pb->d + 1) * sizeof(float);
  if (pb->d % 4 == 0 && pb->d <= 64 && !vT && !qT) {
    const int64_t tiles = ((int64_t)K * B + 15) / 16;
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + 3) / 4, 1024);
    if (const int err = launch(socm_prep_mfma_kernel, dim3(blocks), dim3(256), 0, stream, a)) return err;
    return launch(socm_prep_terminal_kernel, dim3((B * pb->d + 255) / 256), dim3(256), 0, stream, a);
  }
  if (pb->d <= 128 && tile_lds <= 160 * 1024) {
    if (const int err = ensure_max_lds(socm_prep_tiled_kernel)) return err;
    const int64_t rows = (int64_t)K * B;
    a.n_tiled_blocks = (int)((rows + 63) / 64);
    return launch(socm_prep_tiled_kernel, dim3((unsigned)(a.n_tiled_blocks + (B * pb->d + 255) / 256)), dim3(256), tile_lds, stream, a);
  }
  const int64_t n = (int64_t)(K + 1) * B;
  return launch(socm_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
}

static int launch_residual(const TargetArgs& a, socmx_stream_t stream) {
  const int d = a.d, K = a.K, B = a.B;
  const size_t lds = (size_t)2 * 64 * (d + 1) * sizeof(float);
  if (lds > 160 * 1024) return SOCMX_E_LDS;
  if (d % 4 == 0 && d <= 64) {
    const int64_t tiles = ((int64_t)(K + 1) * B + 15) / 16;
    const unsigned blocks = (unsigned)std::min<int64_t>((tiles + 3) / 4, 1024);
    return launch(socm_residual_mfma_kernel, dim3(blocks), dim3(256), 0, stream, a);
  }
  if (const int err = ensure_max_lds(socm_residual_kernel)) return err;
  return launch(socm_residual_kernel, dim3(K + 1, (B + 63) / 64), dim3(64), lds, stream, a);
}

static int launch_target_fwd(const socmx_problem* pb, int32_t K, int32_t B, const float* M_all, const float* dM_all,
                             const float* delta, const float* gamma, const float* q, const float* v,
                             const float* gT, const float* nablaV, const float* w, float inv_norm, float* target,
                             float* G, float* objective, float* workspace, socmx_stream_t stream) {
  if (!pb || !M_all || !dM_all || !q || !v || !gT || !nablaV || !w || !G || !objective || !target || !pb->sigma || !workspace)
    return SOCMX_E_NULL;
  const int d = pb->d;
  if (d < 1 || d > 1024 || K < 1 || B < 1) return SOCMX_E_DIM;
  TargetArgs a;
  a.d = d; a.K = K; a.B = B; a.KG = 0; a.inv_norm = inv_norm;
  a.sigma = pb->sigma; a.M_all = M_all; a.dM_all = dM_all; a.q = q; a.v = v; a.gT = gT;
  a.nablaV = nablaV; a.w = w; a.target = target; a.G = G; a.objective = objective; a.ws = workspace;
  a.delta = delta; a.gamma = gamma;
  a.fuse_residual = 0;
  void* const st0 = stream;
  bool launched = false;
  int lerr = 0;
  if (d > 16 && d <= 64 && B >= 256) {                 // wide form: all k-blocks and 8 x 2 batch tiles per workgroup
    const dim3 wgrid(K + 1, (B + 16 * 2 * kTargetWaves - 1) / (16 * 2 * kTargetWaves));
    const dim3 wblk(64 * kTargetWaves);
    const bool al4 = d % 4 == 0;
#define SOCMX_LDS_LAUNCH(NETV, KBV) \
  lerr = al4 ? launch(socm_target_lds4_kernel<NETV, KBV>, wgrid, dim3(64 * (kTargetWaves + kStageWaves)), 0, st0, a) \
             : launch(socm_target_lds_kernel<NETV, KBV>, wgrid, wblk, 0, st0, a)
    if (delta) {
      if (d <= 32) SOCMX_LDS_LAUNCH(true, 2);
      else         SOCMX_LDS_LAUNCH(true, 4);
    } else {
      if (d <= 32) SOCMX_LDS_LAUNCH(false, 2);
      else         SOCMX_LDS_LAUNCH(false, 4);
    }
#undef SOCMX_LDS_LAUNCH
    launched = true;
  }
  // 16-column batch tiles per wave (developer override SOCMX_TARGET_CT, read once)
  static const int ct_env = [] { const char* e = getenv("SOCMX_TARGET_CT"); return e ? atoi(e) : 0; }();
  // (four column tiles per wave amortise the pair-matrix requests best -- 50.9 / 65.1 / 82.7 us for 4 / 2 / 1 at configs[2] -- unless
  //  that leaves most of the chip without a workgroup: configs[1] has 26 x 2 = 52 of them at four tiles, 17.1 us against 13.4 at two)
  const int ct4 = (B > 32 && ((K + 2) / 2) * ((B + 63) / 64) < 128) ? 2 : 4;
  const int ct = (ct_env == 1 || ct_env == 2 || ct_env == 4) ? ct_env : (B > 32 ? ct4 : (B > 16 ? 2 : 1));
  dim3 grid((K + 2) / 2, (B + 16 * ct - 1) / (16 * ct), (d + 15) / 16);
  const dim3 blk(64 * kTargetWaves);
#define SOCMX_TARGET_LAUNCH(NETV, CTV, N1V) \
  lerr = launch(socm_target_mfma_kernel<NETV, CTV, N1V>, grid, blk, 0, stream, a)
  const int nsv = d <= 4 ? 1 : (d <= 12 ? 3 : (d <= 16 ? 4 : 0));      // MFMAs per l-block (0: several l-blocks)
  // d <= 16: objective and G leave with the target rows (no socm_residual_kernel launch behind the contraction; developer
  // override SOCMX_TARGET_FUSE=0, read once)
  static const int fuse_env = [] { const char* e = getenv("SOCMX_TARGET_FUSE"); return e ? atoi(e) : 1; }();
  a.fuse_residual = (!launched && nsv != 0 && fuse_env) ? 1 : 0;
#define SOCMX_TARGET_NS(NETV, CTV) \
  do { if (nsv == 1) SOCMX_TARGET_LAUNCH(NETV, CTV, 1); else if (nsv == 3) SOCMX_TARGET_LAUNCH(NETV, CTV, 3); \
       else if (nsv == 4) SOCMX_TARGET_LAUNCH(NETV, CTV, 4); else SOCMX_TARGET_LAUNCH(NETV, CTV, 0); } while (0)
  if (launched) {
  } else if (delta) {
    if (ct == 4)      SOCMX_TARGET_NS(true, 4);
    else if (ct == 2) SOCMX_TARGET_NS(true, 2);
    else              SOCMX_TARGET_NS(true, 1);
  } else {
    if (ct == 4)      SOCMX_TARGET_NS(false, 4);
    else if (ct == 2) SOCMX_TARGET_NS(false, 2);
    else              SOCMX_TARGET_NS(false, 1);
  }
#undef SOCMX_TARGET_NS
#undef SOCMX_TARGET_LAUNCH
  if (lerr) return lerr;
  if (a.fuse_residual) return 0;
  return launch_residual(a, stream);
}

// objective = sum w |sigma^T (nabla_V - target)|^2 inv_norm and G = d objective / d nabla_V from a target that is already in
// HBM: the residual kernels of the SOCM loss, for the matching-family baselines (socmx_matching_target_f32) whose targets
// come from other kernels (method.py:289-478, 722-749 end in the same least-squares form as method.py:702-720)
// floats of the objective workspace (ticket + one slot per contributor of whichever kernel the sizes select: the fused contraction
// has (K+1) * ceil(B/16) at most, the residual kernels (K+1) * ceil(B/64) or 4 * 1024)
extern "C" int64_t socmx_socm_objective_workspace_floats(int32_t K, int32_t B) {
  if (K < 1 || B < 1) return 0;
  return 2 + std::max<int64_t>((int64_t)(K + 1) * ((B + 15) / 16), 4096);
}

extern "C" int socmx_socm_residual_f32(const socmx_problem* pb, int32_t K, int32_t B, const float* target,
                                       const float* nablaV, const float* w, float inv_norm, float* G, float* objective,
                                       float* workspace, socmx_stream_t stream) {
  if (!pb || !target || !nablaV || !w || !G || !objective || !pb->sigma || !workspace) return SOCMX_E_NULL;
  if (pb->d < 1 || pb->d > 1024 || K < 1 || B < 1) return SOCMX_E_DIM;
  TargetArgs a{};
  a.d = pb->d; a.K = K; a.B = B; a.inv_norm = inv_norm; a.sigma = pb->sigma; a.nablaV = nablaV; a.w = w;
  a.target = const_cast<float*>(target); a.G = G; a.objective = objective; a.ws = workspace;
  return launch_residual(a, stream);
}

extern "C" int socmx_socm_target_fwd_f32(const socmx_problem* pb, int32_t K, int32_t B, const float* M_all,
                                         const float* dM_all, const float* q, const float* v, const float* gT,
                                         const float* nablaV, const float* w, float inv_norm, float* target,
                                         float* G, float* objective, float* workspace, socmx_stream_t stream) {
  return launch_target_fwd(pb, K, B, M_all, dM_all, nullptr, nullptr, q, v, gT, nablaV, w, inv_norm, target, G,
                           objective, workspace, stream);
}

extern "C" int socmx_socm_target_fwd_net_f32(const socmx_problem* pb, int32_t K, int32_t B, const float* net,
                                             const float* dnet, const float* delta, const float* gamma,
                                             const float* q, const float* v, const float* gT,
                                             const float* nablaV, const float* w, float inv_norm, float* target,
                                             float* G, float* objective, float* workspace, socmx_stream_t stream) {
  if (!delta || !gamma) return SOCMX_E_NULL;
  return launch_target_fwd(pb, K, B, net, dnet, delta, gamma, q, v, gT, nablaV, w, inv_norm, target, G,
                           objective, workspace, stream);
}

static int launch_target_bwd(int32_t d, int32_t K, int32_t B, const float* G, const float* q, const float* v,
                             const float* gT, const float* gout, const float* net, const float* dnet,
                             const float* delta, const float* gamma, float* gM, float* gdM, float* ggamma_part,
                             socmx_stream_t stream) {
  if (!G || !q || !v || !gT || !gM || !gdM) return SOCMX_E_NULL;
  if (d < 1 || K < 1 || B < 1) return SOCMX_E_DIM;
  TargetBwdArgs a;
  a.d = d; a.K = K; a.B = B; a.G = G; a.q = q; a.v = v; a.gT = gT; a.gout = gout; a.gM = gM; a.gdM = gdM;
  a.net = net; a.dnet = dnet; a.delta = delta; a.gamma = gamma; a.ggamma_part = ggamma_part;
  const int64_t np = socmx_num_pairs(K);
  if (d > 16 && d <= 64 && d % 4 == 0) {
    // (reads of whole 16-byte pieces stay inside the rows because d % 4 == 0)
    const int lb = (d + 15) / 16;
    const int64_t r2n = (K + 2) / 2;
    const dim3 lgrid((unsigned)(r2n * (K + 2 - r2n)));                              // items (row pair, j)
#define SOCMX_BWD_LDS(LBV) \
  return net ? launch(socm_target_bwd_lds2_kernel<true, LBV>, lgrid, dim3(256), 0, stream, a) \
             : launch(socm_target_bwd_lds2_kernel<false, LBV>, lgrid, dim3(256), 0, stream, a)
    if (lb == 2) { SOCMX_BWD_LDS(2); }
    if (lb == 3) { SOCMX_BWD_LDS(3); }
    SOCMX_BWD_LDS(4);
#undef SOCMX_BWD_LDS
  }
  if (d > 16) {
    const int nblk = (d + 15) / 16;
    const int lbw = nblk >= 4 ? 4 : 2;                               // l-blocks per wave
    dim3 wgrid((unsigned)np, (nblk + 3) / 4, (nblk + lbw - 1) / lbw);
    if (net) return lbw == 4 ? launch(socm_target_bwd_wide_kernel<true, 4>, wgrid, dim3(256), 0, stream, a)
                             : launch(socm_target_bwd_wide_kernel<true, 2>, wgrid, dim3(256), 0, stream, a);
    return lbw == 4 ? launch(socm_target_bwd_wide_kernel<false, 4>, wgrid, dim3(256), 0, stream, a)
                    : launch(socm_target_bwd_wide_kernel<false, 2>, wgrid, dim3(256), 0, stream, a);
  }
  // two pairs per wave pay off once a wave has enough 16-row chunks to stream (B = 1,024: 0.40 -> 0.36 ms at d = 10, K = 200);
  // at a training batch of 128 the halved wave count costs more than the shared fragments save (58 -> 61 us)
  if (B >= 512) {
    const int64_t r2n = (K + 2) / 2, nitems = r2n * (K + 2 - r2n);                // items (row pair, j): two pairs per wave
    dim3 grid2((unsigned)((nitems + 3) / 4), (d + 15) / 16, (d + 15) / 16);
    return net ? launch(socm_target_bwd_mfma2_kernel<true>, grid2, dim3(256), 0, stream, a, nitems)
               : launch(socm_target_bwd_mfma2_kernel<false>, grid2, dim3(256), 0, stream, a, nitems);
  }
  dim3 grid((unsigned)((np + 3) / 4), (d + 15) / 16, (d + 15) / 16);
  return net ? launch(socm_target_bwd_mfma_kernel<true>, grid, dim3(256), 0, stream, a)
             : launch(socm_target_bwd_mfma_kernel<false>, grid, dim3(256), 0, stream, a);
}

extern "C" int socmx_socm_target_bwd_f32(int32_t d, int32_t K, int32_t B, const float* G, const float* q,
                                         const float* v, const float* gT, float* gM, float* gdM,
                                         socmx_stream_t stream) {
  return launch_target_bwd(d, K, B, G, q, v, gT, nullptr, nullptr, nullptr, nullptr, nullptr, gM, gdM, nullptr,
                           stream);
}

extern "C" int socmx_socm_target_bwd_net_f32(int32_t d, int32_t K, int32_t B, const float* G, const float* q,
                                             const float* v, const float* gT, const float* gout, const float* net,
                                             const float* dnet, const float* delta, const float* gamma,
                                             float* g_net, float* g_dnet, float* g_gamma_part,
                                             socmx_stream_t stream) {
  if (!net || !dnet || !delta || !gamma || !g_gamma_part) return SOCMX_E_NULL;
  return launch_target_bwd(d, K, B, G, q, v, gT, gout, net, dnet, delta, gamma, g_net, g_dnet, g_gamma_part, stream);
}

extern "C" int32_t socmx_colsum_blocks(int64_t R, int32_t C) {
  if (R < 1 || C < 1) return 0;
  const int cw = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : (C > 32 ? 64 : (C > 16 ? 32 : 16))));
  const int nrl = 256 / cw;
  const int64_t want = (R + (int64_t)nrl * 8 - 1) / ((int64_t)nrl * 8);   // >= 8 rows per row lane
  return (int32_t)(want < 1 ? 1 : (want > 512 ? 512 : want));
}

extern "C" int socmx_colsum_f32(const float* x, int64_t R, int32_t C, float* partial, float* out,
                                socmx_stream_t stream) {
  if (!x || !partial) return SOCMX_E_NULL;
  if (R < 1 || C < 1) return SOCMX_E_DIM;
  const int cw = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : (C > 32 ? 64 : (C > 16 ? 32 : 16))));
  const int nblk = socmx_colsum_blocks(R, C);
  if (const int err = launch(colsum_partial_kernel<false>, dim3(nblk), dim3(256), 0, stream, x, nullptr, nullptr, R,
                             (int)C, cw, partial))
    return err;
  if (!out) return 0;                     // partials only: socmx_linear_bwd_finish_f32 adds them up
  return launch(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, partial, nblk, (int)C, out);
}

extern "C" int socmx_relu_bwd_colsum_f32(const float* gy, const float* y, int64_t R, int32_t C, float* gz,
                                         float* partial, float* out, socmx_stream_t stream) {
  if (!gy || !y || !gz || !partial) return SOCMX_E_NULL;
  if (R < 1 || C < 1) return SOCMX_E_DIM;
  const int cw = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : (C > 32 ? 64 : (C > 16 ? 32 : 16))));
  const int nblk = socmx_colsum_blocks(R, C);
  if (const int err = launch(colsum_partial_kernel<true>, dim3(nblk), dim3(256), 0, stream, gy, y, gz, R, (int)C, cw,
                             partial))
    return err;
  if (!out) return 0;
  return launch(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, partial, nblk, (int)C, out);
}

extern "C" int socmx_linear_bwd_finish_f32(const float* gw_parts, int32_t S, int64_t N, const float* tail, float* gw,
                                           const float* partial, int32_t nblk, int32_t C, float* gb,
                                           socmx_stream_t stream) {
  if (!gw_parts || !gw || !partial || !gb) return SOCMX_E_NULL;
  if (S < 1 || N < 1 || nblk < 1 || C < 1) return SOCMX_E_DIM;
  const int nb1 = (int)((N + 255) / 256);
  return launch(linear_bwd_finish_kernel, dim3(nb1 + (C + 15) / 16), dim3(256), 0, stream, gw_parts, (int)S, N, tail,
                gw, partial, (int)nblk, (int)C, gb, nb1);
}
