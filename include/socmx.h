/*
 * socmx.h -- C ABI of the MI355X-native SOC-matching hot path (libsocmx.so).
 *
 * The reference (facebookresearch/SOC-matching) is pure Python/PyTorch and has
 * no FFI of its own; the "operator API" of its hot path is a set of Python call
 * sites.  Each entry point below replaces one of them and says which
 * (file:line are into the reference checkout):
 *
 *   socmx_rollout_f32            SOC_matching/utils.py:17-128  stochastic_trajectories()
 *                                 + SOC_matching/method.py:58-80 NeuralSDE.control() (2-D branch)
 *                                 + SOC_matching/models.py:233-242 FullyConnectedUNet.forward()
 *                                 + experiment_settings/{OU_quadratic,OU_linear,double_well,
 *                                   molecular_dynamics}.py  b(), f(), g(), Phi()
 *   socmx_unet_pack_f32          (no counterpart: re-lays nn.Linear weights, models.py:212-228,
 *                                 into MFMA fragment order; needed whenever the weights changed -- the host package
 *                                 re-packs before every rollout, ~3 us)
 *   socmx_unet_forward_f32       SOC_matching/method.py:272-278 (nabla_V on the trajectory rows)
 *   socmx_unet_backward_f32      autograd of the same lines: d objective / d (weights, biases) of models.py:212-242
 *   socmx_weights_stats_f32      SOC_matching/method.py:258-262, 903-904 (w, mean(w), std(w))
 *   socmx_socm_prep_f32          SOC_matching/method.py:591-646 operand preparation
 *                                 (nabla_f, nabla_b . v, nabla_g, sigma^-T noise / control)
 *   socmx_socm_target_fwd_f32    SOC_matching/method.py:591-720 (least-squares target, residual,
 *                                 weighted reduction to the scalar objective)
 *   socmx_socm_target_bwd_f32    autograd of the above w.r.t. M, dM/ds (nabla_V grad comes out of fwd)
 *   socmx_socm_target_{fwd,bwd}_net_f32   the same with SOC_matching/models.py:263-275 (SigmoidMLP.forward: the
 *                                 exp(-gamma (s-t)) blend of I and the network output) and its d/ds fused in
 *   socmx_colsum_f32, socmx_relu_bwd_colsum_f32   autograd of the nn.Linear(+ReLU) layers of models.py:212-228,
 *                                 253-257 on the trajectory / pair rows: bias gradient, ReLU backward
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = invalid argument (SOCMX_E_*), > 0 = hipError_t of THIS call's launch
 *     (hipLaunchKernel's own status: an error left pending by unrelated work is not picked up);
 *   - the device that owns the buffers must be the current HIP device of the calling thread (kernels launch on the
 *     current device; the Python binding wraps every call in a device guard);
 *   - never throws, never allocates, never synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*); safe to capture in a hipGraph;
 *   - every pointer marked "device" is caller-owned device memory (e.g. torch tensor storage),
 *     fp32, contiguous, row-major in the shape given; structs themselves live on the host;
 *   - re-entrant: the only mutable state is a mutex-protected per-process cache of which kernels already had their
 *     dynamic-LDS limit raised (hipFuncSetAttribute once per kernel and device, not per call); a few getenv()
 *     switches for A/B runs are read once per process: SOCMX_GENERIC, SOCMX_NOFAST, SOCMX_WAVES, SOCMX_PROF_WAVE,
 *     SOCMX_TARGET_WIDE_REGS, SOCMX_TARGET_BWD_REGS, SOCMX_TARGET_CT, SOCMX_TARGET_FUSE.
 *
 * Reference-side binding: see INTEGRATION.md (ctypes stub).
 */
#ifndef SOCMX_H
#define SOCMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOCMX_VERSION 149 /* 0.2.1: the one-row rollout can save the control network's activations and ReLU signs for the backward -- socmx_rollout_extra grew (act_workspace, act_records), + socmx_rollout_saves_activations, socmx_unet_backward_saved_f32; 0.2.0: + socmx_iteration_scalars_hist_f32, socmx_adam_step_scalars_hist_f32 (the iteration's scalars also land in row `itr` of a history array); 0.1.9: the objective is summed without float atomics -- socmx_socm_target_fwd_f32 / _fwd_net_f32 / socmx_socm_residual_f32 take a caller-owned workspace (socmx_socm_objective_workspace_floats); socmx_shard_stats_f32 carries per-rank (n, mean, M2) slots instead of shifted sums; 0.1.8: + socmx_weights_stats_scalars_f32; 0.1.7: the packed U-Net image carries the folded skip behind the nine layers -- F = up_0 res_1, f = up_0 b_res_1, cat = [F | up_0] (socmx_unet_packed_floats grew); the transposed image holds F^T in res_1^T's place (socmx_unet_packed_bwd_floats shrank); socmx_unet_backward_sizes: + the fold's scratch */

#define SOCMX_E_NULL (-1)      /* required pointer is NULL            */
#define SOCMX_E_DIM (-2)       /* dimension out of the supported range */
#define SOCMX_E_KIND (-3)      /* unknown problem kind                 */
#define SOCMX_E_WORKSPACE (-4) /* workspace too small                  */
#define SOCMX_E_LDS (-5)       /* configuration does not fit in 160 KiB of LDS (rollout, hidden widths 256/128/64:
                                  d <= 64 OU_quadratic, 74 OU_linear, 96 double_well, 80 molecular_dynamics) */

typedef void* socmx_stream_t; /* hipStream_t */

/* method.setting -> kind (experiment_settings/settings.py:143-207) */
enum {
  SOCMX_OU_QUADRATIC = 0,       /* OU_quadratic_easy / OU_quadratic_hard : b=Ax, f=x'Px, g=x'Qx */
  SOCMX_OU_LINEAR = 1,          /* OU_linear                             : b=Ax, f=0,    g=omega.x */
  SOCMX_DOUBLE_WELL = 2,        /* double_well   : b=-4k x(x^2-1), f=0, g=sum nu (x^2-1)^2 */
  SOCMX_MOLECULAR_DYNAMICS = 3  /* molecular_dynamics : DW drift, f=1, g=0, Phi(x)=-x_0 (stopping) */
};

/* Closed-form problem constants; unused members may be NULL. All device, fp32. */
#define SOCMX_SIGMA_IDENTITY 1 /* flags: sigma is exactly the identity (lets the rollout skip the sigma products) */

typedef struct socmx_problem {
  int32_t kind;
  int32_t d;
  int32_t flags;            /* SOCMX_SIGMA_IDENTITY or 0; a hint: 0 is always correct */
  int32_t reserved;
  const float* sigma;       /* (d,d) */
  const float* sigma_inv_t; /* (d,d) transpose(inverse(sigma)); only the loss entry points read it */
  const float* A;           /* (d,d)  OU_*            */
  const float* P;           /* (d,d)  OU_QUADRATIC    */
  const float* Q;           /* (d,d)  OU_QUADRATIC    */
  const float* omega;       /* (d,)   OU_LINEAR       */
  const float* kappa;       /* (d,)   DOUBLE_WELL, MD */
  const float* nu;          /* (d,)   DOUBLE_WELL     */
} socmx_problem;

/* FullyConnectedUNet parameters in torch layout: weight (out,in) row-major, bias (out,).
 * Index order is the module construction order of models.py:212-228. */
enum {
  SOCMX_L_DOWN0 = 0, SOCMX_L_DOWN1 = 1, SOCMX_L_DOWN2 = 2,
  SOCMX_L_RES0 = 3, SOCMX_L_RES1 = 4, SOCMX_L_RES2 = 5,
  SOCMX_L_UP2 = 6, SOCMX_L_UP1 = 7, SOCMX_L_UP0 = 8
};
typedef struct socmx_unet {
  int32_t d;        /* state dimension: input is d+1 = [t, x], output is d */
  int32_t hdims[3]; /* arch.hdims */
  const float* weight[9];
  const float* bias[9];
} socmx_unet;

int socmx_version(void);

/* Writes up to `cap` bytes of a NUL-terminated description ("gfx950 wave64 mfma_f32_16x16x4 ...")
 * and returns the number of bytes needed. */
int socmx_capabilities(char* buf, int cap);

/* ---- control network --------------------------------------------------- */

/* Number of floats of the MFMA-fragment-ordered image of a U-Net: the nine layers of models.py:212-228, then THE FOLD --
 * F = W_up_0 W_res_1 (out_pad x h0) in the same fragment order, f = W_up_0 b_res_1, and cat = [F | W_up_0] (out_pad x 2 h0).
 * models.py:239-240 reads out = relu(up_0 (relu(up_1 o2 + b) + res_1 r1 + b) + b) + ...: the skip res_1 r1 + b reaches the ReLU
 * only through the linear up_0, so every kernel uses F r1 + f in its place (a d x 256 product instead of a 256 x 256 one):
 * the one-row rollout multiplies r1 by F, the MFMA tile kernels multiply the LDS row [r1 | relu(up_1 o2 + b)] by cat in ONE
 * split-K GEMM, the backward kernels use F and F^T (socmx_unet_pack_bwd_f32) and recover dW_res_1, db_res_1 and the skip's share
 * of dW_up_0 from the d x h0 product ZU0^T R1.  The pack kernels form F with fp64 accumulation. */
size_t socmx_unet_packed_floats(int32_t d, const int32_t hdims[3]);

/* packed (device, socmx_unet_packed_floats floats) <- net. Re-run after every optimizer step. */
int socmx_unet_pack_f32(const socmx_unet* net, float* packed, socmx_stream_t stream);

/* out (N,d) = UNet(tx (N,d+1)).  Same kernel body as the rollout's per-step evaluation. */
int socmx_unet_forward_f32(const float* packed, int32_t d, const int32_t hdims[3],
                           const float* tx, int64_t N, float* out, socmx_stream_t stream);

/*
 * Parameter gradients of the U-Net over many rows -- autograd of models.py:233-242 as applied in method.py:272-278
 * (nabla_V on the (K+1)*B trajectory rows, whose inputs are detached: utils.py:103-115).  Given
 *     gout (N,d) = d objective / d nabla_V(row)
 * it writes d objective / d (weight, bias) of the nine layers into `grads`, a flat buffer in SOCMX_L_* order
 * [weight_0 (out,in) | bias_0 | weight_1 | bias_1 | ...] (torch layouts, socmx_unet_backward_sizes tells its length).
 * Row r is the network input [ts[r / rows_per_t], x[r,:]]  (x (N,d); for the trajectory tensor `states` (K+1,B,d)
 * pass ts (K+1,) and rows_per_t = B; rows_per_t = 1 gives every row its own time).
 * packed / packedT: the fragment-ordered images of the CURRENT weights and of their transposes
 * (socmx_unet_pack_f32 / socmx_unet_pack_bwd_f32).  workspace: socmx_unet_backward_sizes floats, contents undefined
 * afterwards.  The forward activations are recomputed tile by tile in LDS (no (N, width) tensors are saved between the
 * forward and the backward); the summation order over rows is fixed (deterministic, no atomics).
 * SOCMX_E_LDS: this (d, hdims) does not fit kernel A's LDS tiles -- callers fall back to library autograd.
 */
size_t socmx_unet_packed_bwd_floats(int32_t d, const int32_t hdims[3]);
int socmx_unet_pack_bwd_f32(const socmx_unet* net, float* packedT, socmx_stream_t stream);
int socmx_unet_backward_sizes(int32_t d, const int32_t hdims[3], int64_t N, int64_t* workspace_floats,
                              int64_t* grad_floats);
int socmx_unet_backward_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                            const float* x, const float* ts, int32_t rows_per_t, int64_t N, const float* gout,
                            float* workspace, float* grads, socmx_stream_t stream);
/* The same with gout multiplied by the DEVICE scalar gout_scale[0] as the gradient tiles are read (NULL: 1) -- the
 * d loss / d objective = 1 / running normaliser of main.py:313-320, which lives on the device: one elementwise launch less on the
 * iteration's critical path; the products are the fp32 products that launch would have written. */
int socmx_unet_backward_scaled_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                   const float* x, const float* ts, int32_t rows_per_t, int64_t N, const float* gout,
                                   const float* gout_scale, float* workspace, float* grads, socmx_stream_t stream);
/* The same from SAVED activations: `workspace` already holds the activation slabs and `records` the ReLU signs of the N rows, written by
 * the rollout that produced x (socmx_rollout_ex_f32: act_workspace / act_records, same weights): kernel A runs its five backward
 * stages only -- no forward re-computation (half of its multiply-adds), no activation exports.  N must be a multiple of 16 and
 * (d, hdims) the library's constexpr-specialised default widths with d <= 15 (SOCMX_E_DIM otherwise: callers use the plain entry). */
int socmx_unet_backward_saved_f32(const float* packed, const float* packedT, int32_t d, const int32_t hdims[3],
                                  const float* x, const float* ts, int32_t rows_per_t, int64_t N, const float* gout,
                                  const float* gout_scale, const uint32_t* records, float* workspace, float* grads,
                                  socmx_stream_t stream);

/*
 * The pair-grid network of the SOCM loss: SigmoidMLP.sigmoid_layers (models.py:245-257: Linear(2,h0) ReLU Linear(h0,h1) ReLU
 * Linear(h1,d*d)) evaluated on the Np pairs (t_p, s_p) TOGETHER with its derivative in s, which the reference obtains by
 * functorch.jacrev (method.py:510-515):
 *     net[p]  = L2 relu(L1 relu(L0 [t,s] + b0) + b1) + b2            (Np, d*d)
 *     dnet[p] = L2 ( m2 (.) L1 ( m1 (.) L0[:,1] ) )                   (Np, d*d)   m = the ReLU signs of the value path
 * (the exp(-gamma (s-t)) blend of models.py:263-275 is applied inside socmx_socm_target_*_net_f32).  Forward tangent and
 * value share every weight fragment (two MFMA column tiles per fragment).  The backward takes g_net, g_dnet
 * = d objective / d (net, dnet) (what socmx_socm_target_bwd_net_f32 writes) and returns the parameter gradients as one flat
 * buffer [W0 (h0,n_in) | b0 | W1 (h1,h0) | b1 | W2 (d*d,h1) | b2] (torch layouts); the forward is recomputed tile by tile.
 * packed: socmx_mnet_packed_floats floats, rebuilt by socmx_mnet_pack_f32 whenever the weights changed.
 * n_in = 2: SigmoidMLP; n_in = 3: TwoBoundarySigmoidMLP.sigmoid_layers (models.py:278-310: Linear(3,h0) ...), whose third
 * input z[p] is the stopped / running flag of models.py:313-336 -- both evaluations of the stopping-time loss are ONE call
 * on 2 Np rows (z = 0 on the first Np, 1 on the rest).  z may be NULL when n_in = 2.
 * d*d outputs beyond an LDS tile (d > 26 with 128-wide hidden layers; BASELINE configs[4]: d = 64) take the WIDE kernels:
 * the last layer's outputs leave the forward kernel from the accumulators, the backward reads g_net / g_dnet from HBM as
 * MFMA operands (split-K over the waves) and forms the last layer's weight gradient in its own kernel; any d (rows of d*d
 * floats that are not whole 16-byte pieces are read shifted back into the row and stored element by element); this needs
 * h1 <= 256 (beyond 128 units the backward kernel passes over the gradient rows twice), otherwise SOCMX_E_LDS (callers keep
 * library autograd).
 */
size_t socmx_mnet_packed_floats(int32_t d, const int32_t hdims_M[2]);
int socmx_mnet_pack_f32(int32_t d, const int32_t hdims_M[2], int32_t n_in, const float* w0, const float* b0,
                        const float* w1, const float* b1, const float* w2, const float* b2, float* packed,
                        socmx_stream_t stream);
int socmx_mnet_forward_f32(const float* packed, int32_t d, const int32_t hdims_M[2], const float* t, const float* s,
                           const float* z, int64_t Np, float* net, float* dnet, socmx_stream_t stream);
int socmx_mnet_backward_sizes(int32_t d, const int32_t hdims_M[2], int32_t n_in, int64_t Np, int64_t* workspace_floats,
                              int64_t* grad_floats);
int socmx_mnet_backward_f32(const float* packed, int32_t d, const int32_t hdims_M[2], int32_t n_in, const float* t,
                            const float* s, const float* z, int64_t Np, const float* g_net, const float* g_dnet,
                            float* workspace, float* grads, socmx_stream_t stream);

/* ---- rollout ------------------------------------------------------------ */

/*
 * K Euler-Maruyama steps of B trajectories (utils.py:17-128), one launch.
 *   x0        (B,d)  device   initial states (the reference passes x0.repeat(B,1))
 *   ts        (K+1,) device   time grid; dt_k = ts[k+1]-ts[k] in fp32 (utils.py:38)
 *   noise_in  (K,B,d) device or NULL.  NULL => Philox4x32-10 keyed by (seed, offset,
 *             row0 + row, step): counter = (row0+row, step, dim/4, offset), key = seed;
 *             Box-Muller in fp32.  Independent of how rows are split over launches / GPUs.
 *   outputs (all device, step-major exactly as torch.stack in utils.py:103-128):
 *     states (K+1,B,d)  noises (K,B,d)  controls (K,B,d)
 *     stop_indicators (K+1,B)  fractional_timesteps (K,B)  lpd, lps, ltw (B,)
 *   The stopping-time branch (utils.py:42-44, 49-75) is taken iff kind == MOLECULAR_DYNAMICS,
 *   mirroring `hasattr(sde, "Phi")` (utils.py:33).
 *   Costs-only launch: states, noises, controls, stop_indicators and fractional_timesteps may ALL be NULL; then only
 *   lpd / lps / ltw are written (what the evaluation bursts utils.py:131-231 and method.py:185-221 consume; at
 *   d = 64, K = 400 the trajectory of 65,536 rows would be 20 GB).  Any other mix of NULLs is SOCMX_E_NULL.
 *   Tile shape (internal; a row's result does not depend on it beyond fp32 summation order in the network), chosen by
 *   rollout_launch (csrc/socmx_rollout.hip) from B, d, sigma and the library's constexpr-specialised hidden widths:
 *     ONE row per workgroup   B <= 256 at d <= 15 (any sigma without a stopping time) and at 17 <= d <= 31 with sigma = I:
 *                             matrix-vector stages on v_fmac_f32_dpp, weights resident in registers / LDS
 *                             (csrc/socmx_rollout1.hip) -- the default for training batches: BASELINE configs[1], [2]
 *     4 rows                  up to 64 tiles of 16 rows otherwise (B <= 1024; B <= 256 at d >= 32), d <= 64:
 *                             v_mfma_f32_4x4x1_16b_f32 on the same packed image, B / 4 workgroups
 *     16 rows                 everything else (v_mfma_f32_16x16x4_f32); more than 256 tiles at d <= 31: two tiles per
 *                             workgroup (csrc/socmx_rollout32.hip: the evaluation bursts)
 */
int socmx_rollout_f32(const socmx_problem* problem, const float* packed_unet, const int32_t hdims[3],
                      const float* x0, const float* ts, int32_t B, int32_t K, float lmbd,
                      uint64_t seed, uint64_t offset, int64_t row0, const float* noise_in,
                      float* states, float* noises, float* controls, float* stop_indicators,
                      float* fractional_timesteps, float* lpd, float* lps, float* ltw,
                      socmx_stream_t stream);

/*
 * The same rollout with optional extras (any member may be NULL; extra == NULL is socmx_rollout_f32):
 *   key      device uint64[2] = {seed, offset}: the Philox key is read from DEVICE memory when the kernel starts and the
 *            by-value seed / offset are ignored.  A launch captured in a hipGraph therefore draws fresh noise on every
 *            replay once socmx_philox_advance (key[1] += inc, a one-thread node on the same stream) follows it -- by-value
 *            arguments are frozen into a captured node.  (The reference advances torch's global generator: utils.py:39.)
 *   nabla_v  device (K+1,B,d): nabla_V(t_k, X_k) for k = 0..K -- the network outputs the integrator computes anyway, plus
 *            one evaluation at (T, X_K).  These are the forward values of method.py:272-278 (nabla_V on all (K+1) B
 *            trajectory rows), so the loss needs no second forward pass; needs the trajectory buffers (not costs-only).
 */
/*   flags    SOCMX_ROLLOUT_SHARES_CHIP: the caller runs chip-filling kernels beside this launch on another stream (a training
 *            iteration's loss side: pair-grid network, deferred contraction backward).  Where two tile shapes apply
 *            (d >= 32, 256 < rows <= 1024: one GPU's slice of BASELINE configs[4]) the launcher then keeps the rollout on few
 *            CUs (16-row tiles, 7.3 ms at the slice, hidden beside the loss); without the flag -- a stand-alone rollout:
 *            evaluation, a user calling stochastic_trajectories -- it spreads over 4-row tiles (4.9 ms).
 */
#define SOCMX_ROLLOUT_SHARES_CHIP 1u
/*            SOCMX_ROLLOUT_ADVANCES_KEY: `key` is device uint64[3] = {seed, offset, ticket} and the LAUNCH ITSELF performs
 *            key[1] += 1 -- every workgroup takes a ticket once it has read (seed, offset), and the one that draws the last ticket
 *            stores offset + 1 and clears the ticket -- instead of a socmx_philox_advance node behind it (one launch and its
 *            dependency gap less on a captured iteration's critical path).  ticket must be 0 before the first such launch; the
 *            launch leaves it 0.  Not to be combined with another launch reading the same key concurrently.
 */
#define SOCMX_ROLLOUT_ADVANCES_KEY 2u
/*   act_workspace / act_records (both or neither; need nabla_v): the launch SAVES what the control-network backward would otherwise
 *            re-compute.  act_workspace is that backward's workspace (socmx_unet_backward_sizes floats for N = (K+1) B rows): the
 *            activation slabs of every trajectory row -- R1, R2, R3, O2 and A1 = relu(up_1 O2 + b) of models.py:233-242, [16-row tile][unit][16]
 *            -- are written where socmx_unet_backward_saved_f32 and the weight-gradient kernel read them; act_records is device
 *            uint32[(K+1) B][32]: the ReLU signs of a row (layout: csrc/socmx_unet.h).  Honoured by the one-row kernel only (d <= 15,
 *            B <= 256, default widths, (K+1) B a multiple of 16; with or without a stopping time): socmx_rollout_saves_activations answers 1 / 0 for a
 *            launch's arguments without launching, and a launch that cannot honour the request returns SOCMX_E_DIM.  The states, noises,
 *            costs and nabla_v of the launch are bit-identical with and without the request.
 */
typedef struct socmx_rollout_extra {
  const uint64_t* key;
  float* nabla_v;
  uint32_t flags;
  uint32_t reserved;
  float* act_workspace;
  uint32_t* act_records;
} socmx_rollout_extra;
int socmx_rollout_saves_activations(const socmx_problem* problem, const int32_t hdims[3], int32_t B, int32_t K);
int socmx_rollout_ex_f32(const socmx_problem* problem, const float* packed_unet, const int32_t hdims[3],
                         const float* x0, const float* ts, int32_t B, int32_t K, float lmbd,
                         uint64_t seed, uint64_t offset, int64_t row0, const float* noise_in,
                         float* states, float* noises, float* controls, float* stop_indicators,
                         float* fractional_timesteps, float* lpd, float* lps, float* ltw,
                         const socmx_rollout_extra* extra, socmx_stream_t stream);
int socmx_philox_advance(uint64_t* key, uint64_t inc, socmx_stream_t stream);

/*
 * The rollout of utils.py:17-128 under a TABULATED control instead of the network: the ground-truth controls of
 * models.py:10-150 that method.py:103-107 evaluates when `use_learned_control` is off (the optimal-SDE bursts of
 * main.py:137-150).  Same outputs, layouts, Philox contract and costs-only rule as socmx_rollout_f32.
 *   LINEAR    u = table[tidx[k]] (d,d) . x      models.py:10-41  LinearControl (LQ Riccati, utils.py:234-254)
 *   CONSTANT  u = table[tidx[k]] (d,)           models.py:61-83  ConstantControlLinear
 *   TABLE     u_j = table[tidx[k], clamp(floor((x_j + xb) / delta_x), 0, n_x - 1), j]   models.py:98-150 LowDimControl
 * tidx (K,) int32 device: the table row of step k, evaluated by the caller with the reference's fp32 index formula
 * for a scalar time (floor((n-1) t / T), floor(n t / T), ceil(t / delta_t) respectively).
 * molecular_dynamics has no ground truth (settings.py:112-114): SOCMX_E_KIND.
 */
enum { SOCMX_CTRL_LINEAR = 1, SOCMX_CTRL_CONSTANT = 2, SOCMX_CTRL_TABLE = 3 };
typedef struct socmx_control {
  int32_t kind;
  int32_t n_t;          /* rows of the table along time */
  int32_t n_x;          /* TABLE: entries along x */
  int32_t reserved;
  const float* table;   /* LINEAR (n_t,d,d); CONSTANT (n_t,d); TABLE (n_t,n_x,d) */
  const int32_t* tidx;  /* (K,) */
  float xb, delta_x;    /* TABLE only */
} socmx_control;
int socmx_rollout_control_f32(const socmx_problem* problem, const socmx_control* control, const float* x0,
                              const float* ts, int32_t B, int32_t K, float lmbd, uint64_t seed, uint64_t offset,
                              int64_t row0, const float* noise_in, float* states, float* noises, float* controls,
                              float* stop_indicators, float* fractional_timesteps, float* lpd, float* lps,
                              float* ltw, socmx_stream_t stream);

/* Diagnostics: the same rollout instrumented with s_memtime; cycles ((B+15)/16, 64) int64 device receives, per
 * workgroup, shader cycles summed over the K steps for: [0] input tile build, [1..6] the six network stages
 * (down_0, down_1, down_2, up_2+res_2, up_1+res_1, up_0+res_0), [7] control+noise, [8] EM update,
 * [9] costs/state write-back, [16+8*s+{0..4}] wave 0's split of stage s: issue/prologue, GEMM 1, GEMM 2,
 * store+prefetch, closing barrier.  Results are identical to socmx_rollout_f32; it is slower. */
int socmx_rollout_phase_cycles_f32(const socmx_problem* problem, const float* packed_unet, const int32_t hdims[3],
                                   const float* x0, const float* ts, int32_t B, int32_t K, float lmbd,
                                   uint64_t seed, uint64_t offset, int64_t row0, const float* noise_in,
                                   float* states, float* noises, float* controls, float* stop_indicators,
                                   float* fractional_timesteps, float* lpd, float* lps, float* ltw,
                                   int64_t* cycles, socmx_stream_t stream);

/* ---- importance weights --------------------------------------------------- */

/* w[m] = exp(lpd+lps+ltw) (method.py:258-262); stats (5,) = (sum w, sum (w - mean)^2, B, mean, unbiased std):
 * stats[3..4] are torch.mean / torch.std of method.py:903-904 for this shard; across shards stats[0..2] combine
 * with Chan's parallel-variance rule on the host. */
int socmx_weights_stats_f32(const float* lpd, const float* lps, const float* ltw, int32_t B,
                            float* w, float* stats, socmx_stream_t stream);
/* The same launch also forms the three device scalars the loss side of an iteration starts from (each pointer may be NULL):
 *   gam_out[0]  = gamma[0]        the exponent of models.py:265-275 as THIS iteration's contraction reads it (the deferred backward
 *                                 of the same iteration reads this copy after gamma itself may have been stepped),
 *   gout_out[0] = 1 / norm[0]     d loss / d objective of main.py:313-320 (the running normaliser lives in device memory),
 *   obj_zero[0] = 0               the objective's accumulator (socmx_socm_target_fwd_net_f32 adds into it).
 * Three one-element launches on the iteration's critical path otherwise. */
int socmx_weights_stats_scalars_f32(const float* lpd, const float* lps, const float* ltw, int32_t B, float* w, float* stats,
                                    const float* gamma, float* gam_out, const float* norm, float* gout_out, float* obj_zero,
                                    socmx_stream_t stream);

/* The same statistics for a batch SHARD (one rank of a data-parallel run; no reference counterpart -- the reference is
 * single-process), in a form that ONE all_reduce(SUM) combines EXACTLY, so they travel in the flat gradient buffer:
 *   phase 0: tail (1 + 3 world floats): tail[0] = obj[0] (0 if obj is NULL); tail[1 + 3 rank ..] = (B, mean, sum (w - mean)^2) of
 *            this rank's rows (two passes), every other rank's slot = 0 -- summing zeros is exact, so behind the all-reduce
 *            every rank holds every rank's triple bit for bit;
 *   phase 1: mean_std[0..1] = Chan's pooling of the world triples in rank order, in fp64 = torch.mean / torch.std of
 *            method.py:903-904 over the GLOBAL batch, at the accuracy of the one-process statistics whatever the weights' scale. */
int socmx_shard_stats_f32(int32_t phase, const float* w, int32_t B, int32_t rank, int32_t world, const float* obj, float* tail,
                          float* mean_std, socmx_stream_t stream);

/* Number of (t_i <= s_j) pairs: (K+1)(K+2)/2, ordered i-major, j ascending (method.py:533-547).
 * M_all / dM_all below are (Np,d,d) in that order. */
int64_t socmx_num_pairs(int32_t K);

/* ---- SOCM least-squares target + weighted residual -------------------------------- */

/*
 * Operands of the restated contraction, one thread per (j,m):
 *   v[j,m,:] = -( sqrt(lmbd) sqrt(dt_j) S^-T noise[j,m] + dt_j S^-T control[j,m] )
 *   q[j,m,:] = dt_j nabla_f(X[j,m]) + nabla_b(X[j,m])^T v[j,m]      ((nabla_b^T v)_l = sum_n d b_n/d x_l v_n)
 *   gT[m,:]  = nabla_g(X[K,m])
 * batch-major  v,q (K,B,d), gT (B,d): read by socmx_socm_target_fwd_f32 and socmx_socm_target_bwd_f32.
 * vT,qT (K,d,B), gTT (d,B) are optional batch-fastest copies (NULL = not written; no kernel of this library reads them).
 * frac (K,B) may be NULL (=> dt_j = ts[j+1]-ts[j]), else per-sample fractional time steps.
 */
int socmx_socm_prep_f32(const socmx_problem* problem, const float* ts, int32_t K, int32_t B, float lmbd,
                        const float* states, const float* noises, const float* controls,
                        const float* frac, float* v, float* q, float* gT, float* vT, float* qT, float* gTT,
                        socmx_stream_t stream);

/*
 * target[i,m,:] = sum_{j=i}^{K-1} ( M_ij q[j,m] - dM_ij v[j,m] ) + M_iK gT[m]
 * r[i,m,:]      = sigma^T ( nablaV[i,m] - target[i,m] )
 * objective    += inv_norm * sum_{i,m} w[m] |r[i,m]|^2             (inv_norm = 1/((K+1) B_global))
 * G[i,m,:]      = d objective / d nablaV[i,m,:] = 2 w[m] inv_norm sigma r[i,m]   ( = - d objective/d target )
 * objective (1,) is ADDED TO (zero it first) -- once per launch, by the last contributor to finish, with the launch's total:
 * every workgroup leaves its partial sum in a slot of `workspace` and the slots are added in a fixed order, so the value is
 * bit-reproducible run to run like the reference's torch.sum (method.py:717-720); no float atomics.
 * workspace: socmx_socm_objective_workspace_floats(K, B) floats of device memory, ZERO before its first use; every launch leaves
 * its ticket word zero again, so one buffer serves all launches of a stream (never two launches that may run concurrently).
 * target, nablaV, G: (K+1,B,d); target is an output (written by the MFMA contraction, read by the residual pass).
 */
int64_t socmx_socm_objective_workspace_floats(int32_t K, int32_t B);
int socmx_socm_target_fwd_f32(const socmx_problem* problem, int32_t K, int32_t B, const float* M_all,
                              const float* dM_all, const float* q, const float* v, const float* gT,
                              const float* nablaV, const float* w, float inv_norm, float* target,
                              float* G, float* objective, float* workspace, socmx_stream_t stream);

/* gM[p] = -sum_m G[i,m] (x) qx[j,m],  gdM[p] = +sum_m G[i,m] (x) v[j,m]   (p = pair (i,j); qx = q for j<K,
 * gT for j=K; gdM at j=K is 0).  Overwrites gM, gdM (Np,d,d). */
int socmx_socm_target_bwd_f32(int32_t d, int32_t K, int32_t B, const float* G, const float* q,
                              const float* v, const float* gT, float* gM, float* gdM,
                              socmx_stream_t stream);

/*
 * Fused pair-matrix variants for the reference's SigmoidMLP parametrisation (models.py:263-275):
 *     e = exp(-gamma (s_j - t_i)),   M = e I + (1-e) net,   dM/ds = gamma e (net - I) + (1-e) dnet
 * net, dnet (Np,d,d) are the raw outputs of `sigmoid_layers` and of its forward tangent in s; delta (Np,) = s - t;
 * gamma (1,) is read on the device (it is a trained parameter: method.py:134, 160).  M and dM/ds are never written
 * to HBM: the forward forms them in registers while loading its MFMA operands, the backward chains through them in
 * its epilogue.  Same outputs as socmx_socm_target_fwd_f32 applied to the materialised (M, dM).
 */
int socmx_socm_target_fwd_net_f32(const socmx_problem* problem, int32_t K, int32_t B, const float* net,
                                  const float* dnet, const float* delta, const float* gamma, const float* q,
                                  const float* v, const float* gT, const float* nablaV, const float* w,
                                  float inv_norm, float* target, float* G, float* objective, float* workspace,
                                  socmx_stream_t stream);

/* g_net = d obj/d net, g_dnet = d obj/d dnet (Np,d,d), scaled by gout[0] (upstream gradient of the objective on
 * the device; NULL = 1).  g_gamma_part (Np * ceil(d/16)^2,) holds partial sums of d obj/d gamma: the caller adds
 * them up (deterministic, no atomics). */
int socmx_socm_target_bwd_net_f32(int32_t d, int32_t K, int32_t B, const float* G, const float* q,
                                  const float* v, const float* gT, const float* gout, const float* net,
                                  const float* dnet, const float* delta, const float* gamma, float* g_net,
                                  float* g_dnet, float* g_gamma_part, socmx_stream_t stream);

/*
 * The reference's OTHER losses on the same rollout buffers, one launch per family member.
 * Matching family (method.py:289-478, 722-749: SOCM_const_M, SOCM_exp, SOCM_adjoint) -- the least-squares target:
 *   kind 0  target[i] = sum_{j>=i}^{K-1} q_j + nabla_g(X_K)                                        (M = I)
 *   kind 1  target[i] = sum_{j>=i} e^{-gamma (t_j - t_i)} (q_j + gamma v_j) + e^{-gamma (T - t_i)} nabla_g(X_K)
 *           and dtarget = d target / d gamma (gamma (1,) in device memory; method.py:371-478 trains it)
 *   kind 2  the costate recursion a_K = nabla_g(X_K), a_i = a_{i+1} + dt ((nabla_f_i + nabla_f_{i+1}) / 2
 *           + ((nabla_b_i + nabla_b_{i+1}) / 2)^T a_{i+1}) with the constant step dt = T / K (method.py:722-749): all K steps
 *           inside one kernel (d <= 64)
 * q, v (K,B,d), gT (B,d) from socmx_socm_prep_f32; states (K+1,B,d) for kind 2; unused pointers may be NULL.
 * socmx_socm_residual_f32 then gives objective = sum w |sigma^T (nabla_V - target)|^2 inv_norm (ADDED to objective[0]; workspace as for socmx_socm_target_fwd_f32) and
 * G = d objective / d nabla_V for ANY target in HBM (method.py:702-720's form, shared by the three).
 * Girsanov family (method.py:751-856: cross_entropy, variance, log-variance, moment): c (K,B) with
 *   c[i,m] = stop[i,m] ( dt (-<l,u>/lmbd + |l|^2/(2 lmbd) - [with_f] f(X_i)/lmbd) - sqrt(dt/lmbd) <l,eps> ),  l = -sigma^T nabla_V,
 * dt = frac[i,m] or ts[i+1]-ts[i]; the per-sample totals sum_i c[i,m] and the B-long loss formulas stay with the caller;
 * the backward writes G (K+1,B,d) = d obj / d nabla_V from gtotal (B,) = d obj / d total_m.
 */
int socmx_matching_target_f32(int32_t kind, const socmx_problem* problem, int32_t K, int32_t B, const float* ts, float T,
                              float dt, const float* gamma, const float* q, const float* v, const float* gT,
                              const float* states, float* target, float* dtarget, socmx_stream_t stream);
int socmx_socm_residual_f32(const socmx_problem* problem, int32_t K, int32_t B, const float* target, const float* nablaV,
                            const float* w, float inv_norm, float* G, float* objective, float* workspace,
                            socmx_stream_t stream);
int socmx_girsanov_fwd_f32(const socmx_problem* problem, int32_t K, int32_t B, float lmbd, int32_t with_f, const float* ts,
                           const float* nablaV, const float* noises, const float* controls, const float* states,
                           const float* frac, const float* stop, float* c, socmx_stream_t stream);
int socmx_girsanov_bwd_f32(const socmx_problem* problem, int32_t K, int32_t B, float lmbd, const float* ts,
                           const float* nablaV, const float* noises, const float* controls, const float* frac,
                           const float* stop, const float* gtotal, float* G, socmx_stream_t stream);

/*
 * Stopping-time SOCM target (method.py:484-507, 548-564, 597-613, 649-660, 692-701 with models.py:278-393
 * TwoBoundarySigmoidMLP): the pair matrices depend on the SAMPLE through its stopping time tau_m (method.py:525-530),
 *     M[p,m]     =              w I +  c0 N0[p] +  c1 N1[p]
 *     dM/ds[p,m] = nan_to_num( dw I + dc0 N0[p] + c0 dN0[p] + dc1 N1[p] + c1 dN1[p] )      (entry-wise: method.py:553-555)
 *     target[i,m] = sum_{j>=i} ( M[p,m] qx[j,m] - dM/ds[p,m] vx[j,m] ),  qx = q | nabla_g at j = K,  vx = v | 0.
 * The scalar gates (w, c0, c1) = (factor1 | exp_gamma3, fun_gamma2(factor1), 1 - exp_gamma3) of models.py:341-392 and their
 * s-derivatives are formed per (pair, sample) INSIDE the kernels from pair_t, pair_s (Np,) = the pair grid of
 * method.py:533-547, tau (B,), gammas (3,) = (gamma, gamma2, gamma3) in device memory and T_model = the model's own T
 * (models.py:287: 1.0 whatever cfg.method.T is) -- neither the reference's (Np, B, d, d) tensors nor (Np, B) gate fields
 * exist.  N0, N1, dN0, dN1 (Np, d, d) = the two network evaluations and their s-tangents (socmx_mnet_forward_f32, n_in = 3);
 * q, v (K,B,d) built with the per-sample fractional time steps (socmx_socm_prep_f32 with `frac`).  d <= 16.
 * The backward returns d obj / d (N0, N1, dN0, dN1) and ggamma_part (Np, 3): per-pair partial sums of
 * d obj / d (gamma, gamma2, gamma3) (the caller adds the Np rows), from gtarget = d obj / d target; deterministic.
 */
int socmx_socm_stopping_target_fwd_f32(int32_t d, int32_t K, int32_t B, const float* pair_t, const float* pair_s,
                                       const float* tau, const float* gammas, float T_model, const float* N0,
                                       const float* N1, const float* dN0, const float* dN1, const float* q,
                                       const float* v, const float* gT, float* target, socmx_stream_t stream);
int socmx_socm_stopping_target_bwd_f32(int32_t d, int32_t K, int32_t B, const float* pair_t, const float* pair_s,
                                       const float* tau, const float* gammas, float T_model, const float* N0,
                                       const float* N1, const float* dN0, const float* dN1, const float* q,
                                       const float* v, const float* gT, const float* gtarget, float* gN0, float* gN1,
                                       float* gdN0, float* gdN1, float* ggamma_part, socmx_stream_t stream);

/*
 * The scalar bookkeeping of one training iteration (main.py:313-322, 325-345, 354-359 with compute_EMA, utils.py:389-396) on
 * DEVICE-resident state, so that a captured iteration needs no host arithmetic: itr (1,) the iteration counter, norm (1,) the
 * EMA normalisation constant, ema_gn (1,) the EMA of the squared gradient norm.
 *   phase 0: ab[0..1] = the coefficients (A, B) of  ema_grad <- A ema_grad + B grad  for the current itr.
 *   phase 1: out[0..6] = [objective / norm, mean(w), std(w), gn, EMA(gn), gne, norm used]; then norm <- EMA(mean(w)),
 *            itr <- itr + 1.   gn / gne / ema_gn may be NULL (no gradient telemetry).
 */
int socmx_iteration_scalars_f32(int32_t phase, float* itr, float* norm, float* ema_gn, const float* w_mean,
                                const float* w_std, const float* obj, const float* gn, const float* gne,
                                double c_norm, double c_grad, float* ab, float* out, socmx_stream_t stream);
/* The same with a HISTORY: phase 1 also writes out[0..6] and extra[0] (0 if extra is NULL) into row itr of hist (hist_rows, 8) while
 * itr < hist_rows -- the iteration's values stay where they were written (main.py keeps every iteration's loss, weight statistics
 * and telemetry, main.py:398-413), so a replayed hipGraph's static `out` needs no device copy between two replays.  hist NULL: as above. */
int socmx_iteration_scalars_hist_f32(int32_t phase, float* itr, float* norm, float* ema_gn, const float* w_mean,
                                     const float* w_std, const float* obj, const float* gn, const float* gne,
                                     double c_norm, double c_grad, float* ab, float* out, float* hist, int32_t hist_rows,
                                     const float* extra, socmx_stream_t stream);

/*
 * Adam update of one parameter group from a FLAT gradient buffer, fused with the gradient telemetry of main.py:325-345:
 * torch.optim.Adam's rule as main.py:174-238, 347-349 configure it (no weight decay, no amsgrad) -- ONE launch instead of the
 * optimiser's multi-tensor launches, two dot products, an EMA update and a coefficient kernel per iteration.
 *   tensors (ntensors <= 64 records in DEVICE memory): per parameter tensor its data / exp_avg / exp_avg_sq / step pointers,
 *     its element count and its offset into grad; the step tensors hold the number of updates so far (float, all equal) and
 *     are advanced by the call.
 *   grad (total,): the flat gradient (socmx_unet_backward_f32's output order).
 *   ema_grad (total,) or NULL: telemetry  ema <- A ema + B grad, (A, B) from the iteration counter itr (1,) exactly as
 *     socmx_iteration_scalars_f32 phase 0 computes them (c_grad = the EMA coefficient).
 *   sums_out[0..1] = |grad|^2, |ema_grad|^2 (0 without ema_grad): per-workgroup partials combined in a fixed order by the
 *     last workgroup to finish -- deterministic, no floating-point atomics.
 *   scratch: 4 + 2 * ceil(total / 1024) floats of caller-owned device memory; scratch[0] (the finished-workgroup ticket)
 *     must be zero before the first call, and the call leaves it zero again.
 */
typedef struct socmx_adam_tensor {
  float* p;
  float* exp_avg;
  float* exp_avg_sq;
  float* step;
  int64_t n;
  int64_t offset;
} socmx_adam_tensor;
int socmx_adam_step_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                        float* ema_grad, const float* itr, double c_grad, float lr, float beta1, float beta2, float eps,
                        float* scratch, float* sums_out, socmx_stream_t stream);
/* The same launch followed, in its last workgroup to finish, by phase 1 of socmx_iteration_scalars_f32 with gn = sums_out[0] and
 * gne = sums_out[1] (both absent when ema_grad is NULL): out[0..6], ema_gn, norm and itr are updated exactly as that call would
 * -- one launch less at the end of every iteration.  itr is read by the Adam part (EMA coefficients) and advanced afterwards. */
int socmx_adam_step_scalars_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                                float* ema_grad, float* itr, double c_grad, float lr, float beta1, float beta2, float eps,
                                float* scratch, float* sums_out, float* norm, float* ema_gn, const float* w_mean,
                                const float* w_std, const float* obj, double c_norm, float* out, socmx_stream_t stream);
/* ... and with the history of socmx_iteration_scalars_hist_f32 */
int socmx_adam_step_scalars_hist_f32(const socmx_adam_tensor* tensors, int32_t ntensors, int64_t total, const float* grad,
                                     float* ema_grad, float* itr, double c_grad, float lr, float beta1, float beta2, float eps,
                                     float* scratch, float* sums_out, float* norm, float* ema_gn, const float* w_mean,
                                     const float* w_std, const float* obj, double c_norm, float* out, float* hist,
                                     int32_t hist_rows, const float* extra, socmx_stream_t stream);

/*
 * Column sums of a tall row-major (R, C) matrix: out[c] = sum_r x[r][c].  Bias gradients of the nn.Linear layers
 * (models.py:212-228, 253-257) over the (K+1)*B trajectory rows / the Np pair rows; no reference counterpart beyond
 * autograd's reduction.  partial is a caller-owned workspace of socmx_colsum_blocks(R, C) * C floats; the result
 * is deterministic (fixed summation order).  out == NULL: leave the partials for socmx_linear_bwd_finish_f32.
 */
int32_t socmx_colsum_blocks(int64_t R, int32_t C);
int socmx_colsum_f32(const float* x, int64_t R, int32_t C, float* partial, float* out, socmx_stream_t stream);

/* ReLU backward fused with the bias-gradient reduction of a Linear+ReLU layer (models.py:212-228, 253-257):
 * gz = gy * (y > 0) with y the layer's ReLU output, out[c] = sum_r gz[r][c].  gz must not alias gy. */
int socmx_relu_bwd_colsum_f32(const float* gy, const float* y, int64_t R, int32_t C, float* gz, float* partial,
                              float* out, socmx_stream_t stream);

/*
 * Finishes the backward of one nn.Linear over many rows in ONE launch: gw (N = out*in) = tail (nullable) + the sum of
 * the S split-K slabs gw_parts (S, N) of the weight gradient, and gb (C) = the sum of the nblk column-sum partials
 * that socmx_colsum_f32 / socmx_relu_bwd_colsum_f32 leave in `partial` when called with out == NULL.
 */
int socmx_linear_bwd_finish_f32(const float* gw_parts, int32_t S, int64_t N, const float* tail, float* gw,
                                const float* partial, int32_t nblk, int32_t C, float* gb, socmx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SOCMX_H */
