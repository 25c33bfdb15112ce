// Micro-benchmark for the ONE-ROW rollout design: a matrix-vector stage chain on the VALU.
//   lane (q, n) = (k-group, neuron of a 16-block);  acc[lane] += x[k] * w[neuron][k]  as  v_fmac_f32_dpp  with the activation
//   broadcast inside each 16-lane row by DPP row_newbcast:j  (xv[lane] = x[k0 + 16 q + (lane & 15)]): 64 MACs per instruction,
//   every weight distinct (used once per step) -> weights live in VGPRs (resident), or arrive from L2 (ring) / LDS.
// Questions: (1) is row_newbcast what we think (checked against the CPU); (2) cycles per 16-fmac block with 2 waves per SIMD;
// (3) what a concurrent L2 weight stream / LDS weight reads cost.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_row1.hip -o tools/ubench/_bin/valu_row1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 fmacs on one activation register: weights w[0..15] <-> broadcast lanes 0..15, four accumulators round-robin
#define FM(J, A, W) "v_fmac_f32_dpp %" #A ", %4, %" #W " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void fmac16(float& a0, float& a1, float& a2, float& a3, float x, const float* w) {
  asm("s_nop 1\n\t" FM(0, 0, 5) FM(1, 1, 6) FM(2, 2, 7) FM(3, 3, 8) FM(4, 0, 9) FM(5, 1, 10) FM(6, 2, 11) FM(7, 3, 12)
          FM(8, 0, 13) FM(9, 1, 14) FM(10, 2, 15) FM(11, 3, 16) FM(12, 0, 17) FM(13, 1, 18) FM(14, 2, 19) FM(15, 3, 20)
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
      : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]),
        "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
}

// wave-uniform base (SGPR pair) + 32-bit lane offset + immediate: no 64-bit VGPR pointer per request
template <int IMM>
__device__ __forceinline__ f32x4 gload(const f32x4* sbase, uint32_t lane_off) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(lane_off), "s"(sbase), "n"(IMM) : "memory");
  return v;
}
#define GLOAD4(dst, base, off)           \
  dst[0] = gload<0>(base, off);          \
  dst[1] = gload<1024>(base, off);       \
  dst[2] = gload<2048>(base, off);       \
  dst[3] = gload<3072>(base, off);
template <int N>
__device__ __forceinline__ void vmwait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

// NRES resident blocks (16 regs each), NSTR streamed blocks per step (ring of PD blocks), NLDS blocks read from LDS per step
template <int NW, int NRES, int NSTR, int NLDS, int PD, int NBAR>
__global__ __launch_bounds__(NW * 64) void k(const float* __restrict__ wimg, const f32x4* __restrict__ simg, int steps,
                                             float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* X = lds;                       // 256 activations
  float* LW = lds + 256;                // NW * NLDS * 16 * 64 floats of LDS-resident weights
  for (int i = tid; i < 256; i += NW * 64) X[i] = 0.01f * (i % 7) - 0.02f;
  for (int i = tid; i < NW * NLDS * 1024; i += NW * 64) LW[i] = wimg[i % 4096];
  float wres[NRES > 0 ? NRES * 16 : 1];
#pragma unroll
  for (int r = 0; r < NRES * 16; ++r) wres[r] = wimg[(size_t)(wave * NRES * 16 + r) * 64 + lane];
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const f32x4* sb = simg + (size_t)wave * NSTR * 4 * 64;      // wave-uniform
  const uint32_t loff = lane * 16;
  f32x4 ring[PD > 0 ? PD : 1][4];
  if (NSTR > 0) {
#pragma unroll
    for (int s = 0; s < PD; ++s) { GLOAD4(ring[s], sb + (size_t)s * 256, loff) }
  }
  const long long t0 = clock64();
  for (int st = 0; st < steps; ++st) {
    float xv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xv[m] = X[(lane >> 4) * 64 + m * 16 + (lane & 15)];
    constexpr int NB = NRES + NSTR + NLDS;
    int ir = 0, is = 0, il = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // interleave the three sources proportionally
      const bool take_s = is < NSTR && (is * NB <= b * NSTR);
      const bool take_l = !take_s && il < NLDS && (il * NB <= b * NLDS);
      if (take_s) {
        const int s = is % PD;
        vmwait<(PD - 1) * 4>(ring[s][0], ring[s][1], ring[s][2], ring[s][3]);
        float w[16];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) w[c * 4 + e] = ring[s][c][e];
        fmac16(a0, a1, a2, a3, xv[b & 3], w);
        const int nb = (is + PD) % NSTR;   // wraps into the next step
        GLOAD4(ring[s], sb + (size_t)nb * 256, loff)
        ++is;
      } else if (take_l) {
        float w[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(LW + ((size_t)((wave * NLDS + il) * 4 + c) * 64 + lane) * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) w[c * 4 + e] = v[e];
        }
        fmac16(a0, a1, a2, a3, xv[b & 3], w);
        ++il;
      } else {
        fmac16(a0, a1, a2, a3, xv[b & 3], wres + ir * 16);
        ++ir;
      }
      if (NBAR > 0 && (b + 1) % ((NB + NBAR - 1) / NBAR) == 0) {
        if (lane < 16) X[wave * 16 + lane] = fmaxf(a0 + a1, 0.f) * 1e-3f;
        __syncthreads();
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * NW + wave] = t1 - t0;
  out[(size_t)blockIdx.x * NW * 64 + tid] = a0 + a1 + a2 + a3;
}

static float* d_w;
static f32x4* d_s;
static float* d_out;
static long long* d_cyc;

template <int NW, int NRES, int NSTR, int NLDS, int PD, int NBAR>
void run(int blocks, double ghz) {
  const int steps = 200;
  const size_t lds = (256 + (size_t)NW * NLDS * 1024) * 4;
  auto kern = k<NW, NRES, NSTR, NLDS, PD, NBAR>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, d_w, d_s, 3, d_out, d_cyc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, d_w, d_s, steps, d_out, d_cyc);
  hipEventRecord(e1);
  hipError_t err = hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * NW);
  hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  const int nb = NRES + NSTR + NLDS;
  const double cyc_step = ms * 1e-3 * ghz * 1e9 / steps;
  printf("NW=%d res=%d str=%d lds=%d PD=%d bar=%d blocks=%d: %.3f ms, %.0f cycles/step (wall*%.1fGHz), clock64 max %lld/step; "
         "%d fmac/wave/step -> ideal %d cyc; %.2f cyc per fmac per SIMD; stream %.0f KB/step = %.1f B/clk/CU  [%s]\n",
         NW, NRES, NSTR, NLDS, PD, NBAR, blocks, ms, cyc_step, mx / steps, ghz, nb * 16, nb * 16 * (NW / 4) * 4,
         cyc_step / (nb * 16 * (NW / 4)), NW * NSTR * 4.0, NW * NSTR * 4096.0 / cyc_step, hipGetErrorString(err));
}

__global__ void check_k(const float* x, const float* w, float* o) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  float wv[16];
  for (int j = 0; j < 16; ++j) wv[j] = w[j * 64 + threadIdx.x];
  fmac16(a0, a1, a2, a3, x[threadIdx.x], wv);
  o[threadIdx.x] = (a0 + a1) + (a2 + a3);
}

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  hipMalloc(&d_w, 64 << 20);
  hipMalloc(&d_s, 64 << 20);
  hipMalloc(&d_out, 4 << 20);
  hipMalloc(&d_cyc, 1 << 20);
  std::vector<float> hw(16 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f) - 0.5f;
  hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_s, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  {  // semantics: o[l] = sum_j x[(l & ~15) + j] * w[j][l]
    std::vector<float> hx(64), ho(64);
    for (int l = 0; l < 64; ++l) hx[l] = 0.1f * l - 1.f;
    float *dx, *dout;
    hipMalloc(&dx, 256);
    hipMalloc(&dout, 256);
    hipMemcpy(dx, hx.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check_k, dim3(1), dim3(64), 0, 0, dx, d_w, dout);
    hipMemcpy(ho.data(), dout, 256, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l) {
      double ref = 0;
      for (int j = 0; j < 16; ++j) ref += (double)hx[(l & ~15) + j] * hw[j * 64 + l];
      worst = fmax(worst, fabs(ref - ho[l]));
    }
    printf("row_newbcast check: max |err| = %.3g %s\n", worst, worst < 1e-5 ? "OK" : "MISMATCH");
  }
  // (2) pure VALU, resident weights, 8 waves (2 per SIMD), 12 blocks = 192 regs
  run<8, 12, 0, 0, 0, 0>(128, ghz);
  run<8, 12, 0, 0, 0, 6>(128, ghz);
  run<4, 12, 0, 0, 0, 0>(128, ghz);
  // (3) resident + L2 stream
  run<8, 10, 5, 0, 2, 0>(128, ghz);
  run<8, 10, 5, 0, 2, 6>(128, ghz);
  run<8, 10, 10, 0, 2, 6>(128, ghz);
  run<8, 10, 10, 0, 2, 6>(256, ghz);
  run<8, 9, 10, 0, 3, 6>(128, ghz);
  // (4) + LDS-resident weights (4 blocks = 64 regs-equivalents per lane = 128 KiB)
  run<8, 10, 0, 4, 0, 6>(128, ghz);
  run<8, 10, 5, 4, 2, 6>(128, ghz);
  run<8, 10, 7, 4, 2, 6>(128, ghz);
  run<8, 10, 7, 4, 2, 6>(256, ghz);
  return 0;
}
