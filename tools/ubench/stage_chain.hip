// Micro-benchmark: the FLOOR of the rollout kernel's per-step network evaluation -- a 6-stage, 8-wave, 16-row chain of
// v_mfma_f32_16x16x4_f32 stages with exactly the default network's per-wave MFMA counts and work split
// (socmx_unet.h: stage s = [MFMAs fed by ds_read_b128 activation fragments] -> ReLU -> ds_write_b128 -> s_barrier),
// but with every weight fragment ALREADY IN REGISTERS (MODE 0: an infinitely fast L2) or streamed from L2 through the
// same 8-fragment ring as the product (MODE 1).  No SDE step, no noise, no trajectory stores.
//   hipcc --offload-arch=gfx950 -O3 -o stage_chain stage_chain.hip && ./stage_chain
// Prints shader cycles per step; the pure MFMA issue time of the six stages is 676 MFMAs per SIMD x 32 = 21,632 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S1 = 260, S2 = 132, S3 = 68, S0 = 20;

// one GEMM of a stage for this wave: NB output blocks x KC input chunks, activations from the LDS tile X (row stride S)
// (MODE 0: the NB fragments in `wreg` were loaded before the timed loop and are reused for every chunk -- distinct per block,
//  or the compiler would merge the blocks' accumulators)
template <int NB, int MODE>
__device__ __forceinline__ void gemm(f32x4 (&acc)[NB], const float* X, int S, int KC, const f32x4* wb, int lane,
                                     const f32x4* wreg) {
  const float* xrow = X + (lane & 15) * S + 4 * (lane >> 4);
  f32x4 ring[2][NB];
  for (int j = 0; j < NB; ++j) { ring[0][j] = MODE ? wb[(size_t)j * 64] : wreg[j]; ring[1][j] = ring[0][j]; }
  for (int kc = 0; kc < KC; ++kc) {
    const f32x4 bx = *reinterpret_cast<const f32x4*>(xrow + kc * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[kc & 1][j][i], bx[i], acc[j], 0, 0, 0);
    if (MODE)
#pragma unroll
      for (int j = 0; j < NB; ++j) ring[kc & 1][j] = wb[(size_t)(((kc + 2) * NB + j) & 1023) * 64];
  }
}

template <int NB>
__device__ __forceinline__ void finish(f32x4 (&acc)[NB], float* Y, int SY, int blk0, int lane) {
#pragma unroll
  for (int j = 0; j < NB; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = acc[j][i] < 0.f ? 0.f : acc[j][i];
    *reinterpret_cast<f32x4*>(Y + (lane & 15) * SY + (blk0 + j * 8) * 16 + 4 * (lane >> 4)) = acc[j];
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void chain(const f32x4* __restrict__ w, int steps, float* out, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[16 * (S0 + S1 + S2 + S3 + S2 + S1 + S0) + 8 * 256];
  float* X0 = lds; float* R1 = X0 + 16 * S0; float* R2 = R1 + 16 * S1; float* R3 = R2 + 16 * S2;
  float* O2 = R3 + 16 * S3; float* O1 = O2 + 16 * S2; float* GV = O1 + 16 * S1; float* P = GV + 16 * S0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 16 * (S0 + S1 + S2 + S3 + S2 + S1 + S0); i += 512) lds[i] = 1e-3f * (i & 63);
  __syncthreads();
  const f32x4* wb = w + (size_t)wave * 1024 * 64 + lane;
  f32x4 wr[2];
  for (int j = 0; j < 2; ++j) wr[j] = wb[(size_t)j * 64];       // runtime values (zeros in this harness), opaque to the compiler
  asm volatile("" : "+v"(wr[0]), "+v"(wr[1]));
  const long long t0 = clock64();
  for (int k = 0; k < steps; ++k) {
    { f32x4 a[2] = {}; gemm<2, MODE>(a, X0, S0, 1, wb, lane, wr); finish<2>(a, R1, S1, wave, lane); }            // down_0: 16 -> 256
    __syncthreads();
    { f32x4 a[1] = {}; gemm<1, MODE>(a, R1, S1, 16, wb, lane, wr); finish<1>(a, R2, S2, wave, lane); }           // down_1: 256 -> 128
    __syncthreads();
    if (wave < 4) { f32x4 a[1] = {}; gemm<1, MODE>(a, R2, S2, 8, wb, lane, wr); finish<1>(a, R3, S3, wave, lane); }  // down_2
    __syncthreads();
    { f32x4 a[1] = {}; gemm<1, MODE>(a, R3, S3, 4, wb, lane, wr); gemm<1, MODE>(a, R2, S2, 8, wb, lane, wr); finish<1>(a, O2, S2, wave, lane); }
    __syncthreads();
    { f32x4 a[2] = {}; gemm<2, MODE>(a, O2, S2, 8, wb, lane, wr); gemm<2, MODE>(a, R1, S1, 16, wb, lane, wr); finish<2>(a, O1, S1, wave, lane); }
    __syncthreads();
    {                                                                                                           // up_0: split-K over the 8 waves
      f32x4 a[1] = {};
      gemm<1, MODE>(a, O1 + wave * 32, S1, 2, wb, lane, wr);
      *reinterpret_cast<f32x4*>(P + wave * 256 + lane * 4) = a[0];
      __syncthreads();
      if (threadIdx.x < 256) {
        float v = 0.f;
        for (int p = 0; p < 8; ++p) v += P[p * 256 + threadIdx.x];
        X0[(threadIdx.x >> 4) * S0 + 1 + (threadIdx.x & 15) % 15] = v * 1e-6f;     // feeds the next step (as the SDE update does)
      }
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = X0[threadIdx.x % 256] + R1[threadIdx.x];
}

template <int MODE>
void run(const char* name, const f32x4* w, float* out, long long* cyc, int blocks) {
  const int steps = 200;
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(512), 0, 0, w, 10, out, cyc);
  hipLaunchKernelGGL(chain<MODE>, dim3(blocks), dim3(512), 0, 0, w, steps, out, cyc);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * 8);
  hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto c : h) mx = c > mx ? c : mx;
  printf("%-58s %2d workgroups: %8.0f cycles per step (MFMA issue floor 21632)  -> %.0f %% MFMA-busy\n", name, blocks, mx / steps,
         100.0 * 21632 / (mx / steps));
}

int main() {
  f32x4* w; float* out; long long* cyc;
  hipMalloc(&w, (size_t)8 * 1024 * 1024 + (1 << 20)); hipMemset(w, 0, (size_t)8 * 1024 * 1024 + (1 << 20));
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  run<0>("weights in registers (stage chain only)", w, out, cyc, 8);
  run<1>("weights streamed from L2 (2-deep ring)", w, out, cyc, 8);
  run<0>("weights in registers, two workgroups per CU", w, out, cyc, 512);
  return 0;
}
