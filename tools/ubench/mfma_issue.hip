// Micro-benchmark: issue rate of independent v_mfma_f32_16x16x4_f32 from W waves per SIMD (registers only, NB accumulators
// per wave), one workgroup per CU on `blocks` CUs.  Prints shader cycles per MFMA per SIMD (pipe-bound ideal: 32) and the
// shader clock implied by the wall time.    hipcc -O3 --offload-arch=gfx950 -o /tmp/mi tools/ubench/mfma_issue.hip && /tmp/mi
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int NB>
__global__ __launch_bounds__(WAVES * 64) void k(int iters, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NB];
  for (int j = 0; j < NB; ++j) acc[j] = f32x4{0, 0, 0, 0};
  float a = 1.f + lane, b = 2.f - lane;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
  float r = 0;
  for (int j = 0; j < NB; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}

template <int WAVES, int NB>
void run(int blocks, float* out, long long* cyc) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<WAVES, NB>), dim3(blocks), dim3(WAVES * 64), 0, 0, 100, out, cyc);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<WAVES, NB>), dim3(blocks), dim3(WAVES * 64), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * WAVES);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto c : h) mx = c > mx ? c : mx;
  const double per_simd = (double)iters * 4 * NB * (WAVES / 4.0);
  const double tf = 2048.0 * iters * 4 * NB * WAVES * blocks / (ms * 1e-3) / 1e12;
  printf("blocks=%4d waves/SIMD=%d NB=%d : %.1f counter ticks per MFMA per SIMD, %.3f ms, %.1f TFLOP/s (%.1f ns per MFMA per SIMD)\n", blocks,
         WAVES / 4, NB, mx / per_simd, ms, tf, ms * 1e6 / per_simd);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 20);
  for (int blocks : {8, 256, 1024}) {
    run<4, 8>(blocks, out, cyc);
    run<8, 8>(blocks, out, cyc);
    run<8, 4>(blocks, out, cyc);
    run<16, 4>(blocks, out, cyc);
    run<4, 16>(blocks, out, cyc);
  }
  return 0;
}
