// Micro-benchmark: the rollout kernel's inner loop in isolation.  8 waves (2/SIMD) on one CU, each runs
// dependent v_mfma_f32_16x16x4_f32 chains fed by (a) registers only, (b) + LDS activation fragments,
// (c) + weight fragments streamed from L2.  Prints cycles per MFMA per SIMD (ideal: 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NB, int MODE, int PD>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ w, int chunks, float* out, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float X[16 * 260];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 16 * 260; i += 512) X[i] = 0.001f * i;
  __syncthreads();
  const float* xrow = X + (lane & 15) * 260 + 4 * (lane >> 4);
  f32x4 acc[NB];
  for (int j = 0; j < NB; ++j) acc[j] = f32x4{0, 0, 0, 0};
  const f32x4* wb = w + (size_t)wave * 4096 * 64 + lane;
  f32x4 ring[PD][NB];
  if (MODE >= 2) for (int s = 0; s < PD; ++s) for (int j = 0; j < NB; ++j) ring[s][j] = wb[(size_t)(s * NB + j) * 64];
  else for (int s = 0; s < PD; ++s) for (int j = 0; j < NB; ++j) ring[s][j] = f32x4{1.f, 2.f, 3.f, 4.f};
  const long long t0 = clock64();
  f32x4 bx = {1.f, 1.f, 1.f, 1.f};
  for (int kc = 0; kc + PD <= chunks; kc += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      if (MODE >= 1) bx = *reinterpret_cast<const f32x4*>(xrow + ((kc + s) & 15) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[s][j][i], bx[i], acc[j], 0, 0, 0);
      if (MODE >= 2)
#pragma unroll
        for (int j = 0; j < NB; ++j) ring[s][j] = wb[(size_t)(((kc + s + PD) & 1023) * NB + j) * 64];
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  float r = 0;
  for (int j = 0; j < NB; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int NB, int MODE, int PD>
void run(const char* name, const f32x4* w, float* out, long long* cyc) {
  const int chunks = 4096;
  hipLaunchKernelGGL((k<NB, MODE, PD>), dim3(8), dim3(512), 0, 0, w, 64, out, cyc);
  hipLaunchKernelGGL((k<NB, MODE, PD>), dim3(8), dim3(512), 0, 0, w, chunks, out, cyc);
  hipDeviceSynchronize();
  std::vector<long long> h(64);
  hipMemcpy(h.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto c : h) mx = c > mx ? c : mx;
  const double mfma_per_simd = (double)chunks * 4 * NB * 2;  // 2 waves per SIMD
  printf("%-40s NB=%d PD=%d : %.1f cycles per MFMA per SIMD\n", name, NB, PD, mx / mfma_per_simd);
}

int main() {
  f32x4* w; float* out; long long* cyc;
  hipMalloc(&w, (size_t)8 * 4096 * 1024 + (1 << 20));
  hipMemset(w, 0, (size_t)8 * 4096 * 1024 + (1 << 20));
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  run<1, 0, 8>("registers only", w, out, cyc);
  run<2, 0, 4>("registers only", w, out, cyc);
  run<4, 0, 2>("registers only", w, out, cyc);
  run<1, 1, 8>("+ LDS activation reads", w, out, cyc);
  run<2, 1, 4>("+ LDS activation reads", w, out, cyc);
  run<1, 2, 8>("+ LDS + L2 weight stream", w, out, cyc);
  run<2, 2, 4>("+ LDS + L2 weight stream", w, out, cyc);
  run<2, 2, 8>("+ LDS + L2 weight stream", w, out, cyc);
  run<4, 2, 2>("+ LDS + L2 weight stream", w, out, cyc);
  run<4, 2, 4>("+ LDS + L2 weight stream", w, out, cyc);
  return 0;
}
