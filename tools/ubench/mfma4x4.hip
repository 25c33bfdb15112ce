// Micro-benchmark for the small-batch design: 4-row tiles with v_mfma_f32_4x4x1_16b_f32 (64 neurons x 4 rows x 1 k
// per instruction), weights streamed from L2 (1 KiB per 4 MFMAs), activations broadcast from LDS.
// Prints L2 bytes/clk per CU and cycles per 1-KiB fragment per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NWAVES, int PD, int MODE>
__global__ __launch_bounds__(NWAVES * 64) void k(const f32x4* __restrict__ w, int frags_per_wave, int reps, float* out,
                                                 long long* cyc) {
  __shared__ __attribute__((aligned(16))) float X[4 * 264];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4 * 264; i += NWAVES * 64) X[i] = 0.001f * i;
  __syncthreads();
  const float* xrow = X + (lane & 3) * 264;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  const f32x4* wb = w + (size_t)wave * frags_per_wave * 64 + lane;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    f32x4 ring[PD];
#pragma unroll
    for (int s = 0; s < PD; ++s) ring[s] = wb[(size_t)s * 64];
    for (int f = 0; f + PD <= frags_per_wave; f += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
        f32x4 bx = {1.f, 1.f, 1.f, 1.f};
        if (MODE >= 1) bx = *reinterpret_cast<const f32x4*>(xrow + ((f + s) & 63) * 4);
        const f32x4 a = ring[s];
        if (s & 1) {
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bx[0], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bx[1], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bx[2], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], bx[3], acc1, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bx[0], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bx[1], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bx[2], acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[3], bx[3], acc0, 0, 0, 0);
        }
        const int nf = f + s + PD;
        ring[s] = wb[(size_t)(nf < frags_per_wave ? nf : frags_per_wave - 1) * 64];
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * NWAVES + wave] = t1 - t0;
  out[blockIdx.x * NWAVES * 64 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int NWAVES, int PD, int MODE>
void run(const char* name, const f32x4* w, float* out, long long* cyc, int blocks) {
  const int total_frags = 672;  // 672 KiB image split over the waves
  const int fpw = total_frags / NWAVES / PD * PD;
  const int reps = 100;
  hipLaunchKernelGGL((k<NWAVES, PD, MODE>), dim3(blocks), dim3(NWAVES * 64), 0, 0, w, fpw, 2, out, cyc);
  hipLaunchKernelGGL((k<NWAVES, PD, MODE>), dim3(blocks), dim3(NWAVES * 64), 0, 0, w, fpw, reps, out, cyc);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * NWAVES);
  hipMemcpy(h.data(), cyc, blocks * NWAVES * 8, hipMemcpyDeviceToHost);
  double mx = 0;
  for (auto c : h) mx = c > mx ? c : mx;
  const double bytes = (double)fpw * NWAVES * 1024 * reps;
  printf("%-22s blocks=%3d waves=%2d PD=%d : %6.1f B/clk per CU, %6.1f cycles per 1KiB fragment per wave, pass=%.0f cycles\n",
         name, blocks, NWAVES, PD, bytes / mx, mx / ((double)fpw * reps), mx / reps);
}

int main() {
  f32x4* w; float* out; long long* cyc;
  hipMalloc(&w, (size_t)1 << 22); hipMemset(w, 0, (size_t)1 << 22);
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 1 << 16);
  for (int blocks : {8, 32}) {
    run<16, 6, 0>("mfma+L2", w, out, cyc, blocks);
    run<16, 6, 1>("mfma+L2+LDS", w, out, cyc, blocks);
    run<16, 4, 1>("mfma+L2+LDS", w, out, cyc, blocks);
    run<16, 8, 1>("mfma+L2+LDS", w, out, cyc, blocks);
    run<8, 6, 1>("mfma+L2+LDS", w, out, cyc, blocks);
    run<8, 12, 1>("mfma+L2+LDS", w, out, cyc, blocks);
    run<4, 12, 1>("mfma+L2+LDS", w, out, cyc, blocks);
  }
  return 0;
}
