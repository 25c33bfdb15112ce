// Micro-benchmark: how fast can ONE workgroup on one CU stream an L2-resident buffer with 16-byte
// per-lane loads (the rollout kernel's weight-streaming pattern)?  Prints bytes/clk per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MODE>
__global__ void stream_kernel(const f32x4* __restrict__ buf, int n_frag /*1KB fragments*/, int reps, float* out,
                              long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int f = wave * DEPTH; f + DEPTH <= n_frag; f += nw * DEPTH) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) {
        const f32x4* p = buf + (size_t)(f + s) * 64 + lane;
        if (MODE == 0) v[s] = *p;
        else if (MODE == 1) v[s] = __builtin_nontemporal_load(p);
      }
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) acc += v[s];
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int DEPTH, int MODE>
void run(const char* name, const f32x4* buf, int n_frag, int blocks, int threads, float* out, long long* cyc) {
  const int reps = 200;
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(blocks), dim3(threads), 0, 0, buf, n_frag, 2, out, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(blocks), dim3(threads), 0, 0, buf, n_frag, reps, out, cyc);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= blocks;
  const double bytes = (double)n_frag * 1024 * reps;
  printf("%-28s blocks=%3d threads=%4d depth=%d : %.1f B/clk per CU (%.0f cycles per pass)\n", name, blocks, threads,
         DEPTH, bytes / avg, avg / reps);
}

int main() {
  const int n_frag = 672;  // 672 KB ~ the U-Net image
  f32x4* buf; float* out; long long* cyc;
  hipMalloc(&buf, (size_t)n_frag * 1024);
  hipMemset(buf, 0, (size_t)n_frag * 1024);
  hipMalloc(&out, 1024 * 1024 * 4);
  hipMalloc(&cyc, 4096 * 8);
  for (int blocks : {1, 8, 256}) {
    run<4, 0>("plain", buf, n_frag, blocks, 512, out, cyc);
    run<8, 0>("plain", buf, n_frag, blocks, 512, out, cyc);
    run<8, 0>("plain", buf, n_frag, blocks, 1024, out, cyc);
    run<16, 0>("plain", buf, n_frag, blocks, 512, out, cyc);
    run<8, 0>("plain", buf, n_frag, blocks, 256, out, cyc);
    run<8, 1>("nontemporal", buf, n_frag, blocks, 512, out, cyc);
  }
  return 0;
}
