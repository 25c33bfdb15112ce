// Micro-benchmark: cost of a workgroup barrier (and of an LDS write + barrier + LDS read hand-off) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int mode, int iters, float* out, long long* cyc) {
  __shared__ float buf[4096];
  float v = threadIdx.x;
  buf[threadIdx.x] = v;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (mode == 0) { __syncthreads(); }
    else if (mode == 1) { buf[threadIdx.x] = v; __syncthreads(); v += buf[(threadIdx.x + 64) & (blockDim.x - 1)]; __syncthreads(); }
    else if (mode == 2) { __builtin_amdgcn_s_barrier(); }
    else if (mode == 3) { long long t = clock64(); v += (float)(t & 1); }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096);
  const char* names[] = {"__syncthreads()", "LDS write+sync+read+sync", "raw s_barrier", "clock64() read"};
  for (int threads : {256, 512, 1024})
    for (int mode = 0; mode < 4; ++mode) {
      const int iters = 2000;
      hipLaunchKernelGGL(k, dim3(8), dim3(threads), 0, 0, mode, iters, out, cyc);
      hipDeviceSynchronize();
      long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
      printf("threads=%4d %-28s : %.1f cycles per iteration\n", threads, names[mode], (double)h[0] / iters);
    }
  return 0;
}
