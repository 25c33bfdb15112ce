// Micro-benchmark: what can the second wave of a SIMD issue while the first one streams independent v_mfma_f32_16x16x4_f32?
// One workgroup of 8 waves on one CU: waves 0..3 run NMF back-to-back MFMAs (one wave per SIMD), waves 4..7 run a chain of
// N instructions of one kind; every wave reports its own cycle count.  Printed: cycles of the MFMA waves alone, of the other
// kind alone, and of both together.   hipcc -O3 --offload-arch=gfx950 -o /tmp/ov tools/ubench/mfma_valu_overlap.hip && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int do_mfma, int do_other, int n, float* out, long long* cyc) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float r = 0.f;
  const long long t0 = clock64();
  if (wave < 4) {
    if (do_mfma) {
      f32x4 acc[8];
      for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
      float a = 1.f + lane, b = 2.f - lane;
      for (int it = 0; it < n; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    }
  } else if (do_other) {
    if (KIND == 0) {            // dependent fp32 FMA chain
      float x = lane;
      for (int it = 0; it < n; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
      r = x;
    } else if (KIND == 1) {     // dependent integer add chain
      int x = lane;
      for (int it = 0; it < n; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
      r = x;
    } else if (KIND == 3) {     // LDS reads
      float x = 0.f;
      for (int it = 0; it < n; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) x += lds[(lane + j * 64 + it) & 4095];
      r = x;
    } else if (KIND == 4) {     // 4 independent FMA chains
      float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      for (int it = 0; it < n; ++it)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x0));
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x1));
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x2));
          asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x3));
        }
      r = x0 + x1 + x2 + x3;
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
  const int n = 4000;
  long long h[3][8];
  for (int m = 0; m < 3; ++m) {
    const int dm = m != 1, dother = m != 0;
    hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(512), 0, 0, dm, dother, n, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h[m], cyc, 64, hipMemcpyDeviceToHost);
  }
  printf("%-28s per instruction: MFMA alone %5.1f | other alone %5.1f | together: MFMA %5.1f, other %5.1f cycles\n", name,
         h[0][0] / (8.0 * n), h[1][4] / (8.0 * n), h[2][0] / (8.0 * n), h[2][4] / (8.0 * n));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  run<0>("dependent v_fma_f32", out, cyc);
  run<4>("4 independent v_fma_f32", out, cyc);
  run<1>("dependent v_add_u32", out, cyc);
  run<3>("ds_read_b32 + add", out, cyc);
  return 0;
}
