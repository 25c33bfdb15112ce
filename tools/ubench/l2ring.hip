// Micro-benchmark: steady-state streaming of an L2-resident image by ONE workgroup, each wave keeping D 1-KiB fragments in
// flight (a) in a VGPR ring (asm loads, written-out vmcnt), (b) LDS-direct (global_load_lds_dwordx4, no VGPRs for the data in
// flight) followed by a ds_read_b128 of the landed fragment.  Prints bytes/clk per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ void ring_kernel(const f32x4* __restrict__ buf, int frags_per_wave, int reps, float* out, long long* cycles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* p = buf + (size_t)wave * frags_per_wave * 64 + lane;
  f32x4 acc = {0, 0, 0, 0};
  f32x4 ring[D];
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int s = 0; s < D; ++s) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[s]) : "v"(p + (size_t)s * 64) : "memory");
    for (int f = 0; f < frags_per_wave; f += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        asm volatile("" : "+v"(ring[s]));
        acc += ring[s];
        const int nf = f + s + D < frags_per_wave ? f + s + D : frags_per_wave - 1;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ring[s]) : "v"(p + (size_t)nf * 64) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int D>
__global__ void lds_kernel(const f32x4* __restrict__ buf, int frags_per_wave, int reps, float* out, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* p = buf + (size_t)wave * frags_per_wave * 64 + lane;
  float* mine = lds + wave * D * 256;                       // D slots of 1 KiB
  f32x4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int s = 0; s < D; ++s)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)s * 64),
                                       (__attribute__((address_space(3))) void*)(mine + s * 256), 16, 0, 0);
    for (int f = 0; f < frags_per_wave; f += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        const f32x4 v = *reinterpret_cast<const f32x4*>(mine + s * 256 + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        const int nf = f + s + D < frags_per_wave ? f + s + D : frags_per_wave - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)nf * 64),
                                         (__attribute__((address_space(3))) void*)(mine + s * 256), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int D, bool LDS>
void run(const f32x4* buf, int n_frag, int blocks, int waves, float* out, long long* cyc) {
  const int reps = 100, fpw = n_frag / waves / D * D;
  for (int it = 0; it < 2; ++it) {
    if (LDS) hipLaunchKernelGGL((lds_kernel<D>), dim3(blocks), dim3(waves * 64), waves * D * 1024, 0, buf, fpw, it ? reps : 2, out, cyc);
    else hipLaunchKernelGGL((ring_kernel<D>), dim3(blocks), dim3(waves * 64), 0, 0, buf, fpw, it ? reps : 2, out, cyc);
    hipDeviceSynchronize();
  }
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto c : h) avg += c;
  avg /= blocks;
  printf("%-10s blocks=%3d waves=%2d in flight per wave=%2d KiB : %6.1f B/clk per CU\n", LDS ? "lds-direct" : "vgpr ring", blocks, waves, D,
         (double)fpw * waves * 1024 * reps / avg);
}

int main() {
  const int n_frag = 672;
  f32x4* buf; float* out; long long* cyc;
  hipMalloc(&buf, (size_t)n_frag * 1024); hipMemset(buf, 0, (size_t)n_frag * 1024);
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096 * 8);
  hipFuncSetAttribute((const void*)lds_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int blocks : {1, 32, 256}) {
    run<4, false>(buf, n_frag, blocks, 8, out, cyc);
    run<8, false>(buf, n_frag, blocks, 8, out, cyc);
    run<12, false>(buf, n_frag, blocks, 8, out, cyc);
    run<16, false>(buf, n_frag, blocks, 8, out, cyc);
    run<8, false>(buf, n_frag, blocks, 16, out, cyc);
    run<4, true>(buf, n_frag, blocks, 8, out, cyc);
    run<8, true>(buf, n_frag, blocks, 8, out, cyc);
    run<16, true>(buf, n_frag, blocks, 8, out, cyc);
    run<8, true>(buf, n_frag, blocks, 16, out, cyc);
  }
  return 0;
}
