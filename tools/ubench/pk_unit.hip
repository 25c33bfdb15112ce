// Micro-benchmark: one "unit" of the packed-fma one-row rollout kernel (csrc/socmx_rollout1p.hip) in isolation -- sixteen
// v_pk_fma_f32 of two resident weight blocks against sixteen activations -- with the activations (a) gathered by v_readlane_b32
// into SGPR pairs, (b) read from LDS as broadcast ds_read_b128 into VGPRs one half ahead.  Cycles per unit, one and two waves
// per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_unit.hip -o tools/ubench/_bin/pk_unit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int UNITS_PER_GATHER>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ w, int iters, float* out, long long* cyc, int nw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 1024; i += blockDim.x) lds[i] = 1e-3f * (i & 15);
  __syncthreads();
  f32x4 wa[4][4];
  for (int b = 0; b < 4; ++b)
    for (int c = 0; c < 4; ++c) wa[b][c] = w[(wave * 16 + b * 4 + c) * 64 + lane] * 1e-3f;
  float y = 1e-3f * lane;
  f32x2 A[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
      float x[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) x[kk] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), kk));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNITS_PER_GATHER; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 x0 = {x[4 * c], x[4 * c + 1]}, x1 = {x[4 * c + 2], x[4 * c + 3]};
          A[2 * (u & 1)] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1)][c][0], wa[2 * (u & 1)][c][1]}, x0, A[2 * (u & 1)]);
          A[2 * (u & 1) + 1] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1) + 1][c][0], wa[2 * (u & 1) + 1][c][1]}, x0, A[2 * (u & 1) + 1]);
          A[2 * (u & 1)] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1)][c][2], wa[2 * (u & 1)][c][3]}, x1, A[2 * (u & 1)]);
          A[2 * (u & 1) + 1] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1) + 1][c][2], wa[2 * (u & 1) + 1][c][3]}, x1, A[2 * (u & 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      f32x4 X[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) X[c] = *reinterpret_cast<const f32x4*>(lds + 64 * wave + ((it & 3) * 16) + 4 * c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNITS_PER_GATHER; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x2 x0 = {X[c][0], X[c][1]}, x1 = {X[c][2], X[c][3]};
          A[2 * (u & 1)] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1)][c][0], wa[2 * (u & 1)][c][1]}, x0, A[2 * (u & 1)]);
          A[2 * (u & 1) + 1] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1) + 1][c][0], wa[2 * (u & 1) + 1][c][1]}, x0, A[2 * (u & 1) + 1]);
          A[2 * (u & 1)] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1)][c][2], wa[2 * (u & 1)][c][3]}, x1, A[2 * (u & 1)]);
          A[2 * (u & 1) + 1] = __builtin_elementwise_fma(f32x2{wa[2 * (u & 1) + 1][c][2], wa[2 * (u & 1) + 1][c][3]}, x1, A[2 * (u & 1) + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    y += A[0].x * 1e-9f;
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + tid] = (A[0].x + A[0].y) + (A[1].x + A[1].y) + (A[2].x + A[2].y) + (A[3].x + A[3].y);
}

template <int MODE, int U>
void run(const f32x4* d_w, float* d_out, long long* d_cyc, const char* name) {
  const int iters = 2000, blocks = 128;
  for (int nw : {4, 8}) {
    hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(nw * 64), 4096, 0, d_w, 10, d_out, d_cyc, nw);
    hipLaunchKernelGGL((k<MODE, U>), dim3(blocks), dim3(nw * 64), 4096, 0, d_w, iters, d_out, d_cyc, nw);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * nw);
    hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("%-58s %d unit(s) per 16 activations, %d waves/CU: %7.1f cycles per iteration = %6.1f per unit (16 v_pk_fma_f32)\n", name, U, nw,
           (double)mx / iters, (double)mx / iters / U);
  }
}

int main() {
  f32x4* d_w; float* d_out; long long* d_cyc;
  hipMalloc(&d_w, 1 << 22); hipMalloc(&d_out, 1 << 22); hipMalloc(&d_cyc, 1 << 20);
  std::vector<float> hw(1 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f) - 0.5f;
  hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  run<0, 1>(d_w, d_out, d_cyc, "activations by v_readlane_b32 -> SGPR pairs");
  run<0, 2>(d_w, d_out, d_cyc, "activations by v_readlane_b32 -> SGPR pairs");
  run<1, 1>(d_w, d_out, d_cyc, "activations by broadcast ds_read_b128 (no lead)");
  run<1, 2>(d_w, d_out, d_cyc, "activations by broadcast ds_read_b128 (no lead)");
  return 0;
}
