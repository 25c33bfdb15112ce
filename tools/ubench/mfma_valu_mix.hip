// Micro-benchmark: VALU work placed BETWEEN the MFMAs of the same wave (v_mfma_f32_16x16x4_f32, 32 cycles of pipe each):
// how many VALU instructions per MFMA are free, with one and with two such waves per SIMD; and whether s_setprio lets a
// VALU-only wave through beside an MFMA-streaming wave (tools/ubench/mfma_valu_overlap.hip: one VALU per MFMA without it).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/_bin/mfma_valu_mix tools/ubench/mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int KIND>   // KIND 0: v_fma_f32 on 4 chains, 1: v_mul_hi_u32 / v_mul_lo / v_xor mix (Philox-like), 4 chains
__global__ __launch_bounds__(512) void mix(int waves_active, int n, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < waves_active) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float a = 1.f + lane, b = 2.f - lane;
    float x[4] = {(float)lane, lane + 1.f, lane + 2.f, lane + 3.f};
    unsigned u[4] = {(unsigned)lane, lane + 1u, lane + 2u, lane + 3u};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[v & 3]));
          else if ((v % 3) == 0) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[v & 3]) : "v"(0xD2511F53u));
          else if ((v % 3) == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[v & 3]) : "v"(0xCD9E8D57u));
          else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[v & 3]) : "v"(0x9E3779B9u));
        }
      }
    }
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    r += x[0] + x[1] + x[2] + x[3] + (float)(u[0] ^ u[1] ^ u[2] ^ u[3]);
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = r;
}

// waves 0..3 stream MFMAs, waves 4..7 a dependent VALU chain at priority PRIO
template <int PRIO>
__global__ __launch_bounds__(512) void prio(int n, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float a = 1.f + lane, b = 2.f - lane;
    for (int it = 0; it < n; ++it)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
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
  const long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = r;
}


// memory instructions between the MFMAs of a wave: per group of GROUP MFMAs (four accumulators in turn), NL global_load_dwordx4 of
// an L2-resident image (KIND 0) or NL ds_read_b128 (KIND 1); every loaded register is consumed one group later
template <int NL, int KIND, int GROUP>
__global__ __launch_bounds__(512) void memmix(int waves_active, int n, const f32x4* __restrict__ buf, float* out, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < 8192; e += 512) lds[e] = e;
  float r = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  if (wave < waves_active) {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float a = 1.f + lane, b = 2.f - lane;
    f32x4 v[NL > 0 ? NL : 1];
    for (int l = 0; l < NL; ++l) v[l] = f32x4{0, 0, 0, 0};
    const f32x4* p = buf + (size_t)wave * 4096 + lane;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        a += v[l][0];                                             // consume last group's data
        if (KIND == 0) v[l] = p[(size_t)((it * NL + l) & 63) * 64];
        else v[l] = *reinterpret_cast<const f32x4*>(lds + ((it * NL + l) & 7) * 1024 + lane * 4 + (wave & 3) * 256);
      }
#pragma unroll
      for (int m = 0; m < GROUP; ++m) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
    }
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = r;
}

template <int NL, int KIND, int GROUP>
void runm(const f32x4* buf, float* out, long long* cyc) {
  const int n = 2000;
  long long h[8];
  double res[2];
  for (int m = 0; m < 2; ++m) {
    const int wa = m ? 8 : 4;
    hipLaunchKernelGGL((memmix<NL, KIND, GROUP>), dim3(1), dim3(512), 0, 0, wa, n, buf, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < wa; ++w) mx = h[w] > mx ? h[w] : mx;
    res[m] = (double)mx / ((double)GROUP * n) / (m ? 2 : 1);
  }
  printf("%d %s per %d MFMAs in the same wave: %5.1f cycles per MFMA per SIMD with one wave per SIMD, %5.1f with two\n", NL,
         KIND ? "ds_read_b128" : "global_load_dwordx4", GROUP, res[0], res[1]);
}

template <int NV, int KIND>
void run(float* out, long long* cyc) {
  const int n = 2000;
  long long h[8];
  double res[2];
  for (int m = 0; m < 2; ++m) {
    const int wa = m ? 8 : 4;
    hipLaunchKernelGGL((mix<NV, KIND>), dim3(1), dim3(512), 0, 0, wa, n, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < wa; ++w) mx = h[w] > mx ? h[w] : mx;
    res[m] = (double)mx / (4.0 * n) / (m ? 2 : 1);     // per MFMA per SIMD
  }
  printf("%s x%d per MFMA in the same wave: %5.1f cycles per MFMA per SIMD with one wave per SIMD, %5.1f with two\n",
         KIND ? "integer (mul_hi / mul_lo / xor)" : "v_fma_f32", NV, res[0], res[1]);
}

template <int PRIO>
void runp(float* out, long long* cyc) {
  const int n = 4000;
  long long h[8];
  hipLaunchKernelGGL((prio<PRIO>), dim3(1), dim3(512), 0, 0, n, out, cyc);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("VALU wave at s_setprio %d beside an MFMA-streaming wave: MFMA %5.1f, v_fma_f32 %5.1f cycles per instruction\n", PRIO,
         h[0] / (8.0 * n), h[4] / (8.0 * n));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  run<0, 0>(out, cyc);
  run<2, 0>(out, cyc);
  run<4, 0>(out, cyc);
  run<6, 0>(out, cyc);
  run<8, 0>(out, cyc);
  run<2, 1>(out, cyc);
  run<4, 1>(out, cyc);
  run<6, 1>(out, cyc);
  f32x4* buf; hipMalloc(&buf, 8 * 4096 * 16 * 64 / 64 * 16); hipMemset(buf, 0, 8 * 4096 * 16);
  runm<0, 0, 8>(buf, out, cyc);
  runm<1, 0, 8>(buf, out, cyc);
  runm<2, 0, 8>(buf, out, cyc);
  runm<4, 0, 8>(buf, out, cyc);
  runm<1, 0, 4>(buf, out, cyc);
  runm<1, 1, 8>(buf, out, cyc);
  runm<2, 1, 8>(buf, out, cyc);
  runm<4, 1, 8>(buf, out, cyc);
  runm<2, 1, 4>(buf, out, cyc);
  runp<0>(out, cyc);
  runp<3>(out, cyc);
  return 0;
}
