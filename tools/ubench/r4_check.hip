// Check of the 4-row tile arithmetic: one 16-neuron x 16-input fragment times a 4-row activation tile with
// v_mfma_f32_4x4x1_16b_f32 + the k-group sum (permlane16/32 swaps), against the CPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float kg_reduce(const f32x4 v) {      // the butterfly of socmx_unet.h: lane l keeps component {0,2,1,3}[l >> 4]
  float a = v[0], b = v[1], c = v[2], d = v[3];
  asm volatile(
      "s_nop 7\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_permlane32_swap_b32 %2, %3\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_add_f32 %2, %2, %3\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_add_f32 %0, %0, %2"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  return a;
}
__device__ __forceinline__ int kg_comp(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }
__global__ void k(const float* W /*16x16*/, const float* X /*4x16*/, float* Y /*4x16*/, float* S /*64*/, float* R /*64x4 raw*/) {
  const int l = threadIdx.x;
  f32x4 w, x;
  for (int i = 0; i < 4; ++i) { w[i] = W[(l & 15) * 16 + 4 * (l >> 4) + i]; x[i] = X[(l & 3) * 16 + 4 * (l >> 4) + i]; }
  f32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w[i], x[i], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) R[l * 4 + i] = acc[i];
  Y[(l & 3) * 16 + 4 * ((l >> 2) & 3) + kg_comp(l)] = kg_reduce(acc);
  S[l] = kg_reduce(f32x4{(float)l, 100.f + l, 200.f + l, 300.f + l});
}
int main() {
  float hW[256], hX[64], hY[64], hS[64], hR[256], *W, *X, *Y, *S, *R;
  for (int i = 0; i < 256; ++i) hW[i] = sinf(i * 0.37f);
  for (int i = 0; i < 64; ++i) hX[i] = cosf(i * 0.11f);
  hipMalloc(&W, 1024); hipMalloc(&X, 256); hipMalloc(&Y, 256); hipMalloc(&S, 256); hipMalloc(&R, 1024);
  hipMemcpy(W, hW, 1024, hipMemcpyHostToDevice); hipMemcpy(X, hX, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, W, X, Y, S, R);
  hipMemcpy(hY, Y, 256, hipMemcpyDeviceToHost); hipMemcpy(hS, S, 256, hipMemcpyDeviceToHost); hipMemcpy(hR, R, 1024, hipMemcpyDeviceToHost);
  double err = 0;
  for (int r = 0; r < 4; ++r) for (int n = 0; n < 16; ++n) {
    double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hW[n * 16 + kk] * hX[r * 16 + kk];
    err = fmax(err, fabs(s - hY[r * 16 + n]));
  }
  printf("max |Y - ref| = %.3e\n", err);
  // lane l keeps component c = {0,2,1,3}[l >> 4] of (l, 100 + l, 200 + l, 300 + l) summed over l, l^16, l^32, l^48
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int c = ((l >> 4) & 1) * 2 + (l >> 5);
    const float want = 4.f * (100.f * c + (l & 15)) + 96.f;
    if (hS[l] != want) ++bad;
  }
  printf("kg_reduce on lane ids: %d of 64 lanes wrong\n", bad);
  // raw check: lane l reg i should be sum over its kg of W[4ng+i][4kg+m] x[j][4kg+m]
  double e2 = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    const int kg = l >> 4, ng = (l >> 2) & 3, j = l & 3;
    double s = 0; for (int m = 0; m < 4; ++m) s += (double)hW[(4 * ng + i) * 16 + 4 * kg + m] * hX[j * 16 + 4 * kg + m];
    e2 = fmax(e2, fabs(s - hR[l * 4 + i]));
  }
  printf("raw max err = %.3e\n", e2);
  return 0;
}
