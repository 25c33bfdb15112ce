// Micro-benchmark: what one matrix-VECTOR multiply-add costs on the gfx950 VALU, form by form (cycles per wave-instruction
// per SIMD, one and two waves per SIMD, one workgroup per CU).  Round 5's question: the one-row rollout kernel issues
// v_fmac_f32_dpp row_newbcast (64 MACs, 5.08 cycles measured) -- is a PACKED form (v_pk_fma_f32: 128 MACs per instruction,
// the activation pair replicated in every lane, read from LDS as a broadcast) faster end to end, including what feeds it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/_bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DPP(J, A, W) "v_fmac_f32_dpp %" #A ", %4, %" #W " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define DPP8(J, A, W) "v_fmac_f32_dpp %" #A ", %8, %" #W " row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define FMA(A, W) "v_fmac_f32 %" #A ", %4, %" #W "\n\t"
#define PK(A, W, X) "v_pk_fma_f32 %" #A ", %" #W ", %" #X ", %" #A "\n\t"
#define PKB(A, W, X) "v_pk_fma_f32 %" #A ", %" #W ", %" #X ", %" #A " op_sel_hi:[1,0,1]\n\t"

enum { V_DPP2, V_DPP4, V_DPP8, V_FMAC, V_FMAC_S, V_PK4, V_PK4_S, V_PK4_B, V_PK2, V_PK8, V_PK_XB4, V_PK_XB8, V_PK_WL2, V_PK_WL4, V_PK_WL8, V_PK_XB8_WL4, V_N };
static const char* names[] = {"fmac_dpp row_newbcast, 2 accumulators", "fmac_dpp row_newbcast, 4 accumulators", "fmac_dpp row_newbcast, 8 accumulators", "v_fmac_f32 (VGPR x)", "v_fmac_f32 (SGPR x)",
                              "v_pk_fma_f32, 4 acc pairs", "v_pk_fma_f32, SGPR x pair", "v_pk_fma_f32 op_sel lo-broadcast x", "v_pk_fma_f32, 2 acc pairs", "v_pk_fma_f32, 8 acc pairs",
                              "pk + 1 broadcast ds_read_b128 (x) per 4", "pk + 1 broadcast ds_read_b128 (x) per 8", "pk + 1 per-lane ds_read_b128 (W) per 2 [all W from LDS]",
                              "pk + 1 per-lane ds_read_b128 (W) per 4 [half]", "pk + 1 per-lane ds_read_b128 (W) per 8 [quarter]", "pk + x broadcast per 8 + W per 4"};

// one "group" = 16 instructions of the form under test
template <int V>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ wimg, int iters, float* out, long long* cyc, int nw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = wimg[i] * 1e-3f;
  __syncthreads();
  float w[16];
  f32x2 w2[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    w[r] = wimg[(wave * 16 + r) * 64 + lane] * 1e-3f;
    w2[r] = f32x2{wimg[(wave * 32 + 2 * r) * 64 + lane], wimg[(wave * 32 + 2 * r + 1) * 64 + lane]} * 1e-3f;
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f, x = wimg[lane] * 1e-3f;
  f32x2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f32x2{0.f, 0.f};
  f32x2 x2 = f32x2{wimg[lane], wimg[64 + lane]} * 1e-3f;
  const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
  const float sy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x2.y)));
  const f32x2 sx2 = f32x2{sx, sy};
  const float* xb = lds + 64 * wave;            // broadcast source: one address for the whole wave
  const f32x4* wl = reinterpret_cast<const f32x4*>(lds + 1024) + lane;   // per-lane source: 1 KiB per wave-instruction
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == V_DPP2) {
      asm volatile("s_nop 1\n\t" DPP(0, 0, 5) DPP(1, 1, 6) DPP(2, 0, 7) DPP(3, 1, 8) DPP(4, 0, 9) DPP(5, 1, 10) DPP(6, 0, 11) DPP(7, 1, 12)
                   DPP(8, 0, 13) DPP(9, 1, 14) DPP(10, 0, 15) DPP(11, 1, 16) DPP(12, 0, 17) DPP(13, 1, 18) DPP(14, 0, 19) DPP(15, 1, 20)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    } else if constexpr (V == V_DPP4) {
      asm volatile("s_nop 1\n\t" DPP(0, 0, 5) DPP(1, 1, 6) DPP(2, 2, 7) DPP(3, 3, 8) DPP(4, 0, 9) DPP(5, 1, 10) DPP(6, 2, 11) DPP(7, 3, 12)
                   DPP(8, 0, 13) DPP(9, 1, 14) DPP(10, 2, 15) DPP(11, 3, 16) DPP(12, 0, 17) DPP(13, 1, 18) DPP(14, 2, 19) DPP(15, 3, 20)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    } else if constexpr (V == V_DPP8) {
      asm volatile("s_nop 1\n\t" DPP8(0, 0, 9) DPP8(1, 1, 10) DPP8(2, 2, 11) DPP8(3, 3, 12) DPP8(4, 4, 13) DPP8(5, 5, 14) DPP8(6, 6, 15) DPP8(7, 7, 16)
                   DPP8(8, 0, 17) DPP8(9, 1, 18) DPP8(10, 2, 19) DPP8(11, 3, 20) DPP8(12, 4, 21) DPP8(13, 5, 22) DPP8(14, 6, 23) DPP8(15, 7, 24)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3)
                   : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    } else if constexpr (V == V_FMAC) {
      asm volatile(FMA(0, 5) FMA(1, 6) FMA(2, 7) FMA(3, 8) FMA(0, 9) FMA(1, 10) FMA(2, 11) FMA(3, 12) FMA(0, 13) FMA(1, 14) FMA(2, 15) FMA(3, 16) FMA(0, 17) FMA(1, 18) FMA(2, 19) FMA(3, 20)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "v"(x), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    } else if constexpr (V == V_FMAC_S) {
      asm volatile(FMA(0, 5) FMA(1, 6) FMA(2, 7) FMA(3, 8) FMA(0, 9) FMA(1, 10) FMA(2, 11) FMA(3, 12) FMA(0, 13) FMA(1, 14) FMA(2, 15) FMA(3, 16) FMA(0, 17) FMA(1, 18) FMA(2, 19) FMA(3, 20)
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "s"(sx), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    } else if constexpr (V == V_PK4 || V == V_PK4_S || V == V_PK4_B || V == V_PK2 || V == V_PK8) {
#define PKOPS : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
#define PKW "v"(w2[0]), "v"(w2[1]), "v"(w2[2]), "v"(w2[3]), "v"(w2[4]), "v"(w2[5]), "v"(w2[6]), "v"(w2[7]), "v"(w2[8]), "v"(w2[9]), "v"(w2[10]), "v"(w2[11]), "v"(w2[12]), "v"(w2[13]), "v"(w2[14]), "v"(w2[15])
      // operands: 0..7 acc pairs, 8 x, 9..24 w
      if constexpr (V == V_PK4)
        asm volatile(PK(0, 9, 8) PK(1, 10, 8) PK(2, 11, 8) PK(3, 12, 8) PK(0, 13, 8) PK(1, 14, 8) PK(2, 15, 8) PK(3, 16, 8) PK(0, 17, 8) PK(1, 18, 8) PK(2, 19, 8) PK(3, 20, 8) PK(0, 21, 8) PK(1, 22, 8) PK(2, 23, 8) PK(3, 24, 8) PKOPS : "v"(x2), PKW);
      else if constexpr (V == V_PK4_S)
        asm volatile(PK(0, 9, 8) PK(1, 10, 8) PK(2, 11, 8) PK(3, 12, 8) PK(0, 13, 8) PK(1, 14, 8) PK(2, 15, 8) PK(3, 16, 8) PK(0, 17, 8) PK(1, 18, 8) PK(2, 19, 8) PK(3, 20, 8) PK(0, 21, 8) PK(1, 22, 8) PK(2, 23, 8) PK(3, 24, 8) PKOPS : "s"(sx2), PKW);
      else if constexpr (V == V_PK4_B)
        asm volatile(PKB(0, 9, 8) PKB(1, 10, 8) PKB(2, 11, 8) PKB(3, 12, 8) PKB(0, 13, 8) PKB(1, 14, 8) PKB(2, 15, 8) PKB(3, 16, 8) PKB(0, 17, 8) PKB(1, 18, 8) PKB(2, 19, 8) PKB(3, 20, 8) PKB(0, 21, 8) PKB(1, 22, 8) PKB(2, 23, 8) PKB(3, 24, 8) PKOPS : "v"(x2), PKW);
      else if constexpr (V == V_PK2)
        asm volatile(PK(0, 9, 8) PK(1, 10, 8) PK(0, 11, 8) PK(1, 12, 8) PK(0, 13, 8) PK(1, 14, 8) PK(0, 15, 8) PK(1, 16, 8) PK(0, 17, 8) PK(1, 18, 8) PK(0, 19, 8) PK(1, 20, 8) PK(0, 21, 8) PK(1, 22, 8) PK(0, 23, 8) PK(1, 24, 8) PKOPS : "v"(x2), PKW);
      else
        asm volatile(PK(0, 9, 8) PK(1, 10, 8) PK(2, 11, 8) PK(3, 12, 8) PK(4, 13, 8) PK(5, 14, 8) PK(6, 15, 8) PK(7, 16, 8) PK(0, 17, 8) PK(1, 18, 8) PK(2, 19, 8) PK(3, 20, 8) PK(4, 21, 8) PK(5, 22, 8) PK(6, 23, 8) PK(7, 24, 8) PKOPS : "v"(x2), PKW);
    } else if constexpr (V == V_PK_XB4 || V == V_PK_XB8) {
      // the activations as broadcast LDS reads (every lane the same address): 4 values = 2 pairs per ds_read_b128, requested one group ahead
      constexpr int NR = V == V_PK_XB4 ? 4 : 2;
      f32x4 xq[4];
#pragma unroll
      for (int r = 0; r < NR; ++r) xq[r] = *reinterpret_cast<const f32x4*>(xb + ((it + r) & 15) * 4);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const f32x2 xa = f32x2{xq[r][0], xq[r][1]}, xc = f32x2{xq[r][2], xq[r][3]};
        if constexpr (V == V_PK_XB4) {
          asm volatile(PK(0, 6, 4) PK(1, 7, 4) PK(2, 8, 5) PK(3, 9, 5) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(xa), "v"(xc), "v"(w2[4 * r]), "v"(w2[4 * r + 1]), "v"(w2[4 * r + 2]), "v"(w2[4 * r + 3]));
        } else {
          asm volatile(PK(0, 6, 4) PK(1, 7, 4) PK(2, 8, 4) PK(3, 9, 4) PK(0, 10, 5) PK(1, 11, 5) PK(2, 12, 5) PK(3, 13, 5)
                       : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                       : "v"(xa), "v"(xc), "v"(w2[8 * r]), "v"(w2[8 * r + 1]), "v"(w2[8 * r + 2]), "v"(w2[8 * r + 3]), "v"(w2[8 * r + 4]), "v"(w2[8 * r + 5]), "v"(w2[8 * r + 6]), "v"(w2[8 * r + 7]));
        }
      }
    } else if constexpr (V == V_PK_WL2 || V == V_PK_WL4 || V == V_PK_WL8 || V == V_PK_XB8_WL4) {
      // weights from LDS, per lane: one ds_read_b128 = two weight pairs
      constexpr int NL = V == V_PK_WL2 ? 8 : (V == V_PK_WL4 || V == V_PK_XB8_WL4) ? 4 : 2;
      f32x4 wq[8];
#pragma unroll
      for (int r = 0; r < NL; ++r) wq[r] = wl[((it & 3) * 8 + r) * 64];
      f32x2 xa = x2, xc = x2;
      if constexpr (V == V_PK_XB8_WL4) {
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(xb + (it & 15) * 4);
        xa = f32x2{q0[0], q0[1]};
        xc = f32x2{q0[2], q0[3]};
      }
      f32x2 ww[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ww[r] = w2[r];
#pragma unroll
      for (int r = 0; r < NL; ++r) { ww[2 * r] = f32x2{wq[r][0], wq[r][1]}; ww[2 * r + 1] = f32x2{wq[r][2], wq[r][3]}; }
      asm volatile(PK(0, 6, 4) PK(1, 7, 4) PK(2, 8, 5) PK(3, 9, 5) PK(0, 10, 4) PK(1, 11, 4) PK(2, 12, 5) PK(3, 13, 5) PK(0, 14, 4) PK(1, 15, 4) PK(2, 16, 5) PK(3, 17, 5) PK(0, 18, 4) PK(1, 19, 4) PK(2, 20, 5) PK(3, 21, 5)
                   : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                   : "v"(xa), "v"(xc), "v"(ww[0]), "v"(ww[1]), "v"(ww[2]), "v"(ww[3]), "v"(ww[4]), "v"(ww[5]), "v"(ww[6]), "v"(ww[7]), "v"(ww[8]), "v"(ww[9]), "v"(ww[10]), "v"(ww[11]), "v"(ww[12]), "v"(ww[13]), "v"(ww[14]), "v"(ww[15]));
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * nw + wave] = t1 - t0;
  float s = a0 + a1 + a2 + a3 + e0 + e1 + e2 + e3;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[(size_t)blockIdx.x * 1024 + tid] = s;
}

static float *d_w, *d_out;
static long long* d_cyc;

template <int V>
void run(int nw, double ghz) {
  const int iters = 4000, blocks = 256;
  const size_t lds = 8192 * 4;
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(nw * 64), lds, 0, d_w, 10, d_out, d_cyc, nw);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(nw * 64), lds, 0, d_w, iters, d_out, d_cyc, nw);
  hipEventRecord(e1);
  hipError_t err = hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * nw);
  hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
  long long mx = 0;
  for (auto v : h) mx = v > mx ? v : mx;
  const double per = (double)mx / iters / 16.0 / (nw / 4.0);     // cycles per instruction per SIMD (clock64 ticks at 100 MHz? reported raw)
  const double wall = ms * 1e-3 * ghz * 1e9 / iters / 16.0 / (nw / 4.0);
  const int macs = (V <= V_FMAC_S) ? 64 : 128;
  printf("%-62s %2d waves/CU: %6.2f cyc/instr/SIMD (wall x %.1f GHz) -> %5.1f MACs/clk/SIMD   [clock64 %.2f] %s\n", names[V], nw, wall, ghz, macs / wall, per,
         err == hipSuccess ? "" : hipGetErrorString(err));
}

template <int V>
void both(double ghz) {
  run<V>(4, ghz);
  run<V>(8, ghz);
}

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
  hipMalloc(&d_w, 4 << 20);
  hipMalloc(&d_out, 8 << 20);
  hipMalloc(&d_cyc, 1 << 20);
  std::vector<float> hw(1 << 20);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((i * 2654435761u) >> 8 & 0xffff) * (1.f / 65536.f) - 0.5f;
  hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  both<V_DPP2>(ghz);
  both<V_DPP4>(ghz);
  both<V_DPP8>(ghz);
  run<V_DPP4>(12, ghz); run<V_DPP4>(16, ghz); run<V_DPP8>(12, ghz); run<V_DPP8>(16, ghz); run<V_DPP2>(16, ghz);
  both<V_FMAC>(ghz); run<V_FMAC>(12, ghz); run<V_FMAC>(16, ghz);
  both<V_FMAC_S>(ghz); run<V_FMAC_S>(12, ghz); run<V_FMAC_S>(16, ghz);
  both<V_PK4>(ghz); run<V_PK4>(12, ghz); run<V_PK4>(16, ghz);
  both<V_PK4_S>(ghz); run<V_PK4_S>(12, ghz); run<V_PK4_S>(16, ghz);
  both<V_PK_XB8>(ghz); run<V_PK_XB8>(16, ghz);
  both<V_PK_WL4>(ghz); run<V_PK_WL4>(16, ghz);
  return 0;
}
