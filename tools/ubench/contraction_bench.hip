// Stand-alone timing of the SOCM contraction kernels at one shape (developer tool):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include [-DSOCMX_CONTRACTION_PROF] -o /tmp/cb tools/ubench/contraction_bench.hip
//   /tmp/cb 64 400 512        (d K B [repetitions]; -DSOCMX_CONTRACTION_PROF adds the forward kernel's in-kernel cycle counters)
// Includes the product translation unit, so it times exactly the shipped kernels through the C ABI entry points.
#include "../../soc-matching_amd/csrc/socmx_loss.hip"
#include <cstdio>
#include <vector>

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
  float* p; hipMalloc(&p, n * sizeof(float)); hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int d = argc > 1 ? atoi(argv[1]) : 64, K = argc > 2 ? atoi(argv[2]) : 400, B = argc > 3 ? atoi(argv[3]) : 512;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const size_t np = (size_t)(K + 1) * (K + 2) / 2, dd = (size_t)d * d;
  float *net = dev_rand(np * dd, 0.1f, 1), *dnet = dev_rand(np * dd, 0.1f, 2), *delta = dev_rand(np, 0.5f, 3);
  float *q = dev_rand((size_t)K * B * d, 1.f, 4), *v = dev_rand((size_t)K * B * d, 1.f, 5), *gT = dev_rand((size_t)B * d, 1.f, 6);
  float *nV = dev_rand((size_t)(K + 1) * B * d, 1.f, 7), *w = dev_rand(B, 0.2f, 8), *sig = dev_rand(dd, 0.1f, 9);
  float *target, *G, *obj, *gM, *gdM, *ggp, *gam, *ows;
  hipMalloc(&target, (size_t)(K + 1) * B * d * 4); hipMalloc(&G, (size_t)(K + 1) * B * d * 4); hipMalloc(&obj, 4);
  hipMalloc(&gM, np * dd * 4); hipMalloc(&gdM, np * dd * 4);
  { const size_t nws = (size_t)socmx_socm_objective_workspace_floats(K, B); hipMalloc(&ows, nws * 4); hipMemset(ows, 0, nws * 4); }
  const int nb = (d + 15) / 16;
  hipMalloc(&ggp, np * nb * nb * 4); hipMalloc(&gam, 4);
  const float g1 = 2.f; hipMemcpy(gam, &g1, 4, hipMemcpyHostToDevice);
  socmx_problem pb = {}; pb.kind = 1; pb.d = d; pb.sigma = sig;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double fl = 4.0 * B * dd * np;
  for (int pass = 0; pass < 2; ++pass) {
    float ms = 0.f;
    for (int r = 0; r <= reps; ++r) {
      if (r == 1) hipEventRecord(e0, 0);
      int st = pass == 0 ? socmx_socm_target_fwd_net_f32(&pb, K, B, net, dnet, delta, gam, q, v, gT, nV, w, 1e-3f, target, G, obj, ows, 0)
                         : socmx_socm_target_bwd_net_f32(d, K, B, G, q, v, gT, nullptr, net, dnet, delta, gam, gM, gdM, ggp, 0);
      if (st) { printf("status %d\n", st); return 1; }
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%s d=%d K=%d B=%d: %.3f ms  %.1f TFLOP/s\n", pass == 0 ? "fwd(+residual)" : "bwd", d, K, B, ms / reps, fl / (ms / reps) / 1e9);
#ifdef SOCMX_CONTRACTION_PROF
    if (pass == 0) {      // d % 4 == 0, 16 < d <= 64, B >= 256: socm_target_lds4_kernel's counters (workgroup 0 = the longest row)
      long long h[8][4];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(socmx::g_contraction_prof), sizeof(h));
      for (int w2 = 0; w2 < 8; ++w2)
        printf("  wave %d: cycles per iteration: barrier %6.0f  operand wait %6.0f  multiply (64 MFMAs = 2048) %6.0f   (%lld iterations)\n", w2,
               (double)h[w2][0] / h[w2][3], (double)h[w2][1] / h[w2][3], (double)h[w2][2] / h[w2][3], h[w2][3]);
    }
#endif
  }
  return 0;
}
