// Which VALU / memory ops contend with the fp64 MFMA pipe on gfx950?
// A 512-thread WG per CU: waves 0-3 (one per SIMD) issue back-to-back f64 MFMAs, waves 4-7 run OP.
// Reported: MFMA-only time, OP-only time, both together. together ~= max => overlap; ~= sum => shared pipe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ __launch_bounds__(512) void k(double* out, const double* in, int iters, double a0, int mode) {
  __shared__ double lds[4096];
  int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = a0 + i * 1e-6;
  __syncthreads();
  double s = 0;
  bool do_mfma = (mode == 0 && wave < 4) || (mode == 2 && wave < 4);
  bool do_op = (mode == 1 && wave >= 4) || (mode == 2 && wave >= 4);
  if (do_mfma) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = a0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (do_op) {
    double x[8]; float f[8]; int n[8];
    for (int i = 0; i < 8; ++i) { x[i] = a0 + 1e-3 * i + threadIdx.x * 1e-6; f[i] = (float)x[i]; n[i] = i + threadIdx.x; }
    const int R = 16 * iters / 8;  // each outer iter: 8 x 8 ops = 64 ops
    for (int it = 0; it < R; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (OP == 0) x[i] = __builtin_fma(x[i], 0.999999, 1e-7);
          if (OP == 1) x[i] = x[i] * 0.999999;
          if (OP == 2) x[i] = x[i] + 1e-7;
          if (OP == 3) x[i] = __builtin_amdgcn_rcp(x[i]);
          if (OP == 4) f[i] = __builtin_fmaf(f[i], 0.99999f, 1e-7f);
          if (OP == 5) f[i] = __builtin_amdgcn_rcpf(f[i]);
          if (OP == 6) n[i] = n[i] * 3 + (n[i] >> 2);
          if (OP == 7) { f[i] = (float)x[i]; x[i] = (double)f[i] + 0.0; asm volatile("" : "+v"(x[i])); }
          if (OP == 8) { int e; x[i] = frexp(x[i], &e) + 0.25; n[i] += e; }
          if (OP == 9) { x[i] = (n[i] & 1) ? x[i] : x[(i + 1) & 7]; n[i] += it; }
          if (OP == 10) { x[i] += lds[(threadIdx.x + 64 * i + it * 8 + u) & 4095]; }
          if (OP == 11) { x[i] += in[((size_t)blockIdx.x * 512 + threadIdx.x + 512 * (size_t)(i + 8 * (u + 8 * (it & 63)))) & ((1u << 24) - 1)]; }
        }
    }
    for (int i = 0; i < 8; ++i) s += x[i] + f[i] + n[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> static float time_ms(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
  return best;
}

template <int OP> void run(const char* name, double* out, const double* in, int CU) {
  int iters = 1000;
  float m0 = time_ms([&] { k<OP><<<CU, 512>>>(out, in, iters, 0.4, 0); });
  float m1 = time_ms([&] { k<OP><<<CU, 512>>>(out, in, iters, 0.4, 1); });
  float m2 = time_ms([&] { k<OP><<<CU, 512>>>(out, in, iters, 0.4, 2); });
  double ops = 16.0 * iters / 8 * 64;  // wave-ops per OP wave
  printf("%-10s mfma %.3f ms | op %.3f ms (%.1f cyc/wave-op@2.4GHz) | both %.3f ms | overlap %.0f%%\n", name, m0, m1, m1 * 1e-3 * 2.4e9 / ops, m2,
         100.0 * (m0 + m1 - m2) / (m0 < m1 ? m0 : m1));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int CU = p.multiProcessorCount;
  double *out, *in; CK(hipMalloc(&out, 8 * 512 * 1024)); CK(hipMalloc(&in, 8ull << 24)); CK(hipMemset(in, 0, 8ull << 24));
  run<0>("f64_fma", out, in, CU);
  run<1>("f64_mul", out, in, CU);
  run<2>("f64_add", out, in, CU);
  run<3>("f64_rcp", out, in, CU);
  run<4>("f32_fma", out, in, CU);
  run<5>("f32_rcp", out, in, CU);
  run<6>("i32_ops", out, in, CU);
  run<7>("cvt_rt", out, in, CU);
  run<8>("frexp", out, in, CU);
  run<9>("cndmask64", out, in, CU);
  run<10>("lds_b64", out, in, CU);
  run<11>("gload_b64", out, in, CU);
  return 0;
}
