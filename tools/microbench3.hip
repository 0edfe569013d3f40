// Does a VALU instruction placed BETWEEN the f64 MFMAs of a wave cost its issue time, or does it ride in the ~6 cycles
// by which a loop of bare MFMAs falls short of one per 64 (tools/microbench.hip: ~70 cycles per MFMA)?
// Loop body: 8 MFMAs on 4 independent accumulators, NV independent v_fma_f64 spread among them (NV = 0, 4, 8, 16, 32).
// Also the same VALU work issued by OTHER waves of the SIMD (MFMA waves and FMA waves side by side).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench3.hip -o build/microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a0, double b0) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 + 1e-3 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV / 8; ++v) x[(u + v) & 7] = __builtin_fma(x[(u + v) & 7], 0.999999, 1e-7);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0..3 of a 512-thread workgroup (one per SIMD) issue MFMAs, waves 4..7 issue NV FMAs per 8 MFMAs of their neighbour
template <int NV>
__global__ __launch_bounds__(512) void k_side(double* out, int iters, double a0, double b0) {
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 + 1e-3 * i;
  if (threadIdx.x < 256) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int v = 0; v < NV; ++v) x[v & 7] = __builtin_fma(x[v & 7], 0.999999, 1e-7);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float time_ms(F f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int CU = p.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;   // (8 MFMAs each: 4000 = ~1 ms per launch; 400000 = ~90 ms, long enough for the power governor)
  printf("%d iterations of 8 MFMAs per launch\n", iters);
  double* out;
  CK(hipMalloc(&out, sizeof(double) * CU * 4 * 512));
#define RUN(NV, BPC)                                                                                                         \
  {                                                                                                                           \
    float ms = time_ms([&] { k_mix<NV><<<CU * BPC, 256>>>(out, iters, 1.0, 0.5); });                                          \
    printf("same wave : %2d FMAs per 8 MFMAs, %d waves/SIMD : %6.1f cycles per MFMA per SIMD @2.4 GHz\n", NV, BPC,          \
           ms * 1e-3 * 2.4e9 / (iters * 8.0 * BPC));                                                                          \
  }
  if (argc > 2) {   // (the same-wave form: hipcc moves the accumulators between AGPRs and VGPRs around the FMAs -- not a clean test)
    RUN(0, 1) RUN(8, 1) RUN(16, 1) RUN(32, 1)
    RUN(0, 2) RUN(8, 2) RUN(16, 2) RUN(32, 2)
    RUN(0, 3) RUN(8, 3) RUN(16, 3) RUN(32, 3) RUN(64, 3)
  }
#define SIDE(NV, BPC)                                                                                                        \
  {                                                                                                                           \
    float ms = time_ms([&] { k_side<NV><<<CU * BPC, 512>>>(out, iters, 1.0, 0.5); });                                         \
    printf("other wave: %2d FMAs per 8 MFMAs, %d MFMA + %d FMA waves/SIMD : %6.1f cycles per MFMA per SIMD @2.4 GHz\n", NV, \
           BPC, BPC, ms * 1e-3 * 2.4e9 / (iters * 8.0 * BPC));                                                                \
  }
  SIDE(0, 1) SIDE(8, 1) SIDE(16, 1) SIDE(32, 1)
  SIDE(0, 2) SIDE(8, 2) SIDE(16, 2) SIDE(32, 2)
  SIDE(0, 1) SIDE(0, 2) SIDE(0, 1) SIDE(0, 2)
  return 0;
}
