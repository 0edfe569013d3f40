// Microbenchmarks that size the NBMF pass kernel on gfx950: fp64 MFMA issue rate, fp64 VALU
// (fma / divide / log) rates, their overlap, and an exact-integer check of the f64 MFMA
// operand/accumulator lane maps including "accumulator as the next B operand" chaining.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// VALU kernels: OP 0 = fma, 1 = IEEE divide, 2 = fast reciprocal-divide, 3 = log, 4 = fast log candidate
__device__ __forceinline__ double fast_div(double a, double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = a * r;
  double res = __builtin_fma(-d, q, a);
  return __builtin_fma(res, r, q);
}

template <int OP>
__global__ __launch_bounds__(256) void k_valu(double* out, int iters, double x0) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = x0 + 1e-3 * i + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) x[i] = __builtin_fma(x[i], 0.999999, 1e-7);
      if (OP == 1) x[i] = 0.7 / (x[i] + 1e-8);
      if (OP == 2) x[i] = fast_div(0.7, x[i] + 1e-8);
      if (OP == 3) x[i] = 1.5 + 0.1 * log(x[i] + 1e-8);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Mixed: waves 0-3 of a 512-thread WG do MFMA, waves 4-7 do divides -> checks pipe overlap on one SIMD.
__global__ __launch_bounds__(512) void k_mixed(double* out, int iters, double a0, int mode) {
  int wave = threadIdx.x >> 6;
  double s = 0;
  bool do_mfma = (mode == 0) || (mode == 2 && wave < 4);
  bool do_div = (mode == 1) || (mode == 2 && wave >= 4);
  if (do_mfma) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = a0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (do_div) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + 1e-3 * i + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fast_div(0.7, x[i] + 1e-8);
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Layout check. T[i][j] = sum_k A[i][k] B[k][j] over K=8 (two MFMA steps); then G[p][j] = sum_i L[p][i] T[i][j]
// using T's accumulator registers directly as the B operand of step r (rows 4r..4r+3).
__global__ void k_layout(const double* A, const double* B, const double* L, double* T, double* G) {
  int l = threadIdx.x, q = l >> 4, c = l & 15;
  d4 t = {0, 0, 0, 0};
  for (int s = 0; s < 2; ++s) {
    double a = A[c * 8 + 4 * s + q];   // A[row c][k = 4s+q]
    double b = B[(4 * s + q) * 16 + c]; // B[k = 4s+q][col c]
    t = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) T[(q + 4 * r) * 16 + c] = t[r];
  d4 g = {0, 0, 0, 0};
  for (int r = 0; r < 4; ++r) {
    double a = L[c * 16 + 4 * r + q];  // L[row c][i = 4r+q]
    g = __builtin_amdgcn_mfma_f64_16x16x4f64(a, t[r], g, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) G[(q + 4 * r) * 16 + c] = g[r];
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  double* out; CK(hipMalloc(&out, sizeof(double) * 4096 * 512));

  // ---- layout check
  {
    std::vector<double> A(16 * 8), B(8 * 16), L(16 * 16), T(256), G(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 8; ++k) A[i * 8 + k] = (i * 3 + k * 7) % 11 - 5;
    for (int k = 0; k < 8; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (k * 5 + j * 2 + 1) % 13 - 6;
    for (int pp = 0; pp < 16; ++pp) for (int i = 0; i < 16; ++i) L[pp * 16 + i] = (pp * 7 + i * 3 + 2) % 9 - 4;
    double *dA, *dB, *dL, *dT, *dG;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dL, L.size() * 8));
    CK(hipMalloc(&dT, 256 * 8)); CK(hipMalloc(&dG, 256 * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dL, dT, dG);
    CK(hipMemcpy(T.data(), dT, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(G.data(), dG, 256 * 8, hipMemcpyDeviceToHost));
    int badT = 0, badG = 0;
    std::vector<double> Tr(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0; for (int k = 0; k < 8; ++k) s += A[i * 8 + k] * B[k * 16 + j];
      Tr[i * 16 + j] = s; if (s != T[i * 16 + j]) ++badT;
    }
    for (int pp = 0; pp < 16; ++pp) for (int j = 0; j < 16; ++j) {
      double s = 0; for (int i = 0; i < 16; ++i) s += L[pp * 16 + i] * Tr[i * 16 + j];
      if (s != G[pp * 16 + j]) ++badG;
    }
    printf("LAYOUT theta_mismatch %d chain_mismatch %d (0 0 expected)\n", badT, badG);
  }

  const int CU = p.multiProcessorCount;
  // ---- MFMA rate: blocks of 256 threads (1 wave/SIMD) x CU, 1 to 4 blocks per CU
  for (int bpc = 1; bpc <= 4; ++bpc) {
    int iters = 2000;
    {
      float ms = time_ms([&] { k_mfma<1><<<CU * bpc, 256>>>(out, iters, 1.0, 0.5); });
      double fl = (double)CU * bpc * 4 * iters * 8 * 1 * 2048.0;
      printf("MFMA f64 16x16x4 nacc=1 blocks/CU=%d : %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, fl / ms / 1e9,
             ms * 1e-3 * 2.4e9 / (iters * 8.0 * 1 * bpc));
    }
    {
      float ms = time_ms([&] { k_mfma<4><<<CU * bpc, 256>>>(out, iters, 1.0, 0.5); });
      double fl = (double)CU * bpc * 4 * iters * 8 * 4 * 2048.0;
      printf("MFMA f64 16x16x4 nacc=4 blocks/CU=%d : %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, fl / ms / 1e9,
             ms * 1e-3 * 2.4e9 / (iters * 8.0 * 4 * bpc));
    }
    {
      float ms = time_ms([&] { k_mfma<16><<<CU * bpc, 256>>>(out, iters, 1.0, 0.5); });
      double fl = (double)CU * bpc * 4 * iters * 8 * 16 * 2048.0;
      printf("MFMA f64 16x16x4 nacc=16 blocks/CU=%d : %.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, fl / ms / 1e9,
             ms * 1e-3 * 2.4e9 / (iters * 8.0 * 16 * bpc));
    }
  }
  // ---- VALU rates, 8 waves/SIMD-ish occupancy: 8 blocks of 256 per CU
  {
    int iters = 500;
    const char* names[4] = {"fma", "ieee_div", "fast_div", "log"};
    for (int wps = 1; wps <= 4; wps *= 2) {
      float ms[4];
      ms[0] = time_ms([&] { k_valu<0><<<CU * wps, 256>>>(out, iters, 0.3); });
      ms[1] = time_ms([&] { k_valu<1><<<CU * wps, 256>>>(out, iters, 0.3); });
      ms[2] = time_ms([&] { k_valu<2><<<CU * wps, 256>>>(out, iters, 0.3); });
      ms[3] = time_ms([&] { k_valu<3><<<CU * wps, 256>>>(out, iters, 0.3); });
      for (int o = 0; o < 4; ++o) {
        double ops_per_simd = (double)iters * 8 * wps;  // wave-ops per SIMD
        printf("VALU %-8s waves/SIMD=%d : %.1f cyc per wave-op per SIMD @2.4GHz, %.2f Tops/s\n", names[o], wps,
               ms[o] * 1e-3 * 2.4e9 / ops_per_simd, (double)CU * 4 * 64 * ops_per_simd / ms[o] / 1e9);
      }
    }
  }
  // ---- overlap
  {
    int iters = 1000;
    float m0 = time_ms([&] { k_mixed<<<CU, 512>>>(out, iters, 0.4, 0); });
    float m1 = time_ms([&] { k_mixed<<<CU, 512>>>(out, iters, 0.4, 1); });
    float m2 = time_ms([&] { k_mixed<<<CU, 512>>>(out, iters, 0.4, 2); });
    printf("MIXED all-8-waves-mfma %.3f ms, all-8-waves-div %.3f ms, 4 mfma + 4 div %.3f ms (half-work each alone: %.3f / %.3f)\n", m0, m1, m2, m0 / 2, m1 / 2);
  }
  return 0;
}
