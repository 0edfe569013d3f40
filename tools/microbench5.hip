// Round 6: WHY a vector instruction costs 6-8 cycles inside the sweeps against 4.1 in microbench4's straight runs.
// microbench4's fillers were INDEPENDENT of each other.  The ratio block is not: its reciprocal is a serial chain
// (product -> v_rcp_f64 -> three Newton FMAs -> two multiplications), its logarithm another.  Hypothesis: a dependent f64
// vector instruction cannot issue right behind its producer; with other waves resident the SIMD's arbiter puts one of
// THEIR f64 MFMAs into the gap, and the chain's next instruction then pays the MFMA -> vector turnaround (+9.4 cycles)
// again -- once per link of the chain instead of once per run.
// Loop body per wave: 8 MFMAs on 4 accumulators, then a block of NV v_fma_f64 whose dependency distance is DIST
// (DIST = 1: one serial chain; 2, 4: that many interleaved chains; 8: independent within any window of 8), at 1, 2, 3
// waves per SIMD, without and with s_setprio 1 around the block.  Printed: cycles per trip beyond the MFMAs', per vector
// instruction.  Variant READACC: the block's first four instructions read the accumulators the MFMAs just wrote (as the
// ratio block reads Theta); variant FEED: the MFMAs' B operand is the block's last result (as the back-products read R1).
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench5.hip -o build/microbench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, int DIST, int WPS, bool PRIO, int MODE /* 0 plain, 1 READACC, 2 FEED, 3 RCP chain as in the sweeps */>
__global__ __launch_bounds__(256, (WPS < 2 ? 2 : WPS)) void k_chain(double* out, int iters, double a0, double b0) {
  // (at most 256 registers per lane in every variant, so that hipcc keeps the accumulators in VGPRs -- the form the sweeps
  //  have; with AGPR accumulators the bare loop issues ~8 % slower.  One wave per SIMD is enforced by dynamic LDS instead.)
  extern __shared__ double lds_pad[];
  if (iters < 0) lds_pad[threadIdx.x] = a0;
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double x[8], y = 0.999999, z = 1e-9;
  for (int i = 0; i < 8; ++i) x[i] = a0 + 1e-3 * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (MODE == 2) ? x[u & 3] : b, acc[u & 3], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) __asm__ volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[r]) : "v"(acc[3][r]), "v"(z));
    }
    if (MODE == 3) {
      // the byte-code path's ratio block in outline: 4 x (select, difference) side by side, the batch reciprocal's serial
      // chain, 4 x 2 products side by side: 8 + 13 + 8 = 29 instructions of which 7 hang on their predecessor
      double d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) __asm__ volatile("v_add_f64 %0, %1, %2" : "=v"(d[r]) : "v"(acc[3][r]), "v"(y));
#pragma unroll
      for (int r = 0; r < 4; ++r) __asm__ volatile("v_add_f64 %0, |%0|, %1" : "+v"(d[r]) : "v"(z));
      double p01, p23, pp, r0, e, ip, i01, i23;
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(p01) : "v"(d[0]), "v"(d[1]));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(p23) : "v"(d[2]), "v"(d[3]));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(pp) : "v"(p01), "v"(p23));
      __asm__ volatile("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(pp));
      __asm__ volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(pp), "v"(r0));
      __asm__ volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(e));
      __asm__ volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(ip) : "v"(r0), "v"(e));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(i01) : "v"(ip), "v"(p23));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(i23) : "v"(ip), "v"(p01));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(x[0]) : "v"(i01), "v"(d[1]));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(x[1]) : "v"(i01), "v"(d[0]));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(x[2]) : "v"(i23), "v"(d[3]));
      __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(x[3]) : "v"(i23), "v"(d[2]));
#pragma unroll
      for (int r = 0; r < 4; ++r) __asm__ volatile("v_mul_f64 %0, %1, %2" : "=v"(x[4 + r]) : "v"(x[r]), "v"(y));
#pragma unroll
      for (int r = 0; r < 4; ++r) __asm__ volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(z) : "v"(x[4 + r]), "v"(y));
      z *= 1e-30;
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v) __asm__ volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[v % DIST]) : "v"(y));
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = z;
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
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return ms;
}

int g_cu = 256, g_iters = 20000;
double* g_out = nullptr;

template <int NV, int DIST, int WPS, bool PRIO, int MODE>
double trip_cycles() {
  const size_t lds = WPS == 1 ? 100 * 1024 : 0;
  if (lds) CK(hipFuncSetAttribute((const void*)k_chain<NV, DIST, WPS, PRIO, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const float ms = time_ms([&] { k_chain<NV, DIST, WPS, PRIO, MODE><<<g_cu * WPS, 256, lds>>>(g_out, g_iters, 1.0, 0.5); });
  return ms * 1e-3 * 2.4e9 / ((double)g_iters * WPS);   // SIMD cycles per trip of ONE wave (the WPS waves of a SIMD share it)
}
template <int WPS, bool PRIO>
void row() {
  const double base = trip_cycles<0, 1, WPS, false, 0>();
  const double t1 = trip_cycles<32, 1, WPS, PRIO, 0>(), t2 = trip_cycles<32, 2, WPS, PRIO, 0>(), t4 = trip_cycles<32, 4, WPS, PRIO, 0>(),
               t8 = trip_cycles<32, 8, WPS, PRIO, 0>();
  const double ra = trip_cycles<32, 8, WPS, PRIO, 1>(), fd = trip_cycles<32, 8, WPS, PRIO, 2>(), rc = trip_cycles<0, 1, WPS, PRIO, 3>();
  printf("%d wave(s) per SIMD%s: 8 MFMAs alone %.1f cycles per trip (%.1f per MFMA); + 32 v_fma_f64, per instruction: serial chain %.2f | two chains %.2f | "
         "four chains %.2f | independent %.2f | independent, first four read the fresh accumulators %.2f | independent, MFMAs read the block's results %.2f ;  "
         "the sweeps' 29-instruction ratio block: +%.1f cycles per trip = %.2f per instruction\n",
         WPS, PRIO ? ", s_setprio 1 around the block" : "", base, base / 8, (t1 - base) / 32, (t2 - base) / 32, (t4 - base) / 32, (t8 - base) / 32,
         (ra - base) / 36, (fd - base) / 32, rc - base, (rc - base) / 29);
  fflush(stdout);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  g_cu = p.multiProcessorCount;
  g_iters = argc > 1 ? atoi(argv[1]) : 20000;
  printf("%d CUs, %d trips of 8 MFMAs + a vector block; cycles at the nominal 2.4 GHz\n", g_cu, g_iters);
  CK(hipMalloc(&g_out, sizeof(double) * g_cu * 4 * 256));
  row<1, false>();
  row<2, false>();
  row<3, false>();
  row<4, false>();
  row<2, true>();
  row<3, true>();
  return 0;
}
