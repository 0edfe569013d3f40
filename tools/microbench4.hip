// What ONE instruction of each kind costs beside v_mfma_f64_16x16x4_f64 on gfx950: the f64 MFMA runs on the SIMD's
// double-precision lanes for 64 cycles, and tools/microbench3.hip showed that a v_fma_f64 of another wave adds ~4.5
// cycles to it.  Round 5 asks the same of every instruction kind the general path's ratio block is made of: 64-bit
// arithmetic, the quarter-rate operations (v_rcp_f64; are v_frexp_* and v_cvt_f64_i32 among them?), 32-bit integer
// operations (could they ride beside the MFMA?), LDS gathers and scalar instructions.
// Loop body: 8 MFMAs (builtin: hipcc places the wait states) on 4 independent accumulators, NV filler instructions of
// one kind (inline assembly on registers of their own) spread among them; 2 workgroups of 256 threads per CU = 2 waves
// per SIMD, as the general H sweep runs.  Printed: SIMD cycles per MFMA at the nominal 2.4 GHz for NV = 0, 8, 32 and the
// marginal cost per filler instruction.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench4.hip -o build/microbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

enum { OP_FMA64, OP_MUL64, OP_ADD64, OP_RCP64, OP_FREXP_MANT, OP_FREXP_EXP, OP_CVT_F64_I32, OP_AND32, OP_ANDOR32, OP_BFE32,
       OP_CNDMASK, OP_MIN3_I32, OP_LSHR32, OP_MOV32, OP_LDS_B128, OP_LDS_B64, OP_SALU, OP_SNOP, OP_LDEXP64, OP_MAX64, OP_FMA32, OP_CVT_F32_F64, OP_LOG32, OP_GLOAD_X4, OP_GLOAD_X2, OP_LDS_DMA, OP_COUNT };
const char* const OP_NAME[OP_COUNT] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_frexp_mant_f64", "v_frexp_exp_i32_f64",
                                       "v_cvt_f64_i32", "v_and_b32", "v_and_or_b32", "v_bfe_u32", "v_cndmask_b32", "v_min3_i32",
                                       "v_lshrrev_b32", "v_mov_b32", "ds_read_b128", "ds_read_b64", "s_add_u32", "s_nop 0", "v_ldexp_f64", "v_max_f64", "v_fma_f32", "v_cvt_f32_f64", "v_log_f32", "global_load_dwordx4", "global_load_dwordx2", "global_load_lds_dwordx4"};

struct MemOps {   // what the memory fillers work on: a streaming pointer per WAVE (scalar: advancing it costs no vector
  unsigned long long sbase;   // instruction, as in pass_kernel's STAGE_DMA), the lane's offset, an LDS landing zone per wave
  unsigned lane_off;
  __attribute__((address_space(3))) char* lds_dst;
};
template <int OP>
__device__ __forceinline__ void filler(double& x, double& y, uint32_t& i, uint32_t& j, uint32_t lds_addr, d2& lv, uint32_t& sreg, float& f, MemOps& mo) {
  if (OP == OP_GLOAD_X4) { __asm__ volatile("global_load_dwordx4 %0, %1, %2" : "=v"(lv) : "v"(mo.lane_off), "s"(mo.sbase)); mo.sbase += 1024; }
  if (OP == OP_GLOAD_X2) { __asm__ volatile("global_load_dwordx2 %0, %1, %2" : "=v"(x) : "v"(mo.lane_off), "s"(mo.sbase)); mo.sbase += 1024; }
  if (OP == OP_LDS_DMA) {
    __asm__ volatile("" : "+s"(mo.sbase));
    __asm__ volatile("" : "+v"(mo.lane_off));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) char*)mo.sbase + mo.lane_off, mo.lds_dst, 16, 0, 0);
    mo.sbase += 1024;
  }

  if (OP == OP_FMA64) __asm__ volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if (OP == OP_MUL64) __asm__ volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
  if (OP == OP_ADD64) __asm__ volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
  if (OP == OP_RCP64) __asm__ volatile("v_rcp_f64 %0, %0" : "+v"(x));
  if (OP == OP_FREXP_MANT) __asm__ volatile("v_frexp_mant_f64 %0, %0" : "+v"(x));
  if (OP == OP_FREXP_EXP) __asm__ volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(i) : "v"(x));
  if (OP == OP_CVT_F64_I32) __asm__ volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(i));
  if (OP == OP_AND32) __asm__ volatile("v_and_b32 %0, %0, %1" : "+v"(i) : "v"(j));
  if (OP == OP_ANDOR32) __asm__ volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(i) : "v"(j));
  if (OP == OP_BFE32) __asm__ volatile("v_bfe_u32 %0, %1, 20, 11" : "=v"(i) : "v"(j));
  if (OP == OP_CNDMASK) __asm__ volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i) : "v"(j));
  if (OP == OP_MIN3_I32) __asm__ volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(i) : "v"(j));
  if (OP == OP_LSHR32) __asm__ volatile("v_lshrrev_b32 %0, 6, %1" : "=v"(i) : "v"(j));
  if (OP == OP_MOV32) __asm__ volatile("v_mov_b32 %0, %1" : "=v"(i) : "v"(j));
  if (OP == OP_LDS_B128) __asm__ volatile("ds_read_b128 %0, %1" : "=v"(lv) : "v"(lds_addr));
  if (OP == OP_LDS_B64) __asm__ volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(lds_addr));
  if (OP == OP_SALU) __asm__ volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
  if (OP == OP_SNOP) __asm__ volatile("s_nop 0");
  if (OP == OP_LDEXP64) __asm__ volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(i));
  if (OP == OP_MAX64) __asm__ volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(y));
  if (OP == OP_FMA32) __asm__ volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(j));
  if (OP == OP_CVT_F32_F64) __asm__ volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x));
  if (OP == OP_LOG32) __asm__ volatile("v_log_f32 %0, %0" : "+v"(f));
}

template <int OP, int NV>
__global__ __launch_bounds__(256, 2) void k_mix(double* out, int iters, double a0, double b0, const char* stream, size_t stream_bytes) {
  __shared__ double tab[2048 + 4 * 128];
  for (int t = threadIdx.x; t < 2048; t += 256) tab[t] = a0 + t;
  __syncthreads();
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double x[8], y = 0.999999;
  uint32_t iv[8], j = 0x3FE12345u + threadIdx.x, sreg = 0;
  float f[8];
  d2 lv[8];
  // (a random 16-byte slot per lane, as the logarithm table's gathers have)
  const uint32_t lds_addr = (uint32_t)(uintptr_t)tab + (((threadIdx.x * 2654435761u) >> 20) & 0x3FF0u);
  for (int i = 0; i < 8; ++i) {
    x[i] = a0 + 1e-3 * i;
    iv[i] = i;
    f[i] = 1.0f + i;
    lv[i] = d2{0, 0};
  }
  // (memory fillers: every wave streams its own contiguous region of a buffer far larger than the caches, wrapping around)
  const size_t per_wave = stream_bytes / ((size_t)gridDim.x * 4);
  const unsigned long long g0v = (unsigned long long)(stream + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * per_wave);
  const unsigned long long g0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(g0v >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)g0v);
  MemOps mo;
  mo.sbase = g0;
  mo.lane_off = (threadIdx.x & 63) * 16;
  mo.lds_dst = (__attribute__((address_space(3))) char*)(uintptr_t)((uint32_t)(uintptr_t)tab + 2048 * 8 + (threadIdx.x >> 6) * 1024);
  constexpr bool MEM = OP == OP_GLOAD_X4 || OP == OP_GLOAD_X2 || OP == OP_LDS_DMA;
  for (int it = 0; it < iters; ++it) {
    if (MEM && (size_t)(mo.sbase - g0) + 64 * 1024 > per_wave) mo.sbase = g0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < (NV >= 8 ? NV / 8 : (u < NV ? 1 : 0)); ++v) filler<OP>(x[(u + v) & 7], y, iv[(u + v) & 7], j, lds_addr, lv[(u + v) & 7], sreg, f[(u + v) & 7], mo);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (OP == OP_LDS_B128 || OP == OP_LDS_B64) __asm__ volatile("s_waitcnt lgkmcnt(0)");
    if (MEM) __asm__ volatile("s_waitcnt vmcnt(4)");   // (keeps a run of loads in flight, as a prefetching sweep does)
  }
  __asm__ volatile("s_waitcnt vmcnt(0)");
  double s = sreg;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i] + iv[i] + f[i] + lv[i][0] + lv[i][1];
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
char* g_stream = nullptr;
size_t g_stream_bytes = (size_t)2 << 30;
template <int OP>
void run_op() {
  const int BPC = 2;
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / (g_iters * 8.0 * BPC); };
  const float t0 = time_ms([&] { k_mix<OP, 0><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
  const float t8 = time_ms([&] { k_mix<OP, 8><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
  const float t32 = time_ms([&] { k_mix<OP, 32><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
  printf("%-22s cycles per MFMA with 0 / 1 / 4 of them per MFMA: %6.1f %6.1f %6.1f   => +%.1f for the first, +%.1f for each further one\n",
         OP_NAME[OP], cyc(t0), cyc(t8), cyc(t32), cyc(t8) - cyc(t0), (cyc(t32) - cyc(t8)) / 3.0);
  fflush(stdout);
  if (OP == OP_GLOAD_X4 || OP == OP_GLOAD_X2 || OP == OP_LDS_DMA) {
    // memory instructions at the density the sweeps have (one or two per 8 MFMAs: far below the bandwidth limit the rows
    // above run into), so that what is measured is the instruction's own cost to the issuing SIMD
    const float t1 = time_ms([&] { k_mix<OP, 1><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
    const float t2 = time_ms([&] { k_mix<OP, 2><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
    const float t4 = time_ms([&] { k_mix<OP, 4><<<g_cu * BPC, 256>>>(g_out, g_iters, 1.0, 0.5, g_stream, g_stream_bytes); });
    printf("%-22s cycles per 8 MFMAs with 0 / 1 / 2 / 4 of them per 8 MFMAs: %6.1f %6.1f %6.1f %6.1f  => +%.1f, +%.1f, +%.1f each\n", OP_NAME[OP],
           8 * cyc(t0), 8 * cyc(t1), 8 * cyc(t2), 8 * cyc(t4), 8 * (cyc(t1) - cyc(t0)), 8 * (cyc(t2) - cyc(t0)) / 2, 8 * (cyc(t4) - cyc(t0)) / 4);
    fflush(stdout);
  }
}
template <int OP>
void run_all() {
  run_op<OP>();
  if constexpr (OP + 1 < OP_COUNT) run_all<OP + 1>();
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  g_cu = p.multiProcessorCount;
  g_iters = argc > 1 ? atoi(argv[1]) : 20000;   // (8 MFMAs each at 2 waves per SIMD: 20000 = ~9 ms per launch)
  printf("%d CUs, %d iterations of 8 MFMAs, 2 waves per SIMD\n", g_cu, g_iters);
  CK(hipMalloc(&g_out, sizeof(double) * g_cu * 2 * 256));
  CK(hipMalloc(&g_stream, g_stream_bytes + (1 << 20)));
  CK(hipMemset(g_stream, 0, g_stream_bytes + (1 << 20)));
  run_all<0>();
  return 0;
}
